#!/bin/bash
# every library build (shipped + _variants/): config 2 pipelined and one frame at a time (bench.py), config 4 at 4 spp
export KERNELS=${KERNELS:-0}
for lib in shader-ray_amd/libshray_hip.so shader-ray_amd/_variants/*.so; do
  echo "== $lib"
  SHRAY_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --trials 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  config 2 pipelined  %.4f ms  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
  SHRAY_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --trials 3 --frames-in-flight 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  config 2 one frame  %.4f ms  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
  SHRAY_HIP_LIB=$PWD/$lib python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel [0-9]:" | sed 's/^/  config 4 4spp /'
done
