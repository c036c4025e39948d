#!/bin/bash
# every _variants/ build: config 2 throughput (bench default), one frame, config 3 at 8 spp, config 4 at 4 spp
export KERNELS=0
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  A=$(python3 bench.py --no-cpu-baseline --trials 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  B=$(python3 bench.py --no-cpu-baseline --trials 3 --frames-in-flight 1 --frames-per-launch 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  C=$(python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  D=$(python3 profiles/config_probe.py 3 8 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "$(basename $lib) : config2 throughput $A ms, one frame $B ms; config4 4spp $C ms; plaster 8spp $D ms"
done
