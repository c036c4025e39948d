#!/bin/bash
export KERNELS=0
for lib in shader-ray_amd/_variants/*.so; do
  for KEEP in 0; do
  export SHRAY_KEEP_WALKING=$KEEP
  A=$(SHRAY_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --trials 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  B=$(SHRAY_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --trials 3 --frames-in-flight 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  C=$(SHRAY_HIP_LIB=$PWD/$lib python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "$(basename $lib) keep $KEEP : config2 pipelined $A ms, one frame $B ms; config4 4spp $C ms"
  done
done
