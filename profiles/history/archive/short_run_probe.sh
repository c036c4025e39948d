#!/bin/bash
# the driver's short form (--steps 20 --warmup 5) and the default, per build under _variants
export GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  A=$(python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  B=$(python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  C=$(python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch 4 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "$(basename $lib) : --steps 20: $A ms/step; default (200 steps): $B; --steps 20, 4 frames per launch: $C"
done; done
