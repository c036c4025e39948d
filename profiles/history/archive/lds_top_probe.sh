#!/bin/bash
# A/B of the LDS top-of-tree stage (make variant HIP_EXTRA=-DSHRAY_LDS_TOP=n): parity first, then timings
for V in top0 top31 top127; do
  export SHRAY_HIP_LIB=$PWD/shader-ray_amd/_variants/libshray_hip_$V.so
  T=$(timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or bunny_full or million or config1 or hand_built" 2>&1 | tail -1)
  A=$(python3 bench.py --no-cpu-baseline --trials 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  B=$(python3 bench.py --no-cpu-baseline --trials 3 --frames-in-flight 1 --frames-per-launch 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  C=$(KERNELS=0 python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "$V : parity [$T]; config 2 pipelined $A ms, one frame $B ms; config 4 4spp $C ms"
done
