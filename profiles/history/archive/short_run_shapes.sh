#!/bin/bash
# the driver's short form (--steps 20 --warmup 5) with other launch shapes: frames per launch x streams
export GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for shape in "2 4" "2 2" "2 3" "3 2" "4 2" "1 4" "1 8" "2 6" "5 2" "4 3"; do
  set -- $shape
  A=$(python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $1 --frames-in-flight $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  echo "frames per launch $1 x streams $2: --steps 20: $A ms/step"
done; done
