#!/bin/bash
# the driver's short form (--steps 20 --warmup 5): which frame-loop shape loses least to pipeline fill and drain
for FPL in 1 2 4 5 10; do for FIF in 2 3 4; do
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $FPL --frames-in-flight $FIF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 20: frames/launch $FPL streams $FIF : %.4f ms/frame  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
done; done
