#!/bin/bash
for FIF in 2 3 4; do
  python3 bench.py --no-cpu-baseline --trials 5 --frames-per-launch 2 --frames-in-flight $FIF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps 200: frames/launch 2 streams $FIF : %.4f ms/frame  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
done
