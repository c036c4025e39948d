#!/bin/bash
# the N = 1 frame loop in its forms: frames per launch x streams
for FPL in 1 2 4; do for FIF in 1 2; do
  python3 bench.py --no-cpu-baseline --trials 3 --frames-per-launch $FPL --frames-in-flight $FIF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/launch $FPL streams $FIF : %.4f ms/frame  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
done; done
KERNELS=0 python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0"
python3 profiles/run_configs.py r02 0 3,5 2>&1 | grep -E "^\{" | cut -c1-200
