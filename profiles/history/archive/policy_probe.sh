#!/bin/bash
# leaf-stage policy under one-wave workgroups: default policy, always plain, always dealt; then frames/launch x streams
export GPU_MAX_HW_QUEUES=8
bash profiles/variant_probe3.sh
bash profiles/variant_probe3.sh
unset SHRAY_HIP_LIB
for FPL in 1 2 3 4; do for FIF in 2 4 6; do
  python3 bench.py --no-cpu-baseline --trials 5 --frames-per-launch $FPL --frames-in-flight $FIF 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/launch $FPL streams $FIF : %.4f ms/frame  %.0f Mrays/s' % (d['ms_per_step'], d['value']))"
done; done
