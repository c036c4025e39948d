#!/bin/bash
# runtime scheduling knobs of the stack kernel (SceneView::deal_max_parked, keep_walking): config 2 pipelined and one
# frame at a time (bench.py), config 4 at 4 spp
export KERNELS=0
for DEAL in ${DEALS:-0 4 8 16 32}; do for KEEP in ${KEEPS:-28 40}; do
  export SHRAY_DEAL_MAX_PARKED=$DEAL SHRAY_KEEP_WALKING=$KEEP
  A=$(python3 bench.py --no-cpu-baseline --trials 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  B=$(python3 bench.py --no-cpu-baseline --trials 3 --frames-in-flight 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
  C=$(python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "deal<=$DEAL keep $KEEP : config2 pipelined $A ms, one frame $B ms; config4 4spp $C ms"
done; done
