#!/bin/bash
# A/B of the learnt dispatch order (heaviest patches first; capi.hip: DispatchOrder) through the run-time switch
# SHRAY_DISPATCH_ORDER=0|1, same library, same box: a rank share of 20 / 200 frames at N = 8, 4, 2 (compute side of
# bench.py --gpus N) and the N = 1 bench loop in the driver form and the default one
for on in 0 1; do
  export SHRAY_DISPATCH_ORDER=$on
  for n in 8 4 2; do for k in 20 200; do
    echo "== order $on N $n K $k"; timeout -k 10 300 python profiles/rank_share_shapes.py $n $k 2>/dev/null | grep "N = 1 loop\| $((4*n)) frames per launch x 4\| $((2*n)) frames per launch x 4"
  done; done
  for rep in 1 2; do for form in "--steps 20 --warmup 5" ""; do
    timeout -k 10 300 python bench.py --no-cpu-baseline $form 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order $on', '${form:-default (200 steps)}'.ljust(24), d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms')"
  done; done
done
