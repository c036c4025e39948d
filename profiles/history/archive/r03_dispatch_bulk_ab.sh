#!/bin/bash
# which costs keep their row-major order (SHRAY_DISPATCH_BULK = class from which on patches count as the bulk; 15 = every
# class sorted, 11 = costs below a quarter of the largest are the bulk, 7 = below a half)
for bulk in 15 13 11 9 7; do
  export SHRAY_DISPATCH_BULK=$bulk
  for n in 8 4; do for k in 20 200; do
    echo "== bulk $bulk N $n K $k"; timeout -k 10 300 python profiles/rank_share_shapes.py $n $k 2>/dev/null | grep "N = 1 loop\|$((4*n)) frames per launch x 4\| $((2*n)) frames per launch x 4"
  done; done
  for form in "--steps 20 --warmup 5" ""; do
    timeout -k 10 300 python bench.py --no-cpu-baseline $form 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bulk $bulk', '${form:-default (200 steps)}'.ljust(24), d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms')"
  done
done
