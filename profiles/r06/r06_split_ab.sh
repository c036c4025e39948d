#!/bin/bash
# round 6: the driver's form (--steps 20 --warmup 5) under other splits of the twenty frames into launches
for rep in 1 2; do
for f in 4 5 10 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames per launch $f', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s', d.get('trial_ms'), flush=True)" || exit 1
done; done
