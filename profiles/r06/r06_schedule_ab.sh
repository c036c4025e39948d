#!/bin/bash
# round 6: the new re-sort schedule (1, 2, 3, 4, 8, 16, then every 32nd launch) against round 3's (1, 2, 3, then every 8th = SHRAY_DISPATCH_PERIOD=8
# -- with period 8 the doubling start adds only launch 4): a lone frame of the orbit, config 4, and four ranks' tile sets on one GPU (gloo)
for rep in 1 2; do
for p in 8 32; do
  echo -n "period $p: "
  SHRAY_DISPATCH_PERIOD=$p bash profiles/r05/r05_quick_ab.sh || exit 1
  SHRAY_DISPATCH_PERIOD=$p timeout -k 10 400 python profiles/run_configs.py ab_sched 0 2,4 2>/dev/null | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('   back to back', d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', flush=True)" || exit 1
done; done
for p in 8 32; do
  echo -n "period $p, four ranks on one GPU (gloo): "
  SHRAY_DISPATCH_PERIOD=$p SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_TRANSPORT=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 16 --warmup 4 --trials 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('value'), d.get('ms_per_step'), (d.get('stages') or {}), flush=True)" || exit 1
done
