#!/bin/bash
# round 6 (R6.10): launches of four whole frames in a dispatch order too?  the driver's form and the 200-step form; bulk classes 1, 2, 4, all
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps $1 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2'.ljust(44), 'steps $1', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; verified', d['frames_verified'], d['frames_mismatched'], flush=True)" || exit 1
}
for rep in 1 2; do
  run 20 "row-major (shipped)"
  for bulk in 1 2 4 99; do
    SHRAY_DISPATCH_BATCHES=1 SHRAY_DISPATCH_BULK=$bulk run 20 "ordered batches, bulk class $bulk"
  done
done
run 200 "row-major (shipped)"
for bulk in 1 4 99; do
  SHRAY_DISPATCH_BATCHES=1 SHRAY_DISPATCH_BULK=$bulk run 200 "ordered batches, bulk class $bulk"
done
