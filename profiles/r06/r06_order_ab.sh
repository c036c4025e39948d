#!/bin/bash
# round 6: is the heaviest-first dispatch order still worth it for a lone frame?  (order on / off, the bench's latency form and run_configs' back-to-back form)
for rep in 1 2; do
for o in 1 0; do
  echo "SHRAY_DISPATCH_ORDER=$o"
  SHRAY_DISPATCH_ORDER=$o bash profiles/r05/r05_quick_ab.sh || exit 1
  SHRAY_DISPATCH_ORDER=$o timeout -k 10 400 python profiles/run_configs.py ab_order 0 2,4 2>/dev/null | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('  back to back'.ljust(30), d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', flush=True)" || exit 1
done; done
