#!/bin/bash
# round 6: independent launches over 2 ... 8 HIP streams (four frames per launch), the driver's form and the 200-step form
run() {
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps $1 --warmup 5 --frames-in-flight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $2', 'steps $1', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s', d['config'].get('streams'), flush=True)" || exit 1
}
for rep in 1 2; do for f in 4 2 3 5 6 8; do run 20 $f; done; done
for f in 4 3 6 8; do run 200 $f; done
