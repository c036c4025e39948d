#!/bin/bash
# A/B of the pair traversal (kernel id 3: both children of a node per turn) against the shipped selection (kernel id 0),
# same library, same box: the BASELINE configurations one launch at a time, and the bench loop (throughput + one frame at a time)
mkdir -p gpurun_out
for k in ${KERNELS:-0 3}; do
  timeout -k 10 400 python profiles/run_configs.py r03 $k 2>/dev/null | grep -v "^1M-triangle" | python -c "
import json,sys
for line in sys.stdin:
    r=json.loads(line); print('kernel $k', r['config'][:44].ljust(44), r['ms_per_frame'], 'ms', r['mrays_per_s'], 'Mrays/s')"
done
for k in ${KERNELS:-0 3}; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --kernel $k 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel $k bench: throughput form', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one frame at a time', d['latency']['ms'], 'ms')"
done
