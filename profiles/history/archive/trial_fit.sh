#!/bin/bash
# time of one timed region against its number of steps: T(K) = a K + c; c is what a short run loses to fill and drain
export GPU_MAX_HW_QUEUES=8
for K in 2 4 8 16 24 40 80; do
  python3 bench.py --no-cpu-baseline --steps $K --warmup 5 2>/dev/null | K=$K python3 -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); K = int(os.environ['K'])
print('steps %3d: %.4f ms/step, region %.3f ms' % (K, d['ms_per_step'], d['ms_per_step'] * K))"
done
