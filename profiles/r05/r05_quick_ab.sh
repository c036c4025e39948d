#!/bin/bash
# round 5: the 200-step throughput form only, for the shipped library and the variants named (bash profiles/r05/r05_quick_ab.sh v1 v2 ...)
LIBS=("")
for v in "$@"; do LIBS+=("shader-ray_amd/_variants/libshray_hip_$v.so"); done
for lib in "${LIBS[@]}"; do
  name=${lib:-shipped}; name=${name##*/}
  SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name'.ljust(30), 'steps 200', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms', flush=True)"
done
