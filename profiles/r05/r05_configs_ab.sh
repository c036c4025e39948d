#!/bin/bash
# round 5: configs only (no bench) for the shipped library and the variants named:  bash profiles/r05/r05_configs_ab.sh 3,4,5 v1 v2 ...
CONFIGS=$1; shift
LIBS=("")
for v in "$@"; do LIBS+=("shader-ray_amd/_variants/libshray_hip_$v.so"); done
for lib in "${LIBS[@]}"; do
  name=${lib:-shipped}; name=${name##*/}
  SHRAY_HIP_LIB=$lib timeout -k 10 600 python profiles/run_configs.py ab_$name 0 $CONFIGS 2>&1 | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('$name'.ljust(30), d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', d['mrays_per_s'], 'Mrays/s', flush=True)"
done
