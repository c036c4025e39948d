#!/bin/bash
# round 5 A/B: bench (throughput form, 200 and 20 steps, + one frame at a time) and configs 3/4/5 of the shipped library
# and of the libraries under _variants/ named on the command line (all of them when none is named); same box, same call
#   bash profiles/r05/r05_ab.sh [-c configs] [variant ...]     e.g. -c 4  multi7 multi6
CONFIGS=""
if [ "$1" = "-c" ]; then CONFIGS=$2; shift 2; fi
LIBS=("")
if [ $# -gt 0 ]; then for v in "$@"; do LIBS+=("shader-ray_amd/_variants/libshray_hip_$v.so"); done
else for f in shader-ray_amd/_variants/*.so; do [ -e "$f" ] && LIBS+=("$f"); done; fi
for lib in "${LIBS[@]}"; do
  name=${lib:-shipped}; name=${name##*/}
  for steps in 200 20; do
    SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name'.ljust(30), 'steps $steps', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms', flush=True)"
  done
  if [ -n "$CONFIGS" ]; then
    SHRAY_HIP_LIB=$lib timeout -k 10 600 python profiles/run_configs.py ab_$name 0 $CONFIGS 2>&1 | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('$name'.ljust(30), d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', d['mrays_per_s'], 'Mrays/s', flush=True)"
  fi
done
