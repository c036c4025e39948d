#!/bin/bash
# round 5: the throughput form with glazed plaster at 1 spp (the one-sample diffuse / shadow-ray instances) for the shipped library and variants
LIBS=("")
for v in "$@"; do LIBS+=("shader-ray_amd/_variants/libshray_hip_$v.so"); done
for lib in "${LIBS[@]}"; do
  name=${lib:-shipped}; name=${name##*/}
  SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 5 --material 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name'.ljust(30), 'plaster 1 spp, steps 200', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms', flush=True)"
done
