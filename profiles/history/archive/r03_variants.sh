#!/bin/bash
# bench (throughput form + one frame at a time) of the shipped library and of every library under _variants/
for lib in "" $(ls shader-ray_amd/_variants/*.so | grep -v ldspad); do
  SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-shipped}'.split('/')[-1].ljust(34), 'throughput', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms')"
done
