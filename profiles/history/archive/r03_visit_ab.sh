#!/bin/bash
# A/B of the node visit's bookkeeping (SHRAY_TIED_TOP, SHRAY_PARK_RAW; wave_traversal.h): the shipped library against
# the variants under _variants/ -- the BASELINE configurations one launch at a time, then the bench loop twice each
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -x -q > gpurun_out/visit_ab_tests.log 2>&1; echo "parity exit $?"; tail -2 gpurun_out/visit_ab_tests.log; fi
for lib in "" $(ls shader-ray_amd/_variants/*.so); do
  name=$(basename "${lib:-shipped}")
  SHRAY_HIP_LIB=$lib timeout -k 10 400 python profiles/run_configs.py tmp_ab 0 2>/dev/null | grep -v "^1M-triangle" | python -c "
import json,sys
for line in sys.stdin:
    r=json.loads(line); print('$name'.ljust(28), r['config'][:40].ljust(40), r['ms_per_frame'], 'ms')"
done
for rep in 1 2; do
for lib in "" $(ls shader-ray_amd/_variants/*.so); do
  name=$(basename "${lib:-shipped}")
  SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name'.ljust(28), 'throughput', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one at a time', d['latency']['ms'], 'ms')"
done
done
rm -f profiles/tmp_ab_configs.json
