#!/bin/bash
# round 6: one variant against the shipped library -- the variant's parity suites first, then the 200-step form twice each (interleaved),
# then the other configurations.   bash profiles/r06/r06_variant_ab.sh <variant> [configs, e.g. 2,4]
v=$1; only=${2:-2,3,4,5}
lib=shader-ray_amd/_variants/libshray_hip_$v.so
mkdir -p gpurun_out
SHRAY_HIP_LIB=$lib timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/ab_${v}_parity.log 2>&1 || { tail -30 gpurun_out/ab_${v}_parity.log; exit 1; }
tail -1 gpurun_out/ab_${v}_parity.log
bash profiles/r05/r05_quick_ab.sh $v && bash profiles/r05/r05_quick_ab.sh $v || exit 1
for l in "" $lib; do
  name=${l:-shipped}; name=${name##*/}
  SHRAY_HIP_LIB=$l timeout -k 10 400 python profiles/run_configs.py ab_$v 0 $only 2>/dev/null | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('$name'.ljust(30), d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', d['mrays_per_s'], 'Mrays/s', flush=True)" || exit 1
done
