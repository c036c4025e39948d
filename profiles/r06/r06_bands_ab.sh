for rep in 1 2; do
for l in "" shader-ray_amd/_variants/libshray_hip_bands.so; do
  name=${l:-shipped}; name=${name##*/}
  SHRAY_DISPATCH_ORDER=0 SHRAY_HIP_LIB=$l timeout -k 10 400 python profiles/run_configs.py ab_bands 0 2,3,4 2>/dev/null | grep '"config"' | python -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('$name'.ljust(30), 'order off', d['config'][:40].ljust(42), d['ms_per_frame'], 'ms', flush=True)" || exit 1
done; done
