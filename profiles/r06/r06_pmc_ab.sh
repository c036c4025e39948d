#!/bin/bash
# round 6: instruction counters (passes 1, 2 of run_profile.sh) + kernel trace of the driver-form bench for the shipped library and the variants named
set -u
LIBS=("")
for v in "$@"; do LIBS+=("shader-ray_amd/_variants/libshray_hip_$v.so"); done
for lib in "${LIBS[@]}"; do
  name=${lib:-shipped}; name=${name##*/}; name=${name%.so}
  SHRAY_HIP_LIB=$lib SHRAY_PROFILE_PASSES="1 2" bash profiles/run_profile.sh r06_$name 20 > gpurun_out/r06_pmc_$name.txt 2>&1
  grep -A40 "PMC counters" gpurun_out/r06_pmc_$name.txt | grep -B1 -A22 "dense_kernel" | head -30
done
