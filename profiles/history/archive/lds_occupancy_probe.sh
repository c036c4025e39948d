#!/bin/bash
# config 4 (26-level stacks: 6.6 KB of LDS per wave) against padded LDS allocations: how much does it gain per resident wave?
export KERNELS=0 GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  C=$(python3 profiles/config_probe.py 4 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "$(basename $lib) : config4 4spp $C ms"
done; done
