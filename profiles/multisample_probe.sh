#!/bin/bash
# the multi-sample gold instance (config 5's per-pixel work: bunny, gold, 16 spp at 1080p) per build under _variants
export KERNELS=0 GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  C=$(python3 profiles/config_probe.py 2 16 2>&1 | grep "kernel 0:" | awk '{print $3}')
  D=$(python3 profiles/config_probe.py 2 4 2>&1 | grep "kernel 0:" | awk '{print $3}')
  echo "$(basename $lib) : bunny gold 16 spp $C ms, 4 spp $D ms"
done; done
