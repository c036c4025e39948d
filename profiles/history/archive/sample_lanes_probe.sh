#!/bin/bash
# lanes per pixel capped at 64 / 32 / 16 / 4 (builds under _variants): plaster 64 spp (config 3), gold 16 spp, 1M triangles 4 spp
export GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  A=$(python3 profiles/tail_probe.py 3 64 4 2>&1 | grep "1 stream" | awk '{print $7}')
  B=$(python3 profiles/tail_probe.py 2 16 8 2>&1 | grep "1 stream" | awk '{print $7}')
  C=$(python3 profiles/tail_probe.py 4 4 8 2>&1 | grep "1 stream" | awk '{print $7}')
  echo "$(basename $lib) : plaster 64 spp $A ms; gold 16 spp $B ms; 1M triangles 4 spp $C ms"
done; done
