#!/bin/bash
# multi-sample frames, one launch at a time, per build under _variants: gold 16 spp, plaster 8 spp (config 3's work), the
# 1M-triangle scene at 4 spp (config 4), gold 4 spp
export GPU_MAX_HW_QUEUES=8
for pass in 1 2; do
for lib in shader-ray_amd/_variants/*.so; do
  export SHRAY_HIP_LIB=$PWD/$lib
  A=$(python3 profiles/tail_probe.py 2 16 8 2>&1 | grep "1 stream" | awk '{print $7}')
  B=$(python3 profiles/tail_probe.py 3 8 8 2>&1 | grep "1 stream" | awk '{print $7}')
  C=$(python3 profiles/tail_probe.py 4 4 8 2>&1 | grep "1 stream" | awk '{print $7}')
  D=$(python3 profiles/tail_probe.py 2 4 8 2>&1 | grep "1 stream" | awk '{print $7}')
  echo "$(basename $lib) : gold 16 spp $A ms; plaster 8 spp $B ms; 1M triangles 4 spp $C ms; gold 4 spp $D ms"
done; done
