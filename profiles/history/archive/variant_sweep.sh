#!/bin/bash
# usage: bash profiles/variant_sweep.sh <config> <spp> : config_probe.py under the shipped library and every _variants/ build
CFG=${1:-4}; SPP=${2:-1}
echo "== shipped"; python3 profiles/config_probe.py $CFG $SPP || exit 1
for lib in shader-ray_amd/_variants/*.so; do
  echo "== $(basename $lib)"; KERNELS=${KERNELS:-0} SHRAY_HIP_LIB=$PWD/$lib python3 profiles/config_probe.py $CFG $SPP | grep kernel || exit 1
done
