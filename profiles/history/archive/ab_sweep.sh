#!/bin/bash
# usage: bash profiles/ab_sweep.sh [config] : ab_probe.py under the shipped library and every _variants/ build, twice
CFG=${1:-2}
for round in 1 2; do
  python3 profiles/ab_probe.py $CFG 2>&1 | grep "one at a time" || exit 1
  for lib in shader-ray_amd/_variants/*.so; do SHRAY_HIP_LIB=$PWD/$lib python3 profiles/ab_probe.py $CFG 2>&1 | grep "one at a time" || exit 1; done
done
