#!/bin/bash
# experiment builds under shader-ray_amd/_variants: parity subset with the in-tree build, then timings of all variants
set -e
export GPU_MAX_HW_QUEUES=8
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -m gpu -x -q 2>&1 | tail -2
bash profiles/variant_probe3.sh
bash profiles/variant_probe3.sh
