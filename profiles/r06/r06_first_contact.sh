#!/bin/bash
# round 6, first contact of the hand-scheduled node stage (visit_asm.h): the smoke render, the parity suite's core (under a short timeout: a
# kernel that never ends must not hold the box), then the throughput form shipped / -DSHRAY_ASM_VISIT=0.
set -u
mkdir -p gpurun_out
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/r06_smoke.log; exit 1; }
tail -2 gpurun_out/r06_smoke.log
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_driver.py -x -q -m gpu > gpurun_out/r06_parity.log 2>&1 || { echo "parity failed"; tail -40 gpurun_out/r06_parity.log; exit 1; }
tail -3 gpurun_out/r06_parity.log
bash profiles/r05/r05_quick_ab.sh noasm 2>&1 | tee gpurun_out/r06_visit_asm_ab.txt
