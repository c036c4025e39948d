#!/bin/bash
# kernel trace + issue counters of kernels 0 and 2 on config 4 (1 spp) and config 2
export KERNELS=0,2
ARGS="4 1" bash profiles/probe_counters.sh pool_c4 profiles/config_probe.py > gpurun_out/pool_c4.txt 2>&1
ARGS="2 1" bash profiles/probe_counters.sh pool_c2 profiles/config_probe.py > gpurun_out/pool_c2.txt 2>&1
grep -E "kernel<|SQ_INSTS_VALU|SQ_THREAD_CYCLES|SQ_WAIT|SQ_WAVE_CYCLES|SQ_BUSY|kernel [02]:|calls" gpurun_out/pool_c4.txt | cut -c1-160
grep -E "kernel<|SQ_INSTS_VALU|SQ_THREAD_CYCLES|SQ_WAIT|SQ_WAVE_CYCLES|SQ_BUSY|kernel [02]:|calls" gpurun_out/pool_c2.txt | cut -c1-160
