#!/bin/bash
# instruction-cache counters of the default bench command
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_icache; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # before rocprofv3: its preloaded library initialises HIP ahead of bench.py
cd /tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/pmc1" -- python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --frames-in-flight 1 > "$OUT/pmc1.log" 2>&1
cd "$REPO"; python3 profiles/summarize_profile.py "$OUT" | grep -A8 "batch_kernel" | head -12
