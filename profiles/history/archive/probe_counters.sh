#!/bin/bash
# usage: bash profiles/probe_counters.sh <tag> <python script> : kernel trace + cache/issue counters for any driver script
set -u
# ARGS="4 1" passes arguments to the script
TAG=$1; SCRIPT=$2
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # before rocprofv3: its preloaded library initialises HIP ahead of bench.py
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/$SCRIPT ${ARGS:-} > "$OUT/trace.log" 2>&1
i=0
for GROUP in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc$i" -- python3 $REPO/$SCRIPT ${ARGS:-} > "$OUT/pmc$i.log" 2>&1
done
cd "$REPO"; python3 profiles/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt" | cut -c1-150; tail -2 "$OUT/trace.log"
