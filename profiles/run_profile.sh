#!/bin/bash
# Profiles bench.py on the GPU box (run via gpurun from the repo root):
#   pass 1: kernel trace + stats  -> per-kernel average duration
#   pass 2..: PMC counters, one group per pass (no tracing domains alongside --pmc)
# Outputs land in gpurun_out/prof_<tag>/; copy the summaries worth keeping into profiles/.
set -u
TAG=${1:-r01}
STEPS=${2:-50}
EXTRA=${3:-}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8   # before rocprofv3: its preloaded library initialises HIP ahead of bench.py
cd /tmp
BENCH="python3 $REPO/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-live-counters $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace pass exit $?"
i=0
# SHRAY_PROFILE_PASSES="1 2 7 8": only those counter passes (default: all)
for GROUP in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TD_BUSY_avr" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  if [ -n "${SHRAY_PROFILE_PASSES:-}" ] && ! echo " $SHRAY_PROFILE_PASSES " | grep -q " $i "; then continue; fi
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
  echo "pmc pass $i ($GROUP) exit $?"
done
cd "$REPO"
python3 profiles/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
