#!/bin/bash
# rocprofv3 passes (kernel trace, then counter groups, one per pass) of one configuration rendered one launch at a time:
#   bash profiles/r04/r04_config_profile.sh <tag> [VAR=value ...]      (profiles/config_probe.py reads SCENE WIDTH HEIGHT SPP MATERIAL)
# e.g.  bash profiles/r04/r04_config_profile.sh config3 SPP=64 MATERIAL=6
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r04_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 REPS=${REPS:-8}
cd /tmp
CMD="python3 $REPO/profiles/config_probe.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1; echo "trace pass exit $?"; tail -1 "$OUT/trace.log"
i=0
for GROUP in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TA_BUSY_avr" "TD_TD_BUSY_sum TD_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1; echo "pmc pass $i exit $?"
done
cd "$REPO"
python3 profiles/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -A28 "PMC counters" "$OUT/summary.txt" | head -40
