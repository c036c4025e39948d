#!/bin/bash
# (a pass with TA_BUFFER_WAVEFRONTS_sum / TA_ADDR_STALLED_BY_TC_CYCLES_sum made rocprofv3 abort and the run hang: left out)
# is the vector-memory front end (TA / TCP / TD) a co-bottleneck of the headline kernel?  busy counters of the default bench command
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_ta; mkdir -p "$OUT"
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
cd /tmp
BENCH="python3 $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --trials 2"
i=0
for GROUP in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TD_TD_BUSY_sum TD_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1
  echo "pass $i ($GROUP) exit $?"
done
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/prof_ta/pmc*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "batch_kernel<true, true, false>" not in row["Kernel_Name"]:
            continue
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:42s} n={n:4d} avg {v / n:16.1f}")
PY
