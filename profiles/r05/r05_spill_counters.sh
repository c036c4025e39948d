#!/bin/bash
# round 5: what a configuration writes to and fetches from HBM and how many vector-memory instructions it issues, one launch at a
# time (three rocprofv3 --pmc passes of profiles/config_probe.py, no tracing beside them):
#   bash profiles/r05/r05_spill_counters.sh <tag> [VAR=value ...]     (SCENE WIDTH HEIGHT SPP MATERIAL, SHRAY_HIP_LIB for a variant)
set -u
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
[ -n "${SHRAY_HIP_LIB:-}" ] && export SHRAY_HIP_LIB=$(realpath "$SHRAY_HIP_LIB")   # (the passes run from /tmp)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r05_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8 REPS=${REPS:-4}
cd /tmp
CMD="python3 $REPO/profiles/config_probe.py"
i=0
for GROUP in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $GROUP --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1; echo "pmc pass $i exit $?"; tail -1 "$OUT/pmc$i.log"
done
cd "$REPO"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sums, n = collections.defaultdict(float), collections.defaultdict(int)
for path in glob.glob(os.path.join(out, "pmc*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        if "trace_stack_batch" in r["Kernel_Name"]:
            sums[(r["Kernel_Name"][:70], r["Counter_Name"])] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:70], r["Counter_Name"])] += 1
for (k, c), v in sorted(sums.items()):
    print(f"{tag:18s} {k:70s} {c:22s} n={n[(k, c)]:3d} avg {v / n[(k, c)]:.1f}")
PY
