#!/bin/bash
# hardware counters of one-frame-at-a-time launches: the shipped dealt instance (kernel 0) against the pair traversal (3)
mkdir -p gpurun_out; export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
REPO=$(pwd); cd /tmp
for k in 0 3; do
  for GROUP in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
               "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $GROUP | cut -c1-12 | tr ' ' '_')
    rocprofv3 --pmc $GROUP --output-format csv -d $REPO/gpurun_out/pairpmc_k${k}_$tag -- python3 $REPO/bench.py --steps 40 --warmup 5 --trials 1 --no-cpu-baseline --same-view --frames-in-flight 1 --frames-per-launch 1 --kernel $k > $REPO/gpurun_out/pairpmc_k${k}_$tag.log 2>&1
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for k in (0, 3):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pairpmc_k{k}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "batch" in name and "tally" not in name:
                acc[(name[:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (name, c), v in sorted(acc.items()):
        print(k, name, c, f"n={len(v)} avg {sum(v)/len(v):.4g}")
PY
