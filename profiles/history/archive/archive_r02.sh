#!/bin/bash
# copies what profiles/r02_final.sh left under gpurun_out/ into profiles/r02/ (run in the container after the gpurun call)
set -e
cp gpurun_out/pmc_headline.json profiles/r02/pmc_headline.json
cp gpurun_out/prof_r02_default/summary.txt profiles/r02/default_bench_command_summary.txt
cp gpurun_out/prof_r02_single/summary.txt profiles/r02/one_frame_at_a_time_summary.txt
cp gpurun_out/prof_r02_config4/summary.txt profiles/r02/config4_4spp_summary.txt
for d in r02_default r02_single; do f=$(ls -t gpurun_out/prof_$d/trace/*/*kernel_stats.csv | head -1); cp "$f" profiles/r02/${d}_kernel_stats.csv; done
cp gpurun_out/bench_default.json profiles/r02/bench_default.json
cp gpurun_out/bench_steps20.json profiles/r02/bench_steps20.json
cp gpurun_out/bench_one_at_a_time.json profiles/r02/bench_one_frame_at_a_time.json
cp gpurun_out/r02_configs_k0.json profiles/r02/configs.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, "profiles")
from buildhash import kernel_source_hash
d = json.load(open("profiles/r02/pmc_headline.json"))
print("counter file matches the current kernel sources:", d["build_hash"] == kernel_source_hash(), "| kernel trace avg us", d["kernel_trace_avg_us"])
b = json.loads(open("profiles/r02/bench_default.json").read().strip().splitlines()[-1])
print("bench:", b["value"], "Mrays/s, roofline frac", b["roofline"]["frac"], "lane_util", b["roofline"]["lane_util"], "hbm_frac", b["roofline"]["hbm_frac"])
PY
