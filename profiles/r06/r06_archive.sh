#!/bin/bash
# copies what profiles/r06/r06_evidence.sh left under gpurun_out/ to the names profiles/README.md lists
R=profiles/r06
cp gpurun_out/prof_r06_default/summary.txt $R/default_bench_command_summary.txt
for t in config3 config4 config5 lone_frame; do [ -e gpurun_out/prof_r04_$t/summary.txt ] && cp gpurun_out/prof_r04_$t/summary.txt $R/${t}_summary.txt; done
f=$(ls -t gpurun_out/prof_r06_default/trace/runc/*_kernel_stats.csv | head -1); [ -n "$f" ] && cp "$f" $R/r06_default_kernel_stats.csv
cp gpurun_out/bench_default.json gpurun_out/bench_steps20.json gpurun_out/bench_same_view.json gpurun_out/bench_dist1.json gpurun_out/bench_gloo4.json \
   gpurun_out/vector_cache_probe.json $R/
cp gpurun_out/r06_configs_k0.json $R/configs.json
cp gpurun_out/pmc_headline.json $R/pmc_headline.json
