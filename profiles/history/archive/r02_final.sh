#!/bin/bash
# Round-2 evidence run (one gpurun call): GPU tests, smoke, bench lines, rocprofv3 passes of the default bench command
# (-> profiles/r02/pmc_headline.json via profiles/make_pmc_json.py), of the one-frame-at-a-time form and of config 4.
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests exit $?"; tail -2 gpurun_out/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 bash profiles/run_profile.sh r02_default 50 > gpurun_out/profile_default.txt 2>&1; grep -c "exit 0" gpurun_out/profile_default.txt
timeout -k 10 900 bash profiles/run_profile.sh r02_single 50 "--frames-per-launch 1 --frames-in-flight 1" > gpurun_out/profile_single.txt 2>&1; grep -c "exit 0" gpurun_out/profile_single.txt
KERNELS=0 ARGS="4 4" timeout -k 10 600 bash profiles/probe_counters.sh r02_config4 profiles/config_probe.py > gpurun_out/profile_config4.txt 2>&1
python profiles/make_pmc_json.py gpurun_out/prof_r02_default/summary.txt profiles/r02/pmc_headline.json > /dev/null && cp profiles/r02/pmc_headline.json gpurun_out/pmc_headline.json
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-1500
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_steps20.json 2>/dev/null; tail -1 gpurun_out/bench_steps20.json | cut -c1-200
timeout -k 10 300 python bench.py --frames-per-launch 1 --frames-in-flight 1 --no-cpu-baseline > gpurun_out/bench_one_at_a_time.json 2>/dev/null; tail -1 gpurun_out/bench_one_at_a_time.json | cut -c1-200
timeout -k 10 400 python profiles/run_configs.py r02 0 > gpurun_out/configs.log 2>&1; grep -E "^\{" gpurun_out/configs.log | cut -c1-220
