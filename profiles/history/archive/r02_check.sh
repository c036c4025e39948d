#!/bin/bash
# one GPU-box call: GPU tests, smoke, the default bench line (driver's --steps 20 form too), one frame at a time,
# and the rocprofv3 passes the bench's roofline object reads (profiles/r02/pmc_headline.json)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests exit $?"; tail -4 gpurun_out/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-600
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_steps20.json 2>/dev/null; tail -1 gpurun_out/bench_steps20.json | cut -c1-300
timeout -k 10 300 python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/bench_one_at_a_time.json 2>/dev/null; tail -1 gpurun_out/bench_one_at_a_time.json | cut -c1-300
timeout -k 10 900 bash profiles/run_profile.sh r02_default 50 > gpurun_out/profile_default.txt 2>&1; tail -40 gpurun_out/profile_default.txt
