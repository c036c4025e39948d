set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-400
timeout -k 10 300 python bench.py --frames-in-flight 1 --no-cpu-baseline > gpurun_out/bench_one_at_a_time.json 2>/dev/null; tail -1 gpurun_out/bench_one_at_a_time.json | cut -c1-300
timeout -k 10 600 bash profiles/run_profile.sh r01_default 50 > gpurun_out/profile_default.txt 2>&1; tail -5 gpurun_out/profile_default.txt
timeout -k 10 600 bash profiles/run_profile.sh r01_single 50 "--frames-in-flight 1" > gpurun_out/profile_single.txt 2>&1; grep "trace_stack_batch_kernel<false, true, true>" gpurun_out/profile_single.txt | head -3
