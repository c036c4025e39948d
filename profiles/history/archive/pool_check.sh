#!/bin/bash
# pool kernel (id 2): parity tests first, then A/B timings against the stack kernel on configs 2 and 4
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; rc=$?; echo "gpu tests exit $rc"; tail -5 gpurun_out/gputests.log
[ $rc -eq 0 ] || exit 1
for K in 0 2; do
  timeout -k 10 300 python bench.py --kernel $K --no-cpu-baseline --trials 5 > gpurun_out/bench_k$K.json 2>/dev/null && tail -1 gpurun_out/bench_k$K.json | cut -c1-200
  timeout -k 10 300 python bench.py --kernel $K --no-cpu-baseline --trials 5 --frames-in-flight 1 > gpurun_out/bench_k${K}_single.json 2>/dev/null && tail -1 gpurun_out/bench_k${K}_single.json | cut -c1-200
  timeout -k 10 400 python profiles/run_configs.py pool $K 2,3,4 2>&1 | grep -v "^\[" | cut -c1-250
done
