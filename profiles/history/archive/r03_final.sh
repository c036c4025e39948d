#!/bin/bash
# round-end evidence run: tests, smoke, bench lines, the rocprofv3 passes behind profiles/r03/pmc_headline.json
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests exit $?"; tail -3 gpurun_out/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 900 bash profiles/run_profile.sh r03_default 60 "--warmup 20" > gpurun_out/profile_r03_default.txt 2>&1; grep "trace_stack_batch_dense_kernel" gpurun_out/profile_r03_default.txt | head -2
python profiles/make_pmc_json.py gpurun_out/prof_r03_default/summary.txt profiles/r03/pmc_headline.json "trace_stack_batch_dense_kernel" 4 > /dev/null && cp profiles/r03/pmc_headline.json gpurun_out/pmc_headline.json
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err; tail -1 gpurun_out/bench_steps20.json | cut -c1-300
timeout -k 10 300 python bench.py --same-view --no-cpu-baseline > gpurun_out/bench_same_view.json 2>/dev/null; tail -1 gpurun_out/bench_same_view.json | cut -c1-300
SHRAY_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SHRAY_BENCH_CHECK=1 timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist1 exit $?"; grep -h "assembled frames" gpurun_out/bench_dist1.err
for mode in rotate root0; do
SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_TRANSPORT=gloo SHRAY_BENCH_CHECK=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 3 --steps 12 --warmup 3 --trials 2 --root-mode $mode > gpurun_out/bench_gloo3_$mode.json 2> gpurun_out/bench_gloo3_$mode.err; echo "gloo3 $mode exit $?"; grep -h "assembled frames" gpurun_out/bench_gloo3_$mode.err
done
