#!/bin/bash
# one GPU-box call: GPU tests, smoke, the bench lines (default, the driver's --steps 20 form, the multi-rank control flow
# rehearsed on this one GPU)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "gpu tests exit $?"; tail -5 gpurun_out/gputests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -3 gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | cut -c1-1500
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err; tail -1 gpurun_out/bench_steps20.json | cut -c1-400
SHRAY_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 SHRAY_BENCH_CHECK=1 timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist1 exit $?"; tail -2 gpurun_out/bench_dist1.err; tail -1 gpurun_out/bench_dist1.json | cut -c1-400
for mode in rotate root0; do
SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_TRANSPORT=gloo SHRAY_BENCH_CHECK=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 3 --steps 12 --warmup 3 --trials 2 --root-mode $mode > gpurun_out/bench_gloo3_$mode.json 2> gpurun_out/bench_gloo3_$mode.err; echo "gloo3 $mode exit $?"; grep -h "assembled frames" gpurun_out/bench_gloo3_$mode.err; tail -1 gpurun_out/bench_gloo3_$mode.json | cut -c1-300
done
