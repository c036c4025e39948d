set -e
# multi-rank rehearsal on ONE GPU (gloo): the GPU box allows at most 6 processes on the card and the
# launcher counts as one, so world sizes up to 5
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "batches or tile_sets or error_paths or deinterleave" 2>&1 | tail -5
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
SHRAY_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 20 --warmup 4 --frames-per-launch 3 2>&1 | tail -1 | cut -c1-900
SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_BACKEND=gloo SHRAY_BENCH_CHECK=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 3 --frames-per-launch 3 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -4 | cut -c1-900
SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_BACKEND=gloo SHRAY_BENCH_CHECK=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 3 --steps 9 --warmup 3 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -4 | cut -c1-900
