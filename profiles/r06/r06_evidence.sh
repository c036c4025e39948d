#!/bin/bash
# round 6 evidence run (one gpurun call): GPU tests, the rocprofv3 passes behind profiles/r06/pmc_headline.json, bench lines, configs
mkdir -p gpurun_out profiles/r06
timeout -k 10 900 bash profiles/run_profile.sh r06_default 60 "--warmup 20" > gpurun_out/profile_r06_default.txt 2>&1; grep "trace_stack_batch_dense_kernel" gpurun_out/profile_r06_default.txt | head -2
python profiles/make_pmc_json.py gpurun_out/prof_r06_default/summary.txt gpurun_out/pmc_headline.json "trace_stack_batch_dense_kernel" 4 > /dev/null
cp gpurun_out/pmc_headline.json profiles/r06/pmc_headline.json      # (on the box: the bench lines below read it)
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default.json | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err; tail -1 gpurun_out/bench_steps20.json | cut -c1-300
timeout -k 10 300 python bench.py --same-view --no-cpu-baseline > gpurun_out/bench_same_view.json 2>/dev/null; tail -1 gpurun_out/bench_same_view.json | cut -c1-300
timeout -k 10 600 python profiles/run_configs.py r06 > gpurun_out/r06_configs.log 2>&1; grep '"config"' gpurun_out/r06_configs.log | cut -c1-160
timeout -k 10 300 bash profiles/r04/r04_config_profile.sh config3 SPP=64 MATERIAL=6 REPS=4 > gpurun_out/profile_r06_config3.txt 2>&1
timeout -k 10 300 bash profiles/r04/r04_config_profile.sh config5 WIDTH=3840 HEIGHT=2160 SPP=16 REPS=4 > gpurun_out/profile_r06_config5.txt 2>&1
timeout -k 10 300 bash profiles/r04/r04_config_profile.sh config4 SCENE=million SPP=4 REPS=8 > gpurun_out/profile_r06_config4.txt 2>&1
timeout -k 10 300 bash profiles/r04/r04_config_profile.sh lone_frame REPS=12 > gpurun_out/profile_r06_lone_frame.txt 2>&1
SHRAY_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 40 --no-cpu-baseline > gpurun_out/bench_dist1.json 2> gpurun_out/bench_dist1.err; echo "dist1 exit $?"; tail -1 gpurun_out/bench_dist1.json | cut -c1-200
SHRAY_BENCH_ONE_GPU=1 SHRAY_BENCH_TRANSPORT=gloo timeout -k 10 600 python bench.py --gpus 4 --steps 16 --warmup 4 --trials 2 > gpurun_out/bench_gloo4.json 2> gpurun_out/bench_gloo4.err; echo "gloo4 exit $?"; tail -1 gpurun_out/bench_gloo4.json | cut -c1-300
python profiles/vector_cache_probe.py > /dev/null 2>&1; python profiles/scene_turnaround.py > /dev/null 2>&1
