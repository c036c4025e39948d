#!/bin/bash
# issue counters of two library builds on the default (pipelined) bench command
for V in A_nodeal_w7_k28 C_deal_w6_k48; do
  export SHRAY_HIP_LIB=$PWD/shader-ray_amd/_variants/libshray_hip_$V.so
  bash profiles/run_profile.sh ab_$V 30 > gpurun_out/ab_$V.txt 2>&1
  echo "== $V"; grep -A30 "batch_kernel<true, true>" gpurun_out/ab_$V.txt | grep -E "calls|SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_VMEM_RD|SQ_THREAD_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_ACTIVE_INST_ANY|VGPR" | cut -c1-150
done
