#!/bin/bash
# Evidence for the two designs that were analysed and not built (profiles/EXPERIMENTS.md, round 3):
#  (1) both children per node turn needs the far child's entry distance on the ray's stack (64-bit entries): what the
#      doubled LDS footprint costs the headline kernel in resident waves -- the shipped library against one whose
#      workgroups take 3,840 more bytes of LDS each (SHRAY_LDS_PAD), same sources otherwise
#  (2) a bounce-split (wavefront) form cannot shorten a lone frame if the frame already lasts as long as its slowest
#      wave's dependent chain: the per-wave timeline of one frame (diagnostic build)
mkdir -p gpurun_out
for lib in "" shader-ray_amd/_variants/libshray_hip_ldspad.so; do
  for rep in 1 2; do
    SHRAY_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${lib:-shipped}', 'throughput form', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s; one frame at a time', d['latency']['ms'], 'ms')"
  done
done
timeout -k 10 300 python profiles/timeline.py > gpurun_out/timeline_lone_frame.txt 2>&1; grep -v "^ *[0-9]* *[0-9]* *[0-9]*$" gpurun_out/timeline_lone_frame.txt | head -40
timeout -k 10 300 python profiles/timeline.py --million > gpurun_out/timeline_million.txt 2>&1; grep "kernel span\|heavy wave\|wave duration" gpurun_out/timeline_million.txt | head -8
