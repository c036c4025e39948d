# usage: bash profiles/param_sweep.sh "<flags1>" "<flags2>" ...   (rebuilds libshray_hip.so per variant on the GPU box)
run() { python bench.py --kernel ${KERNEL:-0} --no-cpu-baseline --steps 200 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$1', d['value'], d['roofline']['kernel_ms_avg'])"; }
for v in "$@"; do
  make -C shader-ray_amd -B hip HIP_EXTRA="$v -Rpass-analysis=kernel-resource-usage" 2>&1 | grep -A8 "${KSYM:-trace_stack_kernelILb0}" | grep -E "VGPRs:|Scratch" | sed 's/.*remark: *//' | tr '\n' ' '; echo
  run "$v"
done
