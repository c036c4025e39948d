run() { python bench.py --kernel 0 --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('$1', d['value'], d['roofline']['kernel_ms_avg'])"; }
for v in "-DSHRAY_MIN_WAVES=1" "-DSHRAY_MIN_WAVES=5" "-DSHRAY_MIN_WAVES=6" "-DSHRAY_MIN_WAVES=8" "-DSHRAY_LDS_PAD=45000" "-DSHRAY_LDS_PAD=70000"; do
  make -C shader-ray_amd -B hip HIP_EXTRA="$v -Rpass-analysis=kernel-resource-usage" 2>&1 | grep -A8 "trace_stack_kernelILb0" | grep -E "VGPRs:|Scratch|Occupancy" | tr '\n' ' '; echo
  run "$v"
done
