for rep in 1 2; do
for st in 1 2 3 4 1000000; do
  echo -n "report stride $st: "
  SHRAY_DISPATCH_REPORT_STRIDE=$st bash profiles/r05/r05_quick_ab.sh || exit 1
done; done
