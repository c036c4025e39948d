for rep in 1 2; do
for shape in "2 4" "4 4" "5 4" "10 2" "4 2" "20 1" "7 3"; do
  set -- $shape
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --frames-per-launch $1 --frames-in-flight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep frames/launch $1 streams $2 steps 20:', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s', 'min trial', min(d['trial_ms']))"
done
done
