for shape in "2 4" "4 2" "4 4" "8 2" "3 4" "2 6" "1 8"; do
  set -- $shape
  for steps in 200 20; do
  python bench.py --no-cpu-baseline --steps $steps --warmup 5 --frames-per-launch $1 --frames-in-flight $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frames/launch $1 streams $2 steps $steps:', d['ms_per_step'], 'ms/frame', d['value'], 'Mrays/s')"
  done
done
