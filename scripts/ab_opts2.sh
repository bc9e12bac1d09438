#!/bin/bash
# same-box A/B of bench OPTIONS on the shipped library: scripts/ab_opts2.sh <workload> "<opts A>" "<opts B>" ...   (default schedule, no kernel timing, 2 reps)
W=$1; shift
for rep in 1 2; do
for o in "$@"; do
  timeout 400 python bench.py --workload $W --steps 40 --warmup 24 --windows 3 --no-cpu-baseline --kernel-timing 0 $o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('rep$rep $W [$o] %7.0f Mrays/s  ms/step %.3f' % (j['value'], j['ms_per_step']))
"
done
done
