run() { timeout 300 python bench.py --steps 120 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']
        print('$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],4), {a: round(b,3) for a,b in k.items()})
"; }
for i in 1 2 3; do run --eager-bump 1; run --eager-bump 0; done
run --eager-bump 1 --kernel-timing 0; run --eager-bump 0 --kernel-timing 0
