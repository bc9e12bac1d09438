run() { timeout 300 python bench.py --steps 120 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']
        print('$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],4), {a: round(b,3) for a,b in k.items()}, j['kernel_ms_avg_source']['extra_untimed_pass'])
"; }
for t in 3 2 0 3; do run --kernel-timing $t; done
run --kernel-timing 2 --workload conference
run --kernel-timing 1 --workload conference
