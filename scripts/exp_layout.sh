run() { timeout 300 python bench.py --steps 40 --warmup 16 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']
        print('$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],3), 'ext', round(k['extend'],3), 'sh', round(k['shadow'],3), 'span', round(k.get('trace_span',0),3))
"; }
run --node-layout 0 --overlap 0
run --node-layout 1 --overlap 0
run --node-layout 0
run --node-layout 1
run --node-layout 1 --workload conference
run --node-layout 0 --workload conference
run --node-layout 1 --workload courtyard-1440p
run --node-layout 0 --workload courtyard-1440p
