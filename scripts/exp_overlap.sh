run() { timeout 300 python bench.py --steps 60 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']
        print(os.environ.get('FLX_STREAM_PRIORITY'), '$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],3), {a: round(b,3) for a,b in k.items()})
"; }
for pr in 0 1; do export FLX_STREAM_PRIORITY=$pr
for w in kitchen conference courtyard-1440p; do
for o in 1 2; do run --overlap $o --workload $w; done
done; done
