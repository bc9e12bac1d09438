run() { timeout 300 python bench.py --steps 40 --warmup 16 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']; s=j['roofline']['simd_efficiency'] or {}; r=j['roofline']
        print('$*', '->', round(j['value']), 'ext', round(k['extend'],3), 'sh', round(k['shadow'],3), 'span', round(k.get('trace_span',0),3), 'eff', {a: round(b,3) for a,b in s.items() if isinstance(b,float) and b})
"; }
run --trace-mode 0 --overlap 0
for im in 8 16 24 32 48; do
run --trace-mode 2 --overlap 0 --stream-refill 16 --stream-inner-min $im
done
run --trace-mode 2 --overlap 0 --stream-refill 8 --stream-inner-min 24
run --trace-mode 2 --overlap 0 --stream-refill 32 --stream-inner-min 24
run --trace-mode 2 --overlap 1 --stream-refill 16 --stream-inner-min 24
