# A/B: thread-per-ray (mode 0) vs chunk-refill traversal (mode 2 while-while, mode 3 unified work items)
run() { timeout 300 python bench.py --steps 40 --warmup 16 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']; s=j['roofline']['simd_efficiency'] or {}
        print('$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],3), 'ext', round(k['extend'],3), 'sh', round(k['shadow'],3), 'span', round(k.get('trace_span',0),3), 'eff', {a: round(b,3) for a,b in s.items() if isinstance(b,float) and b})
"; }
run --trace-mode 0 --overlap 0
for r in ${REFILLS:-4 12 24 40}; do
  run --trace-mode 3 --overlap 0 --stream-refill $r
done
run --trace-mode 3 --overlap 1 --stream-refill 12
run --trace-mode 3 --overlap 0 --stream-refill 12 --stream-waves-ext 24 --stream-waves-shadow 24
run --trace-mode 3 --overlap 0 --stream-refill 12 --stream-waves-ext 48 --stream-waves-shadow 48
run --trace-mode 3 --overlap 1 --stream-refill 12 --stream-waves-ext 12 --stream-waves-shadow 12
