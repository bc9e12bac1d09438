run() { timeout 300 python bench.py --steps 40 --warmup 16 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']; s=j['roofline']['simd_efficiency'] or {}; r=j['roofline']
        print(os.environ.get('FLX_BENCH_TRIS'), '$*', '->', round(j['value']), 'ext', round(k['extend'],3), 'sh', round(k['shadow'],3), 'inner/ray', round(r['avg_inner_visits'],1), 'Grec/s', round((r['avg_inner_visits']+r['avg_tri_tests'])*j['rays']['extension']/j['steps']/k['extend']/1e6,1), 'eff', {a: round(b,3) for a,b in s.items() if isinstance(b,float) and b})
"; }
for t in 3000 150000 495648; do
export FLX_BENCH_TRIS=$t
run --trace-mode 0 --overlap 0
run --trace-mode 3 --overlap 0 --stream-refill 12
run --trace-mode 2 --overlap 0 --stream-refill 12
done
