run() { timeout 300 python bench.py --steps 80 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); k=j['kernel_ms_avg']
        print(os.path.basename(os.environ.get('FLX_HIP_LIB','default')), '$*', '->', round(j['value']), 'ms/step', round(j['ms_per_step'],4), {a: round(b,3) for a,b in k.items()})
"; }
for v in ${VARIANTS}; do
  export FLX_HIP_LIB=$PWD/variants/libfluctus_hip_$v.so
  run --overlap 0 --kernel-timing 1; run
done
