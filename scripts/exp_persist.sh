for n in p1 p5 p6 p7; do
  export FLX_HIP_LIB=$PWD/variants/libfluctus_hip_$n.so
  for a in "--persist 0" "--persist 3" "--persist 3 --refill-min 16" "--persist 1" "--persist 2"; do
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline --kernel-timing 1 --overlap 0 $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('$n $a serial  %7.0f Mrays/s  ms/step %.3f | ' % (j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"
  done
done
