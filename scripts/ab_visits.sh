# like ab.sh, plus the 4-wide kernel's own visit counts: scripts/ab_visits.sh "<bench args>" name...
ARGS=$1; shift
for n in "$@"; do
  if [ "$n" = shipped ]; then unset FLX_HIP_LIB; else export FLX_HIP_LIB=$PWD/variants/libfluctus_hip_$n.so; fi
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline --kernel-timing 1 --overlap 0 $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']; r = j['roofline']
        print('$n serial %5.0f Mrays/s | extend=%.3f shadow=%.3f | node visits %.2f leaf %.2f tri %.2f' % (j['value'], k['extend'], k['shadow'], r['own_avg_wide_node_visits'], r['own_avg_leaf_visits'], r['own_avg_tri_tests']))
"
done
