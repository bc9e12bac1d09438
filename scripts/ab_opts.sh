#!/bin/bash
# A/B of bench.py option sets inside ONE gpurun call: scripts/ab_opts.sh "<common bench args>" "<opts A>" "<opts B>" ...
# each set is run on the serial schedule with every kernel timed (kernel times + the 4-wide kernel's own visit counts) and on the default schedule
COMMON=$1; shift
for rep in 1 2; do
for o in "$@"; do
  python bench.py --steps 30 --warmup 24 --windows 1 --no-cpu-baseline --kernel-timing 1 --overlap 0 $COMMON $o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']; r = j['roofline']
        print('[$o] rep$rep serial  %5.0f Mrays/s | extend=%.3f shadow=%.3f logic=%.3f | node visits %.2f leaf %.2f tri %.2f | upload opt %s ms' % (j['value'], k['extend'], k['shadow'], k.get('logic_fused', 0) or 0, r['own_avg_wide_node_visits'], r['own_avg_leaf_visits'], r['own_avg_tri_tests'], j['config'].get('wide_opt_upload_ms')))
"
  python bench.py --steps 30 --warmup 24 --windows 3 --no-cpu-baseline $COMMON $o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('[$o] rep$rep default %5.0f Mrays/s  ms/step %.3f  windows %s' % (j['value'], j['ms_per_step'], ' '.join('%.0f' % x for x in j['windows']['Mrays_s'])))
"
done
done
