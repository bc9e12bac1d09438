#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box variation is +-4 %): scripts/ab.sh "<bench args>" name1 name2 ...
# name = a file variants/libfluctus_hip_<name>.so, or "shipped" for fluctus_amd/libfluctus_hip.so.  Serial schedule, every kernel timed.
ARGS=$1; shift
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = shipped ]; then unset FLX_HIP_LIB; else export FLX_HIP_LIB=$PWD/variants/libfluctus_hip_$n.so; fi
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline --kernel-timing 1 --overlap 0 $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('$n rep$rep serial  %7.0f Mrays/s  ms/step %.3f | ' % (j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('$n rep$rep overlap %7.0f Mrays/s  ms/step %.3f' % (j['value'], j['ms_per_step']))
"
done
done
