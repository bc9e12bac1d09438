#!/bin/bash
# quick A/B of bench variants on the GPU box: scripts/sweep.sh "<args1>" "<args2>" ...
for a in "$@"; do
  python bench.py --steps 40 --warmup 20 --no-cpu-baseline $a 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-58s Mrays/s %.0f  ms/step %.3f  ext %.3f shadow %.3f logic %.3f mat %.3f raygen %.3f  roofline %.3f' % ('$a', j['value'], j['ms_per_step'], k.get('extend',0), k.get('shadow',0), k.get('logic',0), k.get('materials',0), k.get('raygen',0), j['roofline']['frac']))
"
done
