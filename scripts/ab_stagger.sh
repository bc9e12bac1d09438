#!/bin/bash
# A/B: state record arrays staggered inside their allocations (FLX_STATE_STAGGER, float4 elements) -- run-to-run spread of the memory-bound passes
for rep in 1 2 3; do
for st in 0 272 4112 65808; do
  export FLX_STATE_STAGGER=$st
  for w in kitchen; do
  timeout 300 python bench.py --workload $w --steps 30 --warmup 24 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('stagger %-6s %-12s rep$rep %7.0f Mrays/s  ms/step %.3f | ' % ('$st', '$w', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"
  done
done; done
