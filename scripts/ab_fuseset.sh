#!/bin/bash
# A/B (one box): BSDF types the fused pass inlines (fuse_set 1 diffuse | 31 all) x extension-queue order, round-3 kernels and schedule
for rep in 1 2; do
for w in courtyard-1440p kitchen courtyard-2160p; do
for cfg in "" "--fuse-set 31 --ext-order 1" "--fuse-set 31 --ext-order 0" "--fuse-set 1 --ext-order 1"; do
  timeout 400 python bench.py --workload $w --steps 30 --warmup 24 --no-cpu-baseline $cfg 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-16s %-32s rep$rep %7.0f Mrays/s  ms/step %.3f | ' % ('$w', '$cfg' or '(default)', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"
done; done; done
