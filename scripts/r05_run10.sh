#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_partition.py tests/test_tracer.py -q -x -p no:cacheprovider 2>&1 | tail -8 | cut -c1-900 > gpurun_out/r05_regen_tests.log
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-900 >> gpurun_out/r05_regen_tests.log
cat gpurun_out/r05_regen_tests.log
for rep in 1 2; do for rg in 0 1; do
for w in kitchen conference; do
  timeout 300 python bench.py --regen $rg --workload $w --steps 40 --warmup 24 --windows 3 --no-cpu-baseline --kernel-timing 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('rep$rep $w regen $rg overlap %7.0f Mrays/s  ms/step %.3f' % (j['value'], j['ms_per_step']))
"
done; done; done 2>&1 | tee gpurun_out/r05_regen_ab.txt
