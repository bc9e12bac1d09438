#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in shipped prio; do
for o in "" "--shadow-split 8"; do
  if [ $lib = shipped ]; then unset FLX_HIP_LIB; else export FLX_HIP_LIB=$PWD/variants/libfluctus_hip_$lib.so; fi
  for w in kitchen conference; do
  timeout 400 python bench.py --workload $w --steps 40 --warmup 24 --windows 3 --no-cpu-baseline --kernel-timing 0 $o 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('rep$rep $w $lib [$o] %7.0f Mrays/s  ms/step %.3f' % (j['value'], j['ms_per_step']))
"
  done
done; done; done 2>&1 | tee gpurun_out/r05_stream_prio_ab.txt
