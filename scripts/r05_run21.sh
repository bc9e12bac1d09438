#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider -k "in_kernel_regeneration" 2>&1 | tail -3 | cut -c1-300
  bash scripts/ab_opts2.sh kitchen "" "--num-tasks 12582912" "--num-tasks 16777216" "--num-tasks 6291456"
  bash scripts/ab_opts2.sh conference "" "--num-tasks 12582912" "--num-tasks 16777216" ) 2>&1 | tee gpurun_out/r05_num_tasks.txt
