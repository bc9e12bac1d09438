#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( SECONDS=0; timeout 1500 python -m pytest tests/test_gpu_wide.py tests/test_bench_contract.py -q -x -p no:cacheprovider -k "bench_path_count or contract" 2>&1 | tail -5 | cut -c1-400; echo "elapsed $SECONDS s"
  bash scripts/ab_opts2.sh kitchen "--num-tasks 8388608" "" ) 2>&1 | tee gpurun_out/r05_16M_check.txt
