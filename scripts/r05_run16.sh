#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_partition.py tests/test_tracer.py tests/test_gpu_denoiser.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-900 > gpurun_out/r05_defer_tests.log
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-900 >> gpurun_out/r05_defer_tests.log
cat gpurun_out/r05_defer_tests.log
( bash scripts/ab_opts2.sh kitchen "--defer-splat 0" "--defer-splat 1"
  bash scripts/ab_opts2.sh conference "--defer-splat 0" "--defer-splat 1" ) 2>&1 | tee gpurun_out/r05_defer_splat_ab.txt
