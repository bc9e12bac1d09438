#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-600 > gpurun_out/r05_full_tests.log
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-600 >> gpurun_out/r05_full_tests.log
cat gpurun_out/r05_full_tests.log
bash scripts/ab.sh "--workload kitchen" full0 shipped > gpurun_out/r05_full_stores_ab.txt 2>&1
bash scripts/ab.sh "--workload conference" full0 shipped >> gpurun_out/r05_full_stores_ab.txt 2>&1
cat gpurun_out/r05_full_stores_ab.txt
