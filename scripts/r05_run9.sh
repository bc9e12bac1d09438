#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_microkernel.py tests/test_gpu_wide.py -q -x -p no:cacheprovider -k "not full_size" 2>&1 | tail -4 | cut -c1-600 > gpurun_out/r05_alias_tests.log
cat gpurun_out/r05_alias_tests.log
bash scripts/ab.sh "--workload kitchen" shipped nonee 2>&1 | tee gpurun_out/r05_aliasrec_ab.txt
