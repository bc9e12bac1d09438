#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_microkernel.py -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-900 > gpurun_out/r05_neerec_tests.log
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider 2>&1 | tail -5 | cut -c1-900 >> gpurun_out/r05_neerec_tests.log
cat gpurun_out/r05_neerec_tests.log
bash scripts/ab.sh "--workload kitchen" shipped nonee 2>&1 | tee gpurun_out/r05_neerec_ab.txt
