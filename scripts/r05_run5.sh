#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -p no:cacheprovider 2>&1 | tail -8 | cut -c1-1500 > gpurun_out/r05_fuzz.log
cat gpurun_out/r05_fuzz.log
bash scripts/ab.sh "--workload kitchen" v00 noenv d250 d500 > gpurun_out/r05_logic_probes_ab.txt 2>&1
cat gpurun_out/r05_logic_probes_ab.txt
