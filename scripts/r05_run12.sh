#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "free_running or lockstep_env or egyptcat" 2>&1 | tail -3 | cut -c1-400 > gpurun_out/r05_alias8_tests.log
cat gpurun_out/r05_alias8_tests.log
bash scripts/ab.sh "--workload kitchen" shipped nonee notex nosplat noold nt1 2>&1 | tee gpurun_out/r05_logic_probes2_ab.txt
