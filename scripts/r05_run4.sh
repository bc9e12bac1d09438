#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300 > gpurun_out/r05_mlp_tests.log
timeout 900 python -m pytest tests/test_gpu_wide.py -q -x -p no:cacheprovider -k "bench_launch_chain or default_path_free_run" 2>&1 | tail -6 | cut -c1-300 >> gpurun_out/r05_mlp_tests.log
FLX_HIP_LIB=$PWD/variants/libfluctus_hip_lb64.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "free_running or lockstep_simple" 2>&1 | tail -3 | cut -c1-300 >> gpurun_out/r05_mlp_tests.log
cat gpurun_out/r05_mlp_tests.log
bash scripts/ab.sh "--workload kitchen" v00 v10 v01 shipped lb64 lb128 > gpurun_out/r05_logic_variants_ab.txt 2>&1
bash scripts/ab.sh "--workload conference" v00 v10 v01 shipped lb64 >> gpurun_out/r05_logic_variants_ab.txt 2>&1
cat gpurun_out/r05_logic_variants_ab.txt
