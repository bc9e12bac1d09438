export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -s -k "arithmetic_contract" 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r04_fork.log
timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x -s -k "bench_launch_chain" 2>&1 | grep -v "^$" | grep -E "fork|passed|failed|Error" >> gpurun_out/r04_fork.log
cat gpurun_out/r04_fork.log; cat gpurun_out/r04_wide_flips.json | tail -30
