cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "raw_hit or refill" > gpurun_out/r03_suite_e.log 2>&1
tail -30 gpurun_out/r03_suite_e.log
