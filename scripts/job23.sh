cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "empty_and_tiny or raw_hit" > gpurun_out/r03_suite_f.log 2>&1
tail -30 gpurun_out/r03_suite_f.log
bash scripts/ab.sh "" shipped rgtemp > gpurun_out/r03_raygen_temporal_ab.txt 2>&1
cat gpurun_out/r03_raygen_temporal_ab.txt
