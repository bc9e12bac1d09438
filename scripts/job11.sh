cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r03_gpu_suite_a.log 2>&1
tail -40 gpurun_out/r03_gpu_suite_a.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
tail -3 gpurun_out/r03_smoke.log
