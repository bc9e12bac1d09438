cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./variants/valu_pairs > gpurun_out/r03_valu_pairs.txt 2>&1
cat gpurun_out/r03_valu_pairs.txt
timeout 1200 python -m pytest tests/test_gpu_rccl_fake.py -x -q -m gpu > gpurun_out/r03_rccl_fake.log 2>&1
tail -30 gpurun_out/r03_rccl_fake.log
