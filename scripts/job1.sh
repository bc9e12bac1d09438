set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./variants/valu_rate > gpurun_out/r03_valu_rate.txt 2>&1
FLX_HIP_LIB=$PWD/variants/libfluctus_hip_mix.so timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/r03_mix_wide_tests.log 2>&1
tail -5 gpurun_out/r03_mix_wide_tests.log
bash scripts/ab.sh "" shipped w7 folde mix mixw7 > gpurun_out/r03_ab1.txt 2>&1
cat gpurun_out/r03_ab1.txt
