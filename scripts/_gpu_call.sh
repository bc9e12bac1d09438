export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fuzz.py -q -x -s 2>&1 | grep -v "^$" | tail -12 > gpurun_out/r04_fuzz.log
cat gpurun_out/r04_fuzz.log
bash scripts/gpu_validate.sh r04a
