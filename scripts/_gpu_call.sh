export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/gpu_validate.sh r04
FLX_SOAK_ITERS=100 timeout 1500 python -m pytest tests/test_gpu_wide.py -q -x -k "bench_launch_chain" 2>&1 | tail -3 > gpurun_out/r04_soak100.txt
cat gpurun_out/r04_wide_flips.json >> gpurun_out/r04_soak100.txt
timeout 900 python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
timeout 900 python bench.py --workload egyptcat --num-tasks 1048576 --no-cpu-baseline > gpurun_out/r04_egyptcat_1M_bench.json 2>/dev/null
FLX_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_forced_dist.json 2> gpurun_out/r04_bench_forced_dist.err
tail -3 gpurun_out/r04_soak100.txt; cut -c1-700 gpurun_out/r04_final_bench.json; cut -c1-300 gpurun_out/r04_egyptcat_1M_bench.json; cut -c1-300 gpurun_out/r04_bench_forced_dist.json
