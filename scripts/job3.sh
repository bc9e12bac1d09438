cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 150 ./variants/valu_pairs2 > gpurun_out/r03_valu_pairs.txt 2>&1
cat gpurun_out/r03_valu_pairs.txt
timeout 900 python -m pytest tests/test_gpu_rccl_fake.py -x -q -m gpu > gpurun_out/r03_rccl_fake.log 2>&1
tail -30 gpurun_out/r03_rccl_fake.log
timeout 1200 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "refill" > gpurun_out/r03_refill_tests.log 2>&1
tail -5 gpurun_out/r03_refill_tests.log
run() { timeout 300 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-60s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
{
for rep in 1 2; do
run --kernel-timing 1 --overlap 0
run --kernel-timing 1 --overlap 0 --refill-extend 8
run --kernel-timing 1 --overlap 0 --refill-extend 16
run --kernel-timing 1 --overlap 0 --refill-extend 32
run --kernel-timing 1 --overlap 0 --refill-extend 4112
run --kernel-timing 1 --overlap 0 --refill-extend 8208
run --kernel-timing 1 --overlap 0 --refill-extend 4104
run --kernel-timing 1 --overlap 0 --refill-extend 2056
run --kernel-timing 1 --overlap 0 --refill-extend 16 --refill-shadow 16
run --kernel-timing 1 --overlap 0 --refill-extend 16 --refill-shadow 32
run
run --refill-extend 16
run --refill-extend 16 --refill-shadow 16
run --refill-extend 4112 --refill-shadow 4112
done
run --kernel-timing 1 --overlap 0 --workload conference
run --kernel-timing 1 --overlap 0 --workload conference --refill-extend 16
run --kernel-timing 1 --overlap 0 --workload courtyard-1440p
run --kernel-timing 1 --overlap 0 --workload courtyard-1440p --refill-extend 16 --refill-shadow 16
} > gpurun_out/r03_refill_ab.txt 2>&1
cat gpurun_out/r03_refill_ab.txt
