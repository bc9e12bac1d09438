cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "refill" > gpurun_out/r03_refill_tests.log 2>&1
tail -5 gpurun_out/r03_refill_tests.log
run() { timeout 300 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-70s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
E() { echo $(( $1 | ($2 << 8) )); }
{
for rep in 1 2; do
run --kernel-timing 1 --overlap 0
run --kernel-timing 1 --overlap 0 --refill-extend $(E 16 32)
run --kernel-timing 1 --overlap 0 --refill-extend $(E 16 40)
run --kernel-timing 1 --overlap 0 --refill-extend $(E 24 40)
run --kernel-timing 1 --overlap 0 --refill-extend $(E 8 32)
run --kernel-timing 1 --overlap 0 --refill-extend $(E 16 32) --refill-shadow $(E 16 32)
run
run --refill-extend $(E 16 32)
run --refill-extend $(E 16 32) --overlap 1
run --refill-extend $(E 16 32) --refill-shadow $(E 16 32)
run --refill-extend $(E 16 32) --refill-shadow $(E 16 32) --overlap 0
done
run --workload courtyard-1440p
run --workload courtyard-1440p --refill-extend $(E 16 32)
run --workload courtyard-1440p --refill-extend $(E 16 32) --refill-shadow $(E 16 32) --overlap 0
run --workload courtyard-1440p --kernel-timing 1 --overlap 0 --refill-extend $(E 16 32) --refill-shadow $(E 16 32)
} > gpurun_out/r03_refill_ab4.txt 2>&1
cat gpurun_out/r03_refill_ab4.txt
