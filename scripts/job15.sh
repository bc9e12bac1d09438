cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "refill or default_path" > gpurun_out/r03_suite_c.log 2>&1
tail -3 gpurun_out/r03_suite_c.log
run() { timeout 200 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-62s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
E() { echo $(( $1 | ($2 << 8) )); }
for rep in 1 2; do
run; run --kernel-timing 1 --overlap 0
run --workload conference; run --workload conference --kernel-timing 1 --overlap 0
run --workload courtyard-1440p; run --workload courtyard-1440p --kernel-timing 1
done
run --refill-extend $(E 16 24); run --refill-extend $(E 8 32); run --refill-extend $(E 24 32); run --refill-extend $(E 16 40); run --overlap 2
