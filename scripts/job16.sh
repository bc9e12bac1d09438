cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "refill" > gpurun_out/r03_suite_d.log 2>&1
tail -3 gpurun_out/r03_suite_d.log
run() { timeout 200 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-62s %7.0f Mrays/s  ms/step %.3f | ' % ('$W $*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
for rep in 1 2; do
W=""; run; run --kernel-timing 1 --overlap 0
run --workload conference --kernel-timing 1 --overlap 0
run --workload courtyard-1440p
done
for w in 24 20 16 12; do export FLX_PERSISTENT_WAVES_PER_CU=$w; W="waves/CU=$w"; run; run --overlap 2; run --workload conference; done
