cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_partition.py tests/test_tracer.py -x -q -m gpu > gpurun_out/r03_suite_b.log 2>&1
tail -4 gpurun_out/r03_suite_b.log
FLX_REFILL_SHADOW=8208 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not full_size" > gpurun_out/r03_parity_refill.log 2>&1
tail -3 gpurun_out/r03_parity_refill.log
run() { timeout 200 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-50s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
for rep in 1 2; do run --workload conference; run --workload conference --kernel-timing 1 --overlap 0; run; done
