cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 300 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-70s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
E() { echo $(( $1 | ($2 << 8) )); }
D() { echo $(( $1 | ($2 << 8) | 65536 )); }
{
for rep in 1 2; do
run
run --refill-extend $(D 16 32)
run --refill-extend $(D 16 32) --overlap 1
run --refill-extend $(D 16 32) --overlap 0
run --refill-extend $(D 16 32) --refill-shadow $(E 16 32) --overlap 0
run --refill-extend $(D 16 40)
run --refill-extend $(D 24 40)
run --refill-extend $(D 8 32)
done
run --kernel-timing 1 --overlap 0 --refill-extend $(D 64 32)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 64 16)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 32 32)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 16 24)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 16 40)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 24 40)
run --kernel-timing 1 --overlap 0 --refill-extend $(D 24 48)
run --workload conference --refill-extend $(D 16 32)
run --workload courtyard-1440p --refill-extend $(D 16 32)
} > gpurun_out/r03_refill_ab3.txt 2>&1
cat gpurun_out/r03_refill_ab3.txt
