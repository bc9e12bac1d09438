cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 200 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-86s %7.0f Mrays/s  ms/step %.3f | ' % ('$*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
E() { echo $(( $1 | ($2 << 8) )); }
{
for wl in kitchen conference courtyard-1440p; do
run --workload $wl
for ov in 1 2 0; do
for ext in "16 32" "24 32" "32 32" "24 40"; do
run --workload $wl --overlap $ov --refill-extend $(E $ext)
done
run --workload $wl --overlap $ov --refill-extend $(E 16 32) --refill-shadow $(E 16 32)
run --workload $wl --overlap $ov --refill-extend $(E 24 32) --refill-shadow $(E 16 48)
done
done
run --num-tasks 8388608
run --num-tasks 8388608 --overlap 1 --refill-extend $(E 16 32)
run --num-tasks 6291456 --overlap 1 --refill-extend $(E 16 32)
run --num-tasks 2097152 --overlap 1 --refill-extend $(E 16 32)
} > gpurun_out/r03_refill_ab7.txt 2>&1
cat gpurun_out/r03_refill_ab7.txt
