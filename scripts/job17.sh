cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { timeout 200 python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('%-62s %7.0f Mrays/s  ms/step %.3f | ' % ('$W $*', j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"; }
{
for rep in 1 2; do
for w in 28 24 20 16; do export FLX_PERSISTENT_WAVES_PER_CU=$w; W="waves/CU=$w";
for wl in kitchen conference; do run --workload $wl --overlap 1; run --workload $wl --overlap 2; done
done
done
for w in 28 24 20; do export FLX_PERSISTENT_WAVES_PER_CU=$w; W="waves/CU=$w"; run --workload courtyard-1440p; run --workload courtyard-1440p --overlap 1 --refill-shadow 0; run --workload courtyard-1440p --overlap 2 --refill-shadow 0; done
} > gpurun_out/r03_wave_slots_sweep.txt 2>&1
cat gpurun_out/r03_wave_slots_sweep.txt
