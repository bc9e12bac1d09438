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
W=""; run; run --ctx-per-gpu 2; run --ctx-per-gpu 2 --num-tasks 8388608; run --ctx-per-gpu 4 --num-tasks 8388608
for w in 24 20 16 12; do export FLX_PERSISTENT_WAVES_PER_CU=$w; W="waves/CU=$w"; run --ctx-per-gpu 2; run --ctx-per-gpu 2 --num-tasks 8388608; done
unset FLX_PERSISTENT_WAVES_PER_CU; W=""; run --workload conference; run --workload conference --ctx-per-gpu 2 --num-tasks 8388608
export FLX_PERSISTENT_WAVES_PER_CU=16; W="waves/CU=16"; run --workload conference --ctx-per-gpu 2 --num-tasks 8388608
} > gpurun_out/r03_two_wavefronts.txt 2>&1
cat gpurun_out/r03_two_wavefronts.txt
