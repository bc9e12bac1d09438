cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ab.sh "" shipped shcap32k shcap48k > gpurun_out/r03_shadow_grid_cap_ab.txt 2>&1
cat gpurun_out/r03_shadow_grid_cap_ab.txt
for rep in 1 2 3; do for w in kitchen conference courtyard-1440p; do
python bench.py --workload $w --steps 40 --warmup 24 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('$w %7.0f Mrays/s  ms/step %.3f | ' % (j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b), '| alone %.3f' % j['roofline']['launch_ms_alone'])
"
done; done
