export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r04_num_tasks.txt
for w in kitchen conference courtyard-1440p; do
for n in 4194304 6291456 8388608 16777216; do
  python bench.py --workload $w --num-tasks $n --steps 30 --windows 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('$w paths $n: %5.0f Mrays/s  ms/step %.3f  windows %s  ext alone %.3f' % (j['value'], j['ms_per_step'], ' '.join('%.0f' % x for x in j['windows']['Mrays_s']), j['roofline']['launch_ms_alone']))
" >> gpurun_out/r04_num_tasks.txt
done; done
cat gpurun_out/r04_num_tasks.txt
