export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r04_knobs_8m.txt
run() { python bench.py --steps 30 --windows 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('$*: %5.0f Mrays/s  ms/step %.3f  windows %s' % (j['value'], j['ms_per_step'], ' '.join('%.0f' % x for x in j['windows']['Mrays_s'])))
" >> gpurun_out/r04_knobs_8m.txt; }
for w in kitchen conference courtyard-1440p; do
  run --workload $w
  run --workload $w --refill-extend $((16 | 24 << 8))
  run --workload $w --refill-extend $((24 | 32 << 8))
  run --workload $w --refill-extend $((8 | 32 << 8))
  run --workload $w --refill-extend $((16 | 40 << 8))
  run --workload $w --fuse-set 1 --ext-order 0
  run --workload $w --fuse-set 31 --ext-order 1
  run --workload $w --overlap 1
  run --workload $w --refill-shadow $((16 | 32 << 8))
  run --workload $w
done
cat gpurun_out/r04_knobs_8m.txt
timeout 1200 python -m pytest tests/test_gpu_wide.py -q -x -k "bench_path_count" 2>&1 | tail -3
