# A/B of the fused logic+material pass on one workload: scripts/fuse_ab.sh [bench args]
set -u
mkdir -p gpurun_out
run() {  # name args
  n=$1; shift
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline --kernel-timing 1 --overlap 0 "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); k = j['kernel_ms_avg']
        print('$n serial  %7.0f Mrays/s  ms/step %.3f | ' % (j['value'], j['ms_per_step']) + ' '.join('%s=%.3f' % (a, b) for a, b in k.items() if b))
"
  python bench.py --steps 30 --warmup 24 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('$n overlap %7.0f Mrays/s  ms/step %.3f' % (j['value'], j['ms_per_step']))
"
}
for rep in 1 2; do
  run unfused --fuse 0 "$@"
  for v in 1 31; do run set$v --fuse 1 --fuse-set $v "$@"; done
done
