for t in 3000 30000 150000 495648; do
  FLX_BENCH_TRIS=$t timeout 300 python bench.py --steps 40 --warmup 16 --no-cpu-baseline --overlap 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; k=j['kernel_ms_avg']
        print('tris',j['config']['triangles'],'nodes',j['config']['bvh_nodes'],'val',round(j['value']),'ext',round(k['extend'],3),'sh',round(k['shadow'],3),'inner',round(r['avg_inner_visits'],1),'tri',round(r['avg_tri_tests'],1),'extrays/step',j['rays']['extension']/j['steps'],'ns/inner-visit-per-Mray', round(k['extend']*1e6/(j['rays']['extension']/j['steps'])/r['avg_inner_visits'],3))
"
done
