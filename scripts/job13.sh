cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ab.sh "" shipped fast > gpurun_out/r03_fastmath_ab.txt 2>&1
cat gpurun_out/r03_fastmath_ab.txt
bash scripts/ab.sh "--workload conference" shipped fast > gpurun_out/r03_fastmath_ab_conference.txt 2>&1
cat gpurun_out/r03_fastmath_ab_conference.txt
for w in kitchen conference courtyard-1440p courtyard-2160p; do
  timeout 600 python bench.py --workload $w > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err
  python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r03_bench_$w.json") if l.startswith("{")][-1])
r = j["roofline"]
print("$w", round(j["value"]), "Mrays/s", round(j["ms_per_step"], 3), "ms | frac", round(r["frac"], 3), r["frac_source"], "| alone", r["launch_ms_alone"], "| cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline_port", {}).get("value"))
PY
done
