cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ab.sh "" shipped neeearly > gpurun_out/r03_logic_nee_early_ab.txt 2>&1
cat gpurun_out/r03_logic_nee_early_ab.txt
timeout 3300 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r03_gpu_suite_a.log 2>&1
tail -14 gpurun_out/r03_gpu_suite_a.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1
tail -1 gpurun_out/r03_smoke.log
for w in kitchen conference courtyard-1440p; do
  timeout 1500 bash scripts/profile_r03.sh $w > gpurun_out/r03_profile_$w.log 2>&1
  tail -1 gpurun_out/r03_profile_$w.log
done
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_*_kernel_stats.csv profiles/r03_*_counters.txt profiles/traffic_*.json gpurun_out/profiles_r03/ 2>/dev/null
for w in kitchen conference courtyard-1440p courtyard-2160p; do
  timeout 600 python bench.py --workload $w > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err
  python - <<PY
import json
j = json.loads([l for l in open("gpurun_out/r03_bench_$w.json") if l.startswith("{")][-1])
r = j["roofline"]
print("$w", round(j["value"]), "Mrays/s", round(j["ms_per_step"], 3), "ms | frac", round(r["frac"], 3), r["frac_source"], "| alone", round(r["launch_ms_alone"], 3), "| cpu", j.get("cpu_baseline", {}).get("value"), j.get("cpu_baseline_port", {}).get("value"))
PY
done
