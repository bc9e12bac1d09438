cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ab.sh "" shipped fast fastdiv > gpurun_out/r03_fastmath_ab.txt 2>&1
cat gpurun_out/r03_fastmath_ab.txt
bash scripts/ab.sh "--workload conference" shipped fast > gpurun_out/r03_fastmath_ab_conference.txt 2>&1
cat gpurun_out/r03_fastmath_ab_conference.txt
for w in kitchen conference courtyard-1440p; do
  timeout 1500 bash scripts/profile_r03.sh $w > gpurun_out/r03_profile_$w.log 2>&1
  tail -3 gpurun_out/r03_profile_$w.log
done
ls -la profiles | tail -20
mkdir -p gpurun_out/profiles_r03; cp profiles/r03_*_bench.json profiles/r03_*_kernel_stats.csv profiles/r03_*_counters.txt profiles/traffic_*.json gpurun_out/profiles_r03/ 2>/dev/null
