export TMPDIR=/tmp
mkdir -p gpurun_out
for w in courtyard-1440p courtyard-2160p; do
  bash scripts/profile_round.sh r04 $w > gpurun_out/r04_profile_$w.log 2>&1
  tail -2 gpurun_out/r04_profile_$w.log
done
mkdir -p gpurun_out/profiles_r04; cp profiles/r04_*_bench.json profiles/r04_*_kernel_stats.csv profiles/r04_*_counters.txt profiles/traffic_*.json gpurun_out/profiles_r04/ 2>/dev/null
ls gpurun_out/profiles_r04
