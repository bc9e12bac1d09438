#!/bin/bash
# One GPU-box call for a round's evidence: scripts/round_capture.sh <tag> "<workloads>" [suite]
#   [suite]      the full -m gpu suite + smoke() first (scripts/gpu_validate.sh)
#   per workload scripts/profile_round.sh (kernel stats, PMC passes -> traffic_<w>.json with the per-pass blocks), then the bench line AGAIN
#                so that it quotes the capture just made (no committed line with frac null), and the per-iteration timeline of the trace
#   the default `python bench.py` line (cpu baseline included) as <tag>_final_bench.json
# Everything that has to survive the call is copied to gpurun_out/profiles_<tag>/ (profiles/ on the box is not merged back).
set -u
TAG=${1:-r05}; WLS=${2:-kitchen}; SUITE=${3:-}
export TMPDIR=/tmp
REPO=$PWD
DST=$REPO/gpurun_out/profiles_$TAG
mkdir -p $DST
if [ -n "$SUITE" ]; then
  bash scripts/gpu_validate.sh $TAG; echo "validate rc=$?"
  cp gpurun_out/${TAG}_gpu_suite.log gpurun_out/${TAG}_smoke.log $DST/ 2>/dev/null
fi
for W in $WLS; do
  bash scripts/profile_round.sh $TAG $W > gpurun_out/${TAG}_profile_$W.log 2>&1
  tail -2 gpurun_out/${TAG}_profile_$W.log
  python bench.py --workload $W --no-cpu-baseline 2>/dev/null | grep '^{' > profiles/${TAG}_${W}_bench.json
  T=$(ls gpurun_out/prof_${TAG}_$W/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python scripts/timeline.py $T > profiles/${TAG}_${W}_timeline.txt 2>&1
  cp profiles/${TAG}_${W}_* profiles/traffic_$W.json $DST/ 2>/dev/null
  rm -rf gpurun_out/prof_${TAG}_$W                  # (raw traces: tens of MB per workload; everything judged is in $DST by now)
done
python bench.py 2>gpurun_out/${TAG}_final_bench.err | grep '^{' > $DST/${TAG}_final_bench.json
python - <<PY
import json
j = json.loads(open("$DST/${TAG}_final_bench.json").readline())
r = j["roofline"]
print("final: %.0f Mrays/s  %.3f ms/step  bound %s  hbm frac %s  valu %s" % (j["value"], j["ms_per_step"], r["bound"], r["frac"], (r.get("valu") or {}).get("issue_frac")))
for k, o in (r.get("other_kernels") or {}).items():
    print("  ", k, o["launch_ms"], o["bound"], o["hbm"]["frac"], (o.get("valu") or {}).get("issue_frac"))
PY
