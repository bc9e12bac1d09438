#!/bin/bash
# Capture for one workload: scripts/profile_round.sh <round tag, e.g. r04> <workload> [extra bench args]
#   (1) rocprofv3 --kernel-trace --stats of `python bench.py --workload <w>` (the bench line is kept beside it)
#   (2) fabric-side request counters of the traversal kernels, separate --pmc passes with --kernel-trace only (the guide's HBM recipe:
#       request counts by size; = 2 x FETCH_SIZE + WRITE_SIZE with the gfx950 correction)  -> profiles/traffic_<workload>.json
#   (3) SQ counters of the traversal kernels (instructions, lanes per instruction, waits)
# Everything lands in profiles/<tag>_<workload>_*; run from the repo root on the GPU box.
set -u
TAG=${1:-r04}; shift
W=${1:-kitchen}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/prof_${TAG}_$W
rm -rf $OUT; mkdir -p $OUT $REPO/profiles
export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload $W --steps 20 --warmup 16 --windows 1 --no-cpu-baseline --no-ref-gpu-baseline $*"      # (the reference's OpenCL kernels are not what is profiled here)
cd /tmp
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log > $REPO/profiles/${TAG}_${W}_bench.json
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
done
cd $REPO
python scripts/summarize_round.py $OUT $TAG $W "$*"
# gpurun merges at most 64 MiB back: keep the kernel trace (the timeline is made from it), drop the eight counter passes' raw CSVs once they are summarised
rm -rf $OUT/p[0-9]*
