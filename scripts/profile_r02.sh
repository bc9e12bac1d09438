#!/bin/bash
# Round-2 capture for one workload: scripts/profile_r02.sh <tag> [bench args, e.g. --workload courtyard-1440p]
#   (1) rocprofv3 --kernel-trace --stats   (2) fabric-side request counters of the traversal kernels (scripts/pmc_hbm.sh)
#   (3) TCC hit/miss + SQ occupancy.  All --pmc passes are separate runs with --kernel-trace only.
set -u
TAG=${1:-r02}; shift
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 16 --no-cpu-baseline $*"
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
grep '^{' $OUT/trace.log > $OUT/bench.json
timeout -k 5 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cd $REPO
bash scripts/pmc_hbm.sh $TAG $* > $OUT/fabric_bytes.txt 2>&1
cat $OUT/summary.txt | head -40
tail -30 $OUT/fabric_bytes.txt
