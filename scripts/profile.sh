#!/bin/bash
# rocprofv3 capture of the bench workload on the GPU box.  Usage: scripts/profile.sh <tag>
# Writes gpurun_out/prof_<tag>/{trace,pmc_fetch,pmc_write,pmc_sq,pmc_tcc}; summaries are copied to profiles/ by hand.
set -u
TAG=${1:-r01}
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 16 --no-cpu-baseline"
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
# counters: one pass each (TCC slot limits), kernel-trace only
timeout -k 5 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- $CMD > $OUT/pmc_tcc.log 2>&1
timeout -k 5 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -5 $OUT/trace.log
cat $OUT/summary.txt
