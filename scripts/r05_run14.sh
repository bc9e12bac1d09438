#!/bin/bash
export TMPDIR=/tmp
REPO=$PWD
mkdir -p gpurun_out
cd /tmp
for tag in base split8; do
  opt=""; [ $tag = split8 ] && opt="--shadow-split 8"
  rm -rf $REPO/gpurun_out/r05_tl_$tag
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/r05_tl_$tag -- python $REPO/bench.py --steps 20 --warmup 16 --windows 1 --no-cpu-baseline --kernel-timing 0 $opt > $REPO/gpurun_out/r05_tl_$tag.log 2>&1
  T=$(ls $REPO/gpurun_out/r05_tl_$tag/*/*kernel_trace.csv | head -1)
  python $REPO/scripts/timeline.py $T > $REPO/gpurun_out/r05_tl_$tag.txt 2>&1
  tail -16 $REPO/gpurun_out/r05_tl_$tag.txt
done
