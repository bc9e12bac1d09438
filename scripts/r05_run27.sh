#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( bash scripts/ab_opts2.sh kitchen "" "--shadow-split 8" "--shadow-split 12" "--shadow-split 6"
  bash scripts/ab_opts2.sh conference "" "--shadow-split 12" "--shadow-split 16" ) 2>&1 | tee gpurun_out/r05_split_16M.txt
