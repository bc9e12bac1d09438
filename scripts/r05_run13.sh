#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( bash scripts/ab_opts2.sh kitchen "" "--fuse-set 31" "--shadow-split 8" "--fuse-set 31 --shadow-split 8"
  bash scripts/ab_opts2.sh conference "" "--shadow-split 8" "--shadow-split 12"
  bash scripts/ab_opts2.sh courtyard-1440p "" "--shadow-split 8" ) 2>&1 | tee gpurun_out/r05_options_ab.txt
