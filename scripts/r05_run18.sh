#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( bash scripts/ab_opts2.sh kitchen "--overlap 2" "--overlap 1" "--overlap 2 --shadow-split 8" "--overlap 1 --shadow-split 8" "--overlap 1 --shadow-split 12"
  bash scripts/ab_opts2.sh conference "--overlap 2" "--overlap 1" "--overlap 1 --shadow-split 12" ) 2>&1 | tee gpurun_out/r05_overlap_split_ab.txt
