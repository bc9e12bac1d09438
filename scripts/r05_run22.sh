#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( free -g | head -3; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null; nproc
  bash scripts/ab_opts2.sh kitchen "" "--num-tasks 16777216" "--num-tasks 25165824" "--num-tasks 33554432"
  bash scripts/ab_opts2.sh courtyard-1440p "" "--num-tasks 16777216"
  bash scripts/ab_opts2.sh egyptcat "" "--num-tasks 16777216" ) 2>&1 | tee gpurun_out/r05_num_tasks2.txt
