#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/ab.sh "--workload kitchen" shipped mb6 mb7 > gpurun_out/r05_logic_occupancy_ab.txt 2>&1
bash scripts/ab.sh "--workload conference" shipped mb6 mb7 >> gpurun_out/r05_logic_occupancy_ab.txt 2>&1
cat gpurun_out/r05_logic_occupancy_ab.txt
