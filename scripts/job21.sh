cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/ab.sh "" shipped wait32 wait24 wait16 wait8 > gpurun_out/r03_shadow_wait_ab.txt 2>&1
cat gpurun_out/r03_shadow_wait_ab.txt
bash scripts/ab.sh "--workload conference" shipped wait24 wait16 > gpurun_out/r03_shadow_wait_ab_conference.txt 2>&1
cat gpurun_out/r03_shadow_wait_ab_conference.txt
bash scripts/ab.sh "--workload courtyard-1440p" shipped wait24 wait16 > gpurun_out/r03_shadow_wait_ab_courtyard.txt 2>&1
cat gpurun_out/r03_shadow_wait_ab_courtyard.txt
