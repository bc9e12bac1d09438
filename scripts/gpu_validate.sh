#!/bin/bash
# Round-end validation on the GPU box: the full -m gpu suite and smoke(); logs under gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r03_gpu_suite.log
echo "rc=$?" >> gpurun_out/r03_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1
tail -3 gpurun_out/r03_gpu_suite.log; tail -2 gpurun_out/r03_smoke.log
