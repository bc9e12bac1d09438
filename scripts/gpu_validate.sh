#!/bin/bash
# Round-end validation on the GPU box: the full -m gpu suite and smoke(); logs under gpurun_out/.  Exits non-zero when either fails.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r04}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/${TAG}_gpu_suite.log
rc=$?                                   # pipefail: pytest's (or timeout's) status, not tail's
echo "rc=$rc" >> gpurun_out/${TAG}_gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1
rc2=$?
echo "rc=$rc2" >> gpurun_out/${TAG}_smoke.log
tail -3 gpurun_out/${TAG}_gpu_suite.log; tail -2 gpurun_out/${TAG}_smoke.log
[ $rc -eq 0 ] && [ $rc2 -eq 0 ]
