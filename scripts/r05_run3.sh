#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
./variants/bin/lds_dma_probe > gpurun_out/r05_lds_dma_probe.txt 2>&1
cat gpurun_out/r05_lds_dma_probe.txt
./variants/bin/logic_stream > gpurun_out/r05_logic_stream.txt 2>&1
cat gpurun_out/r05_logic_stream.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "free_running_render_bit_identical and binary-shadow-serial-unfused-0" 2>&1 | grep -v "^$" | head -80 | cut -c1-400 > gpurun_out/r05_mlp_fail.log
head -60 gpurun_out/r05_mlp_fail.log
