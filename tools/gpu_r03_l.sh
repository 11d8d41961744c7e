#!/bin/bash
# Round 3, twelfth run: which HIP streams share a hardware queue?  The pipelined leg with different queue choices.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_l; rm -rf $o; mkdir -p $o
run() { name=$1; shift; echo "== $name" | tee -a $o/variants.txt; env "$@" timeout 120 python tools/e2e_pipe_run.py 256 24 2>&1 | grep -v amdgpu.ids | tee -a $o/variants.txt; }
run default WELSHIP_X=1
run queues_1_2 WELSHIP_PIPE_QUEUES=1,2
run queues_5_10 WELSHIP_PIPE_QUEUES=5,10
run prio WELSHIP_STREAM_PRIO=1
run hwq16 GPU_MAX_HW_QUEUES=16
run hwq16_prio GPU_MAX_HW_QUEUES=16 WELSHIP_STREAM_PRIO=1
run hwq2 GPU_MAX_HW_QUEUES=2
