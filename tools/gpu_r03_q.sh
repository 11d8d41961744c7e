#!/bin/bash
# Round 3, seventeenth run: the pipelined group 1..3 steps ahead, one or two upload queues
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_q; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests/test_multi_rank.py -m gpu -x -q 2>&1 | tail -3 | tee $o/pytest_pipelined.txt
run() { name=$1; shift; echo "== $name" | tee -a $o/variants.txt; env "$@" timeout 120 python tools/e2e_pipe_run.py 256 60 12 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $o/variants.txt; }
run ahead1 WELSHIP_PIPE_AHEAD=1
run ahead2 WELSHIP_PIPE_AHEAD=2
run ahead3 WELSHIP_PIPE_AHEAD=3
run ahead1_two_upload_queues WELSHIP_PIPE_AHEAD=1 WELSHIP_PIPE_QUEUES=1,2,3
run ahead2_two_upload_queues WELSHIP_PIPE_AHEAD=2 WELSHIP_PIPE_QUEUES=1,2,3
run ahead2_two_upload_queues_5 WELSHIP_PIPE_AHEAD=2 WELSHIP_PIPE_QUEUES=1,2,5
WELSHIP_PIPE_AHEAD=2 WELSHIP_PIPE_TRACE=1 timeout 120 python tools/e2e_pipe_run.py 256 20 12 2>&1 | grep "welship pipe" | tail -4 | tee $o/host_trace_ahead2.txt
