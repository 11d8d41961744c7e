#!/bin/bash
# Round 3, thirteenth run: timeline of the pipelined group with its transfer queues on hardware queues of their own
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_m; rm -rf $o; mkdir -p $o
R=$PWD
( cd /tmp && WELSHIP_PIPE_TRACE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$o/trace -- python $R/tools/e2e_pipe_run.py 256 20 > $R/$o/run.txt 2> $R/$o/run.err )
cat $o/run.txt; grep "welship pipe" $o/run.err | tail -5
python tools/trace_timeline.py $o/trace 4 1.0 > $o/timeline.txt; cat $o/timeline.txt
rm -rf $o/trace
for t in 16 32 64; do timeout 200 python tools/e2e_pipe_run.py 256 30 $t 2>&1 | grep -v amdgpu.ids | tee -a $o/untraced.txt; done
