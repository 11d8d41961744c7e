#!/bin/bash
# Round 3, tenth run: timeline of the pipelined end-to-end leg (kernels / H2D / D2H: what overlaps what) + the host's view of every call
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_j; rm -rf $o; mkdir -p $o
R=$PWD
( cd /tmp && WELSHIP_PIPE_TRACE=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$o/trace -- python $R/tools/e2e_pipe_run.py 256 8 > $R/$o/run.txt 2> $R/$o/run.err )
cat $o/run.txt; grep "welship pipe" $o/run.err | tail -6
python tools/trace_timeline.py $o/trace 4 1.0 > $o/timeline.txt; cat $o/timeline.txt
rm -rf $o/trace
timeout 200 python tools/e2e_pipe_run.py 256 12 | tee $o/untraced.txt
