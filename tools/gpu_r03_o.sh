#!/bin/bash
# Round 3, fifteenth run: do running copies slow the kernels?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_o; rm -rf $o; mkdir -p $o
timeout 200 python tools/micro/bench_under_copies.py 2>&1 | grep -v amdgpu.ids | tee $o/bench_under_copies.txt
