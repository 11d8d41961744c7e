#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02e; mkdir -p $o
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/time.txt; tail -c 6000 $o/bench_default.json; tail -5 $o/bench_default.err; cat $o/time.txt
