#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ad; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q > $o/pytest_gpu.txt 2>&1; tail -2 $o/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; tail -c 300 $o/bench_default.json; tail -3 $o/bench_default.time
