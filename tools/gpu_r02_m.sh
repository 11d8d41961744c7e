#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02m; mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -4 $o/pytest_gpu.txt
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/time.txt; tail -c 5000 $o/bench_default.json; tail -3 $o/bench_default.err; cat $o/time.txt
WELSHIP_COMPACT=0 timeout 300 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("no compaction: e2e", d["e2e"])'
