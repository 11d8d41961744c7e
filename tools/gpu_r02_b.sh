#!/bin/bash
# Round-2 GPU call B: deblocking bands -- parity tier + bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02b; mkdir -p $o
timeout 600 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
x=$o/experiments.txt; : > $x
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "frames/s", d["roofline"].get("events_ms"))'
run() { echo "== $*" >> $x; ( env "$@" timeout 90 python bench.py --no-cpu-baseline 2>>$o/err.txt | python -c "$pr" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_DB_WAVES=12
echo "== idc2" >> $x; timeout 90 python bench.py --no-cpu-baseline --deblock-idc 2 2>>$o/err.txt | python -c "$pr" >> $x 2>&1
echo "== s8" >> $x; timeout 90 python bench.py --no-cpu-baseline --sessions 8 2>>$o/err.txt | python -c "$pr" >> $x 2>&1
echo "== s64" >> $x; timeout 90 python bench.py --no-cpu-baseline --sessions 64 2>>$o/err.txt | python -c "$pr" >> $x 2>&1
cat $x
