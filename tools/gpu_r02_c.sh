#!/bin/bash
# Round-2 GPU call C: write-through seam hand-off (no agent fences), band geometry A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02c; mkdir -p $o
timeout 600 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
x=$o/experiments.txt; : > $x
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "frames/s", d["roofline"].get("events_ms"))'
run() { echo "== $*" >> $x; ( env "$@" timeout 90 python bench.py --no-cpu-baseline $EXTRA 2>>$o/err.txt | python -c "$pr" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_DB_BY_SLICE=1 WELSHIP_DB_BAND_ROWS=64
run WELSHIP_DB_BY_SLICE=1
run WELSHIP_DB_BAND_ROWS=14
run WELSHIP_DB_BAND_ROWS=12
run WELSHIP_DB_BAND_ROWS=8
EXTRA="--deblock-idc 2"
run WELSHIP_NOP=1
run WELSHIP_DB_BAND_ROWS=64
EXTRA="--sessions 8"
run WELSHIP_NOP=1
run WELSHIP_DB_BAND_ROWS=8
cat $x
