#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02g; mkdir -p $o
x=$o/experiments.txt; : > $x
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "frames/s", d["roofline"].get("events_ms"))'
run() { echo "== $*" >> $x; ( env "$@" timeout 120 python bench.py --quick --steps 30 --content res $EXTRA 2>>$o/err.txt | python -c "$pr" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_P_WAVES=12
EXTRA="--sessions 192"; run WELSHIP_NOP=1; run WELSHIP_P_WAVES=12
EXTRA="--sessions 256"; run WELSHIP_NOP=1; run WELSHIP_P_WAVES=12; run WELSHIP_P_WAVES=6
EXTRA="--sessions 384"; run WELSHIP_NOP=1; run WELSHIP_P_WAVES=6
EXTRA="--sessions 64"; run WELSHIP_NOP=1
cat $x
