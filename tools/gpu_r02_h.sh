#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02h; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
x=$o/experiments.txt; : > $x
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "frames/s", d["roofline"].get("events_ms"), d.get("verified"))'
run() { echo "== $EXTRA $*" >> $x; ( env "$@" timeout 200 python bench.py --no-extra --no-cpu-baseline --steps 30 $EXTRA 2>>$o/err.txt | python -c "$pr" ) >> $x 2>&1; }
EXTRA="--content res"
run WELSHIP_NOP=1
run WELSHIP_MD_ASSIGN=0
run WELSHIP_MD_SLOTS=1
run WELSHIP_MD_SLOTS=1 WELSHIP_P_WAVES=6
EXTRA="--content res --sessions 256"
run WELSHIP_NOP=1
run WELSHIP_MD_SLOTS=2
EXTRA=""
run WELSHIP_NOP=1
run WELSHIP_MD_ASSIGN=0
EXTRA="--sessions 8"
run WELSHIP_NOP=1
cat $x
