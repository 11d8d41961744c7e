#!/bin/bash
# Round-2 GPU call A: baseline + knobs that were never measured on the device (MB_BAND, WH_DB_FAST_LINES candidate).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02a; mkdir -p $o
x=$o/experiments.txt; : > $x
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"],1), "frames/s", d["roofline"].get("events_ms"))'
run() { echo "== $*" >> $x; ( env "$@" timeout 90 python bench.py --no-cpu-baseline 2>>$o/err.txt | python -c "$pr" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_MB_BAND=2
run WELSHIP_MB_BAND=3
run WELSHIP_MB_BAND=4
run WELSHIP_MB_BAND=6
run WELSHIP_MB_BAND=8
run WELSHIP_MB_BAND=6 WELSHIP_P_WAVES=12
cand=$(python -c "from openh264_amd import build as B; print(B.build_hip(verbose=False, defines=('WH_DB_FAST_LINES',), tag='wh_db_fast_lines'))")
echo "== candidate $cand: parity" >> $x; timeout 300 python tools/fuzz_parity.py --lib $cand --cases 24 --seed 7 2>&1 | tail -1 >> $x
run WELSHIP_LIB=$cand
run WELSHIP_LIB=$cand WELSHIP_MB_BAND=4
cat $x
