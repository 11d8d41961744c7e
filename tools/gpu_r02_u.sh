#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02u; mkdir -p $o
cand=openh264_amd/libwelship_wh_db_row_pass.so
x=$o/experiments.txt; : > $x
echo "== candidate $cand: parity" >> $x; timeout 600 python tools/fuzz_parity.py --lib $cand --cases 60 --seed 7 2>&1 | tail -1 >> $x
run() { echo "== $*" >> $x; ( env "$@" timeout 120 python bench.py --quick --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_LIB=$cand
run WELSHIP_NOP=2
run WELSHIP_LIB=$cand WELSHIP_NOP=2
cat $x
WELSHIP_LIB=$cand timeout 300 python tools/phase_profile.py 128 2>&1 | grep -A14 "deblocking kernel"
