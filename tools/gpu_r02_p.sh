#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02p; mkdir -p $o
x=$o/experiments.txt; : > $x
run() { echo "== $*" >> $x; ( env "$@" timeout 120 python bench.py --quick --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_P_WAVES=10
run WELSHIP_P_WAVES=8
run WELSHIP_MD_SLOTS=1 WELSHIP_P_WAVES=6
echo "== res clip" >> $x; timeout 200 python bench.py --quick --steps 40 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
echo "== res clip P_WAVES=10" >> $x; WELSHIP_P_WAVES=10 timeout 200 python bench.py --quick --steps 40 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
cat $x
