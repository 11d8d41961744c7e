#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02y; mkdir -p $o
x=$o/sessions_sweep.txt; : > $x
for s in 128 192 256 320 384 512; do echo "== sessions $s" >> $x; timeout 200 python bench.py --quick --sessions $s --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1; done
echo "== res clip 256" >> $x; timeout 200 python bench.py --quick --sessions 256 --steps 30 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
echo "== res clip 128" >> $x; timeout 200 python bench.py --quick --sessions 128 --steps 30 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
cat $x
