#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02t; mkdir -p $o
timeout 600 python -m pytest tests/test_frame_parity.py tests/test_fuzz_parity.py tests/test_reference_content.py -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
x=$o/experiments.txt; : > $x
run() { echo "== $*" >> $x; ( env "$@" timeout 120 python bench.py --quick --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" ) >> $x 2>&1; }
run WELSHIP_NOP=1
cat $x
