#!/bin/bash
# Round 3, twentieth run: one deblocking band per picture for large batches (default for >= 192 pictures per launch)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_t; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests/test_frame_parity.py tests/test_multi_rank.py -m gpu -q -x 2>&1 | tail -3 | tee $o/parity.txt
run() { name=$1; shift
  env "$@" timeout 120 python bench.py --quick --steps 60 $EXTRA > $o/bench_$name.json 2> $o/bench_$name.err
  echo "$name: $(python -c "import json; d=json.loads(open('$o/bench_$name.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/db_whole.txt; }
run default_256 WELSHIP_X=1
run by_slice_256 WELSHIP_DB_WHOLE=0
EXTRA="--sessions 192" run default_192 WELSHIP_X=1
EXTRA="--sessions 192" run by_slice_192 WELSHIP_DB_WHOLE=0
EXTRA="--sessions 128" run whole_128 WELSHIP_DB_WHOLE=1
EXTRA="--sessions 128" run by_slice_128 WELSHIP_DB_WHOLE=0
