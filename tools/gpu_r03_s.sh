#!/bin/bash
# Round 3, nineteenth run: deblocking bands that ignore the slices (idc 0: the filter crosses slice edges anyway): 2 x 34 rows, 3 x 23 rows
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_s; rm -rf $o; mkdir -p $o
run() { name=$1; shift
  env "$@" timeout 120 python bench.py --quick --steps 60 > $o/bench_$name.json 2> $o/bench_$name.err
  echo "$name: $(python -c "import json; d=json.loads(open('$o/bench_$name.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/db_bands.txt; }
run slices_17_rows_12_waves WELSHIP_X=1
run whole_34_rows_12_waves WELSHIP_DB_BY_SLICE=0 WELSHIP_DB_BAND_ROWS=34
run whole_34_rows_16_waves WELSHIP_DB_BY_SLICE=0 WELSHIP_DB_BAND_ROWS=34 WELSHIP_DB_WAVES=16
run whole_23_rows_12_waves WELSHIP_DB_BY_SLICE=0 WELSHIP_DB_BAND_ROWS=23
run whole_68_rows_16_waves WELSHIP_DB_BY_SLICE=0 WELSHIP_DB_BAND_ROWS=68 WELSHIP_DB_WAVES=16
WELSHIP_DB_BY_SLICE=0 WELSHIP_DB_BAND_ROWS=34 timeout 300 python -m pytest tests/test_frame_parity.py tests/test_fuzz_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $o/parity_34_rows.txt
