#!/bin/bash
# Round 3, 22nd run: the P kernel's time before / after the one-band deblocking change (kernel argument layout unchanged now)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_u; rm -rf $o; mkdir -p $o
for v in a b; do
  timeout 120 python bench.py --quick --steps 60 > $o/bench_$v.json 2> $o/bench_$v.err
  echo "$v: $(python -c "import json; d=json.loads(open('$o/bench_$v.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/ab.txt
done
