#!/bin/bash
# Round 3, eighteenth run: A/B of the P-kernel candidate WH_FLAT_NB_LOADS (a macroblock's neighbour loads as one batch)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_r; rm -rf $o; mkdir -p $o
timeout 200 python tools/fuzz_parity.py --lib openh264_amd/libwelship_flatnb.so --cases 60 --seed 21 2>&1 | tail -2 | tee $o/fuzz_flatnb.txt
for v in base flatnb base flatnb; do
  lib=openh264_amd/libwelship.so; [ $v = flatnb ] && lib=openh264_amd/libwelship_flatnb.so
  WELSHIP_LIB=$PWD/$lib timeout 120 python bench.py --quick --steps 60 > $o/bench_$v.json 2> $o/bench_$v.err
  echo "$v: $(python -c "import json; d=json.loads(open('$o/bench_$v.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/ab.txt
done
