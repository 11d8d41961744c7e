#!/bin/bash
# Round 3, last GPU seconds: the PLAIN variant of the P kernel (-DWH_PLAIN_KERNEL=1: 30 k instead of 44 k instructions) against the default, same box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_plain; rm -rf $o; mkdir -p $o
for v in default plain default plain; do
  lib=openh264_amd/libwelship.so; [ $v = plain ] && lib=openh264_amd/libwelship_plain.so
  WELSHIP_LIB=$PWD/$lib timeout 60 python bench.py --quick --steps 40 > $o/bench_$v.json 2> $o/bench_$v.err
  echo "$v: $(python -c "import json; d=json.loads(open('$o/bench_$v.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/ab.txt
done
