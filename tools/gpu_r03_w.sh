#!/bin/bash
# Round 3, 24th run: the job descriptor in registers (WH_JOB_REGS) against the LDS copy, same box
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_w; rm -rf $o; mkdir -p $o
timeout 200 python tools/fuzz_parity.py --lib openh264_amd/libwelship.so --cases 60 --seed 33 2>&1 | tail -1 | tee $o/fuzz.txt
for v in regs lds regs lds; do
  lib=openh264_amd/libwelship.so; [ $v = lds ] && lib=openh264_amd/libwelship_nojobregs.so
  WELSHIP_LIB=$PWD/$lib timeout 120 python bench.py --quick --steps 60 > $o/bench_$v.json 2> $o/bench_$v.err
  echo "$v: $(python -c "import json; d=json.loads(open('$o/bench_$v.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/ab.txt
done
WELSHIP_LIB=$PWD/openh264_amd/libwelship.so timeout 120 python bench.py --quick --steps 40 --content res > $o/bench_res.json 2> $o/bench_res.err; python -c "import json; d=json.loads(open('$o/bench_res.json').read().strip().splitlines()[-1]); print('res clip regs', round(d['value']), d['roofline']['events_ms'])" | tee -a $o/ab.txt
WELSHIP_LIB=$PWD/openh264_amd/libwelship_nojobregs.so timeout 120 python bench.py --quick --steps 40 --content res > $o/bench_res2.json 2> $o/bench_res2.err; python -c "import json; d=json.loads(open('$o/bench_res2.json').read().strip().splitlines()[-1]); print('res clip lds', round(d['value']), d['roofline']['events_ms'])" | tee -a $o/ab.txt
