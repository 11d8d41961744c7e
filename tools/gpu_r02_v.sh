#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02v; mkdir -p $o
cand=openh264_amd/libwelship_wh_no_cwin.so
x=$o/experiments.txt; : > $x
echo "== candidate $cand: parity" >> $x; timeout 600 python tools/fuzz_parity.py --lib $cand --cases 30 --seed 7 2>&1 | tail -1 >> $x
run() { echo "== $*" >> $x; ( env "$@" timeout 120 python bench.py --quick --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" ) >> $x 2>&1; }
run WELSHIP_NOP=1
run WELSHIP_LIB=$cand
echo "== res clip default" >> $x; timeout 200 python bench.py --quick --steps 40 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
echo "== res clip candidate" >> $x; WELSHIP_LIB=$cand timeout 200 python bench.py --quick --steps 40 --content res 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'frames/s', d['roofline'].get('events_ms'))" >> $x 2>&1
cat $x
for lib in openh264_amd/libwelship.so $cand; do
  WELSHIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_$(basename $lib .so) -- python bench.py --quick --steps 2 --warmup 1 > /dev/null 2>&1
  python tools/pmc_summary.py $o/pmc_$(basename $lib .so) | grep "k_inter_pool"
done
