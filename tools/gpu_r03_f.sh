#!/bin/bash
# Round 3, sixth run: is it the sliding or the run scheduling that costs time; sub-phase profile of claim / neighbour loads / P_Skip test.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_f; rm -rf $o; mkdir -p $o
t0=$(date +%s); lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
q() { name=$1; shift
  env "$@" timeout 120 python bench.py --quick --steps 60 > $o/bench_quick_$name.json 2> $o/bench_quick_$name.err
  echo "$name: $(python -c "import json; d=json.loads(open('$o/bench_quick_$name.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"; }
q tickets_default WELSHIP_X=1
q run4_slides WELSHIP_MD_RUN=4
q run4_no_slides WELSHIP_MD_RUN=4 WELSHIP_LIB=$PWD/openh264_amd/libwelship_noslide.so
q run8_no_slides WELSHIP_MD_RUN=8 WELSHIP_LIB=$PWD/openh264_amd/libwelship_noslide.so
lap "variants"
WELSHIP_LIB=$PWD/openh264_amd/libwelship_profdetail.so WELSHIP_PROF_DETAIL=1 timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles_detail_tickets.txt 2>&1; head -20 $o/phase_cycles_detail_tickets.txt
WELSHIP_LIB=$PWD/openh264_amd/libwelship_profdetail.so WELSHIP_PROF_DETAIL=1 timeout 200 python tools/phase_profile.py 256 res > $o/phase_cycles_detail_tickets_res.txt 2>&1; head -20 $o/phase_cycles_detail_tickets_res.txt | tail -16
lap "detail profile"
