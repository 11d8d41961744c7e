#!/bin/bash
# Round 3, the FIRST run on the MI355X: what round 2 left unmeasured (its GPU budget was spent before these were written).
# Build the candidate libraries in the container first (they travel with the snapshot):
#   python -c "from openh264_amd import build as B; B.build_hip(); B.build_hip(defines=('WH_FLAT_NB_LOADS=1',), tag='flatnb'); B.build_hip(defines=('WH_EARLY_CLAIM=1',), tag='early')"
# 1. GPU tier + smoke at HEAD
# 2. size-limited slices (WELS_HIP_DYNSLICE=1) on the device for the first time: the SHA1 table's 512 rows, random sessions
# 3. A/B of the P-kernel candidates that passed the emulation: one-batch neighbour loads (WH_FLAT_NB_LOADS), early claim
# 4. the default bench line (the e2e legs with the host entropy coder that reads the packed records in place; `host_thread_ms_per_picture`)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_first; rm -rf $o; mkdir -p $o
timeout 120 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt
timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
WELSHIP_TEST_UNVERIFIED=1 timeout 120 python -m pytest tests/test_hooks_dynslice.py -m gpu -q > $o/pytest_dynslice_gpu.txt 2>&1; tail -2 $o/pytest_dynslice_gpu.txt
timeout 200 python tools/sha1_table_rows.py --dynslice --workers 16 > $o/size_limited_rows_mi355x.txt 2>&1; tail -4 $o/size_limited_rows_mi355x.txt
timeout 200 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 60 --seed 7 --workers 16 > $o/fuzz_dynslice_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_mi355x.txt
timeout 200 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 40 --seed 5 --threads 4 --workers 4 > $o/fuzz_dynslice_threads_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_threads_mi355x.txt
timeout 200 python tools/sha1_table_rows.py --table adobe --dynslice --workers 16 --stride 4 > $o/size_limited_rows_screen_mi355x.txt 2>&1; tail -4 $o/size_limited_rows_screen_mi355x.txt
timeout 200 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 40 --seed 21 --screen --workers 16 > $o/fuzz_dynslice_screen_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_screen_mi355x.txt
timeout 200 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 24 --seed 11 --low-qp --workers 16 > $o/fuzz_dynslice_lowqp_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_lowqp_mi355x.txt
WELSHIP_FRAME_STATS=1 timeout 120 python tools/config5_sessions.py 8 40 dynslice > $o/config5_dynslice_8sessions.json 2> $o/config5_dynslice.err; cut -c1-700 $o/config5_dynslice_8sessions.json
for tag in "" _flatnb _early; do
  lib=$PWD/openh264_amd/libwelship$tag.so; [ -f $lib ] || continue
  WELSHIP_LIB=$lib timeout 60 python bench.py --quick --steps 60 > $o/bench_quick$tag.json 2> $o/bench_quick$tag.err
  echo "quick${tag:-_default}: $(python -c "import json,sys; d=json.loads(open('$o/bench_quick$tag.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"
done
for tag in _flatnb; do      # a candidate only counts when it is still bit-exact on the device
  lib=$PWD/openh264_amd/libwelship$tag.so; [ -f $lib ] || continue
  timeout 120 python tools/fuzz_parity.py --lib $lib --cases 40 --seed 3 > $o/fuzz$tag.txt 2>&1; tail -1 $o/fuzz$tag.txt
done
( time timeout 240 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; tail -c 400 $o/bench_default.json; tail -3 $o/bench_default.time
