#!/bin/bash
# Round 3, the FIRST run on the MI355X: everything round 2 left verified on the CPU test build only, on libwelship.so:
#   1. GPU tier + smoke at HEAD                          2. size-limited slices (WELS_HIP_DYNSLICE=1): both tables' -slcmd 3 rows, four fuzz sets
#   3. the screen-content table, all 896 device rows     4. GOM-level QP inside the kernel (--gom 2) over the camera table's device rows
#   5. calibration of the traffic counters for 64-byte rows (tools/micro/l2_granule)          6. quick A/B + the default bench line
# Build first (in the container; the binaries travel with the snapshot):
#   python -c "from openh264_amd import build as B; B.build_hip(); B.build_hip(defines=('WH_FLAT_NB_LOADS=1',), tag='flatnb')"
#   (cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o l2_granule l2_granule.hip)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_first; rm -rf $o; mkdir -p $o
t0=$(date +%s); lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
timeout 240 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt; lap "gpu tier"
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.txt 2>&1; tail -1 $o/smoke.txt
WELSHIP_TEST_UNVERIFIED=1 timeout 150 python -m pytest tests/test_hooks_dynslice.py -m gpu -q > $o/pytest_dynslice_gpu.txt 2>&1; tail -2 $o/pytest_dynslice_gpu.txt; lap "dynslice tests"
timeout 240 python tools/sha1_table_rows.py --dynslice --workers 16 > $o/size_limited_rows_mi355x.txt 2>&1; tail -4 $o/size_limited_rows_mi355x.txt; lap "camera table, size-limited rows"
timeout 150 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 60 --seed 7 --workers 16 > $o/fuzz_dynslice_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_mi355x.txt
timeout 150 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 40 --seed 5 --threads 4 --workers 4 > $o/fuzz_dynslice_threads_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_threads_mi355x.txt
timeout 150 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 40 --seed 21 --screen --workers 16 > $o/fuzz_dynslice_screen_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_screen_mi355x.txt
timeout 150 python tools/fuzz_dynslice.py --lib openh264_amd/libwelship.so --cases 24 --seed 11 --low-qp --workers 16 > $o/fuzz_dynslice_lowqp_mi355x.txt 2>&1; tail -1 $o/fuzz_dynslice_lowqp_mi355x.txt; lap "fuzz sets"
timeout 420 python tools/sha1_table_rows.py --table adobe --workers 16 > $o/screen_table_all_device_rows_mi355x.txt 2>&1; tail -4 $o/screen_table_all_device_rows_mi355x.txt; lap "screen table, 896 rows"
timeout 300 python tools/sha1_table_rows.py --table adobe --dynslice --workers 16 > $o/size_limited_rows_screen_mi355x.txt 2>&1; tail -4 $o/size_limited_rows_screen_mi355x.txt; lap "screen table, size-limited rows"
timeout 300 python tools/sha1_table_rows.py --gom 2 --workers 16 > $o/camera_table_gom2_mi355x.txt 2>&1; tail -5 $o/camera_table_gom2_mi355x.txt; lap "camera table, GOM QP inside the kernel"
timeout 120 tools/micro/l2_granule > $o/l2_granule.txt 2>&1; cat $o/l2_granule.txt
( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$o/l2_granule_fetch -- $OLDPWD/tools/micro/l2_granule > $OLDPWD/$o/l2_granule_fetch.log 2>&1 )
python tools/pmc_summary.py $o/l2_granule_fetch > $o/l2_granule_fetch_summary.txt 2>&1; python - <<PY
import csv, glob
for f in glob.glob("$o/l2_granule_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("dispatch", r["Dispatch_Id"], r["Counter_Name"], r["Counter_Value"])
PY
lap "calibration"
for tag in "" _flatnb; do
  lib=$PWD/openh264_amd/libwelship$tag.so; [ -f $lib ] || continue
  WELSHIP_LIB=$lib timeout 90 python bench.py --quick --steps 60 > $o/bench_quick$tag.json 2> $o/bench_quick$tag.err
  echo "quick${tag:-_default}: $(python -c "import json,sys; d=json.loads(open('$o/bench_quick$tag.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"
done
( time timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; tail -c 600 $o/bench_default.json; tail -3 $o/bench_default.time; lap "bench"
