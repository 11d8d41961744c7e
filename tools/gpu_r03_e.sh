#!/bin/bash
# Round 3, fifth run: where the run scheduler loses time (variants), the pre-processing on the device (down-sampling + statistics), configs 4 / 5 at BASELINE's sizes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_e; rm -rf $o; mkdir -p $o
t0=$(date +%s); lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
q() { # name, env...
  name=$1; shift
  env "$@" timeout 120 python bench.py --quick --steps 60 > $o/bench_quick_$name.json 2> $o/bench_quick_$name.err
  echo "$name: $(python -c "import json; d=json.loads(open('$o/bench_quick_$name.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"
}
q run4 WELSHIP_MD_RUN=4
q run1_rows_kernel WELSHIP_MD_ROWS=1 WELSHIP_MD_RUN=1
q run4_fetch_after_mb WELSHIP_MD_RUN=4 WELSHIP_LIB=$PWD/openh264_amd/libwelship_slidetop.so
q run2 WELSHIP_MD_RUN=2
q tickets WELSHIP_MD_ROWS=0
lap "variants"
timeout 300 python -m pytest tests/test_vaa.py tests/test_hooks_simulcast.py tests/test_downsample_gpu.py -m gpu -q > $o/pytest_preproc.txt 2>&1; tail -3 $o/pytest_preproc.txt; lap "pre-processing tests"
WELS_HIP_CHECK_VAA=1 timeout 300 python -m pytest tests/test_hooks_sha1.py -m gpu -q -k "table_rows_on_the_mi355x" > $o/pytest_sha1_check_vaa.txt 2>&1; tail -3 $o/pytest_sha1_check_vaa.txt; lap "sha1 rows with the device's pre-analysis checked against the reference's"
for dsv in 1 0; do
  WELS_HIP_DOWNSAMPLE=$dsv WELS_HIP_VAA=$dsv timeout 200 python tools/config5_sessions.py 1 30 simulcast 1080p > $o/config4_one_session_preproc$dsv.json 2> $o/config4_one_session_preproc$dsv.err; cut -c1-900 $o/config4_one_session_preproc$dsv.json
done
timeout 200 python tools/config5_sessions.py 8 30 simulcast 1080p > $o/config4_8_sessions.json 2> $o/config4_8.err; cut -c1-900 $o/config4_8_sessions.json
timeout 200 python tools/config5_sessions.py 8 30 plain 1080p > $o/config5_8_sessions_1080p.json 2> $o/config5_8.err; cut -c1-900 $o/config5_8_sessions_1080p.json
lap "configs 4 and 5"
