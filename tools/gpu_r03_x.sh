#!/bin/bash
# Round 3: the frame API's ranged calls wait outside the device-wide lock -- the tests that drive them from several threads, on the MI355X
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_x; rm -rf $o; mkdir -p $o
timeout 400 python -m pytest tests/test_hooks_dynslice.py tests/test_hooks_sha1.py tests/test_hooks_cabac_threads.py tests/test_hooks_screen.py tests/test_frame_api_retry.py -m gpu -q -n 4 2>&1 | tail -3 | tee $o/pytest_hooks.txt
timeout 100 python tools/config5_sessions.py 8 30 gom > $o/gom8.json 2> $o/gom8.err; cut -c1-600 $o/gom8.json
