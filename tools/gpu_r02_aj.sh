#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02aj; mkdir -p $o
timeout 900 python -m pytest tests/test_hooks_sha1.py tests/test_hooks_simulcast.py -m gpu -q -x 2>&1 | tail -2
for n in 1 2 4 8 16; do echo "== $n simulcast sessions (1080p -> 4 layers)"; timeout 400 python tools/config5_sessions.py $n 54 simulcast 2>$o/err$n.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"], "C path", d["reference_c_path"]["sum_of_session_encode_fps"], d["reference_c_path"]["min_session_fps"], "same", d["same_bitstreams"])'; done 2>&1 | tee $o/config4_simulcast.txt
for n in 8 32; do echo "== $n 720p sessions"; timeout 300 python tools/config5_sessions.py $n 90 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"], "same", d["same_bitstreams"])'; done 2>&1 | tee -a $o/config4_simulcast.txt
