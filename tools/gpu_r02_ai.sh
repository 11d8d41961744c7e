#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ai; mkdir -p $o
for n in 1 2 4 8 16; do echo "== $n simulcast sessions (1080p -> 4 layers)"; timeout 400 python tools/config5_sessions.py $n 54 simulcast 2>$o/err$n.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"], "C path", d["reference_c_path"]["sum_of_session_encode_fps"], d["reference_c_path"]["min_session_fps"], "same", d["same_bitstreams"])'; done 2>&1 | tee $o/config4_simulcast.txt
