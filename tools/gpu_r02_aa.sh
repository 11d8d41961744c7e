#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02aa; mkdir -p $o
timeout 900 python -m pytest tests/test_hooks_sha1.py tests/test_hooks_simulcast.py tests/test_zz_dropin_gpu.py -m gpu -q -x > $o/pytest_hooks.txt 2>&1; tail -3 $o/pytest_hooks.txt
for n in 1 2 4 8 16 32; do echo "== $n sessions"; WELSHIP_TRACE=1 timeout 300 python tools/config5_sessions.py $n 90 2>$o/err$n.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"], "C path sum", d["reference_c_path"]["sum_of_session_encode_fps"], "same", d["same_bitstreams"])'; done 2>&1 | tee $o/config5_batched.txt
echo "== 8 sessions, gather 0 / 300 us" | tee -a $o/config5_batched.txt
for g in 0 300; do WELSHIP_FRAME_GATHER_US=$g timeout 300 python tools/config5_sessions.py 8 90 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"])'; done 2>&1 | tee -a $o/config5_batched.txt
