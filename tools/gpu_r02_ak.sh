#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ak; mkdir -p $o
for g in 2000 4000 8000 16000; do for n in 8 16; do echo "== gather $g us, $n simulcast sessions"; WELSHIP_TRACE=1 WELSHIP_FRAME_GATHER_US=$g timeout 400 python tools/config5_sessions.py $n 54 simulcast 2>$o/err_${g}_$n.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"], "same", d["same_bitstreams"])'; done; done 2>&1 | tee $o/gather.txt
echo "== gather 8000, 8 x 720p"; WELSHIP_FRAME_GATHER_US=8000 timeout 300 python tools/config5_sessions.py 8 90 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"])' | tee -a $o/gather.txt
