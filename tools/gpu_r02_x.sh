#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02x; mkdir -p $o
for q in 4 8 16; do echo "== GPU_MAX_HW_QUEUES=$q, 8 sessions"; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/config5_sessions.py 8 60 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"])'; done 2>&1 | tee $o/config5_hwq.txt
for n in 2 4; do echo "== default queues, $n sessions"; timeout 300 python tools/config5_sessions.py $n 60 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["hooks_on_device"])'; done 2>&1 | tee -a $o/config5_hwq.txt
