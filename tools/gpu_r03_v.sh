#!/bin/bash
# Round 3, 23rd run: where a picture's time goes in config 5 at 1080p through the binding (1 and 8 sessions)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_v; rm -rf $o; mkdir -p $o
for n in 1 8; do
  WELS_HIP_TRACE=2 WELSHIP_FRAME_STATS=1 timeout 200 python tools/config5_sessions.py $n 30 plain 1080p > $o/config5_$n.json 2> $o/config5_$n.err
  echo "== $n sessions"; cut -c1-400 $o/config5_$n.json; sort $o/config5_$n.err | uniq -c | sort -rn | head -12 | cut -c1-300
done 2>&1 | tee $o/summary.txt
