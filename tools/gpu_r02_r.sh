#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02r; mkdir -p $o
x=$o/e2e_sweep.txt; : > $x
for cfg in "2 64" "3 64" "4 64" "6 64" "2 128" "3 128" "4 128" "3 96" "8 32"; do
  set -- $cfg
  echo "== groups $1 x $2 sessions" >> $x
  GPU_MAX_HW_QUEUES=${HWQ:-4} timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 --e2e-groups $1 --e2e-group-sessions $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["e2e_overlapped"]["frames_per_s"]), "sync:", round(d["e2e"]["frames_per_s"]))' >> $x
done
echo "== groups 3 x 64, GPU_MAX_HW_QUEUES=8" >> $x
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 --e2e-groups 3 --e2e-group-sessions 64 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["e2e_overlapped"]["frames_per_s"]))' >> $x
cat $x
timeout 600 python tools/config5_sessions.py 8 60 > $o/config5_8sessions.json 2>$o/config5.err; cat $o/config5_8sessions.json; tail -2 $o/config5.err
