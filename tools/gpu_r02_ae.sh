#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ae; mkdir -p $o
x=$o/e2e_pool.txt; : > $x
for cfg in "3 64 32" "4 64 32" "3 128 32" "4 128 32" "3 64 48" "6 64 24"; do
  set -- $cfg
  echo "== groups $1 x $2 sessions, $3 host threads per group" >> $x
  timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 --e2e-groups $1 --e2e-group-sessions $2 --host-threads $3 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["e2e_overlapped"]["frames_per_s"]), "sync:", round(d["e2e"]["frames_per_s"]), "latency1", round(d["latency"]["sessions_1"]["ms_per_frame"],2))' >> $x
done
cat $x
