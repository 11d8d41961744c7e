#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02af; mkdir -p $o
x=$o/e2e_pool_ab.txt; : > $x
for pool in 1 0 1 0; do
  echo "== WELSHIP_POOL=$pool, 3 x 64" >> $x
  WELSHIP_POOL=$pool timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["e2e_overlapped"]["frames_per_s"]), "sync:", round(d["e2e"]["frames_per_s"]))' >> $x
done
cat $x
