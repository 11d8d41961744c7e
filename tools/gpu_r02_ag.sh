#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02ag; mkdir -p $o
x=$o/numa.txt; : > $x
lscpu | grep -i "numa\|socket\|model name" >> $x
for d in /sys/class/drm/card*/device; do echo "$d numa_node $(cat $d/numa_node 2>/dev/null)" >> $x; done
which numactl taskset >> $x 2>&1
for cpus in "all" "0-63,128-191" "64-127,192-255" "0-63" "64-127"; do
  echo "== cpus $cpus" >> $x
  if [ "$cpus" = all ]; then pre=""; else pre="taskset -c $cpus"; fi
  $pre timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["e2e_overlapped"]["frames_per_s"]), "sync:", round(d["e2e"]["frames_per_s"]), "lat1", round(d["latency"]["sessions_1"]["ms_per_frame"],2))' >> $x
done
cat $x
