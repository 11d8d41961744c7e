#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02q; mkdir -p $o
timeout 300 python -m pytest tests/test_multi_rank.py -m gpu -q > $o/pytest_groups.txt 2>&1; tail -2 $o/pytest_groups.txt
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/time.txt; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02q/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["events_ms"], "verified", d.get("verified"))
print("e2e", d.get("e2e"))
print("e2e_overlapped", d.get("e2e_overlapped"))
print("latency", d.get("latency"))
print("res", d.get("res_clip"))
PY
tail -3 $o/bench_default.err; cat $o/time.txt
for g in 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 20 --e2e-groups $g 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["e2e_overlapped"])'; done
timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 20 --e2e-groups 4 --host-threads 48 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["e2e_overlapped"], d["e2e"]["frames_per_s"])'
