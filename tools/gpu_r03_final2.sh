#!/bin/bash
# Round 3, last evidence run: the camera SHA1 table (all 1792 device rows) at the round's last kernel code, the GPU tier, the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_final2; rm -rf $o; mkdir -p $o
T0=$SECONDS; lap() { echo "[$((SECONDS - T0)) s] $1"; }
timeout 330 python tools/sha1_table_rows.py --workers 16 > $o/camera_table_1792_rows.txt 2>&1; tail -3 $o/camera_table_1792_rows.txt | cut -c1-220; lap "camera table"
timeout 300 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -3 $o/pytest_gpu.txt; lap "gpu tier"
timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "verified", d.get("verified"), d["roofline"]["events_ms"])
for k in ("e2e", "e2e_overlapped", "e2e_pipelined"):
    print(k, {x: d[k][x] for x in d[k] if x in ("frames_per_s", "frames_per_s_second_half", "steps_ahead")}, d[k]["bitstream_vs_reference"]["match"])
for k in d:
    if k.startswith("config") and k != "config": print(k, d[k].get("device_frames_per_s"), d[k].get("c_path_frames_per_s"), d[k].get("same_bitstreams"))
PY
lap "bench default"
