#!/bin/bash
# Round 3, last call: the frame API's queues with the hardware-queue classes they had before, now by construction; the default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_z2; rm -rf $o; mkdir -p $o
timeout 120 python -m pytest tests/test_hooks_simulcast.py tests/test_vaa.py -m gpu -q -n 4 2>&1 | tail -2 | tee $o/pytest_hooks.txt
timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "verified", d.get("verified"), d["roofline"]["events_ms"])
for k in ("e2e", "e2e_overlapped", "e2e_pipelined"):
    print(k, {x: d[k][x] for x in d[k] if x in ("frames_per_s", "frames_per_s_second_half", "steps_ahead")}, d[k]["bitstream_vs_reference"]["match"])
for k in d:
    if k.startswith("config") and k != "config": print(k, d[k].get("device_frames_per_s"), d[k].get("c_path_frames_per_s"), d[k].get("same_bitstreams"))
PY
