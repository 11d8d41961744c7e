#!/bin/bash
# Round 3, sixteenth run: the pipelined leg on bench.py's content, host threads, and the whole default bench line
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_p; mkdir -p $o
for t in 6 8 12 24; do echo "threads per half $t" | tee -a $o/pipelined_60_frames.txt; timeout 200 python tools/e2e_pipe_run.py 256 60 $t 2>&1 | grep -v amdgpu.ids | tee -a $o/pipelined_60_frames.txt; done
S0=$SECONDS
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err
echo "bench.py default run: $((SECONDS - S0)) s" | tee $o/bench_default.time
python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "verified", d.get("verified"))
for k in ("e2e", "e2e_overlapped", "e2e_pipelined"):
    print(k, {x: d[k][x] for x in d[k] if x in ("frames_per_s", "frames_per_s_second_half", "bitstream_vs_reference", "host_thread_ms_per_picture")})
for k in d:
    if k.startswith("config"): print(k, json.dumps(d[k])[:300])
print("cpu_baseline", d.get("cpu_baseline"))
PY
tail -3 $o/bench_default.err
