#!/bin/bash
# Round 3, ninth run: the pipelined session group (WelsHipGroupEncodeFramesPipelined) -- parity tests, then the end-to-end legs.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_i; rm -rf $o; mkdir -p $o
timeout 300 python -m pytest tests/test_multi_rank.py -m gpu -x -q 2>&1 | tail -5 | tee $o/pytest_pipelined.txt
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $o/bench_$name.json 2> $o/bench_$name.err
  python - <<PY
import json
d = json.loads(open("$o/bench_$name.json").read().strip().splitlines()[-1])
print("$name: value", round(d["value"]), "e2e", round(d["e2e"]["frames_per_s"]), "e2e_overlapped", round(d["e2e_overlapped"]["frames_per_s"]),
      "e2e_pipelined", round(d["e2e_pipelined"]["frames_per_s"]), d["e2e_pipelined"]["bitstream_vs_reference"], d["e2e_pipelined"]["host_thread_ms_per_picture"])
PY
  tail -3 $o/bench_$name.err
}
run default WELSHIP_X=1

# host threads: the entropy coder needs ~0.8 ms of one core per picture
for t in 64 128; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify --host-threads $t > $o/bench_t$t.json 2> $o/bench_t$t.err
  python - <<PY
import json
d = json.loads(open("$o/bench_t$t.json").read().strip().splitlines()[-1])
print("threads $t: e2e", round(d["e2e"]["frames_per_s"]), "e2e_pipelined", round(d["e2e_pipelined"]["frames_per_s"]), d["e2e_pipelined"]["host_thread_ms_per_picture"])
PY
done
