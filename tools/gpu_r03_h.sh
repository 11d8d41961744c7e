#!/bin/bash
# Round 3, eighth run: do the transfers of the end-to-end legs run on the SDMA engines or as shader copies behind the kernels?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_h; rm -rf $o; mkdir -p $o
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify > $o/bench_$name.json 2> $o/bench_$name.err
  python - <<PY
import json
d = json.loads(open("$o/bench_$name.json").read().strip().splitlines()[-1])
print("$name: value", round(d["value"]), "e2e", round(d["e2e"]["frames_per_s"]), "e2e_overlapped", round(d["e2e_overlapped"]["frames_per_s"]), "latency1", round(d["latency"]["sessions_1"]["ms_per_frame"], 2))
PY
}
run default WELSHIP_X=1
run sdma0 HSA_ENABLE_SDMA=0
run sdma1 HSA_ENABLE_SDMA=1
run hwq8 GPU_MAX_HW_QUEUES=8
