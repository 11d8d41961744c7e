#!/bin/bash
# Round 3, fourteenth run: waves per deblocking workgroup (73 VGPRs: one 16-wave workgroup per CU -> 1024 bands take four rounds), and the longer pipelined leg
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_n; rm -rf $o; mkdir -p $o
for W in 16 12 8 6 4; do
  WELSHIP_DB_WAVES=$W timeout 120 python bench.py --quick --steps 60 > $o/bench_quick_db$W.json 2> $o/bench_quick_db$W.err
  echo "db waves $W: $(python -c "import json; d=json.loads(open('$o/bench_quick_db$W.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")" | tee -a $o/db_waves.txt
done
timeout 200 python tools/e2e_pipe_run.py 256 80 2>&1 | grep -v amdgpu.ids | tee -a $o/pipelined_80_frames.txt
