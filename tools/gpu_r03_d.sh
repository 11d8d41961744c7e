#!/bin/bash
# Round 3, fourth run: run scheduler (runs of 4 macroblocks by default) vs run lengths 2 / 8 / 16 / whole rows vs tickets.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_d; rm -rf $o; mkdir -p $o
t0=$(date +%s); lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
for L in 4 2 8 16 0; do
  if [ $L = 0 ]; then export WELSHIP_MD_ROWS=0; unset WELSHIP_MD_RUN; else unset WELSHIP_MD_ROWS; export WELSHIP_MD_RUN=$L; fi
  timeout 120 python bench.py --quick --steps 60 > $o/bench_quick_run$L.json 2> $o/bench_quick_run$L.err
  timeout 120 python bench.py --quick --steps 40 --content res > $o/bench_quick_res_run$L.json 2> $o/bench_quick_res_run$L.err
  for f in bench_quick_run$L bench_quick_res_run$L; do echo "$f: $(python -c "import json; d=json.loads(open('$o/$f.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"; done
done
unset WELSHIP_MD_ROWS WELSHIP_MD_RUN
lap "A/B"
timeout 400 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -5 $o/pytest_gpu.txt; lap "gpu tier"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OLDPWD/$o/pmc_$c -- python $OLDPWD/bench.py --quick --steps 8 --warmup 4 > $OLDPWD/$o/pmc_$c.log 2>&1 )
  python tools/pmc_summary.py $o/pmc_$c | grep -E "inter_|deblock|k_tile|k_expand|src_tile"
done > $o/pmc_traffic.txt 2>&1; cat $o/pmc_traffic.txt; lap "pmc"
timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles.txt 2>&1; head -22 $o/phase_cycles.txt; lap "phase cycles"
( time timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", d["ms_per_step"], "roofline", d["roofline"], "verified", d.get("verified"))
for k in ("res_clip", "e2e", "e2e_overlapped", "latency", "intra_720p"):
    if k in d: print(k, json.dumps(d[k])[:300])
PY
lap "bench default"
