#!/bin/bash
# Round 3, third run: row scheduler one slice at a time (default) vs spread (WELSHIP_MD_ROWS=2) vs tickets (=0); macroblock-tiled sources.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_c; rm -rf $o; mkdir -p $o
t0=$(date +%s); lap() { echo "[$(( $(date +%s) - t0 )) s] $*"; }
for mode in 1 2 0; do
  WELSHIP_MD_ROWS=$mode timeout 120 python bench.py --quick --steps 60 > $o/bench_quick_rows$mode.json 2> $o/bench_quick_rows$mode.err
  WELSHIP_MD_ROWS=$mode timeout 120 python bench.py --quick --steps 40 --content res > $o/bench_quick_res_rows$mode.json 2> $o/bench_quick_res_rows$mode.err
  for f in bench_quick_rows$mode bench_quick_res_rows$mode; do echo "$f: $(python -c "import json; d=json.loads(open('$o/$f.json').read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['events_ms'])")"; done
done
lap "A/B"
timeout 400 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -5 $o/pytest_gpu.txt; lap "gpu tier"
for mode in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && WELSHIP_MD_ROWS=$mode timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OLDPWD/$o/pmc_rows${mode}_$c -- python $OLDPWD/bench.py --quick --steps 8 --warmup 4 > $OLDPWD/$o/pmc_rows${mode}_$c.log 2>&1 )
    python tools/pmc_summary.py $o/pmc_rows${mode}_$c | grep -E "inter_|deblock|k_tile|k_expand|src_tile" | sed "s/^/rows=$mode /"
  done
done > $o/pmc_traffic.txt 2>&1; cat $o/pmc_traffic.txt; lap "pmc"
WELSHIP_PROF_GROUPS=1 timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles_rows.txt 2>&1; head -24 $o/phase_cycles_rows.txt; lap "phase cycles"
( time timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", d["ms_per_step"], "roofline", d["roofline"], "verified", d.get("verified"))
for k in ("res_clip", "e2e", "e2e_overlapped", "latency", "intra_720p"):
    if k in d: print(k, json.dumps(d[k])[:300])
PY
lap "bench default"
