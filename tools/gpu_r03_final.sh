#!/bin/bash
# Round 3, evidence run at the round's last kernel code: GPU tier, default bench line, rocprofv3 kernel statistics, PMC traffic passes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_final; rm -rf $o; mkdir -p $o
T0=$SECONDS; lap() { echo "[$((SECONDS - T0)) s] $1"; }
timeout 500 python -m pytest tests -m gpu -q -n 4 > $o/pytest_gpu.txt 2>&1; tail -4 $o/pytest_gpu.txt; lap "gpu tier"
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; python - <<PY
import json
d = json.loads(open("$o/bench_default.json").read().strip().splitlines()[-1])
print("default: value", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 4), "traffic x", d["roofline"].get("traffic_over_algorithmic"), "verified", d.get("verified"))
for k in ("e2e", "e2e_overlapped", "e2e_pipelined"):
    print(k, {x: d[k][x] for x in d[k] if x in ("frames_per_s", "frames_per_s_second_half", "steps_ahead")}, d[k]["bitstream_vs_reference"]["match"])
for k in d:
    if k.startswith("config") and k != "config": print(k, d[k].get("device_frames_per_s"), d[k].get("c_path_frames_per_s"), d[k].get("same_bitstreams"))
print("res_clip", d.get("res_clip", {}).get("value"), "intra_720p", d.get("intra_720p", {}).get("value"), "latency", d.get("latency"))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"))
PY
lap "bench default"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$o/stats -- python $OLDPWD/bench.py --quick --steps 6 --warmup 2 > $OLDPWD/$o/prof_bench.json 2> $OLDPWD/$o/prof_bench.err )
f=$(find $o/stats -name "*kernel_stats.csv" | head -1); cp $f $o/kernel_stats.csv 2>/dev/null; head -14 $o/kernel_stats.csv | cut -c1-160
python -c "import json; d=json.loads(open('$o/prof_bench.json').read().strip().splitlines()[-1]); print('events of that run: avg MD launch ms', d['roofline']['avg_launch_ms'])"
lap "kernel stats"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OLDPWD/$o/pmc_$c -- python $OLDPWD/bench.py --quick --steps 8 --warmup 4 > $OLDPWD/$o/pmc_$c.log 2>&1 )
  python tools/pmc_summary.py $o/pmc_$c | grep -E "inter_|deblock|k_tile|k_expand|src_tile"
done > $o/pmc_traffic.txt 2>&1; cat $o/pmc_traffic.txt; lap "pmc"
rm -rf $o/stats $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE
timeout 200 python tools/phase_profile.py 256 > $o/phase_cycles.txt 2>&1; head -24 $o/phase_cycles.txt; lap "phase cycles"
