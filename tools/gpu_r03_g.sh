#!/bin/bash
# Round 3, seventh run: where do the end-to-end legs lose their time -- do the transfers overlap the kernels at all?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r03_g; rm -rf $o; mkdir -p $o
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$o/trace -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-verify > $OLDPWD/$o/bench.json 2> $OLDPWD/$o/bench.err )
ls $o/trace/*/ | head; for f in $(find $o/trace -name "*stats.csv"); do echo "== $f"; head -12 $f | cut -c1-200; done
python - <<PY
import csv, glob
for f in glob.glob("$o/trace/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    import collections
    by = collections.defaultdict(lambda: [0, 0, 0])
    for r in rows:
        k = r.get("Direction", "?")
        b = int(r.get("Bytes", r.get("Size", "0")) or 0)
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[k][0] += 1; by[k][1] += b; by[k][2] += d
    for k, v in by.items(): print(" ", k, "n", v[0], "MB", round(v[1] / 1e6, 1), "busy ms", round(v[2] / 1e6, 2), "GB/s while busy", round(v[1] / max(v[2], 1), 2))
PY
tail -c 1500 $o/bench.json
