#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02z; mkdir -p $o
yuv=/tmp/c5.yuv; oracle/_ref/ref_dec oracle/_ref/res/VID_1280x720_cavlc_temporal_direct.264 $yuv > /dev/null 2>&1
for n in 1 8; do
  WELSHIP_LIB=$PWD/openh264_amd/libwelship.so WELS_HIP_TRACE=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $o/trace$n -- oracle/_ref/ref_enc_hip -parallel $n -i $yuv -w 1280 -h 720 -o /tmp/c5.264 -frames 60 -fps 30 -rc 1 -bitrate 1500000 -slcmd 2 -slcmbnum 900 -threads 1 -iper 0 -quiet 2>&1 | grep -v "hooks: did" | tail -12 > $o/run$n.txt
  cat $o/run$n.txt | cut -c1-200
  python - <<PY
import csv, glob
f = glob.glob("$o/trace$n/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _, _ in ev[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
import collections
by = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in ev: by[k][0] += 1; by[k][1] += e - s
print("n=$n kernels", len(ev), "span ms", (t1 - t0) / 1e6, "union busy ms", busy / 1e6, "sum of durations ms", tot / 1e6, "avg concurrency while busy", tot / max(busy, 1), "queues", len({q for _, _, _, q in ev}))
for k, (c, d) in sorted(by.items(), key=lambda x: -x[1][1])[:6]: print("   %-42s calls %5d avg %.3f ms" % (k, c, d / c / 1e6))
PY
done 2>&1 | tee $o/summary.txt
