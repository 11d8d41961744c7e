#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02an; mkdir -p $o
yuv=/tmp/c4.yuv; oracle/_ref/ref_dec oracle/_ref/res/VID_1920x1080_cavlc_temporal_direct.264 $yuv > /dev/null 2>&1
for n in 4 8; do
  WELSHIP_LIB=$PWD/openh264_amd/libwelship.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $o/trace$n -- oracle/_ref/ref_enc_hip -parallel $n -i $yuv -w 1920 -h 1080 -o /tmp/c4.264 -frames 30 -fps 30 -rc 1 -bitrate 1500000 -threads 1 -iper 0 -quiet -slcmd 1 -slcnum 4 -simulcast 240 135 -simulcast 480 270 -simulcast 960 540 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$o/trace$n/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]; k = "MD" if "k_inter_pool" in k else "deblock" if "k_deblock" in k else "intra" if "k_intra" in k else "expand" if "k_expand" in k else "other"
    g = (int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Grid_Size_Y", 0) or 0))
    by[(k, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("== n=$n kernels", len(rows))
for (k, g), v in sorted(by.items(), key=lambda x: -sum(x[1]))[:14]:
    print("  %-8s grid %-14s calls %5d avg %.3f ms  total %.1f ms" % (k, g, len(v), sum(v) / len(v), sum(v)))
m = glob.glob("$o/trace$n/**/*memory_copy_trace.csv", recursive=True)
if m:
    rows = list(csv.DictReader(open(m[0])))
    tot = collections.defaultdict(lambda: [0, 0.0, 0])
    for r in rows:
        d = r.get("Direction", "?"); t = tot[d]; t[0] += 1; t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; t[2] += int(r.get("Size", 0) or 0)
    for d, (c, ms, b) in tot.items(): print("  copies %-28s calls %6d total %.1f ms  %.1f MB  -> %.1f GB/s" % (d, c, ms, b / 1e6, b / 1e9 / max(ms / 1e3, 1e-9)))
PY
done 2>&1 | tee $o/summary.txt
