#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02j; mkdir -p $o
timeout 1200 python -m pytest tests -m gpu -q -x > $o/pytest_gpu.txt 2>&1; tail -4 $o/pytest_gpu.txt
cat > /tmp/rows.py <<'PY'
import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import test_hooks_sha1 as T, subprocess, pathlib, tempfile
from openh264_amd import build as B
lib = B.build_hip()
d = pathlib.Path(tempfile.mkdtemp())
subprocess.check_call([os.path.join(T.REF, "ref_dec"), os.path.join(T.RES, "BA_MW_D.264"), str(d / "BA_MW_D.264.yuv")], stdout=subprocess.DEVNULL)
for k in range(4): (d / ("layer%d.cfg" % k)).write_bytes(open(os.path.join(T.RES, "layer2.cfg"), "rb").read())
(d / "welsenc.cfg").write_bytes(open(os.path.join(T.RES, "welsenc.cfg"), "rb").read())
allrows = T._device_rows()
sets = {"slcmd 1, bgd 1 (frame-constant QP + background detection)": [r for r in allrows if r[4]["-slcmd 0"] == "1" and r[4]["bgd"] == "1"],
        "slcmd 0 and 2, every 4th row (GOM-level QP)": [r for r in allrows if r[4]["-slcmd 0"] in ("0", "2")][::4]}
for name, rows in sets.items():
    bad = 0; t0 = time.time()
    for i, r in enumerate(rows):
        got, pics, err = T._run_row(d, lib, r, "x")
        if got != r[0] or pics < 40: bad += 1; print("BAD", i, r[4], got, pics)
    print(name, ": rows", len(rows), "bad", bad, "seconds", round(time.time() - t0, 1)); sys.stdout.flush()
PY
timeout 1500 python /tmp/rows.py > $o/sha1_table_rows_bgd_and_gom.txt 2>&1; tail -4 $o/sha1_table_rows_bgd_and_gom.txt
