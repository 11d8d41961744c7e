#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r02i; mkdir -p $o
timeout 900 python -m pytest tests/test_hooks_sha1.py -m gpu -q -x > $o/pytest_hooks.txt 2>&1; tail -5 $o/pytest_hooks.txt
# all 512 device rows on the MI355X, once
cat > /tmp/allrows.py <<'PY'
import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import test_hooks_sha1 as T, subprocess, pathlib, tempfile
from openh264_amd import build as B
lib = B.build_hip()
d = pathlib.Path(tempfile.mkdtemp())
subprocess.check_call([os.path.join(T.REF, "ref_dec"), os.path.join(T.RES, "BA_MW_D.264"), str(d / "BA_MW_D.264.yuv")], stdout=subprocess.DEVNULL)
for k in range(4): (d / ("layer%d.cfg" % k)).write_bytes(open(os.path.join(T.RES, "layer2.cfg"), "rb").read())
(d / "welsenc.cfg").write_bytes(open(os.path.join(T.RES, "welsenc.cfg"), "rb").read())
rows = T._device_rows(); bad = 0; t0 = time.time()
for i, r in enumerate(rows):
    got, pics, err = T._run_row(d, lib, r, "x")
    if got != r[0] or pics < 40: bad += 1; print("BAD", i, r[4], got, pics)
print("device rows", len(rows), "bad", bad, "seconds", round(time.time() - t0, 1))
PY
timeout 1500 python /tmp/allrows.py > $o/sha1_table_all_device_rows.txt 2>&1; tail -3 $o/sha1_table_all_device_rows.txt
