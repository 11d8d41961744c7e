#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=$PWD/gpurun_out/r02l; mkdir -p $o
R=$PWD/oracle/_ref
export WELSHIP_LIB=$PWD/openh264_amd/libwelship.so
python3 -c "
import sys; sys.path.insert(0,'.')
from openh264_amd.utils.synth import synth_sequence
open('/tmp/s320x184.yuv','wb').write(synth_sequence(320,184,6))"
cd /tmp
python3 -c "
import sys; sys.path.insert(0,'/root/repo')
from openh264_amd.utils.synth import synth_sequence
open('/tmp/s640.yuv','wb').write(synth_sequence(640,368,6))"
WELS_HIP_DUMP_RECORDS=/tmp/hip $R/ref_enc_hip -i s640.yuv -w 640 -h 368 -o o.264 -fps 30 -quiet -rc -1 -qp 28 -complexity 1 -simulcast 320 192 >/dev/null 2>&1
# the same small picture through the session API (round-1 path)
python3 - <<'PY2' > $o/session_320x184.txt 2>&1
import sys, subprocess; sys.path.insert(0, '/root/repo')
import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence
for (w, h, c) in ((320, 184, 1), (320, 184, 2), (640, 360, 1), (322, 182, 1)):
    yuv = synth_sequence(w, h, 6)
    open('/tmp/x.yuv', 'wb').write(yuv)
    subprocess.check_call(['/root/repo/oracle/_ref/ref_enc', '-i', '/tmp/x.yuv', '-w', str(w), '-h', str(h), '-o', '/tmp/x.264', '-rc', '-1', '-qp', '28', '-fps', '30', '-complexity', str(c), '-quiet'], stdout=subprocess.DEVNULL)
    bs, _ = oh.encode_sequence(yuv, w, h, iDLayerQp=28, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000, iComplexityMode=c)
    print(w, h, c, 'session API', 'SAME' if bs == open('/tmp/x.264', 'rb').read() else 'DIFF')
PY2
cat $o/session_320x184.txt
python3 - <<'PY' > $o/recdiff.txt
import glob, os, struct
def key(r):
    t, cbp = r[0], r[1]
    if t == 6: return (bytes(r[0:3]), r[120:124])
    parts = [bytes(r[0:144 - 16]), r[128:129]]
    allowed = 0
    if t == 1:
        allowed |= 1 << 16
        if cbp & 15: allowed |= 0xffff
    else:
        for k in range(4):
            if (cbp >> k) & 1: allowed |= 0xf << (4 * k)
    if (cbp >> 4) >= 1: allowed |= 1 << 25
    if (cbp >> 4) == 2: allowed |= 0xff << 17
    for b in range(25):
        if (allowed >> b) & 1: parts.append(bytes(r[144 + 32 * b: 144 + 32 * b + 32]))
    if (allowed >> 25) & 1: parts.append(bytes(r[144 + 800: 144 + 816]))
    return tuple(parts)
for f in sorted(glob.glob("/tmp/hip_*.rec")):
    e = f.replace("/tmp/hip_", "/root/repo/oracle/_ref/dbg/emu_")
    a, b = open(f, "rb").read(), open(e, "rb").read()
    n = len(a) // 960
    bad = [mb for mb in range(n) if key(a[mb*960:(mb+1)*960]) != key(b[mb*960:(mb+1)*960])]
    print(os.path.basename(f), "differing MBs:", bad[:10], "of", n)
    for mb in bad[:3]:
        x, y = a[mb*960:(mb+1)*960], b[mb*960:(mb+1)*960]
        first = next(i for i in range(960) if x[i] != y[i])
        print("   mb", mb, "first byte", first, "hip", list(x[0:8]), "emu", list(y[0:8]), "cost", struct.unpack_from("<i", x, 120)[0], struct.unpack_from("<i", y, 120)[0], "mvd0", struct.unpack_from("<hh", x, 32), struct.unpack_from("<hh", y, 32), "nzc", list(x[96:120]), list(y[96:120]))
PY
cat $o/recdiff.txt
