"""Repeat the pipelined-group GPU tests in one process, alternating between two picture sizes / contents the way the tier does, and say
which mode (synchronous / pipelined), session and access unit deviates from the first run's streams when one does.
   python tools/stress_pipelined.py <iterations> [<lib>]
(round 6: test_hip_pipelined_group_reencodes_after_cavlc_overflow[2] failed once in a tier run and never in 600 repeats of itself alone)"""
import sys, os, time, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openh264_amd as oh
from openh264_amd.utils.synth import make_sequence

lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.getcwd(), "openh264_amd", "libwelship.so")


def run(w, h, frames, qp, contents, ring, intra_period=0, threads=2, ahead=1):
    fsz = w * h * 3 // 2
    seqs = [make_sequence(c, w, h, frames) for c in contents]
    e = oh.Encoder(lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = w, h, qp, intra_period, 30.0, 5000000
    p.bEnableSceneChangeDetect = False
    out = {}
    for mode in ("sync", "pipe"):
        g = oh.EncoderGroup(p, len(seqs), ring_slots=ring, host_threads=threads, lib_path=lib)
        if mode == "pipe":
            g.set_pipelined(ahead)
        got = [[] for _ in seqs]
        for f in range(frames):
            pics = g.make_pictures([s[f * fsz:(f + 1) * fsz] for s in seqs])
            res = g.encode_frames(pics, want_bytes=True) if mode == "sync" else g.encode_frames_pipelined(pics, want_bytes=True)
            if res is not None:
                for s, bs in enumerate(res):
                    got[s].append(hashlib.sha1(bs).hexdigest()[:10])
        while mode == "pipe":
            res = g.encode_frames_pipelined(None, want_bytes=True)
            if res is None:
                break
            for s, bs in enumerate(res):
                got[s].append(hashlib.sha1(bs).hexdigest()[:10])
        out[mode] = (got, hashlib.sha1(g.recon(0)).hexdigest()[:10])
        g.close()
    return out


cases = {
    "A320": dict(w=320, h=192, frames=8, qp=26, contents=("synth", "checker5", "synth", "pan7"), ring=3, intra_period=5, threads=4),
    "B64": dict(w=64, h=64, frames=4, qp=3, contents=("synth", "checker5", "synth", "checker8"), ring=3),
}
n = int(sys.argv[1]); bad = 0; t0 = time.time(); want = {}
for i in range(n):
    for name, kw in cases.items():
        for ahead in (1, 2):
            out = run(ahead=ahead, **kw)
            for mode in ("sync", "pipe"):
                key = (name, )
                if key not in want:
                    want[key] = out[mode]
                if out[mode] != want[key]:
                    bad += 1
                    for s, (a, b) in enumerate(zip(out[mode][0], want[key][0])):
                        for f, (x, y) in enumerate(zip(a, b)):
                            if x != y:
                                print("iteration", i, name, "ahead", ahead, mode, "session", s, "access unit", f, "differs", flush=True)
                    if out[mode][1] != want[key][1]:
                        print("iteration", i, name, "ahead", ahead, mode, "reconstruction of session 0 differs", flush=True)
print("iterations", n, "deviating runs", bad, "WELSHIP_MD_SPLIT", os.environ.get("WELSHIP_MD_SPLIT"), "%.1f s" % (time.time() - t0))
