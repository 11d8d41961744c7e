"""Randomised parity of sessions with size-limited slices (SM_SIZELIMITED_SLICE) through the dispatch-table binding
(TEST INFRASTRUCTURE).

Camera-like synthetic clips at random sizes, encoded by the unmodified reference (oracle/_ref/ref_enc) and by the reference with
this engine behind SWelsFuncPtrList (oracle/_ref/ref_enc_hip, WELS_HIP_DYNSLICE=1) with `-slcmd 3` and a random slice size
(421 .. 3000 bytes: from a few macroblocks per slice -- slices shorter than a macroblock row, dozens of slices per picture -- to
one slice per picture), random rate-control mode / QP, complexity, temporal layers, LTR, denoising, background and
scene-change detection, deblocking mode, intra period and entropy coder.  The two bitstreams must be identical and the hooks must have coded
every picture (several device calls for a picture with several slices).

usage: fuzz_dynslice.py [--lib path] [--seed S] [--cases N] [--workers W] [-v]
"""
import argparse
import os
import random
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def one_case(seed, lib, tmp, verbose=False, max_threads=1, low_qp=False, screen=False, force=None, env=None, ref_runs=None):
    """One random session.  `force`: flag -> value put over the drawn ones (e.g. {"-threads": "1"}: the same session without slice
    threads); `env`: extra environment of the hooked run; `ref_runs`: how often the unmodified reference runs (default: once without
    slice threads, three times with them).

    The reference runs FIRST, `ref_runs` times, and the SET of its outputs is what the hooked run is compared with: the result line says
    how many different streams the reference itself produced ("reference: 1 stream in 3 runs").  A constant-QP session (-rc -1) must
    give one stream however its slice tasks interleave; more than one fails the case as a finding about the reference, not as a pass."""
    from openh264_amd.utils.synth import make_sequence
    rng = random.Random(seed)
    w = 16 * rng.randint(4, 40) - rng.choice((0, 0, 0, 2, 8))
    h = 16 * rng.randint(3, 23) - rng.choice((0, 0, 0, 2, 6))
    if (w // 16) * (h // 16) > 500:
        h = max(48, 16 * (500 // max(1, w // 16)))
    n = rng.randint(4, 10)
    content = rng.choice(("synth", "synth", "pan", "checker"))
    yuv = make_sequence(content, w, h, n)
    rc = rng.choice((-1, -1, 0, 1, 3))
    threads = max_threads if max_threads <= 1 else rng.choice((1, 2, 3, 4)[:max_threads])
    flags = ["-slcmd", "3", "-slcsize", str(rng.choice((421, 450, 500, 600, 800, 1200, 1500, 3000))), "-threads", str(threads), "-loadbalancing", "0",
             "-rc", str(rc), "-complexity", str(rng.randint(0, 2)), "-numtl", str(rng.choice((1, 1, 2, 3))),
             "-deblock", str(rng.choice((0, 0, 1, 2))), "-iper", str(rng.choice((0, 0, 3, 5))),
             "-bgd", str(rng.randint(0, 1)), "-scene", str(rng.randint(0, 1)), "-ltr", str(rng.choice((0, 0, 1))),
             "-denoise", str(rng.choice((0, 0, 1)))]
    if low_qp:         # constant QP 0 .. 12 on saturated content: CAVLC level overflows, macroblocks coded again at QP + 2 (TRY_REENCODING)
        rc = -1
        flags[flags.index("-rc") + 1] = "-1"
        yuv = make_sequence(rng.choice(("checker", "checker2", "checker5", "synth")), w, h, n)
        flags += ["-qp", str(rng.randint(0, 12))]
    elif rc == -1:
        flags += ["-qp", str(rng.choice((14, 20, 24, 28, 34, 40)))]
    else:
        flags += ["-bitrate", str(rng.choice((150000, 400000, 1000000, 3000000))), "-frameskip", str(rng.randint(0, 1))]
    if rng.randint(0, 3) == 0:
        flags += ["-nalsize", str(rng.choice((800, 1000, 1500)))]
    if rng.randint(0, 3) == 0:
        flags += ["-cabac", "1", "-profile", str(rng.choice((77, 100)))]
    if screen:         # screen content (tools/fuzz_screen.py's synthetic documents: still, scrolling, jumping): static / scrolled skips, the
        import fuzz_screen         # cross and feature searches and the chain of the 8x8 searches' costs, all under size-limited slices
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        w, h = max(64, w & ~1), max(64, h & ~1)
        n = rng.randint(5, 10)
        yuv = fuzz_screen.make_clip(w, h, n, seed)
        flags += ["-usage", "1"]
        if "-bitrate" in flags:
            flags[flags.index("-bitrate") + 1] = str(rng.choice((150000, 450000, 2400000)))
    for k, v in (force or {}).items():
        if k in flags:
            flags[flags.index(k) + 1] = str(v)
        else:
            flags += [k, str(v)]
    threads = int(flags[flags.index("-threads") + 1])
    rc = int(flags[flags.index("-rc") + 1])
    tag = "%d_%d" % (seed, os.getpid()) + "_%x" % (id(flags) & 0xffffff)
    src = os.path.join(tmp, "c%s.yuv" % tag)
    open(src, "wb").write(yuv)
    base = ["-i", src, "-w", str(w), "-h", str(h), "-fps", "30", "-quiet"] + flags
    out = os.path.join(tmp, "o%s.264" % tag)

    def run(exe, env_extra):
        p = subprocess.run([os.path.join(REF, exe)] + base + ["-o", out], env=dict(os.environ, **env_extra), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        err = p.stderr.decode(errors="replace")
        data = None
        if p.returncode == 0:
            data = open(out, "rb").read()
        if os.path.exists(out):
            os.remove(out)
        return p.returncode, err, data

    import hashlib
    h8 = lambda d: hashlib.md5(d).hexdigest()[:8]
    # Where the UNMODIFIED reference is not deterministic (measured on the MI355X box's host, 3 slice threads, constant QP, screen content:
    # "2 streams in 3 runs", profiles/r04_size_limited_slices_screen_threads_reference_varies.txt): SSlice::uiSliceFMECostDown is only ever
    # added to, never reset (svc_motion_estimate.cpp:1082-1091, slice.h:195); a slice task takes its SSlice from the buffer of the POOL THREAD
    # that happens to run it (wels_task_encoder.cpp:238-239, m_iThreadIdx), so which partition's history a struct carries depends on which
    # thread picked up which task; CountFMECostDown sums the structs of the current picture (svc_motion_estimate.cpp:1027-1041) and that sum
    # switches the feature search of the following pictures on or off (UpdateFMESwitch :1054-1058, CalcFMESwitchFlag).  Rate control with
    # slice threads varies as well (the slices' bit counts arrive in completion order).  Everything else must give ONE stream.
    may_vary = threads > 1 and (screen or "-usage" in flags or rc != -1)
    try:
        # the unmodified reference first: the set of streams it produces for this command line
        k_ref = ref_runs if ref_runs else (1 if threads <= 1 else 3)
        ref_set = set()
        for _ in range(k_ref):
            rcode, err, data = run("ref_enc", {})
            if rcode != 0:          # a parameter combination the reference itself refuses: not a case
                return seed, "skipped (reference: %s)" % err.strip().splitlines()[-1][:60] if err.strip() else "skipped", True
            ref_set.add(data)
        if not may_vary and len(ref_set) > 1:
            return seed, "REFERENCE VARIES where it must not (%d streams in %d runs)  %s" % (len(ref_set), k_ref, " ".join(flags)), False
        hip_env = {"WELSHIP_LIB": lib, "WELS_HIP_DYNSLICE": "1", "WELS_HIP_TRACE": "1"}
        hip_env.update(env or {})
        rcode, err, hip_out = run("ref_enc_hip", hip_env)
        if rcode != 0:
            return seed, "FAILED rc=%d %s\n%s" % (rcode, " ".join(flags), err[-800:]), False
        done = [l for l in err.splitlines() if "picture complete" in l]
        slices = sum(int(l.split("complete:")[1].split()[0]) for l in done)
        calls = sum(int(l.split("slices,")[1].split()[0]) for l in done)
        info = "%dx%d %d frames, %d pictures on the device, %d slices, %d device calls" % (w, h, n, len(done), slices, calls)
        again = err.count("coded again at QP")
        if again:
            info += ", %d macroblock passes repeated after a CAVLC overflow" % again
        if "welship hooks: installed" not in err or not done:
            return seed, "NOT ON THE DEVICE %s\n%s" % (" ".join(flags), err[-400:]), False
        ok = hip_out in ref_set
        if not ok and may_vary:
            # a session of the class above: the reference's set is widened -- four runs at a time, so that its slice tasks interleave
            # differently -- until it contains the hooked run's stream or 32 runs have not produced it.  The line names every stream.
            from concurrent.futures import ThreadPoolExecutor as _TPE
            outs2 = [out + ".%d" % i for i in range(4)]

            def run_ref_to(o):
                p = subprocess.run([os.path.join(REF, "ref_enc")] + base + ["-o", o], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                d = open(o, "rb").read() if p.returncode == 0 and os.path.exists(o) else None
                if os.path.exists(o):
                    os.remove(o)
                return d
            while not ok and k_ref < 32:
                with _TPE(4) as ex2:
                    for d in ex2.map(run_ref_to, outs2):
                        k_ref += 1
                        if d is not None:
                            ref_set.add(d)
                ok = hip_out in ref_set
        keep = os.environ.get("FUZZ_DYNSLICE_KEEP")          # developer aid: the streams of a failing case, for tools/h264_parse.py
        if not ok and keep:
            os.makedirs(keep, exist_ok=True)
            open(os.path.join(keep, "%s_hip.264" % tag), "wb").write(hip_out)
            for d in ref_set:
                open(os.path.join(keep, "%s_ref_%s.264" % (tag, h8(d))), "wb").write(d)
            open(os.path.join(keep, "%s_hip.log" % tag), "w").write(err)
        info += " (reference: %d stream%s in %d run%s {%s}; with the hooks: %s)" % (len(ref_set), "" if len(ref_set) == 1 else "s", k_ref, "" if k_ref == 1 else "s",
                                                                                   ",".join(sorted(h8(d) for d in ref_set)), h8(hip_out))
        return seed, ("ok   " if ok else "DIFF ") + info + ("" if ok and not verbose else "  " + " ".join(flags)), ok
    finally:
        if os.path.exists(src):
            os.remove(src)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("-v", action="store_true")
    ap.add_argument("--low-qp", action="store_true", help="constant QP 0..12 on saturated content: the re-encode after a CAVLC level overflow inside size-limited slices")
    ap.add_argument("--screen", action="store_true", help="screen-content sessions (-usage 1) on synthetic documents")
    ap.add_argument("--threads", type=int, default=1, help="slice threads up to this many (one partition of the picture per thread)")
    a = ap.parse_args()
    from openh264_amd import build as B
    lib = os.path.abspath(a.lib) if a.lib else B.build_emu()
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(a.workers) as ex:
            for seed, msg, ok in ex.map(lambda s: one_case(s, lib, tmp, a.v, a.threads, a.low_qp, a.screen), range(a.seed * 1000, a.seed * 1000 + a.cases)):
                print("%6d %s" % (seed, msg), flush=True)
                bad += 0 if ok else 1
    print("%d cases, %d failed (library %s)" % (a.cases, bad, os.path.basename(lib)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
