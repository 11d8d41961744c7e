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


def one_case(seed, lib, tmp, verbose=False, max_threads=1, low_qp=False, screen=False):
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
    src = os.path.join(tmp, "c%d.yuv" % seed)
    open(src, "wb").write(yuv)
    base = ["-i", src, "-w", str(w), "-h", str(h), "-fps", "30", "-quiet"] + flags
    outs = []
    info = ""
    for exe, env_extra in (("ref_enc", {}), ("ref_enc_hip", {"WELSHIP_LIB": lib, "WELS_HIP_DYNSLICE": "1", "WELS_HIP_TRACE": "1"})):
        out = os.path.join(tmp, "o%d_%s.264" % (seed, exe))
        p = subprocess.run([os.path.join(REF, exe)] + base + ["-o", out], env=dict(os.environ, **env_extra), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        err = p.stderr.decode(errors="replace")
        if p.returncode != 0:
            if exe == "ref_enc":          # a parameter combination the reference itself refuses: not a case
                os.remove(src)
                return seed, "skipped (reference: %s)" % err.strip().splitlines()[-1][:60] if err.strip() else "skipped", True
            os.remove(src)
            return seed, "FAILED rc=%d %s\n%s" % (p.returncode, " ".join(flags), err[-800:]), False
        outs.append(open(out, "rb").read())
        os.remove(out)
        if exe == "ref_enc_hip":
            done = [l for l in err.splitlines() if "picture complete" in l]
            slices = sum(int(l.split("complete:")[1].split()[0]) for l in done)
            calls = sum(int(l.split("slices,")[1].split()[0]) for l in done)
            info = "%dx%d %d frames, %d pictures on the device, %d slices, %d device calls" % (w, h, n, len(done), slices, calls)
            again = err.count("coded again at QP")
            if again:
                info += ", %d macroblock passes repeated after a CAVLC overflow" % again
            if "welship hooks: installed" not in err or not done:
                os.remove(src)
                return seed, "NOT ON THE DEVICE %s\n%s" % (" ".join(flags), err[-400:]), False
    ok = outs[0] == outs[1]
    if not ok and threads > 1:
        # With slice threads the unmodified reference is not always deterministic itself (rate control, screen content: its output depends
        # on how the slice tasks interleave -- seen as two to four different streams of one command line on a loaded machine, from the
        # reference alone and from the reference with the hooks alike): accept what the reference produces on any of a few more runs -- and say so
        for k in range(12):
            out = os.path.join(tmp, "o%d_again.264" % seed)
            subprocess.run([os.path.join(REF, "ref_enc")] + base + ["-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            again = open(out, "rb").read()
            os.remove(out)
            if again == outs[1]:
                ok = True
                info += " (the reference's own output varies between runs of this command: matched on run %d)" % (k + 2)
                break
    os.remove(src)
    return seed, ("ok   " if ok else "DIFF ") + info + ("" if ok and not verbose else "  " + " ".join(flags)), ok


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
