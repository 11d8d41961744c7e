"""Repeat given size-limited-slice sessions of tools/fuzz_dynslice.py against one or several libraries (TEST INFRASTRUCTURE, developer aid).

usage: repro_dynslice.py <seeds>      seeds: comma separated; prefix s = screen content, q = low QP (e.g. s21001,q11012,5003)
environment:
  LIBS      libraries to run, colon separated (default: the product library); e.g. build/libs/libwelship_5840cd5.so:openh264_amd/libwelship.so
  RUNS      runs per (library, seed, variant)   (default 6)
  VARIANTS  comma separated: plain (as drawn), t1 (the same session with one slice thread), sync (WELSHIP_TRACE=1: every launch
            synchronised), load (four copies at a time)      (default plain,t1,sync,load)
Every line says how many different streams the unmodified reference produced for the command line (it runs first).
"""
import os
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import fuzz_dynslice
    seeds = []
    for tok in sys.argv[1].split(","):
        kind = tok[0] if tok[0] in "sq" else ""
        seeds.append((int(tok[len(kind):]), kind))
    libs = [os.path.abspath(p) for p in os.environ.get("LIBS", os.path.join(ROOT, "openh264_amd", "libwelship.so")).split(":")]
    runs = int(os.environ.get("RUNS", "6"))
    variants = os.environ.get("VARIANTS", "plain,t1,sync,load").split(",")
    bad_total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for lib in libs:
            for seed, kind in seeds:
                for var in variants:
                    kw = dict(verbose=True, max_threads=4, low_qp=kind == "q", screen=kind == "s")
                    if var == "t1":
                        kw["force"] = {"-threads": "1"}
                    if var == "sync":
                        kw["env"] = {"WELSHIP_TRACE": "1"}
                    workers = 4 if var == "load" else 1
                    with ThreadPoolExecutor(workers) as ex:
                        res = list(ex.map(lambda _: fuzz_dynslice.one_case(seed, lib, tmp, **kw), range(runs)))
                    bad = sum(0 if ok else 1 for _, _, ok in res)
                    bad_total += bad
                    print("== %s seed %s%d %-5s: %d of %d runs differ" % (os.path.basename(lib), kind, seed, var, bad, runs), flush=True)
                    for _, msg, ok in res:
                        if not ok:
                            print("   " + msg[:400], flush=True)
                    if res:
                        print("   e.g. " + res[0][1][:300], flush=True)
    print("summary: %d differing runs" % bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
