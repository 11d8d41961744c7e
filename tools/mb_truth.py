"""Per-macroblock ground truth: where do two runs of the same command line part ways?  

The hooked reference (oracle/_ref/ref_enc_hip = the reference with integration/openh264_hip.patch + welship_hooks.cpp) writes one line per
macroblock the entropy writer is handed when WELS_HIP_MB_TRACE=<file> is set (welship_hooks.cpp TraceWriteMbSyn, wrapped around
SWelsFuncPtrList::pfWelsSpatialWriteMbSyn): picture, layer, address, type, cbp, QP, the vector differences, reference indices, total_coeff,
intra modes and a hash of the coded coefficient levels.  This tool runs the command line twice -- the reference's own C path (WELS_HIP=0) and
the path under test (the device library, or the wave emulation of the same kernels with --emu) -- and prints the first macroblocks whose
lines differ, with both lines, and the totals per picture.  Byte-identical streams give "0 macroblocks differ".

usage: mb_truth.py [--emu | --lib <libwelship*.so>] [--keep <dir>] [--env NAME=VALUE ...] -- <ref_enc flags ...>
  e.g. mb_truth.py --emu -- -i clip.yuv -w 320 -h 192 -rc -1 -qp 26 -frames 8 -slcmd 1 -slcnum 2
The flags are those of oracle/ref_enc_driver.cpp (-i, -w, -h and -o are required by it; -o is supplied here)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")


def run(flags, out_dir, tag, env_extra):
    env = dict(os.environ, WELS_HIP_MB_TRACE=os.path.join(out_dir, tag + ".mbs"), WELS_HIP_TRACE="1")
    env.update(env_extra)
    cmd = [os.path.join(REF, "ref_enc_hip")] + flags + ["-o", os.path.join(out_dir, tag + ".264"), "-quiet"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=3600)
    if p.returncode != 0:
        raise RuntimeError("%s: exit code %d\n%s" % (tag, p.returncode, p.stderr.decode(errors="replace")[-2000:]))
    return p.stderr.decode(errors="replace")


def load(path):
    """{(pic, layer, mb): line} -- a macroblock coded twice (size-limited slices) keeps its last line."""
    mbs, order = {}, []
    for line in open(path):
        f = line.split()
        if len(f) < 6 or f[0] != "pic":
            continue
        key = (int(f[1]), int(f[3]), int(f[6]))
        if key not in mbs:
            order.append(key)
        mbs[key] = line.rstrip("\n")
    return mbs, order


def compare(a_path, b_path, limit=8, out=sys.stdout):
    a, order = load(a_path)
    b, _ = load(b_path)
    bad = [k for k in order if b.get(k) != a[k]] + [k for k in b if k not in a]
    per_pic = {}
    for k in order:
        per_pic.setdefault(k[:2], [0, 0])[0] += 1
    for k in bad:
        per_pic.setdefault(k[:2], [0, 0])[1] += 1
    print("%d macroblocks in %d pictures, %d differ" % (len(order), len(per_pic), len(bad)), file=out)
    for k in bad[:limit]:
        print("  picture %d layer %d macroblock %d\n    C path: %s\n    tested: %s" % (k[0], k[1], k[2], a.get(k, "(missing)"), b.get(k, "(missing)")), file=out)
    if bad:
        first = sorted(p for p, c in per_pic.items() if c[1])[:6]
        print("  first pictures with differences (picture, layer: macroblocks): " + ", ".join("%d,%d: %d" % (p[0], p[1], per_pic[p][1]) for p in first), file=out)
    return len(bad)


def main(argv):
    if "--" not in argv:
        print(__doc__)
        return 2
    own, flags = argv[:argv.index("--")], argv[argv.index("--") + 1:]
    env_t = {}
    keep = None
    i = 0
    while i < len(own):
        if own[i] == "--emu":
            from openh264_amd import build as B
            env_t["WELSHIP_LIB"] = B.build_emu()
        elif own[i] == "--lib":
            env_t["WELSHIP_LIB"] = os.path.abspath(own[i + 1]); i += 1
        elif own[i] == "--keep":
            keep = own[i + 1]; i += 1
        elif own[i] == "--env":
            k, v = own[i + 1].split("=", 1); env_t[k] = v; i += 1
        else:
            print("unknown option", own[i]); return 2
        i += 1
    work = keep or tempfile.mkdtemp(prefix="mb_truth_")
    os.makedirs(work, exist_ok=True)
    run(flags, work, "c_path", {"WELS_HIP": "0"})
    err = run(flags, work, "tested", env_t)
    on_device = err.count("welship hooks: did")
    same = open(os.path.join(work, "c_path.264"), "rb").read() == open(os.path.join(work, "tested.264"), "rb").read()
    print("streams %s; %d pictures went through the hooks%s" % ("equal" if same else "DIFFER", on_device, "" if on_device else " (the installer declined: both runs are the C path)"))
    n = compare(os.path.join(work, "c_path.mbs"), os.path.join(work, "tested.mbs"))
    if keep is None and n == 0 and same:
        import shutil
        shutil.rmtree(work, ignore_errors=True)
    else:
        print("files kept in", work)
    return 0 if (n == 0 and same) else 1


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    sys.exit(main(sys.argv[1:]))
