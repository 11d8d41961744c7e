"""Every device row of one of the reference's bitstream-regression tables (test/encoder_binary_comparison/SHA1Table/:
BA_MW_D.264_AllCases_SHA1_Table.csv, camera video; Adobe_PDF_sample_a_1024x768_50Frms.264_AllCases_SHA1_Table.csv, screen
content) through the dispatch-table binding, the way tests/test_hooks_sha1.py / tests/test_hooks_screen.py run their samples.
usage: sha1_table_rows.py [--table ba|adobe] [--lib path] [--workers N] [--stride K] [--dynslice]      (default library: openh264_amd/libwelship.so)
--dynslice: the table's size-limited rows (-slcmd 3: 512 / 256; half of them with as many slice threads as the machine has cores, up to four)
with WELS_HIP_DYNSLICE=1 instead."""
import argparse, os, pathlib, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_hooks_sha1 as T_BA
import test_hooks_screen as T_ADOBE


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "openh264_amd", "libwelship.so"))
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--table", default="ba", choices=("ba", "adobe"))
    ap.add_argument("--gom", type=int, default=1, help="WELS_HIP_GOM: 1 = single-slice rate-controlled pictures group by group, 2 = their QP recursion inside the kernel")
    ap.add_argument("--dynslice", action="store_true")
    a = ap.parse_args()
    a.lib = os.path.abspath(a.lib)
    T = T_BA if a.table == "ba" else T_ADOBE
    clip = "BA_MW_D.264" if a.table == "ba" else T_ADOBE.CLIP
    min_pics = 40 if a.table == "ba" else 1       # the screen table's 450 kbps rows skip most of their 30 frames
    d = pathlib.Path(tempfile.mkdtemp())
    subprocess.check_call([os.path.join(T.REF, "ref_dec"), os.path.join(T.RES, clip), str(d / (clip + ".yuv"))], stdout=subprocess.DEVNULL)
    for k in range(4):
        (d / ("layer%d.cfg" % k)).write_bytes(open(os.path.join(T.RES, "layer2.cfg"), "rb").read())
    (d / "welsenc.cfg").write_bytes(open(os.path.join(T.RES, "welsenc.cfg"), "rb").read())
    rows = (T._size_limited_rows(("0", "1")) if a.dynslice else T._device_rows())[::a.stride]
    t0 = time.time()

    def one(ir):
        i, r = ir
        got, pics, err = T._run_row(d, a.lib, r, "w%d" % i, {"WELS_HIP_GOM": str(a.gom), "WELS_HIP_DYNSLICE": "1" if a.dynslice else "0"})
        if a.dynslice:
            done = [l for l in err.splitlines() if "picture complete" in l]
            dyn[0] += sum(int(l.split("complete:")[1].split()[0]) for l in done); dyn[1] += sum(int(l.split("slices,")[1].split()[0]) for l in done)
            dyn[2] += sum(int(l.split("calls,")[1].split()[0]) for l in done); dyn[3] += sum(int(l.split("coded for")[1].split()[0]) for l in done)
        if a.table == "ba": os.remove(str(d / ("t_w%d.264" % i)))
        return i, got == r[0] and pics >= min_pics and "welship hooks: installed" in err, r[4]["-slcmd 0"], got, pics, err.count("GOM-level QP")

    bad, by_mode, total_pics, ranged = [], {}, 0, 0
    dyn = [0, 0, 0, 0]
    with ThreadPoolExecutor(a.workers) as ex:
        for i, ok, mode, got, pics, rg in ex.map(one, enumerate(rows)):
            total_pics += pics
            ranged += rg
            by_mode.setdefault(mode, [0, 0])
            by_mode[mode][0] += 1
            if not ok:
                by_mode[mode][1] += 1
                bad.append((i, rows[i][4], got, pics))
    for mode in sorted(by_mode):
        print("-slcmd %s : rows %d bad %d" % (mode, by_mode[mode][0], by_mode[mode][1]))
    print("table %s: device rows %d of the table's %d, bad %d, %d pictures coded on the device, %.1f s, %d workers, library %s" % (os.path.basename(T.TABLE), len(rows), len(T._rows()), len(bad), total_pics, time.time() - t0, a.workers, os.path.basename(a.lib)))
    print("WELS_HIP_GOM=%d: %d pictures were coded group by group (one device call per group of macroblocks)" % (a.gom, ranged))
    if a.dynslice: print("WELS_HIP_DYNSLICE=1: %d slices, %d device calls that coded macroblocks (+ one closing call per picture); %d macroblocks coded for %d (x%.2f: what coding ahead of the writer costs)" % (dyn[0], dyn[1], dyn[2], dyn[3], dyn[2] / max(1, dyn[3])))
    for b in bad[:10]:
        print("BAD", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
