"""Screen content (iUsageType == SCREEN_CONTENT_REAL_TIME) through the dispatch-table binding: SURVEY 8(f) 4.

The reference's second checked-in bitstream-regression table, test/encoder_binary_comparison/SHA1Table/
Adobe_PDF_sample_a_1024x768_50Frms.264_AllCases_SHA1_Table.csv (1152 rows, `-utype 1`, rate control on in every row), run
exactly like the BA_MW_D table of tests/test_hooks_sha1.py: the reference's console encoder with this repository's engine
behind SWelsFuncPtrList, the row's options on the command line, SHA1 of the bitstream against the table.  The unmodified
reference built from C reproduces the table (checked when this test was written), so the table pins the C path this engine
restates: static / scrolled P_Skip (svc_mode_decision.cpp:326-541), the cross search of the 16x16 block, the directional
(scroll) vector and the feature search of the 8x8 blocks (svc_motion_estimate.cpp:385-1097), TryModeMerge and the sub-block
scoring of merged partitions (svc_mode_decision.cpp:553-667, md.cpp:575-640).

Device rows: slice modes 0, 1, 2 (896 of the 1152); the 256 size-limited-slice rows keep the C path as before.  Also here: the
screen-content entries of the API golden hashes (test/api/encoder_test.cpp:146-157) whose input files exist in res/.

CPU tier: the wave emulation of the kernel sources (tests/emu); GPU tier (-m gpu): libwelship.so on the MI355X.
"""
import csv
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
RES = os.path.join(REF, "res")
CLIP = "Adobe_PDF_sample_a_1024x768_50Frms.264"
TABLE = os.path.join(RES, CLIP + "_AllCases_SHA1_Table.csv")
H264ENC = os.path.join(REF, "h264enc_hiphooks")
YUV_SHA1 = "9aa9a4d9598eb3e1093311826844f37c43e4c521"
pytestmark = pytest.mark.skipif(not (os.path.exists(TABLE) and os.path.exists(H264ENC)), reason="oracle/_ref (hooked reference + table) not built")


def _rows():
    rows = list(csv.reader(open(TABLE)))
    hdr = [h.strip() for h in rows[0]]
    out = []
    for r in rows[1:]:
        vals = [v.strip() for v in r]
        out.append((vals[0], vals[1], hdr[2:], vals[2:], dict(zip(hdr, vals))))
    return out


def _device_rows():
    return [r for r in _rows() if r[4]["-slcmd 0"] in ("0", "1", "2")]


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, ref_tools):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    d = tmp_path_factory.mktemp("sha1table_screen")
    subprocess.check_call([ref_tools["dec"], os.path.join(RES, CLIP), str(d / (CLIP + ".yuv"))], stdout=subprocess.DEVNULL)
    assert hashlib.sha1((d / (CLIP + ".yuv")).read_bytes()).hexdigest() == YUV_SHA1       # the table's InputYUVSHA1 column
    for k in range(4):
        (d / ("layer%d.cfg" % k)).write_bytes(open(os.path.join(RES, "layer2.cfg"), "rb").read())
    (d / "welsenc.cfg").write_bytes(open(os.path.join(RES, "welsenc.cfg"), "rb").read())
    return d


def _run_row(workdir, lib, row, tag, extra_env=None):
    sha, yuv_sha, keys, vals, _ = row
    opts = []
    for k, v in zip(keys, vals):
        opts += k.split() + [v]
    opts = [o if o != "bgd" else "-bgd" for o in opts]
    out = str(workdir / ("t_%s.264" % tag))
    env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")
    env.update(extra_env or {})
    p = subprocess.run([H264ENC, "welsenc.cfg", "-lconfig", "0", "layer0.cfg", "-lconfig", "1", "layer1.cfg", "-lconfig", "2", "layer2.cfg",
                        "-lconfig", "3", "layer3.cfg", "-bf", out, "-org", str(workdir / (CLIP + ".yuv"))] + opts,
                       cwd=str(workdir), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    got = hashlib.sha1(open(out, "rb").read()).hexdigest()
    os.remove(out)
    return got, err.count("welship hooks: did"), err


def _check(workdir, lib, rows, workers=8):
    from concurrent.futures import ThreadPoolExecutor

    def one(ir):
        i, row = ir
        got, pictures, err = _run_row(workdir, lib, row, str(i))
        # 30 frames per case; the 450 kbps rows skip most of them (frame skipping is on in every row), but the IDR is always coded
        return (row[4], row[0], got, pictures) if (got != row[0] or pictures < 1 or "welship hooks: installed" not in err) else None

    with ThreadPoolExecutor(workers) as ex:
        bad = [b for b in ex.map(one, enumerate(rows)) if b]
    assert not bad, "%d of %d rows differ, first: %s" % (len(bad), len(rows), bad[0])


def _sample(rows, n):
    step = max(1, len(rows) // n)
    return [rows[(i * step + (i % 5)) % len(rows)] for i in range(n)]


def test_table_shape():
    rows = _rows()
    assert len(rows) == 1152 and all(r[1] == YUV_SHA1 and r[4]["-utype"] == "1" for r in rows)
    assert len(_device_rows()) == 896
    assert len({r[0] for r in _device_rows()}) >= 20          # distinct streams among them


def test_screen_table_rows_on_emulation(workdir, emu_lib):
    """A 32-row sample (tools/sha1_table_rows.py --table adobe runs all 896: profiles/r02_sha1_table_screen_*)."""
    _check(workdir, emu_lib, _sample(_device_rows(), 32))


def test_every_screen_content_path_is_reached(workdir, emu_lib):
    """The table's clip reaches every screen-content branch of the kernels (the test build counts them): a row that passes has
    exercised static and scrolled skips, the P16x16 of a static block, both line searches, the feature search (with hits), the
    directional vector, P8x8 with fixed-vector blocks and merged partitions."""
    row = _device_rows()[0]
    got, pictures, err = _run_row(workdir, emu_lib, row, "stat", {"WELSHIP_SCC_STATS": "1"})
    assert got == row[0]
    stat = {l.split()[3]: int(l.split()[4]) for l in err.splitlines() if l.startswith("welship scc stat")}
    assert len(stat) == 11 and all(v > 0 for v in stat.values()), stat


def test_switched_off_size_limited_rows_stay_on_the_c_path(workdir, emu_lib):
    row = [r for r in _rows() if r[4]["-slcmd 0"] == "3"][0]
    got, pictures, err = _run_row(workdir, emu_lib, row, "c0", {"WELS_HIP_DYNSLICE": "0"})
    assert "not installed" in err and pictures == 0 and got == row[0]


def _size_limited_rows(threads=("1",)):
    return [r for r in _rows() if r[4]["-slcmd 0"] == "3" and r[4]["-thread"] in threads]


def test_size_limited_rows_on_emulation(workdir, emu_lib):
    """... unless asked for (WELS_HIP_DYNSLICE=1): the device codes ahead of the entropy writer, the chain of the 8x8 searches
    is kept per macroblock.  A sample of the 128 single-thread rows (and, on a machine with four cores, of the 128 that take the core
    count); all 256: profiles/r02_size_limited_slices_emulation.txt (tools/sha1_table_rows.py --table adobe --dynslice)."""
    from concurrent.futures import ThreadPoolExecutor
    rows = _size_limited_rows()
    assert len(rows) == 128
    rows = rows[::8] + (_size_limited_rows(("0",))[::16] if (os.cpu_count() or 1) >= 4 else [])

    def one(ir):
        i, row = ir
        got, pictures, err = _run_row(workdir, emu_lib, row, "d%d" % i, {"WELS_HIP_DYNSLICE": "1"})
        return None if (got == row[0] and "welship hooks: installed" in err and err.count("picture complete") >= 1) else (row[4], got)

    with ThreadPoolExecutor(8) as ex:
        bad = [b for b in ex.map(one, enumerate(rows)) if b]
    assert not bad, "%d rows differ, first: %s" % (len(bad), bad[0])


API_GOLDEN_SCREEN = [  # test/api/encoder_test.cpp:146-157 (SEncParamBase, SCREEN_CONTENT_REAL_TIME: RC quality mode, 5 Mbps, one slice)
    ("CiscoVT2people_320x192_12fps.yuv", 320, 192, 12.0, "fd57470eebb9b334e8edcb8b47f7fb5b5868f111"),
    ("CiscoVT2people_160x96_6fps.yuv", 160, 96, 6.0, "5f63e723c3ec82fad186b48fcbcfb54730ce3b26"),
    ("Static_152_100.yuv", 152, 100, 6.0, "e77a5b0ffb48753556e617544616fb06a049e9be"),
]


def _api_hash(exe, lib, tmp_path, name, w, h, fps):
    out = str(tmp_path / "o.264")
    env = dict(os.environ, WELSHIP_LIB=lib or "", WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")
    p = subprocess.run([os.path.join(REF, exe), "-i", os.path.join(RES, name), "-w", str(w), "-h", str(h), "-o", out, "-base", "-usage", "1", "-rc", "0",
                        "-fps", str(fps), "-quiet"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    return hashlib.sha1(open(out, "rb").read()).hexdigest(), err


@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN_SCREEN)
def test_oracle_reproduces_the_screen_api_hashes(ref_tools, tmp_path, name, w, h, fps, sha):
    """Pins oracle/_ref itself: the unmodified reference built from C."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    assert _api_hash("ref_enc", None, tmp_path, name, w, h, fps)[0] == sha


@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN_SCREEN)
def test_screen_api_hash_through_the_hooks_on_emulation(emu_lib, tmp_path, name, w, h, fps, sha):
    got, err = _api_hash("ref_enc_hip", emu_lib, tmp_path, name, w, h, fps)
    assert "welship hooks: installed" in err and err.count("welship hooks: did") >= 5
    assert got == sha


@pytest.mark.gpu
def test_size_limited_screen_rows_on_the_mi355x(workdir, hip_lib):
    """Slice mode 3 with screen content on the device: eight of the table's single-thread rows (one stream from the unmodified reference)."""
    rows = _size_limited_rows()[5::16]
    assert len(rows) == 8
    for i, row in enumerate(rows):
        got, pictures, err = _run_row(workdir, hip_lib, row, "dg%d" % i)
        assert "welship hooks: installed" in err and err.count("picture complete") >= 1, err[-1500:]
        assert got == row[0], (row[4], got)


@pytest.mark.gpu
def test_screen_table_rows_on_the_mi355x(workdir, hip_lib):
    """A 96-row sample of the 896 device rows (round 6: 24 before; seconds per row on the MI355X).  The whole table: tools/sha1_table_rows.py --table adobe."""
    _check(workdir, hip_lib, _sample(_device_rows(), 96), workers=16)


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN_SCREEN)
def test_screen_api_hash_through_the_hooks_on_the_mi355x(hip_lib, tmp_path, name, w, h, fps, sha):
    got, err = _api_hash("ref_enc_hip", hip_lib, tmp_path, name, w, h, fps)
    assert "welship hooks: installed" in err and err.count("welship hooks: did") >= 5
    assert got == sha


# ---- randomised screen-content sessions (tools/fuzz_screen.py): synthetic scrolling documents, random parameters ------------
def _fuzz(lib, tmp_path, seeds, usage=1, qp_max=None):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_screen
    from concurrent.futures import ThreadPoolExecutor
    fuzz_screen.QP_MAX = qp_max
    try:
        with ThreadPoolExecutor(8) as ex:
            res = list(ex.map(lambda s: fuzz_screen.run_case(s, lib, str(tmp_path), usage), seeds))
    finally:
        fuzz_screen.QP_MAX = None
    bad = [r for r in res if r[1] == "DIFF"]
    assert not bad, bad[0]
    assert sum(r[1] == "ok" for r in res) >= len(res) // 2        # (the rest: parameter sets the reference itself rejects)


def test_screen_fuzz_on_emulation(emu_lib, tmp_path):
    _fuzz(emu_lib, tmp_path, range(0, 16))


@pytest.mark.gpu
def test_screen_fuzz_on_the_mi355x(hip_lib, tmp_path):
    _fuzz(hip_lib, tmp_path, range(100, 124))


# ---- TRY_REENCODING through the binding: constant QP 0..12 on the same noisy synthetic clips makes CAVLC levels overflow; the
# reference codes such a macroblock again at QP + 2 (svc_encode_slice.cpp:564-576,1845-1867), the hooks repeat the picture on the
# device with that macroblock's QP raised (WelsHipFrameJob::bRetry).  A third of these cases re-encode macroblocks (tens of passes).
def test_reencoding_after_cavlc_overflow_on_emulation(emu_lib, tmp_path):
    _fuzz(emu_lib, tmp_path, range(7000, 7016), usage=0, qp_max=12)
    _fuzz(emu_lib, tmp_path, range(8000, 8008), usage=1, qp_max=12)


@pytest.mark.gpu
def test_reencoding_after_cavlc_overflow_on_the_mi355x(hip_lib, tmp_path):
    _fuzz(hip_lib, tmp_path, range(7016, 7032), usage=0, qp_max=12)
