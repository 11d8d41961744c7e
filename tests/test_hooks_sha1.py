"""The metric's own parity gate: rows of the reference's checked-in bitstream-regression table
(test/encoder_binary_comparison/SHA1Table/BA_MW_D.264_AllCases_SHA1_Table.csv) run through the reference's own console
encoder with this repository's engine installed behind SWelsFuncPtrList (integration/openh264_hip.patch +
integration/welship_hooks.cpp -> oracle/_ref/h264enc_hiphooks), exactly the way
test/encoder_binary_comparison/Scripts/run_BinarySHA1Comparison.sh:165-241 runs them: welsenc.cfg + layer2.cfg copied as
layer0..3.cfg, the row's options on the command line, SHA1 of the bitstream against the table's first column.

Rows the dispatch-table binding takes to the device: every option combination of the table (rate-control mode 1 and 3,
1 and 3 temporal layers, LTR, denoising, scene-change detection, background detection, frame skipping) in all four slice modes.
Slice modes 0, 1 and 2 are 1792 of the 2304 rows; the 512 size-limited rows (-slcmd 3) run on the device by default as well
(WELS_HIP_DYNSLICE=0 hands them back to the reference's C path -- the hooks report that, and a test checks it).  A deterministic
sample of the size-limited rows (one slice thread) sits here, early in the GPU tier; the randomised sessions are tests/test_hooks_dynslice.py.

Also here: the reference's API-level golden hashes (test/api/encoder_test.cpp:104-115) and its stock testbin/welsenc.cfg
through the same binding.

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
TABLE = os.path.join(RES, "BA_MW_D.264_AllCases_SHA1_Table.csv")
H264ENC = os.path.join(REF, "h264enc_hiphooks")
pytestmark = pytest.mark.skipif(not (os.path.exists(TABLE) and os.path.exists(H264ENC)), reason="oracle/_ref (hooked reference + table) not built")


def _rows():
    rows = list(csv.reader(open(TABLE)))
    hdr = [h.strip() for h in rows[0]]
    out = []
    for r in rows[1:]:
        vals = [v.strip() for v in r]
        d = dict(zip(hdr, vals))
        out.append((vals[0], vals[1], hdr[2:], vals[2:], d))
    return out


def _device_rows():
    """Rows the binding installs the hooks for: everything but size-limited slices (-slcmd 3).  -slcmd 1 (4 or 7 slices) has a
    frame-constant QP; -slcmd 0 and -slcmd 2 (one slice of 960 MBs = the whole QCIF picture) run GOM-level rate control,
    which the binding codes group by group."""
    return [r for r in _rows() if r[4]["-slcmd 0"] in ("0", "1", "2")]


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, ref_tools):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    d = tmp_path_factory.mktemp("sha1table")
    subprocess.check_call([ref_tools["dec"], os.path.join(RES, "BA_MW_D.264"), str(d / "BA_MW_D.264.yuv")], stdout=subprocess.DEVNULL)
    for k in range(4):
        (d / ("layer%d.cfg" % k)).write_bytes(open(os.path.join(RES, "layer2.cfg"), "rb").read())
    (d / "welsenc.cfg").write_bytes(open(os.path.join(RES, "welsenc.cfg"), "rb").read())
    return d


def _run_row(workdir, lib, row, tag, extra_env=None):
    sha, yuv_sha, keys, vals, _ = row
    opts = []
    for k, v in zip(keys, vals):
        opts += k.split() + [v]
    opts = [o if o != "bgd" else "-bgd" for o in opts]       # the table's header spells this one column without its dash
    out = str(workdir / ("t_%s.264" % tag))
    env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")      # product defaults: the one-slice rows run their GOM-level QP inside the kernel (P pictures) or group by group
    env.update(extra_env or {})
    p = subprocess.run([H264ENC, "welsenc.cfg", "-lconfig", "0", "layer0.cfg", "-lconfig", "1", "layer1.cfg", "-lconfig", "2", "layer2.cfg",
                        "-lconfig", "3", "layer3.cfg", "-bf", out, "-org", str(workdir / "BA_MW_D.264.yuv")] + opts,
                       cwd=str(workdir), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    pictures = err.count("welship hooks: did")
    return hashlib.sha1(open(out, "rb").read()).hexdigest(), pictures, err


def _check(workdir, lib, rows, workers=8):
    from concurrent.futures import ThreadPoolExecutor

    def one(ir):
        i, row = ir
        got, pictures, err = _run_row(workdir, lib, row, str(i))
        os.remove(str(workdir / ("t_%d.264" % i)))
        return (row[4], row[0], got, pictures) if (got != row[0] or pictures < 40) else None      # 50 frames per case; rate control may skip a few

    with ThreadPoolExecutor(workers) as ex:
        bad = [b for b in ex.map(one, enumerate(rows)) if b]
    assert not bad, "%d of %d rows differ, first: %s" % (len(bad), len(rows), bad[0])


def test_table_shape():
    rows = _rows()
    assert len(rows) == 2304
    dev = _device_rows()
    assert len(dev) == 1792
    assert len({r[0] for r in dev}) >= 8            # distinct streams among them
    assert all(r[1] == "afd7a9765961ca241bb4bdf344b31397bec7465a" for r in rows)


def _sample(rows, n):
    """An even sample that keeps every option of the table represented (the row order cycles through them)."""
    step = max(1, len(rows) // n)
    return [rows[(i * step + (i % 7)) % len(rows)] for i in range(n)]


def test_sha1_table_rows_on_emulation(workdir, emu_lib):
    """Every fourth device row (448 rows; tools/sha1_table_rows.py runs all 1792: profiles/r02_sha1_table_all_*)."""
    _check(workdir, emu_lib, _device_rows()[1::4])


def test_switched_off_rows_stay_on_the_c_path(workdir, emu_lib):
    """With WELS_HIP_DYNSLICE=0 the hooks decline size-limited slices: the reference codes the stream itself and still matches the table."""
    rows = [r for r in _rows() if r[4]["-slcmd 0"] == "3"][:3]
    for i, row in enumerate(rows):
        got, pictures, err = _run_row(workdir, emu_lib, row, "c%d" % i, {"WELS_HIP_DYNSLICE": "0"})
        assert "not installed" in err and pictures == 0
        assert got == row[0]


def _size_limited_rows(threads=("1",)):
    """The table's -slcmd 3 rows: 256 that name one slice thread (-thread 1) and 256 that take the machine's core count (-thread 0: up to
    four slice threads, one partition of the picture each -- the table's hashes are those of a machine with at least four cores)."""
    return [r for r in _rows() if r[4]["-slcmd 0"] == "3" and r[4]["-thread"] in threads]


def test_size_limited_rows_on_emulation(workdir, emu_lib):
    """Size-limited slices on request (WELS_HIP_DYNSLICE=1): the device codes ahead of the entropy writer, and where
    the writer ends a slice (DynSlcJudgeSliceBoundaryStepBack, svc_encode_slice.cpp:1741-1790) the next slice begins with another device
    call that codes the macroblocks from there on again -- without the neighbours that now belong to the slice before.  Every second
    of the 256 single-thread rows and every eighth of the others here; all 512: profiles/r02_size_limited_slices_emulation.txt
    (tools/sha1_table_rows.py --dynslice)."""
    from concurrent.futures import ThreadPoolExecutor
    rows = _size_limited_rows()
    assert len(rows) == 256
    if (os.cpu_count() or 1) >= 4:          # the -thread 0 rows: four partitions per picture, coded by four slice tasks at once
        rows = rows + _size_limited_rows(("0",))[::4]

    def one(ir):
        i, row = ir
        got, pictures, err = _run_row(workdir, emu_lib, row, "d%d" % i, {"WELS_HIP_DYNSLICE": "1"})
        os.remove(str(workdir / ("t_d%d.264" % i)))
        multi = sum(1 for l in err.splitlines() if "picture complete" in l and " 1 slices" not in l)
        return None if (got == row[0] and "welship hooks: installed" in err and err.count("picture complete") >= 40) else (row[4], got), multi

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, enumerate(rows[::2])))
    bad = [b for b, _ in res if b]
    assert not bad, "%d rows differ, first: %s" % (len(bad), bad[0])
    assert sum(m for _, m in res) > 100          # pictures of more than one slice really occurred (the IDR pictures at least)


def test_gom_sessions_can_be_switched_off(emu_lib, tmp_path):
    """Rate control with one slice per picture (the reference's DEFAULT parameter set) runs on the device by default -- the groups' QP
    recursion inside the kernel for P pictures, one call per group otherwise; WELS_HIP_GOM=0 leaves such sessions to the C path and
    the installer says so."""
    out = str(tmp_path / "o.264")
    env = dict(os.environ, WELSHIP_LIB=emu_lib, WELS_HIP_TRACE="1")
    env["WELS_HIP_GOM"] = "0"
    name, w, h, fps, sha = API_GOLDEN[0]
    p = subprocess.run([os.path.join(REF, "ref_enc_hip"), "-i", os.path.join(RES, name), "-w", str(w), "-h", str(h), "-o", out, "-base", "-rc", "0",
                        "-fps", str(fps), "-quiet"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "not installed" in err and "GOM" in err and "welship hooks: did" not in err
    assert hashlib.sha1(open(out, "rb").read()).hexdigest() == sha


@pytest.mark.gpu
def test_sha1_table_rows_on_the_mi355x(workdir, hip_lib):
    _check(workdir, hip_lib, _sample(_device_rows(), 128))


@pytest.mark.gpu
def test_size_limited_rows_on_the_mi355x(workdir, hip_lib):
    """Slice mode 3 on the device (the installer's default): sixteen of the table's single-thread rows -- one slice thread, so the unmodified
    reference writes ONE stream for them and the table's hash is the only acceptable answer (DynSlcJudgeSliceBoundaryStepBack,
    svc_encode_slice.cpp:1741-1793; WelsMdInterMbLoopOverDynamicSlice :1901-2010)."""
    rows = _size_limited_rows()[3::16]
    assert len(rows) == 16
    for i, row in enumerate(rows):
        got, pictures, err = _run_row(workdir, hip_lib, row, "dg%d" % i)
        os.remove(str(workdir / ("t_dg%d.264" % i)))
        assert "welship hooks: installed" in err and err.count("picture complete") >= 40, err[-1500:]
        assert got == row[0], (row[4], got)


@pytest.mark.gpu
def test_sha1_table_rows_on_the_mi355x_second_sample(workdir, hip_lib):
    """Another sample of the device rows (the odd half of a 96-row draw; rounds 3-4 ran it through the run scheduler, removed in round 5)."""
    _check(workdir, hip_lib, _sample(_device_rows(), 96)[1::2])


# ---- the one-slice rows with the groups' QP recursion INSIDE the kernel (WELS_HIP_GOM=2) ------------------------
# The device counts every macroblock's CAVLC bits (kernels/cavlc_bits.h), the last macroblock of a group of macroblocks adds the
# skip-run / mb_qp_delta terms in coding order and runs RcCalculateGomQp / RcGomTargetBits (common/gom_rc.h), the next group reads
# its QP -- one device call per picture (P pictures since round 2, I pictures since round 5: k_intra_slice walks the picture's own order,
# waits for the group before and closes its group the same way).  The reference's own rate control still runs in the slice loop on the real bit
# positions: the hooks compare every macroblock's QP with the record's (and, with WELS_HIP_CHECK_BITS, every bit count).
def _gom_rows():
    return [r for r in _device_rows() if r[4]["-slcmd 0"] in ("0", "2")]


def _check_gom_kernel(workdir, lib, rows):
    from concurrent.futures import ThreadPoolExecutor

    def one(ir):
        i, row = ir
        got, pictures, err = _run_row(workdir, lib, row, "g%d" % i, {"WELS_HIP_GOM": "2"})
        os.remove(str(workdir / ("t_g%d.264" % i)))
        by_group = err.count("GOM-level QP")          # pictures that still went group by group: none since round 5 (I pictures run the recursion in the intra kernel)
        return (row[4], row[0], got, pictures, by_group) if (got != row[0] or pictures < 40 or by_group != 0) else None

    with ThreadPoolExecutor(8) as ex:
        bad = [b for b in ex.map(one, enumerate(rows)) if b]
    assert not bad, "%d of %d rows differ, first: %s" % (len(bad), len(rows), bad[0])


def test_gom_rate_control_inside_the_kernel_on_emulation(workdir, emu_lib):
    rows = _gom_rows()
    assert len(rows) == 768
    _check_gom_kernel(workdir, emu_lib, rows[2::8])


@pytest.mark.gpu
def test_gom_rate_control_inside_the_kernel_on_the_mi355x(workdir, hip_lib):
    _check_gom_kernel(workdir, hip_lib, _gom_rows()[5::16])


def _check_gom_per_group(workdir, lib, rows):
    """WELS_HIP_GOM=1: every group of macroblocks is a device call of its own (what I pictures, screen content and CABAC sessions
    always do; forced here for the P pictures too)."""
    for i, row in enumerate(rows):
        got, pictures, err = _run_row(workdir, lib, row, "pg%d" % i, {"WELS_HIP_GOM": "1"})
        os.remove(str(workdir / ("t_pg%d.264" % i)))
        assert got == row[0] and pictures >= 40 and err.count("GOM-level QP") > pictures // 2, (row[4], got, pictures)


def test_gom_rate_control_group_by_group_on_emulation(workdir, emu_lib):
    _check_gom_per_group(workdir, emu_lib, [r for r in _gom_rows() if r[4]["-slcmd 0"] == "0"][3::64])


@pytest.mark.gpu
def test_gom_rate_control_group_by_group_on_the_mi355x(workdir, hip_lib):
    _check_gom_per_group(workdir, hip_lib, [r for r in _gom_rows() if r[4]["-slcmd 0"] == "0"][7::64])


# ---- the API-level golden hashes and the stock configuration through the binding ---------------------------------------------
API_GOLDEN = [  # test/api/encoder_test.cpp:104-115 (SEncParamBase: RC quality mode, 5 Mbps, one slice -> GOM-level QP)
    ("CiscoVT2people_160x96_6fps.yuv", 160, 96, 6.0, "08ade1853e4e49d50be675393780e75519586143"),
    ("CiscoVT2people_320x192_12fps.yuv", 320, 192, 12.0, "672a52fb6b6e6d52b5b3f3480d13d44e88481fb9"),
    ("Static_152_100.yuv", 152, 100, 6.0, "e60f12e3c24500d4306d812b0811d3c21855dd1c"),
]


def _api_hash(lib, tmp_path, name, w, h, fps, gom=None):
    out = str(tmp_path / "o.264")
    env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")
    env.pop("WELS_HIP_GOM", None)          # default: the groups' QP recursion inside the kernel for P pictures (= "2")
    if gom: env["WELS_HIP_GOM"] = gom
    p = subprocess.run([os.path.join(REF, "ref_enc_hip"), "-i", os.path.join(RES, name), "-w", str(w), "-h", str(h), "-o", out, "-base", "-rc", "0",
                        "-fps", str(fps), "-quiet"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    assert "welship hooks: installed" in err and err.count("welship hooks: did") >= 5
    if gom is None:          # quality-mode rate control gives I pictures a GOM-level QP too: the IDR picture of these sessions is ONE device call as well
        assert err.count("GOM-level QP") == 0, err[-1500:]
    return hashlib.sha1(open(out, "rb").read()).hexdigest()


@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN)
def test_api_golden_hash_through_the_hooks_on_emulation(emu_lib, tmp_path, name, w, h, fps, sha):
    assert _api_hash(emu_lib, tmp_path, name, w, h, fps) == sha


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN)
def test_api_golden_hash_through_the_hooks_on_the_mi355x(hip_lib, tmp_path, name, w, h, fps, sha):
    assert _api_hash(hip_lib, tmp_path, name, w, h, fps) == sha


@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN[:2])
def test_api_golden_hash_gom_group_by_group_on_emulation(emu_lib, tmp_path, name, w, h, fps, sha):
    assert _api_hash(emu_lib, tmp_path, name, w, h, fps, gom="1") == sha


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,fps,sha", API_GOLDEN[:2])
def test_api_golden_hash_gom_group_by_group_on_the_mi355x(hip_lib, tmp_path, name, w, h, fps, sha):
    assert _api_hash(hip_lib, tmp_path, name, w, h, fps, gom="1") == sha


def _stock_cfg(lib, tmp_path):
    """testbin/welsenc.cfg as it is (RCMode 0, two temporal layers, LTR, background and scene-change detection, adaptive
    quantisation on): the stock front-end on the reference and on the reference with the hooks must write the same file."""
    cfg = open(os.path.join(RES, "welsenc.cfg")).read().replace("../res/CiscoVT2people_320x192_12fps.yuv", os.path.join(RES, "CiscoVT2people_320x192_12fps.yuv"))
    (tmp_path / "welsenc.cfg").write_text(cfg)
    (tmp_path / "layer2.cfg").write_bytes(open(os.path.join(RES, "layer2.cfg"), "rb").read())
    subprocess.check_call([os.path.join(REF, "h264enc_ref"), "welsenc.cfg", "-bf", "ref.264"], cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")
    p = subprocess.run([H264ENC, "welsenc.cfg", "-bf", "hip.264"], cwd=str(tmp_path), env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "welship hooks: installed" in err and err.count("welship hooks: did") >= 5, err[-2000:]
    assert (tmp_path / "ref.264").read_bytes() == (tmp_path / "hip.264").read_bytes()


def test_stock_welsenc_cfg_on_emulation(emu_lib, tmp_path):
    _stock_cfg(emu_lib, tmp_path)


@pytest.mark.gpu
def test_stock_welsenc_cfg_on_the_mi355x(hip_lib, tmp_path):
    _stock_cfg(hip_lib, tmp_path)


# ---- several encoder instances in one process: their pictures are launched together (FrameShared, csrc/host/frame_api.cpp) ------
def _parallel_sessions(lib, tmp_path, n):
    yuv = os.path.join(RES, "CiscoVT2people_320x192_12fps.yuv")
    base = [os.path.join(REF, "ref_enc_hip"), "-parallel", str(n), "-i", yuv, "-w", "320", "-h", "192", "-fps", "12", "-rc", "1", "-bitrate", "300000",
            "-slcmd", "1", "-slcnum", "3", "-bgd", "1", "-numtl", "2", "-quiet"]
    subprocess.check_call(base + ["-o", str(tmp_path / "c.264")], env=dict(os.environ, WELS_HIP="0"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run(base + ["-o", str(tmp_path / "d.264")], env=dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    assert err.count("welship hooks: installed") == n and err.count("welship hooks: did") >= 5 * n
    want = (tmp_path / "c.264.0").read_bytes()
    for k in range(n):
        assert (tmp_path / ("c.264.%d" % k)).read_bytes() == want
        assert (tmp_path / ("d.264.%d" % k)).read_bytes() == want, "session %d differs" % k


def test_concurrent_sessions_on_emulation(emu_lib, tmp_path):
    _parallel_sessions(emu_lib, tmp_path, 3)


@pytest.mark.gpu
def test_concurrent_sessions_on_the_mi355x(hip_lib, tmp_path):
    _parallel_sessions(hip_lib, tmp_path, 6)


# ---- pictures of more macroblocks than the record packer takes (advisor finding, round 4) ----------------------------------------------
def _large_picture_through_the_hooks(lib, tmp_path):
    """2048x1200 = 9600 macroblocks, above WELSHIP_PACKED_MAX_MB: the library hands such a picture's records over unpacked and says so
    (WelsHipFrameJob::pbRecordsPacked); the slice loop must read them as what they are -- the stream is the unmodified reference's."""
    from openh264_amd.utils.synth import synth_sequence
    w, h, n = 2048, 1200, 2
    yuv = str(tmp_path / "in.yuv")
    open(yuv, "wb").write(synth_sequence(w, h, n))
    base = ["-i", yuv, "-w", str(w), "-h", str(h), "-rc", "-1", "-qp", "30", "-fps", "30", "-iper", "0", "-slcmd", "1", "-slcnum", "2", "-quiet"]
    subprocess.check_call([os.path.join(REF, "ref_enc")] + base + ["-o", str(tmp_path / "c.264")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([os.path.join(REF, "ref_enc_hip")] + base + ["-o", str(tmp_path / "d.264")], env=dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "welship hooks: installed" in err and err.count("welship hooks: did") >= n, err[-2000:]
    assert (tmp_path / "c.264").read_bytes() == (tmp_path / "d.264").read_bytes()


def test_large_picture_through_the_hooks_on_emulation(emu_lib, tmp_path):
    _large_picture_through_the_hooks(emu_lib, tmp_path)


@pytest.mark.gpu
def test_large_picture_through_the_hooks_on_the_mi355x(hip_lib, tmp_path):
    _large_picture_through_the_hooks(hip_lib, tmp_path)


def test_unpacked_records_on_request_on_emulation(emu_lib, workdir):
    """WELSHIP_COMPACT=0 (the library keeps full records) under hooks that ask for packed ones: the answer's format is reported, not assumed."""
    row = _device_rows()[5]
    got, pictures, err = _run_row(workdir, emu_lib, row, "nc", {"WELSHIP_COMPACT": "0"})
    assert got == row[0] and pictures >= 40


# ---- I pictures with a GOM-level QP (quality-mode / timestamp rate control, one slice) in ONE device call; buffer-based rate control ------------
def _intra_gom_sessions(lib, tmp_path):
    """Rate-control modes that give I pictures a QP per group of macroblocks (RC_QUALITY_MODE 0, RC_TIMESTAMP_MODE 3; bitrate mode switches it off
    for I slices, ratectl.cpp:1199-1204): since round 5 the intra kernel runs the groups' recursion itself (k_intra_slice: the picture's own order,
    a group's first macroblock waits for the group before, its last one closes it -- kernels/frame_kernels.h, inter_mb.h wh_gom_close_if_last), so
    no picture of such a session goes group by group any more; the reference's own rate control still checks every macroblock's QP and bit count.
    RC_BUFFERBASED_MODE (2) has a frame-constant QP (WelsRcMbInitDisable): any number of slices, plain calls."""
    from openh264_amd.utils.synth import make_sequence
    cases = [(320, 192, 9, "pan7", 0, 3, []), (352, 288, 7, "checker8", 0, 0, ["-scene", "1"]), (640, 368, 6, "synth", 3, 4, []), (176, 144, 9, "checker5", 0, 5, ["-complexity", "2"]),
             (320, 192, 6, "synth", 2, 3, ["-slcmd", "1", "-slcnum", "3"]), (176, 144, 6, "pan7", 2, 0, [])]
    for k, (w, h, frames, content, rc, iper, extra) in enumerate(cases):
        yuv = str(tmp_path / "in.yuv")
        open(yuv, "wb").write(make_sequence(content, w, h, frames))
        flags = ["-i", yuv, "-w", str(w), "-h", str(h), "-rc", str(rc), "-bitrate", "600000", "-fps", "30", "-iper", str(iper), "-quiet"] + extra
        subprocess.check_call([os.path.join(REF, "ref_enc")] + flags + ["-o", str(tmp_path / "c.264")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        p = subprocess.run([os.path.join(REF, "ref_enc_hip")] + flags + ["-o", str(tmp_path / "d.264")], env=dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1"),
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        err = p.stderr.decode(errors="replace")
        assert p.returncode == 0 and "welship hooks: installed" in err, (k, err[-1500:])
        assert err.count(" I picture") >= (2 if iper else 1) and err.count("GOM-level QP") == 0, (k, err.count(" I picture"), err.count("GOM-level QP"))
        assert (tmp_path / "c.264").read_bytes() == (tmp_path / "d.264").read_bytes(), k


def test_intra_pictures_with_gom_level_qp_on_emulation(emu_lib, tmp_path):
    _intra_gom_sessions(emu_lib, tmp_path)


@pytest.mark.gpu
def test_intra_pictures_with_gom_level_qp_on_the_mi355x(hip_lib, tmp_path):
    _intra_gom_sessions(hip_lib, tmp_path)
