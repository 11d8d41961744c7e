"""CABAC sessions (Main / High profile) and slice threads through the dispatch-table binding.

Entropy coding stays on the host (north_star): the reference's own WelsSpatialWriteMbSynCabac codes the device's macroblock
records -- the hooks start the slice's arithmetic coder (WelsInitSliceCabac, svc_encode_slice.cpp:550-554,1824-1828) and hand over
what the writer's contexts read from the SMB array (types, cbp, the intra chroma mode of the neighbours; the vector
differences and iCbpDc it leaves there itself).  Mode decision does not depend on the entropy coder, so the device path is
the same as for CAVLC.

Checked here: the API golden hashes of test/api/encoder_test.cpp that need the SEncParamExt fixture of
test/api/BaseEncoderTest.cpp:25-71 (`ref_enc -ext`): CABAC (`d31a7239...`, :166-169), denoising (`913e49c7...`, :120-123) and
one slice per macroblock row on two slice threads (`266de2d0...`, :116-119) -- first against the unmodified reference (pins
oracle/_ref), then through the hooks; and a set of CABAC
configurations (constant QP and rate control, 1-4 slices, raster slices, temporal layers, LTR, background detection, complexity
0-2, screen content) byte for byte against the reference run live.

Slice threads (iMultipleThreadIdc > 1): the picture is coded on the device before the reference starts its slice tasks; every
task then entropy-codes its slice from the picture's records, and the per-slice in-loop filter the tasks would run
(wels_task_encoder.cpp:184) is the device's picture-wide pass in filter mode 2.  THREAD_CONFIGS: 2-4 threads over fixed and raster
slices, rate control, CABAC, screen content, and the cases where the reference ends up with one slice (no task runs: nothing is
filtered) -- against the reference really running its threads (load balancing off: it re-partitions by measured slice times).

CPU tier: the wave emulation (tests/emu); GPU tier (-m gpu): libwelship.so on the MI355X.
"""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
RES = os.path.join(REF, "res")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_enc_hip")), reason="oracle/_ref (hooked reference) not built")

CISCO = ("CiscoVT2people_320x192_12fps.yuv", 320, 192, 12)
EXT_GOLDEN = [  # flags after -ext, hash, must the hooks be installed
    (["-cabac", "1"], "d31a72395a4ca760c5b86a06901a2557e0373e76", True),
    (["-denoise", "1"], "913e49c787a0abdb378e9bc55bcffc27da89b965", True),
    (["-slcmd", "2"], "266de2d059a00ad2f28304e7eb378543ea7d85ab", True),       # one slice per macroblock row, two slice threads
]


def _enc(exe, lib, src, w, h, fps, flags, out):
    env = dict(os.environ, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1")
    if lib:
        env["WELSHIP_LIB"] = lib
    p = subprocess.run([os.path.join(REF, exe), "-i", src, "-w", str(w), "-h", str(h), "-fps", str(fps), "-o", out, "-quiet"] + flags,
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    return open(out, "rb").read(), err


@pytest.mark.parametrize("flags,sha,device", EXT_GOLDEN)
def test_oracle_reproduces_the_ext_fixture_hashes(tmp_path, flags, sha, device):
    bs, _ = _enc("ref_enc", None, os.path.join(RES, CISCO[0]), CISCO[1], CISCO[2], CISCO[3], ["-ext"] + flags, str(tmp_path / "r.264"))
    assert hashlib.sha1(bs).hexdigest() == sha


def _ext_through_hooks(lib, tmp_path, flags, sha, device):
    bs, err = _enc("ref_enc_hip", lib, os.path.join(RES, CISCO[0]), CISCO[1], CISCO[2], CISCO[3], ["-ext"] + flags, str(tmp_path / "h.264"))
    assert hashlib.sha1(bs).hexdigest() == sha
    if device:
        assert "welship hooks: installed" in err and err.count("welship hooks: did") >= 5
    else:
        assert "not installed" in err and "welship hooks: did" not in err


@pytest.mark.parametrize("flags,sha,device", EXT_GOLDEN)
def test_ext_fixture_hashes_through_the_hooks_on_emulation(emu_lib, tmp_path, flags, sha, device):
    _ext_through_hooks(emu_lib, tmp_path, flags, sha, device)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,sha,device", EXT_GOLDEN)
def test_ext_fixture_hashes_through_the_hooks_on_the_mi355x(hip_lib, tmp_path, flags, sha, device):
    _ext_through_hooks(hip_lib, tmp_path, flags, sha, device)


CABAC_CONFIGS = [  # clip ("cisco" / "ba" = BA_MW_D decoded, QCIF, 100 frames), flags
    ("cisco", "-cabac 1 -profile 77 -rc -1 -qp 30 -slcmd 1 -slcnum 3 -iper 16"),
    ("cisco", "-cabac 1 -profile 100 -rc 1 -bitrate 300000 -slcmd 1 -slcnum 4 -bgd 1 -numtl 2 -scene 1 -complexity 1"),
    ("ba", "-cabac 1 -profile 77 -rc -1 -qp 26 -complexity 2 -ltr 1 -numtl 3"),
    ("ba", "-cabac 1 -profile 77 -rc 0 -bitrate 200000 -bgd 1 -scene 1"),                 # one slice, rate control: GOM-level QP
    ("ba", "-cabac 1 -profile 77 -rc -1 -qp 12 -slcmd 2 -slcmbnum 33 -deblock 2"),
    ("cisco", "-cabac 1 -profile 77 -usage 1 -rc 1 -bitrate 400000 -slcmd 1 -slcnum 2"),   # screen content
]


THREAD_CONFIGS = [
    ("cisco", "-rc -1 -qp 28 -slcmd 1 -slcnum 4 -threads 4 -loadbalancing 0"),
    ("ba", "-rc 1 -bitrate 200000 -slcmd 1 -slcnum 3 -threads 2 -loadbalancing 0 -bgd 1 -numtl 2 -scene 1"),
    ("ba", "-rc -1 -qp 26 -slcmd 2 -slcmbnum 22 -threads 3 -complexity 1 -deblock 1"),
    ("ba", "-rc -1 -qp 30 -slcmd 1 -slcnum 4 -threads 4 -loadbalancing 0 -deblock 0 -cabac 1 -profile 77 -numtl 3"),
    ("cisco", "-usage 1 -rc 1 -bitrate 400000 -slcmd 1 -slcnum 2 -threads 2 -loadbalancing 0"),
    ("ba", "-rc -1 -qp 26 -slcmd 2 -slcmbnum 99 -threads 4"),            # raster slices that end up as ONE slice: no task, no filtering
    ("ba", "-rc 0 -bitrate 150000 -slcmd 0 -threads 4"),                 # single-slice mode (GOM-level QP) with a thread count
]


@pytest.fixture(scope="module")
def ba_yuv(tmp_path_factory, ref_tools):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    p = tmp_path_factory.mktemp("cabac") / "ba.yuv"
    subprocess.check_call([ref_tools["dec"], os.path.join(RES, "BA_MW_D.264"), str(p)], stdout=subprocess.DEVNULL)
    return str(p)


def _cabac_config(lib, tmp_path, ba_yuv, clip, flags):
    src, w, h, fps = (os.path.join(RES, CISCO[0]), CISCO[1], CISCO[2], CISCO[3]) if clip == "cisco" else (ba_yuv, 176, 144, 30)
    want, _ = _enc("ref_enc", None, src, w, h, fps, flags.split(), str(tmp_path / "r.264"))
    got, err = _enc("ref_enc_hip", lib, src, w, h, fps, flags.split(), str(tmp_path / "h.264"))
    assert "welship hooks: installed" in err and err.count("welship hooks: did") >= 5, err[-1000:]
    assert got == want


@pytest.mark.parametrize("clip,flags", CABAC_CONFIGS)
def test_cabac_sessions_on_emulation(emu_lib, tmp_path, ba_yuv, clip, flags):
    _cabac_config(emu_lib, tmp_path, ba_yuv, clip, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("clip,flags", CABAC_CONFIGS)
def test_cabac_sessions_on_the_mi355x(hip_lib, tmp_path, ba_yuv, clip, flags):
    _cabac_config(hip_lib, tmp_path, ba_yuv, clip, flags)


@pytest.mark.parametrize("clip,flags", THREAD_CONFIGS)
def test_slice_threads_on_emulation(emu_lib, tmp_path, ba_yuv, clip, flags):
    _cabac_config(emu_lib, tmp_path, ba_yuv, clip, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("clip,flags", THREAD_CONFIGS)
def test_slice_threads_on_the_mi355x(hip_lib, tmp_path, ba_yuv, clip, flags):
    _cabac_config(hip_lib, tmp_path, ba_yuv, clip, flags)
