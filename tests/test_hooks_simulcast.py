"""BASELINE config 4 through the dispatch-table binding: a 4-layer spatial simulcast (1920x1080 / 1280x720 / 640x360 / 320x180) from
one 1080p input -- the reference's own frame layer downsamples (codec/processing, C downsampler) and codes the four layers one
after the other; every layer's macroblock work runs on the device in its own context (optionally its own GPU:
WELS_HIP_LAYER_DEVICES=1), and the highest layer takes the inter-layer mode-decision hints of WelsMdInterMbEnhancelayer
(svc_mode_decision.cpp:108-150) from the layer below.  The access units must equal the unpatched reference's byte for byte.

Also covered here: 2- and 3-layer sessions with rate control, several slices, temporal layers and background detection."""
import os
import subprocess

import pytest

from openh264_amd.utils.synth import synth_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
RES = os.path.join(REF, "res")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ref_enc_hip")), reason="oracle/_ref (hooked reference) not built")

CONFIG4 = ["-simulcast", "320", "180", "-simulcast", "640", "360", "-simulcast", "1280", "720"]


def _both(lib, tmp_path, yuv, w, h, flags, min_pictures, extra_env=None):
    fi = str(tmp_path / "in.yuv")
    open(fi, "wb").write(yuv)
    base = ["-i", fi, "-w", str(w), "-h", str(h), "-fps", "30", "-quiet"] + flags
    subprocess.check_call([os.path.join(REF, "ref_enc"), "-o", str(tmp_path / "ref.264")] + base, stdout=subprocess.DEVNULL)
    # (every layer on the device unless a test asks for the binding's default -- layers below 1000 macroblocks of a multi-layer session stay on
    #  the host, WELS_HIP_MIN_LAYER_MBS -- or for a threshold of its own: the small pictures of this file would all stay there)
    env = dict(os.environ, WELSHIP_LIB=lib, WELS_HIP_TRACE="1", WELS_HIP_CHECK_BITS="1", WELS_HIP_MIN_LAYER_MBS="0")
    for k, v in (extra_env or {}).items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    p = subprocess.run([os.path.join(REF, "ref_enc_hip"), "-o", str(tmp_path / "hip.264")] + base, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0, err[-2000:]
    assert "welship hooks: installed" in err and err.count("welship hooks: did") >= min_pictures, err[-2000:]
    # the lower layers' source pictures come from the device's down-sampling cascade (pfHipDownsample in
    # CWelsPreProcess::DownsamplePadding), not from the reference's C functions -- unless switched off
    lower = flags.count("-simulcast")
    if env.get("WELS_HIP_DOWNSAMPLE") == "0":
        assert "down-sampled" not in err
    else:
        assert err.count("down-sampled") >= lower * (min_pictures // (lower + 1)), err[-1500:]
    assert (tmp_path / "ref.264").read_bytes() == (tmp_path / "hip.264").read_bytes()
    return err


SMALL = [
    (["-rc", "1", "-bitrate", "800000", "-simulcast", "160", "96", "-simulcast", "320", "192", "-slcmd", "1", "-slcnum", "2"], 18),
    (["-rc", "0", "-bitrate", "800000", "-simulcast", "106", "62", "-simulcast", "212", "122", "-simulcast", "426", "246", "-bgd", "1", "-numtl", "2"], 24),
    (["-rc", "3", "-bitrate", "500000", "-simulcast", "320", "184", "-complexity", "2", "-scene", "1"], 12),
    (["-rc", "-1", "-qp", "28", "-simulcast", "160", "96", "-simulcast", "320", "184", "-complexity", "1", "-slcmd", "1", "-slcnum", "3"], 18),
]


@pytest.mark.parametrize("flags,pictures", SMALL)
def test_simulcast_sessions_on_emulation(emu_lib, tmp_path, flags, pictures):
    _both(emu_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures)


@pytest.mark.parametrize("flags,pictures", SMALL[2:])
def test_simulcast_sessions_on_the_reverse_lane_emulation(tmp_path, flags, pictures):
    """The complexity > 0 sessions once more on the test build that walks the lanes of a lane block backwards
    (tests/test_frame_parity.py::test_emu_reverse_lane_order): the stale-pSadCost path they exercise is where two lanes once
    stored to one LDS word, which only the MI355X noticed."""
    from openh264_amd import build as B
    _both(B.build_emu(defines=("WH_EMU_REVERSE",), tag="wh_emu_reverse"), tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures)


def test_config4_four_layers_on_emulation(emu_lib, tmp_path):
    _both(emu_lib, tmp_path, synth_sequence(1920, 1080, 3), 1920, 1080, ["-rc", "-1", "-qp", "24"] + CONFIG4, 12)


@pytest.mark.parametrize("flags,pictures", SMALL)
def test_small_layers_left_to_the_host_on_emulation(emu_lib, tmp_path, flags, pictures):
    """pfHipLayerOnDevice (round 6): the layers below the threshold are coded by the reference's own path -- slice loops, in-loop filter, reference
    list as if no hook were installed -- and only the full-size layer goes to the device, which still takes its inter-layer hints from
    the (host-coded) layer below and shares the one pSadCost array with it.  Same access units."""
    layers = flags.count("-simulcast") + 1
    err = _both(emu_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures // layers, {"WELS_HIP_MIN_LAYER_MBS": "500"})
    assert err.count("left to the host") == pictures - pictures // layers and err.count("welship hooks: did") == pictures // layers, err[-1500:]


def test_config4_with_the_default_layer_split_on_emulation(emu_lib, tmp_path):
    """The binding's default for config 4: 1080p and 720p on the device, 360p and 180p (920 and 240 macroblocks) on the host."""
    err = _both(emu_lib, tmp_path, synth_sequence(1920, 1080, 3), 1920, 1080, ["-rc", "-1", "-qp", "24"] + CONFIG4, 6, {"WELS_HIP_MIN_LAYER_MBS": None})
    assert err.count("left to the host") == 6 and err.count("welship hooks: did") == 6 and "pictures of layers below 1000 macroblocks were coded by the host" in err


def test_simulcast_with_the_host_downsampler(emu_lib, tmp_path):
    """WELS_HIP_DOWNSAMPLE=0: the reference's own C down-samplers feed the layers (the pre-round-3 arrangement); same bytes."""
    _both(emu_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, SMALL[1][0], SMALL[1][1], {"WELS_HIP_DOWNSAMPLE": "0"})


@pytest.mark.gpu
@pytest.mark.parametrize("flags,pictures", SMALL)
def test_simulcast_sessions_on_the_mi355x(hip_lib, tmp_path, flags, pictures):
    _both(hip_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("rc", [["-rc", "-1", "-qp", "24"], ["-rc", "1", "-bitrate", "3000000", "-slcmd", "1", "-slcnum", "4"]])
def test_config4_four_layers_of_the_1080p_clip_on_the_mi355x(hip_lib, ref_tools, tmp_path, rc, split):
    """split: the binding's default -- 360p and 180p on the host, 1080p and 720p on the device; else all four layers on the device."""
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    out = str(tmp_path / "clip.yuv")
    subprocess.check_call([ref_tools["dec"], os.path.join(RES, "VID_1920x1080_cavlc_temporal_direct.264"), out], stdout=subprocess.DEVNULL)
    yuv = open(out, "rb").read()[: 1920 * 1080 * 3 // 2 * 12]
    err = _both(hip_lib, tmp_path, yuv, 1920, 1080, rc + CONFIG4, 24 if split else 48, {"WELS_HIP_MIN_LAYER_MBS": None} if split else None)
    assert err.count("left to the host") == (24 if split else 0)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,pictures", SMALL[:2])
def test_small_layers_left_to_the_host_on_the_mi355x(hip_lib, tmp_path, flags, pictures):
    layers = flags.count("-simulcast") + 1
    err = _both(hip_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures // layers, {"WELS_HIP_MIN_LAYER_MBS": "500"})
    assert err.count("left to the host") == pictures - pictures // layers


def test_one_device_per_simulcast_layer(emu_lib, tmp_path):
    """WELS_HIP_LAYER_DEVICES=1: spatial layer d gets a device context on GPU WELS_HIP_DEVICE + d, a backend of its own each (no GPU in
    the CPU tier: the test build of the kernels reports which device index every backend was asked for); without the switch all layers
    share device WELS_HIP_DEVICE.  Same access units either way."""
    import re
    flags, pictures = SMALL[1]                         # three lower layers + the full-size one
    for layer_devices, want in (("1", [2, 3, 4, 5]), ("0", [2]), ("2", [2, 3])):      # ("2": two GPUs for the four layers, d mod 2)
        err = _both(emu_lib, tmp_path, synth_sequence(640, 368, 6), 640, 368, flags, pictures,
                    {"WELS_HIP_DEVICE": "2", "WELS_HIP_LAYER_DEVICES": layer_devices, "WELSHIP_TRACE_DEVICES": "1"})
        got = sorted(set(int(x) for x in re.findall(r"welship emu: backend for device (\d+)", err)))
        # (the picture-level down-sampling entry point keeps its own per-device state and is not a backend)
        assert got == want, (layer_devices, got)
