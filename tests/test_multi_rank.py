"""The N>1 path on CPU: world_size-2 `gloo` processes shard independent sessions with no data-path
collective and must reproduce the single-process result (kernels via the CPU wave-emulation build)."""
import hashlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FRAMES, SESSIONS = 96, 80, 3, 5


def _inputs():
    from openh264_amd.utils.synth import synth_sequence
    fsz = W * H * 3 // 2
    out = []
    for s in range(SESSIONS):
        yuv = synth_sequence(W, H, FRAMES, seed=0x1234 + 17 * s)
        out.append([yuv[f * fsz:(f + 1) * fsz] for f in range(FRAMES)])
    return out


def _make_group_factory(lib):
    import openh264_amd as oh

    def make(n):
        e = oh.Encoder(lib)
        p = e.GetDefaultParams()
        e.close()
        p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = W, H, 26, 0, 30.0, 500000
        return oh.EncoderGroup(p, n, ring_slots=FRAMES, host_threads=2, lib_path=lib)
    return make


def _worker(rank, world, port, lib, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from openh264_amd.parallel import encode_sessions_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dist.barrier()
        d = encode_sessions_sharded(_make_group_factory(lib), _inputs(), FRAMES, rank, world, dist)
        dist.barrier()
        if rank == 0:
            q.put(d)
    finally:
        dist.destroy_process_group()


def test_shard_range():
    from openh264_amd.parallel import shard_range
    for n in (0, 1, 5, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == n
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_ranks_match_single_process(emu_lib):
    import torch.multiprocessing as mp
    from openh264_amd.parallel import encode_sessions_sharded
    single = encode_sessions_sharded(_make_group_factory(emu_lib), _inputs(), FRAMES)
    assert len(single) == SESSIONS and len(set(single)) == SESSIONS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_lib, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == single


def _groups_in_threads(lib):
    """Two session groups driven by two host threads at the same time (bench.py's overlapped end-to-end leg): each group has
    its own backend / device queue, nothing is shared -- the streams must be those of the groups run one after the other."""
    import threading
    from openh264_amd.parallel import encode_sessions_sharded
    inputs = _inputs()
    want = encode_sessions_sharded(_make_group_factory(lib), inputs, FRAMES)
    make = _make_group_factory(lib)
    parts = [inputs[:2], inputs[2:]]
    got = [None, None]

    def run(k):
        g = make(len(parts[k]))
        pics = [g.make_pictures([parts[k][s][f] for s in range(len(parts[k]))]) for f in range(FRAMES)]
        streams = [bytearray() for _ in parts[k]]
        for f in range(FRAMES):
            for s, bs in enumerate(g.encode_frames(pics[f], want_bytes=True)):
                streams[s] += bs
        got[k] = [hashlib.sha1(bytes(b)).hexdigest() for b in streams]
        g.close()

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert got[0] + got[1] == want


def test_groups_in_threads(emu_lib):
    _groups_in_threads(emu_lib)


@pytest.mark.gpu
def test_hip_groups_in_threads(hip_lib):
    _groups_in_threads(hip_lib)


def test_group_equals_single_session(emu_lib):
    """A session inside a group produces the same bitstream as the ISVCEncoder-style object."""
    import openh264_amd as oh
    inputs = _inputs()
    digs = __import__("openh264_amd.parallel", fromlist=["x"]).encode_sessions_sharded(_make_group_factory(emu_lib), inputs, FRAMES)
    for s in (0, SESSIONS - 1):
        bs, _ = oh.encode_sequence(b"".join(inputs[s]), W, H, lib_path=emu_lib, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=500000)
        assert hashlib.sha1(bs).hexdigest() == digs[s]


@pytest.mark.gpu
def test_hip_group_equals_single_session(hip_lib, ref_tools, tmp_path):
    """The same on the MI355X: the group path copies the macroblock records back packed (k_compact, common/compact.h), the
    single session copies them whole -- both must give the stream of the ISVCEncoder-style object, and that stream is oracle/_ref's."""
    import openh264_amd as oh
    inputs = _inputs()
    digs = __import__("openh264_amd.parallel", fromlist=["x"]).encode_sessions_sharded(_make_group_factory(hip_lib), inputs, FRAMES)
    for s in range(SESSIONS):
        bs, _ = oh.encode_sequence(b"".join(inputs[s]), W, H, lib_path=hip_lib, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=500000)
        assert hashlib.sha1(bs).hexdigest() == digs[s]
        if ref_tools:
            assert _reference_stream(ref_tools, tmp_path, b"".join(inputs[s]), W, H, ["-qp", 26, "-iper", 0, "-bitrate", 500000]) == bs


def test_group_reencodes_overflowing_sessions(emu_lib):
    """Sessions of a group whose picture hits a CAVLC overflow are re-encoded individually (WelsHipGroupFinish) and
    still match the single-session result; their neighbours in the group are unaffected."""
    import openh264_amd as oh
    from openh264_amd.utils.synth import make_sequence
    w, h, frames, qp = 64, 64, 3, 3
    fsz = w * h * 3 // 2
    seqs = [make_sequence(c, w, h, frames) for c in ("synth", "checker5", "synth", "checker8")]
    e = oh.Encoder(emu_lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = w, h, qp, 0, 30.0, 5000000
    g = oh.EncoderGroup(p, len(seqs), ring_slots=2, host_threads=2, lib_path=emu_lib)
    got = [bytearray() for _ in seqs]
    for f in range(frames):
        for s, yuv in enumerate(seqs):
            g.upload(s, f & 1, yuv[f * fsz:(f + 1) * fsz])
        for s, bs in enumerate(g.step(f & 1)):
            got[s] += bs
    # LOW complexity P pictures need the previous source picture: re-using its slot is refused, not silently different
    g.upload(0, (frames - 1) & 1, seqs[0][:fsz])
    with pytest.raises(oh.WelsHipError):
        g.step((frames - 1) & 1)
    g.close()
    for s, yuv in enumerate(seqs):
        st = {}
        bs, _ = oh.encode_sequence(yuv, w, h, lib_path=emu_lib, stats=st, iDLayerQp=qp, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000)
        assert bytes(got[s]) == bs
        assert (st["overflow_reencodes"] > 0) == (s in (1, 3))


def test_group_bench_path_on_emulation(emu_lib):
    """The device-only benchmark entry point (bench.py's timed region) with resident sources: repeated calls, then a full
    step -- the same call pattern as bench.py, here on the CPU emulation so that it cannot break unnoticed."""
    import openh264_amd as oh
    from openh264_amd.utils.synth import synth_sequence
    w, h, ring = 64, 48, 4
    fsz = w * h * 3 // 2
    yuv = synth_sequence(w, h, ring)
    e = oh.Encoder(emu_lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = w, h, 24, 0, 30.0, 5000000
    g = oh.EncoderGroup(p, 3, ring_slots=ring, host_threads=1, lib_path=emu_lib)
    for s in range(3):
        for k in range(ring):
            g.upload(s, k, yuv[k * fsz:(k + 1) * fsz])
    g.bench(1, 0)
    g.bench(2, 1)
    ev = g.bench(3, 0)
    assert ev["total_ms"] >= 0.0
    for i in range(3):
        assert all(len(bs) > 0 for bs in g.step(i % ring))
    st = g.host_stats()                  # the host share of the three complete steps: entropy coding straight from the packed records
    assert st["pictures"] == 9 and st["entropy_ms_per_picture"] > 0.0
    assert 0 < st["packed_record_bytes_per_picture"] < 964 * (w // 16) * (h // 16)
    g.close()


def test_group_with_mixed_frame_types(emu_lib):
    """Scene changes (and forced IDRs) make sessions of a group disagree on the frame type: the P and IDR pictures of a
    step then go to their own launches, and every session still matches its single-session stream."""
    import openh264_amd as oh
    from openh264_amd.utils.synth import synth_sequence
    w, h, frames = 64, 48, 20
    fsz = w * h * 3 // 2
    still = synth_sequence(w, h, 1) * frames                      # no motion: never a scene change
    seqs = [synth_sequence(w, h, frames), still, synth_sequence(w, h, frames, seed=77)]
    e = oh.Encoder(emu_lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = w, h, 30, 0, 30.0, 5000000
    p.bEnableSceneChangeDetect = 1
    g = oh.EncoderGroup(p, len(seqs), ring_slots=2, host_threads=2, lib_path=emu_lib)
    got = [bytearray() for _ in seqs]
    for f in range(frames):
        for s, yuv in enumerate(seqs):
            g.upload(s, f & 1, yuv[f * fsz:(f + 1) * fsz])
        for s, bs in enumerate(g.step(f & 1)):
            got[s] += bs
    g.close()
    idrs = []
    for s, yuv in enumerate(seqs):
        bs, _ = oh.encode_sequence(yuv, w, h, lib_path=emu_lib, iDLayerQp=30, uiIntraPeriod=0, fMaxFrameRate=30.0,
                                   iTargetBitrate=5000000, bEnableSceneChangeDetect=1)
        assert bytes(got[s]) == bs
        idrs.append(bs.count(b"\x00\x00\x00\x01\x65"))
    assert idrs[1] == 1 and max(idrs) == 2        # the step with the scene change really was mixed


@pytest.mark.parametrize("world", [2, 8])
def test_bench_line_from_two_ranks_on_the_cpu_test_build(emu_lib, world):
    """bench.py --gpus 2 (and --gpus 8: the node the driver's scaling run uses) end to end without a GPU: the launcher (torch.distributed.run, one process per rank), the barriers, the
    max-over-ranks time and the ONE JSON line of rank 0 with its roofline object -- gloo in place of RCCL and the CPU test build of the
    kernels in place of libwelship.so (bench.py --cpu-launcher-test; the line says it is no measurement).  What the driver's 2 / 4 / 8
    GPU runs execute, minus the device."""
    import json
    import subprocess
    env = dict(os.environ, WELSHIP_LIB=emu_lib)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--sessions", "3", "--width", "176",
                        "--height", "144", "--cpu-launcher-test", "--multi-gpu-legs"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "frames/s"
    assert d["value"] > 0 and abs(d["value"] - world * 3 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6        # all ranks' pictures / the slowest rank's time
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0 and "not a measurement" in d["data"]
    # BASELINE configs 5 and 4 over the ranks (bench.py multi_gpu_legs): every rank hosts its sessions through the dispatch-table binding
    # at once, rank 0 gathers the digests; one simulcast session with its layers spread over the devices (WELS_HIP_LAYER_DEVICES = number of ranks).
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_enc_hip")):
        c5 = d["config5_64_sessions" if world == 8 else "config5_4_sessions"]      # (2 sessions per rank in the launcher test; 8 per GPU on an 8-GPU node: the key is the real run's)
        assert c5["n_ranks_seen"] == world and c5["sessions"] == 2 * world and c5["same_bitstreams"] is True and not c5["errors"]
        assert c5["c_path_comparable_with_single_gpu_leg"] is False
        assert len(c5["per_gpu_device_frames_per_s"]) == world and all(v > 0 for v in c5["per_gpu_device_frames_per_s"])
        assert abs(c5["aggregate_device_frames_per_s"] - sum(c5["per_gpu_device_frames_per_s"])) < 1e-6 and all(v > 0 for v in c5["per_session_latency_ms"])
        c4 = d["config4_layer_per_gpu"]
        n4 = min(world, 4)
        assert c4["same_bitstreams"] is True and c4["layer_devices"] == n4 and c4["devices_seen"] == list(range(n4)) and c4["device_frames_per_s"] > 0


# ---- pipelined groups: WelsHipGroupEncodeFramesPipelined returns step k - 1's streams while the device codes step k -------------
def _pipelined_vs_synchronous(lib, w, h, frames, qp, contents, ring, intra_period=0, threads=2, ahead=1, complexity=0):
    import openh264_amd as oh
    from openh264_amd.utils.synth import make_sequence
    fsz = w * h * 3 // 2
    seqs = [make_sequence(c, w, h, frames) for c in contents]
    e = oh.Encoder(lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.uiIntraPeriod, p.fMaxFrameRate, p.iTargetBitrate = w, h, qp, intra_period, 30.0, 5000000
    p.bEnableSceneChangeDetect = False
    p.iComplexityMode = complexity
    out = {}
    for mode in ("sync", "pipe"):
        g = oh.EncoderGroup(p, len(seqs), ring_slots=ring, host_threads=threads, lib_path=lib)
        if mode == "pipe":
            g.set_pipelined(ahead)
        got = [bytearray() for _ in seqs]
        steps = 0
        for f in range(frames):
            pics = g.make_pictures([s[f * fsz:(f + 1) * fsz] for s in seqs])
            res = g.encode_frames(pics, want_bytes=True) if mode == "sync" else g.encode_frames_pipelined(pics, want_bytes=True)
            if mode == "pipe":
                assert (res is None) == (f < ahead)       # `ahead` calls late
            if res is not None:
                steps += 1
                for s, bs in enumerate(res):
                    got[s] += bs
        while mode == "pipe":
            res = g.encode_frames_pipelined(None, want_bytes=True)
            if res is None:                                   # nothing pending any more
                break
            steps += 1
            for s, bs in enumerate(res):
                got[s] += bs
        assert steps == frames
        out[mode] = ([bytes(b) for b in got], g.recon(0))
        g.close()
    if out["pipe"][0] != out["sync"][0] or out["pipe"][1] != out["sync"][1]:
        _explain_mismatch(lib, seqs, out, w, h, qp, intra_period, complexity, ahead)
    assert out["pipe"][0] == out["sync"][0]
    assert out["pipe"][1] == out["sync"][1]
    return seqs, out["pipe"][0]


def _explain_mismatch(lib, seqs, out, w, h, qp, intra_period, complexity, ahead):
    """A pipelined group and a synchronous one disagree: say which of them left the single-session encoder's stream (same library), for which
    session and from which access unit on, and keep the streams (gpurun_out/ travels back from the GPU box)."""
    import openh264_amd as oh
    d = os.path.join(ROOT, "gpurun_out", "pipelined_mismatch_ahead%d" % ahead)
    os.makedirs(d, exist_ok=True)
    lines = []
    for s, yuv in enumerate(seqs):
        want, _ = oh.encode_sequence(yuv, w, h, lib_path=lib, iDLayerQp=qp, uiIntraPeriod=intra_period, fMaxFrameRate=30.0, iTargetBitrate=5000000,
                                     bEnableSceneChangeDetect=False, iComplexityMode=complexity)
        for mode in ("sync", "pipe"):
            got = out[mode][0][s]
            open(os.path.join(d, "%s_session%d.264" % (mode, s)), "wb").write(got)
            if got != want:
                first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), min(len(got), len(want)))
                lines.append("%s session %d: leaves the single-session stream at byte %d of %d (access unit %d)" % (mode, s, first, len(want), want[:first].count(b"\x00\x00\x00\x01")))
        open(os.path.join(d, "single_session%d.264" % s), "wb").write(want)
    lines.append("reconstruction of session 0: %s" % ("same" if out["pipe"][1] == out["sync"][1] else "differs"))
    open(os.path.join(d, "report.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


@pytest.mark.parametrize("ring,ahead", [(2, 1), (3, 1), (3, 2), (2, 3)])
def test_pipelined_group_matches_the_synchronous_one(emu_lib, ring, ahead):
    _pipelined_vs_synchronous(emu_lib, 64, 48, 7, 26, ("synth", "checker5", "synth"), ring, intra_period=4, ahead=ahead)


@pytest.mark.parametrize("ring,ahead,frames", [(3, 1, 4), (3, 2, 4), (2, 2, 7), (2, 3, 8), (3, 3, 8), (5, 2, 6)])
def test_pipelined_group_reencodes_after_cavlc_overflow(emu_lib, ring, ahead, frames):
    """QP 3 on checkerboards: pictures in the middle and at the end of the stream overflow the CAVLC level range, when their successors
    are already on the device: the picture is coded again, then the successors (they predicted from the replaced reconstruction).
    The repeat reads the picture's source slot and the previous picture's (LOW complexity) while `ahead` later pictures have been tiled
    into the ring: a ring asked for with fewer than ahead + 2 slots is grown by WelsHipGroupSetPipelined (round-3 advisor finding:
    ring 2 / ahead 2 and ring 3 / ahead 3 coded the wrong source in the repeat)."""
    import openh264_amd as oh
    seqs, got = _pipelined_vs_synchronous(emu_lib, 64, 64, frames, 3, ("synth", "checker5", "synth", "checker8"), ring, ahead=ahead)
    for s in (1, 3):
        st = {}
        bs, _ = oh.encode_sequence(seqs[s], 64, 64, lib_path=emu_lib, stats=st, iDLayerQp=3, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=5000000)
        assert st["overflow_reencodes"] > 0 and bs == got[s]


def test_pipelined_group_refuses_what_it_cannot_do(emu_lib):
    import openh264_amd as oh
    e = oh.Encoder(emu_lib)
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate = 64, 48, 26, 30.0, 500000
    p.bEnableSceneChangeDetect = True
    g = oh.EncoderGroup(p, 2, ring_slots=3, host_threads=1, lib_path=emu_lib)
    with pytest.raises(oh.WelsHipError):
        g.set_pipelined()
    g.close()
    # ... and a pipelined group takes its frame steps through the pipelined call only (its records live in per-step buffer sets)
    from openh264_amd.utils.synth import synth_sequence
    p.bEnableSceneChangeDetect = False
    g = oh.EncoderGroup(p, 2, ring_slots=3, host_threads=1, lib_path=emu_lib)
    g.set_pipelined(2)
    yuv = synth_sequence(64, 48, 1)
    pics = g.make_pictures([yuv, yuv])
    with pytest.raises(oh.WelsHipError):
        g.encode_frames(pics)
    assert g.encode_frames_pipelined(pics) is None
    with pytest.raises(oh.WelsHipError):
        g.set_pipelined(3)                       # (a second call is only accepted with the same number of steps ahead)
    g.set_pipelined(2)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", [1, 2])
def test_hip_pipelined_group_matches_the_synchronous_one(hip_lib, ahead):
    _pipelined_vs_synchronous(hip_lib, 320, 192, 8, 26, ("synth", "checker5", "synth", "pan7"), 3, intra_period=5, threads=4, ahead=ahead)


@pytest.mark.gpu
@pytest.mark.parametrize("ahead", [1, 2])
def test_hip_pipelined_group_reencodes_after_cavlc_overflow(hip_lib, ahead):
    _pipelined_vs_synchronous(hip_lib, 64, 64, 4, 3, ("synth", "checker5", "synth", "checker8"), 3, ahead=ahead)


def test_plain_p_kernel_variant_on_emulation(emu_lib):
    """A session group's steps promise the P kernel that their pictures carry no optional per-picture inputs (WH_SEQ_PLAIN, common/wh_types.h)
    and get a body variant that never looks at them (hip_backend.hip WH_PLAIN_KERNEL; the emulation also checks the promise on every picture).
    The default test build takes that variant, a build with -DWH_PLAIN_KERNEL=0 the general body: both must give the streams of the
    single-session encoder (which never takes the variant), synchronous and pipelined."""
    import openh264_amd as oh
    from openh264_amd import build as B
    from openh264_amd.parallel import encode_sessions_sharded
    general = B.build_emu(defines=("WH_PLAIN_KERNEL=0",), tag="noplain")
    inputs = _inputs()
    want = []
    for s in range(SESSIONS):
        bs, _ = oh.encode_sequence(b"".join(inputs[s]), W, H, lib_path=emu_lib, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0, iTargetBitrate=500000)
        want.append(hashlib.sha1(bs).hexdigest())
    for lib in (emu_lib, general):
        assert encode_sessions_sharded(_make_group_factory(lib), inputs, FRAMES) == want
        seqs, got = _pipelined_vs_synchronous(lib, 64, 64, 5, 24, ("synth", "checker5", "pan7"), 3, intra_period=4, ahead=2)
        for s, yuv in enumerate(seqs):
            bs, _ = oh.encode_sequence(yuv, 64, 64, lib_path=emu_lib, iDLayerQp=24, uiIntraPeriod=4, fMaxFrameRate=30.0, iTargetBitrate=5000000, bEnableSceneChangeDetect=False)
            assert bs == got[s]
        # (LOW complexity takes the variant that knows it at compile time, WH_PLAIN_KERNEL == 2; any other complexity the plain variant with the SATD paths)
        seqs, got = _pipelined_vs_synchronous(lib, 64, 64, 5, 24, ("synth", "checker5", "pan7"), 3, intra_period=4, ahead=2, complexity=2)
        for s, yuv in enumerate(seqs):
            bs, _ = oh.encode_sequence(yuv, 64, 64, lib_path=emu_lib, iDLayerQp=24, uiIntraPeriod=4, fMaxFrameRate=30.0, iTargetBitrate=5000000, bEnableSceneChangeDetect=False, iComplexityMode=2)
            assert bs == got[s]


def _reference_stream(ref_tools, tmp_path, yuv, w, h, flags):
    """The stream of oracle/_ref/ref_enc (the reference compiled by oracle/Makefile) for one input, run live."""
    import subprocess
    fi, fo = str(tmp_path / "in.yuv"), str(tmp_path / "ref.264")
    open(fi, "wb").write(yuv)
    subprocess.check_call([ref_tools["enc"], "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-fps", "30", "-quiet"] + [str(f) for f in flags],
                          stdout=subprocess.DEVNULL)
    return open(fo, "rb").read()


def _group_paths_against_the_reference(lib, ref_tools, tmp_path, w, h, frames):
    """Session groups (the PLAIN body variants of the P kernel, packed records, the pipelined steps) against the reference itself: every
    stream a group returns -- synchronous, pipelined, LOW and MEDIUM complexity -- must be the one oracle/_ref/ref_enc writes for that input."""
    from openh264_amd.parallel import encode_sessions_sharded
    inputs = _inputs()
    digs = encode_sessions_sharded(_make_group_factory(lib), inputs, FRAMES)
    for s in range(SESSIONS):
        ref = _reference_stream(ref_tools, tmp_path, b"".join(inputs[s]), W, H, ["-qp", 26, "-iper", 0, "-bitrate", 500000])
        assert hashlib.sha1(ref).hexdigest() == digs[s], "group session %d differs from oracle/_ref" % s
    for complexity in (0, 2):
        seqs, got = _pipelined_vs_synchronous(lib, w, h, frames, 24, ("synth", "checker5", "synth", "pan7"), 3, intra_period=5, threads=4, ahead=2, complexity=complexity)
        for s, yuv in enumerate(seqs):
            ref = _reference_stream(ref_tools, tmp_path, yuv, w, h, ["-qp", 24, "-iper", 5, "-scene", 0, "-complexity", complexity])
            assert ref == got[s], "pipelined group session %d (complexity %d) differs from oracle/_ref" % (s, complexity)


def test_group_paths_against_the_reference_on_emulation(emu_lib, ref_tools, tmp_path):
    if not ref_tools:
        pytest.skip("oracle/_ref not built")
    _group_paths_against_the_reference(emu_lib, ref_tools, tmp_path, 96, 64, 6)


@pytest.mark.gpu
def test_hip_plain_p_kernel_variant(hip_lib, ref_tools, tmp_path):
    """The same on the MI355X for the shipped library, in process: the kernel the headline is timed on (k_inter_pool, PLAIN variant, which only
    session groups launch) meets the oracle here, not only inside bench.py."""
    assert ref_tools, "oracle/_ref must travel to the GPU box (it is the checker)"
    _group_paths_against_the_reference(hip_lib, ref_tools, tmp_path, 320, 192, 8)
