#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X macroblock engine on BASELINE.json's metric ("Measurement").

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (mode decision + reconstruction, in-loop deblocking, border expansion) over one
batch: `--sessions` independent 1080p pictures per GPU (default 128 = two slice workgroups per CU), every session with its
own content, sources already resident in HBM, MB records left in HBM.  value = pictures processed by all ranks / time.

With --gpus N > 1 and no torchrun environment the script launches the N ranks itself (one process per GPU, sessions
sharded over ranks, no data-path collective: "weak" scaling); under torchrun it is one of the ranks.

The JSON line (last line of stdout, rank 0) carries, besides the contract's keys:
  roofline       dominant kernel, algorithmic bytes per launch / HIP-event launch time (events on the launch stream)
  verified       the reconstructions the timed steps left behind (eight sessions spread over 0 .. N-1) equal the reference decoder's output of
                 the reference encoder's stream for the same frame order (oracle/_ref, run live) -- the run fails otherwise
  e2e_pipelined  complete EncodeFrame rate incl. source upload, D2H of the packed MB records and host CAVLC (the PCIe/host-inclusive
                 figure; never `value`), session 0's bitstream compared with the reference encoder's
  config4_* / config5_*   BASELINE configs 4 and 5 through the dispatch-table binding next to the reference's C path; with --gpus N > 1:
                 config5_64_sessions (8 sessions per GPU on every rank) and config4_layer_per_gpu (one simulcast session, its layers spread over the GPUs)
  latency        per-frame wall time of complete EncodeFrame calls with 1 and 8 sessions on the GPU
  res_clip       the same hot-path step on the reference's own 1080p clip (res/VID_1920x1080_cavlc_temporal_direct.264 decoded)
  intra_720p     BASELINE config 2: all-IDR 1280x720 on the reference's 720p clip, hot path, with its own roofline fraction
  cpu_baseline   the reference itself (oracle/_ref, C fallback, no asm) on this box's host cores: 1 thread and 4 slice threads
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# algorithmic bytes per intra MB = 384 src + 384 recon write + 768 deblock r/w + 960 record
BYTES_I_MB_PATH = 2496
BYTES_I_MB_MD = 384 + 384 + 960  # the mode-decision/reconstruction kernel's share (dominant kernel)
BYTES_P_MB_PATH = 2912
BYTES_P_MB_MD = 384 + 384 + 384 + 960 + 32
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
RES_DIR = os.path.join(REF_DIR, "res")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sessions", type=int, default=int(os.environ.get("WELSHIP_BENCH_SESSIONS", "0")),
                    help="independent pictures per GPU per step (default: 128 for the P workload = two slice workgroups per CU, 256 for all-IDR)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=24)
    ap.add_argument("--workload", default="p", choices=["intra", "p"])
    ap.add_argument("--content", default="synthetic", choices=["synthetic", "res"], help="res: the reference's own clip of that size (oracle/_ref/res)")
    ap.add_argument("--quick", action="store_true", help="only the headline leg (no e2e / latency / res clip / intra / cpu baseline / verification)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the e2e, latency, res-clip and intra legs")
    ap.add_argument("--host-threads", type=int, default=min(32, os.cpu_count() or 8))
    ap.add_argument("--multi-gpu-legs", action="store_true",
                    help="with --gpus N > 1: also run BASELINE configs 4 and 5 across the ranks (default there unless --quick; the launcher test asks for them explicitly)")
    ap.add_argument("--cpu-launcher-test", action="store_true",
                    help="TEST ONLY (tests/test_multi_rank.py): exercise the N-rank launch / barrier / max-over-ranks / JSON logic without a GPU -- "
                         "gloo instead of RCCL, WELSHIP_LIB must name the CPU test build of the kernels; implies --quick, the line is marked as no measurement")
    ap.add_argument("--deblock-idc", type=int, default=0, help="disable_deblocking_filter_idc (0: filter across slice boundaries, the reference default)")
    a = ap.parse_args()
    if a.cpu_launcher_test:
        a.quick = True
    if a.quick:
        a.no_cpu_baseline = a.no_verify = a.no_extra = True
    return a


# ---------------------------------------------------------------------------------------------------------- inputs
def decode_res_clip(name):
    """The reference's clip decoded by the reference decoder (oracle/_ref/ref_dec), or None when the oracle is not built."""
    dec, src = os.path.join(REF_DIR, "ref_dec"), os.path.join(RES_DIR, name)
    if not (os.path.exists(dec) and os.path.exists(src)):
        return None
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "clip.yuv")
        subprocess.check_call([dec, src, out], stdout=subprocess.DEVNULL)
        return open(out, "rb").read()


def slot_of(i, ring):
    """Source slot of the i-th step of a WelsHipGroupBench call (ping-pong over the resident ring)."""
    if ring == 1:
        return 0
    period = 2 * (ring - 1)
    k = i % period
    return k if k < ring else period - k


class Content:
    """Which frame sits in which resident slot of which session: session s, slot j -> frame (first(s) + j) of the clip."""

    def __init__(self, frames, fsz, ring, consecutive):
        self.frames, self.fsz, self.ring = frames, fsz, ring
        self.n = len(frames) // fsz
        self.consecutive = consecutive              # real clip: `ring` consecutive frames from a per-session offset
        self._cache = {}

    def frame(self, s, slot):
        if self.consecutive:
            k = (s * 5) % max(1, self.n - self.ring + 1) + slot
        elif self.n >= 2 * self.ring:
            k = (s * 3) % (self.n - self.ring + 1) + slot       # synthetic: every session a different, continuous stretch of the motion
        else:
            k = (slot + s) % self.n
        if k not in self._cache:
            self._cache[k] = self.frames[k * self.fsz:(k + 1) * self.fsz]
        return self._cache[k]


def cpu_info():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"model": model, "logical_cores": os.cpu_count()}


# ------------------------------------------------------------------------------------------------- reference legs
def ref_encode(yuv, w, h, flags, want_fps=False):
    enc = os.path.join(REF_DIR, "ref_enc")
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.yuv"), os.path.join(td, "out.264")
        open(fi, "wb").write(yuv)
        out = subprocess.check_output([enc, "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-fps", "30"] + flags).decode()
        bs = open(fo, "rb").read()
    if want_fps:
        kv = dict(t.split("=") for t in out.split())
        return bs, float(kv["fps"])
    return bs


def ref_decode_last_frame(bs, w, h):
    dec = os.path.join(REF_DIR, "ref_dec")
    with tempfile.TemporaryDirectory() as td:
        fi, fo = os.path.join(td, "in.264"), os.path.join(td, "out.yuv")
        open(fi, "wb").write(bs)
        subprocess.check_call([dec, fi, fo], stdout=subprocess.DEVNULL)
        fsz = w * h * 3 // 2
        with open(fo, "rb") as f:
            f.seek(-fsz, 2)
            return f.read()


def spread_sessions(n, k):
    """k sessions spread evenly over 0 .. n-1, first and last included (the ones whose reconstruction is checked after the timed steps)."""
    if n <= k:
        return tuple(range(n))
    return tuple(sorted({(i * (n - 1)) // (k - 1) for i in range(k)}))


def p_flags(qp, idc):
    return ["-iper", "0", "-qp", str(qp), "-slcmd", "1", "-slcnum", "4", "-deblock", str(idc)]


def cpu_baseline(w, h, qp, workload, content, idc, sessions=(0,)):
    """Time the reference (oracle/_ref/ref_enc) on a bounded sample of the same workload: the frame orders of the sessions whose reconstruction
    the timed steps verified (up to 8, spread over the batch), one after the other on ONE thread; and session 0's on 4 slice threads with
    -loadbalancing 0 and deblocking idc 2 (what the threaded reference needs for a deterministic stream).
    Which reference: oracle/_ref/ref_enc is the C fallback (`-O3`, no asm) -- the image has no nasm, so the SIMD library cannot be built here and,
    /root/reference not existing on the GPU box, not there either.  oracle/Makefile builds oracle/_ref/ref_enc_asm (the reference with its x86
    assembly, build/x86-common.mk) wherever nasm is on PATH at build time; when that binary travels with the snapshot it is timed as well and
    reported as `simd` (kind "reference SSE2/AVX2").  A reported baseline either way, not the optimisation target.
    (Letting the compiler vectorise the C path for AVX2 -- -O3 -march=x86-64-v3 -- is no stand-in for the assembly: measured in round 6, the same 1080p
    frames code at 16.8 instead of 20.2 frames/s on the build container's Xeon, identical stream; not built.)"""
    if not os.path.exists(os.path.join(REF_DIR, "ref_enc")):
        return None
    sessions = tuple(sessions) or (0,)
    n = max(12, (96 if w * h > 1280 * 720 else 200) * 2 // len(sessions)) if len(sessions) > 1 else (96 if w * h > 1280 * 720 else 200)
    ring = content.ring
    base = ["-iper", "1"] if workload == "intra" else p_flags(qp, idc)

    def run(exe_flags, sess, frames):
        yuv = b"".join(content.frame(sess, slot_of(i, ring)) for i in range(frames))
        return ref_encode(yuv, w, h, exe_flags, True)[1]
    per = [run(base + ["-quiet", "-threads", "1"], s_, n) for s_ in sessions]
    fps1 = len(per) / sum(1.0 / f for f in per)            # frames / total time over the sessions (equal frame counts)
    out = {"value": fps1, "unit": "frames/s", "cores": 1, "kind": "reference", "host": cpu_info(), "build": "C fallback, -O3, no asm",
           "per_session": dict(zip((str(s_) for s_ in sessions), per)),
           "sample": "%d frames %dx%d of each of %d sessions' frame orders (sessions %s of the batch), oracle/_ref (reference C fallback, no asm), one thread, timed around EncodeFrame"
                     % (n, w, h, len(sessions), list(sessions))}
    if workload == "p":
        flags4 = ["-iper", "0", "-qp", str(qp), "-slcmd", "1", "-slcnum", "4", "-deblock", "2", "-quiet", "-threads", "4", "-loadbalancing", "0"]
        out["threads4"] = {"value": run(flags4, sessions[0], 96 if w * h > 1280 * 720 else 200), "unit": "frames/s", "cores": 4, "flags": "-threads 4 -loadbalancing 0 -deblock 2"}
    asm = os.path.join(REF_DIR, "ref_enc_asm")
    import shutil
    if os.path.exists(asm):
        def run_asm(flags, frames):
            yuv = b"".join(content.frame(sessions[0], slot_of(i, ring)) for i in range(frames))
            with tempfile.TemporaryDirectory() as td:
                fi, fo = os.path.join(td, "in.yuv"), os.path.join(td, "out.264")
                open(fi, "wb").write(yuv)
                o = subprocess.check_output([asm, "-i", fi, "-w", str(w), "-h", str(h), "-o", fo, "-rc", "-1", "-fps", "30"] + flags).decode()
            return float(dict(t.split("=") for t in o.split())["fps"])
        try:
            out["simd"] = {"value": run_asm(base + ["-quiet", "-threads", "1"], 3 * n), "unit": "frames/s", "cores": 1, "kind": "reference SSE2/AVX2",
                           "sample": "%d frames of session %d, oracle/_ref/ref_enc_asm (the reference with its x86 assembly)" % (3 * n, sessions[0])}
        except Exception as e:          # noqa: BLE001
            out["simd"] = {"available": False, "why": "oracle/_ref/ref_enc_asm failed: %s" % str(e)[:160]}
    else:
        out["simd"] = {"available": False, "nasm_on_this_box": shutil.which("nasm") is not None,
                       "why": "oracle/_ref/ref_enc_asm is not in the snapshot: the build image has no nasm (oracle/Makefile builds it where nasm is on PATH), and "
                              "/root/reference does not exist on the GPU box, so it cannot be built here either"}
    return out


# ----------------------------------------------------------------------------------------------------- GPU legs
def make_group(oh, a, local, w, h, workload, sessions, ring, content, idc=None, host_threads=None):
    e = oh.Encoder()
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate = w, h, a.qp, 30.0, 5000000
    p.iDevice = local
    p.iLoopFilterDisableIdc = a.deblock_idc if idc is None else idc
    if workload == "intra":
        p.uiIntraPeriod = 1
    else:
        p.uiIntraPeriod = 0
        p.uiSliceMode, p.uiSliceNum = 1, 4
    g = oh.EncoderGroup(p, sessions, ring_slots=ring, host_threads=host_threads or a.host_threads)
    if content is not None:
        for s in range(sessions):
            for slot in range(ring):
                g.upload(s, slot, content.frame(s, slot))
    return g


def hot_path_leg(oh, a, local, w, h, workload, sessions, ring, content, steps, warmup, barrier=None, verify_sessions=()):
    """IDR (P workload) + warmup + `steps` timed device-only steps.  Returns (wall seconds of the timed steps, event dict,
    verification result or None)."""
    g = make_group(oh, a, local, w, h, workload, sessions, ring, content)
    order = []                                       # source slots in coding order (the reference is fed the same frames)
    if workload == "p":
        g.bench(1, 0)                                # the IDR that starts every stream is not part of the timed P steps
        order.append(slot_of(0, ring))
    if barrier:
        barrier()
    if warmup > 0:
        g.bench(warmup, 0)
        order += [slot_of(i, ring) for i in range(warmup)]
    if barrier:
        barrier()
    t0 = time.perf_counter()
    ev = g.bench(steps, 0)
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    order += [slot_of(i, ring) for i in range(steps)]
    verified = None
    if verify_sessions:
        def check(s):
            yuv = b"".join(content.frame(s, k) for k in order)
            flags = (["-iper", "1", "-qp", str(a.qp)] if workload == "intra" else p_flags(a.qp, a.deblock_idc)) + ["-quiet", "-threads", "1"]
            return ref_decode_last_frame(ref_encode(yuv, w, h, flags), w, h)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(8, len(verify_sessions))) as ex:          # (the reference runs as a subprocess per session)
            refs = list(ex.map(check, verify_sessions))
        verified = {s: g.recon(s) == r for s, r in zip(verify_sessions, refs)}
    g.close()
    return dt, ev, verified


def e2e_leg(oh, a, local, w, h, sessions, ring, content, frames, check=False):
    """Complete EncodeFrame calls (source upload + device + D2H + host CAVLC) for `sessions` concurrent streams."""
    g = make_group(oh, a, local, w, h, "p", sessions, ring, None)
    pics = [g.make_pictures([content.frame(s, k) for s in range(sessions)]) for k in range(ring)]
    order = [0, 1 % ring] + [slot_of(i + 2, ring) for i in range(frames)]
    bs0 = bytearray()
    bs0 += g.encode_frames(pics[order[0]], want_bytes=True)[0]       # IDR
    bs0 += g.encode_frames(pics[order[1]], want_bytes=True)[0]       # first P: page-locking and first-touch effects stay out of the timing
    t0 = time.perf_counter()
    nbytes = 0
    for i in range(frames):
        out = g.encode_frames(pics[order[i + 2]], want_bytes=check)
        if check:
            bs0 += out[0]
            nbytes += sum(len(b) for b in out)
        else:
            nbytes += out
    dt = time.perf_counter() - t0
    host = g.host_stats() if hasattr(g, "host_stats") else None
    g.close()
    match = None
    if check:      # session 0's bitstream against the reference encoder on the same frames
        import hashlib
        ref = ref_encode(b"".join(content.frame(0, k) for k in order), w, h, p_flags(a.qp, a.deblock_idc) + ["-quiet", "-threads", "1"])
        match = {"match": bytes(bs0) == ref, "sha1": hashlib.sha1(bytes(bs0)).hexdigest(), "reference_sha1": hashlib.sha1(ref).hexdigest(), "frames": len(order)}
    e2e_leg.host = host
    return dt, nbytes, match


def e2e_pipelined_leg(oh, a, local, w, h, sessions, ring, content, frames, check=False):
    """The same complete EncodeFrame work as e2e_leg through WelsHipGroupEncodeFramesPipelined: one group, the call that submits step k
    (staging copy, H2D, kernels) entropy-codes step k - 1 meanwhile.  Timed from the first timed submission to the flush that returns
    the last step's streams: `frames` steps submitted and `frames` steps finished inside the region.  Also returns the rate over the
    second half of the region (the first P pictures after the IDR carry more residual than the later ones)."""
    # (its two halves run at the same time, each on its own team of host threads: staging copies | entropy coding)
    e2e_pipelined_leg.threads = int(os.environ.get("WELSHIP_PIPE_THREADS", "0")) or max(4, a.host_threads // 2)
    g = make_group(oh, a, local, w, h, "p", sessions, ring, None, host_threads=e2e_pipelined_leg.threads)
    ahead = e2e_pipelined_leg.ahead = max(1, min(3, int(os.environ.get("WELSHIP_PIPE_AHEAD", "2"))))
    g.set_pipelined(ahead)
    pics = [g.make_pictures([content.frame(s, k) for s in range(sessions)]) for k in range(ring)]
    order = [0, 1 % ring] + [slot_of(i + 2, ring) for i in range(frames)]
    bs0 = bytearray()
    for k in (0, 1):                                                          # the IDR and the first P picture, then the pipeline is emptied again
        out = g.encode_frames_pipelined(pics[order[k]], want_bytes=True)
        if out is not None:
            bs0 += out[0]
    while True:
        out = g.encode_frames_pipelined(None, want_bytes=True)
        if out is None:
            break
        bs0 += out[0]
    t0 = time.perf_counter()
    nbytes = 0
    stamps = []
    i = done = 0
    while done < frames:
        out = g.encode_frames_pipelined(pics[order[i + 2]] if i < frames else None, want_bytes=check)
        i += 1
        if out is None:
            continue
        done += 1
        stamps.append(time.perf_counter())          # step `done` finished
        if check:
            bs0 += out[0]
            nbytes += sum(len(b) for b in out)
        else:
            nbytes += out
    dt = time.perf_counter() - t0
    half = frames // 2
    steady = sessions * (frames - half) / (stamps[frames - 1] - stamps[half - 1]) if frames >= 4 else None      # steps half+1 .. frames finished in that time
    host = g.host_stats()
    g.close()
    match = None
    if check:
        import hashlib
        ref = ref_encode(b"".join(content.frame(0, k) for k in order), w, h, p_flags(a.qp, a.deblock_idc) + ["-quiet", "-threads", "1"])
        match = {"match": bytes(bs0) == ref, "sha1": hashlib.sha1(bytes(bs0)).hexdigest(), "reference_sha1": hashlib.sha1(ref).hexdigest(), "frames": len(order)}
    e2e_pipelined_leg.steady = steady
    return dt, nbytes, match, host


def binding_leg(args, local, extra_env=None, timeout=400):
    """One run of tools/config5_sessions.py (N ISVCEncoder objects of the patched reference in one process, this engine behind SWelsFuncPtrList,
    next to the same sessions on the reference's C path) -> the digest bench lines carry."""
    env = dict(os.environ, WELS_HIP_DEVICE=str(local))
    for k, v in (extra_env or {}).items():
        if v is None:
            env.pop(k, None)
        else:
            env[k] = v
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "config5_sessions.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, env=env)
    lines = r.stdout.decode(errors="replace").strip().splitlines()
    if r.returncode != 0 or not lines:
        raise RuntimeError("tools/config5_sessions.py %s: exit code %d, stderr tail: %s" % (" ".join(args), r.returncode, r.stderr.decode(errors="replace")[-400:]))
    j = json.loads(lines[-1])
    n = int(args[0])
    dev, c = j["hooks_on_device"], j["reference_c_path"]
    return {"config": j["config"], "same_bitstreams": j["same_bitstreams"],
            "device_frames_per_s": dev["sum_of_session_encode_fps"], "device_slowest_session_fps": dev["min_session_fps"],
            "device_ms_per_frame_per_session": 1e3 * n / dev["sum_of_session_encode_fps"],
            "c_path_frames_per_s": c["sum_of_session_encode_fps"], "c_path_cores": n,
            "pictures_on_device": dev.get("device_pictures"), "pictures_left_to_the_host": dev.get("host_pictures"),
            "note": "frames/s = sum over the sessions of frames / time inside EncodeFrame; every layer of a simulcast frame counts as part of ONE frame"}, r.stderr.decode(errors="replace")


def multi_gpu_legs(rank, world, local, dist, small, visible0=None):
    """BASELINE configs 5 and 4 over the ranks of one node (no data-path collective: all_gather_object of the per-rank digests only).
    config5_64_sessions: every rank hosts 8 concurrent 1080p sessions (rate control, raster slices) on its GPU, all ranks at once -- 8 x N sessions
    per node, 64 on 8 GPUs; the C-path twin of every rank runs at the same time too, so the host's cores are shared by 8 x N reference sessions.
    config4_layer_per_gpu: ONE simulcast session (1080p input, four AVC layers) on rank 0's process with its layers on GPUs 0 .. min(N, 4) - 1
    (WELS_HIP_LAYER_DEVICES, integration/welship_hooks.cpp), the other ranks idle at the barrier.
    small: the launcher test's sizes (2 sessions of 3 frames at 720p / one 2-frame simulcast session) on the CPU test build."""
    import re
    out = {}
    args5 = ["2", "3", "plain"] if small else ["8", "54", "plain", "1080p"]
    try:
        mine, _ = binding_leg(args5, local)
    except Exception as e:          # noqa: BLE001 -- reported in the line
        mine = {"error": str(e)[:200]}
    mine["rank"], mine["device"] = rank, os.environ.get("HIP_VISIBLE_DEVICES", str(local))
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        ok = [p for p in parts if "error" not in p]
        out["config5_64_sessions" if world == 8 else "config5_%d_sessions" % (int(args5[0]) * world)] = {
            "n_ranks_seen": len(parts), "sessions_per_gpu": int(args5[0]), "sessions": int(args5[0]) * world,
            "aggregate_device_frames_per_s": sum(p["device_frames_per_s"] for p in ok), "aggregate_c_path_frames_per_s": sum(p["c_path_frames_per_s"] for p in ok),
            "per_gpu_device_frames_per_s": [p.get("device_frames_per_s") for p in parts],
            "per_session_latency_ms": [p.get("device_ms_per_frame_per_session") for p in parts],
            "same_bitstreams": bool(ok) and len(ok) == len(parts) and all(p["same_bitstreams"] for p in ok), "errors": [p["error"] for p in parts if "error" in p],
            "config": ok[0]["config"] if ok else None,
            "c_path_comparable_with_single_gpu_leg": False,
            "note": "all ranks at once; the C-path sessions of all ranks share the host's cores (%d reference sessions on %s logical cores): the aggregate C-path figure "
                    "is measured on oversubscribed cores and is NOT comparable with the single-GPU leg's (config5_8_sessions_1080p_rc_raster_slices)" % (int(args5[0]) * world, os.cpu_count())}
    dist.barrier()
    if rank == 0:
        n = min(world, 4)
        try:
            leg, err = binding_leg(["1", "2", "simulcast"] if small else ["1", "54", "simulcast", "1080p"], 0,
                                   {"WELS_HIP_LAYER_DEVICES": str(n if n >= 2 else 0), "WELSHIP_TRACE_DEVICES": "1", "WELS_HIP_TRACE": "1",
                                    "HIP_VISIBLE_DEVICES": visible0})          # (this one session sees all the node's GPUs again)
            leg["layer_devices"] = n
            seen = sorted(set(int(x) for x in re.findall(r"backend for device (\d+)", err)))
            if seen:
                leg["devices_seen"] = seen        # (the CPU test build's backend reports the device index it was asked for)
            out["config4_layer_per_gpu"] = leg
        except Exception as e:      # noqa: BLE001
            out["config4_layer_per_gpu"] = {"error": str(e)[:200]}
    dist.barrier()
    return out


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU; this process only launches them and relays rank 0's line
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_test = a.cpu_launcher_test
    # One rank = one GPU, pinned BEFORE anything initialises HIP: the rank sees exactly its own device (index 0), whatever the launcher's
    # HIP_VISIBLE_DEVICES was -- no plumbing left that the first real N-GPU run could trip over (a library or a child process that forgets the
    # device index lands on the right GPU anyway).  `visible0`: the launcher's own setting, for the one leg that spreads ONE session over
    # several GPUs (config4_layer_per_gpu).
    visible0 = os.environ.get("HIP_VISIBLE_DEVICES")
    dev = local
    if world > 1 and not cpu_test:
        ids = [x for x in visible0.split(",") if x != ""] if visible0 else [str(i) for i in range(world)]
        if local >= len(ids):
            raise SystemExit("bench.py: rank %d has no GPU (HIP_VISIBLE_DEVICES=%r)" % (local, visible0))
        os.environ["HIP_VISIBLE_DEVICES"] = ids[local]
        dev = 0
    import torch
    if cpu_test:
        if "emu" not in os.path.basename(os.environ.get("WELSHIP_LIB", "")):
            raise SystemExit("--cpu-launcher-test needs WELSHIP_LIB = the CPU test build (tests/emu/libwelship_emu.so)")
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    else:
        torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        # the harness' barrier and its max-over-ranks of ONE float go over gloo: the path has no exchange step, so no RCCL
        # communicator is brought up at all
        import torch.distributed as dist
        dist.init_process_group("gloo")
    import openh264_amd as oh
    from openh264_amd.utils.synth import synth_sequence

    def barrier():
        if not cpu_test:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if not cpu_test:
            torch.cuda.synchronize()

    workload = a.workload
    if a.sessions <= 0:
        a.sessions = 256        # 1024 slices on 256 CUs: four slices share every 12-wave workgroup of the mode-decision pool
    w, h = a.width, a.height
    mbs = ((w + 15) // 16) * ((h + 15) // 16)
    fsz = w * h * 3 // 2
    ring = 2 if workload == "intra" else 8          # the library keeps at least two source slots per session
    have_ref = os.path.exists(os.path.join(REF_DIR, "ref_enc")) and os.path.exists(os.path.join(REF_DIR, "ref_dec"))
    if a.content == "res":
        clip = decode_res_clip("VID_%dx%d_cavlc_temporal_direct.264" % (w, h))
        if clip is None:
            raise SystemExit("--content res needs oracle/_ref (ref_dec + res/)")
        content = Content(clip, fsz, ring, True)
        data = "res/VID_%dx%d_cavlc_temporal_direct.264 decoded by the reference decoder, %d consecutive frames per session" % (w, h, ring)
    else:
        content = Content(synth_sequence(w, h, 4 if workload == "intra" else 2 * ring), fsz, ring, False)
        data = "synthetic"

    verify_sessions = () if (a.no_verify or not have_ref or rank != 0) else spread_sessions(a.sessions, 8)
    dt, ev, verified = hot_path_leg(oh, a, dev, w, h, workload, a.sessions, ring, content, a.steps, a.warmup, barrier, verify_sessions)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    mg = {}
    if dist is not None and world > 1 and (a.multi_gpu_legs or not a.quick) and os.path.exists(os.path.join(REF_DIR, "ref_enc_hip")):
        mg = multi_gpu_legs(rank, world, dev, dist, cpu_test, visible0)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    def roofline(workload, mbs, sessions, steps, ev):
        b_md = BYTES_I_MB_MD if workload == "intra" else BYTES_P_MB_MD
        b_path = BYTES_I_MB_PATH if workload == "intra" else BYTES_P_MB_PATH
        md_launch_ms = ev["md_ms"] / steps                                   # one launch per pass and step
        achieved = b_md * mbs * sessions / (md_launch_ms * 1e-3) / 1e9       # every MB of every picture in the batch x bytes/MB
        return {"bound": "hbm", "kernel": "k_intra_slice" if workload == "intra" else "k_inter_pool",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "bytes_per_mb": b_md, "avg_launch_ms": md_launch_ms, "launches_per_step": 1,
                "path_achieved_GBs": b_path * mbs * sessions * steps / (ev["total_ms"] * 1e-3) / 1e9,
                "path_frac": b_path * mbs * sessions * steps / (ev["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "events_ms": ev}

    rf = roofline(workload, mbs, a.sessions, a.steps, ev)
    rf["traffic"] = None
    try:        # HBM bytes per launch of the dominant kernel: from the committed rocprofv3 --pmc passes of this very command (not measured in this run)
        tj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))
        if tj.get("workload") == workload and tj.get("sessions") == a.sessions and (w, h) == (tj.get("width"), tj.get("height")):
            want = "tickets" if rf["kernel"] == "k_inter_pool" else None
            for name, v in tj["schedulers"].items():
                if want and name.startswith(want):
                    rf["traffic"] = v["hbm_bytes_per_launch"]
                    rf["traffic_over_algorithmic"] = v["times_algorithmic"]
                    rf["traffic_source"] = "profiles/r06_pmc_traffic.json, '%s' (separate rocprofv3 --pmc passes of this command; not measured in this run)" % name
    except Exception:
        pass
    pics = a.sessions * a.steps * world
    line = {
        "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"],
        "metric_scope": "value = device hot path (MD + reconstruction + deblocking + border expansion), entropy coding on the host excluded; "
                        "`verified` = the timed steps' reconstruction equals the reference's, `e2e_pipelined` = complete EncodeFrame rate with a bitstream SHA1 check",
        "value": pics / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": data if not cpu_test else data + " -- LAUNCHER TEST on the CPU test build of the kernels: not a measurement",
        "config": {"workload": ("%dx%d all-IDR (intra MD + DCT/quant + deblock), QP %d, LOW complexity" % (w, h, a.qp)) if workload == "intra" else
                   ("%dx%d P-frames, diamond ME range 16, 4 slices/frame, QP %d, LOW complexity" % (w, h, a.qp)),
                   "pictures_in_flight_per_gpu": a.sessions,
                   "hot_path": "device MD/recon + deblock + border expand; sources resident in HBM, MB records left in HBM; host CAVLC excluded (see e2e_pipelined)",
                   "note": "throughput needs many concurrent pictures per GPU (measured in round 6: 224 in flight 23.9 k, 160: 20.5 k, 128: 19.4 k, 96: 14.8 k, 64: 11.4 k frames/s); see latency for 1 and 8 sessions",
                   "parallelism": "sessions sharded over %d GPU(s), no collective" % world},
        "roofline": rf,
    }
    if verified is not None:
        line["verified"] = all(verified.values())
        line["verified_detail"] = {"sessions": sorted(verified), "frames_each": 1 + a.warmup + a.steps if workload == "p" else a.warmup + a.steps,
                                   "against": "oracle/_ref: reference encoder + reference decoder on the same frame order, last reconstruction compared byte for byte"}

    if not a.no_extra and world == 1:
        # complete EncodeFrame work end to end (source upload, device passes, D2H of the packed records, host CAVLC): one group as a two-stage pipeline,
        # the host entropy-codes step k - 1 while the device codes step k.  (The synchronous and the several-groups legs of rounds 1-3 are gone:
        # 4.2 k and 4.1 k frames/s against this leg's 13.4 k, profiles/r04_bench_default_final.json.)
        n3 = 60
        dp, nb3, m3, host3 = e2e_pipelined_leg(oh, a, dev, w, h, a.sessions, ring, content, n3, bool(verify_sessions))
        line["e2e_pipelined"] = {"frames_per_s": a.sessions * n3 / dp, "frames_per_s_second_half": e2e_pipelined_leg.steady,
                                 "sessions": a.sessions, "frames_each": n3, "host_threads_per_half": e2e_pipelined_leg.threads, "steps_ahead": e2e_pipelined_leg.ahead,
                                 "bitstream_MB_per_s": nb3 / dp / 1e6, "bitstream_vs_reference": m3, "host_thread_ms_per_picture": host3,
                                 "how": "WelsHipGroupEncodeFramesPipelined: staging copy + H2D + kernels of step k queued, then D2H + CAVLC of step k - 1 under them"}
        lat = {}
        for ns in (1, 8):
            dl, _, _ = e2e_leg(oh, a, dev, w, h, ns, ring, content, 20)
            lat["sessions_%d" % ns] = {"ms_per_frame": dl / 20 * 1e3, "frames_per_s": ns * 20 / dl}
        line["latency"] = lat
        # the same step on the reference's own 1080p clip
        if a.content == "synthetic" and (w, h) == (1920, 1080) and workload == "p":
            clip = decode_res_clip("VID_1920x1080_cavlc_temporal_direct.264")
            if clip is not None:
                c2 = Content(clip, fsz, ring, True)
                st = max(10, min(a.steps, 50))
                d2, ev2, ver2 = hot_path_leg(oh, a, dev, w, h, "p", a.sessions, ring, c2, st, 2, None, (0,) if verify_sessions else ())
                r2 = roofline("p", mbs, a.sessions, st, ev2)
                line["res_clip"] = {"data": "res/VID_1920x1080_cavlc_temporal_direct.264 decoded, %d consecutive frames per session, sessions start 5 frames apart" % ring,
                                    "value": a.sessions * st / d2, "unit": "frames/s", "steps": st, "roofline_frac": r2["frac"], "events_ms": ev2,
                                    "verified": (all(ver2.values()) if ver2 is not None else None)}
        # BASELINE config 2: 720p all-IDR on the reference's 720p clip
        if workload == "p":
            clip = decode_res_clip("VID_1280x720_cavlc_temporal_direct.264")
            if clip is not None:
                w2, h2 = 1280, 720
                c3 = Content(clip[: w2 * h2 * 3 // 2 * 16], w2 * h2 * 3 // 2, 2, True)
                st, ns = 12, 512        # (two single-slice pictures per CU: the intra kernel then runs two 12-wave workgroups on each)
                d3, ev3, ver3 = hot_path_leg(oh, a, dev, w2, h2, "intra", ns, 2, c3, st, 2, None, (0,) if verify_sessions else ())
                r3 = roofline("intra", ((w2 + 15) // 16) * ((h2 + 15) // 16), ns, st, ev3)
                line["intra_720p"] = {"data": "res/VID_1280x720_cavlc_temporal_direct.264 decoded (BASELINE config 2's stand-in clip)", "value": ns * st / d3, "unit": "frames/s",
                                      "sessions": ns, "steps": st, "roofline": {k: r3[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "bytes_per_mb")},
                                      "verified": (all(ver3.values()) if ver3 is not None else None)}
        # BASELINE configs 4 and 5 at the sizes BASELINE.json states, through the dispatch-table binding (the reference's own frame layer,
        # rate control and entropy coder; oracle/_ref/ref_enc_hip -parallel N = N ISVCEncoder objects on N threads of one process), next to
        # the same sessions on the reference's C path on as many host cores.  One GPU's share: config 4 = one 1080p input coded as
        # 1080p / 720p / 360p / 180p simulcast AVC layers (all four layers on this GPU; one layer per GPU is WELS_HIP_LAYER_DEVICES=1),
        # config 5 = 8 of the 64 concurrent 1080p sessions (rate control in bitrate mode, raster slices of 2040 macroblocks).
        if workload == "p" and os.path.exists(os.path.join(REF_DIR, "ref_enc_hip")):
            # (config 4 as the binding runs it by default: the 360p and 180p layers -- 920 and 240 macroblocks -- are coded by the session's own thread on the
            #  reference's path, 1080p and 720p on the device, integration/welship_hooks.cpp pfHipLayerOnDevice; "..._all_layers_on_device" switches that off)
            for key, args, env in (("config4_simulcast_1080p_720p_360p_180p", ["1", "54", "simulcast", "1080p"], None), ("config4_8_sessions", ["8", "54", "simulcast", "1080p"], None),
                                   ("config4_8_sessions_all_layers_on_device", ["8", "54", "simulcast", "1080p"], {"WELS_HIP_MIN_LAYER_MBS": "0"}),
                                   ("config5_8_sessions_1080p_rc_raster_slices", ["8", "54", "plain", "1080p"], None)):      # (the whole 54-frame clip: rounds 2-4 coded its first 30
                # frames, of which the sessions of the device leg spend about ten falling into step -- a running service is the steady state)
                try:
                    line[key], _ = binding_leg(args, dev, extra_env=env, timeout=240)
                except Exception as e:
                    line[key] = {"error": str(e)[:200]}
    # what an encoder delivers, next to the contract's `value` (the device hot path): complete EncodeFrame work with a verified bitstream, and the
    # hot path on the reference's own 1080p content instead of the synthetic pan
    if "e2e_pipelined" in line:
        line["value_e2e"] = line["e2e_pipelined"]["frames_per_s"]
        line["value_e2e_note"] = "frames/s of complete EncodeFrame work (H2D of the sources, device passes, D2H of the packed records, host CAVLC), bitstreams == the reference's; PCIe-bound"
    if "res_clip" in line:
        line["value_res_clip"] = line["res_clip"]["value"]
    line.update(mg)
    if world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(w, h, a.qp, workload, content, a.deblock_idc, verify_sessions or (0,))
        if cb:
            line["cpu_baseline"] = cb
    print(json.dumps(line))
    sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for leg in ("e2e_pipelined",):
        if line.get(leg, {}).get("bitstream_vs_reference") and not line[leg]["bitstream_vs_reference"]["match"]:
            raise SystemExit("bench.py: the %s bitstream of session 0 differs from the reference encoder's" % leg)
    if verified is not None and not all(verified.values()):
        raise SystemExit("bench.py: the timed steps' reconstruction differs from the reference (sessions %s)" % [s for s, ok in verified.items() if not ok])


if __name__ == "__main__":
    main()
