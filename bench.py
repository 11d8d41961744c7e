#!/usr/bin/env python3
"""bench.py -- hot-path throughput of the MI355X macroblock engine ("Measurement").

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path (mode decision + reconstruction, in-loop deblocking, border
expansion) over one batch: `--sessions` independent 1080p pictures per GPU (default 128 = two slice
workgroups per CU), sources already resident in HBM, MB records left in HBM.
value = pictures processed by all ranks / time.
Multi-GPU: one process per GPU, sessions sharded over ranks, no data-path collective ("weak").
Extra objects on the JSON line: `roofline` (dominant kernel, HIP events on the launch stream) and
`cpu_baseline` (the reference itself, oracle/_ref, timed on this box's host cores, rank 0, N=1).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sessions", type=int, default=int(os.environ.get("WELSHIP_BENCH_SESSIONS", "0")),
                    help="independent pictures per GPU per step (default: 128 for the P workload = two slice workgroups per CU, 256 for all-IDR = one picture workgroup per CU)")
    ap.add_argument("--queues", type=int, default=int(os.environ.get("WELSHIP_QUEUES", "1")), help="device queues the sessions are spread over (kernels of different queues overlap)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--qp", type=int, default=24)
    ap.add_argument("--workload", default="auto", choices=["auto", "intra", "p"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true", help="also time the full encode incl. D2H + host CAVLC")
    ap.add_argument("--host-threads", type=int, default=min(32, os.cpu_count() or 8))
    ap.add_argument("--deblock-idc", type=int, default=0, help="disable_deblocking_filter_idc (0: filter across slice boundaries, the reference default)")
    return ap.parse_args()


def cpu_baseline(width, height, qp, workload, frames_src):
    """Time the reference (oracle/_ref/ref_enc, C fallback, 1 thread) on a bounded sample of the same workload."""
    enc = os.path.join(ROOT, "oracle", "_ref", "ref_enc")
    if not os.path.exists(enc):
        return None
    fsz = width * height * 3 // 2
    n_unique = len(frames_src) // fsz
    n = 96 if width * height > 1280 * 720 else 200          # ~ 5-10 s of single-core work
    with tempfile.TemporaryDirectory() as td:
        fi = os.path.join(td, "in.yuv")
        with open(fi, "wb") as f:
            period = max(1, 2 * (n_unique - 1))
            for i in range(n):
                k = i % period if n_unique > 1 else 0
                k = k if k < n_unique else period - k
                f.write(frames_src[k * fsz:(k + 1) * fsz])
        flags = ["-rc", "-1", "-qp", str(qp), "-fps", "30", "-quiet", "-threads", "1"]
        flags += ["-iper", "1"] if workload == "intra" else ["-iper", "0", "-slcmd", "1", "-slcnum", "4"]
        out = subprocess.check_output([enc, "-i", fi, "-w", str(width), "-h", str(height)] + flags).decode()
    kv = dict(t.split("=") for t in out.split())
    return {"value": float(kv["fps"]), "unit": "frames/s", "cores": 1, "kind": "reference",
            "sample": "%d frames %dx%d, oracle/_ref (reference C fallback, no asm), 1 thread, timed around EncodeFrame" % (n, width, height)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    os.environ["WELSHIP_QUEUES"] = str(a.queues)
    import openh264_amd as oh
    from openh264_amd.utils.synth import synth_sequence

    workload = a.workload
    if workload == "auto":
        workload = "p" if getattr(oh, "HAS_INTER_PATH", False) else "intra"
    if a.sessions <= 0:
        a.sessions = 128 if workload == "p" else 256
    w, h = a.width, a.height
    mbs = ((w + 15) // 16) * ((h + 15) // 16)
    ring = 2 if workload == "intra" else 8          # the library keeps at least two source slots per session
    n_unique = 4 if workload == "intra" else ring
    frames = synth_sequence(w, h, n_unique)
    fsz = w * h * 3 // 2

    e = oh.Encoder()
    p = e.GetDefaultParams()
    e.close()
    p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate = w, h, a.qp, 30.0, 5000000
    p.iDevice = local
    p.iLoopFilterDisableIdc = a.deblock_idc
    if workload == "intra":
        p.uiIntraPeriod = 1
    else:
        p.uiIntraPeriod = 0
        p.uiSliceMode, p.uiSliceNum = 1, 4
    g = oh.EncoderGroup(p, a.sessions, ring_slots=ring, host_threads=a.host_threads)
    for s in range(a.sessions):
        for slot in range(ring):
            k = (s + slot) % n_unique if workload == "intra" else slot
            g.upload(s, slot, frames[k * fsz:(k + 1) * fsz])
    if workload == "p":
        g.bench(1, 0)          # the IDR that starts every stream is not part of the timed P-frame steps

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    if a.warmup > 0:
        g.bench(a.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    ev = g.bench(a.steps, 0)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    e2e = None
    if a.e2e and rank == 0:
        t1 = time.perf_counter()
        for i in range(a.steps):
            g.step(i % ring)
        e2e = a.sessions * a.steps / (time.perf_counter() - t1)

    if rank == 0:
        pics = a.sessions * a.steps * world
        nd = 1                                                                # one launch per pass: a workgroup walks a whole slice
        md_launch_ms = ev["md_ms"] / (a.steps * nd)
        b_md = BYTES_I_MB_MD if workload == "intra" else BYTES_P_MB_MD
        b_path = BYTES_I_MB_PATH if workload == "intra" else BYTES_P_MB_PATH
        bytes_per_launch = b_md * mbs * a.sessions / nd                     # every MB of every picture in the batch x bytes/MB
        achieved = bytes_per_launch / (md_launch_ms * 1e-3) / 1e9
        traffic = None
        try:        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this very command
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if tj.get("workload") == workload and tj.get("sessions") == a.sessions and (w, h) == (tj.get("width"), tj.get("height")):
                traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "1080p frames/sec/GPU at QP=24 CBP; encoder_binary_comparison SHA1 pass",
            "value": pics / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("%dx%d all-IDR (intra MD + DCT/quant + deblock), QP %d, LOW complexity" % (w, h, a.qp)) if workload == "intra" else
                       ("%dx%d P-frames, diamond ME range 16, 4 slices/frame, QP %d, LOW complexity" % (w, h, a.qp)),
                       "pictures_in_flight_per_gpu": a.sessions, "device_queues": a.queues, "hot_path": "device MD/recon + deblock + border expand; sources resident in HBM, MB records left in HBM; host CAVLC excluded",
                       "parallelism": "sessions sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "k_intra_slice" if workload == "intra" else "k_inter_slice",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_mb": b_md, "avg_launch_ms": md_launch_ms, "launches_per_step": nd,
                         "path_achieved_GBs": b_path * mbs * a.sessions * a.steps / (ev["total_ms"] * 1e-3) / 1e9,
                         "events_ms": ev},
        }
        if e2e is not None:
            line["e2e_frames_per_s_incl_d2h_and_host_cavlc"] = e2e
            line["host_entropy_threads"] = a.host_threads
        if world == 1 and not a.no_cpu_baseline:
            cb = cpu_baseline(w, h, a.qp, workload, frames)
            if cb:
                line["cpu_baseline"] = cb
        print(json.dumps(line))
    g.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
