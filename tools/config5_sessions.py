"""BASELINE config 5's per-GPU share: N concurrent 720p30 sessions with rate control on and raster slices (N-MB slices), each one
an ISVCEncoder instance of the reference with this engine installed behind SWelsFuncPtrList (oracle/_ref/ref_enc_hip -parallel N:
N instances on N threads of one process, integration/welship_hooks.cpp) -- against the same instances on the reference's C path
(WELS_HIP=0).  One process per GPU hosting its sessions is the deployment: kernels of different sessions then share the device
(different processes would be time-sliced).  Prints one JSON line.

usage: config5_sessions.py [sessions=8] [frames=60] [mode] [1080p]   (needs oracle/_ref incl. res/; the GPU leg needs an MI355X)
A fourth argument "1080p" gives BASELINE's sizes as stated: config 5 = 1920x1080 sessions (raster slices of 2040 MBs = 4 slices, 4 Mbit/s),
config 4 ("simulcast") = 1080p / 720p / 360p / 180p layers instead of the 540 / 270 / 135 ladder."""
import hashlib, json, os, re, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 60
# third argument "simulcast": BASELINE config 4's shape instead -- every session a 1080p input coded as four simulcast AVC layers
# (1920x1080, 960x540, 480x270, 240x135; four slices each), all layers on this one GPU
SIMULCAST = len(sys.argv) > 3 and sys.argv[3] == "simulcast"
# third argument "screen": screen-content sessions instead (the SHA1 table's 1024x768 clip, bitrate mode, four slices)
SCREEN = len(sys.argv) > 3 and sys.argv[3] == "screen"
# third argument "gom": the 720p sessions with ONE slice per picture (GOM-level QP; set WELS_HIP_GOM=1 or 2, else the hooks decline)
GOM = len(sys.argv) > 3 and sys.argv[3] == "gom"
# third argument "dynslice": the 720p sessions with size-limited slices of 1200 bytes (what an RTP caller asks for; WELS_HIP_DYNSLICE=1 is set here)
DYNSLICE = len(sys.argv) > 3 and sys.argv[3] == "dynslice"
LIB = os.environ.get("WELSHIP_LIB") or os.path.join(ROOT, "openh264_amd", "libwelship.so")
FULL = len(sys.argv) > 4 and sys.argv[4] == "1080p"
W, H = (1920, 1080) if (SIMULCAST or (FULL and not SCREEN)) else (1024, 768) if SCREEN else (1280, 720)
LADDER = ["-simulcast", "320", "180", "-simulcast", "640", "360", "-simulcast", "1280", "720"] if FULL else ["-simulcast", "240", "135", "-simulcast", "480", "270", "-simulcast", "960", "540"]


def run(tmp, yuv, hip):
    env = dict(os.environ, WELSHIP_LIB=LIB, WELS_HIP="1" if hip else "0", WELS_HIP_TRACE=os.environ.get("WELS_HIP_TRACE", "1"))
    if DYNSLICE:
        env["WELS_HIP_DYNSLICE"] = "1"
    out = os.path.join(tmp, "s_%d.264" % hip)
    cmd = [os.path.join(REF, "ref_enc_hip"), "-parallel", str(N), "-i", yuv, "-w", str(W), "-h", str(H), "-o", out, "-frames", str(FRAMES),
           "-fps", "30", "-rc", "1", "-bitrate", "4000000" if FULL and not SIMULCAST else "1500000", "-threads", "1", "-iper", "0", "-quiet"]
    cmd += (["-slcmd", "1", "-slcnum", "4"] + LADDER if SIMULCAST
            else ["-usage", "1", "-slcmd", "1", "-slcnum", "4", "-scene", "1", "-denoise", "1", "-frameskip", "1"] if SCREEN
            else ["-slcmd", "0"] if GOM
            else ["-slcmd", "3", "-slcsize", "1200"] if DYNSLICE
            else ["-slcmd", "2", "-slcmbnum", "2040" if FULL else "900"])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr[-2000:]
    if hip and os.environ.get("WELS_HIP_TRACE") == "2":       # where a picture's time went, per session
        sys.stderr.write("".join(l + "\n" for l in p.stderr.decode(errors="replace").splitlines() if "per picture" in l))
    if hip and os.environ.get("WELSHIP_TRACE_DEVICES"):        # which device every backend was opened on (bench.py config4_layer_per_gpu)
        sys.stderr.write("".join(l + "\n" for l in p.stderr.decode(errors="replace").splitlines() if "backend for device" in l))
    if hip and os.environ.get("WELSHIP_FRAME_STATS"):
        sys.stderr.write("".join(l + "\n" for l in p.stderr.decode(errors="replace").splitlines() if l.startswith("welship:")))
    enc_fps = [float(x) for x in re.findall(rb" fps=([0-9.]+)", p.stdout)]
    wall = float(re.search(rb"wall_seconds=([0-9.]+)", p.stdout).group(1))
    pictures = p.stderr.count(b"welship hooks: did")
    host_pictures = p.stderr.count(b"left to the host")      # layers below WELS_HIP_MIN_LAYER_MBS macroblocks of a multi-layer session (pfHipLayerOnDevice)
    if DYNSLICE and hip:
        done = [l for l in p.stderr.decode(errors="replace").splitlines() if "picture complete" in l]
        run.dyn = {"slices": sum(int(l.split("complete:")[1].split()[0]) for l in done), "device_calls": sum(int(l.split("slices,")[1].split()[0]) for l in done)}
    sha = [hashlib.sha1(open("%s.%d" % (out, i), "rb").read()).hexdigest() for i in range(N)]
    return {"wall_s_incl_init": wall, "sum_of_session_encode_fps": sum(enc_fps), "min_session_fps": min(enc_fps), "max_session_fps": max(enc_fps),
            "device_pictures": pictures, "host_pictures": host_pictures}, sha


def main():
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "clip.yuv")
        clip = "Adobe_PDF_sample_a_1024x768_50Frms.264" if SCREEN else "VID_%dx%d_cavlc_temporal_direct.264" % (W, H)
        subprocess.check_call([os.path.join(REF, "ref_dec"), os.path.join(REF, "res", clip), yuv], stdout=subprocess.DEVNULL)
        nfr = os.path.getsize(yuv) // (W * H * 3 // 2)
        c_leg, c_sha = run(tmp, yuv, False)
        h_leg, h_sha = run(tmp, yuv, True)
        print(json.dumps({"config": "%d concurrent sessions, %dx%d, %d frames each (clip has %d), RC bitrate mode %s Mbps per layer, %s, one process, one thread per session" % (N, W, H, FRAMES, nfr, "4" if FULL and not SIMULCAST else "1.5", ("4 simulcast AVC layers (%s) of 4 slices" % ("1080p/720p/360p/180p" if FULL else "1080p/540p/270p/135p")) if SIMULCAST else "screen content, 4 slices" if SCREEN else "one slice (GOM-level QP, WELS_HIP_GOM=%s)" % os.environ.get("WELS_HIP_GOM", "unset") if GOM else "size-limited slices of 1200 bytes (WELS_HIP_DYNSLICE=1): %s" % getattr(run, "dyn", None) if DYNSLICE else "raster slices of %d MBs" % (2040 if FULL else 900)),
                          "reference_c_path": c_leg, "hooks_on_device": h_leg, "same_bitstreams": c_sha == h_sha, "lib": os.path.basename(LIB)}))


if __name__ == "__main__":
    main()
