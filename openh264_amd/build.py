"""Build recipes (in-tree, no JIT cache): libwelship.so for gfx950, the oracle, the CPU test build."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openh264_amd", "csrc")
HOST_SRCS = [os.path.join(CSRC, "host", f) for f in ("encoder.cpp", "frame_api.cpp", "entropy_cavlc.cpp", "headers.cpp")]
# (csrc/hip/leaf.hip is the second half of prims.hip's translation unit: the kernels both layers launch are compiled once)
HIP_SRCS = [os.path.join(CSRC, "hip", "hip_backend.hip"), os.path.join(CSRC, "hip", "prims.hip"), os.path.join(CSRC, "hip", "downsample.hip")]
LIB = os.path.join(ROOT, "openh264_amd", "libwelship.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libwelship_emu.so")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        for base, _, files in (os.walk(d) if os.path.isdir(d) else [(os.path.dirname(d), [], [os.path.basename(d)])]):
            for f in files:
                if os.path.getmtime(os.path.join(base, f)) > t:
                    return True
    return False


class _locked:
    """One builder at a time per output file (pytest -n runs the session fixtures of several workers at once): an exclusive lock on <out>.lock;
    the compiler writes <out>.tmp.<pid>, which is renamed over <out> only when it is complete -- nobody ever maps a half-written library."""
    def __init__(self, out):
        self.out = out

    def __enter__(self):
        import fcntl
        self.f = open(self.out + ".lock", "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)
        return self.out + ".tmp.%d" % os.getpid()

    def __exit__(self, et, ev, tb):
        import fcntl
        tmp = self.out + ".tmp.%d" % os.getpid()
        try:
            if et is None and os.path.exists(tmp):
                os.replace(tmp, self.out)
            elif os.path.exists(tmp):
                os.remove(tmp)
        finally:
            fcntl.flock(self.f, fcntl.LOCK_UN)
            self.f.close()
        return False


def build_hip(force=False, verbose=True, defines=(), tag="", flags=()):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).
    `defines` + `tag`: a second library (libwelship_<tag>.so) with candidate code paths switched on, for A/B runs on the
    device through WELSHIP_LIB / tools/fuzz_parity.py --lib; the product library is always the plain build.
    `flags`: further compiler flags of such a candidate (e.g. "-mllvm", "-amdgpu-enable-max-ilp-scheduling-strategy")."""
    out = LIB if not tag else LIB.replace(".so", "_" + tag + ".so")
    deps = [CSRC, os.path.join(ROOT, "include"), os.path.abspath(__file__)]        # (this file: the source lists)
    if not force and not _newer(out, deps):
        return out
    # NB: no v_ashr_pk_u8_i32 may appear in the device code (see wh_clip255 in csrc/kernels/wave.h): tests/test_abi.py checks the listing.
    with _locked(out) as tmp:
        if not force and not _newer(out, deps):          # (another process built it while this one waited for the lock)
            return out
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] + ["-D" + d for d in defines] + list(flags) + ["-fPIC", "-shared", "-Wno-unused-function",
               "-Wno-unused-variable", "-o", tmp] + HIP_SRCS + HOST_SRCS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=ROOT)
    return out


def build_emu(force=False, verbose=False, defines=(), tag=""):
    """g++ -DWH_EMU test build of the same kernel sources (tests only, never shipped/loaded by the product).
    `defines` + `tag`: a second test build with candidate code paths switched on (e.g. WH_DB_PER_EDGE)."""
    out = EMU_LIB if not tag else EMU_LIB.replace(".so", "_" + tag + ".so")
    deps = [CSRC, os.path.join(ROOT, "tests", "emu", "emu_backend.cpp"), os.path.abspath(__file__)]
    if not force and not _newer(out, deps):
        return out
    with _locked(out) as tmp:
        if not force and not _newer(out, deps):
            return out
        cmd = ["g++", "-O2", "-std=c++17", "-DWH_EMU"] + ["-D" + d for d in defines] + ["-fPIC", "-shared", "-Wno-unused-function", "-Wno-unused-variable",
               "-o", tmp, os.path.join(ROOT, "tests", "emu", "emu_backend.cpp")] + HOST_SRCS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=ROOT)
    return out


def build_oracle(verbose=True):
    """oracle/Makefile: C restatement always; oracle/_ref only where /root/reference exists."""
    cmd = ["make", "-C", os.path.join(ROOT, "oracle"), "-j8", "all"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)


if __name__ == "__main__":
    build_oracle()
    build_hip(force="--force" in sys.argv)
    build_emu(force="--force" in sys.argv)
