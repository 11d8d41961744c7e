"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(hip_lib):
    names = set()
    for h in ("welship.h", "welship_leaf.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names |= set(re.findall(r"\b(WelsHip[A-Za-z0-9]+)\s*\(", hdr))
    assert len(names) >= 30 + 85
    lib = C.CDLL(hip_lib)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_is_loud(hip_lib):
    from conftest import has_gpu
    if has_gpu():
        return
    lib = C.CDLL(hip_lib)
    out = (C.c_int32 * 1)()
    buf = (C.c_uint8 * 1024)()
    off = (C.c_int32 * 1)(0)
    rc = lib.WelsHipPrimSampleSad(0, 1, buf, C.c_size_t(1024), 32, off, buf, C.c_size_t(1024), 32, off, out)
    assert rc == 100      # WELSHIP_ERR_NO_DEVICE


def test_device_code_avoids_ashr_pk(tmp_path):
    """hipcc (ROCm 7.2) mis-uses gfx950's v_ashr_pk_u8_i32 (see wh_clip255 in csrc/kernels/wave.h): the device code of the
    macroblock kernels must not contain it."""
    import shutil
    import subprocess
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not available")
    for unit in ("hip_backend", "prims", "downsample"):       # (leaf.hip is the second half of prims.hip's translation unit)
        out = tmp_path / (unit + ".s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-function",
                               "-Wno-unused-variable", "-Wno-unused-command-line-argument", "-o", str(out),
                               os.path.join(ROOT, "openh264_amd", "csrc", "hip", unit + ".hip")])
        assert "v_ashr_pk" not in out.read_text()


def test_leaf_installer_declines_without_a_device(hip_lib, tmp_path):
    """WELS_HIP_LEAVES=1 (integration/welship_hooks.cpp InstallLeaves) on a machine without an MI355X: nothing is installed, the
    installer says why, the session runs on the reference's C functions.  (With a device: tests/test_leaf_gpu.py.)"""
    import subprocess
    from conftest import has_gpu
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_enc_hip")
    if has_gpu() or not os.path.exists(exe):
        pytest.skip("needs the hooked reference build and no GPU")
    from openh264_amd.utils.synth import synth_sequence
    src = str(tmp_path / "c.yuv")
    open(src, "wb").write(synth_sequence(64, 48, 2))
    p = subprocess.run([exe, "-i", src, "-w", "64", "-h", "48", "-o", str(tmp_path / "o.264"), "-quiet", "-rc", "-1", "-qp", "30"],
                       env=dict(os.environ, WELSHIP_LIB=hip_lib, WELS_HIP_LEAVES="1", WELS_HIP_TRACE="1"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    err = p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "leaf functions not installed (no usable device)" in err, err[-1000:]
    lib = C.CDLL(hip_lib)
    assert lib.WelsHipLeafAvailable() == 100      # WELSHIP_ERR_NO_DEVICE


def test_frame_job_layout_and_versioning(tmp_path):
    """WelsHipFrameJob (include/welship.h, layer 2b): the ctypes mirror the frame-API tests drive the library with has the header's size and field
    offsets, and WELSHIP_FRAMEJOB_MIN_SIZE is the size of the layout up to pbRecordsPacked -- fields are only appended behind it (cbSize)."""
    import shutil
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_frame_api_retry import FrameJob
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    fields = [f[0] for f in FrameJob._fields_]
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main (void) {\n  printf ("%%zu %%u\\n", sizeof (WelsHipFrameJob), WELSHIP_FRAMEJOB_MIN_SIZE);\n%s  return 0;\n}\n'
                   % (os.path.join(ROOT, "include", "welship.h"), "".join('  printf ("%s %%zu\\n", offsetof (WelsHipFrameJob, %s));\n' % (f, f) for f in fields)))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    size, min_size = (int(x) for x in out[0].split())
    assert size == C.sizeof(FrameJob) and min_size <= size
    for line in out[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(FrameJob, name).offset == int(off), name
    assert FrameJob.cbSize.offset == 0
    assert FrameJob.pbRecordsPacked.offset + C.sizeof(C.c_void_p) == min_size        # the first versioned layout ends with pbRecordsPacked


def test_mode_decision_kernels_do_not_spill(hip_lib):
    """Every instantiation of the P kernel that a launch can select keeps its registers: no scratch memory, no spilled VGPR (the 16- and 14-wave
    builds have 128 VGPRs to live in; one innocent-looking change of a helper in round 6 pushed them to 150 and the launch from 7.2 to 8.3 ms).
    The deblocking, intra and pre-processing kernels likewise.  Read from the notes of the code object inside the built library."""
    import shutil
    import subprocess
    tool = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(tool) or not shutil.which("bash"):
        pytest.skip("ROCm LLVM tools not available")
    out = subprocess.check_output(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), hip_lib]).decode()
    rows = [l for l in out.splitlines() if ".name:" in l]
    assert len(rows) > 30
    seen = 0
    for l in rows:
        import re
        name = re.search(r"\.name:\s+(\S+)", l).group(1)
        m = re.search(r"\.private_segment_fixed_size:\s+(\d+)", l)        # (the script's name shortening swallows this field of some rows: the spill count is in all)
        scratch, spills = int(m.group(1)) if m else 0, int(re.search(r"\.vgpr_spill_count:\s+(\d+)", l).group(1))
        if any(k in name for k in ("k_inter_pool", "k_intra_slice", "k_deblock_slices", "k_deblock_pairs", "k_expand", "k_tile", "k_compact", "k_vaa", "k_bgd", "k_ds_")):
            seen += 1
            assert scratch == 0 and spills == 0, "%s: %d bytes of scratch, %d spilled VGPRs" % (name, scratch, spills)
    assert seen >= 20
