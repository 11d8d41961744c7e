"""The diagnostic tools must keep working: the CAVLC syntax dumper parses what the encoder writes."""
import io
import os
import sys

import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_h264_parse_reads_our_streams(emu_lib):
    import h264_parse
    w, h, n = 176, 144, 4
    bs, _ = oh.encode_sequence(synth_sequence(w, h, n), w, h, lib_path=emu_lib, iDLayerQp=26, uiIntraPeriod=0, fMaxFrameRate=30.0,
                               iTargetBitrate=5000000, uiSliceMode=1, uiSliceNum=3, iComplexityMode=1)
    out = io.StringIO()
    h264_parse.parse_stream(bs, out=out, levels=True)
    text = out.getvalue()
    assert text.count("SLICE pic") == 3 * n
    mbs = [ln for ln in text.splitlines() if ln.startswith("  mb ")]
    assert len(mbs) == 99 * n                     # every macroblock of every picture was parsed (skips included)
    assert any("P16x16" in ln or "P_Skip" in ln for ln in mbs) and any("I4x4" in ln or "I16x16" in ln for ln in mbs)


def test_committed_bench_line_keeps_the_contract():
    """profiles/r02_bench_default.json is the output of `python bench.py` on the MI355X: the fields the driver and the judge read
    must all be there (metric/unit from BASELINE.json, roofline and cpu_baseline objects, the self-verification verdicts)."""
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["unit"] == "frames/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "u8" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"] == "k_inter_pool" and r["avg_launch_ms"] * r["launches_per_step"] <= line["ms_per_step"]
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["threads4"]["cores"] == 4
    n = line["config"]["pictures_in_flight_per_gpu"]
    assert abs(line["value"] - n * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) / line["value"] < 1e-6
    # what the run checked about itself
    assert line["verified"] is True and line["e2e"]["bitstream_vs_reference"]["match"] and line["e2e_overlapped"]["bitstream_vs_reference"]["match"]
    assert line["res_clip"]["verified"] is True and line["intra_720p"]["verified"] is True
    assert set(line["latency"]) == {"sessions_1", "sessions_8"}


def test_host_entropy_bench_runs(emu_lib):
    """tools/host_entropy_bench.py times the host share of a session group's frame step (entropy coding straight from the packed
    records) on the CPU test build: it must keep running and keep producing the same bytes for the same input."""
    import subprocess
    out = []
    for _ in range(2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_entropy_bench.py"), "--frames", "3", "--sessions", "2", "--width", "320", "--height", "192",
                            "--lib", emu_lib], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        text = p.stdout.decode()
        assert "host finish:" in text and "WelsHipGroupHostStats" in text
        out.append([l for l in text.splitlines() if l.startswith("sha1 of all bitstreams")][0])
    assert out[0] == out[1]


def test_intra4x4_predictor_table_is_the_generated_one():
    """kernels/intra_mb.h kWhI4Desc (the nine Intra4x4 predictors as a table look-up) equals what tools/gen_tables.py derives from the
    predictors' index arithmetic -- which the generator checks against the predictors themselves on random edge samples."""
    import re
    import gen_tables
    src = open(os.path.join(ROOT, "openh264_amd", "csrc", "kernels", "intra_mb.h")).read()
    body = src.split("kWhI4Desc[36] = {", 1)[1].split("};", 1)[0]
    words = [int(w, 16) for w in re.findall(r"0x([0-9a-fA-F]{8})u", body)]
    assert words == gen_tables.i4_table(check=300)


def test_every_repo_file_a_test_names_exists():
    """Round 4 lost eleven GPU tests to a clean-up commit that deleted a helper script a `gpu`-marked test still shelled out to; nothing on the
    CPU tier noticed.  Every tracked-tree path the test sources name -- os.path.join(ROOT, "tools", "x.py"), "profiles/..." literals, modules
    imported from tools/ -- must exist (built artefacts under oracle/_ref, tests/emu and gpurun_out are made by build(), not tracked)."""
    import ast
    import glob
    import re
    missing = []
    built = ("oracle/_ref", "gpurun_out", "tests/emu/", "build/")
    tool_modules = {os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, "tools")) if f.endswith(".py")}
    top = {"tools", "oracle", "profiles", "integration", "include", "tests", "openh264_amd"}
    for src in sorted(glob.glob(os.path.join(ROOT, "tests", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        tree = ast.parse(open(src).read())
        inserts_tools = 'os.path.join(ROOT, "tools")' in open(src).read()
        for node in ast.walk(tree):
            rel = None
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "join" and len(node.args) >= 2 \
                    and isinstance(node.args[0], ast.Name) and node.args[0].id == "ROOT" and all(isinstance(a, ast.Constant) and isinstance(a.value, str) for a in node.args[1:]):
                rel = "/".join(a.value for a in node.args[1:])
            elif isinstance(node, ast.Constant) and isinstance(node.value, str) and re.fullmatch(r"(tools|oracle|profiles|integration|include|tests/golden)/[\w./-]+\.\w+", node.value):
                rel = node.value
            elif isinstance(node, ast.Import) and inserts_tools:
                for a in node.names:                      # `import h264_parse` after sys.path.insert(tools)
                    if a.name not in sys.stdlib_module_names and a.name not in sys.modules and "." not in a.name and a.name not in tool_modules \
                            and not os.path.exists(os.path.join(ROOT, "tests", a.name + ".py")) and a.name not in ("pytest", "numpy", "torch", "openh264_amd", "conftest"):
                        missing.append("%s imports %s (not in tools/)" % (os.path.basename(src), a.name))
            if rel and "*" not in rel and rel.split("/")[0] in top and not rel.startswith(built) and not os.path.exists(os.path.join(ROOT, rel)):
                missing.append("%s names %s" % (os.path.basename(src), rel))
    assert not missing, missing
