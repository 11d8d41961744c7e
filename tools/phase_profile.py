"""In-kernel phase profile of the P-frame MB kernel (cycle counters, see WH_PROF_MARK).
usage: phase_profile.py [sessions] [synthetic|res] [intra]   (res: the reference's 1080p clip from oracle/_ref/res, as in bench.py; intra: the IDR step)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the cycle accumulators are only compiled into the profiling build (-DWH_PROF, kernels/wave.h): openh264_amd/libwelship_prof.so, built here
# (`python -c "from openh264_amd import build as B; B.build_hip(defines=('WH_PROF',), tag='prof')"`) so that it travels to the GPU box
if "WELSHIP_LIB" not in os.environ:
    _prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openh264_amd", "libwelship_prof.so")
    if not os.path.exists(_prof):
        from openh264_amd import build as _B
        _B.build_hip(defines=("WH_PROF",), tag="prof")
    os.environ["WELSHIP_LIB"] = _prof
import openh264_amd as oh
from openh264_amd.utils.synth import synth_sequence
w, h, S = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 16
e = oh.Encoder(); p = e.GetDefaultParams(); e.close()
p.iPicWidth, p.iPicHeight, p.iDLayerQp, p.fMaxFrameRate, p.iTargetBitrate, p.uiIntraPeriod, p.uiSliceMode, p.uiSliceNum = w, h, 24, 30.0, 5000000, 0, 1, 4
R = 8
g = oh.EncoderGroup(p, S, ring_slots=R)
fsz = w * h * 3 // 2
if len(sys.argv) > 2 and sys.argv[2] == "res":
    import bench
    clip = bench.decode_res_clip("VID_1920x1080_cavlc_temporal_direct.264")
    c = bench.Content(clip, fsz, R, True)
else:
    import bench
    c = bench.Content(synth_sequence(w, h, 2 * R), fsz, R, False)
for s in range(S):
    for k in range(R): g.upload(s, k, c.frame(s, k))
lib = g._lib
lib.WelsHipGroupProfile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
if len(sys.argv) > 3 and sys.argv[3] == "intra":      # the IDR step that starts every stream: the I macroblock body's phases
    lib.WelsHipGroupProfile(g._h, 1, None)
    out = (C.c_ulonglong * 64)()
    one = g.bench(1, 0)
    lib.WelsHipGroupProfile(g._h, 1, out)
    print("IDR step %s" % (one,))
    names = {11: "ticket+order", 12: "dependency wait", 9: "tile load", 2: "I16x16 mode costs", 3: "texture analysis + mode cache", 4: "sixteen I4x4 blocks", 5: "I16x16 encode", 6: "chroma decision + encode",
             15: "decision epilogue", 7: "bits + store", 13: "release+flag", 14: "(body total)"}
    if out[16 + 0]:       # a -DWH_PROF_I4 library: the pair steps' parts are counted on their own, "sixteen I4x4 blocks" keeps the rest (single-block steps, loop)
        names.update({0: "I4x4 pair: tables + candidates + costs", 1: "I4x4 pair: decision trees", 8: "I4x4 pair: transform .. reconstruction"})
    tot = sum(out[i] for i in range(16) if i not in (14,))
    for i, n in names.items():
        print("%-30s %6.2f%%  avg %8.0f cycles  hits %d" % (n, 100.0 * out[i] / max(tot, 1), out[i] / max(out[16 + i], 1), out[16 + i]))
    print("total cycles/MB %.0f" % (tot / max(out[16 + 7], 1)))
    sys.exit(0)
g.bench(1, 0)
lib.WelsHipGroupProfile(g._h, 1, None)
out = (C.c_ulonglong * 64)()
one = g.bench(1, 0)
lib.WelsHipGroupProfile(g._h, 1, out)           # one step: the wave lifetimes of a single mode-decision launch
if out[46]:
    print("one step %s: MD waves alive %.1f%% of the launch's span (first start to last end %.3f ms, %d waves, mean lifetime %.3f ms)"
          % (one, 100.0 * out[45] / (out[46] * max(out[44], 1)), out[44] / 1e5, out[46], out[45] / out[46] / 1e5))
print(g.bench(3, 0))
lib.WelsHipGroupProfile(g._h, 1, out)
if os.environ.get("WELSHIP_PROF_DETAIL"):      # library built with -DWH_PROF_DETAIL (wave.h): sub-phases instead of the search .. store phases
    names = {0: "claim: slots+ticket+order", 2: "claim: cold inputs issued", 11: "claim: windows issued (+body callback)", 12: "dependency wait",
             3: "batch-1: loads issued", 9: "batch-1: wait + commit", 10: "args, nb cache, mvp+window wait", 4: "pskip: luma prediction", 5: "pskip: chroma prediction",
             6: "pskip: SADs + decision", 1: "pskip: DCT / quant test + rest", 7: "search .. store (merged)", 13: "release+flag", 14: "(body total)", 15: "window adopted"}
else:
  names = {11: "ticket+order", 12: "dependency wait", 8: "args+job+slice", 9: "batch-1 loads", 10: "nb cache+ctx", 0: "mvp+window loads", 1: "pskip test", 2: "p16x16 ME", 3: "i16 test",
         4: "fine partitions", 5: "refine+chromaMC", 6: "residual", 7: "store", 13: "release+flag", 14: "(body total)", 15: "window adopted"}
tot = sum(out[i] for i in range(16) if i not in (14,))
for i, n in names.items():
    print("%-18s %6.2f%%  avg %8.0f cycles  hits %d" % (n, 100.0 * out[i] / max(tot, 1), out[i] / max(out[16 + i], 1), out[16 + i]))
print("total cycles/MB %.0f" % (tot / max(out[16 + 7], 1)))

db = {0: "ticket+order", 1: "request own inputs", 2: "wait neighbours (in WG)", 3: "wait neighbours (seam)", 4: "own inputs landed", 5: "strips+tile", 6: "boundary strengths",
      7: "edge filters", 8: "write-back+exchange", 9: "publish+flag", 10: "idle tail (per wave)", 11: "(body total)"}
d = out[32:]
tot = sum(d[i] for i in range(16) if i not in (11,))
print("deblocking kernel:")
for i, n in db.items():
    print("%-26s %6.2f%%  avg %8.0f cycles  hits %d" % (n, 100.0 * d[i] / max(tot, 1), d[i] / max(d[16 + i], 1), d[16 + i]))
print("total wave cycles per MB %.0f" % (tot / max(d[16 + 9], 1)))
