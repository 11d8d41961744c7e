"""Static instruction counts of one kernel by the phase of the P macroblock body they belong to.
usage: isa_by_phase.py <listing.s made with -gline-tables-only> <kernel symbol substring> [--lines]
Every instruction is attributed to the most recent source line of the body function itself (inter_mb.h wh_inter_mb_body_t) or of the kernel
(hip_backend.hip) that a .loc named before it: inlined helpers count for the statement that called them (as far as the scheduler kept them
together).  Loops count once (static); weigh them with the trip counts of tools/mb_stats.py."""
import re, sys, collections
f, key = sys.argv[1], sys.argv[2]
lines = open(f, errors="replace").read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = m.group(3)
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
body = {v: k for k, v in files.items()}
f_inter, f_back = body.get("inter_mb.h"), 0
# the body function's own line range
src = open("openh264_amd/csrc/kernels/inter_mb.h").read().split("\n")
b0 = next(i for i, l in enumerate(src) if "WH_FN void wh_inter_mb_body_t" in l) + 1
b1 = next(i for i, l in enumerate(src) if l.startswith("WH_FN void wh_inter_mb_body (")) + 1
phases = []   # (first line, name) from the WH_PROF_MARK comments
for i in range(b0, b1):
    m = re.search(r"WH_PROF_MARK \(P, M, (\d+)\);\s*//\s*(.*)", src[i - 1])
    if m: phases.append((i, m.group(2).strip()))
def phase_of(fl, ln):
    if fl == f_back: return "kernel: claim / wait / release (hip_backend.hip)"
    prev = "body entry"
    name = None
    for first, nm in phases:
        if ln <= first: return nm
    return "store (tail)"
cnt = collections.defaultdict(lambda: collections.Counter())
anchor = (f_back, 0)
kinds = lambda op: ("valu" if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")) else
                    "lanexfer" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_cbranch", "s_branch", "s_load", "s_nop", "s_endpgm", "s_barrier")) else
                    "branch" if op.startswith(("s_cbranch", "s_branch")) else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "smem" if op.startswith("s_load") else
                    "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
per_line = collections.defaultdict(collections.Counter)
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"): break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        fl, ln = int(m.group(1)), int(m.group(2))
        if ln and ((fl == f_inter and b0 <= ln < b1) or fl == f_back): anchor = (fl, ln)
        continue
    m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
    if not m or l.lstrip().startswith((";", ".")): continue
    k = kinds(m.group(1))
    cnt[phase_of(*anchor)][k] += 1
    per_line[anchor][k] += 1
    if "_dpp" in l: cnt[phase_of(*anchor)]["(dpp)"] += 1
tot = collections.Counter()
print("%-62s %6s %6s %6s %5s %5s %5s %5s" % ("phase", "valu", "salu", "branch", "lds", "vmem", "xfer", "wait"))
order = ["kernel: claim / wait / release (hip_backend.hip)"] + [n for _, n in phases] + ["store (tail)"]
for n in order:
    c = cnt.get(n)
    if not c: continue
    print("%-62s %6d %6d %6d %5d %5d %5d %5d" % (n[:62], c["valu"], c["salu"], c["branch"], c["lds"], c["vmem"], c["lanexfer"], c["wait"]))
    tot.update(c)
print("%-62s %6d %6d %6d %5d %5d %5d %5d" % ("total (static)", tot["valu"], tot["salu"], tot["branch"], tot["lds"], tot["vmem"], tot["lanexfer"], tot["wait"]))
if "--lines" in sys.argv:
    for (fl, ln), c in sorted(per_line.items()):
        if sum(c.values()) >= 25: print("%s:%d  valu %d salu %d lds %d  | %s" % (files.get(fl, "hip_backend.hip"), ln, c["valu"], c["salu"], c["lds"], (src[ln - 1].strip()[:110] if fl == f_inter else "")))
