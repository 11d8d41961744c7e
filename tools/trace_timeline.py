"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run (csv): per engine busy time, copy/kernel overlap, and the last steps as text.
python tools/trace_timeline.py <trace dir> [steps to print]"""
import csv, glob, sys

d = sys.argv[1]
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:28], 0))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kind = "H2D" if "HOST_TO_DEVICE" in r.get("Direction", "") else "D2H" if "DEVICE_TO_HOST" in r.get("Direction", "") else "D2D"
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, kind, int(r.get("Bytes", r.get("Size", 0)) or 0)))
ev.sort()
if not ev:
    raise SystemExit("no events")
t0 = ev[0][0]


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


def inter(u, v):
    i = j = 0
    s = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if a < b:
            s += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return s


# steady state: from the third-last k_inter launch on
md = [e for e in ev if e[2] == "K" and "k_inter" in e[3]]
if len(md) >= nprint + 1:
    lo = md[-(nprint + 1)][0]
else:
    lo = t0
hi = ev[-1][1]
sel = [e for e in ev if e[1] > lo]
K = union([(max(e[0], lo), e[1]) for e in sel if e[2] == "K"])
H = union([(max(e[0], lo), e[1]) for e in sel if e[2] == "H2D"])
D = union([(max(e[0], lo), e[1]) for e in sel if e[2] == "D2H"])
span = hi - lo
print("window %.2f ms: kernels busy %.2f ms, H2D busy %.2f ms (%.1f MB), D2H busy %.2f ms (%.1f MB)" % (
    span / 1e6, length(K) / 1e6, length(H) / 1e6, sum(e[4] for e in sel if e[2] == "H2D") / 1e6, length(D) / 1e6, sum(e[4] for e in sel if e[2] == "D2H") / 1e6))
print("H2D under kernels %.2f ms, D2H under kernels %.2f ms, nothing running %.2f ms" % (inter(K, H) / 1e6, inter(K, D) / 1e6, (span - length(union(K + H + D))) / 1e6))
# one text lane per engine / kernel: busy share of every bucket (' ' idle, '.' < 25 %, '-' < 75 %, '#' more)
bucket = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 0.5e6
nb = int(span / bucket) + 1
lanes = {}
for e in sel:
    lanes.setdefault(e[3], []).append((max(e[0], lo), e[1]))
print("one column = %.2f ms" % (bucket / 1e6))
for name in sorted(lanes, key=lambda k: lanes[k][0][0]):
    u = union(lanes[name])
    if length(u) < 0.02 * bucket * 1 and len(lanes[name]) < 8:
        continue
    row = []
    for k in range(nb):
        a, b = lo + k * bucket, lo + (k + 1) * bucket
        f = inter(u, [[a, b]]) / bucket
        row.append(" " if f == 0 else "." if f < 0.25 else "-" if f < 0.75 else "#")
    print("%-28s|%s| %.2f ms, %d events" % (name, "".join(row), length(u) / 1e6, len(lanes[name])))
