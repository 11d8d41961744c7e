"""Where a kernel waits for a vector-memory load right after issuing it (a latency chain the source did not intend: a load inside a branch whose
VALUE is merged through a phi, a select between loaded values folded into a load at a selected address, ...).
usage: isa_load_waits.py <listing.s made with -gline-tables-only> <kernel symbol substring> [max instructions between load and wait, default 12]"""
import re, sys
f, key = sys.argv[1], sys.argv[2]
near = int(sys.argv[3]) if len(sys.argv) > 3 else 12
lines = open(f, errors="replace").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l)
loc, last_load, n_since = "", None, 0
for i in range(start + 1, len(lines)):
    l = lines[i]
    if l.startswith(".Lfunc_end"): break
    if ".loc" in l:
        m = re.search(r";\s*(.*)$", l); loc = m.group(1) if m else loc
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"): continue
    op = t.split()[0]
    if op.startswith(("global_load", "buffer_load", "flat_load", "global_atomic")) and "lds" not in op:
        last_load, n_since = (i - start, t, loc), 0
        continue
    n_since += 1
    if op == "s_waitcnt" and "vmcnt(0)" in t and last_load and n_since <= near:
        print("+%d %s\n      waited %d instructions later  <%s>" % (last_load[0], last_load[1][:60], n_since, last_load[2][:160]))
        last_load = None
