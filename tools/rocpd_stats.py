#!/usr/bin/env python3
"""Kernel statistics (what `rocprofv3 --stats` prints) from a rocprofv3 rocpd SQLite database."""
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.display_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                       f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
                       f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.display_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS_bytes,Scratch_bytes"]
    for r in rows:
        out.append("\"%s\",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s" % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9]))
    return "\n".join(out)


if __name__ == "__main__":
    print(stats(sys.argv[1]))
