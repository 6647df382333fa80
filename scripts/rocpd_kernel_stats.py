#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into the `--stats` table: per kernel calls / total / avg / min / max.
usage: rocpd_kernel_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info('top_kernels')")]
rows = None
try:
    rows = list(cur.execute("select * from top_kernels"))
except Exception:
    pass
q = """select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start),
              max(d.workgroup_size_x), max(d.grid_size_x), max(d.group_segment_size)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
res = list(cur.execute(q))
tot = sum(r[2] for r in res) or 1
hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "WorkgroupSize", "MaxGridSize", "MaxLDSBytes"]
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(hdr)
for name, n, t, a, mn, mx, wg, grid, lds in res:
    out.writerow([name, n, t, "%.1f" % a, mn, mx, "%.2f" % (100.0 * t / tot), wg, grid, lds])
