#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) into a per-kernel table.
usage: python profiles/summarize_rocprof.py gpurun_out/prof/<name>_results.db > profiles/<round>_<what>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                  "max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'%':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'vgpr':>5s} {'lds':>6s}")
for r in rows:
    print(f"{r[0][:72]:72s} {r[1]:7d} {r[2]:10.2f} {100 * r[2] / tot:6.1f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f} {r[6] or 0:5d} {r[7] or 0:6d}")
