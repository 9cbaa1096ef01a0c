#!/usr/bin/env python3
"""Per-kernel averages out of a rocprofv3 results db: python tools/kernel_avg.py <dir> [name-substring ...]"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
                   "group by s.kernel_name order by 4 desc").fetchall()
want = sys.argv[2:]
for name, n, avg, tot in rows:
    if want and not any(w in name for w in want):
        continue
    print(f"{name[:80]:80s} n={n:4d} avg_us={avg:10.1f} total_us={tot:10.1f}")
