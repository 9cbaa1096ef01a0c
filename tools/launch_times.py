#!/usr/bin/env python3
"""Per-launch durations of the kernels whose name contains a substring, in launch order, out of a rocprofv3 kernel trace csv:
python tools/launch_times.py <dir> <file prefix> <substring> [last N]"""
import csv
import glob
import sys

d, pre, sub = sys.argv[1], sys.argv[2], sys.argv[3]
last = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for f in glob.glob(d + "/**/" + pre + "_kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    if last:
        rows = rows[-last:]
    print("  ", sub, "launches (ms):", " ".join("%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows))
