"""Summarise rocprofv3 --pmc counter_collection CSVs (one per pass) into {kernel: {counter: mean per launch}}.
usage: python tools/pmc_parse.py gpurun_out/<dir> [batches per profiled run] > profiles/<name>.json
With the number of batches a profiled run processed (steps + warmup of bench.py) every kernel also gets `_launches_per_step`
(the split pipeline launches its first stage chunk by chunk: per step = mean per launch x launches per step)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    per = defaultdict(lambda: defaultdict(float))   # (dispatch id, kernel) -> counter -> sum over dimensions
    for row in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"][:40]
        per[(row["Dispatch_Id"], k)][row["Counter_Name"]] += float(row["Counter_Value"])
    for (_, k), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if k.startswith("k_")}
if len(sys.argv) > 2:
    for k in out:
        out[k]["_launches_per_step"] = max(len(v) for v in acc[k].values()) / float(sys.argv[2])
json.dump(out, sys.stdout, indent=1, sort_keys=True)
