#!/usr/bin/env python3
"""Throughput of the device MI grouping (fgx_group_records_device) over a simulated record stream resident in HBM.
usage: python tools/bench_grouping.py [families] [depth]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fgumi_amd import VanillaUmiConsensusCaller  # noqa: E402

fam = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
c = VanillaUmiConsensusCaller("", "A")
dg = c.simulate_on_device(fam, family_size=depth)
rg = c.group_records_device(dg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rg = c.group_records_device(dg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(json.dumps(dict(families=fam, records=dg.n_rec, groups=rg.n_grp, ms=dt * 1e3, records_per_s=dg.n_rec / dt, blob_GBs=dg.blob_len / dt / 1e9)))
c.close()
