#!/usr/bin/env python
"""Why did `share_of_8.configs1_share` read 8.7 ms in one bench line and 3.8 ms in another (same build, same chain: 16 launches, 6 syncs)?
The share step is launch/latency bound, so it is the first number to move when something outside the kernels changes.  This times the same
625 000-family step (20 steps after 2 warm-up, as bench.py does) at several points of what a default bench.py run does before it:
fresh, after the file -> file leg, after idle seconds, after the CPU legs.  Prints one line per point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions

caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True, device=0)


def share(tag, fam=625000, steps=20, warm=2):
    dg = caller.simulate_on_device(fam, read_length=150, first_family=0, family_size=8)
    for _ in range(warm):
        caller.process_batch_device(dg)
    torch.cuda.synchronize()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        caller.process_batch_device(dg)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    k = caller.last_timing
    print(f"{tag:34s} mean {sum(ts)/len(ts):6.2f} ms  min {min(ts):6.2f}  max {max(ts):6.2f}  first3 {[round(t, 2) for t in ts[:3]]}  kernels {k['kernels']:.2f} stage {k['k_family']:.2f} emit {k['k_emit']:.2f}", flush=True)
    del dg
    torch.cuda.empty_cache()


big = caller.simulate_on_device(5000000, read_length=150, first_family=0, family_size=8)
for _ in range(3):
    caller.process_batch_device(big)
torch.cuda.synchronize()
del big
torch.cuda.empty_cache()
share("after the 5 M steps")
share("again")
e = bench.end_to_end(caller, 1000000, 8, 150, "/tmp/fgx_probe_e2e")
print("file -> file", round(e["value"] / 1e6, 1), "M reads/s", flush=True)
share("after file -> file")
share("again")
time.sleep(20)
share("after 20 s idle")
share("again")
c = bench.cpu_baseline(320000, 8, 150, min(os.cpu_count(), bench.cgroup_cpu_quota() or 1 << 30))
print("cpu leg", round(c["value"] / 1e6, 2), "M reads/s", flush=True)
share("after the CPU leg")
share("again")
share("200 steps", steps=200)
