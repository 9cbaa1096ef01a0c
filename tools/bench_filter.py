#!/usr/bin/env python3
"""Throughput of the device consensus-read filter (fgx_filter_last_output_device) on the records a device-resident consensus
batch leaves in HBM.  Each step re-runs the consensus call untimed (the filter masks in place), then times the filter alone.
usage: python tools/bench_filter.py [--caller simplex|duplex|codec] [--families N] [--steps K]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, ConsensusFilter, DuplexConsensusCaller, FilterConfig,  # noqa: E402
                       VanillaUmiConsensusCaller, VanillaUmiConsensusOptions)

ap = argparse.ArgumentParser()
ap.add_argument("--caller", default="simplex")
ap.add_argument("--families", type=int, default=2000000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--rejects", type=int, default=0)
a = ap.parse_args()
if a.caller == "simplex":
    c = VanillaUmiConsensusCaller("c", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2))
    dg = c.simulate_on_device(a.families, family_size=2, family_size_max=12)
    cfg = FilterConfig.new([3], [0.025], [0.1], min_base_quality=20)
elif a.caller == "duplex":
    c = DuplexConsensusCaller("d", "A", [1])
    dg = c.simulate_on_device(a.families, family_size=4, family_size_max=16, duplex=1)
    cfg = FilterConfig.new([6, 3, 2], [0.025], [0.1], min_base_quality=20)
else:
    c = CodecConsensusCaller("x", "A", CodecConsensusOptions(produce_per_base_tags=True))
    dg = c.simulate_on_device(a.families, family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1)
    cfg = FilterConfig.new([2, 1, 1], [0.025], [0.1], min_base_quality=20)
f = ConsensusFilter.on_caller(c, cfg, track_rejects=bool(a.rejects))
times = []
for step in range(a.steps + 1):
    out = c.process_batch_device(dg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = f.filter_last_output_device()
    torch.cuda.synchronize()
    if step:
        times.append(time.perf_counter() - t0)
dt = sum(times) / len(times)
print(json.dumps(dict(caller=a.caller, families=a.families, consensus_records=r.records_count, record_bytes=out.data_len, passed=r.passed_count,
                      bases_masked=r.bases_masked, kept_bytes=r.data_len, ms=dt * 1e3, records_per_s=r.records_count / dt,
                      GBs_in_plus_out=(out.data_len + r.data_len + r.rejects_len) / dt / 1e9)))
try:
    import ctypes as C
    from fgumi_amd._lib import load
    L = load()
    ph = (C.c_uint64 * 8)()
    if L.fgx_debug_filter_phase_cycles(ph, 0) == 0:
        tot = sum(ph) or 1
        print("k_filter_records phase shares (stage, walk, decode, sweep, reduce+read-level):", [round(x / tot, 3) for x in list(ph)[:5]], file=sys.stderr)
except AttributeError:
    pass
c.close()
