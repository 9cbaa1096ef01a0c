#!/usr/bin/env python3
"""PCIe-inclusive host entry (fgx_process_batch: host buffers in, host buffers out).
usage: python tools/bench_host_entry.py [families] [pinned]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads  # noqa: E402
from fgumi_amd._lib import Output, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = simulate_grouped_reads(n, family_size=8)
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"
if pinned:      # the integrator's staging buffers are pinned (north star): hipMemcpyAsync is then a true DMA
    import torch
    keep = [torch.from_numpy(a).pin_memory() for a in (g.blob, g.rec_off.view("int64"), g.rec_len.view("int32"), g.grp_first.view("int32"))]
    g.blob, g.rec_off, g.rec_len, g.grp_first = keep[0].numpy(), keep[1].numpy().view("uint64"), keep[2].numpy().view("uint32"), keep[3].numpy().view("uint32")
c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)


def call():
    out = Output()
    t0 = time.perf_counter()
    rc = lib.fgx_process_batch(c._h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data,
                               g.n_grp, C.byref(out))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.fgx_last_error(c._h)
    return dt, out


res = {}
for mode in ("host_entry",):
    call()
    ts = [call() for _ in range(3)]
    dt = min(t for t, _ in ts)
    o = ts[-1][1]
    res[mode] = dict(s=dt, reads_per_s=g.n_rec / dt, out_bytes=int(o.data_len), kernels_ms=o.ms_kernels, d2h_ms=o.ms_d2h)
print(json.dumps(dict(families=n, pinned_input=pinned, raw_reads=g.n_rec, in_bytes=int(g.blob.size), **res)))
c.close()
