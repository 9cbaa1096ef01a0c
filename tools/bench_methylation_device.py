#!/usr/bin/env python3
"""The methylation-aware mode in the device-resident pipeline (round 4: the streaming kernels of simplex_deep.inc): simulate-shaped families
resident in HBM, a random genome under their coordinates (contig 0; the simulator places molecule m at 1000 + 1000 m), EM-Seq mode, through
`fgx_process_batch_device`.  Prints one JSON line: raw reads/s with the mode, with the mode off (the record / column split pipeline on the
same batch) and through the host entry's general path on a sample (what the mode cost before).

  python tools/bench_methylation_device.py [--families 1000000] [--depth 8] [--steps 5]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from fgumi_amd import MethylationMode, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, lib, simulate_grouped_reads  # noqa: E402


def timed(c, dg, steps):
    c.process_batch_device(dg)                     # warm-up (allocations, images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = c.process_batch_device(dg)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=1000000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import ctypes as C
    lib.fgx_debug_last_meth_device.restype = C.c_uint32
    lib.fgx_debug_last_meth_device.argtypes = [C.c_void_p]
    g = simulate_grouped_reads(a.families, family_size=a.depth)
    dg = g.to_device()
    n_reads = int(g.n_rec)
    rng = np.random.default_rng(7)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=1000 + a.families * 1000 + 2000, dtype=np.uint8)].tobytes()
    line = {"workload": f"{a.families} families x {a.depth} pairs x 150 bp, device-resident, EM-Seq mode, {len(genome) >> 20} MiB genome in HBM"}
    for name, mode in (("em_seq", MethylationMode.EmSeq), ("mode_off", None)):
        kw = dict(min_reads=1, min_consensus_base_quality=2, cell_tag="CB")
        if mode is not None:
            kw["methylation_mode"] = mode
        c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(**kw), overlapping_consensus=True)
        if mode is not None:
            c.set_reference({"chr1": genome}, ["chr1"])
        dt, out = timed(c, dg, a.steps)
        line[name] = {"ms_per_step": round(dt * 1e3, 2), "raw_reads_per_s": round(n_reads / dt), "consensus_records": int(out.count), "output_bytes": int(out.data_len),
                      "deferred_families": int(out.n_deferred), "families_on_the_streaming_kernels": int(lib.fgx_debug_last_meth_device(c._h))}
        c.close()
    print(json.dumps(line))


if __name__ == "__main__":
    main()
