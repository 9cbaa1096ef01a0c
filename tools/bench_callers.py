#!/usr/bin/env python3
"""Throughput of the duplex (BASELINE configs[2] shape) and CODEC (configs[4] shape) callers through the C ABI,
next to the oracle on the host cores.  These are the general paths: host orchestration + every likelihood column on
the device; input and output are host buffers, so the numbers include staging and PCIe.  One JSON line per caller.

usage: python tools/bench_callers.py [--families N] [--steps K] [--callers duplex,codec]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--callers", default="duplex,codec")
    args = ap.parse_args()
    import fgx_opts
    import orc
    from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, simulate_grouped_reads

    threads = os.cpu_count() or 1
    for name in args.callers.split(","):
        if name == "duplex":
            g = simulate_grouped_reads(args.families, family_size=12, duplex=1)
            caller = DuplexConsensusCaller("", "A", [1], cell_tag="CB", overlapping_consensus=True)
            o = fgx_opts.defaults(kind=1)
            o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 1
            batch, shape = 100, f"{args.families} molecules x (6+6) pairs x 150bp, --duplex"
        else:
            g = simulate_grouped_reads(args.families, family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1)
            caller = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True, cell_tag="CB"))
            o = fgx_opts.defaults(kind=2, overlapping_consensus=0)
            batch, shape = 1000, f"{args.families} molecules x 4 pairs x 2x300bp, insert N(350,60)"
        out = caller.process_batch(g)     # warm-up
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = caller.process_batch(g)
        dt = (time.perf_counter() - t0) / args.steps
        tm = dict(caller.last_timing) if hasattr(caller, "last_timing") else {}
        best = None
        for _ in range(2):
            t1 = time.perf_counter()
            ref = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch, threads=threads)
            d = time.perf_counter() - t1
            best = d if best is None else min(best, d)
        same = ref["data"] == out.data
        print(json.dumps({
            "caller": name, "workload": shape, "raw_reads": g.n_rec, "consensus_reads": out.count,
            "value": g.n_rec / dt, "unit": "raw reads/s (host buffers in, host buffers out; general path)", "ms_per_step": dt * 1e3,
            "timing_ms": tm, "byte_identical_to_oracle": bool(same),
            "cpu_baseline": {"value": g.n_rec / best, "unit": "raw reads/s", "cores": threads, "kind": "port",
                             "sample": f"same input, batches of {batch} MI groups over {threads} threads, best of 2"}}))
        caller.close()


if __name__ == "__main__":
    main()
