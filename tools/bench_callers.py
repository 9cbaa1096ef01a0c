#!/usr/bin/env python3
"""Throughput of the GENERAL (host-orchestrated) paths of the duplex and CODEC callers through the C ABI: host buffers in,
host buffers out, `fgx_set_general_only`.  These paths are the fallback for molecules the device pipelines defer; the
device-resident numbers (and the CPU baseline) come from `bench.py --caller duplex|codec`.  One JSON line per caller.

usage: python tools/bench_callers.py [--families N] [--steps K] [--callers duplex,codec]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=50000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--callers", default="duplex,codec")
    args = ap.parse_args()
    from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, simulate_grouped_reads

    for name in args.callers.split(","):
        if name == "duplex":
            g = simulate_grouped_reads(args.families, family_size=12, duplex=1)
            caller = DuplexConsensusCaller("", "A", [1], cell_tag="CB", overlapping_consensus=True)
            shape = f"{args.families} molecules x (6+6) pairs x 150bp, --duplex"
        else:
            g = simulate_grouped_reads(args.families, family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1)
            caller = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True, cell_tag="CB"))
            shape = f"{args.families} molecules x 4 pairs x 2x300bp, insert N(350,60)"
        res = {}
        for mode in ("hybrid", "general"):
            caller.set_general_only(mode == "general")
            out = caller.process_batch(g)     # warm-up
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = caller.process_batch(g)
            dt = (time.perf_counter() - t0) / args.steps
            res[mode] = {"raw_reads_per_s": g.n_rec / dt, "ms_per_step": dt * 1e3, "timing_ms": dict(caller.last_timing), "bytes": len(out.data)}
        print(json.dumps({"caller": name, "workload": shape, "raw_reads": g.n_rec, "consensus_reads": out.count,
                          "unit": "raw reads/s, host buffers in and out (Python mirror included)", **res,
                          "same_bytes_both_paths": res["hybrid"]["bytes"] == res["general"]["bytes"]}))
        caller.close()


if __name__ == "__main__":
    main()
