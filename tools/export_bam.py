#!/usr/bin/env python3
"""Writes the simulated grouped input, and (on a GPU box) the engine's consensus output, as real BGZF BAM files — the two
files `fgumi compare bams` needs to pin whole-BAM parity against a reference build (see tools/ref_pin.sh).

  python tools/export_bam.py --families 100000 --depth 8 --out-dir /tmp/pin            # grouped.bam (+ ours.bam with a GPU)
  python tools/export_bam.py --caller duplex --families 20000 --depth 12 --out-dir /tmp/pin_duplex

grouped.bam  the `simulate grouped-reads`-shaped records of csrc/simgen.h, header `@HD SO:unsorted GO:query
             SS:template-coordinate` + 24 @SQ lines (the simulator places molecules on 24 references)
ours.bam     the engine's records behind `create_unmapped_consensus_header` (consensus_runner.rs:130-173)
Both are level-1 BGZF like the reference's own output; the record bytes inside are exactly what the C ABI takes / returns."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REFS = [(f"chr{i + 1}", 2147483647) for i in range(24)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--caller", choices=["simplex", "duplex", "codec"], default="simplex")
    ap.add_argument("--families", type=int, default=100000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--depth-max", type=int, default=0)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--min-reads", type=int, default=1)
    ap.add_argument("--out-dir", default=".")
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--input-only", action="store_true", help="write grouped.bam only (no GPU needed)")
    a = ap.parse_args()
    from fgumi_amd import bgzf, simulate_grouped_reads
    os.makedirs(a.out_dir, exist_ok=True)
    kw = {}
    if a.depth_max:
        kw["family_size_max"] = a.depth_max
    if a.caller == "duplex":
        kw["duplex"] = 1
    if a.caller == "codec":
        kw.update(insert_mean=350, insert_sd=60, codec=1)
    g = simulate_grouped_reads(a.families, family_size=a.depth, read_length=a.read_length, **kw)
    t0 = time.perf_counter()
    gin = os.path.join(a.out_dir, "grouped.bam")
    size_in = bgzf.write_bam(gin, bgzf.grouped_input_header(REFS), REFS, g.blob, threads=a.threads)
    rep = dict(grouped_bam=gin, grouped_records=int(g.n_rec), grouped_bytes=size_in, write_s=time.perf_counter() - t0)
    have_gpu = False
    if not a.input_only:
        try:
            import torch
            have_gpu = torch.cuda.is_available()
        except ImportError:
            pass
    if have_gpu:
        from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, VanillaUmiConsensusCaller,
                               VanillaUmiConsensusOptions)
        if a.caller == "simplex":
            c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=a.min_reads, min_consensus_base_quality=2), overlapping_consensus=True)
            cmd = f"fgumi simplex -i grouped.bam -o ours.bam --min-reads {a.min_reads}"
        elif a.caller == "duplex":
            c = DuplexConsensusCaller("", "A", [a.min_reads], overlapping_consensus=True)
            cmd = f"fgumi duplex -i grouped.bam -o ours.bam --min-reads {a.min_reads}"
        else:
            c = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True))
            cmd = "fgumi codec -i grouped.bam -o ours.bam"
        out = c.process_batch(g)
        c.close()
        ours = os.path.join(a.out_dir, "ours.bam")
        size_out = bgzf.write_bam(ours, bgzf.consensus_header("A", "Read group", 0, cmd), [], out.data, threads=a.threads)
        rep.update(ours_bam=ours, consensus_records=int(out.count), ours_bytes=size_out)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
