#!/usr/bin/env python3
"""bench.py — simplex consensus throughput on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank per GPU via
torch.distributed.run (RCCL).  One "step" = one pass of the consensus hot path over one batch of
synthetic `simulate`-shaped families resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(n_families, family_size, read_length, threads):
    """Bounded sample of the same workload through the ORACLE (C++ restatement of the reference CPU
    caller, `--threads`-style batches of 50 groups) on the host cores.  Reported, not optimised."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    g = simulate_grouped_reads(n_families, family_size=family_size, read_length=read_length)
    o = fgx_opts.defaults(min_reads=1)
    orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50, threads=threads)  # warm-up
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return dict(value=g.n_rec / best, unit="raw reads/s", cores=threads, kind="port",
                sample=f"{n_families} families x {family_size} pairs x {read_length}bp (compute-only, records in RAM, "
                       f"batches of 50 groups, best of 2)", consensus_reads_per_s=res["count"] / best)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--families", type=int, default=int(os.environ.get("FGX_BENCH_FAMILIES", "200000")))
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--read-length", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads

    # weak scaling: every rank gets its own contiguous shard of the family stream (no data-path collective)
    fam_per_rank = args.families
    g = simulate_grouped_reads(fam_per_rank, family_size=args.depth, read_length=args.read_length, first_family=rank * fam_per_rank)
    caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"),
                                       overlapping_consensus=True, device=local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = caller.process_batch(g)
    barrier()
    t0 = time.perf_counter()
    kern_ms = 0.0
    for _ in range(args.steps):
        out = caller.process_batch(g)
        kern_ms += caller.last_timing["kernels"]
    barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        raw_reads = g.n_rec * world * args.steps
        cons_reads = out.count * world * args.steps
        L = args.read_length
        alg_bytes_per_launch = g.n_rec * ((L + 1) // 2 + L) + out.count * 6 * L
        k_avg_s = kern_ms / args.steps / 1e3
        achieved = alg_bytes_per_launch / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        line = {
            "metric": "simplex consensus, input raw reads/s (depth-8 x 150bp)", "value": raw_reads / dt, "unit": "raw reads/s",
            "consensus_reads_per_s": cons_reads / dt, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"simplex consensus, {fam_per_rank} families/GPU, depth={args.depth}, {L}bp paired, "
                                   f"general host-orchestrated path", "min_reads": 1, "overlapping_consensus": True},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_column_jobs", "kernel_ms": kern_ms / args.steps},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(min(fam_per_rank, 20000), args.depth, L, os.cpu_count() or 1)
        print(json.dumps(line))
    caller.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
