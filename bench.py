#!/usr/bin/env python3
"""bench.py — simplex consensus throughput on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank per GPU via
torch.distributed.run (backend nccl = RCCL).  One "step" = one pass of the consensus hot path
(raw BAM records resident in HBM → consensus BAM records in HBM) over one batch of synthetic
`simulate grouped-reads`-shaped families.  Rank 0 prints ONE JSON line.

Workload at N=1: BASELINE.json configs[1] — simplex, 5 M families, depth 8 (pairs), 150 bp paired.
Weak scaling: every rank processes its own contiguous shard of the family stream (families are
independent → no data-path collective); the only collective is the gather of the per-rank
consensus payload sizes + stats (what a writer needs to concatenate shards in input order).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(n_families, family_size, read_length, threads, duplex=False, codec=False):
    """Bounded sample of the same workload through the ORACLE (C++ restatement of the reference CPU
    caller; `--threads`-style batches of 50 MI groups, one caller object per batch) on this box's
    host cores.  A reported baseline, not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    extra = dict(insert_mean=350, insert_sd=60, codec=1) if codec else {}
    g = simulate_grouped_reads(n_families, family_size=family_size, read_length=read_length, duplex=int(duplex), **extra)
    o = fgx_opts.defaults(min_reads=1, kind=2 if codec else 1 if duplex else 0)
    if codec:
        o.overlapping_consensus = 0
    if duplex:
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 1
    best, res = None, None
    for _ in range(3):
        t0 = time.perf_counter()
        res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=1000 if codec else 100 if duplex else 50, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return dict(value=g.n_rec / best, unit="raw reads/s", cores=threads, kind="port",
                sample=f"{n_families} families x {family_size} pairs x {read_length}bp{' (--duplex)' if duplex else ' (CODEC pairs, insert N(350,60))' if codec else ''}, compute-only (records in RAM → "
                       f"ConsensusOutput bytes), batches of {1000 if codec else 100 if duplex else 50} MI groups over {threads} threads, best of 3",
                consensus_reads_per_s=res["count"] / best)


def pmc_traffic(families, depth, read_length):
    """HBM bytes of one k_family_wave launch from the committed rocprofv3 PMC passes of THIS workload (collected by
    tools/profile_round.sh in separate --pmc runs, summarised by tools/pmc_parse.py): 2 x FETCH_SIZE (the gfx950
    correction for wide coalesced reads, MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KiB.  WRITE_SIZE tallies a
    full request granule per partial-line store, so it over-states the kernel's many small descriptor stores; k_emit's
    streaming stores calibrate it at 0.99 of the true byte count.  None when no profile matches the workload."""
    if (depth, read_length) != (8, 150):
        return None
    import glob
    tag = f"{families // 1000000}M" if families % 1000000 == 0 else str(families)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{tag}_families.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["k_family_wave"]
        return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
    except (KeyError, ValueError):
        return None


def pmc_valu_busy(families, depth, read_length, kernel_ms):
    """Fraction of the launch during which the vector ALUs of a SIMD were issuing (SQ_ACTIVE_INST_VALU counts quad-cycles,
    summed over the 1024 SIMDs of the chip at 2.4 GHz) — the bound that actually binds this kernel.  From the same committed
    PMC passes as `traffic`; None when no profile matches the workload."""
    if (depth, read_length) != (8, 150) or kernel_ms <= 0:
        return None
    import glob
    tag = f"{families // 1000000}M" if families % 1000000 == 0 else str(families)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{tag}_families.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["k_family_wave"]
        return min(1.0, k["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / 2.4e9 / (kernel_ms * 1e-3))
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--caller", choices=["simplex", "duplex", "codec"], default="simplex",
                    help="simplex = BASELINE configs[1] (the headline metric); duplex = configs[2] shape (2M molecules, 6+6 pairs); "
                         "codec = configs[4] shape (1M molecules, 4 pairs of 2x300bp)")
    ap.add_argument("--families", type=int, default=None)
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--read-length", type=int, default=None)
    ap.add_argument("--depth-max", type=int, default=0,
                    help="simplex only: long-tail family sizes in [depth, depth-max] pairs, count ~ size^-1.5 (BASELINE configs[3] shape: --depth 2 --depth-max 50)")
    ap.add_argument("--cpu-sample-families", type=int, default=300000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reassemble", choices=["none", "root"], default="none",
                    help="root: every step also gathers the shard payloads to rank 0 in rank (= input) order over RCCL, inside the timed region "
                         "(the north star's reassembly step; off by default: shard outputs are already in input order, see DESIGN.md 6)")
    args = ap.parse_args()
    duplex, codec = args.caller == "duplex", args.caller == "codec"
    if args.families is None:
        args.families = int(os.environ.get("FGX_BENCH_FAMILIES", "1000000" if codec else "2000000" if duplex else "5000000"))
    if args.depth is None:
        args.depth = 4 if codec else 12 if duplex else 8
    if args.read_length is None:
        args.read_length = 300 if codec else 150

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")

    from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, VanillaUmiConsensusCaller,
                           VanillaUmiConsensusOptions)

    fam = args.families
    sim_extra = dict(family_size_max=args.depth_max) if (args.depth_max and args.caller == "simplex") else {}
    if codec:
        caller = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True, cell_tag="CB"), device=local_rank)
        sim_extra = dict(insert_mean=350, insert_sd=60, codec=1)
    elif duplex:
        caller = DuplexConsensusCaller("", "A", [1], cell_tag="CB", overlapping_consensus=True, device=local_rank)
    else:
        caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"),
                                           overlapping_consensus=True, device=local_rank)
    # synthetic families generated straight into HBM; rank r owns molecules [r*fam, (r+1)*fam)
    dg = caller.simulate_on_device(fam, family_size=args.depth, read_length=args.read_length, first_family=rank * fam, duplex=int(duplex), **sim_extra)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        out = caller.process_batch_device(dg)
    barrier()
    t0 = time.perf_counter()
    k_family_ms = k_emit_ms = k_total_ms = 0.0
    gathered_bytes = 0
    for _ in range(args.steps):
        out = caller.process_batch_device(dg)
        k_family_ms += caller.last_timing["k_family"]
        k_emit_ms += caller.last_timing["k_emit"]
        k_total_ms += caller.last_timing["kernels"]
        if args.reassemble == "root":
            from fgumi_amd.distributed import gather_payload_to_root
            whole = gather_payload_to_root(out.as_tensor(local_rank), root=0)
            if whole is not None:
                gathered_bytes = int(whole.numel())
            del whole
    barrier()
    dt = time.perf_counter() - t0
    from fgumi_amd.distributed import gather_sizes, max_over_ranks
    dt = max_over_ranks(dt, "cuda")                                                   # MAX over ranks
    per_rank = gather_sizes([out.data_len, out.count, dg.n_rec, out.n_deferred], "cuda")   # shard payload sizes, rank (= input) order
    total_bytes, total_cons, total_raw, total_def = [int(v) for v in per_rank.sum(0).tolist()]

    if rank == 0:
        L = args.read_length
        steps = args.steps
        # algorithmic bytes of ONE k_family launch on ONE GPU: per raw read ceil(L/2)+L read,
        # per consensus read 6*Lc written (bases, quals, depth i16, errors i16)
        alg_read = dg.n_rec * ((L + 1) // 2 + L)
        # duplex / CODEC: each record is built from two single-strand column sets (CODEC strands are about one read long)
        alg_write = out.count * 6 * L * (2 if (duplex or codec) else 1)
        k_avg_s = k_family_ms / steps / 1e3
        achieved = (alg_read + alg_write) / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        line = {
            "metric": ("CODEC consensus throughput, input raw reads/s (4 pairs x 2x300bp)" if codec
                       else "duplex consensus throughput, input raw reads/s (depth 6+6 x 150bp)" if duplex
                       else f"simplex consensus throughput, input raw reads/s (depth {args.depth}..{args.depth_max} long tail x {L}bp)" if args.depth_max
                       else f"simplex consensus throughput, input raw reads/s (depth-{args.depth} x {L}bp)"),
            "value": total_raw * steps / dt, "unit": "raw reads/s",
            "consensus_reads_per_s": total_cons * steps / dt,
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"CODEC consensus, {fam} molecules per GPU, {args.depth} pairs of 2x{L}bp, insert N(350,60) (BASELINE configs[4] shape), " if codec else
                                    f"duplex consensus, {fam} molecules per GPU, {args.depth} pairs split over /A and /B, {L}bp paired (BASELINE configs[2] shape), "
                                    if duplex else f"simplex consensus, {fam} families per GPU, depth {args.depth}..{args.depth_max} pairs (long tail), {L}bp paired (BASELINE configs[3] shape), "
                                    if args.depth_max else f"simplex consensus, {fam} families per GPU, depth={args.depth} pairs, {L}bp paired (BASELINE configs[1] shape), ")
                                   + "device-resident: raw BAM records in HBM -> consensus BAM records in HBM",
                       "min_reads": 1, "overlapping_consensus": True, "families_per_gpu": fam, "raw_reads_per_gpu": dg.n_rec,
                       "deferred_families": total_def, "output_bytes": total_bytes, "reassemble": args.reassemble,
                       "reassembled_bytes_on_rank0": gathered_bytes,
                       "columns_needing_call_full_per_step": caller.last_timing.get("full_columns")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if (duplex or codec or args.depth_max) else pmc_traffic(fam, args.depth, L), "kernel": "k_family", "kernel_ms": k_family_ms / steps, "k_emit_ms": k_emit_ms / steps,
                         "device_ms_per_step": k_total_ms / steps, "algorithmic_bytes_per_launch": alg_read + alg_write,
                         "read_only_GBs": alg_read / k_avg_s / 1e9 if k_avg_s > 0 else 0.0,
                         "valu_busy_frac": None if (duplex or codec or args.depth_max) else pmc_valu_busy(fam, args.depth, L, k_family_ms / steps)},
        }
        if not args.no_cpu_baseline and world == 1 and not args.depth_max:
            line["cpu_baseline"] = cpu_baseline(min(fam, args.cpu_sample_families), args.depth, L, os.cpu_count() or 1, duplex, codec)
        print(json.dumps(line))
    if rank == 0:   # profiling builds (-DFGX_PHASE_TIMING=1) expose per-phase cycle totals of k_family_wave
        import ctypes
        from fgumi_amd import lib
        if hasattr(lib, "fgx_debug_phase_cycles"):
            ph = (ctypes.c_uint64 * 16)()
            lib.fgx_debug_phase_cycles(ph, 1)
            tot = float(sum(ph)) or 1.0
            names = ["-", "stage", "parse", "overlap", "geometry", "gates", "columns", "umi", "descriptors"]
            print("phase share: " + "  ".join(f"{names[i]}={100.0 * ph[i] / tot:.1f}%" for i in range(1, 9)), file=sys.stderr)
            bn = ["raw->lds", "parse", "unpack", "overlap", "geometry", "gates", "columns(to umi)"]
            tb = float(sum(ph[9:16])) or 1.0
            print("k_family (workgroup) share: " + "  ".join(f"{bn[i - 9]}={100.0 * ph[i] / tb:.1f}%" for i in range(9, 16)), file=sys.stderr)
    caller.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
