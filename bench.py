#!/usr/bin/env python3
"""bench.py — simplex consensus throughput on MI355X (BASELINE.json metric).

`python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank per GPU via
torch.distributed.run (backend nccl = RCCL).  One "step" = one pass of the consensus hot path
(raw BAM records resident in HBM → consensus BAM records in HBM) over one batch of synthetic
`simulate grouped-reads`-shaped families.  Rank 0 prints ONE JSON line.

Workload at N=1: BASELINE.json configs[1] — simplex, 5 M families, depth 8 (pairs), 150 bp paired.

N>1 (families are independent → no collective on the data path):
  --scaling weak    (default) every rank processes its own 5 M-family shard of the family stream
  --scaling strong  ONE 5 M-family stream is cut into contiguous shards of equal record bytes
                    (`distributed.balanced_shards` over the per-family weights), one per rank
  --reassemble      N>1: `value` is the rate with the shard payloads left on their ranks (a writer per rank; output is
                    SO:unsorted in input order, so rank order IS file order).  `root` also times a second loop in which
                    every step ships the payloads to rank 0 in rank order over RCCL point-to-point — the north star's
                    single-writer reassembly — and reports it beside (`value_with_reassembly_on_root`): one root receiving
                    9 GB per rank and step is bound by its xGMI links, not by the kernels.  `auto` (default) = root beyond one rank.
                    The gather loops run LAST, after everything else of the line has been measured, under a watchdog: a rank that
                    fails or hangs in the gather costs `reassemble_error`, never the line.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F64_LANE_OPS_PER_S = 64 * 1024 * 2.4e9 / 4   # one f64 add per lane per 4 cycles, 1024 SIMDs at 2.4 GHz = 39.3 T/s
E2E_CHUNK_MB = 512              # compressed bytes per chunk of the file -> file leg (FGX_BENCH_E2E_CHUNK_MB; tools/e2e_chunk_sweep.py)
F64_OPS_PER_OBSERVATION = 8     # two Kahan chains (the base seen, any other base) x 4 dependent add/sub


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cgroup_cpu_quota():
    """CPUs' worth of time the container's cgroup grants (cpu.max = "<quota> <period>"), or None when unlimited / unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, -(-int(q) // int(p)))
    except (OSError, ValueError):
        return None


def cpu_baseline(n_families, family_size, read_length, threads, duplex=False, codec=False):
    """Bounded sample of the same workload through the ORACLE (C++ restatement of the reference CPU caller;
    `--threads`-style batches of MI groups, one caller object per batch, Phred tables cached process-wide like the
    reference's OnceLock caches) on this box's host cores: once on ONE thread, once on all of them.
    A reported baseline (kind "port"), not the optimisation target, and not `fgumi` itself."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    extra = dict(insert_mean=350, insert_sd=60, codec=1) if codec else {}
    o = fgx_opts.defaults(min_reads=1, kind=2 if codec else 1 if duplex else 0)
    if codec:
        o.overlapping_consensus = 0
    if duplex:
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 1
    bg = 1000 if codec else 100 if duplex else 50

    def run(nf, T, reps):
        # timed: the worker section inside the oracle (every batch's ConsensusOutput bytes ready: what the reference's Process step
        # hands to its writer) — NOT the harness's single-threaded join of the batches into one buffer, nor the copy into Python
        g = simulate_grouped_reads(nf, family_size=family_size, read_length=read_length, duplex=int(duplex), **extra)
        best, res = None, None
        for _ in range(reps):
            res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=bg, threads=T)
            dt = res["seconds_workers"]
            best = dt if best is None else min(best, dt)
        return g.n_rec / best, res["count"] / best, g.n_rec

    n1 = max(1000, min(n_families, 40000))
    v1, c1, r1 = run(n1, 1, 2)
    vall, call, rall = run(n_families, threads, 3)
    shape = f"{family_size} pairs x {read_length}bp" + (" (--duplex)" if duplex else " (CODEC pairs, insert N(350,60))" if codec else "")
    phys = physical_cores() or threads
    quota = cgroup_cpu_quota()
    if quota is not None:
        phys = min(phys, quota)                  # what the box lets this process use, whatever /proc/cpuinfo lists
    return dict(value=vall, unit="raw reads/s", cores=threads, kind="port", cpu_model=cpu_model(), cgroup_cpu_quota=quota,
                value_1_thread=v1, speedup_all_over_1=vall / v1 if v1 else None,
                # what perfect scaling of the one-thread figure over the box's physical cores would give: the number to hold the GPU
                # against when the measured multi-thread leg falls short of it (memory allocator, SMT, NUMA)
                physical_cores=phys, linear_bound=v1 * phys,
                consensus_reads_per_s=call, consensus_reads_per_s_1_thread=c1,
                sample=f"T={threads}: {n_families} families x {shape} = {rall} reads, best of 3; T=1: {n1} families = {r1} reads, best of 2; "
                       f"compute-only (records in RAM -> per-batch ConsensusOutput bytes), batches of {bg} MI groups pulled by the worker threads")


def cpu_end_to_end(families, depth, read_length, threads, directory):
    """The CPU side of the file -> file leg (BASELINE.md 3.2 / SURVEY 8d (b)): a level-1 BGZF grouped BAM -> consensus BAM through host code
    only — block-parallel zlib inflate and the library's level-1 deflate on `threads` cores, the record chain walk, and the ORACLE for MI
    grouping and the consensus caller (kind "port": the C++ restatement, batches of 50 MI groups pulled by `threads` workers).  The stages
    run one after another here; the reference's pipeline overlaps them, so `value` (records / sum of the stages) is a lower bound of a
    pipelined host path and `value_if_stages_overlapped` (records / the longest stage) an upper bound.  A bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fgx_opts
    import orc
    import numpy as np
    from fgumi_amd import bgzf
    os.makedirs(directory, exist_ok=True)
    gin, gout = os.path.join(directory, "cpu_grouped.bam"), os.path.join(directory, "cpu_consensus.bam")
    n_rec = write_grouped_bam(gin, families, depth, read_length)
    o = fgx_opts.defaults(min_reads=1)
    best = None
    for _ in range(2):
        st = {}
        t0 = time.perf_counter()
        with open(gin, "rb") as f:
            raw = f.read()
        st["read"] = time.perf_counter() - t0
        t = time.perf_counter()
        stream, own = bgzf.native_inflate(raw, threads)
        st["inflate"] = time.perf_counter() - t
        t = time.perf_counter()
        l_text = int.from_bytes(bytes(stream[4:8]), "little")
        p = 8 + l_text
        n_ref = int.from_bytes(bytes(stream[p:p + 4]), "little")
        p += 4
        for _r in range(n_ref):
            l_name = int.from_bytes(bytes(stream[p:p + 4]), "little")
            p += 8 + l_name
        rec_off, rec_len = bgzf.record_boundaries(stream, p)
        st["boundaries"] = time.perf_counter() - t
        t = time.perf_counter()
        k_off, k_len, grp = orc.group_records(stream, rec_off, rec_len)
        st["grouping"] = time.perf_counter() - t
        t = time.perf_counter()
        res = orc.process(o, stream, k_off, k_len, grp, batch_groups=50, threads=threads)
        st["consensus"] = time.perf_counter() - t
        t = time.perf_counter()
        hdr = bgzf.bam_header_bytes(bgzf.consensus_header("A", "Read group", 0, "fgumi simplex"), [])
        body = np.frombuffer(hdr + res["data"], dtype=np.uint8)
        comp, own2 = bgzf.native_deflate(body, 1, threads, with_eof=True)
        st["deflate"] = time.perf_counter() - t
        t = time.perf_counter()
        with open(gout, "wb") as f:
            f.write(memoryview(comp))
        st["write"] = time.perf_counter() - t
        wall = time.perf_counter() - t0
        count = int(res["count"])
        del stream, own, comp, own2, res, body
        if best is None or wall < best[0]:
            best = (wall, st, count)
    wall, st, count = best
    for pth in (gin, gout):
        try:
            os.remove(pth)
        except OSError:
            pass
    return dict(metric="BAM file in -> consensus BAM file out on the host cores (oracle + host BGZF), raw reads/s", value=n_rec / wall, unit="raw reads/s", kind="port",
                cores=threads, families=families, raw_reads=n_rec, total_s=wall, stage_s=st, value_if_stages_overlapped=n_rec / max(st.values()),
                consensus_records=count,
                note="stages run one after another (the reference's pipeline overlaps them): `value` is a lower, `value_if_stages_overlapped` an upper bound; best of two")


def write_grouped_bam(path, families, depth, read_length):
    """The grouped input BAM of the file -> file leg (level-1 BGZF, 125 000 families per slab); returns its record count."""
    from fgumi_amd import bgzf, simulate_grouped_reads
    refs = [(f"chr{i + 1}", 2147483647) for i in range(24)]
    n_rec, slab = 0, 125000
    with open(path, "wb") as f:
        for b in bgzf.bgzf_compress(bgzf.bam_header_bytes(bgzf.grouped_input_header(refs), refs), 1, None):
            f.write(b)
        for lo in range(0, families, slab):
            g = simulate_grouped_reads(min(slab, families - lo), family_size=depth, read_length=read_length, first_family=lo)
            n_rec += int(g.n_rec)
            nat = bgzf.native_deflate(g.blob, 1, 32, with_eof=False)
            f.write(memoryview(nat[0]))
            del g, nat
        f.write(bgzf.BGZF_EOF)
    return n_rec


def end_to_end(caller, families, depth, read_length, directory, chunk_mb=None, grouped=None):
    """BAM file in -> consensus BAM file out through the streaming pipeline (fgx_run_bam: BGZF inflate + boundaries + MI grouping + consensus + block
    CRCs on the device, level-1 deflate on the host cores, five overlapping stages, the next chunk uploading and inflating while the device stage works on this one) on a
    bounded file of the same workload: what a user of the command sees, next to the device-resident `value`.  Best of two runs, input file in
    the page cache.  `grouped` = (path, records) of an input written before (tools/e2e_chunk_sweep.py)."""
    from fgumi_amd import bgzf
    os.makedirs(directory, exist_ok=True)
    gin, gout = os.path.join(directory, "grouped.bam"), os.path.join(directory, "consensus.bam")
    if grouped:
        gin, n_rec = grouped
    else:
        n_rec = write_grouped_bam(gin, families, depth, read_length)
    if chunk_mb is None:
        chunk_mb = int(os.environ.get("FGX_BENCH_E2E_CHUNK_MB", str(E2E_CHUNK_MB)))
    best = None
    for _ in range(2):
        t = time.perf_counter()
        st = caller.run_bam(gin, gout, header_text=bgzf.consensus_header("A", "Read group", 0, "fgumi simplex"), chunk_raw_bytes=chunk_mb << 20)
        wall = time.perf_counter() - t
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    for pth in (gout,) if grouped else (gin, gout):
        try:
            os.remove(pth)
        except OSError:
            pass
    stages = {k: st["seconds_" + k] for k in ("read", "inflate", "device", "deflate", "write")}
    return dict(metric="BAM file in -> consensus BAM file out (fgx_run_bam), raw reads/s", value=n_rec / wall, unit="raw reads/s", families=families, raw_reads=n_rec,
                total_s=wall, chunks=int(st["chunks"]), chunk_mb=chunk_mb, stage_busy_s=stages, bottleneck=max(stages, key=stages.get),
                device_stage_s={k: st["seconds_" + k] for k in ("h2d", "device_inflate", "boundaries", "grouping", "consensus", "d2h")},
                input_bam_bytes=int(st["in_bytes"]), input_uncompressed_bytes=int(st["inflated_bytes"]), output_bam_bytes=int(st["out_file_bytes"]),
                consensus_records=int(st["consensus_records"]), deferred_groups=int(st["deferred_groups"]),
                # (every simulated family is a pair family above --min-reads: two consensus records each — a group cut in two at a chunk border would show here)
                consensus_records_expected=2 * families, records_as_expected=bool(int(st["consensus_records"]) == 2 * families),
                note="bounded sample of the same workload; stages of successive chunks overlap (total_s is below the sum of the busy times); host side = the cores the cgroup grants")


def pmc_profile(families, depth, read_length):
    """Counters of ONE launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS workload (separate --pmc
    runs by tools/profile_round.sh, summarised by tools/pmc_parse.py into profiles/*pmc_<N>M_families.json).  Returns
    (counters, file name) or (None, None): these numbers are read from a committed file, not measured in this run."""
    if (depth, read_length) != (8, 150):
        return None, None
    tag = f"{families // 1000000}M" if families % 1000000 == 0 else str(families)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{tag}_families.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except ValueError:
            continue
        for k in ("k_split_cols", "k_simplex_wave2", "k_family_wave"):
            if k in d:
                # the file holds means per launch and the launches per step (`_launches_per_step`, 1 when absent): per step = mean x launches
                n = float(d[k].get("_launches_per_step", 1.0))
                prof = {c: v * n for c, v in d[k].items() if not c.startswith("_")}
                stage = {}
                for kk in ("k_split_parse", "k_split_cols", "k_split_finish", "k_call_full"):
                    if kk in d and "SQ_INSTS_VALU" in d[kk]:
                        stage[kk] = d[kk]["SQ_INSTS_VALU"] * float(d[kk].get("_launches_per_step", 1.0))
                return dict(prof, kernel=k, stage_valu=stage), os.path.relpath(f, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--caller", choices=["simplex", "duplex", "codec"], default="simplex",
                    help="simplex = BASELINE configs[1] (the headline metric); duplex = configs[2] shape (2M molecules, 6+6 pairs); "
                         "codec = configs[4] shape (1M molecules, 4 pairs of 2x300bp)")
    ap.add_argument("--families", type=int, default=None, help="families per GPU (weak) or in total (strong)")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--read-length", type=int, default=None)
    ap.add_argument("--depth-max", type=int, default=0,
                    help="simplex only: long-tail family sizes in [depth, depth-max] pairs, count ~ size^-1.5 (BASELINE configs[3] shape: --depth 2 --depth-max 50)")
    ap.add_argument("--cpu-sample-families", type=int, default=320000, help="families of the multi-thread CPU leg (320000 x 16 = 5.12 M reads at depth 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--end-to-end-families", type=int, default=1000000, help="N=1, simplex: families of the file -> file leg (`end_to_end` in the line); 0 skips it")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-strong-block", action="store_true", help="weak runs also time the strong-scaling reading (`strong_scaling` in the line); this skips it")
    ap.add_argument("--no-share-block", action="store_true", help="N=1: skip the `share_of_8` block (one rank's share of an 8-way strong-scaling run, timed on this GPU)")
    ap.add_argument("--gather-timeout", type=float, default=240.0, help="N>1: seconds the gather loops (run last) may take before the line is printed without them")
    ap.add_argument("--reassemble", choices=["auto", "none", "root"], default="auto",
                    help="root: a second timed loop also gathers the shard payloads to rank 0 in rank (= input) order over RCCL, reported beside `value`; "
                         "auto = none")
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
    # (test hook, tests/test_gpu_distributed.py: FGX_BENCH_TEST_BACKEND=gloo runs the ranks of a multi-rank invocation on ONE GPU — RCCL refuses two
    # ranks on a device — with the collectives on host tensors; the driver never sets it)
    test_backend = os.environ.get("FGX_BENCH_TEST_BACKEND")
    if test_backend:
        local_rank = 0
    coll_dev = "cpu" if test_backend else "cuda"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(test_backend or "nccl")
    # `auto`: with more than one rank the single-writer gather (sizes all_gather + one batch_isend_irecv group into rank 0) is timed in
    # a SECOND loop and reported beside `value` (value_with_reassembly_on_root / gather_GBs_into_root), never instead of it; a failure of that
    # loop is reported in `reassemble_error` and costs nothing else of the line.  One rank: nothing to gather.
    reassemble = args.reassemble if args.reassemble != "auto" else ("root" if world > 1 else "none")

    from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, VanillaUmiConsensusCaller,
                           VanillaUmiConsensusOptions, simulated_family_bytes)
    from fgumi_amd.distributed import balanced_shards, gather_payload_to_root, gather_sizes, max_over_ranks, sum_over_ranks

    sim_extra = dict(family_size_max=args.depth_max) if (args.depth_max and args.caller == "simplex") else {}
    if codec:
        caller = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True, cell_tag="CB"), device=local_rank)
        sim_extra = dict(insert_mean=350, insert_sd=60, codec=1)
    elif duplex:
        caller = DuplexConsensusCaller("", "A", [1], cell_tag="CB", overlapping_consensus=True, device=local_rank)
    else:
        caller = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"),
                                           overlapping_consensus=True, device=local_rank)
    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def workload(scaling):
        """One workload (weak: `families` per GPU; strong: `families` in total, cut into contiguous shards of equal record bytes),
        generated straight into HBM."""
        if scaling == "strong" and world > 1:
            # one stream of `families` molecules, cut where the running record bytes reach k/world of the total
            w = simulated_family_bytes(args.families, family_size=args.depth, read_length=args.read_length, duplex=int(duplex), **sim_extra)
            lo, hi = balanced_shards(w, world)[rank]
            shard_bytes = int(w[lo:hi].sum())
        else:
            lo, hi = (rank * args.families, (rank + 1) * args.families) if scaling == "weak" else (0, args.families)
            shard_bytes = None
        fam = hi - lo
        dg = caller.simulate_on_device(fam, family_size=args.depth, read_length=args.read_length, first_family=lo, duplex=int(duplex), **sim_extra)
        if shard_bytes is None:
            shard_bytes = int(dg.blob_len)
        return dg, fam, shard_bytes

    def timed_loop(dg, n_steps, gather):
        k_family_ms = k_emit_ms = k_total_ms = 0.0
        gathered = 0
        out = None
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            out = caller.process_batch_device(dg)
            k_family_ms += caller.last_timing["k_family"]
            k_emit_ms += caller.last_timing["k_emit"]
            k_total_ms += caller.last_timing["kernels"]
            if gather:
                if os.environ.get("FGX_BENCH_TEST_GATHER_FAIL") == str(rank):      # (test hook: one rank fails inside the gather loop)
                    raise RuntimeError("FGX_BENCH_TEST_GATHER_FAIL")
                payload = out.as_tensor(local_rank)
                whole = gather_payload_to_root(payload.cpu() if test_backend else payload, root=0)
                if whole is not None:
                    gathered = int(whole.numel())
                del whole
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0, coll_dev)                            # MAX over ranks
        return dt, out, k_family_ms, k_emit_ms, k_total_ms, gathered

    def measure(scaling, steps, warmup):
        """`warmup` untimed + `steps` timed passes over one workload (barrier + synchronize on both sides, MAX over ranks)."""
        dg, fam, shard_bytes = workload(scaling)
        out = None
        for _ in range(warmup):
            out = caller.process_batch_device(dg)
        dt, out, k_family_ms, k_emit_ms, k_total_ms, _ = timed_loop(dg, steps, False)       # the K timed steps: `value`
        per_rank = gather_sizes([out.data_len, out.count, dg.n_rec, out.n_deferred, fam, shard_bytes,
                                 int(round(k_family_ms / steps * 1e3)), int(round(k_emit_ms / steps * 1e3))], coll_dev)   # rank (= input) order
        # the batch counters (ConsensusCallingStats / RejectionReason order, then the overlap CorrectionStats) summed over the ranks
        counters = sum_over_ranks(caller.last_stats_array, coll_dev)
        res = dict(dt=dt, out_count=int(out.count), n_rec=int(dg.n_rec), fam=fam, k_family_ms=k_family_ms, k_emit_ms=k_emit_ms, k_total_ms=k_total_ms,
                   dt_gather=None, gathered_bytes=0, gather_error=None, per_rank=per_rank, counters=counters, full_columns=caller.last_timing.get("full_columns"),
                   chain=last_chain())
        del dg, out
        torch.cuda.empty_cache()
        return res

    def last_chain():
        """Kernel launches / host synchronisations of the last batch's launch chain (fgx_debug_last_chain; None on a library without it)."""
        import ctypes
        from fgumi_amd import lib
        if not hasattr(lib, "fgx_debug_last_chain"):
            return None
        lib.fgx_debug_last_chain.restype = None
        lib.fgx_debug_last_chain.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        o = (ctypes.c_uint32 * 2)()
        lib.fgx_debug_last_chain(caller._h, o)
        return {"kernel_launches": int(o[0]), "host_syncs": int(o[1])}

    def share_of_8():
        """Strong-scaling readiness without an 8-GPU node: the SAME launch chain on what one rank of eight would get —
        configs[1] / 8 (625 000 depth-8 families) and the first of eight byte-balanced shards of the 5 M long-tail stream (configs[3] shape / 8) —
        with ms_per_step, kernel launches and host synchronisations per step, and the speedup eight such ranks would show over the whole
        stream on one GPU (`speedup_if_8` = ms of the whole stream / ms of the share).  N=1 only."""
        import numpy as np
        res = {}

        def run(fam, first, steps, **kw):
            dg = caller.simulate_on_device(fam, read_length=args.read_length, first_family=first, **kw)
            for _ in range(2):
                caller.process_batch_device(dg)
            torch.cuda.synchronize()
            out, per_step = None, []
            for _ in range(steps):               # one rank's step ends with its output on the host's side of a synchronize: time each step that way
                t0 = time.perf_counter()
                out = caller.process_batch_device(dg)
                torch.cuda.synchronize()
                per_step.append((time.perf_counter() - t0) * 1e3)
            # the median: a 4 ms step is short enough for one host hiccup (tools/share_probe.py saw a single 10 ms step in 200) to move a mean of 20
            ms = float(np.median(per_step))
            r = dict(families=int(fam), raw_reads=int(dg.n_rec), ms_per_step=ms, ms_per_step_mean=float(np.mean(per_step)), ms_per_step_max=float(np.max(per_step)), steps=steps,
                     value=dg.n_rec / ms * 1e3, deferred_families=int(out.n_deferred), **(last_chain() or {}))
            del dg, out
            torch.cuda.empty_cache()
            return r
        whole_ms = dt / args.steps * 1e3
        a = run(args.families // 8, 0, 20, family_size=args.depth)
        a["whole_stream_ms"] = whole_ms
        a["speedup_if_8"] = whole_ms / a["ms_per_step"]
        res["configs1_share"] = a
        # the long tail: the whole 5 M-family stream on this GPU, then rank 0's shard of its byte-balanced cut into eight
        lt = dict(family_size=2, family_size_max=50)
        w = simulated_family_bytes(args.families, read_length=args.read_length, **lt)
        lo, hi = balanced_shards(w, 8)[0]
        whole = run(args.families, 0, 3, **lt)
        b = run(hi - lo, lo, 10, **lt)
        b["whole_stream_ms"] = whole["ms_per_step"]
        b["whole_stream_value"] = whole["value"]
        b["whole_stream_chain"] = {k: whole.get(k) for k in ("kernel_launches", "host_syncs")}
        b["speedup_if_8"] = whole["ms_per_step"] / b["ms_per_step"]
        b["shard_bytes_share"] = float(np.asarray(w[lo:hi], dtype=np.float64).sum() / np.asarray(w, dtype=np.float64).sum())
        res["long_tail_share"] = b
        res["note"] = ("one rank's share of an 8-way strong-scaling run, measured on ONE GPU: ranks are independent (no data-path collective), so eight of them "
                       "finish in the time of the slowest share; what does not shrink with the batch — launches, host synchronisations, the tail of every kernel — shows here. "
                       "ms_per_step = the MEDIAN of the timed steps, each ended by a device synchronize (mean and max beside it)")
        return res

    M = measure(args.scaling, args.steps, args.warmup)
    dt, k_family_ms, k_emit_ms, k_total_ms = M["dt"], M["k_family_ms"], M["k_emit_ms"], M["k_total_ms"]
    dt_gather, gathered_bytes, per_rank, counters, fam = M["dt_gather"], M["gathered_bytes"], M["per_rank"], M["counters"], M["fam"]
    total_bytes, total_cons, total_raw, total_def = [int(v) for v in per_rank[:, :4].sum(0).tolist()]
    # The other reading of the metric, in the SAME invocation (the driver runs `bench.py --gpus N` once per N): north_star's target is
    # strong scaling — ONE stream of `families` molecules cut over the ranks.  The weak line's step IS the N=1 workload on every GPU
    # (`families` per GPU), so its step time is this run's own N=1 reference: speedup_vs_n1 = weak ms_per_step / strong ms_per_step.
    S = None
    if args.scaling == "weak" and not args.no_strong_block:
        S = measure("strong", max(1, min(args.steps, 10)), 1)

    if rank == 0:
        L = args.read_length
        steps = args.steps
        # algorithmic bytes of ONE k_family launch on ONE GPU: per raw read ceil(L/2)+L read,
        # per consensus read 6*Lc written (bases, quals, depth i16, errors i16)
        alg_read = M["n_rec"] * ((L + 1) // 2 + L)
        alg_write = M["out_count"] * 6 * L          # (per consensus read, once — also for duplex / CODEC records)
        k_avg_s = k_family_ms / steps / 1e3
        achieved = (alg_read + alg_write) / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        plain = not (duplex or codec or args.depth_max)
        pmc, pmc_file = pmc_profile(fam, args.depth, L) if plain else (None, None)
        # the arithmetic floor of the dominant kernel: every observation (read x position) costs two Kahan chains = 8 dependent
        # f64 add/sub on the vector ALUs (full rate: one per lane per 4 cycles)
        valu_floor_ms = M["n_rec"] * L * F64_OPS_PER_OBSERVATION / F64_LANE_OPS_PER_S * 1e3
        shape = (f"CODEC consensus, {args.depth} pairs of 2x{L}bp, insert N(350,60) (BASELINE configs[4] shape)" if codec else
                 f"duplex consensus, {args.depth} pairs split over /A and /B, {L}bp paired (BASELINE configs[2] shape)" if duplex else
                 f"simplex consensus, depth {args.depth}..{args.depth_max} pairs (long tail), {L}bp paired (BASELINE configs[3] shape)" if args.depth_max else
                 f"simplex consensus, depth={args.depth} pairs, {L}bp paired" + (" (BASELINE configs[1] shape)" if (args.depth, L) == (8, 150) else
                                                                                  " (BASELINE configs[0] shape)" if (args.depth, L) == (3, 150) else ""))
        sizing = (f"{args.families} families in total, cut into {world} shards of equal record bytes" if (args.scaling == "strong" and world > 1)
                  else f"{args.families} families per GPU")
        line = {
            "metric": ("CODEC consensus throughput, input raw reads/s (4 pairs x 2x300bp)" if codec
                       else "duplex consensus throughput, input raw reads/s (depth 6+6 x 150bp)" if duplex
                       else f"simplex consensus throughput, input raw reads/s (depth {args.depth}..{args.depth_max} long tail x {L}bp)" if args.depth_max
                       else f"simplex consensus throughput, input raw reads/s (depth-{args.depth} x {L}bp)"),
            "value": total_raw * steps / dt, "unit": "raw reads/s",
            "consensus_reads_per_s": total_cons * steps / dt,
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{shape}, {sizing}, device-resident: raw BAM records in HBM -> consensus BAM records in HBM",
                       "min_reads": 1, "overlapping_consensus": True, "families_total": int(per_rank[:, 4].sum()),
                       "families_per_rank": [int(v) for v in per_rank[:, 4].tolist()], "raw_reads_per_rank": [int(v) for v in per_rank[:, 2].tolist()],
                       "input_bytes_per_rank": [int(v) for v in per_rank[:, 5].tolist()], "output_bytes_per_rank": [int(v) for v in per_rank[:, 0].tolist()],
                       "deferred_families": total_def, "output_bytes": total_bytes,
                       "reassemble": ("none (payloads stay on their ranks: a writer per rank; --reassemble root times the single-writer gather beside)" if reassemble == "none"
                                      else "none in `value` (payloads stay on their ranks); gather to rank 0 timed beside"),
                       "reassembled_bytes_on_rank0": gathered_bytes, "reassemble_error": M.get("gather_error"),
                       "value_with_reassembly_on_root": (total_raw * steps / dt_gather) if dt_gather else None,
                       "ms_per_step_with_reassembly_on_root": (dt_gather / steps * 1e3) if dt_gather else None,
                       "gather_GBs_into_root": ((gathered_bytes - int(per_rank[0, 0])) * steps / max(dt_gather - dt, 1e-9) / 1e9) if (dt_gather and dt_gather > dt) else None,
                       "k_family_ms_per_rank": [v / 1e3 for v in per_rank[:, 6].tolist()], "k_emit_ms_per_rank": [v / 1e3 for v in per_rank[:, 7].tolist()],
                       "counters_all_ranks": {"total_reads": counters[0], "consensus_reads": counters[1], "filtered_reads": counters[2],
                                              "rejected_by_reason": counters[3:24], "overlap_correction": counters[24:28]},
                       "columns_needing_call_full_per_step": M["full_columns"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes of one launch: 2 x FETCH_SIZE (gfx950 correction for wide coalesced reads) + WRITE_SIZE, KiB → bytes,
                         # from the committed PMC passes (`traffic_source`), NOT measured in this run
                         "traffic": ((2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0) if (pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc) else None,
                         "traffic_source": pmc_file,
                         "kernel": ("k_split_cols" if plain else "k_family_wave" if (duplex or codec) else "k_split_cols"),
                         "kernel_stage": ("the family stage the HIP events bracket: k_split_parse + k_split_cols + k_split_finish | k_simplex_seg, then k_simplex_wave2 / "
                                          "k_family_wave for what is left, k_deep_parse + k_deep_cols for families of more than 64 records, k_family behind them, + k_call_full"
                                          if not (duplex or codec) else "k_family_wave + k_call_full"),
                         "kernel_ms": k_family_ms / steps, "k_emit_ms": k_emit_ms / steps,
                         "device_ms_per_step": k_total_ms / steps, "algorithmic_bytes_per_launch": alg_read + alg_write,
                         "read_only_GBs": alg_read / k_avg_s / 1e9 if k_avg_s > 0 else 0.0,
                         "read_only_frac": (alg_read / k_avg_s / 1e9 / HBM_PEAK_GBS) if k_avg_s > 0 else 0.0,   # north_star's target reads this one (>= 0.40)
                         # the bound that binds this kernel is vector-ALU issue, not HBM:
                         "f64_ops_per_observation": F64_OPS_PER_OBSERVATION,
                         "valu_floor_ms": valu_floor_ms,
                         "frac_of_valu_floor": valu_floor_ms / (k_family_ms / steps) if k_family_ms > 0 else None,
                         "valu_insts_per_family": (pmc["SQ_INSTS_VALU"] / fam) if (pmc and "SQ_INSTS_VALU" in pmc) else None,
                         "valu_insts_per_family_by_kernel": ({k: v / fam for k, v in pmc["stage_valu"].items()} if (pmc and pmc.get("stage_valu")) else None),
                         "valu_insts_per_family_stage": (sum(pmc["stage_valu"].values()) / fam if (pmc and pmc.get("stage_valu")) else None),
                         "salu_insts_per_family": (pmc["SQ_INSTS_SALU"] / fam) if (pmc and "SQ_INSTS_SALU" in pmc) else None,
                         # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs at 2.4 GHz (committed PMC file / this run's time)
                         "valu_busy_frac": (min(1.0, pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / 2.4e9 / k_avg_s)
                                            if (pmc and "SQ_ACTIVE_INST_VALU" in pmc and k_avg_s > 0) else None),
                         "valu_busy_frac_assumes": "4 cycles per vector wave-instruction (SQ_ACTIVE_INST_VALU counts quad-cycles); the measured mix of this kernel (tools/ubench/valu_rate.hip) is ~3 (adds) .. ~4.4 (shifts, bfe, compares, f64) cycles, the guide gives 2 for v_fma_f32: read as an estimate",
                         "pmc_kernel": pmc["kernel"] if pmc else None, "pmc_source": pmc_file},
        }
        if S is not None:
            s_raw = int(S["per_rank"][:, 2].sum())
            s_steps = max(1, min(args.steps, 10))
            s_ms = S["dt"] / s_steps * 1e3
            line["strong_scaling"] = {
                "workload": f"{args.families} families in TOTAL, cut into {world} contiguous shards of equal record bytes (distributed.balanced_shards)",
                "value": s_raw * s_steps / S["dt"], "unit": "raw reads/s", "steps": s_steps, "warmup": 1, "ms_per_step": s_ms,
                # this run's weak step is the N=1 workload on every GPU at once: the N=1 reference measured in the same process, same box
                "speedup_vs_n1": (dt / steps * 1e3) / s_ms if s_ms > 0 else None,
                "n1_reference": "ms_per_step of this line (weak: the same `families` per GPU = the N=1 workload)",
                "families_per_rank": [int(v) for v in S["per_rank"][:, 4].tolist()], "raw_reads_per_rank": [int(v) for v in S["per_rank"][:, 2].tolist()],
                "input_bytes_per_rank": [int(v) for v in S["per_rank"][:, 5].tolist()],
                "k_family_ms_per_rank": [v / 1e3 for v in S["per_rank"][:, 6].tolist()], "k_emit_ms_per_rank": [v / 1e3 for v in S["per_rank"][:, 7].tolist()],
                "deferred_families": int(S["per_rank"][:, 3].sum()),
                # N=1 self-check: the strong reading IS the weak workload there — same counters, value within 2 %
                "n1_self_check": None if world > 1 else {"counters_equal_the_weak_line": S["counters"] == counters,
                                                         "value_within_2pct_of_the_weak_line": abs((s_raw * s_steps / S["dt"]) / (total_raw * steps / dt) - 1.0) < 0.02},
                "value_with_reassembly_on_root": (s_raw * s_steps / S["dt_gather"]) if S["dt_gather"] else None,
            }
        # (the GPU legs back to back, before the host-side ones: the share step is latency bound and the first to feel a busy host)
        if world == 1 and plain and not args.no_share_block and not args.no_cpu_baseline and args.families >= 8:     # (profiling runs — --no-cpu-baseline — time the headline workload alone)
            try:
                line["share_of_8"] = share_of_8()
            except Exception as ex:
                line["share_of_8"] = {"error": str(ex)[:300]}
        if world == 1 and plain and args.end_to_end_families > 0 and not args.no_cpu_baseline:     # (the two extra legs go together: profiling runs switch both off)
            try:
                line["end_to_end"] = end_to_end(caller, min(args.end_to_end_families, fam), args.depth, L, os.environ.get("FGX_BENCH_TMP", "/tmp/fgx_bench_e2e"))
            except Exception as ex:                      # (a full /tmp or a read-only file system must not cost the line)
                line["end_to_end"] = {"error": str(ex)[:300]}
        if not args.no_cpu_baseline and world == 1 and not args.depth_max:
            # threads = the CPUs the container may really use (cgroup quota): oversubscribing a throttled cgroup only adds queueing
            T = min(os.cpu_count() or 1, cgroup_cpu_quota() or 1 << 30)
            line["cpu_baseline"] = cpu_baseline(min(fam, args.cpu_sample_families), args.depth, L, T, duplex, codec)
            if plain and args.end_to_end_families > 0 and isinstance(line.get("end_to_end"), dict) and "value" in line["end_to_end"]:
                try:   # the file -> file leg's CPU side (a quarter of its families: ~1 s of host work)
                    ce = cpu_end_to_end(max(1000, min(args.end_to_end_families, fam) // 4), args.depth, L, T, os.environ.get("FGX_BENCH_TMP", "/tmp/fgx_bench_e2e"))
                    line["cpu_baseline"]["end_to_end"] = ce
                    line["end_to_end"]["vs_cpu_end_to_end"] = line["end_to_end"]["value"] / ce["value"]
                    line["end_to_end"]["vs_cpu_end_to_end_if_its_stages_overlapped"] = line["end_to_end"]["value"] / ce["value_if_stages_overlapped"]
                except Exception as ex:
                    line["cpu_baseline"]["end_to_end"] = {"error": str(ex)[:300]}
    # ---- the gather to rank 0 (single-writer reassembly), LAST and under a watchdog: every other number of the line is final by now.  A rank that
    #      throws in the gather leaves the others blocked inside RCCL, where no exception reaches them — so each rank arms a timer: rank 0's prints the
    #      line as it stands (reassemble_error = the reason) and every rank's ends its process with exit code 0.
    if reassemble == "root" and world > 1:
        import threading
        state = {"line": line if rank == 0 else None, "done": False}

        def bail(reason):
            if state["done"]:
                return
            state["done"] = True
            if rank == 0:
                state["line"]["config"]["reassemble_error"] = reason
                print(json.dumps(state["line"]), flush=True)
            os._exit(0)
        timer = threading.Timer(args.gather_timeout, bail, args=(f"the gather loops did not finish within {args.gather_timeout} s (a rank failed or hung); every other field of the line was measured before them",))
        timer.daemon = True
        timer.start()
        try:
            dg, _fam, _sb = workload(args.scaling)
            caller.process_batch_device(dg)
            r = timed_loop(dg, args.steps, True)
            del dg
            torch.cuda.empty_cache()
            rs = None
            if S is not None:
                dg, _fam, _sb = workload("strong")
                caller.process_batch_device(dg)
                rs = timed_loop(dg, max(1, min(args.steps, 10)), True)
                del dg
        except Exception as ex:
            bail("gather failed on rank %d: %s" % (rank, str(ex)[:300]))
        timer.cancel()
        state["done"] = True
        if rank == 0:
            dt_gather, gathered_bytes = r[0], r[5]
            cfg = line["config"]
            cfg["reassembled_bytes_on_rank0"] = gathered_bytes
            cfg["value_with_reassembly_on_root"] = total_raw * args.steps / dt_gather
            cfg["ms_per_step_with_reassembly_on_root"] = dt_gather / args.steps * 1e3
            cfg["gather_GBs_into_root"] = ((gathered_bytes - int(per_rank[0, 0])) * args.steps / max(dt_gather - dt, 1e-9) / 1e9) if dt_gather > dt else None
            if rs is not None and "strong_scaling" in line:
                s_steps = max(1, min(args.steps, 10))
                line["strong_scaling"]["value_with_reassembly_on_root"] = int(S["per_rank"][:, 2].sum()) * s_steps / rs[0]
    if rank == 0:
        print(json.dumps(line), flush=True)
    if rank == 0:   # profiling builds (-DFGX_PHASE_TIMING=1) expose per-phase cycle totals of the family kernels
        import ctypes
        from fgumi_amd import lib
        if hasattr(lib, "fgx_debug_phase_cycles"):
            ph = (ctypes.c_uint64 * 16)()
            lib.fgx_debug_phase_cycles(ph, 1)
            tot = float(sum(ph[1:9])) or 1.0
            names = ["-", "stage", "parse", "overlap", "geometry", "gates", "columns", "umi", "descriptors"]
            print("phase share: " + "  ".join(f"{names[i]}={100.0 * ph[i] / tot:.1f}%" for i in range(1, 9)), file=sys.stderr)
            ta = float(sum(ph[1:16])) or 1.0
            print("phase cycles, share of all 15 slots (k_split_cols: 9..13 = packed pass rows | finalize | list | stores | items): " + " ".join(f"{i}:{100.0 * ph[i] / ta:.1f}" for i in range(1, 16)), file=sys.stderr)
            bn = ["raw->lds", "parse", "unpack", "overlap", "geometry", "gates", "columns(to umi)"]
            tb = float(sum(ph[9:16])) or 1.0
            print("k_family (workgroup) share: " + "  ".join(f"{bn[i - 9]}={100.0 * ph[i] / tb:.1f}%" for i in range(9, 16)), file=sys.stderr)
    caller.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
