#!/usr/bin/env python3
"""Times the opt-in device paths against the paths they replace, through the C ABI (host buffers in and out), one JSON line per case:

  duplex / codec : a batch in which `--indel-fraction` of the molecules carry indel / soft-clip CIGARs (the others are simulated one-M
                   molecules), through fgx_process_batch with the canonical second pass off (deferred molecules → general path), on with
                   the form computed on the host's cores (FGX_*_CANON=1), and on with the form computed by the device kernel
                   (+ FGX_CANON_DEVICE=1); every variant's bytes are compared with the first's.
  rejects        : a simplex batch with `track_rejects`, through fgx_process_batch with the side kernels off (whole batch on the general
                   path) and on (FGX_REJECTS_DEVICE=1).
  pipeline       : a duplex BAM file with the same share of indel molecules through fgx_run_bam: the whole chunk through the host entry
                   whenever its device batch defers groups (default), only the deferred groups (FGX_PIPE_SUBSET=1), and with the
                   canonical second pass inside the device entry as well (+ FGX_DUPLEX_CANON=1 FGX_CANON_RESIDENT=1); the output files'
                   records are compared.

The indel molecules come from the test generators (tests/test_canon_core.py, tests/test_canon_codec.py: Python, a few thousand per
second), so the batches are small — this measures per-molecule cost, not a roofline.  The environment switches are read per call, so one
process times all variants.  Needs a GPU; `FGX_LIB=tests/hostemu/_build/libapiemu.so` dry-runs it on the CPU (times meaningless).

usage: python tools/bench_opt_in_paths.py [--molecules 3000] [--indel-fraction 0.3] [--steps 3] [--cases duplex,codec,rejects,pipeline]"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLAGS = ("FGX_DUPLEX_CANON", "FGX_CODEC_CANON", "FGX_CANON_DEVICE", "FGX_REJECTS_DEVICE", "FGX_CANON_RESIDENT", "FGX_PIPE_SUBSET")


def run(o, g, steps, **env):
    import numpy as np
    from fgumi_amd._lib import Options, Output, lib
    for k in FLAGS:
        os.environ.pop(k, None)
    os.environ.update({k: "1" for k, v in env.items() if v})
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    if not h:
        raise RuntimeError(lib.fgx_global_error().decode())
    try:
        out = Output()
        best = None
        for i in range(steps + 1):                      # (the first call warms buffers up)
            t0 = time.perf_counter()
            rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
            dt = time.perf_counter() - t0
            if rc != 0:
                raise RuntimeError(lib.fgx_last_error(h).decode())
            if i and (best is None or dt < best):
                best = dt
        d = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d)
        data = C.string_at(out.data, out.data_len) if out.data_len else b""
        rej = C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b""
        return dict(ms=best * 1e3, raw_reads_per_s=g.n_rec / best, first_pass_deferred=int(d[0]), decided_by_second_pass=int(d[1]), records=int(out.count),
                    n_rejects=int(out.n_rejects), kernels_ms=out.ms_kernels, host_prep_ms=out.ms_host_prep), (data, rej, list(out.stats))
    finally:
        lib.fgx_destroy(h)


def pipeline_case(a, rng):
    import tempfile
    import test_canon_core as tc
    from fgumi_amd import DuplexConsensusCaller, GroupedReads, bgzf, simulate_grouped_reads
    sim = simulate_grouped_reads(a.molecules, family_size=4, duplex=1)
    groups, used = [], 0
    for i in range(a.molecules):
        m = tc.duplex_indel_molecule(rng, 10 ** 6 + i) if rng.random() < a.indel_fraction else None
        if not m:
            m, used = sim.records(used), used + 1
        groups.append(m)
    g = GroupedReads.from_groups(groups)
    refs = [("chr%d" % (i + 1), 2147483647) for i in range(24)]
    variants = [("whole chunk through the host entry", {}), ("only the deferred groups", dict(FGX_PIPE_SUBSET=1)),
                ("only the deferred groups, canonical second pass inside the device entry", dict(FGX_PIPE_SUBSET=1, FGX_DUPLEX_CANON=1, FGX_CANON_RESIDENT=1))]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "grouped.bam")
        bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob)
        ref = None
        for name, env in variants:
            for k in FLAGS:
                os.environ.pop(k, None)
            os.environ.update({k: "1" for k in env})
            c = DuplexConsensusCaller("", "A", [1, 1, 0], cell_tag="CB", overlapping_consensus=True)
            dst = os.path.join(d, "out.bam")
            best = None
            for i in range(a.steps + 1):
                t0 = time.perf_counter()
                st = c.run_bam(src, dst, strip_strand_suffix=True, cell_tag=None)
                dt = time.perf_counter() - t0
                if i and (best is None or dt < best):
                    best = dt
            c.close()
            _, _, stream, off, ln = bgzf.read_bam(dst)
            recs = b"".join(bytes(stream[int(o) - 4:int(o) + int(l)]) for o, l in zip(off, ln))
            if ref is None:
                ref = recs
            print(json.dumps({"case": "pipeline", "variant": name, "workload": f"{g.n_grp} duplex molecules in a BAM file, {a.indel_fraction:.0%} with indel CIGARs",
                              "raw_reads": int(g.n_rec), "unit": "file to file, best of steps", "ms": best * 1e3, "raw_reads_per_s": g.n_rec / best,
                              "deferred_groups": st["deferred_groups"], "consensus_records": st["consensus_records"], "same_records_as_first_variant": recs == ref}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--molecules", type=int, default=3000)
    ap.add_argument("--indel-fraction", type=float, default=0.3)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cases", default="duplex,codec,rejects,pipeline")
    a = ap.parse_args()
    import fgx_opts
    import test_canon_codec as tcc
    import test_canon_core as tc
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    rng = random.Random(1)
    for case in a.cases.split(","):
        if case == "pipeline":
            pipeline_case(a, rng)
            continue
        if case == "rejects":
            sim = simulate_grouped_reads(a.molecules, family_size=1, family_size_max=9, seed=2)
            g = GroupedReads.from_groups([sim.records(i) for i in range(sim.n_grp)])
            o = fgx_opts.defaults(kind=0, track_rejects=1, min_reads=2, max_reads=6)
            variants = [("general path (whole batch)", {}), ("side kernels", dict(FGX_REJECTS_DEVICE=1))]
            shape = f"{g.n_grp} simplex families of 1..9 pairs x 150bp, --min-reads 2 --max-reads 6 --rejects"
        else:
            codec = case == "codec"
            sim = (simulate_grouped_reads(a.molecules, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1) if codec
                   else simulate_grouped_reads(a.molecules, family_size=4, duplex=1))
            groups = []
            for i in range(a.molecules):
                m = None
                if rng.random() < a.indel_fraction:
                    m = tcc.codec_molecule(rng, 10 ** 6 + i) if codec else tc.duplex_indel_molecule(rng, 10 ** 6 + i)
                groups.append(m if m else sim.records(i))
            g = GroupedReads.from_groups(groups)
            o = fgx_opts.defaults(kind=2, overlapping_consensus=0) if codec else fgx_opts.defaults(kind=1)
            flag = "FGX_CODEC_CANON" if codec else "FGX_DUPLEX_CANON"
            variants = [("deferred molecules on the general path", {}), ("canonical second pass, form made on the host", {flag: 1}),
                        ("canonical second pass, form made by the device kernel", {flag: 1, "FGX_CANON_DEVICE": 1})]
            shape = f"{g.n_grp} {case} molecules, {a.indel_fraction:.0%} with indel / soft-clip CIGARs"
        ref = None
        for name, env in variants:
            r, payload = run(o, g, a.steps, **env)
            if ref is None:
                ref = payload
            r["same_bytes_and_counters_as_first_variant"] = payload == ref
            print(json.dumps({"case": case, "variant": name, "workload": shape, "raw_reads": int(g.n_rec), "unit": "host buffers in and out, best of steps", **r}))


if __name__ == "__main__":
    main()
