#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: frozen inputs and the consensus bytes / counters the ORACLE produces for them.

The reference itself cannot run in this image (no Rust toolchain), so these are not reference outputs: they freeze
what the restatement — pinned on the reference's own known-answer vectors in tests/test_oracle_*.py — says today, so
that (a) oracle drift is caught on CPU and (b) the GPU box can check the HIP paths without trusting a freshly built
oracle.  The known-answer values that DO come from the reference (its unit-test pins and the fgbio-captured tag
values) live in reference_pins.json, transcribed from the cited file:line.

usage: python tests/golden/make_golden.py        (writes next to itself; needs oracle/_build/liboracle.so)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases            # noqa: E402
import fgx_opts         # noqa: E402
import orc              # noqa: E402
import test_oracle_codec as toc      # noqa: E402
import test_oracle_duplex as tod     # noqa: E402
from fgumi_amd import GroupedReads, simulate_grouped_reads   # noqa: E402


def freeze(name, g, o, batch):
    res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), blob=g.blob, rec_off=g.rec_off, rec_len=g.rec_len, grp_first=g.grp_first,
                        data=np.frombuffer(res["data"], dtype=np.uint8), count=np.uint64(res["count"]), stats=res["stats"])
    print(name, g.n_grp, "groups ->", res["count"], "records,", len(res["data"]), "bytes")


def inputs():
    """(name, GroupedReads, options, batch) for every fixture; shared with tests/test_golden.py."""
    out = []
    out.append(("simplex_crafted", GroupedReads.from_groups(cases.crafted_groups()), fgx_opts.defaults(min_reads=1), 50))
    out.append(("simplex_sim_depth3", simulate_grouped_reads(300, family_size=3), fgx_opts.defaults(min_reads=1), 50))
    o = fgx_opts.defaults(kind=1)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 0
    out.append(("duplex_sim", simulate_grouped_reads(150, family_size=6, duplex=1, error_rate_ppm=10000), o, 100))
    o = fgx_opts.defaults(kind=1, overlapping_consensus=0, read_name_prefix=b"duplex", cell_tag=b"\0\0")
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 1
    out.append(("duplex_fgbio_fixture", tod.duplex_fixture(3, 2, "ACGTACGT", "CCGTACGT", 1), o, 100))
    import test_gpu_codec as tgc
    out.append(("codec_crafted", GroupedReads.from_groups(tgc.crafted_groups()),
                fgx_opts.defaults(kind=2, read_name_prefix=b"codec", overlapping_consensus=0, cell_tag=b"CB", produce_per_base_tags=1), 1000))
    out.append(("codec_sim", simulate_grouped_reads(150, family_size=3, read_length=300, insert_mean=350, insert_sd=60, codec=1),
                fgx_opts.defaults(kind=2, read_name_prefix=b"codec", overlapping_consensus=0, produce_per_base_tags=1), 1000))
    import test_gpu_indels as tgi          # indel / clip / skip CIGARs, minority alignments (alignment filter), --max-reads downsampling
    out.append(("simplex_indels", GroupedReads.from_groups(tgi.indel_groups(seed=21, n_groups=60, max_pairs=8)), fgx_opts.defaults(min_reads=1), 50))
    out.append(("simplex_indels_max_reads", GroupedReads.from_groups(tgi.indel_groups(seed=22, n_groups=40, max_pairs=8)), fgx_opts.defaults(min_reads=2, max_reads=3), 50))
    return out


def filter_inputs():
    """(name, records, oracle filter keyword options) for the `fgumi filter` fixtures; shared with tests/test_golden.py."""
    import test_gpu_filter as tgf
    recs = tgf.crafted()
    return [("filter_crafted_default", recs, dict()),
            ("filter_crafted_duplex_tiers", recs, dict(min_reads=[5, 3, 2], max_read_error_rate=[0.03, 0.02, 0.05], max_base_error_rate=[0.3, 0.2, 0.4],
                                                       min_base_quality=12, track_rejects=True, require_single_strand_agreement=True)),
            ("filter_crafted_single_read", recs, dict(min_reads=[2], filter_by_template=False, track_rejects=True, min_base_quality=10, min_mean_base_quality=30.0))]


def freeze_filter(name, recs, kw):
    import test_oracle_filter as tof
    blob, off, ln = tof.stream(recs)
    res = orc.filter_records(orc.filter_options(**kw), blob, off, ln)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), blob=blob, rec_off=off, rec_len=ln, data=np.frombuffer(res["data"], dtype=np.uint8),
                        rejects=np.frombuffer(res["rejects"], dtype=np.uint8),
                        counts=np.array([res["records"], res["passed"], res["masked"], res["rejected"]], dtype=np.uint64))
    print(name, len(recs), "records ->", res["passed"], "kept,", res["masked"], "bases masked")


if __name__ == "__main__":
    for name, g, o, batch in inputs():
        freeze(name, g, o, batch)
    for name, recs, kw in filter_inputs():
        freeze_filter(name, recs, kw)
