"""GPU parity of the canonical second pass for CODEC molecules (default since round 4, FGX_CODEC_CANON=0 opts out; fgumi_amd/csrc/canon_core.h
`canon_codec_molecule` + api.cpp `canon_second_pass`): molecules with soft clips, indels, skips or pads in their CIGARs, which the
first device pass defers, are rewritten into their canonical form (proved equivalent through the oracle in tests/test_canon_codec.py)
and decided by the device pipeline in a second pass — byte-identical to the oracle, counters included, and the diagnostics show the
second pass took them.  First ran on hardware in the driver's round-3 GPU run (XPASS); each test still runs in a child interpreter."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
import test_canon_codec as tcc
from fgumi_amd import GroupedReads, simulate_grouped_reads, split_records
from isolated import run_isolated
from test_gpu_duplex_canon import product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [dict(), dict(min_input_base_quality=20, produce_per_base_tags=1), dict(codec_min_reads_per_strand=2, cell_tag=b"\0\0"),
                                dict(codec_outer_bases_length=5, codec_has_outer_bases_qual=1, codec_outer_bases_qual=7, codec_min_duplex_length=10)])
def test_codec_indel_molecules_take_the_canonical_second_pass(kw):
    run_isolated("test_gpu_zz_codec_canon", "check_codec_indel_molecules", kw, env={"FGX_CODEC_CANON": "1"})


def check_codec_indel_molecules(kw):
    rng = random.Random(41)
    groups = []
    sim = simulate_grouped_reads(120, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1)     # regular molecules in between
    for g in range(360):
        groups.append(sim.records(g // 3) if g % 3 == 0 else tcc.codec_molecule(rng, 1000 + g))
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=2, overlapping_consensus=0, **kw)
    want = orc.process(o, gr.blob, gr.rec_off, gr.rec_len, gr.grp_first, batch_groups=1000)
    got = product(o, gr)
    assert got["deferred"] > 50 and got["canon"] > 0.2 * got["deferred"], (got["deferred"], got["canon"])
    if got["data"] != want["data"]:
        for i, (a, b) in enumerate(zip(split_records(got["data"]), split_records(want["data"]))):
            if a != b:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(a)}\nwant {bamutil.parse(b)}")
        raise AssertionError("record count / length differs")
    assert got["count"] == want["count"] and np.array_equal(got["stats"], want["stats"]), (got["stats"].tolist(), want["stats"].tolist())


def test_codec_second_pass_can_be_switched_off():
    run_isolated("test_gpu_zz_codec_canon", "check_switched_off", env={"FGX_CODEC_CANON": "0"})


def check_switched_off():
    rng = random.Random(42)
    groups = [tcc.codec_molecule(rng, g) for g in range(80)]
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=2, overlapping_consensus=0)
    want = orc.process(o, gr.blob, gr.rec_off, gr.rec_len, gr.grp_first, batch_groups=1000)
    got = product(o, gr)
    assert got["deferred"] > 0 and got["canon"] == 0 and got["data"] == want["data"] and np.array_equal(got["stats"], want["stats"])
