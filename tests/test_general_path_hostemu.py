"""The general path's host orchestration, end to end on the CPU (tests/hostemu): source reads, annotation jobs, column jobs, duplex strand
combine and record assembly of fgumi_amd/csrc/{simplex,duplex}_host.cpp against the oracle, byte for byte — with the kernels' per-position
work done on the host by the same functions the kernels call.  The `-m gpu` suite runs the same inputs through libfgumi_amd.so."""
import numpy as np
import pytest

import bamutil
import cases
import fgx_opts
import hostemu
import methsim
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records


def oracle(o, contigs, g, batch_groups=50):
    orc.set_reference(contigs)
    try:
        return orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    finally:
        orc.set_reference(None)


def same(o, contigs, groups, batch_groups=50):
    g = groups if isinstance(groups, GroupedReads) else GroupedReads.from_groups(groups)
    want = oracle(o, contigs, g, batch_groups)
    got = hostemu.process(o, contigs, g)
    assert got["count"] == want["count"]
    if got["data"] != want["data"]:
        for i, (a, b) in enumerate(zip(split_records(got["data"]), split_records(want["data"]))):
            if a != b:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(a)}\nwant {bamutil.parse(b)}")
        raise AssertionError("record count / length differs")
    assert np.array_equal(got["stats"], want["stats"]), (got["stats"].tolist(), want["stats"].tolist())
    if o.track_rejects:
        assert got["rejects"] == want["rejects"] and got["n_rejects"] == want["n_rejects"]
    return want


def opts_from(kw):
    kw = dict(kw)
    mr = kw.pop("duplex_min_reads", None)
    o = fgx_opts.defaults(**kw)
    if mr:
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    return o


def test_methylation_reference_unit_test_inputs():
    import test_oracle_methylation_pins as pins
    n = 0
    for kw, contigs, groups in pins.replay_cases():
        same(opts_from(kw), contigs, groups, batch_groups=100 if kw.get("kind") == 1 else 50)
        n += 1
    assert n >= 18


@pytest.mark.parametrize("mode,kw", [
    (1, dict(min_reads=1)), (2, dict(min_reads=1)), (1, dict(min_reads=2, track_rejects=1)), (1, dict(min_reads=1, max_reads=3)),
    (2, dict(min_reads=2, overlapping_consensus=0, produce_per_base_tags=0)), (1, dict(min_reads=1, min_input_base_quality=25, trim=1)),
])
def test_simplex_em_seq_like_batches(mode, kw):
    rng = methsim.seeded(40 + mode)
    contigs = methsim.genome(rng)
    groups = methsim.simplex_groups(rng, contigs, 700)
    same(fgx_opts.defaults(methylation_mode=mode, **kw), contigs, groups)


@pytest.mark.parametrize("mode,min_reads,kw", [
    (1, (1, 1, 0), {}), (2, (1, 1, 0), {}), (1, (2, 1, 1), dict(track_rejects=1)), (1, (1, 1, 0), dict(duplex_max_reads_per_strand=2)),
    (1, (3, 2, 1), dict(produce_per_base_tags=0, overlapping_consensus=0)),
])
def test_duplex_em_seq_like_batches(mode, min_reads, kw):
    rng = methsim.seeded(70 + mode)
    contigs = methsim.genome(rng)
    groups = methsim.duplex_groups(rng, contigs, 500)
    o = fgx_opts.defaults(kind=1, methylation_mode=mode, **kw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = min_reads
    same(o, contigs, groups, batch_groups=100)


def test_mode_on_without_a_reference_is_the_plain_consensus():
    rng = methsim.seeded(21)
    contigs = methsim.genome(rng, n_contigs=1, length=1200)
    groups = methsim.simplex_groups(rng, contigs, 80)
    want = same(fgx_opts.defaults(min_reads=1, methylation_mode=1), None, groups)
    assert want["data"] == oracle(fgx_opts.defaults(min_reads=1), None, GroupedReads.from_groups(groups))["data"]


# ---- the same harness over the inputs the general path had before the methylation mode: its host code is shared -----------------

def test_simplex_general_path_unchanged_on_the_reference_unit_test_inputs_and_crafted_groups():
    import test_oracle_vanilla_pins as pins
    n = 0
    for kw, groups in pins.replay_cases():
        if sum(len(x) for x in groups) > 5000:
            continue                      # (the 40 000-read family: minutes on a scalar host loop; the GPU suite runs it)
        same(opts_from(kw), None, groups)
        n += 1
    assert n >= 25
    for mr in (1, 2):
        for tr in (0, 1):
            same(fgx_opts.defaults(min_reads=mr, track_rejects=tr), None, cases.crafted_groups())


def test_duplex_general_path_unchanged_on_the_reference_unit_test_inputs():
    import test_oracle_duplex_pins as pins
    n = 0
    for kw, groups in pins.replay_cases():
        same(opts_from(kw), None, groups, batch_groups=100)
        n += 1
    assert n >= 10
