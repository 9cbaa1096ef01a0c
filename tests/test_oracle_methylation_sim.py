"""Size-independent properties of the methylation-aware mode on seeded EM-Seq-like families (tests/methsim.py), checked on the oracle:
what the GPU parity suite compares byte for byte is, here, at least well-formed — MM / ML agree with the tracked bases of SEQ, the count
arrays cover the consensus, and switching the mode off gives the plain consensus of the same reads wherever no reference cytosine is hit."""
import numpy as np
import pytest

import bamutil
import fgx_opts
import methsim
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records


def run(opts, contigs, groups, batch_groups=50):
    g = GroupedReads.from_groups(groups)
    orc.set_reference(contigs)
    try:
        res = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    finally:
        orc.set_reference(None)
    return res, [bamutil.parse(r) for r in split_records(res["data"])]


def check_tags(rec, prefix_mm="MM", ml="ML", cu="cu", ct="ct", seq=None):
    t = {k: v[1] for k, v in rec["tags"].items()}
    seq = seq if seq is not None else rec["seq"]
    if cu in t:
        assert len(t[cu]) == len(seq) == len(t[ct])
        assert all(0 <= v <= 32767 for v in t[cu] + t[ct])
    if prefix_mm in t:
        mm = t[prefix_mm]
        assert mm[:3] in ("C+m", "G-m") and mm.endswith(";")
        skips = [int(x) for x in mm[4:-1].split(",")]
        tracked = sum(1 for b in seq if b == mm[0])
        assert sum(skips) + len(skips) <= tracked
        if ml in t:
            assert len(t[ml]) == len(skips)
        # every listed base is a reference cytosine with evidence
        idx = [i for i, b in enumerate(seq) if b == mm[0]]
        k = -1
        for s in skips:
            k += s + 1
            assert t[cu][idx[k]] + t[ct][idx[k]] > 0


@pytest.mark.parametrize("mode", [1, 2])
def test_simplex_tags_are_well_formed_and_the_run_is_thread_invariant(mode):
    rng = methsim.seeded(11 + mode)
    contigs = methsim.genome(rng)
    groups = methsim.simplex_groups(rng, contigs, 400)
    o = fgx_opts.defaults(min_reads=1, methylation_mode=mode)
    res, recs = run(o, contigs, groups)
    assert res["count"] > 300 and sum("cu" in r["tags"] for r in recs) > 250 and sum("MM" in r["tags"] for r in recs) > 100
    for r in recs:
        check_tags(r)
    g = GroupedReads.from_groups(groups)
    orc.set_reference(contigs)
    try:
        again = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=7, threads=4)
    finally:
        orc.set_reference(None)
    assert again["data"] == res["data"] and np.array_equal(again["stats"], res["stats"])


def test_duplex_tags_are_well_formed():
    rng = methsim.seeded(5)
    contigs = methsim.genome(rng)
    groups = methsim.duplex_groups(rng, contigs, 300)
    o = fgx_opts.defaults(kind=1, methylation_mode=1)
    res, recs = run(o, contigs, groups, batch_groups=100)
    assert res["count"] > 200
    n_both = 0
    for r in recs:
        t = {k: v[1] for k, v in r["tags"].items()}
        check_tags(r)
        if "au" in t and "bu" in t:
            n_both += 1
            assert [a + b for a, b in zip(t["au"], t["bu"])] == t["cu"] and [a + b for a, b in zip(t["at"], t["bt"])] == t["ct"]
        assert ("au" in t) == ("at" in t) and ("bu" in t) == ("bt" in t)
    assert n_both > 50


def test_mode_off_ignores_the_reference():
    rng = methsim.seeded(3)
    contigs = methsim.genome(rng)
    groups = methsim.simplex_groups(rng, contigs, 100)
    o = fgx_opts.defaults(min_reads=1, methylation_mode=0)
    with_ref, _ = run(o, contigs, groups)
    without, _ = run(o, None, groups)
    assert with_ref["data"] == without["data"]
    assert not any(t in r["tags"] for r in map(bamutil.parse, split_records(with_ref["data"])) for t in ("MM", "ML", "cu", "ct"))
