"""Duplex caller oracle pinned on the fgbio-CAPTURED expectations the reference holds
(crates/fgumi-consensus/src/duplex_caller.rs:7466-7748: values recorded from a real fgbio run on
programmatically built fixtures), plus structural checks on simulated --duplex families."""
import struct

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads, simulate_grouped_reads, split_records


def duplex_fixture(n_ab, n_ba, bases, variant=None, variant_count=0, start2=200):
    """build_duplex_fixture (duplex_caller.rs:7475-7524), template-coordinate order: /A and /B pairs share
    coordinates, so records sort by (strand orientation, name); mates stay adjacent."""
    recs = []
    for i in range(n_ab):
        b1 = variant if (variant and i < variant_count) else bases
        recs += list(bamutil.pair2(f"a{i:07d}", b1, 40, bases, 40, "mol/A", 100, start2, rev1=False, rev2=True))
    for i in range(n_ba):
        recs += list(bamutil.pair2(f"b{i:07d}", bases, 40, bases, 40, "mol/B", 100, start2, rev1=True, rev2=False))
    return GroupedReads.from_groups([recs])


def run_duplex(g, min_reads=(1, 1, 1), **kw):
    base = dict(overlapping_consensus=0, read_name_prefix=b"duplex", cell_tag=b"\0\0")
    base.update(kw)
    o = fgx_opts.defaults(kind=1, **base)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = min_reads
    return orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)


def tags(rec):
    return {k: v[1] for k, v in bamutil.parse(rec)["tags"].items()}


def check(rec, cd, cm, ce, ad, am, ae, bd, bm, be, ad_b, bd_b, ae_b, be_b):
    t = tags(rec)
    assert (t["cD"], t["cM"], t["aD"], t["aM"], t["bD"], t["bM"]) == (cd, cm, ad, am, bd, bm)
    for got, want in ((t["cE"], ce), (t["aE"], ae), (t["bE"], be)):
        assert abs(got - want) < 1e-6, (got, want)
    assert t["ad"] == ad_b and t["bd"] == bd_b and t["ae"] == ae_b and t["be"] == be_b


def test_open_interval_depth_saturation_fgbio():  # duplex_caller.rs:7561-7573
    res = run_duplex(duplex_fixture(33000, 20000, "ACGT"))
    recs = split_records(res["data"])
    assert len(recs) == 2
    r1 = [r for r in recs if bamutil.parse(r)["flag"] & 0x40][0]
    check(r1, 52767, 52767, 0.0, 32767, 32767, 0.0, 20000, 20000, 0.0, [32767] * 4, [20000] * 4, [0] * 4, [0] * 4)


def test_error_numerator_saturation_fgbio():  # duplex_caller.rs:7578-7589
    res = run_duplex(duplex_fixture(73000, 20000, "ACGTACGT", "CCGTACGT", 33000))
    recs = split_records(res["data"])
    r1 = [r for r in recs if bamutil.parse(r)["flag"] & 0x40][0]
    check(r1, 52767, 52767, 0.07762191, 32767, 32767, 0.125, 20000, 20000, 0.0, [32767] * 8, [20000] * 8, [32767] + [0] * 7, [0] * 8)


def test_strand_split_fgbio():  # duplex_caller.rs:7621-7700
    res = run_duplex(duplex_fixture(3, 2, "ACGTACGT", "CCGTACGT", 1))
    recs = split_records(res["data"])
    r1 = [r for r in recs if bamutil.parse(r)["flag"] & 0x40][0]
    r2 = [r for r in recs if not bamutil.parse(r)["flag"] & 0x40][0]
    check(r1, 5, 5, 0.025, 3, 3, 0.0416667, 2, 2, 0.0, [3] * 8, [2] * 8, [1] + [0] * 7, [0] * 8)
    check(r2, 5, 5, 0.0, 3, 3, 0.0, 2, 2, 0.0, [3] * 8, [2] * 8, [0] * 8, [0] * 8)
    res = run_duplex(duplex_fixture(1, 1, "ACGTACGT"))
    recs = split_records(res["data"])
    for r in recs:
        check(r, 2, 2, 0.0, 1, 1, 0.0, 1, 1, 0.0, [1] * 8, [1] * 8, [0] * 8, [0] * 8)
    p = bamutil.parse(recs[0])
    assert p["name"] == "duplex:mol" and p["tag_order"] == ["MI", "RG", "aD", "aE", "aM", "ac", "ad", "ae", "aq", "bD", "bE", "bM", "bc", "bd", "be", "bq", "cD", "cE", "cM"]


def test_empty_ba_strand_emits_nothing_like_fgbio():  # duplex_caller.rs:7738-7748
    res = run_duplex(duplex_fixture(2, 0, "ACGTACGT"))
    assert res["count"] == 0 and res["data"] == b""
    assert res["stats"][0] == 4 and res["stats"][3 + 1] == 4          # InsufficientReads for the whole group
    # with min_reads [1,1,0] the AB-only molecule is emitted (single-strand consensus allowed)
    res = run_duplex(duplex_fixture(2, 0, "ACGTACGT"), min_reads=(1, 1, 0))
    assert res["count"] == 2
    t = tags(split_records(res["data"])[0])
    assert (t["aD"], t["bD"], t["cD"]) == (2, 0, 2) and "bc" not in t


def test_duplex_rejections():
    # fragment reads are rejected; MI without /A,/B is fatal; potential strand collision
    frag = bamutil.frag("f", "ACGTACGT", 30, "mol/A")
    g = GroupedReads.from_groups([[frag] + list(bamutil.pair2("a0", "ACGTACGT", 40, "ACGTACGT", 40, "mol/A", 100, 200)) +
                                  list(bamutil.pair2("b0", "ACGTACGT", 40, "ACGTACGT", 40, "mol/B", 100, 200, rev1=True, rev2=False))])
    res = run_duplex(g)
    assert res["count"] == 2 and res["stats"][3 + 0] == 1
    bad = GroupedReads.from_groups([list(bamutil.pair2("a0", "ACGT", 40, "ACGT", 40, "mol", 100, 200))])
    with pytest.raises(RuntimeError, match="suffix"):
        run_duplex(bad)
    coll = GroupedReads.from_groups([list(bamutil.pair2("a0", "ACGTACGT", 40, "ACGTACGT", 40, "mol/A", 100, 200)) +
                                     list(bamutil.pair2("b0", "ACGTACGT", 40, "ACGTACGT", 40, "mol/B", 100, 200, rev1=False, rev2=True))])
    res = run_duplex(coll)
    assert res["count"] == 0 and res["stats"][3 + 17] == 4          # PotentialCollision


def test_duplex_on_simulated_families():
    g = simulate_grouped_reads(300, family_size=6, duplex=1)
    res = run_duplex(g, overlapping_consensus=1, cell_tag=b"CB", read_name_prefix=b"")
    recs = [bamutil.parse(r) for r in split_records(res["data"])]
    assert res["count"] == len(recs) == 600 and res["stats"][0] == g.n_rec
    r = recs[0]
    assert r["tag_order"][:5] == ["MI", "RG", "aD", "aE", "aM"] and r["tag_order"][-1] == "RX" and r["name"] == ":0"
    assert r["tags"]["aD"][1] + r["tags"]["bD"][1] >= r["tags"]["cD"][1] >= 1
    # threads / batches do not change bytes
    o = fgx_opts.defaults(kind=1)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 1
    a = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100, threads=1)
    b = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=7, threads=3)
    assert a["data"] == b["data"] and np.array_equal(a["stats"], b["stats"])
