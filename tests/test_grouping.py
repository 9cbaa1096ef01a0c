"""MI grouping (`MiGrouper`, src/lib/mi_group.rs): the oracle restatement pinned on the reference's tests, and the device
kernels against the oracle."""
import numpy as np
import pytest

import bamutil
import orc
from fgumi_amd import GroupedReads, simulate_grouped_reads


def _stream(recs):
    g = GroupedReads.from_groups([recs])
    return g.blob, g.rec_off, g.rec_len


def _rec(mi=None, cb=None, flag=0, name="q"):
    tags = ([("MI", "Z", mi)] if mi is not None else []) + ([("CB", "Z", cb)] if cb is not None else [])
    return bamutil.make_record(name, "ACGT", [30] * 4, flag=flag, tags=tags)


def _sizes(grp):
    return np.diff(grp).tolist()


def test_multiple_groups_and_skips():   # mi_group.rs:624-669
    off, ln, grp = orc.group_records(*_stream([_rec("0"), _rec("0"), _rec("1"), _rec("1"), _rec("1"), _rec("2")]), cell_tag=None)
    assert _sizes(grp) == [2, 3, 1]
    off, ln, grp = orc.group_records(*_stream([_rec("0"), _rec(), _rec("0"), _rec(), _rec("1")]), cell_tag=None)
    assert _sizes(grp) == [2, 1] and len(off) == 3


def test_cell_tag_composite_key():   # mi_group.rs:848-942
    s = _stream([_rec("1", "ACGT"), _rec("1", "ACGT"), _rec("1", "TGCA"), _rec("1", "TGCA")])
    assert _sizes(orc.group_records(*s, cell_tag=b"CB")[2]) == [2, 2]
    assert _sizes(orc.group_records(*s, cell_tag=None)[2]) == [4]
    s = _stream([_rec("1"), _rec("1"), _rec("1", "ACGT")])                       # a missing cell tag keys as "1\t"
    assert _sizes(orc.group_records(*s, cell_tag=b"CB")[2]) == [2, 1]
    s = _stream([_rec("1/A", "ACGT"), _rec("1/B", "ACGT"), _rec("1/A", "TGCA")])
    assert _sizes(orc.group_records(*s, cell_tag=b"CB", strip_strand_suffix=True)[2]) == [2, 1]


def test_record_filter_and_transform():   # mi_group.rs:1104-1147, commands/common.rs:384-397, fgumi-umi lib.rs:370-375
    s = _stream([_rec("1/A"), _rec("1/A", flag=0x100), _rec("1/B"), _rec("1/B", flag=0x800), _rec("1/A", flag=0x4)])
    off, ln, grp = orc.group_records(*s, cell_tag=None, strip_strand_suffix=True)
    assert _sizes(grp) == [2]
    off, ln, grp = orc.group_records(*s, cell_tag=None, strip_strand_suffix=True, allow_unmapped=True)
    assert _sizes(grp) == [3]
    s = _stream([_rec("/A"), _rec("/B"), _rec("a/b/A"), _rec("a/b/B"), _rec("a/b")])      # a leading '/' is not a suffix; only the LAST '/' cuts
    assert _sizes(orc.group_records(*s, cell_tag=None, strip_strand_suffix=True)[2]) == [1, 1, 2, 1]
    off, ln, grp = orc.group_records(*_stream([]), cell_tag=None)
    assert len(off) == 0 and grp.tolist() == [0]


def test_simulated_streams_regroup_exactly():
    for kw in (dict(family_size=3), dict(family_size=4, duplex=1)):
        g = simulate_grouped_reads(500, **kw)
        off, ln, grp = orc.group_records(g.blob, g.rec_off, g.rec_len, cell_tag=b"CB", strip_strand_suffix=bool(kw.get("duplex")))
        assert np.array_equal(off, g.rec_off) and np.array_equal(ln, g.rec_len) and np.array_equal(grp, g.grp_first)


def _mixed_stream():
    import random
    rng = random.Random(5)
    recs = []
    for m in range(400):
        n = rng.randint(1, 6)
        cbs = [rng.choice(["AAAA", "CCCC", None, ""]) for _ in range(2)]
        for i in range(n):
            strand = rng.choice("AB")
            flag = rng.choice([0, 0, 0, 0x100, 0x800, 0x4, 0x10])
            mi = None if rng.random() < 0.05 else f"{m // 2}/{strand}" if rng.random() < 0.8 else f"{m // 2}"
            recs.append(_rec(mi, cbs[i % 2] if rng.random() < 0.9 else None, flag=flag, name=f"r{m}_{i}"))
    return _stream(recs)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(cell_tag="CB"), dict(cell_tag=None), dict(cell_tag="CB", strip_strand_suffix=True),
                                dict(cell_tag=None, strip_strand_suffix=True, allow_unmapped=True)])
def test_device_grouping_matches_oracle(kw):
    from fgumi_amd import VanillaUmiConsensusCaller
    c = VanillaUmiConsensusCaller("", "A")
    okw = dict(kw)
    okw["cell_tag"] = kw["cell_tag"].encode() if kw["cell_tag"] else None
    for blob, off, ln in (_mixed_stream(), _stream([]), _stream([_rec()]), _stream([_rec("7", "X")])):
        want = orc.group_records(blob, off, ln, **okw)
        got = c.group_records(blob, off, ln, **kw)
        assert np.array_equal(got.rec_off, want[0]) and np.array_equal(got.rec_len, want[1]) and np.array_equal(got.grp_first, want[2])
    # a simulated stream on the device: regrouping reproduces the generator's own boundaries, and feeds the caller
    dg = c.simulate_on_device(3000, family_size=4)
    rg = c.group_records_device(dg)
    assert rg.n_grp == 3000 and rg.n_rec == dg.n_rec
    import torch
    assert torch.equal(rg.grp_first[:3001].cpu(), dg.grp_first.cpu()) and torch.equal(rg.rec_off[:dg.n_rec].cpu(), dg.rec_off[:dg.n_rec].cpu())
    out = c.process_batch_device(rg)
    assert out.count == 6000 and out.n_deferred == 0
    c.close()
