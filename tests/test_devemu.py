"""The new lane-per-item device paths driven END TO END on the CPU (tests/devemu: reject_device.hip and canon_device.hip compiled for the
host, a launch = a serial loop): slab sizing, grid-stride loops, the scan of the groups' bytes, offsets and totals of the `--rejects` side
kernels against the oracle's rejects; the canonicalisation kernel with api.cpp's slot layout against the per-molecule host entries.  What
this cannot show is the hardware running them — tests/test_gpu_zz_rejects_device.py and tests/test_gpu_zz_canon_device.py do."""
import random

import numpy as np
import pytest

import devemu
import fgx_opts
import orc
import test_canon_codec as tcc
import test_canon_core as tc
import test_general_path_fuzz as fuzz
import test_gpu_zz_rejects_device as tgr
from fgumi_amd import GroupedReads, simulate_grouped_reads


@pytest.mark.parametrize("kw", tgr.KWS)
def test_reject_kernels_equal_the_oracle(kw):
    g = tgr.batch(11)                                              # the batch the GPU test sends
    o = fgx_opts.defaults(kind=0, track_rejects=1, **kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    oos, got, cnt = devemu.simplex_rejects(o, g)
    assert oos == 0 and cnt == want["n_rejects"] > 0 and got == want["rejects"]


@pytest.mark.parametrize("seed", range(12))
def test_reject_kernels_on_hostile_groups(seed):
    rng = random.Random(31000 + seed)
    groups = [x for x in (fuzz.random_group(rng, g, "simplex", rng.random() < 0.7) for g in range(80)) if x]
    o = fuzz.random_options(rng, "simplex")
    o.track_rejects = 1
    g = GroupedReads.from_groups(groups)
    try:
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)
    except RuntimeError:
        want = None
    oos, got, cnt = devemu.simplex_rejects(o, g)
    if oos:
        # some group is out of scope: exactly the groups the host function refuses
        import test_reject_core as trc
        bad = sum(trc.product_rejects(o, GroupedReads.from_groups([x]))[0] != 0 for x in groups)
        assert oos == bad
        keep = [x for x in groups if trc.product_rejects(o, GroupedReads.from_groups([x]))[0] == 0]
        g = GroupedReads.from_groups(keep)
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)
        oos, got, cnt = devemu.simplex_rejects(o, g)
    assert want is not None and oos == 0
    assert cnt == want["n_rejects"] and got == want["rejects"]


def test_reject_kernels_edge_shapes():
    o = fgx_opts.defaults(kind=0, track_rejects=1, min_reads=2)
    big = simulate_grouped_reads(1, family_size=70, seed=3)                       # 140 records: out of scope, and it must not size the slabs
    small = simulate_grouped_reads(3, family_size=1, seed=4)                       # below --min-reads: rejected whole, original bytes
    g = GroupedReads.from_groups([small.records(0), big.records(0), small.records(1)])
    assert devemu.simplex_rejects(o, g)[0] == 1
    g = GroupedReads.from_groups([small.records(i) for i in range(3)])
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    assert devemu.simplex_rejects(o, g) == (0, want["rejects"], want["n_rejects"])
    many = simulate_grouped_reads(700, family_size=1, family_size_max=4, seed=5)   # more groups than one block of lanes
    g = GroupedReads.from_groups([many.records(i) for i in range(many.n_grp)])
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    assert devemu.simplex_rejects(o, g) == (0, want["rejects"], want["n_rejects"])


def _canon_batch(rng, maker, n):
    sim = simulate_grouped_reads(n, family_size=2, duplex=1, seed=7)
    groups, deferred = [], []
    for i in range(n):
        groups.append(sim.records(i))
        m = maker(rng, 4000 + i)
        if m:
            deferred.append(len(groups))
            groups.append(m)
    return groups, deferred


@pytest.mark.parametrize("kw,mr", [(dict(overlapping_consensus=1), (1, 1, 0)), (dict(overlapping_consensus=0, min_input_base_quality=20), (2, 1, 1))])
def test_canon_kernel_equals_the_host_entry_duplex(kw, mr):
    rng = random.Random(51)
    groups, deferred = _canon_batch(rng, tc.duplex_indel_molecule, 90)
    o = fgx_opts.defaults(kind=1, **kw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    g = GroupedReads.from_groups(groups)
    res = devemu.canon(o, False, g, deferred)
    ok = 0
    for gi, (st, recs, pre, delta) in zip(deferred, res):
        rc, want, wd = tc.canonicalise(o, groups[gi])
        assert st == rc
        if rc != 0:
            continue
        ok += 1
        assert [r for r in recs if r] == list(want)                               # (the host helper lists the kept records only)
        assert all(p == len(r) for r, p in zip(recs, pre) if r)                  # the block_size prefixes the kernel writes itself
        assert delta == [int(x) for x in wd]
    assert ok > 30


@pytest.mark.parametrize("kw", [dict(), dict(codec_min_reads_per_strand=2, cell_tag=b"\0\0")])
def test_canon_kernel_equals_the_host_entry_codec(kw):
    rng = random.Random(52)
    groups, deferred = _canon_batch(rng, tcc.codec_molecule, 90)
    o = fgx_opts.defaults(kind=2, overlapping_consensus=0, **kw)
    g = GroupedReads.from_groups(groups)
    res = devemu.canon(o, True, g, deferred)
    ok = 0
    for gi, (st, recs, pre, delta) in zip(deferred, res):
        rc, want = tcc.canonicalise(o, groups[gi])
        assert st == rc
        if rc != 0:
            continue
        ok += 1
        assert recs == want and all(p == len(r) for r, p in zip(recs, pre))
    assert ok > 20
