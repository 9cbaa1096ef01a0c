"""reject_core.h on the host: the `--rejects` stream of the simplex caller computed from the records alone must equal the reference
restatement's, byte for byte and in order — on simulated families (pairs, overlapping mates, families below --min-reads), on families
with indels / minority alignments / soft clips, with downsampling, quality trimming and the overlap pre-correction on and off, and on
the hostile groups of the general-path fuzz (unmapped reads, missing mates, secondary / supplementary records, zero-length reads).
This is the function the device kernels of reject_device.hip run, one lane per group."""
import ctypes as C
import random

import numpy as np
import pytest

import fgx_opts
import orc
import test_general_path_fuzz as fuzz
from fgumi_amd import GroupedReads, simulate_grouped_reads
from fgumi_amd._lib import lib


def product_rejects(o, g):
    n = C.c_uint64(0)
    cnt = C.c_uint64(0)
    args = (C.addressof(o), g.blob.ctypes.data, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.grp_first.ctypes.data, g.n_grp)
    rc = lib.fgx_simplex_rejects_host(*args, None, 0, C.addressof(n), C.addressof(cnt))
    if rc != 0:
        return rc, b"", 0
    out = np.zeros(n.value + 16, dtype=np.uint8)
    rc = lib.fgx_simplex_rejects_host(*args, out.ctypes.data, n.value, C.addressof(n), C.addressof(cnt))
    assert rc == 0
    return 0, bytes(out[:n.value]), cnt.value


def check(o, groups, must_be_in_scope=True):
    o.track_rejects = 1
    g = GroupedReads.from_groups(groups)
    try:
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=max(50, len(groups)))
    except RuntimeError:
        return None                                   # the reference refuses the batch: the general path reports it
    rc, got, cnt = product_rejects(o, g)
    if rc != 0:
        assert not must_be_in_scope
        return None
    assert cnt == want["n_rejects"], (cnt, want["n_rejects"])
    assert got == want["rejects"]
    return cnt


@pytest.mark.parametrize("kw", [dict(min_reads=1), dict(min_reads=2), dict(min_reads=3, overlapping_consensus=0), dict(min_reads=1, max_reads=2),
                                dict(min_reads=2, max_reads=3, trim=1, min_input_base_quality=25), dict(min_reads=1, min_input_base_quality=36), dict(min_reads=1, min_input_base_quality=50),
                                dict(min_reads=4, max_reads=2)])
def test_simulated_families(kw):
    total = 0
    for seed, (fs, fmax) in enumerate([(1, 0), (3, 0), (2, 9), (8, 0)]):
        sim = simulate_grouped_reads(150, family_size=fs, family_size_max=fmax, seed=40 + seed) if fmax else simulate_grouped_reads(150, family_size=fs, seed=40 + seed)
        groups = [sim.records(i) for i in range(sim.n_grp)]
        total += check(fgx_opts.defaults(kind=0, **kw), groups)
    if kw.get("min_reads", 1) > 1 or "max_reads" in kw or kw.get("min_input_base_quality", 10) > 45:
        assert total > 0


@pytest.mark.parametrize("seed", range(8))
def test_read_through_and_indel_families(seed):
    """Short inserts (mates overlap and run past each other: the pre-correction changes the rejected bytes, the mate clip the lengths)
    and families whose reads disagree on the alignment (minority alignments are rejects)."""
    import test_canon_core as tc
    rng = random.Random(900 + seed)
    sim = simulate_grouped_reads(80, family_size=4, read_length=151, insert_mean=120, insert_sd=30, seed=seed)
    groups = [sim.records(i) for i in range(sim.n_grp)]
    for g in range(80):
        m = tc.duplex_indel_molecule(rng, 5000 + g)
        if m:
            groups.append(m)                           # (MI values with /A /B suffixes are just strings to the simplex caller)
    rng.shuffle(groups)
    kw = dict(min_reads=rng.choice([1, 2, 3]), max_reads=rng.choice([-1, -1, 2, 4]), overlapping_consensus=rng.randint(0, 1), trim=rng.randint(0, 1),
              min_input_base_quality=rng.choice([0, 10, 20, 30]))
    n = check(fgx_opts.defaults(kind=0, **kw), groups)
    assert n is not None


@pytest.mark.parametrize("seed", range(40))
def test_hostile_groups(seed):
    """The fuzz's exotic groups carry CIGARs of 17 and more ops now and then (out of scope: such a group sends the batch to the general
    path); they are taken out one by one, and everything else must match as a batch."""
    rng = random.Random(7000 + seed)
    exotic = rng.random() < 0.7
    groups = [x for x in (fuzz.random_group(rng, g, "simplex", exotic) for g in range(60)) if x]
    o = fuzz.random_options(rng, "simplex")
    o.track_rejects = 1
    keep = [x for x in groups if product_rejects(o, GroupedReads.from_groups([x]))[0] == 0]
    assert len(keep) >= 0.8 * len(groups)
    if keep:
        check(o, keep)


def test_out_of_scope_and_bad_arguments():
    sim = simulate_grouped_reads(2, family_size=70, seed=3)           # 140 records per group > 128
    groups = [sim.records(i) for i in range(sim.n_grp)]
    g = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=0, min_reads=1)
    assert product_rejects(o, g)[0] == 1
    n = C.c_uint64(0)
    assert lib.fgx_simplex_rejects_host(None, None, None, None, None, 0, None, 0, C.addressof(n), C.addressof(n)) == 2


# ---- the duplex / CODEC callers' rejects (reject_core.h duplex_reject_codes / codec_reject_mask): whether a molecule gave its consensus is an
# ---- INPUT (on the device the pipeline's output slots say so); here the reference restatement, asked molecule by molecule, says so.

def strand_rejects(o, g, kept):
    n = C.c_uint64(0)
    cnt = C.c_uint64(0)
    k = np.asarray(kept, dtype=np.uint8)
    args = (C.addressof(o), g.blob.ctypes.data, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.grp_first.ctypes.data, g.n_grp, k.ctypes.data)
    rc = lib.fgx_strand_rejects_host(*args, None, 0, C.addressof(n), C.addressof(cnt))
    if rc != 0:
        return rc, b"", 0
    out = np.zeros(n.value + 16, dtype=np.uint8)
    rc = lib.fgx_strand_rejects_host(*args, out.ctypes.data, n.value, C.addressof(n), C.addressof(cnt))
    assert rc == 0
    return 0, bytes(out[:n.value]), cnt.value


def check_strand(o, groups, must_be_in_scope=True):
    """The batch's rejects (a CODEC batch: one oracle run — its running name counter spans the batch; the per-molecule runs only say who was kept)."""
    o.track_rejects = 1
    kept, want, n_want = [], [], 0
    for m in groups:
        g1 = GroupedReads.from_groups([m])
        try:
            w = orc.process(o, g1.blob, g1.rec_off, g1.rec_len, g1.grp_first, batch_groups=50)
        except RuntimeError:
            return None                               # the reference refuses a molecule: the general path reports it
        kept.append(1 if w["count"] else 0)
        want.append(w["rejects"])
        n_want += w["n_rejects"]
    g = GroupedReads.from_groups(groups)
    rc, got, cnt = strand_rejects(o, g, kept)
    if rc != 0:
        assert not must_be_in_scope
        return None
    assert cnt == n_want, (cnt, n_want)
    assert got == b"".join(want)
    return cnt, sum(kept)


@pytest.mark.parametrize("kw", [dict(), dict(duplex_min_reads=(3, 2, 1)), dict(duplex_min_reads=(6, 3, 3), overlapping_consensus=0), dict(min_input_base_quality=36),
                                dict(duplex_min_reads=(1, 1, 1), trim=1, min_input_base_quality=25), dict(duplex_min_reads=(2, 1, 0), min_input_base_quality=50)])
def test_duplex_simulated_molecules(kw):
    kw = dict(kw)
    mr = kw.pop("duplex_min_reads", None)
    total = kept = 0
    for seed, (fs, fmax, ins) in enumerate([(2, 0, 300), (3, 8, 120), (6, 0, 160), (1, 4, 300)]):
        sim = simulate_grouped_reads(120, family_size=fs, family_size_max=fmax, duplex=1, read_length=151, insert_mean=ins, insert_sd=30, seed=70 + seed)
        groups = [sim.records(i) for i in range(sim.n_grp)]
        o = fgx_opts.defaults(kind=1, **kw)
        if mr:
            o.duplex_min_reads = (C.c_uint32 * 3)(*mr)
        r = check_strand(o, groups)
        assert r is not None
        total += r[0]; kept += r[1]
    assert kept > 0
    if mr and mr[0] > 1 or kw.get("min_input_base_quality", 10) > 45:
        assert total > 0


@pytest.mark.parametrize("seed", range(8))
def test_duplex_indel_molecules_and_fragments(seed):
    """Molecules whose reads disagree on the alignment (minority alignments are rejects of a KEPT molecule, written after its zero-length reads) with
    fragment reads thrown in (always rejects, written first)."""
    import test_canon_core as tc
    rng = random.Random(1900 + seed)
    groups = []
    for g in range(120):
        m = tc.duplex_indel_molecule(rng, 9000 + g)
        if m:
            groups.append(m)
    sim = simulate_grouped_reads(60, family_size=3, duplex=1, read_length=100, insert_mean=140, insert_sd=25, seed=seed)
    for i in range(sim.n_grp):
        m = sim.records(i)
        if rng.random() < 0.5:                            # a fragment: the first record with the pairing flags taken off
            r = bytearray(m[0])
            fl = int.from_bytes(r[14:16], "little") & ~(0x1 | 0x2 | 0x8 | 0x20 | 0x40 | 0x80)
            r[14:16] = fl.to_bytes(2, "little")
            m = m[:1] + [bytes(r)] + m[1:]
        groups.append(m)
    rng.shuffle(groups)
    o = fgx_opts.defaults(kind=1, overlapping_consensus=rng.randint(0, 1), trim=rng.randint(0, 1), min_input_base_quality=rng.choice([0, 10, 20, 30]))
    o.duplex_min_reads = (C.c_uint32 * 3)(*rng.choice([(1, 1, 0), (2, 1, 1), (4, 2, 1)]))
    r = check_strand(o, groups)
    assert r is not None and r[0] > 0


@pytest.mark.parametrize("kind", ["duplex", "codec"])
@pytest.mark.parametrize("seed", range(25))
def test_strand_callers_hostile_groups(kind, seed):
    rng = random.Random(8000 + seed + (500 if kind == "codec" else 0))
    exotic = rng.random() < 0.7
    groups = [x for x in (fuzz.random_group(rng, g, kind, exotic) for g in range(50)) if x]
    o = fuzz.random_options(rng, kind)
    o.methylation_mode = 0
    o.track_rejects = 1
    if kind == "codec":
        o.codec_max_reads_per_strand = -1                 # (a cap is out of the side function's scope)
    keep = []
    for x in groups:
        g1 = GroupedReads.from_groups([x])
        try:
            w = orc.process(o, g1.blob, g1.rec_off, g1.rec_len, g1.grp_first, batch_groups=50)
        except RuntimeError:
            continue
        if strand_rejects(o, g1, [1 if w["count"] else 0])[0] == 0:
            keep.append(x)
    assert len(keep) >= 0.6 * len(groups)
    if keep:
        check_strand(o, keep)


@pytest.mark.parametrize("kw", [dict(), dict(codec_min_reads_per_strand=3), dict(codec_min_duplex_length=200)])
def test_codec_simulated_molecules(kw):
    total = kept = 0
    for seed, fs in enumerate([1, 2, 4]):
        sim = simulate_grouped_reads(100, family_size=fs, read_length=150, insert_mean=200, insert_sd=40, codec=1, seed=90 + seed)
        groups = [sim.records(i) for i in range(sim.n_grp)]
        o = fgx_opts.defaults(kind=2)
        for k, v in kw.items():
            setattr(o, k, v)
        r = check_strand(o, groups)
        assert r is not None
        total += r[0]; kept += r[1]
    if kw:
        assert total > 0
    else:
        assert kept > 0
