"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the
same inputs.  Bit-exact for every byte (integer/byte work; f64 column math must match glibc)."""
import ctypes as C
import math
import random

import numpy as np
import pytest

import bamutil
import cases
import fgx_opts
import orc
from fgumi_amd import (GroupedReads, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads, split_records)
from fgumi_amd._lib import default_options, lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    o = default_options()
    h = lib.fgx_create(C.byref(o))
    assert h, lib.fgx_global_error().decode()
    yield h
    lib.fgx_destroy(h)


# ---- device libm ≡ glibc ----------------------------------------------------------------------
@pytest.mark.parametrize("op,fn,ranges", [
    (0, math.exp, [(-40.0, 20.0), (-745.5, -700.0), (700.0, 710.0), (-1e-3, 1e-3)]),
    (1, math.log, [(1e-300, 1e-290), (1e-9, 10.0), (0.93, 1.07), (1.0, 1e9)]),
    (2, math.log1p, [(-0.99999, 2.0), (-1e-6, 1e-6), (1.0, 1e12)]),
    (3, math.expm1, [(-0.7, 0.7), (-40.0, 5.0), (-1e-9, 1e-9)]),
])
def test_device_libm_bit_exact(handle, op, fn, ranges):
    rng = np.random.default_rng(op + 11)
    x = np.concatenate([rng.uniform(lo, hi, 250000) for lo, hi in ranges] + [np.array([0.0, -0.0, 1.0, -1.0 if op in (0, 3) else 0.5])])
    y = np.zeros_like(x)
    assert lib.fgx_device_libm(handle, op, x.ctypes.data, y.ctypes.data, x.size) == 0
    libm = C.CDLL("libm.so.6")   # the box's real glibc, not Python's wrappers (which raise on log(0), exp overflow)
    f = getattr(libm, fn.__name__)
    f.restype, f.argtypes = C.c_double, [C.c_double]
    ref = np.array([f(float(v)) for v in x])
    bad = np.nonzero((y.view(np.uint64) != ref.view(np.uint64)) & ~(np.isnan(y) & np.isnan(ref)))[0]
    assert bad.size == 0, f"{bad.size} mismatches, first x={x[bad[0]]!r} got={y[bad[0]]!r} want={ref[bad[0]]!r}"


def test_device_tables_equal_oracle(handle):
    b = orc.Builder(45, 40)
    for which in range(4):
        want, cap = b.table(which)
        got = np.zeros(94)
        gcap = C.c_uint32()
        lib.fgx_get_table(handle, which, got.ctypes.data, C.byref(gcap))
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and gcap.value == cap


# ---- column level: ConsensusBaseBuilder -----------------------------------------------------
def _columns(handle, bases, quals, pre=45, post=40, tie=0):
    n_cols, depth = bases.shape
    o = default_options(error_rate_pre_umi=pre, error_rate_post_umi=post, tie_rule=tie)
    h = lib.fgx_create(C.byref(o))
    assert h
    try:
        ob, oq = np.zeros(n_cols, np.uint8), np.zeros(n_cols, np.uint8)
        od, oe = np.zeros(n_cols, np.uint32), np.zeros(n_cols, np.uint32)
        b, q = np.ascontiguousarray(bases), np.ascontiguousarray(quals)
        assert lib.fgx_call_columns(h, b.ctypes.data, q.ctypes.data, n_cols, depth, ob.ctypes.data, oq.ctypes.data, od.ctypes.data, oe.ctypes.data) == 0
    finally:
        lib.fgx_destroy(h)
    wb, wq = np.zeros(n_cols, np.uint8), np.zeros(n_cols, np.uint8)
    wd, we = np.zeros(n_cols, np.uint32), np.zeros(n_cols, np.uint32)
    orc.lib.orc_call_columns(pre, post, tie, orc.ptr(b), orc.ptr(q), n_cols, depth, orc.ptr(wb), orc.ptr(wq), orc.ptr(wd), orc.ptr(we))
    return (ob, oq, od, oe), (wb, wq, wd, we)


@pytest.mark.parametrize("pre,post,tie", [(45, 40, 0), (45, 40, 1), (90, 90, 0), (93, 93, 0), (20, 10, 0), (70, 5, 0), (2, 2, 0), (50, 50, 0)])
def test_random_columns_match_oracle(handle, pre, post, tie):
    rng = np.random.default_rng(pre * 100 + post + tie)
    for depth in [1, 2, 3, 4, 8, 16, 50]:
        n = 20000
        truth = rng.integers(0, 4, n)
        err = rng.random((n, depth)) < 0.08
        alt = rng.integers(0, 4, (n, depth))
        idx = np.where(err, alt, truth[:, None])
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[idx]
        bases = np.where(rng.random((n, depth)) < 0.03, ord("N"), bases).astype(np.uint8)
        quals = rng.integers(0, 60, (n, depth)).astype(np.uint8)
        quals[rng.random((n, depth)) < 0.3] = 37            # equal qualities provoke near-ties
        got, want = _columns(handle, bases, quals, pre, post, tie)
        for g, w, what in zip(got, want, ("base", "qual", "depth", "errors")):
            bad = np.nonzero(g != w)[0]
            assert bad.size == 0, f"{what}: {bad.size} mismatches at depth {depth}; first col {bad[0]}: bases={bases[bad[0]].tobytes()} quals={quals[bad[0]].tolist()} got={g[bad[0]]} want={w[bad[0]]}"


def test_reference_column_pins_on_device(handle):
    # base_builder.rs:2500-2526, 1510-1603, 2682-2708 replayed through the kernel
    def one(obs, pre, post, tie=0):
        bases = np.array([[ord(b) for b, _ in obs]], dtype=np.uint8)
        quals = np.array([[q for _, q in obs]], dtype=np.uint8)
        got, want = _columns(handle, bases, quals, pre, post, tie)
        assert all(int(g[0]) == int(w[0]) for g, w in zip(got, want))
        return chr(got[0][0]), int(got[1][0])

    for pre, post, obs, depth, exp in [(45, 2, 2, 50, 16), (70, 5, 5, 15, 65), (70, 5, 5, 40, 70), (93, 40, 20, 3, 69), (20, 10, 10, 4, 19),
                                       (45, 40, 40, 50, 45), (93, 93, 93, 100, 93), (2, 2, 2, 5, 2)]:
        assert one([("A", obs)] * depth, pre, post) == ("A", exp)
    assert one([("C", 37), ("C", 37), ("T", 37), ("T", 37)], 45, 40, 0) == ("T", 3)
    assert one([("C", 37), ("C", 37), ("T", 37), ("T", 37)], 45, 40, 1) == ("N", 2)
    assert one([("A", 20), ("C", 20)], 93, 93) == ("N", 2)
    assert one([("A", 30), ("A", 30)], 45, 40) == ("A", 44)
    assert one([("A", 30), ("A", 30), ("C", 0)], 45, 40) == ("A", 44)
    assert one([("C", 20)] * 1000 + [("T", 20)] * 10, 50, 50) == ("C", 50)
    assert one([("A", 20)], 50, 50) == ("A", 20)


def test_deep_unanimous_columns_match_oracle(handle):
    # contiguous depths straddle the gap-table bracket edges (base_builder.rs:2042-2083)
    for pre, post, q in [(45, 3, 3), (70, 5, 7), (90, 93, 93), (93, 93, 40)]:
        depths = list(range(1, 400))
        D = max(depths)
        bases = np.full((len(depths), D), ord("N"), dtype=np.uint8)
        for i, d in enumerate(depths):
            bases[i, :d] = ord("G")
        quals = np.full((len(depths), D), q, dtype=np.uint8)
        got, want = _columns(handle, bases, quals, pre, post)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)


# ---- whole batches: process_fn ------------------------------------------------------------------
def _oracle(grouped, **optkw):
    o = fgx_opts.defaults(**optkw)
    return orc.process(o, grouped.blob, grouped.rec_off, grouped.rec_len, grouped.grp_first)


MODE = {"general_only": False}


@pytest.fixture(params=["fast+general", "general-only"], autouse=True)
def _path_mode(request):
    """Every batch test runs twice: through the device-resident fast path (general path only for the
    families it defers) and with every family forced through the general host-orchestrated path."""
    MODE["general_only"] = request.param == "general-only"
    yield


def _device(grouped, min_reads=1, overlapping=True, track_rejects=False, prefix="", **optkw):
    vo = VanillaUmiConsensusOptions(min_reads=min_reads, min_consensus_base_quality=optkw.pop("min_consensus_base_quality", 2),
                                    cell_tag="CB", **optkw)
    c = VanillaUmiConsensusCaller(prefix, "A", vo, track_rejects=track_rejects, overlapping_consensus=overlapping)
    c.set_general_only(MODE["general_only"])
    out = c.process_batch(grouped)
    stats = c.last_batch_statistics()
    rej = c.take_rejected_reads()
    c.close()
    return out, stats, rej


def _assert_same(grouped, min_reads=1, overlapping=True, track_rejects=False, **kw):
    okw = dict(min_reads=min_reads, overlapping_consensus=int(overlapping), track_rejects=int(track_rejects))
    for k, v in kw.items():
        if k == "max_reads":
            okw["max_reads"] = -1 if v is None else v
        elif k == "trim":
            okw["trim"] = int(v)
        elif k == "prefix":
            okw["read_name_prefix"] = v.encode()
        elif k == "produce_per_base_tags":
            okw[k] = int(v)
        else:
            okw[k] = v
    want = _oracle(grouped, **okw)
    out, stats, rej = _device(grouped, min_reads=min_reads, overlapping=overlapping, track_rejects=track_rejects, **kw)
    assert out.count == want["count"]
    if out.data != want["data"]:
        a, b = split_records(out.data), split_records(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError("record count/length differs")
    arr = np.zeros(28, dtype=np.uint64)
    arr[0], arr[1], arr[2] = stats.total_reads, stats.consensus_reads, stats.filtered_reads
    for r, v in stats.rejection_reasons.items():
        arr[3 + int(r)] = v
    arr[24:28] = [stats.overlapping[k] for k in ("overlapping_bases", "bases_agreeing", "bases_disagreeing", "bases_corrected")]
    assert np.array_equal(arr, want["stats"]), (arr.tolist(), want["stats"].tolist())
    if track_rejects:
        assert rej == split_records(want["rejects"])
    return out, want


def test_config1_simplex_10k_families_depth3(handle):
    """BASELINE.json configs[0]: 10k families, depth 3, 150 bp — byte-identical ConsensusOutput."""
    g = simulate_grouped_reads(10000, family_size=3)
    out, want = _assert_same(g, min_reads=1)
    assert out.count == 20000


@pytest.mark.parametrize("kw", [dict(min_reads=2), dict(min_reads=3), dict(overlapping=False), dict(min_reads=1, track_rejects=True),
                                dict(min_reads=2, track_rejects=True, min_consensus_base_quality=40), dict(trim=True),
                                dict(max_reads=2, track_rejects=True), dict(min_reads=2, max_reads=2), dict(produce_per_base_tags=False),
                                dict(error_rate_pre_umi=30, error_rate_post_umi=20), dict(tie_rule=1), dict(min_input_base_quality=30)])
def test_simplex_option_matrix(handle, kw):
    g = simulate_grouped_reads(600, family_size=4, error_rate_ppm=20000)
    _assert_same(g, **kw)


def test_simplex_depth8_and_long_tail(handle):
    _assert_same(simulate_grouped_reads(2000, family_size=8))
    _assert_same(simulate_grouped_reads(1500, family_size=2, family_size_max=50), min_reads=2)
    _assert_same(simulate_grouped_reads(300, family_size=1), min_reads=1)                 # single-read LUT path
    _assert_same(simulate_grouped_reads(300, family_size=3, read_length=300, insert_mean=350, insert_sd=60))
    _assert_same(simulate_grouped_reads(300, family_size=3, read_length=151, insert_mean=120, insert_sd=30))   # read-through clips


def test_reference_caller_unit_test_inputs(handle):
    """The inputs of the reference's own caller-level unit tests (vanilla_caller.rs `mod tests`; their assertions are replayed on the
    oracle in tests/test_oracle_vanilla_pins.py) through the HIP path: byte-identical records and statistics — including the
    40 000-read family whose depths saturate at i16::MAX, pairs without MC tags, indel minorities and orphan ends."""
    import test_oracle_vanilla_pins as pins
    cases = pins.replay_cases()
    assert len(cases) >= 25
    for kw, groups in cases:
        o = dict(kw)
        _assert_same(GroupedReads.from_groups(groups), min_reads=o.pop("min_reads"), overlapping=bool(o.pop("overlapping_consensus")),
                     track_rejects=bool(o.pop("track_rejects", 0)), prefix=o.pop("read_name_prefix").decode(),
                     produce_per_base_tags=bool(o.pop("produce_per_base_tags", 1)), trim=bool(o.pop("trim", 0)),
                     **{k: v for k, v in o.items() if k != "cell_tag"})


@pytest.mark.parametrize("read_length", [20, 32, 33, 64, 65, 96, 97, 129, 160, 161, 200])
def test_column_pass_schedule_over_read_lengths(handle, read_length):
    """k_simplex_wave2 walks 64 columns per pass and takes the two tails of a pair family in ONE merged pass when both are at most
    32 columns long: lengths on either side of every boundary (no full pass at all, tail of exactly 32 / 33, no tail), with mates
    that overlap (lengths change with the mate clip) and mates that do not, at depths 2 and 5."""
    for insert_mean in (read_length * 3, read_length + read_length // 3):
        for depth in (2, 5):
            g = simulate_grouped_reads(120, family_size=depth, read_length=read_length, insert_mean=insert_mean, insert_sd=max(2, read_length // 8),
                                       error_rate_ppm=15000, seed=read_length * 7 + depth)
            _assert_same(g, min_reads=1)
    _assert_same(simulate_grouped_reads(120, family_size=4, read_length=read_length, insert_mean=read_length * 3, insert_sd=5, seed=read_length), min_reads=2,
                 overlapping=False, min_input_base_quality=25)


@pytest.mark.parametrize("kw", [dict(min_reads=1), dict(min_reads=3), dict(min_reads=40), dict(overlapping=False, min_reads=2), dict(trim=True), dict(max_reads=20),
                                dict(min_reads=2, min_input_base_quality=30, produce_per_base_tags=False)])
def test_large_families_workgroup_kernel(handle, kw):
    """66..128 records per family: beyond one wavefront, the workgroup-per-family kernel (parallel gates, ballot-built member
    lists, wave-parallel mate pairing) decides them."""
    _assert_same(simulate_grouped_reads(150, family_size=33, family_size_max=64, error_rate_ppm=5000), **kw)


def test_crafted_edge_cases(handle):
    groups = cases.crafted_groups()
    g = GroupedReads.from_groups(groups)
    for mr in (1, 2):
        for ov in (True, False):
            _assert_same(g, min_reads=mr, overlapping=ov, track_rejects=True, prefix="lib1")
    _assert_same(g, min_reads=1, trim=True, track_rejects=True)
    _assert_same(g, min_reads=1, max_reads=1, track_rejects=True)


def test_device_resident_pipeline_matches_oracle(handle):
    """Input generated in HBM by the device generator, consensus records left in HBM: the measured
    configuration of bench.py.  The same molecules regenerated on the host feed the oracle."""
    for kw in [dict(n_families=3000, family_size=8), dict(n_families=2000, family_size=2, family_size_max=40),
               dict(n_families=500, family_size=3, read_length=300, insert_mean=350, insert_sd=60)]:
        c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"),
                                      overlapping_consensus=True)
        dg = c.simulate_on_device(**kw)
        out = c.process_batch_device(dg)
        assert out.n_deferred == 0
        data = out.to_host()
        g = simulate_grouped_reads(**kw)
        want = _oracle(g, min_reads=1)
        assert out.count == want["count"] and data == want["data"]
        assert bytes(out.as_tensor().cpu().numpy()) == data                      # zero-copy tensor view of the records in HBM (what a collective takes)
        st = c.last_batch_statistics()
        assert st.total_reads == int(want["stats"][0]) and st.consensus_reads == int(want["stats"][1])
        assert st.overlapping["bases_corrected"] == int(want["stats"][27])
        # repeated passes over the same resident input are identical (input is never modified)
        out2 = c.process_batch_device(dg)
        assert out2.to_host() == data
        c.close()


def test_fatal_errors_match_reference(handle):
    # missing MI tag (vanilla_caller.rs:1901-1904), absent qualities (:1119-1124), over-long read name (:1795-1797)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1))
    with pytest.raises(RuntimeError, match="Missing UMI tag"):
        c.consensus_reads([bamutil.make_record("x", "ACGT", [30] * 4)])
    with pytest.raises(RuntimeError, match="missing base qualities"):
        c.consensus_reads([bamutil.make_record("x", "ACGT", None, tags=[("MI", "Z", "1")])])
    with pytest.raises(RuntimeError, match="read name"):
        c.consensus_reads([bamutil.frag("x", "ACGT", 30, "U" * 300)])
    assert c.consensus_reads([]).count == 0
    c.close()


def _clipped_groups():
    """Families whose reads carry soft / hard clips around ONE aligned block (what an aligner gives a read without indels):
    overlapping mates, clips on either end and strand, =/X ops, MC tags with clips, fragments."""
    import random
    rng = random.Random(11)
    tmpl = "".join(rng.choice("ACGT") for _ in range(600))
    groups = []

    def pr(name, p1, c1, p2, c2, q1=32, q2=30, mi="m", **kw):
        def qlen(c):
            return sum(o >> 4 for o in bamutil.cigar_ops(c) if (o & 15) in (0, 1, 4, 7, 8))
        return list(bamutil.pair(name, tmpl[p1:p1 + qlen(c1)], q1, tmpl[p2:p2 + qlen(c2)], q2, mi, pos1=1000 + p1, pos2=1000 + p2, cigar1=c1, cigar2=c2, **kw))

    groups.append(pr("a", 0, "5S95M", 60, "90M10S", mi="c1") + pr("b", 0, "5S95M", 60, "90M10S", q1=35, q2=28, mi="c1") + pr("c", 0, "100M", 60, "100M", mi="c1"))
    groups.append(pr("a", 10, "3H4S50=1X45=2S", 40, "2S98M5H", mi="c2", rx="ACGT-AAAA") + pr("b", 10, "7S96M2S", 40, "2S98M", mi="c2", rx="ACGT-AAAT"))
    groups.append(pr("a", 0, "20S80M", 30, "70M30S", mi="c3") + pr("b", 0, "100M", 30, "100M", mi="c3") + pr("c", 2, "10S90M", 35, "95M5S", mi="c3"))
    groups.append([bamutil.frag("f1", tmpl[:80], 33, "c4", cigar="10S70M"), bamutil.frag("f2", tmpl[:80], 30, "c4", cigar="80M"),
                   bamutil.frag("f3", tmpl[:80], 31, "c4", cigar="5H10S60M10S", flag=0x10)])
    groups.append(pr("a", 100, "50M50S", 110, "40S60M", mi="c5") + pr("b", 100, "100M", 110, "100M", mi="c5"))      # read-through shaped clips
    return GroupedReads.from_groups(groups)


@pytest.mark.parametrize("kw", [dict(), dict(min_reads=2), dict(overlapping=False), dict(trim=True, min_input_base_quality=31)])
def test_clipped_single_block_cigars(kw):
    _assert_same(_clipped_groups(), **kw)


def test_clipped_single_block_cigars_stay_on_the_device():
    if MODE["general_only"]:
        pytest.skip("device-resident entry only")
    g = _clipped_groups()
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    out = c.process_batch_device(g.to_device())
    want = _oracle(g, min_reads=1)
    assert out.n_deferred == 0
    assert out.count == want["count"] and out.to_host() == want["data"]
    c.close()


def test_full_size_config2_properties():
    """BASELINE.json configs[1] at full size (5 M families x 8 pairs x 150 bp = 80 M reads, 26 GB of records in HBM): the properties that
    do not need the oracle at that size.  (1) Sharding invariance, a checksum of checksums: the batch cut into five 1 M-family shards
    gives outputs whose concatenation IS the output of the whole batch (byte for byte), and whose counters add up; (2) idempotence: a
    second pass over the same resident input is identical; (3) the accounting identities of the simulated shape.  Small-scale parity
    against the oracle (same generator, same kernels) is what the rest of this file establishes."""
    import os
    if os.environ.get("FGX_SKIP_FULL_SIZE"):
        pytest.skip("FGX_SKIP_FULL_SIZE set")
    import torch
    fam, shard = 5_000_000, 1_000_000
    if torch.cuda.mem_get_info()[1] < 120 * 2**30:
        pytest.skip("needs a 288 GB-class GPU")
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    dg = c.simulate_on_device(fam, family_size=8)
    out = c.process_batch_device(dg)
    assert out.n_deferred == 0 and out.count == 2 * fam
    st = c.last_batch_statistics()
    assert st.total_reads == 16 * fam and st.consensus_reads == 2 * fam and st.filtered_reads == 0
    full = out.to_host()
    again = c.process_batch_device(dg)
    assert again.data_len == len(full) and again.to_host() == full                     # idempotent, input untouched
    del dg, again
    torch.cuda.empty_cache()
    off = 0
    total_reads = 0
    overlap = 0
    for k in range(fam // shard):
        dk = c.simulate_on_device(shard, family_size=8, first_family=k * shard)
        ok = c.process_batch_device(dk)
        part = ok.to_host()
        assert full[off:off + len(part)] == part, f"shard {k} differs from its slice of the whole batch"
        off += len(part)
        sk = c.last_batch_statistics()
        total_reads += sk.total_reads
        overlap += sk.overlapping["bases_corrected"]
        del dk
    assert off == len(full) and total_reads == st.total_reads and overlap == st.overlapping["bases_corrected"]
    c.close()



def test_noisy_batch_grows_the_call_full_pool_instead_of_deferring(monkeypatch):
    """3 % substitution errors at depth 8: a fifth of the columns needs `call_full`.  With a pool far too small for that (test knobs)
    every list fills up; the batch must be run again with more room until nothing is deferred — and still equal the oracle."""
    monkeypatch.setenv("FGX_POOL_DIV", "4096")
    monkeypatch.setenv("FGX_POOL_SLACK", "1")
    g = simulate_grouped_reads(3000, family_size=8, error_rate_ppm=30000)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    out = c.process_batch_device(g.to_device())
    assert out.n_deferred == 0
    assert out.to_host() == want["data"]
    c.close()


def test_long_read_name_prefix_takes_the_per_field_writer():
    """A read-name prefix of 70 characters: names no longer fit the lanes of the one-store field writers (k_emit's pair writer and its per-record fast
    path refuse the record) — emit_generic, field after field."""
    g = simulate_grouped_reads(300, family_size=5)
    _assert_same(g, prefix="p" * 70)
