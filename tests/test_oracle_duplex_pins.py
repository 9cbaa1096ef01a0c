"""Oracle pins: the process-level unit tests of the reference's `DuplexConsensusCaller`
(`crates/fgumi-consensus/src/duplex_caller.rs`, `#[cfg(test)] mod tests`, `:3764-4800`), transcribed case by case — the same
four record builders (`ab_r1 / ab_r2 / ba_r1 / ba_r2`, `:3764-3882`), the same constructor arguments
(`DuplexConsensusCaller::new(prefix, read group, min_reads, min_input_base_quality, per-base tags, trim, max reads per strand, cell
tag, track rejects, pre-UMI, post-UMI)`, `:449-461`), the same assertions.  `replay_cases()` hands the valid inputs to the GPU parity
suite."""
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records

F_PAIRED, F_REVERSE, F_MATE_REVERSE, F_FIRST, F_LAST = 0x1, 0x10, 0x20, 0x40, 0x80
ST_TOTAL, ST_CONSENSUS, ST_FILTERED, ST_REASON0 = 0, 1, 2, 3
Q20 = [20] * 10
A10, C10 = "A" * 10, "C" * 10


def _rec(name, seq, quals, cigar, mi, flag, pos, mate_pos, extra):
    tags = [("MI", "Z", mi), ("RG", "Z", "A")] + [(t, "Z", v) for t, v in extra]
    return bamutil.make_record(name, seq, None if quals is None else list(quals), flag=flag, ref_id=0, pos=pos, mapq=60, cigar=cigar,
                               mate_ref=0, mate_pos=mate_pos, tags=tags)


def ab_r1(name, seq, quals, cigar, mi, extra=()):  # duplex_caller.rs:3764-3792
    return _rec(name, seq, quals, cigar, mi, F_PAIRED | F_FIRST | F_MATE_REVERSE, 99, 199, extra)


def ab_r2(name, seq, quals, cigar, mi, extra=()):  # :3794-3822
    return _rec(name, seq, quals, cigar, mi, F_PAIRED | F_LAST | F_REVERSE, 199, 99, extra)


def ba_r1(name, seq, quals, cigar, mi, extra=()):  # :3824-3852
    return _rec(name, seq, quals, cigar, mi, F_PAIRED | F_FIRST | F_REVERSE, 199, 99, extra)


def ba_r2(name, seq, quals, cigar, mi, extra=()):  # :3854-3882
    return _rec(name, seq, quals, cigar, mi, F_PAIRED | F_LAST | F_MATE_REVERSE, 99, 199, extra)


def duplex_molecule(majority, gapped_a, gapped_b):  # duplex_caller.rs:3992-4047
    reads = []
    for i in range(majority):
        reads += [ab_r1(f"ab{i}", A10, Q20, "10M", "foo/A"), ab_r2(f"ab{i}", C10, Q20, "10M", "foo/A"),
                  ba_r1(f"ba{i}", C10, Q20, "10M", "foo/B"), ba_r2(f"ba{i}", A10, Q20, "10M", "foo/B")]
    if gapped_a:
        reads += [ab_r1("minority_a", A10, Q20, "4M1D6M", "foo/A"), ab_r2("minority_a", C10, Q20, "4M1D6M", "foo/A")]
    if gapped_b:
        reads += [ba_r1("minority_b", C10, Q20, "4M1D6M", "foo/B"), ba_r2("minority_b", A10, Q20, "4M1D6M", "foo/B")]
    return reads


def caller_opts(min_reads, min_input_base_quality=10, per_base_tags=False, trim=False, cell_tag=None, track_rejects=False, pre=45, post=40):
    """The arguments of `DuplexConsensusCaller::new` behind the oracle's option block: min_reads padded like fgbio's padTo(3, last)."""
    mr = list(min_reads) + [min_reads[-1]] * (3 - len(min_reads))
    kw = dict(kind=1, overlapping_consensus=0, read_name_prefix=b"consensus", read_group_id=b"RG1", cell_tag=(cell_tag or b"\0\0"),
              min_input_base_quality=min_input_base_quality, produce_per_base_tags=int(per_base_tags), trim=int(trim), track_rejects=int(track_rejects),
              error_rate_pre_umi=pre, error_rate_post_umi=post)
    o = fgx_opts.defaults(**kw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    o._kw = dict(kw, duplex_min_reads=tuple(mr))
    return o


_REPLAY = []


def call(opts, reads):
    g = GroupedReads.from_groups([reads])
    res = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)
    _REPLAY.append((dict(opts._kw), [list(reads)]))
    return res, [bamutil.parse(r) for r in split_records(res["data"])]


def reasons(res):
    import test_oracle_vanilla_pins as v
    names = v._reason_names()
    return {n: int(res["stats"][ST_REASON0 + i]) for n, i in names.items()}


def test_not_create_records_from_fragments():  # duplex_caller.rs:3884-3942
    frags = [bamutil.make_record(n, A10, [30] * 10, flag=0, ref_id=0, pos=0, mapq=60, tags=[("MI", "Z", mi)]) for n, mi in (("frag1", "foo/A"), ("frag2", "foo/B"))]
    res, _ = call(caller_opts([1]), frags)
    assert res["count"] == 0


def test_create_simple_double_stranded_consensus():  # duplex_caller.rs:3946-3982
    res, _ = call(caller_opts([1]), [ab_r1("q1", A10, Q20, "10M", "foo/A"), ab_r2("q1", C10, Q20, "10M", "foo/A"),
                                    ba_r1("q2", C10, Q20, "10M", "foo/B"), ba_r2("q2", A10, Q20, "10M", "foo/B")])
    assert res["count"] == 2


def test_single_strand_rejections_reach_the_duplex_statistics():  # duplex_caller.rs:4068-4118
    reads = duplex_molecule(3, True, False)
    res, _ = call(caller_opts([1], track_rejects=True), reads)
    want_rej = [r for r in reads if bamutil.parse(r)["name"] == "minority_a"]
    assert len(want_rej) == 2 and res["count"] == 2
    assert reasons(res)["MinorityAlignment"] == 2 and res["stats"][ST_FILTERED] == 2
    assert res["stats"][ST_TOTAL] == len(reads) and res["stats"][ST_CONSENSUS] == 2
    assert split_records(res["rejects"]) == want_rej and res["n_rejects"] == res["stats"][ST_FILTERED]


def test_zero_length_after_trimming_reads_reach_stats_and_rejects():  # duplex_caller.rs:4122-4188
    dropped = [ab_r1("trimmed", A10, [2] * 10, "10M", "foo/A"), ab_r2("trimmed", C10, [2] * 10, "10M", "foo/A")]
    reads = duplex_molecule(3, False, False) + dropped
    res, _ = call(caller_opts([1], trim=True, track_rejects=True), reads)
    assert res["count"] == 2 and reasons(res)["ZeroLengthAfterTrimming"] == 2 and res["stats"][ST_FILTERED] == 2
    assert res["stats"][ST_TOTAL] == len(reads) and res["stats"][ST_CONSENSUS] == 2
    assert split_records(res["rejects"]) == dropped and res["n_rejects"] == res["stats"][ST_FILTERED]


def test_whole_group_rejection_preserves_single_strand_reasons():  # duplex_caller.rs:4192-4238
    res, _ = call(caller_opts([4], track_rejects=True), duplex_molecule(4, False, False))
    assert res["count"] == 2          # the control: four clean templates per strand clear min_reads = 4
    reads = duplex_molecule(3, True, True)
    res, _ = call(caller_opts([4], track_rejects=True), reads)
    r = reasons(res)
    assert res["count"] == 0 and res["stats"][ST_FILTERED] == len(reads)
    assert r["MinorityAlignment"] == 4 and r["InsufficientReads"] == len(reads) - 4
    assert res["n_rejects"] == res["stats"][ST_FILTERED]


def test_whole_group_rejection_splits_all_single_strand_reasons():  # duplex_caller.rs:4242-4298
    reads = duplex_molecule(3, True, True) + [ab_r1("trimmed", A10, [2] * 10, "10M", "foo/A"), ab_r2("trimmed", C10, [2] * 10, "10M", "foo/A")]
    res, _ = call(caller_opts([4], trim=True, track_rejects=True), reads)
    r = reasons(res)
    assert res["count"] == 0 and r["MinorityAlignment"] == 4 and r["ZeroLengthAfterTrimming"] == 2
    assert r["InsufficientReads"] == len(reads) - 4 - 2 and res["stats"][ST_FILTERED] == len(reads)
    assert res["n_rejects"] == res["stats"][ST_FILTERED]


def test_whole_group_rejection_counts_are_independent_of_rejects_tracking():  # duplex_caller.rs:4302-4322
    a, _ = call(caller_opts([4], track_rejects=True), duplex_molecule(3, True, True))
    b, _ = call(caller_opts([4], track_rejects=False), duplex_molecule(3, True, True))
    assert a["stats"][ST_FILTERED] == b["stats"][ST_FILTERED] and reasons(a) == reasons(b)


def test_absent_base_qualities_abort_duplex_consensus():  # duplex_caller.rs:4326-4343
    reads = duplex_molecule(2, False, False) + [ab_r1("absent", A10, None, "10M", "foo/A"), ab_r2("absent", C10, None, "10M", "foo/A")]
    g = GroupedReads.from_groups([reads])
    with pytest.raises(RuntimeError):
        orc.process(caller_opts([1], track_rejects=True), g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)


def test_consensus_reads_mixed_paired_and_fragments():  # duplex_caller.rs:4347-4405
    frag = lambda n, mi: bamutil.make_record(n, A10, Q20, flag=0, ref_id=0, pos=99, mapq=60, tags=[("MI", "Z", mi)])
    reads = [ab_r1("q1", A10, Q20, "10M", "foo/A"), ab_r2("q1", C10, Q20, "10M", "foo/A"), ba_r1("q2", C10, Q20, "10M", "foo/B"),
             ba_r2("q2", A10, Q20, "10M", "foo/B"), frag("frag1", "foo/A"), frag("frag2", "foo/B")]
    res, _ = call(caller_opts([1]), reads)
    assert res["count"] == 2 and reasons(res)["FragmentRead"] == 2


def test_preserve_cell_barcode():  # duplex_caller.rs:4409-4448
    cb = [("CB", "ACGT")]
    res, recs = call(caller_opts([1], cell_tag=b"CB"), [ab_r1("q1", A10, Q20, "10M", "foo/A", cb), ab_r2("q1", C10, Q20, "10M", "foo/A", cb),
                                                        ba_r1("q2", C10, Q20, "10M", "foo/B", cb), ba_r2("q2", A10, Q20, "10M", "foo/B", cb)])
    assert res["count"] == 2 and all(r["tags"]["CB"][1] == "ACGT" for r in recs)


@pytest.mark.parametrize("rx_a,rx_b", [("ACT-", "-ACT"), ("-ACT", "ACT-")])
def test_handle_absent_umi(rx_a, rx_b):  # duplex_caller.rs:4452-4490 (right), 4494-4532 (left): RX inherited from the A family
    ea, eb = [("RX", rx_a)], [("RX", rx_b)]
    res, recs = call(caller_opts([1]), [ab_r1("q1", A10, Q20, "10M", "foo/A", ea), ab_r2("q1", C10, Q20, "10M", "foo/A", ea),
                                        ba_r1("q2", C10, Q20, "10M", "foo/B", eb), ba_r2("q2", A10, Q20, "10M", "foo/B", eb)])
    assert res["count"] == 2 and all(r["tags"]["RX"][1] == rx_a for r in recs)


def _single_strand(a):
    reads = []
    for i in (1, 2, 3):
        n = "q" + chr(i)          # the reference names them [b'q', i]
        reads += ([ab_r1(n, A10, Q20, "10M", "foo/A"), ab_r2(n, C10, Q20, "10M", "foo/A")] if a else
                  [ba_r1(n, C10, Q20, "10M", "foo/B"), ba_r2(n, A10, Q20, "10M", "foo/B")])
    return reads


def test_create_single_strand_consensus_a_only():  # duplex_caller.rs:4536-4580
    res, recs = call(caller_opts([1, 1, 0]), _single_strand(True))
    assert res["count"] == 2
    for r in recs:
        assert r["tags"]["aD"][1] == 3 and ("bD" not in r["tags"] or r["tags"]["bD"][1] == 0)


def test_create_single_strand_consensus_b_only():  # duplex_caller.rs:4584-4624 (the lone BA strand is reported as AB after the swap)
    res, recs = call(caller_opts([1, 1, 0]), _single_strand(False))
    assert res["count"] == 2 and all(r["tags"]["aD"][1] == 3 for r in recs if "aD" in r["tags"])


def test_reject_single_strand_when_min_reads_requires_both():  # duplex_caller.rs:4628-4660
    res, _ = call(caller_opts([1, 1, 1]), _single_strand(True))
    assert res["count"] == 0


def test_min_reads_hard_filter_after_alignment_filtering():  # duplex_caller.rs:4664-4745
    reads = []
    for i in (1, 2, 3):
        reads += [ab_r1("ab" + chr(i), A10, Q20, "10M", "foo/A"), ab_r2("ab" + chr(i), C10, Q20, "10M", "foo/A")]
    for i in (4, 5):
        reads += [ba_r1("ba" + chr(i), C10, Q20, "10M", "foo/B"), ba_r2("ba" + chr(i), A10, Q20, "10M", "foo/B")]
    assert call(caller_opts([3]), reads)[0]["count"] == 0          # only two BA templates
    assert call(caller_opts([2]), reads)[0]["count"] == 2
    dissimilar = [ba_r1("ba6", C10, Q20, "5M1D5M", "foo/B"), ba_r2("ba6", A10, Q20, "10M", "foo/B")]
    assert call(caller_opts([3]), reads + dissimilar)[0]["count"] == 0   # the dissimilar read is filtered out before the count


def replay_cases():
    """(oracle option keywords incl. `duplex_min_reads`, MI groups) of every valid run above, for the GPU parity suite."""
    if not _REPLAY:
        for name, fn in sorted(globals().items()):
            if not (name.startswith("test_") and callable(fn)):
                continue
            marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
            for a in ([a if isinstance(a, tuple) else (a,) for a in marks[0].args[1]] if marks else [()]):
                try:
                    fn(*a)
                except Exception:
                    pass
    seen, out = set(), []
    for kw, groups in _REPLAY:
        key = (tuple(sorted(kw.items())), tuple(tuple(x) for x in groups))
        if key not in seen:
            seen.add(key)
            out.append((kw, groups))
    return out


# ---- duplex_consensus on two single-strand reads (duplex_caller.rs:931-1108): the reference's direct tests --------------------------
import numpy as np


def _duplex(ab, ba):
    """ab / ba = (bases, quals, depths, errors) or None → (bases, quals, errors, has_ba, ba_only) or None."""
    def arr(v):
        if v is None:
            z8, z16 = np.zeros(1, np.uint8), np.zeros(1, np.uint16)
            return z8, z8, z16, z16, 0
        b, q, d, e = v
        return np.frombuffer(b.encode(), np.uint8).copy(), np.array(q, np.uint8), np.array(d, np.uint16), np.array(e, np.uint16), len(b)
    a, b = arr(ab), arr(ba)
    ob, oq, oe = np.zeros(64, np.uint8), np.zeros(64, np.uint8), np.zeros(64, np.uint16)
    import ctypes as C
    fl = C.c_uint32()
    n = orc.lib.orc_duplex_consensus(orc.ptr(a[0]), orc.ptr(a[1]), orc.ptr(a[2]), orc.ptr(a[3]), a[4], orc.ptr(b[0]), orc.ptr(b[1]), orc.ptr(b[2]), orc.ptr(b[3]), b[4],
                                     orc.ptr(ob), orc.ptr(oq), orc.ptr(oe), 64, C.byref(fl))
    if n < 0:
        return None
    return bytes(ob[:n]).decode(), oq[:n].tolist(), oe[:n].tolist(), bool(fl.value & 1), bool(fl.value & 2)


def test_duplex_consensus_single_strands_pass_through():  # duplex_caller.rs:5196-5238 (ab_only, ba_only), :5241-5245 (none_none)
    ab = ("ACGT", [30, 31, 32, 33], [5] * 4, [0, 1, 0, 1])
    assert _duplex(ab, None) == ("ACGT", [30, 31, 32, 33], [0, 1, 0, 1], False, False)
    ba = ("TGCA", [25, 26, 27, 28], [4] * 4, [1, 0, 1, 0])
    assert _duplex(None, ba) == ("TGCA", [25, 26, 27, 28], [1, 0, 1, 0], False, True)          # the lone BA strand becomes the AB of the output
    assert _duplex(None, None) is None


def test_duplex_consensus_different_lengths():  # duplex_caller.rs:5248-5276: the shorter strand decides
    r = _duplex(("ACGTAC", [30] * 6, [5] * 6, [0] * 6), ("ACGT", [25] * 4, [4] * 4, [0] * 4))
    assert len(r[0]) == 4 and len(r[1]) == 4


def test_duplex_consensus_error_calculation():  # duplex_caller.rs:5279-5310 (agreement), :5313-5345 (disagreement)
    r = _duplex(("ACGT", [30] * 4, [5] * 4, [1, 0, 2, 0]), ("ACGT", [25] * 4, [4] * 4, [0, 1, 0, 2]))
    assert r[2] == [1, 1, 2, 2]
    r = _duplex(("AT", [30, 40], [5, 5], [1, 2]), ("AC", [25, 30], [4, 4], [0, 1]))
    assert r[0] == "AT" and r[2][1] == 5          # AB wins position 1: its 2 errors + the 3 BA reads that agree with BA's own call


def test_cap_quality():  # duplex_caller.rs:5420-5427
    assert [orc.lib.orc_duplex_cap_quality(v) for v in (-5, 0, 2, 50, 93, 100)] == [2, 2, 2, 50, 93, 93]


def _ss(seq, quals, depth):
    """`create_ss_consensus` (duplex_caller.rs:2953-2968) as a single-strand read: depth cD at every position, no errors."""
    return (seq, list(quals), [depth] * len(seq), [0] * len(seq))


@pytest.mark.parametrize("a,b,seq,quals", [
    (_ss("AAAA", [20, 30, 40, 50], 3), _ss("AAAA", [20, 30, 40, 50], 2), "AAAA", [40, 60, 80, 93]),     # ..._quality_scores_agreement :2971-2996 (sum, capped at 93)
    (_ss("ACGT", [30] * 4, 3), _ss("TGCA", [10, 15, 20, 25], 2), "ACGT", [20, 15, 10, 5]),               # ..._disagreement_unequal :2999-3026 (higher wins, difference)
    (_ss("ACGT", [20] * 4, 3), _ss("TGCA", [20] * 4, 2), "NNNN", [2] * 4),                               # ..._disagreement_equal :3029-3054
    (_ss("AAA", [50, 60, 93], 3), _ss("AAA", [50, 60, 93], 2), "AAA", [93] * 3),                         # ..._quality_capping :3160-3183
    (_ss("ACGT", [5, 4, 3, 10], 3), _ss("TGCA", [3, 2, 2, 8], 2), "NNNN", [2] * 4),                      # ..._quality_difference_at_threshold :3186-3214 (a difference of 2 is the floor: masked)
    (_ss("AAAA", [45] * 4, 50), _ss("AAAA", [20] * 4, 1), "AAAA", [65] * 4),                             # ..._with_deep_coverage :3217-3243
    (_ss("AAAA", [25] * 4, 3), _ss("TTTT", [25] * 4, 2), "NNNN", [2] * 4),                               # ..._zero_quality_difference :3449-3474
])
def test_duplex_consensus_quality_rules(a, b, seq, quals):  # the `call_duplex_from_ss_pair` tests whose subject is the A/B combine
    r = _duplex(a, b)
    assert r[0] == seq and r[1] == quals


def test_duplex_consensus_n_bases_and_lengths():  # :3093-3125 (n_bases), :3128-3157 (length_mismatch), :3420-3446 (mixed_bases_and_n)
    r = _duplex(_ss("ANAA", [20] * 4, 3), _ss("AANA", [20] * 4, 2))
    assert r[0] == "ANNA" and r[1][1] == 2 and r[1][2] == 2
    assert _duplex(_ss("AAAA", [20] * 4, 3), _ss("AAA", [20] * 3, 2))[0] == "AAA"
    r = _duplex(_ss("ANGT", [30] * 4, 3), _ss("TNCG", [25] * 4, 2))
    assert r[0] == "ANGT" and r[1] == [5, 2, 5, 5]
