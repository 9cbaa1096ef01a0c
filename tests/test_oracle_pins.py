"""Pins the ORACLE against the known-answer vectors the reference's own unit tests hold
(SURVEY.md §8c).  Each test cites the reference test it replays."""
import ctypes as C
import math
import struct

import numpy as np
import pytest

import orc
from orc import Builder, lib, ptr

INF = float("inf")
EPS = 2.220446049250313e-16


def bits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


# ---- phred.rs ----------------------------------------------------------------------------
def test_phred_conversions_fgbio():  # phred.rs:686-702
    assert lib.orc_ln_prob_to_phred(-INF) == 93
    assert lib.orc_ln_prob_to_phred(math.log(0.1)) == 10
    assert lib.orc_ln_prob_to_phred(math.log(0.5)) == 3
    assert lib.orc_ln_prob_to_phred(0.0) == 2
    assert lib.orc_ln_prob_to_phred(math.log(0.01)) == 20
    assert lib.orc_ln_prob_to_phred(math.log(0.001)) == 30
    assert lib.orc_ln_prob_to_phred(math.log(1e-20)) == 93


def test_error_two_trials_comprehensive_fgbio():  # phred.rs:491-513
    for i in range(1, 101):
        for j in range(1, 101):
            p1, p2 = 1.0 / i, 1.0 / j
            exp = p1 * (1 - p2) + (1 - p1) * p2 + p1 * p2 * (2.0 / 3.0)
            got = lib.orc_ln_error_prob_two_trials(math.log(p1), math.log(p2))
            assert abs(math.exp(got) - exp) < 1e-4


def test_ln_sum_exp_fgbio():  # phred.rs:516-541
    assert abs(math.exp(lib.orc_ln_sum_exp(10.0, 10.0)) - math.exp(10) * 2) < 1e-5
    assert abs(math.exp(lib.orc_ln_sum_exp(10.0, 20.0)) - (math.exp(10) + math.exp(20))) < 1e-5
    assert abs(math.exp(lib.orc_ln_sum_exp(20.0, 10.0)) - (math.exp(10) + math.exp(20))) < 1e-5
    assert abs(math.exp(lib.orc_ln_sum_exp(10.0, -INF)) - math.exp(10)) < 1e-5
    assert abs(math.exp(lib.orc_ln_sum_exp(-INF, 10.0)) - math.exp(10)) < 1e-5
    assert abs(lib.orc_ln_sum_exp(-718.3947756282423, -8.404216861178751) + 8.404216861178751) < 1e-5


def test_ln_a_minus_b_fgbio():  # phred.rs:545-563, 583-620
    q10, q20 = lib.orc_phred_to_ln_error_prob(10), lib.orc_phred_to_ln_error_prob(20)
    out = C.c_double()
    assert lib.orc_ln_a_minus_b(10.0, 10.0, C.byref(out)) == 0 and out.value == -INF
    assert lib.orc_ln_a_minus_b(q10, q10, C.byref(out)) == 0 and out.value == -INF
    assert lib.orc_ln_a_minus_b(q10, q20, C.byref(out)) == 0 and abs(math.exp(out.value) - 0.09) < 1e-5
    assert lib.orc_ln_a_minus_b(math.log(10.0), -INF, C.byref(out)) == 0 and abs(out.value - math.log(10)) < 1e-5
    # a < b within EPSILON → -inf ; beyond → the reference panics
    assert lib.orc_ln_a_minus_b(1.0, 1.0 + EPS / 4, C.byref(out)) == 0 and out.value == -INF
    assert lib.orc_ln_a_minus_b(1.0, 2.0, C.byref(out)) == 1


def test_ln_one_minus_exp_fgbio():  # phred.rs:655-683
    q10, q20 = lib.orc_phred_to_ln_error_prob(10), lib.orc_phred_to_ln_error_prob(20)
    assert abs(math.exp(lib.orc_ln_not(q10)) - 0.9) < 1e-5
    assert abs(math.exp(lib.orc_ln_not(q20)) - 0.99) < 1e-5
    assert abs(math.exp(lib.orc_ln_not(math.log(0.90))) - 0.1) < 1e-5
    assert abs(math.exp(lib.orc_ln_not(math.log(0.99))) - 0.01) < 1e-5
    assert abs(math.exp(lib.orc_ln_not(-INF)) - 1.0) < 1e-5


def test_log1pexp_zero_constant_is_bit_exact():  # phred.rs:880-887
    assert bits(lib.orc_log1pexp(0.0)) == bits(math.log(2.0)) == bits(0.6931471805599453)


@pytest.mark.parametrize("a", [0.0, -0.0, -2.5, -1e-8, -700.0, 3.25, 5e-324, 1.7976931348623157e308, -1.7976931348623157e308])
def test_ln_sum_exp_equal_inputs_matches_general_path(a):  # phred.rs:893-912
    expected = a + lib.orc_log1pexp(a - a)
    assert bits(lib.orc_ln_sum_exp(a, a)) == bits(expected)


@pytest.mark.parametrize("a,b", [(0.0, 0.0), (-2.5, -2.5), (-700.0, -700.0), (-5e-324, -5e-324), (-3.0, -1.0), (-1.0, -3.0),
                                 (-2.5, -2.5000001), (-0.5, -650.0), (-INF, -3.0), (-3.0, -INF), (-INF, -INF)])
def test_ln_sum_exp_matches_fgbio_or_baseline(a, b):  # phred.rs:931-977
    def l1pe(v):
        if v <= -37.0:
            return math.exp(v)
        if v <= 18.0:
            return math.log1p(math.exp(v))
        if v <= 33.3:
            return v + math.exp(-v)
        return v

    def f_or(x, y):
        if x == -INF:
            return y
        if y == -INF:
            return x
        if y < x:
            return f_or(y, x)
        return x + l1pe(y - x)

    assert bits(lib.orc_ln_sum_exp(a, b)) == bits(f_or(a, b))


def test_ln_sum_exp_signed_zero_and_neg_inf():  # phred.rs:982-1003
    for a, b in [(0.0, -0.0), (-0.0, 0.0)]:
        assert bits(lib.orc_ln_sum_exp(a, b)) == bits(a + lib.orc_log1pexp(b - a))
    assert lib.orc_ln_sum_exp(-INF, -INF) == -INF
    assert lib.orc_ln_sum_exp(-INF, -3.0) == -3.0 and lib.orc_ln_sum_exp(-3.0, -INF) == -3.0


def test_ln_sum_exp_array_skips_neg_inf_lane():  # phred.rs:357-384 doc
    v = np.array([-1.0, -INF, -2.0, -3.0])
    got = lib.orc_ln_sum_exp_array(ptr(v), 4)
    assert abs(got - math.log(math.exp(-1) + math.exp(-2) + math.exp(-3))) < 1e-12
    v = np.array([-INF] * 4)
    assert lib.orc_ln_sum_exp_array(ptr(v), 4) == -INF


# ---- base_builder.rs: tie rules ------------------------------------------------------------
def _ll(v):
    return np.array(v, dtype=np.float64)


def step_away_from_zero(at, ulps):
    m = struct.unpack("<d", struct.pack("<Q", bits(abs(at)) + ulps))[0]
    return -m if at < 0 else m


def one_ulp(x):
    return abs(step_away_from_zero(x, 1) - x)


@pytest.mark.parametrize("ll,exp", [([-500.0, -500.0, -600.0, -700.0], -1), ([-700.0, -600.0, -500.0, -500.0], -1),
                                    ([-500.0] * 4, -1), ([-100.0, -600.0, -500.0, -700.0], 0), ([-600.0, -500.0, -700.0, -100.0], 3),
                                    ([-0.01, -1.0, -2.0, -3.0], 0), ([float("nan"), -3.0, float("nan"), -9.0], 1)])
def test_unambiguous_pileups_match_fgbio(ll, exp):  # base_builder.rs:2566-2580
    a = _ll(ll)
    assert lib.orc_unique_max_index(ptr(a)) == exp
    assert lib.orc_fgbio_unique_max_index(ptr(a)) == exp


def test_tie_rule_divergences():  # base_builder.rs:1249-1356, 2584-2668
    desc = _ll([-1.0, -1.0 - EPS, -1.0e9, -1.0e9])
    asc = _ll([-1.0 - EPS, -1.0, -1.0e9, -1.0e9])
    assert lib.orc_unique_max_index(ptr(desc)) == -1 and lib.orc_unique_max_index(ptr(asc)) == -1
    assert lib.orc_fgbio_unique_max_index(ptr(desc)) == -1 and lib.orc_fgbio_unique_max_index(ptr(asc)) == 1
    for mag in [-1e-4, -0.01, -1.0, -10.0, -500.0, -5000.0]:
        a = _ll([mag, mag - one_ulp(mag), -1e9, -1e9])
        assert lib.orc_unique_max_index(ptr(a)) == -1
        b = _ll([mag, mag - 1.0, -1e9, -1e9])
        assert lib.orc_unique_max_index(ptr(b)) == 0
    for mag in [-500.0, -5000.0]:
        a = _ll([mag, mag - one_ulp(mag), -1e9, -1e9])
        assert lib.orc_fgbio_unique_max_index(ptr(a)) == 0
    for mag in [-0.1, -0.01, -1e-4]:
        sep = step_away_from_zero(mag, 5)
        a = _ll([mag, sep, -1e9, -1e9])
        assert lib.orc_unique_max_index(ptr(a)) == 0
        assert lib.orc_fgbio_unique_max_index(ptr(a)) == -1
    a = _ll([-INF] * 4)
    assert lib.orc_unique_max_index(ptr(a)) == -1 and lib.orc_fgbio_unique_max_index(ptr(a)) == 0
    a = _ll([-INF, -3.0, -INF, -INF])
    assert lib.orc_unique_max_index(ptr(a)) == 1
    a = _ll([float("nan")] * 4)
    assert lib.orc_unique_max_index(ptr(a)) == -1 and lib.orc_fgbio_unique_max_index(ptr(a)) == -1


def test_two_two_split_real_data_pin():  # base_builder.rs:2682-2708 (idt-cfdna library:502, fgbio 4.0.0 → T,Q3)
    for rule, exp in [(0, ("T", 3)), (1, ("N", 2))]:
        b = Builder(45, 40, rule)
        for base in "CCTT":
            b.add(base, 37)
        assert b.call() == exp
        ll = b.likelihoods()
        # the reference documents the two lanes as "exactly one ULP" apart (C below T); the decimal
        # literals in its comment were printed on another libm, so pin the structure, not the digits
        assert ll[1] < ll[3] and bits(float(ll[1])) - bits(float(ll[3])) == 1
        assert abs(ll[1] + 18.42461843127378) < 1e-13
    for rule in (0, 1):
        b = Builder(93, 93, rule)
        b.add("A", 20)
        b.add("C", 20)
        assert b.call()[0] == "N"


# ---- base_builder.rs: column pins ---------------------------------------------------------
@pytest.mark.parametrize("pre,post,obs,depth,exp", [(45, 2, 2, 50, 16), (70, 5, 5, 15, 65), (70, 5, 5, 40, 70), (93, 40, 20, 3, 69),
                                                    (20, 10, 10, 4, 19), (45, 40, 40, 50, 45), (93, 93, 93, 100, 93), (2, 2, 2, 5, 2)])
def test_unanimous_quality_pins(pre, post, obs, depth, exp):  # base_builder.rs:2500-2526
    b = Builder(pre, post)
    b.add("A", obs, depth)
    assert b.call() == ("A", exp)
    assert b.call_full() == ("A", exp)


def test_equal_likelihood_no_call():  # base_builder.rs:1510-1524
    b = Builder(93, 93)
    assert b.call() == ("N", 2)
    b.add("A", 20)
    b.add("C", 20)
    assert b.call() == ("N", 2)


def test_massive_pileup():  # base_builder.rs:1529-1556
    b = Builder(50, 50)
    b.add("C", 20, 1000)
    assert b.call() == ("C", 50)
    assert b.contributions() == 1000 and b.observations_for_base("C") == 1000 and b.observations_for_base("A") == 0
    b.add("T", 20, 10)
    assert b.call() == ("C", 50)
    assert b.contributions() == 1010 and b.observations_for_base("T") == 10


def test_conflicting_evidence():  # base_builder.rs:1560-1570
    b = Builder(50, 50)
    b.add("A", 30)
    b.add("C", 28)
    base, q = b.call()
    assert base == "A" and q <= 5


def test_neg_inf_lane_does_not_inflate_quality():  # base_builder.rs:1579-1603
    b = Builder(45, 40)
    b.add("A", 30, 2)
    assert b.call() == ("A", 44)
    b.add("C", 0)
    assert b.call() == ("A", 44)


def test_single_base_and_reset():  # base_builder.rs:1607-1622
    b = Builder(50, 50)
    b.add("A", 20)
    assert b.call() == ("A", 20) and b.contributions() == 1
    b.reset()
    b.add("C", 20)
    assert b.call() == ("C", 20) and b.contributions() == 1


def test_scale_base_qualities_post_umi():  # base_builder.rs:1687-1708
    for q_in, q_exp in zip([20, 15, 10, 5], [9, 8, 7, 4]):
        b = Builder(93, 10)
        b.add("A", q_in)
        q = b.call()[1]
        assert q <= q_in and abs(q - q_exp) <= 1


def test_ignored_bases_and_qual_clamp():  # base_builder.rs:836-845
    b = Builder(45, 40)
    for base in "NnRY.=":
        b.add(base, 30)
    assert b.contributions() == 0 and b.call() == ("N", 2)
    b.add("a", 200)  # lower case accepted, quality clamped to 93
    b2 = Builder(45, 40)
    b2.add("A", 93)
    assert np.array_equal(b.likelihoods(), b2.likelihoods())


def test_tables_match_inline_formula():  # base_builder.rs:1119-1154
    for post in [0, 10, 40, 45, 93, 255]:
        b = Builder(45, post)
        corr, _ = b.table(0)
        err, _ = b.table(1)
        ln_post = lib.orc_phred_to_ln_error_prob(post)
        for q in range(94):
            adj = lib.orc_ln_error_prob_two_trials(ln_post, lib.orc_phred_to_ln_error_prob(q))
            assert bits(corr[q]) == bits(lib.orc_ln_not(adj))
            assert bits(err[q]) == bits(adj - math.log(3.0))


def test_gap_table_invariants():  # base_builder.rs:1811-1847, 2251-2386
    for pre in [2, 20, 45, 70, 93]:
        b = Builder(pre, 40)
        thr, cap = b.table(2)
        cerr, _ = b.table(3)
        assert cap == lib.orc_ln_prob_to_phred(lib.orc_phred_to_ln_error_prob(pre)) and cap >= 2
        fin = thr[np.isfinite(thr)]
        assert np.all(np.diff(fin) >= 0)
        assert np.all(np.isfinite(thr[: cap + 1])) and np.all(np.isinf(thr[cap + 1:]))
        assert thr[0] == 0.0
        assert np.all(cerr[:cap] > 0) and np.all(cerr[cap:] == 0.0)
        assert bits(cerr[cap - 1]) == bits(lib.orc_consensus_error(thr[cap]))
        for q in range(cap + 1):  # threshold is the least gap reaching q
            if thr[q] > 0:
                assert lib.orc_unanimous_quality_from_gap(thr[q], pre) >= q
    assert lib.orc_consensus_error(0.0) == 0.75


@pytest.mark.parametrize("which,min_cases", [(0, 30720), (2, 300)])
def test_fast_path_equals_call_full_sweeps(which, min_cases):  # base_builder.rs:1986-2012, 2092-2123
    n = C.c_uint64()
    assert lib.orc_sweep_fast_vs_full(which, C.byref(n)) == 0
    assert n.value >= min_cases


def test_fast_path_equals_call_full_dense():  # base_builder.rs:2042-2083
    n = C.c_uint64()
    assert lib.orc_sweep_fast_vs_full(1, C.byref(n)) == 0
    assert n.value > 100000


# ---- other layers --------------------------------------------------------------------------
@pytest.mark.parametrize("name,exp", [("q0", -593808727), ("q2", -974105965), ("q3", -1135430185), ("q9", 98042550),
                                      ("read0", 916970908), ("read5", -1573193749)])
def test_read_name_rank_htsjdk_vectors(name, exp):  # raw-bam/hash.rs:102-111
    assert lib.orc_read_name_rank(name.encode(), len(name)) == exp


@pytest.mark.parametrize("cigar,exp", [("40M", (0, 40, 0)), ("5S10M", (5, 10, 0)), ("10M3S", (0, 10, 3)), ("5S10M3S", (5, 10, 3)),
                                       ("5H10M2H", (0, 10, 0)), ("2H5S10M3S2H", (5, 10, 3)), ("5M2D3N4M", (0, 14, 0)), ("5M2I3M", (0, 8, 0)),
                                       ("268435455M" * 9, (0, 2147483647, 0)), ("", None), ("10", None), ("M", None), ("0M", None),
                                       ("10M5", None), ("5M3S4M", None), ("3M2H4M", None), ("10S", None), ("10Q", None), ("268435456M", None)])
def test_parse_mc_cigar(cigar, exp):  # raw-bam/overlap.rs parse_mc_cigar_ops tests
    out = np.zeros(3, dtype=np.int32)
    ok = lib.orc_parse_mc(cigar.encode(), ptr(out))
    if exp is None:
        assert ok == 0
    else:
        assert ok == 1 and tuple(out) == exp


def test_mate_clip_ops_indel_pair():  # raw-bam/overlap.rs ops_core_matches_record_entry_for_indel_pair
    enc = lambda t, n: (n << 4) | t
    this_ops = np.array([enc(0, 70), enc(1, 10), enc(0, 20)], dtype=np.uint32)
    mate_ops = np.array([enc(4, 80), enc(0, 20)], dtype=np.uint32)
    assert lib.orc_mate_clip_ops(0, 1000, ptr(this_ops), 3, 1019, ptr(mate_ops), 2) == 61


def test_consensus_umis():  # simple_umi.rs tests
    def cu(umis):
        buf = C.create_string_buffer(256)
        n = lib.orc_consensus_umis("\n".join(umis).encode(), buf, 256)
        return None if n < 0 else buf.value.decode()

    assert cu(["ACGT"]) == "ACGT"
    assert cu(["ACGT", "ACGT", "ACGA"]) == "ACGT"
    assert cu(["AAAA-CCCC", "AAAA-CCCC", "AAAT-CCCC"]) == "AAAA-CCCC"
    assert cu(["ACGT", "ACGA"]) == "ACGN"
    assert cu(["ACGT", "ACG"]) is None
    assert cu(["NNNN", "NNNN"]) == "NNNN"
    # SimpleConsensusCaller::call_consensus, the reference's own cases (simple_umi.rs `mod tests`):
    assert cu(["A", "AC"]) is None                                            # test_fail_if_sequences_have_different_lengths (panics)
    assert cu(["GATT-ACA", "GATT-ACA", "GATTAACA"]) is None                   # test_fail_if_mixed_dna_and_non_dna
    assert cu(["GATT-ACA", "GATT+ACA"]) is None                               # test_fail_if_non_dna_chars_differ
    assert cu(["A", "A"]) == "A" and cu(["GATTACA", "GATTACA"]) == "GATTACA"  # test_consensus_from_sequences_that_agree
    assert cu(["A", "C", "G", "T"]) == "N"                                    # test_consensus_from_sequences_that_differ
    assert cu(["A", "C", "C", "C"]) == "C" and cu(["C", "C", "C", "A"]) == "C"
    assert cu(["GATTACA", "GATTACA", "GATTACA", "NNNNNNN"]) == "GATTACA"
    assert cu(["GATT-ACA"] * 3) == "GATT-ACA" and cu(["XGAT", "XGAT"]) == "XGAT" and cu(["GATY", "GATY"]) == "GATY"   # test_gracefully_handle_non_acgtn_bases


def test_quality_trim_point():  # vanilla_caller.rs:992-1016 (htsjdk TrimmingUtil semantics)
    def tp(q, t):
        a = np.array(q if q else [0], dtype=np.uint8)
        return lib.orc_quality_trim_point(ptr(a), len(q), t)

    assert tp([30] * 10, 0) == 0
    assert tp([], 10) == 0
    assert tp([30] * 10, 10) == 10
    assert tp([30] * 5 + [2] * 5, 10) == 5
    assert tp([2] * 10, 10) == 0


def test_single_input_quals():  # vanilla_caller.rs:469-501
    import fgx_opts

    o = fgx_opts.defaults()
    out = np.zeros(94, dtype=np.uint8)
    lib.orc_single_input_quals(C.addressof(o), ptr(out))
    assert out[60] <= 42 and out[2] == 2 and np.all(np.diff(out.astype(int)) >= 0)
    assert out[30] in (29, 30)


def _murmur3_unencoded_chars(units, seed=42):
    """htsjdk `Murmur3.hashUnencodedChars` (Guava's Murmur3_32 over UTF-16 code units, two per block), as a signed 32-bit int."""
    M = 0xFFFFFFFF

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & M

    def mix_k1(k1):
        k1 = (k1 * 0xCC9E2D51) & M
        k1 = rotl(k1, 15)
        return (k1 * 0x1B873593) & M

    h1 = seed
    for i in range(0, len(units) - 1, 2):
        k1 = mix_k1(units[i] | (units[i + 1] << 16))
        h1 ^= k1
        h1 = rotl(h1, 13)
        h1 = (h1 * 5 + 0xE6546B64) & M
    if len(units) & 1:
        h1 ^= mix_k1(units[-1])
    h1 ^= 2 * len(units)
    h1 ^= h1 >> 16
    h1 = (h1 * 0x85EBCA6B) & M
    h1 ^= h1 >> 13
    h1 = (h1 * 0xC2B2AE35) & M
    h1 ^= h1 >> 16
    return h1 - (1 << 32) if h1 & 0x80000000 else h1


@pytest.mark.parametrize("name", ["A", "H0164ALXX140820:2:1101:10003:23260", "abc", "abcd", "q0", "read5"])
def test_read_name_rank_matches_utf16_murmur3(name):  # raw-bam/hash.rs:114-127 (ascii_fast_path_matches_utf16_widening)
    units = [ord(c) for c in name]
    assert lib.orc_read_name_rank(name.encode(), len(name)) == _murmur3_unencoded_chars(units)


@pytest.mark.parametrize("left,right", [("H0164ALXX140820:2:1101:10003:23260", "H0164ALXX140820:2:1101:10003:23261"),
                                        ("A0164ALXX140820:2:1101:10003:23260", "B0164ALXX140820:2:1101:10003:23260"),
                                        ("H0164ALXX140820:2:1101:10003:23260", "H0164ALXX140820:2:2101:10003:23260"), ("frag:1", "frag:10")])
def test_distinct_names_rank_distinctly(left, right):  # raw-bam/hash.rs:129-141
    assert lib.orc_read_name_rank(left.encode(), len(left)) != lib.orc_read_name_rank(right.encode(), len(right))
