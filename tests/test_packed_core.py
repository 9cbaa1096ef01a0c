"""The packed column pass of k_split_cols (fgumi_amd/csrc/packed_core.h, round 5) on the HOST — tests/devemu compiles the very functions
the kernel inlines (acc_row: the per-row step on eight packed columns; finalize: slots, flags, output bytes; fill_t1: the table of the
one-observation column):

  * against a column-by-column restatement in plain Python: codes, qualities, depths and the "k_call_full's" flag of every column, for
    forward and reverse ends, ragged consensus lengths, quality floors up to 128, rows of 1 .. 17 reads, codes of every kind (ACGT, N, IUPAC,
    '=');
  * against the oracle's ConsensusBaseBuilder (oracle/oracle_phred.hpp, restating base_builder.rs:836-1081): every column the pass ANSWERS
    carries the reference's call — "a flag too many costs time, never a byte";
  * the one-observation table against the builder's call of one observation, quality by quality.
"""
import numpy as np
import pytest

import devemu
import orc

COMP = {1: 8, 2: 4, 4: 2, 8: 1}
BASE = {1: "A", 2: "C", 4: "G", 8: "T"}
SETTINGS = [(45, 40), (30, 30), (60, 50), (20, 45)]


def _tile(rng, m, len_e, profile):
    """Rows of an end: codes (rows, len_e) 4-bit, qualities (rows, len_e); profile picks how dirty the columns are."""
    truth = rng.choice([1, 2, 4, 8], size=len_e)
    codes = np.tile(truth, (m, 1)).astype(np.uint8)
    qual = rng.integers(25, 42, size=(m, len_e)).astype(np.uint8)
    if profile >= 1:      # sequencing errors, masked bases, no-observation codes
        err = rng.random((m, len_e)) < 0.04
        codes[err] = rng.choice([1, 2, 4, 8, 15, 0, 3, 5, 10], size=int(err.sum()))
        low = rng.random((m, len_e)) < 0.06
        qual[low] = rng.choice([0, 1, 2, 5, 9, 10, 11, 19, 20], size=int(low.sum()))
    if profile >= 2:      # sparse rows (clipped / cleared positions: code 0) and extreme quality bytes
        hole = rng.random((m, len_e)) < rng.random() * 0.9
        codes[hole] = 0
        hi = rng.random((m, len_e)) < 0.03
        qual[hi] = rng.choice([93, 94, 127, 128, 200, 254], size=int(hi.sum()))
    return codes, qual


def _pack(codes, qual, rng):
    """The tile rows as k_split_cols stages them: two codes per byte (even position = high nibble), row strides padded to multiples of 16; the
    padding holds whatever followed the read in its record."""
    m, len_e = codes.shape
    qs = ((len_e + 15) // 16) * 16
    ss = (((len_e + 1) // 2 + 15) // 16) * 16
    q = rng.integers(0, 256, size=(m, qs)).astype(np.uint8)
    q[:, :len_e] = qual
    c = np.zeros((m, 2 * ss), dtype=np.uint8)
    c[:, :len_e] = codes
    c[:, len_e:] = rng.integers(0, 16, size=(m, 2 * ss - len_e))
    seq = ((c[:, 0::2] << 4) | c[:, 1::2]).astype(np.uint8)
    return seq, q


def _restate(codes, qual, cnt_e, rev, min_bq, nsafe, cap, min_cons_bq, min_reads):
    """What run_cols_packed leaves per consensus column, one column at a time: (code, qual, depth, flagged, observations in read orientation)."""
    m, len_e = codes.shape
    out = []
    for c in range(cnt_e):
        p = len_e - 1 - c if rev else c
        acc = [(int(codes[j, p]), int(qual[j, p])) for j in range(m) if qual[j, p] >= min_bq and codes[j, p] != 0]
        kinds = set(cc for cc, _ in acc)
        n = len(acc)
        if n == 0:
            out.append((15, 0 if min_reads > 0 else 2, 0, 0, acc))
        elif len(kinds) == 1 and next(iter(kinds)) in COMP and n >= nsafe:
            b = next(iter(kinds))
            b = COMP[b] if rev else b
            ans = (15, 0) if n < min_reads else (15, 2) if cap < min_cons_bq else (b, cap)
            out.append((ans[0], ans[1], n, 0, acc))
        else:
            out.append((None, None, None, 1, acc))
    return out


@pytest.mark.parametrize("profile", [0, 1, 2])
def test_packed_pass_equals_the_column_by_column_restatement(profile):
    rng = np.random.default_rng(100 + profile)
    answered = flagged = 0
    for t in range(400):
        m = int(rng.integers(1, 33)); len_e = int(rng.integers(1, 257)); rev = bool(rng.integers(0, 2))      # (round 6: ends of up to 31 rows; 32 shows the counters' limit is respected by the caller)
        if m > 31:
            m = 31
        cnt_e = int(rng.integers(0, len_e + 1)) if t % 3 else len_e
        min_bq = int(rng.choice([0, 2, 10, 20, 30, 128])); nsafe = int(rng.integers(1, m + 1)); cap = int(rng.choice([30, 40, 45, 90]))
        min_cons_bq = int(rng.choice([2, 2, 2, 40, 50])); min_reads = int(rng.integers(0, nsafe + 1))
        codes, qual = _tile(rng, m, len_e, profile)
        seq, q = _pack(codes, qual, rng)
        code, qo, dep, fl = devemu.packed_end(seq, q, m, len_e, cnt_e, rev, min_bq, nsafe, cap, min_cons_bq, min_reads)
        want = _restate(codes, qual, cnt_e, rev, min_bq, nsafe, cap, min_cons_bq, min_reads)
        for c, (wc, wq, wd, wf, _) in enumerate(want):
            assert fl[c] == wf, (t, c, fl[c], wf)
            if wf:
                flagged += 1
                continue
            assert (code[c], qo[c], dep[c]) == (wc, wq, wd), (t, c, (code[c], qo[c], dep[c]), (wc, wq, wd), rev, m, nsafe)
            answered += 1
    assert answered > 15000 and (profile == 0 or flagged > 2000)


@pytest.mark.parametrize("pre,post", SETTINGS)
@pytest.mark.parametrize("min_bq", [10, 20])
def test_every_answered_column_is_the_reference_call(pre, post, min_bq):
    """nsafe = unanimous_cap_depth of the caller's tables: the (base, cap) the pass writes is what ConsensusBaseBuilder calls for the column's
    observations (reverse ends: complemented, in read order), and the depth its contributions."""
    nsafe, cap = devemu.cap_depth(pre, post, min_bq)
    if nsafe > 31:
        pytest.skip("no end is deep enough for the packed pass at these tables")
    rng = np.random.default_rng(pre * 1000 + post * 10 + min_bq)
    b = orc.Builder(pre, post)
    checked = 0
    for t in range(60):
        m = int(rng.integers(nsafe, 32)); len_e = int(rng.integers(20, 161)); rev = bool(rng.integers(0, 2))
        codes, qual = _tile(rng, m, len_e, 1 + t % 2)
        seq, q = _pack(codes, qual, rng)
        code, qo, dep, fl = devemu.packed_end(seq, q, m, len_e, len_e, rev, min_bq, nsafe, cap, 2, 1)
        for c in range(len_e):
            if fl[c]:
                continue
            p = len_e - 1 - c if rev else c
            b.reset()
            n = 0
            for j in range(m):
                cc, qq = int(codes[j, p]), int(qual[j, p])
                if qq < min_bq or cc not in COMP:          # (below the floor: masked to N when the source read is made; not ACGT: ignored by add)
                    continue
                b.add(BASE[COMP[cc] if rev else cc], min(qq, 93))
                n += 1
            assert dep[c] == n == b.contributions(), (t, c, dep[c], n)
            if n == 0:
                assert (code[c], qo[c]) == (15, 0)
                continue
            base, ql = b.call()
            assert (BASE.get(int(code[c])), int(qo[c])) == (base, ql), (t, c, code[c], qo[c], base, ql, n)
            checked += 1
    assert checked > 1000


@pytest.mark.parametrize("pre,post", SETTINGS + [(93, 93)])
def test_one_observation_table_is_the_builders_call(pre, post):
    t1 = devemu.t1_table(pre, post)
    b = orc.Builder(pre, post)
    answered = 0
    for q in range(0, 94):
        b.reset()
        b.add("A", q)
        base, ql = b.call()
        if t1[q] == 0xFF:
            assert base != "A" or q == 0, (q, base, ql)      # (no table answer: the call does not come back with the observed base / ln 0 in the sums)
        else:
            assert (base, ql) == ("A", int(t1[q])), (q, base, ql, t1[q])
            answered += 1
    assert answered >= 80
    assert (t1[94:] == 0xFF).all()


@pytest.mark.parametrize("pre,post", SETTINGS + [(93, 93)])
def test_two_observation_table_is_the_builders_call(pre, post):
    """Round 6: S2Image::t2 — two observations of one base, qualities in file order — against the oracle's ConsensusBaseBuilder, every pair of qualities."""
    t2 = devemu.t2_table(pre, post)
    b = orc.Builder(pre, post)
    answered = 0
    for q1 in range(94):
        for q2 in range(94):
            b.reset()
            b.add("C", q1)
            b.add("C", q2)
            base, ql = b.call()
            if t2[q1, q2] == 0xFF:
                assert base != "C" or q1 == 0 or q2 == 0, (q1, q2, base, ql)
            else:
                assert (base, ql) == ("C", int(t2[q1, q2])), (q1, q2, base, ql, t2[q1, q2])
                answered += 1
    assert answered >= 80 * 80
