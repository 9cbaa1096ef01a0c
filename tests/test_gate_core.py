"""The unanimous-column gates of the simplex kernels (fgumi_amd/csrc/gate_core.h) on the HOST, against the oracle's ConsensusBaseBuilder
(oracle/oracle_phred.hpp, restating base_builder.rs:836-994) — tests/devemu compiles the very functions the kernels inline:

  * unanimous_call_lds (exact Kahan sums: k_simplex_wave2, k_simplex_seg, k_deep_cols) answers exactly what the reference's
    try_unanimous_fast_path answers, and defers exactly where it defers;
  * unanimous_call_approx (f32 sums: k_split_cols) never answers anything else: where it answers, the exact gate gives the same quality —
    the property the round-4 f32 hot loop rests on ("a `false` too many costs time, never a byte");
  * s2_cap_pregate (seven f32 instructions ahead of it) implies unanimous_call_approx's cap answer, for every member count the bound may
    be built for;
  * the f32 table's finite stand-in for ln 0 keeps a column that observes a quality 0 away from every gate.
"""
import numpy as np
import pytest

import devemu
import orc

SETTINGS = [(45, 40), (30, 30), (60, 50), (93, 93), (20, 45)]


def _columns(rng, count, stride=64):
    """A mix of columns: clean deep ones, shallow ones, low qualities, wide ranges, long runs of one quality (sums near the gate's brackets)."""
    n = np.empty(count, dtype=np.uint32)
    q = np.zeros((count, stride), dtype=np.uint8)
    kinds = rng.integers(0, 6, size=count)
    for k in range(6):
        idx = np.nonzero(kinds == k)[0]
        if k == 0:      # sequencer-like: 1 - 16 reads of Q25 - Q41
            n[idx] = rng.integers(1, 17, size=idx.size); q[idx] = rng.integers(25, 42, size=(idx.size, stride))
        elif k == 1:    # shallow and poor: 1 - 4 reads of Q2 - Q20 (gaps among the low brackets)
            n[idx] = rng.integers(1, 5, size=idx.size); q[idx] = rng.integers(2, 21, size=(idx.size, stride))
        elif k == 2:    # anything: 1 - 64 reads of Q1 - Q93
            n[idx] = rng.integers(1, 65, size=idx.size); q[idx] = rng.integers(1, 94, size=(idx.size, stride))
        elif k == 3:    # one quality repeated (with one odd read): sums that are multiples of a table entry
            n[idx] = rng.integers(1, 41, size=idx.size); q[idx] = rng.integers(2, 61, size=(idx.size, 1)); q[idx, 0] = rng.integers(2, 61, size=idx.size)
        elif k == 4:    # deep: 32 - 64 reads of Q10 - Q45
            n[idx] = rng.integers(32, 65, size=idx.size); q[idx] = rng.integers(10, 46, size=(idx.size, stride))
        else:           # bytes above the table (clamped to 93) and very low ones
            n[idx] = rng.integers(1, 9, size=idx.size); q[idx] = rng.choice(np.array([1, 2, 3, 90, 93, 94, 120, 200, 254], dtype=np.uint8), size=(idx.size, stride))
    return q, n


@pytest.mark.parametrize("pre,post", SETTINGS)
def test_exact_gate_equals_the_reference_fast_path(pre, post):
    rng = np.random.default_rng(pre * 100 + post)
    q, n = _columns(rng, 6000)
    qe, _, _, _ = devemu.gates(pre, post, q, n)
    b = orc.Builder(pre, post)
    answered = 0
    for i in range(q.shape[0]):
        b.reset()
        for j in range(int(n[i])):
            b.add("A", int(min(q[i, j], 93)))
        got = b.fast_path()
        want = None if qe[i] < 0 else ("A", int(qe[i]))
        assert got == want, (i, n[i], q[i, :n[i]].tolist(), got, want)
        answered += got is not None
    assert answered > 2000          # (the comparison is not vacuous)


@pytest.mark.parametrize("pre,post", SETTINGS)
def test_approximate_gate_never_answers_anything_else(pre, post):
    rng = np.random.default_rng(7 + pre * 100 + post)
    total = agree = exact_yes = approx_yes = pre_yes = 0
    for _ in range(4):
        q, n = _columns(rng, 250000)
        m = np.minimum(n + rng.integers(0, 49, size=n.size).astype(np.uint32), 64).astype(np.uint32)      # the run's member count: >= the observations a lane accepted
        qe, qa, qp, sums = devemu.gates(pre, post, q, n, m)
        assert np.isfinite(sums).all()
        a = qa >= 0
        bad = np.nonzero(a & (qa != qe))[0]
        assert bad.size == 0, (bad[:5], qa[bad[:5]], qe[bad[:5]], n[bad[:5]])
        p = qp >= 0
        bad = np.nonzero(p & (qa != qp))[0]                       # the pre-gate says "the cap": so does the gate behind it
        assert bad.size == 0, (bad[:5], qa[bad[:5]], qp[bad[:5]])
        total += n.size; exact_yes += int((qe >= 0).sum()); approx_yes += int(a.sum()); pre_yes += int(p.sum())
    # the approximate gate gives up little of what the exact one decides (what it gives up is redone exactly by k_call_full)
    assert approx_yes >= 0.97 * exact_yes, (approx_yes, exact_yes)
    assert pre_yes > 0 or pre >= 90


def test_the_usual_column_is_decided_by_the_pre_gate():
    """Depth-8 families of Q30 - Q40 reads (the benchmark's shape): every column's answer is the cap, and the f32 pre-gate finds it."""
    rng = np.random.default_rng(3)
    q = rng.integers(30, 41, size=(100000, 16)).astype(np.uint8)
    n = np.full(100000, 8, dtype=np.uint32)
    qe, qa, qp, _ = devemu.gates(45, 40, q, n, np.full(100000, 16, dtype=np.uint32))
    assert (qe == 45).all() and (qa == 45).all() and (qp == 45).mean() > 0.999


def test_a_quality_zero_observation_keeps_every_gate_silent():
    rng = np.random.default_rng(5)
    q, n = _columns(rng, 50000)
    pos = rng.integers(0, 64, size=n.size) % n
    q[np.arange(n.size), pos] = 0                                  # one observation of quality 0 in every column: correct[0] = ln 0
    qe, qa, qp, sums = devemu.gates(45, 40, q, n)
    assert np.isfinite(sums).all() and (sums[:, 0] < -1e29).all()
    assert (qa < 0).all() and (qp < 0).all() and (qe < 0).all()


@pytest.mark.parametrize("pre,post", SETTINGS[:3])
def test_every_two_quality_column_up_to_40_reads(pre, post):
    """Exhaustive over a family of columns whose gaps land everywhere among the brackets: k reads of quality a and one of quality b
    (a, b in 2 .. 60, k in 0 .. 40; both orders) — 280 000 columns per setting; the approximate gate agrees with the exact one wherever it
    answers, the exact one with the oracle on a sample."""
    a, b, k = np.meshgrid(np.arange(2, 61), np.arange(2, 61), np.arange(0, 41), indexing="ij")
    a, b, k = a.ravel(), b.ravel(), k.ravel()
    cnt = a.size
    q = np.zeros((2 * cnt, 64), dtype=np.uint8)
    q[:cnt] = a[:, None]
    q[np.arange(cnt), k] = b                                        # the odd read last
    q[cnt:] = a[:, None]
    q[cnt:, 0] = b                                                  # the odd read first
    n = np.concatenate([k + 1, k + 1]).astype(np.uint32)
    qe, qa, qp, _ = devemu.gates(pre, post, q, n, np.minimum(n + 7, 64).astype(np.uint32))
    ans = qa >= 0
    assert (qa[ans] == qe[ans]).all()
    assert ((qp < 0) | (qa == qp)).all()
    assert ans.sum() >= 0.97 * (qe >= 0).sum()
    rng = np.random.default_rng(11)
    bld = orc.Builder(pre, post)
    for i in rng.choice(2 * cnt, size=1500, replace=False):
        bld.reset()
        for j in range(int(n[i])):
            bld.add("C", int(q[i, j]))
        got = bld.fast_path()
        assert got == (None if qe[i] < 0 else ("C", int(qe[i]))), (i, got, qe[i])


@pytest.mark.parametrize("pre,post", SETTINGS)
@pytest.mark.parametrize("min_bq", [0, 2, 10, 20, 40])
def test_from_the_cap_depth_on_every_single_base_column_is_the_cap(pre, post, min_bq):
    """unanimous_cap_depth (round 5: k_split_cols decides the usual column from the OR of its bases and a count): n_safe agreeing observations
    of ANY qualities >= the floor make the reference's fast path answer (base, cap) — checked on the oracle's ConsensusBaseBuilder with the
    worst quality of the range repeated, random qualities, bytes above 93, and depths up to the 64 the bound is built for; one observation
    fewer of the worst quality does NOT (the depth is tight, i.e. the function is not vacuous)."""
    n_safe, cap = devemu.cap_depth(pre, post, min_bq)
    b = orc.Builder(pre, post)
    if min_bq == 0:
        assert n_safe is None                                      # quality 0 is ln 0: never
        return
    if n_safe is None:
        return                                                     # (a table that never reaches the budget: the kernel keeps to the sums)
    assert 1 <= n_safe <= 64
    rng = np.random.default_rng(pre * 1000 + post * 10 + min_bq)

    def call(quals):
        b.reset()
        for q in quals:
            b.add("G", int(min(q, 93)))
        return b.fast_path()
    # the worst quality of the range: the one whose n_safe-fold column has the smallest gap = the floor itself or 93 (the table is monotone
    # in neither direction at the extremes), so try every quality of the range at exactly n_safe
    for q in range(min(min_bq, 93), 94):
        assert call([q] * n_safe) == ("G", cap), (q, n_safe)
        assert call([q] * 64) == ("G", cap), (q, 64)
    for _ in range(3000):
        n = int(rng.integers(n_safe, 65))
        quals = rng.integers(min_bq, 256 if rng.random() < 0.1 else 94, size=n)
        assert call(quals) == ("G", cap), (n, quals.tolist())
    if n_safe > 1:
        assert any(call([q] * (n_safe - 1)) != ("G", cap) for q in range(min(min_bq, 93), 94))
