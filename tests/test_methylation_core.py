"""The methylation-aware mode's device code and host pieces, without a device: `fgumi_amd/csrc/methylation_core.h` is host + device
source — the annotation kernel's per-position body is run here lane by lane through `fgx_methylation_annotate_host` and compared
with the oracle's restatement of `annotate_simplex_methylation` + the normalisation loop (methylation.rs:193-242,
vanilla_caller.rs:838-852); the aligned runs the host derives from an anchor read against `query_to_ref_positions`
(methylation.rs:116-178); the MM / ML builder against `build_mm_ml_tags` (:264-329) and the reference's own unit-test vectors."""
import ctypes as C
import random

import numpy as np
import pytest

import bamutil
import orc
from fgumi_amd._lib import lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def product_runs(simplified, start, is_reverse, original):
    s = np.array(bamutil.cigar_ops(simplified), dtype=np.uint32)
    o = np.array(bamutil.cigar_ops(original), dtype=np.uint32)
    runs = np.zeros((len(s) + 1, 4), dtype=np.int64)
    n = lib.fgx_methylation_runs_host(ptr(s), len(s), start, int(is_reverse), ptr(o), len(o), ptr(runs), len(runs))
    return runs[:n]


def expand(runs, n_query):
    pos = [None] * n_query
    for q0, ln, r0, step in runs:
        for k in range(int(ln)):
            pos[int(q0) + k] = int(r0) + int(step) * k
    return pos


def random_simplified(rng, n_ops):
    """A simplified CIGAR (M / I / D / N only, no two adjacent ops of a kind), starting and ending with M."""
    ops, last = [], None
    for i in range(n_ops):
        kinds = "M" if i in (0, n_ops - 1) else rng.choice(["M", "I", "D", "N", "M"])
        if kinds == last:
            kinds = "M" if last != "M" else "D"
        ops.append((rng.randint(1, 12), kinds))
        last = kinds
    return ops


def cigar_str(ops):
    return "".join(f"{n}{k}" for n, k in ops)


def truncate(ops, qlen):
    out, rem = [], qlen
    for n, k in ops:
        if rem == 0:
            break
        if k in "MI":
            t = min(n, rem)
            out.append((t, k))
            rem -= t
        else:
            out.append((n, k))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_aligned_runs_equal_query_to_ref_positions(seed):
    rng = random.Random(seed)
    for _ in range(300):
        orig = random_simplified(rng, rng.randint(1, 7))
        rev = rng.random() < 0.5
        simp = list(reversed(orig)) if rev else list(orig)
        qlen_full = sum(n for n, k in simp if k in "MI")
        simp = truncate(simp, rng.randint(1, qlen_full))
        start = rng.randint(0, 5000)
        want = orc.meth_query_to_ref_positions(cigar_str(simp), start, rev, cigar_str(orig))
        got = expand(product_runs(cigar_str(simp), start, rev, cigar_str(orig)), len(want))
        assert got == want


def test_aligned_runs_of_the_reference_unit_tests():  # methylation.rs:461-525
    assert expand(product_runs("10M", 100, False, "10M"), 10) == [100 + i for i in range(10)]
    assert expand(product_runs("5M2I3M", 100, False, "5M2I3M"), 10) == [100, 101, 102, 103, 104, None, None, 105, 106, 107]
    assert expand(product_runs("5M2D5M", 100, False, "5M2D5M"), 10) == [100, 101, 102, 103, 104, 107, 108, 109, 110, 111]
    assert expand(product_runs("10M", 100, True, "10M"), 10) == [109 - i for i in range(10)]


def annotate_host(reads, runs, contig, top, n_pos):
    """Stage the reads the way the general path does (bases then quals per read) and run the kernel body on every position."""
    stage = bytearray()
    off, lens = [], []
    for r in reads:
        off.append(len(stage))
        lens.append(len(r))
        stage += r.encode() + bytes([30] * len(r))
    buf = np.frombuffer(bytes(stage) or b"\0", dtype=np.uint8).copy()
    off_a, len_a = np.array(off, dtype=np.uint64), np.array(lens, dtype=np.uint32)
    runs_a = np.ascontiguousarray(np.array(runs, dtype=np.int64).reshape(-1, 4))
    ctg = np.frombuffer(contig or b"\0", dtype=np.uint8).copy()
    flag, u, t = np.zeros(max(1, n_pos), np.uint8), np.zeros(max(1, n_pos), np.uint32), np.zeros(max(1, n_pos), np.uint32)
    rc = lib.fgx_methylation_annotate_host(ptr(buf), ptr(off_a), ptr(len_a), len(reads), ptr(runs_a), len(runs_a), ptr(ctg), len(contig), int(top), n_pos,
                                           ptr(flag), ptr(u), ptr(t))
    assert rc == 0
    out_reads = [bytes(buf[o:o + n]).decode() for o, n in zip(off, lens)]
    quals_ok = all(bytes(buf[o + n:o + 2 * n]) == bytes([30] * n) for o, n in zip(off, lens))
    return [bool(x) for x in flag[:n_pos]], [int(x) for x in u[:n_pos]], [int(x) for x in t[:n_pos]], out_reads, quals_ok


@pytest.mark.parametrize("seed", range(8))
def test_annotation_kernel_body_equals_the_oracle(seed):
    rng = random.Random(100 + seed)
    for _ in range(120):
        contig = "".join(rng.choice("ACGTacgtN") for _ in range(rng.randint(30, 200))).encode()
        orig = random_simplified(rng, rng.randint(1, 5))
        rev = rng.random() < 0.5
        top = rng.random() < 0.5
        simp = list(reversed(orig)) if rev else list(orig)
        n_pos = rng.randint(1, sum(n for n, k in simp if k in "MI"))
        simp = truncate(simp, n_pos)
        start = rng.randint(-5, len(contig))                      # runs may leave the contig at either end
        reads = ["".join(rng.choice("ACGTN") for _ in range(rng.randint(1, n_pos))) for _ in range(rng.randint(1, 9))]
        reads[rng.randrange(len(reads))] = "".join(rng.choice("ACGT") for _ in range(n_pos))      # the anchor
        positions = orc.meth_query_to_ref_positions(cigar_str(simp), start, rev, cigar_str(orig))
        ref_bases = [chr(contig[p]) if (p is not None and 0 <= p < len(contig)) else None for p in positions]
        want_c, want_u, want_t = orc.meth_annotate(n_pos, reads, ref_bases, top)
        unconv, conv = ("C", "T") if top else ("G", "A")
        want_reads = ["".join(unconv if (want_c[i] and b == conv) else b for i, b in enumerate(r)) for r in reads]
        runs = product_runs(cigar_str(simp), start, rev, cigar_str(orig))
        got_c, got_u, got_t, got_reads, quals_ok = annotate_host(reads, runs, contig, top, n_pos)
        assert (got_c, got_u, got_t) == (want_c, want_u, want_t)
        assert got_reads == want_reads and quals_ok


def product_mm_ml(bases, evidence, top, mode):
    b = np.frombuffer(bases.encode() + b"\0", dtype=np.uint8).copy()
    c = np.array([int(e[0]) for e in evidence] + [0], dtype=np.uint8)
    u = np.array([e[1] for e in evidence] + [0], dtype=np.uint32)
    t = np.array([e[2] for e in evidence] + [0], dtype=np.uint32)
    mm = C.create_string_buffer(16 + 12 * (len(bases) + 1))
    ml = np.zeros(len(bases) + 1, dtype=np.uint8)
    r = lib.fgx_methylation_mm_ml_host(ptr(b), len(bases), ptr(c), ptr(u), ptr(t), int(top), mode, mm, len(mm), ptr(ml), len(ml))
    return None if r < 0 else (mm.value.decode(), [int(x) for x in ml[:r]])


def test_mm_ml_builder_on_the_reference_vectors():  # methylation.rs:622-800
    ev = [(0, 0, 0), (1, 3, 0), (0, 0, 0), (1, 0, 3), (0, 0, 0), (0, 0, 0)]
    assert product_mm_ml("ACGCAC", ev, True, 1) == ("C+m,0,0;", [255, 0])
    assert product_mm_ml("AGCGAG", ev, False, 1) == ("G-m,0,0;", [255, 0])
    assert product_mm_ml("AGGT", [(0, 0, 0)] * 4, True, 1) is None
    assert product_mm_ml("CCACC", [(1, 5, 0), (0, 0, 0), (0, 0, 0), (1, 0, 5), (0, 0, 0)], True, 1) == ("C+m,0,1;", [255, 0])
    assert product_mm_ml("CCCCC", [(1, 0, 3)] * 5, True, 2)[1] == [255] * 5
    assert product_mm_ml("CCCCC", [(1, 3, 0)] * 5, True, 2)[1] == [0] * 5
    assert product_mm_ml("CCCCC", [(1, 3, 0)] * 5, True, 1)[1] == [255] * 5


@pytest.mark.parametrize("seed", range(4))
def test_mm_ml_builder_equals_the_oracle(seed):
    rng = random.Random(500 + seed)
    for _ in range(400):
        n = rng.randint(1, 60)
        bases = "".join(rng.choice("ACGTNacgt") for _ in range(n))
        ev = [(rng.random() < 0.4, rng.choice([0, 0, 1, 2, 7, 40000, 0xFFFFFFFF]), rng.choice([0, 0, 1, 3, 9, 70000, 0xFFFFFFFF])) for _ in range(n)]
        for top in (True, False):
            for mode in (1, 2):
                assert product_mm_ml(bases, ev, top, mode) == orc.meth_build_mm_ml(bases, ev, top, mode)
