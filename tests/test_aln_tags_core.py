"""The product's scalar core of `fgumi filter --ref` (fgumi_amd/csrc/aln_tags_core.h: what a GPU lane per record runs in filter.hip) against the
oracle's restatement of regenerate_alignment_tags_raw, through fgx_regenerate_alignment_tags_host: the reference's own unit-test inputs, and
thousands of random records — random CIGARs (M I D N S H P = X), masked bases, lower-case reference stretches, alignments at the contig's ends,
stale NM / UQ / MD entries of every integer type and string length in every position among other tags, duplicates, unmapped records, records
without a reference id — byte for byte (the product lays the edited tag block out analytically, the oracle splices a vector as the reference
does).  Error statuses must agree with the oracle's fatal errors."""
import ctypes as C
import random

import pytest

import bamutil
import test_oracle_alignment_tags_pins as pins
from fgumi_amd import lib

lib.fgx_regenerate_alignment_tags_host.restype = C.c_int
lib.fgx_regenerate_alignment_tags_host.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]


def product(rec: bytes, contigs):
    bufs = [C.create_string_buffer(s, max(1, len(s))) for s in contigs]
    ptrs = (C.c_void_p * max(1, len(bufs)))(*[C.cast(b, C.c_void_p).value for b in bufs])
    lens = (C.c_uint64 * max(1, len(bufs)))(*[len(s) for s in contigs])
    out = C.create_string_buffer(len(rec) + 8192)
    n = C.c_uint32()
    st = lib.fgx_regenerate_alignment_tags_host(rec, len(rec), len(contigs), ptrs, lens, out, len(out), C.byref(n))
    return st, out.raw[:n.value] if st in (0, 1) else b""


def both(rec, contigs):
    pins.set_ref(*contigs) if contigs else pins.orc.lib.orc_set_reference(0, None, None)
    st, got = product(rec, contigs)
    try:
        rc, want = pins.regen(rec)
    except RuntimeError as e:
        assert st >= 2, (st, str(e))
        return st, None
    assert st == (0 if rc == 1 else 1), (st, rc)
    if got != want:
        raise AssertionError(f"product differs from the oracle:\n got {bamutil.parse(got)}\nwant {bamutil.parse(want)}\n rec {bamutil.parse(rec)}")
    return st, got


@pytest.mark.parametrize("name,seq,quals,cigar,nm,uq,md", pins.CASES)
def test_reference_unit_test_inputs(name, seq, quals, cigar, nm, uq, md):
    st, out = both(pins.mapped(seq, quals, cigar, 1), [pins.REF])
    t = bamutil.parse(out)["tags"]
    assert st == 0 and (t["NM"][1], t["UQ"][1], t["MD"][1]) == (nm, uq, md)


def _random_record(rng, contigs):
    ref_id = rng.randrange(len(contigs))
    contig = contigs[ref_id]
    ops, span, qlen = [], 0, 0
    for _ in range(rng.randrange(1, 7)):
        t = rng.choice("MMMMIDNS=XHP")
        n = rng.randrange(1, 40 if t in "M=X" else 6)
        ops.append((n, t))
        span += n if t in "MDN=X" else 0
        qlen += n if t in "MIS=X" else 0
    if qlen == 0:
        ops.append((5, "M")); span += 5; qlen += 5
    start = rng.randrange(0, max(1, len(contig) - span + 1)) if rng.random() < 0.9 else max(0, len(contig) - span + rng.randrange(-2, 3))
    # a read that mostly follows the reference
    seq, rp = [], start
    for n, t in ops:
        for _ in range(n if t in "MIS=X" else 0):
            if t in "M=X" and rp < len(contig) and rng.random() < 0.85:
                seq.append(chr(contig[rp]).upper() if rng.random() < 0.9 else chr(contig[rp]).lower())
            else:
                seq.append(rng.choice("ACGTN"))
            if t in "M=X":
                rp += 1
        if t in "DN":
            rp += n
    quals = [rng.randrange(0, 60) for _ in seq]
    def stale():
        out = []
        if rng.random() < 0.6:
            out.append(("NM", "raw", rng.choice([b"c\x05", b"C\xc8", b"s\x10\x27", b"S\x10\x27", b"i\x07\x00\x00\x00", b"I\x07\x00\x00\x00"])))
        if rng.random() < 0.6:
            out.append(("UQ", "raw", rng.choice([b"c\x05", b"S\x10\x27", b"i\x07\x00\x00\x00", b"I\xff\x00\x00\x00"])))
        if rng.random() < 0.6:
            out.append(("MD", "Z", rng.choice(["5", "10A3", "0", "3^AC12T0", "x" * rng.randrange(1, 12)])))
        return out
    tags = stale() + [("RG", "Z", "A"), ("cD", "i", rng.randrange(1, 300)), ("cd", "raw", b"Bs" + (3).to_bytes(4, "little") + bytes(6))]
    if rng.random() < 0.2:
        tags += [("NM", "i", 3)]                                    # a second occurrence: never touched
    rng.shuffle(tags)
    flag = rng.choice([0, 16, 0x41, 0x91]) | (4 if rng.random() < 0.08 else 0)
    rid = -1 if rng.random() < 0.05 else ref_id
    return bamutil.make_record(f"r{rng.randrange(10**6)}", "".join(seq), quals, flag=flag, ref_id=rid, pos=start, cigar="".join(f"{n}{t}" for n, t in ops), tags=tags)


def test_random_records_equal_the_oracle():
    rng = random.Random(20260922)
    contigs = []
    for L in (400, 1500, 90):
        s = bytearray(rng.choice(b"ACGT") for _ in range(L))
        for _ in range(6):                                          # soft-masked (lower-case) stretches and N runs, as FASTA files have them
            a = rng.randrange(L); b = min(L, a + rng.randrange(1, 30))
            s[a:b] = bytes(s[a:b]).lower() if rng.random() < 0.7 else b"N" * (b - a)
        contigs.append(bytes(s))
    seen = {0: 0, 1: 0, "err": 0}
    for _ in range(6000):
        st, _out = both(_random_record(rng, contigs), contigs)
        seen[st if st in (0, 1) else "err"] += 1
    assert seen[0] > 4000 and seen[1] > 300 and seen["err"] > 20, seen


def test_errors_agree_with_the_oracle():
    rec = pins.mapped("ACGTACGT", [30] * 8, "8M", 1)
    assert both(rec[:32 + 2 + 4 + 4], [pins.REF])[0] == 6                      # truncated
    assert both(bytes(10), [pins.REF])[0] == 2                                 # too short
    assert both(pins.mapped("ACGT", [30] * 4, "4M", 1, ref_id=3), [pins.REF])[0] == 3
    assert both(pins.mapped("ACGT", [30] * 4, "4M", 15), [pins.REF])[0] == 5   # off the contig's end
    assert both(pins.mapped("ACGT", [30] * 4, "6M", 1), [pins.REF])[0] == 7    # CIGAR longer than the sequence
    assert both(pins.mapped("ACGT", [30] * 4, "4M", 1), [b""])[0] == 5         # a contig the FASTA lacks
