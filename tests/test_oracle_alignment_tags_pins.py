"""The oracle's restatement of `regenerate_alignment_tags_raw` (crates/fgumi-sam/src/alignment_tags.rs:259-433) against the known answers of
the reference's own unit tests (alignment_tags.rs:550-1114; each case names the test it transcribes) — the NM / UQ / MD values the
RecordBuf tests assert hold for the raw path too (alignment_tags.rs:1020-1075 checks that equivalence) — and the raw tag editing rules of
crates/fgumi-raw-bam/src/tags.rs:808-888 (in-place overwrite of a 4-byte integer / an equally long string, splice at the same place,
remove-and-append with the smallest signed-first integer type) on crafted records."""
import ctypes as C

import numpy as np
import pytest

import bamutil
import orc

REF = b"ACGTACGTACGTACGT"          # create_test_reference (:488-495)

orc.lib.orc_regenerate_alignment_tags.restype = C.c_int
orc.lib.orc_regenerate_alignment_tags.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]


def set_ref(*seqs):
    bufs = [C.create_string_buffer(s, max(1, len(s))) for s in seqs]
    ptrs = (C.c_void_p * len(bufs))(*[C.cast(b, C.c_void_p).value for b in bufs])
    lens = (C.c_uint64 * len(bufs))(*[len(s) for s in seqs])
    orc.lib.orc_set_reference(len(bufs), ptrs, lens)


def regen(rec: bytes):
    out = C.create_string_buffer(len(rec) + 4096)
    n = C.c_uint32()
    rc = orc.lib.orc_regenerate_alignment_tags(rec, len(rec), out, len(out), C.byref(n))
    if rc < 0:
        raise RuntimeError(orc.lib.orc_last_error().decode())
    return rc, out.raw[:n.value]


def mapped(seq, quals, cigar, start, tags=(), flag=0, ref_id=0):
    """create_mapped_record (:511-519): reference id 0, 1-based alignment start."""
    return bamutil.make_record("q", seq, quals, flag=flag, ref_id=ref_id, pos=start - 1, cigar=cigar, tags=tags)


@pytest.fixture(autouse=True)
def _reference():
    set_ref(REF)
    yield
    orc.lib.orc_set_reference(0, None, None)


CASES = [   # (reference test, sequence, qualities, CIGAR, NM, UQ, MD)
    ("test_perfect_match", "ACGT", [30] * 4, "4M", 0, 0, "4"),
    ("test_one_mismatch", "ATGT", [30] * 4, "4M", 1, 30, "1C2"),
    ("test_masked_base", "ANGT", [30, 0, 30, 30], "4M", 1, 0, "1C2"),
    ("test_insertion", "ACTTGT", [30, 30, 25, 25, 30, 30], "2M2I2M", 2, 0, "4"),
    ("test_deletion", "ACAC", [30] * 4, "2M2D2M", 2, 0, "2^GT2"),
    ("test_soft_clip", "TTACGTGG", [20, 20, 30, 30, 30, 30, 20, 20], "2S4M2S", 0, 0, "4"),
    ("test_hard_clip", "ACGT", [30] * 4, "2H4M2H", 0, 0, "4"),
    ("test_multiple_mismatches", "AATT", [30, 25, 20, 35], "4M", 2, 45, "1C0G1"),
    ("test_multiple_masked_bases", "ANNN", [30, 0, 0, 0], "4M", 3, 0, "1C0G0T0"),
    ("test_complex_cigar", "ACTCAC", [30, 30, 25, 30, 30, 30], "2M1I1M1D2M", 3, 30, "2G0^T2"),
    ("test_sequence_match_and_mismatch_ops", "ACTT", [30, 30, 25, 30], "2=1X1=", 1, 25, "2G1"),
    ("test_pad_operation", "ACGT", [30] * 4, "2M2P2M", 0, 0, "4"),
    ("test_skip_operation", "ACGT", [30] * 4, "2M2N2M", 2, 60, "2A0C0"),
    ("test_insertion_at_end", "ACGTTT", [30, 30, 30, 30, 20, 20], "4M2I", 2, 0, "4"),
    ("test_deletion_at_end", "AC", [30, 30], "2M2D", 2, 0, "2^GT0"),
    ("test_mixed_matches_and_masks", "ACNTTC", [30, 30, 0, 30, 25, 30], "6M", 2, 25, "2G1A1"),
    ("test_regenerate_alignment_tags_raw_happy_path", "ATGT", [30, 30, 25, 30], "4M", 1, 30, "1C2"),
    ("test_regenerate_alignment_tags_zero_ref_span_returns_true", "ACGT", [30] * 4, "4I", 0, 0, "0"),
]


@pytest.mark.parametrize("name,seq,quals,cigar,nm,uq,md", CASES)
def test_reference_known_answers(name, seq, quals, cigar, nm, uq, md):
    rc, out = regen(mapped(seq, quals, cigar, 1))
    assert rc == 1
    t = bamutil.parse(out)["tags"]
    assert (t["NM"][1], t["UQ"][1], t["MD"][1]) == (nm, uq, md), name
    assert bamutil.parse(out)["tag_order"] == ["NM", "UQ", "MD"]                  # appended in this order to a record without them


def test_regenerate_tags_fgbio_equivalent():
    """:641-673 — an all-A reference, stale tags planted: NM 7 -> 2, MD 6A7C8T9G -> 3A4A1, UQ 237 -> 40."""
    set_ref(b"A" * 20)
    rec = mapped("AAACAAAATA", [20] * 10, "10M", 1, tags=[("NM", "raw", b"i" + (7).to_bytes(4, "little")), ("MD", "Z", "6A7C8T9G"),
                                                          ("UQ", "raw", b"i" + (237).to_bytes(4, "little"))])
    rc, out = regen(rec)
    p = bamutil.parse(out)
    assert rc == 1 and (p["tags"]["NM"][1], p["tags"]["MD"][1], p["tags"]["UQ"][1]) == (2, "3A4A1", 40)
    # the 4-byte integers are overwritten in place, MD (other length) is spliced where it was: the order of the tags does not change
    assert p["tag_order"] == ["NM", "MD", "UQ"] and p["tags"]["NM"][0] == "i" and p["tags"]["UQ"][0] == "i"


def test_unmapped_read_loses_the_tags():
    """test_unmapped_read (:615-635): NM / MD / UQ removed, everything else stays where it was."""
    rec = bamutil.make_record("q", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1, tags=[("RG", "Z", "A"), ("NM", "i", 7), ("MD", "Z", "6A7C8T9G"), ("XY", "i", 300), ("UQ", "i", 237)])
    rc, out = regen(rec)
    p = bamutil.parse(out)
    assert rc == 0 and p["tag_order"] == ["RG", "XY"] and p["tags"]["XY"][1] == 300 and p["seq"] == "ACGT"


def test_negative_ref_id_on_a_mapped_record_strips_stale_tags():
    """test_regenerate_alignment_tags_raw_strips_stale_tags_on_negative_ref_id (:928-972)."""
    rec = mapped("ACGTACGT", [30] * 8, "8M", 1, tags=[("NM", "i", 99), ("UQ", "i", 12345), ("MD", "Z", "8")], ref_id=-1)
    rc, out = regen(rec)
    p = bamutil.parse(out)
    assert rc == 0 and p["tag_order"] == [] and not (p["flag"] & 4)


def test_bounds_and_short_records_are_errors():
    """test_regenerate_alignment_tags_raw_validates_bounds (:974-992) and ..._rejects_short_record (:994-1006)."""
    rec = mapped("ACGTACGT", [30] * 8, "8M", 1)
    qual_off = 32 + 2 + 4 + 4
    with pytest.raises(RuntimeError, match="Truncated"):
        regen(rec[:qual_off])
    with pytest.raises(RuntimeError, match="too short"):
        regen(bytes(10))
    with pytest.raises(RuntimeError, match="not found in header"):
        regen(mapped("ACGT", [30] * 4, "4M", 1, ref_id=3))
    with pytest.raises(RuntimeError, match="region"):
        regen(mapped("ACGT", [30] * 4, "4M", 15))                      # the alignment runs off the end of the 16-base contig


def test_tag_editing_rules():
    """tags.rs:808-888: an 'i' / 'I' integer is overwritten in place; any other integer type is removed and the new value appended with the
    smallest signed-first type; a string of the same length is overwritten in place (type byte kept), another length is spliced in place."""
    seq, q = "ATGTACGTAC", [30] * 10                                       # one mismatch (T for C at position 2): NM 1, UQ 30, MD 1C8
    rec = mapped(seq, q, "10M", 1, tags=[("NM", "i", 5), ("XA", "Z", "keep"), ("UQ", "raw", b"I" + (9).to_bytes(4, "little")), ("MD", "Z", "abc")])
    p = bamutil.parse(regen(rec)[1])
    assert p["tag_order"] == ["XA", "UQ", "MD", "NM"]                                       # NM was a 'c' (5 fits a byte): removed, appended
    assert p["tags"]["NM"] == ("c", 1) and p["tags"]["UQ"] == ("I", 30) and p["tags"]["MD"] == ("Z", "1C8") and p["tags"]["XA"] == ("Z", "keep")
    rec = mapped(seq, q, "10M", 1, tags=[("MD", "Z", "longer-than-new"), ("NM", "raw", b"S" + (700).to_bytes(2, "little")), ("ZZ", "i", 70000)])
    p = bamutil.parse(regen(rec)[1])
    assert p["tag_order"] == ["MD", "ZZ", "NM", "UQ"] and p["tags"]["MD"] == ("Z", "1C8") and p["tags"]["NM"] == ("c", 1) and p["tags"]["UQ"] == ("c", 30)
    big = mapped("N" * 16, [40] * 16, "16M", 1)                             # UQ 640: an unsigned 16-bit value ('S'), NM 16 ('c')
    p = bamutil.parse(regen(big)[1])
    assert p["tags"]["UQ"] == ("S", 640) and p["tags"]["NM"] == ("c", 16) and p["tags"]["MD"][1] == "0A0C0G0T0A0C0G0T0A0C0G0T0A0C0G0T0"


def test_lowercase_reference_bases_go_into_md_as_they_are():
    """eq_ignore_ascii_case for the comparison, `ref_base as char` for the MD text (reference.rs keeps the FASTA's case: :589-594)."""
    set_ref(b"acgtACGTacgtACGT")
    p = bamutil.parse(regen(mapped("ATGTTCGT", [30] * 8, "8M", 1))[1])
    assert (p["tags"]["NM"][1], p["tags"]["MD"][1]) == (2, "1c2A3")
