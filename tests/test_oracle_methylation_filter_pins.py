"""The oracle's restatement of the methylation (EM-Seq / TAPs) filters of `fgumi filter` (crates/fgumi-consensus/src/filter.rs:925-1340)
against the known answers of the reference's own unit tests (filter.rs:1617-2170; each case names the test it transcribes): the depth masks
of simplex and duplex records, the strand agreement at reference CpGs, the conversion fraction at non-CpG cytosines in both modes, and the
reference bases of a record's query positions."""
import ctypes as C
import struct

import pytest

import bamutil
import orc

L = orc.lib
L.orc_filter_mask_methylation_depth.restype = C.c_int64
L.orc_filter_mask_methylation_depth.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
L.orc_filter_resolve_ref_bases.restype = C.c_int
L.orc_filter_resolve_ref_bases.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_int16), C.c_uint32, C.POINTER(C.c_uint32)]
L.orc_filter_mask_strand_methylation_agreement.restype = C.c_int64
L.orc_filter_mask_strand_methylation_agreement.argtypes = [C.c_char_p, C.c_uint32]
L.orc_filter_check_conversion_fraction.restype = C.c_int
L.orc_filter_check_conversion_fraction.argtypes = [C.c_char_p, C.c_uint32, C.c_double, C.c_int]

DISABLED, EM_SEQ, TAPS = 0, 1, 2


def i16(tag, values):
    """RawSamBuilder::add_array_i16: a B:s array."""
    return (tag, "raw", b"B" + b"s" + struct.pack("<I", len(values)) + b"".join(struct.pack("<h", v) for v in values))


def rec(seq, cigar, tags=(), ref_id=0, pos=0, flag=0):
    return bamutil.make_record("q", seq, [30] * len(seq), flag=flag, ref_id=ref_id, pos=pos, cigar=cigar, tags=tags)


def set_ref(*seqs):
    bufs = [C.create_string_buffer(s, max(1, len(s))) for s in seqs]
    ptrs = (C.c_void_p * len(bufs))(*[C.cast(b, C.c_void_p).value for b in bufs])
    lens = (C.c_uint64 * len(bufs))(*[len(s) for s in seqs])
    L.orc_set_reference(len(bufs), ptrs, lens)


@pytest.fixture(autouse=True)
def _no_reference_left():
    yield
    L.orc_set_reference(0, None, None)


def depth(record, duplex, thr):
    buf = C.create_string_buffer(record, len(record))
    n = L.orc_filter_mask_methylation_depth(buf, len(record), int(duplex), (C.c_uint32 * 3)(*thr))
    return n, buf.raw


def test_methylation_depth_thresholds_expand_from_the_last_value():
    """test_methylation_depth_thresholds_{single_value,two_values,three_values} (:1621-1643)."""
    assert list(orc.filter_options(min_methylation_depth=[5]).min_methylation_depth) == [5, 5, 5]
    assert list(orc.filter_options(min_methylation_depth=[10, 3]).min_methylation_depth) == [10, 3, 3]
    assert list(orc.filter_options(min_methylation_depth=[10, 5, 2]).min_methylation_depth) == [10, 5, 2]


SIMPLEX_DEPTH = [   # (reference test, cu, ct, min depth, masked, masked positions)
    ("test_mask_methylation_depth_simplex_all_pass", [5, 5, 5, 5], [3, 3, 3, 3], 5, 0, []),
    ("test_mask_methylation_depth_simplex_some_fail", [5, 1, 0, 10], [3, 1, 0, 0], 5, 2, [1, 2]),
]


@pytest.mark.parametrize("name,cu,ct,min_depth,want,where", SIMPLEX_DEPTH, ids=[c[0] for c in SIMPLEX_DEPTH])
def test_mask_methylation_depth_simplex(name, cu, ct, min_depth, want, where):
    r = rec("ACGT", "4M", [i16("cu", cu), i16("ct", ct)])
    n, out = depth(r, False, [min_depth] * 3)
    assert n == want
    p = bamutil.parse(out)
    assert [i for i, b in enumerate(p["seq"]) if b == "N"] == where
    assert [q for q in p["quals"]] == [2 if i in where else 30 for i in range(4)]


def test_mask_methylation_depth_simplex_no_tags_no_masking():
    """(:1676-1686)"""
    r = rec("ACGT", "4M")
    n, out = depth(r, False, [5, 5, 5])
    assert n == 0 and out == r


DUPLEX_DEPTH = [   # (reference test, au, masked)
    ("test_mask_methylation_depth_duplex_all_pass", [5, 5, 5, 5], 0),
    ("test_mask_methylation_depth_duplex_ab_fails", [5, 0, 5, 5], 1),
]


@pytest.mark.parametrize("name,au,want", DUPLEX_DEPTH, ids=[c[0] for c in DUPLEX_DEPTH])
def test_mask_methylation_depth_duplex(name, au, want):
    r = rec("ACGT", "4M", [i16("cu", [10] * 4), i16("ct", [2] * 4), i16("au", au), i16("at", [1] * 4), i16("bu", [5] * 4), i16("bt", [1] * 4)])
    n, out = depth(r, True, [5, 3, 3])
    assert n == want
    assert [i for i, b in enumerate(bamutil.parse(out)["seq"]) if b == "N"] == ([1] if want else [])


def strand(record):
    buf = C.create_string_buffer(record, len(record))
    return L.orc_filter_mask_strand_methylation_agreement(buf, len(record)), buf.raw


def cpg_record(seq, bu6, bt6):
    z = [0] * 11
    au, at, bu, bt = list(z), list(z), list(z), list(z)
    au[5], at[5], bu[6], bt[6] = 10, 1, bu6, bt6
    return rec(seq, "11M", [i16("au", au), i16("at", at), i16("bu", bu), i16("bt", bt)])


def test_strand_methylation_agreement_concordant():
    """(:1730-1758) both strands call the CpG at 5 / 6 methylated."""
    set_ref(b"AAAAACGAAAA")
    n, out = strand(cpg_record("AAAAACGAAAA", 10, 1))
    assert n == 0


def test_strand_methylation_agreement_discordant():
    """(:1761-1789) top methylated, bottom not: both bases of the CpG are masked."""
    set_ref(b"AAAAACGAAAA")
    r = cpg_record("AAAAACGAAAA", 1, 10)
    n, out = strand(r)
    assert n == 2
    p = bamutil.parse(out)
    assert p["seq"] == "AAAAANNAAAA" and list(p["quals"]) == [30] * 5 + [2, 2] + [30] * 4


def test_strand_methylation_agreement_non_cpg_ignored():
    """(:1792-1818)"""
    set_ref(b"ACAGTGCATGA")
    z = [0] * 11
    r = rec("ACAGTGCATGA", "11M", [i16("au", z), i16("at", z), i16("bu", z), i16("bt", z)])
    n, out = strand(r)
    assert n == 0 and out == r


def conv(record, frac, mode):
    return bool(L.orc_filter_check_conversion_fraction(record, len(record), frac, mode))


HI_CT = ([0, 1, 0, 0, 0, 1, 0, 0, 0], [0, 9, 0, 0, 0, 9, 0, 0, 0])     # (cu, ct): 18 of 20 converted at the non-CpG Cs 1 and 5
HI_CU = ([0, 9, 0, 0, 0, 9, 0, 0, 0], [0, 1, 0, 0, 0, 1, 0, 0, 0])     # 2 of 20 converted
CONVERSION = [   # (reference test, (cu, ct), threshold, mode, passes)
    ("test_conversion_fraction_passes_high_conversion", HI_CT, 0.8, EM_SEQ, True),
    ("test_conversion_fraction_fails_low_conversion", HI_CU, 0.8, EM_SEQ, False),
    ("test_conversion_fraction_taps_passes_high_non_conversion", HI_CU, 0.8, TAPS, True),
    ("test_conversion_fraction_taps_fails_low_non_conversion", HI_CT, 0.8, TAPS, False),
    ("test_conversion_fraction_taps_vs_emseq_inverted/em-seq", HI_CU, 0.8, EM_SEQ, False),
    ("test_conversion_fraction_taps_vs_emseq_inverted/taps", HI_CU, 0.8, TAPS, True),
    ("test_conversion_fraction_disabled_mode_passes", HI_CU, 0.8, DISABLED, True),
]


@pytest.mark.parametrize("name,counts,frac,mode,want", CONVERSION, ids=[c[0] for c in CONVERSION])
def test_conversion_fraction(name, counts, frac, mode, want):
    set_ref(b"ACATACATA")
    r = rec("ACATACATA", "9M", [i16("cu", counts[0]), i16("ct", counts[1])])
    assert conv(r, frac, mode) is want


def test_conversion_fraction_skips_cpg():
    """(:1893-1924) the only C with evidence sits in a CpG: nothing to evaluate, passes."""
    set_ref(b"AAAACGAAA")
    r = rec("AAAACGAAA", "9M", [i16("cu", [0, 0, 0, 0, 10, 0, 0, 0, 0]), i16("ct", [0] * 9)])
    assert conv(r, 0.9, EM_SEQ) is True


def test_conversion_fraction_no_methylation_tags_passes():
    """(:1928-1955)"""
    set_ref(b"ACATACATA")
    assert conv(rec("ACATACATA", "9M"), 0.9, EM_SEQ) is True


def test_conversion_fraction_unmapped_passes():
    """(:1959-1983) RawSamBuilder's default record: no reference id, no CIGAR."""
    set_ref(b"ACATACATA")
    r = bamutil.make_record("q", "ACATACATA", [30] * 9, flag=0, ref_id=-1, pos=-1, cigar="", tags=[i16("cu", HI_CU[0]), i16("ct", HI_CU[1])])
    assert conv(r, 0.9, EM_SEQ) is True
    r = bamutil.make_record("q", "ACATACATA", [30] * 9, flag=0x4, ref_id=0, pos=0, cigar="9M", tags=[i16("cu", HI_CU[0]), i16("ct", HI_CU[1])])
    assert conv(r, 0.9, EM_SEQ) is True                       # (the unmapped flag alone: resolve_ref_bases_for_record :1077-1080)


def ref_bases(record):
    out = (C.c_int16 * 64)()
    n = C.c_uint32()
    some = L.orc_filter_resolve_ref_bases(record, len(record), out, 64, C.byref(n))
    return None if not some else [None if out[i] < 0 else chr(out[i]) for i in range(n.value)]


def test_resolve_ref_bases_simple_match():
    """(:2138-2151)"""
    set_ref(b"ACGTACGTAC")
    assert ref_bases(rec("ACGT", "4M")) == ["A", "C", "G", "T"]


def test_resolve_ref_bases_with_insertion():
    """(:2155-2172) 2M2I2M: the inserted bases have no reference base."""
    set_ref(b"ACGTACGTAC")
    assert ref_bases(rec("ACNNGT", "2M2I2M")) == ["A", "C", None, None, "G", "T"]


def test_resolve_ref_bases_beyond_the_restated_tests():
    """Deletions and skips advance the reference, soft clips do not; lower-case reference bases come back upper-cased; past the contig end
    and past the CIGAR there is no base (filter.rs:1098-1130)."""
    set_ref(b"acgtACGTac")
    assert ref_bases(rec("TTACGT", "2S2M2D2M")) == [None, None, "A", "C", "A", "C"]
    assert ref_bases(rec("ACGT", "2M3N2M")) == ["A", "C", "C", "G"]
    assert ref_bases(rec("ACGT", "4M", pos=8)) == ["A", "C", None, None]
    assert ref_bases(rec("ACGT", "2M")) == ["A", "C", None, None]
    assert ref_bases(rec("ACGT", "4M", ref_id=-1)) is None
