"""Oracle pins: the unit tests of `num_bases_extending_past_mate_raw` (`crates/fgumi-raw-bam/src/overlap.rs`, `mod tests`,
`:1224-2160`) — the mate-overlap clip every source read of the simplex / duplex callers goes through (MC-tag path), which the
device kernels evaluate in closed form for `<n>M` reads and op by op in `k_family`.  Records are built like the reference's
`make_bam_bytes[_with_tlen]` (`testutil.rs:193-262`: zeroed sequence / qualities, MAPQ 0, the aux block as given); the oracle
entry is `orc_mate_clip` = `oracle_bam.hpp`'s restatement of `overlap.rs:181-357`."""
import struct

import numpy as np
import pytest

import bamutil
import orc

PAIRED, UNMAPPED, MATE_UNMAPPED, REVERSE, MATE_REVERSE, FIRST, LAST = 0x1, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80


def rec(pos0, flag, cigar, mate_tid, mate_pos0, tlen=0, mc=None, tid=0, name="rea"):
    ops = bamutil.cigar_ops(cigar)
    l_seq = sum(o >> 4 for o in ops if (o & 15) in (0, 1, 4, 7, 8))
    name_b = name.encode() + b"\0"
    head = struct.pack("<iiBBHHHIiii", tid, pos0, len(name_b), 0, 0, len(ops), flag, l_seq, mate_tid, mate_pos0, tlen)
    aux = (b"MCZ" + mc.encode() + b"\0") if mc is not None else b""
    return head + name_b + b"".join(struct.pack("<I", o) for o in ops) + bytes((l_seq + 1) // 2) + bytes(l_seq) + aux


def clip(r):
    a = np.frombuffer(r, dtype=np.uint8).copy()
    return int(orc.lib.orc_mate_clip(orc.ptr(a), len(a)))


def test_not_paired_unmapped_mate_unmapped_same_strand_other_reference_no_mc():  # overlap.rs:1224-1318: every one of them clips nothing
    assert clip(rec(100, 0, "10M", 0, 200)) == 0
    assert clip(rec(100, PAIRED | UNMAPPED | MATE_REVERSE, "10M", 0, 200)) == 0
    assert clip(rec(100, PAIRED | MATE_UNMAPPED | MATE_REVERSE, "10M", -1, -1)) == 0
    assert clip(rec(100, PAIRED, "10M", 0, 200, mc="10M")) == 0
    assert clip(rec(100, PAIRED | MATE_REVERSE, "10M", 1, 200, mc="10M")) == 0
    assert clip(rec(100, PAIRED | MATE_REVERSE, "10M", 0, 200, tlen=110)) == 0


def test_positive_strand_overlap_and_no_overlap():  # overlap.rs:1321-1376
    assert clip(rec(100, PAIRED | MATE_REVERSE, "20M", 0, 105, tlen=20, mc="10M")) == 5
    assert clip(rec(100, PAIRED | MATE_REVERSE, "10M", 0, 200, tlen=110, mc="10M")) == 0


def test_negative_strand_overlap_no_overlap_and_soft_clip_gap():  # overlap.rs:1402-1469
    assert clip(rec(100, PAIRED | REVERSE, "20M", 0, 105, mc="10M")) == 5
    assert clip(rec(200, PAIRED | REVERSE, "10M", 0, 100, mc="10M")) == 0
    assert clip(rec(110, PAIRED | REVERSE, "3S10M", 0, 105, mc="10M")) == 0


def test_oversized_mc_leading_soft_clip_does_not_overflow():  # overlap.rs:1472-1489
    assert clip(rec(100, PAIRED | REVERSE, "20M", 0, 0, mc="9999999999S10M")) == 0


def test_positive_strand_gap_with_soft_clip():  # overlap.rs:1536-1558
    assert clip(rec(100, PAIRED | MATE_REVERSE, "10M3S", 0, 200, tlen=110, mc="10M")) == 0


def test_non_fr_chimeric_reads_clip_nothing():  # overlap.rs:1605-1630, 2138-2160
    assert clip(rec(11576620, PAIRED | MATE_REVERSE | FIRST, "145M124S", 0, 11576412, tlen=-28, mc="87S182M")) == 0
    assert clip(rec(11576412, PAIRED | REVERSE | LAST, "87S182M", 0, 11576620, tlen=28, mc="145M124S")) == 0


def test_soft_only_mate_end():  # overlap.rs:1633-1654: a hard clip after the mate's last aligned base does not extend its unclipped end
    assert clip(rec(100, PAIRED | MATE_REVERSE, "40M", 0, 100, tlen=40, mc="30M5H")) == 10


def test_symmetric_on_dovetail_forward():  # overlap.rs:1770-1802 (the MC path of it)
    assert clip(rec(100, PAIRED | MATE_REVERSE | FIRST, "50M50S", 0, 60, tlen=-90, mc="100M", name="dtl")) == 40


def _deletion_at_boundary_pair():  # overlap.rs:1827-1870 (fgumi#752 / fgbio#1090)
    fwd = rec(83585780, PAIRED | MATE_REVERSE | FIRST, "2S124M1D3M", 0, 83585779, tlen=124, mc="3S124M2S", name="del")
    rev = rec(83585779, PAIRED | REVERSE | LAST, "3S124M2S", 0, 83585780, tlen=-124, mc="2S124M1D3M", name="del")
    return fwd, rev


def test_deletion_at_mate_boundary_clips_the_query_distance():  # overlap.rs:1873-1922
    fwd, rev = _deletion_at_boundary_pair()
    assert clip(fwd) == 2 and clip(rev) == 2 and clip(fwd) < 129


def test_insertion_before_mate_boundary_is_not_under_clipped():  # overlap.rs:1925-1963
    assert clip(rec(1234499, PAIRED | MATE_REVERSE | FIRST, "70M10I23M47S", 0, 1234499, tlen=93, mc="50S70M30S", name="ins")) == 50


def test_ungapped_overlap_is_unchanged():  # overlap.rs:1966-2019
    assert clip(rec(100, PAIRED | MATE_REVERSE | FIRST, "40M", 0, 100, tlen=40, mc="10S30M10S", name="pln")) == 0
    assert clip(rec(100, PAIRED | REVERSE | LAST, "10S30M10S", 0, 100, tlen=-40, mc="40M", name="pln")) == 10


def test_read_entirely_past_mate():  # overlap.rs:2022-2047
    assert clip(rec(300, PAIRED | MATE_REVERSE, "20M", 0, 100, tlen=100, mc="10M5S")) == 0


def _clips_for_fr_pair(s1, c1, s2, c2):  # overlap.rs:2080-2129
    def ref_len(c):
        return sum(o >> 4 for o in bamutil.cigar_ops(c) if (o & 15) in (0, 2, 3, 7, 8))
    tlen = (s2 + ref_len(c2) - 1) - s1 + 1
    fwd = rec(s1 - 1, PAIRED | MATE_REVERSE | FIRST, c1, 0, s2 - 1, tlen=tlen, mc=c2, name="pair")
    rev = rec(s2 - 1, PAIRED | REVERSE | LAST, c2, 0, s1 - 1, tlen=-tlen, mc=c1, name="pair")
    return clip(fwd), clip(rev)


def test_disjoint_alignments_still_clip_their_read_through():  # overlap.rs:2132-2135
    assert _clips_for_fr_pair(1000, "20M80S", 1019, "80S20M") == (61, 61)
    assert _clips_for_fr_pair(1000, "20M80S", 1020, "80S20M") == (60, 60)
