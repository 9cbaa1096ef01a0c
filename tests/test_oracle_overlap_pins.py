"""Oracle pins: the unit tests of the reference's overlapping-bases consensus (`crates/fgumi-consensus/src/overlapping.rs`,
`#[cfg(test)] mod tests`, `:690-1460`) for the strategy pair the consensus commands use (agreement = Consensus, disagreement =
Consensus), transcribed case by case: `make_raw_bam` / `create_raw_test_record` (`:744-815`) build the same records (no aux data,
mate fields -1), `caller.call(r1, r2)` is the oracle's `overlapping_call`, `apply_overlapping_consensus` its group-level step.
(`test_raw_disagreement_strategy_consensus` runs with agreement = PassThrough in the reference; the position it asserts is a
disagreement, where the two strategy pairs do the same thing.)"""
import struct

import numpy as np

import bamutil
import orc

F_PAIRED, F_UNMAPPED, F_FIRST, F_LAST, F_SECONDARY, F_SUPPLEMENTARY = 0x1, 0x4, 0x40, 0x80, 0x100, 0x800


def make_raw_bam(name, flag, tid, pos0, cigar, seq, qual):
    """`make_raw_bam` (overlapping.rs:744-796): bin 0, mapq 60, mate tid / pos -1, tlen 0, no aux."""
    rec = bytearray(bamutil.make_record(name, seq, list(qual), flag=flag, ref_id=tid, pos=pos0, mapq=60, cigar=cigar, mate_ref=-1, mate_pos=-1, tlen=0))
    rec[10:12] = struct.pack("<H", 0)
    return rec


def raw(seq, qual, start_1based, cigar):
    """`create_raw_test_record` (overlapping.rs:805-815): name "rea", flag 0, reference 0."""
    return make_raw_bam("rea", 0, 0, start_1based - 1, cigar, seq, qual)


def call(r1, r2):
    a, b = np.frombuffer(bytes(r1), dtype=np.uint8).copy(), np.frombuffer(bytes(r2), dtype=np.uint8).copy()
    st = np.zeros(4, dtype=np.uint64)
    ok = orc.lib.orc_overlap_pair(orc.ptr(a), len(a), orc.ptr(b), len(b), orc.ptr(st))
    pa, pb = bamutil.parse(bytes(a)), bamutil.parse(bytes(b))
    return bool(ok), pa, pb, dict(overlapping=int(st[0]), agreeing=int(st[1]), disagreeing=int(st[2]), corrected=int(st[3]))


def apply(records):
    blob = np.frombuffer(b"".join(bytes(r) for r in records), dtype=np.uint8).copy()
    lens = np.array([len(r) for r in records], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    st = np.zeros(4, dtype=np.uint64)
    orc.lib.orc_apply_overlapping(orc.ptr(blob), orc.ptr(offs), orc.ptr(lens), len(records), orc.ptr(st))
    out = [bamutil.parse(bytes(blob[int(o):int(o) + int(n)])) for o, n in zip(offs, lens)]
    return out, dict(overlapping=int(st[0]), agreeing=int(st[1]), disagreeing=int(st[2]), corrected=int(st[3]))


def test_raw_agreement_strategy_consensus():  # overlapping.rs:817-833
    ok, a, b, _ = call(raw("ACGT", [30] * 4, 100, "4M"), raw("ACGT", [20] * 4, 100, "4M"))
    assert ok and a["quals"][0] == 50 and b["quals"][0] == 50


def test_raw_disagreement_strategy_consensus():  # overlapping.rs:874-894
    ok, a, b, _ = call(raw("ACGT", [30] * 4, 100, "4M"), raw("GCTA", [20] * 4, 100, "4M"))
    assert ok and a["seq"][0] == "A" and b["seq"][0] == "A" and a["quals"][0] == 10 and b["quals"][0] == 10


def test_raw_disagreement_consensus_equal_quality():  # overlapping.rs:987-1007
    ok, a, b, _ = call(raw("ACGT", [30] * 4, 100, "4M"), raw("GCTA", [30] * 4, 100, "4M"))
    assert ok and a["seq"][0] == "N" and b["seq"][0] == "N" and a["quals"][0] == 2 and b["quals"][0] == 2


def test_raw_no_overlap():  # overlapping.rs:1010-1023
    assert not call(raw("ACGT", [30] * 4, 100, "4M"), raw("ACGT", [20] * 4, 200, "4M"))[0]


def test_raw_unmapped_reads():  # overlapping.rs:1026-1047
    assert not call(make_raw_bam("rea", F_UNMAPPED, 0, 99, "4M", "ACGT", [30] * 4), raw("ACGT", [20] * 4, 100, "4M"))[0]


def test_raw_different_references():  # overlapping.rs:1050-1063
    assert not call(make_raw_bam("rea", 0, 0, 99, "4M", "ACGT", [30] * 4), make_raw_bam("rea", 0, 1, 99, "4M", "ACGT", [20] * 4))[0]


def test_raw_quality_capping_at_93():  # overlapping.rs:1066-1081
    ok, a, _, _ = call(raw("ACGT", [50] * 4, 100, "4M"), raw("ACGT", [50] * 4, 100, "4M"))
    assert ok and a["quals"][0] == 93


def test_raw_stats_tracking():  # overlapping.rs:1084-1101
    ok, _, _, st = call(raw("ACGT", [30] * 4, 100, "4M"), raw("ACGT", [20] * 4, 100, "4M"))
    assert ok and st["overlapping"] > 0 and st["agreeing"] == st["overlapping"] and st["disagreeing"] == 0 and st["corrected"] == st["overlapping"]


def test_raw_stats_tracking_with_disagreements():  # overlapping.rs:1104-1120
    ok, _, _, st = call(raw("ACGT", [30] * 4, 100, "4M"), raw("TGCA", [20] * 4, 100, "4M"))
    assert ok and st["overlapping"] > 0 and st["agreeing"] == 0 and st["disagreeing"] == st["overlapping"]


def test_raw_overlap_different_start_positions():  # overlapping.rs:1123-1143
    ok, a, _, st = call(raw("ACGT", [30] * 4, 100, "4M"), raw("GTAC", [20] * 4, 102, "4M"))
    assert ok and st["overlapping"] == 2 and st["agreeing"] == 2 and a["quals"][2] == 50 and a["quals"][3] == 50


def test_raw_cigar_with_soft_clips():  # overlapping.rs:1146-1168
    ok, a, _, st = call(raw("NNACGT", [2, 2, 30, 30, 30, 30], 100, "2S4M"), raw("ACGT", [20] * 4, 100, "4M"))
    assert ok and st["overlapping"] == 4 and st["agreeing"] == 4 and a["quals"][2:6] == [50] * 4


def test_raw_cigar_with_insertions():  # overlapping.rs:1171-1185
    assert call(raw("ACTTGG", [30] * 6, 100, "2M2I2M"), raw("ACGG", [20] * 4, 100, "4M"))[0]


def test_raw_cigar_with_deletions():  # overlapping.rs:1188-1202
    assert call(raw("ACGG", [30] * 4, 100, "2M2D2M"), raw("ACTTGG", [20] * 6, 100, "6M"))[0]


def test_raw_disagreement_consensus_min_quality():  # overlapping.rs:1205-1221: a difference of 1 is raised to the minimum quality 2
    ok, a, b, _ = call(raw("ACGT", [30] * 4, 100, "4M"), raw("GCTA", [29] * 4, 100, "4M"))
    assert ok and a["quals"][0] == 2 and b["quals"][0] == 2


def test_apply_overlapping_consensus_pair():  # overlapping.rs:1224-1262
    out, st = apply([make_raw_bam("rea", F_PAIRED | F_FIRST, 0, 99, "4M", "ACGT", [30] * 4), make_raw_bam("rea", F_PAIRED | F_LAST, 0, 99, "4M", "ACGT", [20] * 4)])
    assert out[0]["quals"][0] == 50 and out[1]["quals"][0] == 50 and st["overlapping"] > 0


def test_apply_overlapping_consensus_no_pair():  # overlapping.rs:1265-1303: different names are not a pair
    out, st = apply([make_raw_bam("rea", F_PAIRED | F_FIRST, 0, 99, "4M", "ACGT", [30] * 4), make_raw_bam("reb", F_PAIRED | F_LAST, 0, 99, "4M", "ACGT", [20] * 4)])
    assert out[0]["quals"][0] == 30 and out[1]["quals"][0] == 20 and st["overlapping"] == 0


def test_apply_overlapping_consensus_reversed_indices():  # overlapping.rs:1306-1346: R2 stored before R1
    out, st = apply([make_raw_bam("rea", F_PAIRED | F_LAST, 0, 99, "4M", "ACGT", [20] * 4), make_raw_bam("rea", F_PAIRED | F_FIRST, 0, 99, "4M", "ACGT", [30] * 4)])
    assert st["overlapping"] > 0 and out[0]["quals"][0] == out[1]["quals"][0]


def test_apply_overlapping_consensus_skips_supplementary_and_secondary():  # overlapping.rs:1349-1405, 1408-1461
    for extra in (F_SUPPLEMENTARY, F_SECONDARY):
        out, _ = apply([make_raw_bam("rea", F_PAIRED | F_FIRST, 0, 99, "4M", "ACGT", [30] * 4), make_raw_bam("rea", F_PAIRED | F_LAST, 0, 99, "4M", "ACGT", [20] * 4),
                        make_raw_bam("rea", F_PAIRED | F_FIRST | extra, 0, 99, "4M", "ACGT", [10] * 4)])
        assert out[0]["quals"][0] == 50 and out[1]["quals"][0] == 50 and out[2]["quals"][0] == 10
