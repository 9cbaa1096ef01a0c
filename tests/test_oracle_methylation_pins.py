"""Oracle pins: the methylation-aware mode (EM-Seq / TAPs).  The unit tests of `crates/fgumi-consensus/src/methylation.rs`
(`#[cfg(test)] mod tests`, `:461-926`), the simplex caller's methylation tests (`vanilla_caller.rs:5856-6250`) and the duplex caller's
(`duplex_caller.rs:6741-7170`), transcribed case by case: same reads, same reference sequence, same options, same assertions.
`replay_cases()` hands the caller-level inputs to the GPU parity suite (tests/test_gpu_methylation.py)."""
import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records

F_PAIRED, F_REVERSE, F_MATE_REVERSE, F_FIRST, F_LAST = 0x1, 0x10, 0x20, 0x40, 0x80
EM, TAPS, OFF = orc.METH_EM_SEQ, orc.METH_TAPS, orc.METH_DISABLED

_REPLAY = []   # (option keywords, reference contigs, MI groups)


# ---- query_to_ref_positions (methylation.rs:461-525) -------------------------------------------------------------------------

def test_query_to_ref_positions_all_matches():
    assert orc.meth_query_to_ref_positions("10M", 100, False, "10M") == [100 + i for i in range(10)]


def test_query_to_ref_positions_with_insertion():
    p = orc.meth_query_to_ref_positions("5M2I3M", 100, False, "5M2I3M")
    assert p == [100, 101, 102, 103, 104, None, None, 105, 106, 107]


def test_query_to_ref_positions_with_deletion():
    p = orc.meth_query_to_ref_positions("5M2D5M", 100, False, "5M2D5M")
    assert p == [100, 101, 102, 103, 104, 107, 108, 109, 110, 111]


def test_query_to_ref_positions_reverse_strand():
    assert orc.meth_query_to_ref_positions("10M", 100, True, "10M") == [109 - i for i in range(10)]


def test_query_to_ref_positions_reverse_strand_with_a_deletion_and_a_truncated_read():
    # not a reference test: the reverse walk over a reversed, truncated CIGAR (create_source_read reverses 6M2D4M to 4M2D6M and a
    # mate clip cuts it to 7 query bases); the span comes from the ORIGINAL ops: 100 + 12 - 1 = 111
    assert orc.meth_query_to_ref_positions("4M2D3M", 100, True, "6M2D4M") == [111, 110, 109, 108, 105, 104, 103]


# ---- annotate_simplex_methylation (methylation.rs:527-620) -------------------------------------------------------------------

def test_annotate_simplex_all_methylated():
    c, u, t = orc.meth_annotate(4, ["ACGT", "ACGT"], list("ACGT"), True)
    assert c[1] and u[1] == 2 and t[1] == 0
    assert c == [False, True, False, False]


def test_annotate_simplex_all_unmethylated():
    c, u, t = orc.meth_annotate(4, ["ATGT", "ATGT"], list("ACGT"), True)
    assert c[1] and u[1] == 0 and t[1] == 2


def test_annotate_simplex_mixed():
    c, u, t = orc.meth_annotate(4, ["ACGT", "ATGT"], list("ACGT"), True)
    assert c[1] and u[1] == 1 and t[1] == 1


def test_annotate_simplex_non_c_positions():
    c, u, t = orc.meth_annotate(4, ["AGGT"], list("AGGT"), True)
    assert not any(c)


def test_annotate_simplex_reverse_strand():
    c, u, t = orc.meth_annotate(4, ["CAGT"], list("TGCA"), False)
    assert c[1] and u[1] == 0 and t[1] == 1


def test_annotate_positions_past_the_reference_or_inside_insertions_are_not_reference_cytosines():
    c, u, t = orc.meth_annotate(5, ["CCCCC", "CCC"], ["C", None, "c"], True)      # lower-case reference bases count (to_ascii_uppercase)
    assert c == [True, False, True, False, False] and u == [2, 0, 2, 0, 0]


# ---- build_mm_ml_tags (methylation.rs:622-800) -------------------------------------------------------------------------------

def test_build_mm_ml_tags_basic():
    ev = [(0, 0, 0), (1, 3, 0), (0, 0, 0), (1, 0, 3), (0, 0, 0), (0, 0, 0)]
    assert orc.meth_build_mm_ml("ACGCAC", ev, True, EM) == ("C+m,0,0;", [255, 0])


def test_build_mm_ml_tags_no_modifications():
    assert orc.meth_build_mm_ml("AGGT", [(0, 0, 0)] * 4, True, EM) is None


def test_build_mm_tag_no_ml():
    mm, _ = orc.meth_build_mm_ml("ACGT", [(0, 0, 0), (1, 2, 1), (0, 0, 0), (0, 0, 0)], True, EM)
    assert mm == "C+m,0;"


def test_build_mm_ml_with_skips():
    ev = [(1, 5, 0), (0, 0, 0), (0, 0, 0), (1, 0, 5), (0, 0, 0)]
    assert orc.meth_build_mm_ml("CCACC", ev, True, EM) == ("C+m,0,1;", [255, 0])


def test_build_mm_ml_tags_bottom_strand():
    ev = [(0, 0, 0), (1, 3, 0), (0, 0, 0), (1, 0, 3), (0, 0, 0), (0, 0, 0)]
    assert orc.meth_build_mm_ml("AGCGAG", ev, False, EM) == ("G-m,0,0;", [255, 0])


def test_build_mm_ml_tags_taps_all_methylated():
    mm, ml = orc.meth_build_mm_ml("CCCCC", [(1, 0, 3)] * 5, True, TAPS)
    assert mm.startswith("C+m") and ml == [255] * 5


def test_build_mm_ml_tags_taps_all_unmethylated():
    assert orc.meth_build_mm_ml("CCCCC", [(1, 3, 0)] * 5, True, TAPS)[1] == [0] * 5


def test_build_mm_ml_tags_emseq_unchanged():
    assert orc.meth_build_mm_ml("CCCCC", [(1, 3, 0)] * 5, True, EM)[1] == [255] * 5


def test_build_mm_ml_a_reference_cytosine_without_evidence_is_skipped_and_length_mismatch_panics():
    assert orc.meth_build_mm_ml("CCC", [(1, 0, 0), (1, 1, 2), (1, 0, 0)], True, EM) == ("C+m,1;", [85])     # 1 * 255 / 3
    with pytest.raises(RuntimeError):
        orc.meth_build_mm_ml("CCC", [(1, 1, 0)] * 2, True, EM)


def test_is_top_strand():  # methylation.rs:676-686
    assert orc.lib.orc_meth_is_top_strand(F_PAIRED | F_FIRST)
    assert not orc.lib.orc_meth_is_top_strand(F_PAIRED | F_FIRST | F_REVERSE)
    assert not orc.lib.orc_meth_is_top_strand(F_PAIRED | F_LAST)
    assert orc.lib.orc_meth_is_top_strand(F_PAIRED | F_LAST | F_REVERSE)


def test_combine_methylation_annotations():  # :688-708
    got = orc.meth_combine([(1, 2, 1), (0, 0, 0)], [(1, 1, 2), (0, 0, 0)], 2)
    assert got == [(True, 3, 3), (False, 0, 0)]


def test_methylation_counters_saturate():  # :774-790
    m = 0xFFFFFFFF
    assert orc.meth_combine([(1, m, m)], [(1, m, m)], 1) == [(True, m, m)]


@pytest.mark.parametrize("ref,pos,top,want", [
    (b"ACGT", 1, True, True), (b"ACAT", 1, True, False), (b"ACCT", 1, True, False), (b"ACTT", 1, True, False), (b"AAC", 2, True, False),
    (b"AGGT", 1, True, False), (b"ACGT", 2, False, True), (b"AAGT", 2, False, False), (b"AGGT", 2, False, False), (b"ATGT", 2, False, False),
    (b"GAC", 0, False, False), (b"ACAT", 1, False, False), (b"acgt", 1, True, True), (b"acgt", 2, False, True), (b"CG", 2, True, False),
    (b"CG", 2, False, False), (b"CG", 100, True, False), (b"CG", 100, False, False), (b"", 0, True, False), (b"", 0, False, False)])
def test_is_cpg_context(ref, pos, top, want):  # methylation.rs:846-920
    assert bool(orc.lib.orc_meth_is_cpg_context(ref, len(ref), pos, int(top))) == want


# ---- the simplex caller (vanilla_caller.rs:5856-6250) ------------------------------------------------------------------------

def meth_opts(mode, **kw):
    """`create_methylation_caller` (vanilla_caller.rs:5862-5878): min_reads 1, min consensus base quality 0, the library defaults
    otherwise (no cell tag, prefix "consensus", no overlap pre-step: that belongs to the commands)."""
    base = dict(min_reads=1, min_consensus_base_quality=0, overlapping_consensus=0, cell_tag=b"\0\0", read_name_prefix=b"consensus", methylation_mode=mode)
    base.update(kw)
    o = fgx_opts.defaults(**base)
    o._kw = dict(base)
    return o


def test_read(name, bases, quals, umi):
    """`create_consensus_test_read` (vanilla_caller.rs:3054-3066): unpaired, ref 0, pos 99, one M op, MI tag."""
    return bamutil.make_record(name, bases, list(quals), flag=0, ref_id=0, pos=99, tags=[("MI", "Z", umi)])


test_read.__test__ = False


def call(opts, reference, groups, batch_groups=50):
    g = GroupedReads.from_groups(groups)
    orc.set_reference(reference)
    try:
        res = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    finally:
        orc.set_reference(None)
    _REPLAY.append((dict(opts._kw), [bytes(s) for s in (reference or [])], [list(x) for x in groups]))
    return res, [bamutil.parse(r) for r in split_records(res["data"])]


REF_C = b"N" * 99 + b"CCCCCCCCCC"
Q30 = [30] * 10


def _tags(rec):
    return {k: v[1] for k, v in rec["tags"].items()}


@pytest.mark.parametrize("mode,reads,bases,mm,ml,cu,ct", [
    (EM, ["C" * 10] * 3, "C" * 10, "C+m" + ",0" * 10 + ";", [255] * 10, [3] * 10, [0] * 10),        # test_simplex_em_seq_all_methylated :5887
    (EM, ["T" * 10] * 3, "C" * 10, "C+m" + ",0" * 10 + ";", [0] * 10, [0] * 10, [3] * 10),          # ..._all_unmethylated :5933
    (EM, ["C" * 10, "C" * 10, "T" * 10], "C" * 10, "C+m" + ",0" * 10 + ";", [170] * 10, [2] * 10, [1] * 10),   # ..._mixed_methylation :5970
    (TAPS, ["C" * 10] * 3, "C" * 10, "C+m" + ",0" * 10 + ";", [0] * 10, [3] * 10, [0] * 10),        # test_simplex_taps_all_unmethylated :6120
    (TAPS, ["T" * 10] * 3, "C" * 10, "C+m" + ",0" * 10 + ";", [255] * 10, [0] * 10, [3] * 10),      # test_simplex_taps_all_methylated :6158
    (TAPS, ["C" * 10, "C" * 10, "T" * 10], "C" * 10, "C+m" + ",0" * 10 + ";", [85] * 10, [2] * 10, [1] * 10),   # ..._taps_mixed :6194, taps_vs_emseq :6225
])
def test_simplex_methylation_over_a_run_of_reference_cytosines(mode, reads, bases, mm, ml, cu, ct):
    res, recs = call(meth_opts(mode), [REF_C], [[test_read(f"r{i + 1}", s, Q30, "UMI1") for i, s in enumerate(reads)]])
    assert res["count"] == 1
    t = _tags(recs[0])
    assert recs[0]["seq"] == bases
    assert t["MM"] == mm and t["ML"] == ml and t["cu"] == cu and t["ct"] == ct
    assert recs[0]["tags"]["ML"][0] == "B" and recs[0]["tag_order"][-4:] == ["MM", "ML", "cu", "ct"]


def test_simplex_em_seq_non_c_positions():  # vanilla_caller.rs:6003-6030
    res, recs = call(meth_opts(EM), [b"N" * 99 + b"AAAAAAAAAA"], [[test_read("r1", "A" * 10, Q30, "UMI1"), test_read("r2", "A" * 10, Q30, "UMI1")]])
    t = _tags(recs[0])
    assert res["count"] == 1 and recs[0]["seq"] == "A" * 10
    assert "MM" not in t and "ML" not in t and t["cu"] == [0] * 10 and t["ct"] == [0] * 10


def test_simplex_em_seq_disabled():  # :6034-6061
    res, recs = call(meth_opts(OFF), [REF_C], [[test_read("r1", "T" * 10, Q30, "UMI1"), test_read("r2", "T" * 10, Q30, "UMI1")]])
    t = _tags(recs[0])
    assert res["count"] == 1 and recs[0]["seq"] == "T" * 10
    assert not ({"MM", "ML", "cu", "ct"} & set(t))


def test_simplex_em_seq_longest_read_used_for_mapping():  # :6066-6107
    reads = [test_read("r1", "C" * 5, [30] * 5, "UMI1"), test_read("r2", "C" * 10, Q30, "UMI1"), test_read("r3", "C" * 10, Q30, "UMI1")]
    res, recs = call(meth_opts(EM), [REF_C], [reads])
    t = _tags(recs[0])
    assert res["count"] == 1 and len(recs[0]["seq"]) == 10
    assert t["cu"] == [3] * 5 + [2] * 5 and t["ct"] == [0] * 10 and t["MM"].startswith("C+m") and len(t["ML"]) == 10


def test_simplex_methylation_mode_without_a_reference_calls_the_plain_consensus():
    # annotate_and_normalize returns (None, reads) when no reference was set (vanilla_caller.rs:792-797)
    res, recs = call(meth_opts(EM), None, [[test_read("r1", "T" * 10, Q30, "UMI1"), test_read("r2", "T" * 10, Q30, "UMI1")]])
    assert recs[0]["seq"] == "T" * 10 and not ({"MM", "ML", "cu", "ct"} & set(_tags(recs[0])))


def test_simplex_anchor_outside_the_header_or_unplaced_gets_no_annotation():
    # ref_id beyond ref_names / negative ref_id or start → (None, reads) (:812-818)
    r = [bamutil.make_record(f"r{i}", "T" * 10, Q30, flag=0, ref_id=3, pos=99, tags=[("MI", "Z", "U")]) for i in range(2)]
    _, recs = call(meth_opts(EM), [REF_C], [r])
    assert recs[0]["seq"] == "T" * 10 and "cu" not in _tags(recs[0])
    r = [bamutil.make_record(f"r{i}", "T" * 10, Q30, flag=0, ref_id=-1, pos=-1, tags=[("MI", "Z", "U")]) for i in range(2)]
    _, recs = call(meth_opts(EM), [REF_C], [r])
    assert recs[0]["seq"] == "T" * 10 and "cu" not in _tags(recs[0])


def test_simplex_reverse_strand_fragment_counts_g_and_a_against_reference_g():
    # A reverse-strand fragment is the bottom strand (is_top_strand: reverse == R2 is false).  Its source read is the reverse
    # complement of the stored bases, the reference walk runs from the alignment end down, and the reference bases are compared AS
    # THE FASTA HOLDS THEM (not complemented): target G, unconverted G, converted A in the source read's own orientation
    # (annotate_simplex_methylation, methylation.rs:208-236).
    ref = b"N" * 99 + b"AAGAAAAGAA"                       # 0-based 99..108, G at 101 and 106
    stored = ["AACAAAACAA", "AACAAAACAA", "AATAAAATAA"]  # source reads: TTGTTTTGTT x 2, TTATTTTATT
    reads = [bamutil.make_record(f"r{i}", s, Q30, flag=F_REVERSE, ref_id=0, pos=99, tags=[("MI", "Z", "U")]) for i, s in enumerate(stored)]
    res, recs = call(meth_opts(EM), [ref], [reads])
    t = _tags(recs[0])
    assert recs[0]["seq"] == "TTGTTTTGTT"               # A normalised to G at the two reference G's (walked 108 → 99: query 2 and 7)
    assert t["cu"] == [0, 0, 2, 0, 0, 0, 0, 2, 0, 0] and t["ct"] == [0, 0, 1, 0, 0, 0, 0, 1, 0, 0]
    assert t["MM"] == "G-m,0,0;" and t["ML"] == [170, 170]
    # a reverse read that simply matches the reference shows C where the reference has G: nothing is counted, no MM tag
    reads = [bamutil.make_record(f"r{i}", "AAGAAAAGAA", Q30, flag=F_REVERSE, ref_id=0, pos=99, tags=[("MI", "Z", "U")]) for i in range(2)]
    _, recs = call(meth_opts(EM), [ref], [reads])
    t = _tags(recs[0])
    assert recs[0]["seq"] == "TTCTTTTCTT" and t["cu"] == [0] * 10 and t["ct"] == [0] * 10 and "MM" not in t


def test_simplex_paired_family_annotates_each_end_on_its_own_strand():
    # R1 forward = top strand (C / T against reference C); R2 reverse = top strand too (reverse == R2): `is_top_strand`.
    ref = b"N" * 200 + b"ACGTCCGTAC" + b"N" * 40 + b"GGCATCGTCA" + b"N" * 50
    s1 = ["ACGTCCGTAC", "ATGTCTGTAC", "ACGTCCGTAC"]
    s2 = ["GGCATCGTCA", "GGTATTGTCA", "GGCATCGTCA"]
    reads = []
    for i in range(3):
        reads += list(bamutil.pair(f"p{i}", s1[i], 30, s2[i], 30, "U", pos1=200, pos2=250))
    res, recs = call(meth_opts(EM, min_reads=1), [ref], [reads])
    assert res["count"] == 2
    t1, t2 = _tags(recs[0]), _tags(recs[1])
    assert recs[0]["seq"] == "ACGTCCGTAC" and t1["cu"] == [0, 2, 0, 0, 3, 2, 0, 0, 0, 3] and t1["ct"] == [0, 1, 0, 0, 0, 1, 0, 0, 0, 0]
    assert t1["MM"] == "C+m,0,0,0,0;" and t1["ML"] == [170, 255, 170, 255]
    # R2: source reads = revcomp(stored) = TGACGATGCC / TGACAATACC / TGACGATGCC, reference walked 259 → 250 = A C T G C T A C G G:
    # reference C at query 1, 4, 7, where the source reads show G / G / A,G: neither C nor T, so nothing is counted, no MM tag
    assert recs[1]["seq"] == "TGACGATGCC"
    assert t2["cu"] == [0] * 10 and t2["ct"] == [0] * 10 and "MM" not in t2 and "ML" not in t2


def test_simplex_downsampled_family_annotates_the_retained_reads_only():
    # process_subgroup annotates after downsample_filtered_source_reads (vanilla_caller.rs:1560-1614): the counts cover max_reads reads
    reads = [test_read(f"read{i}", "C" * 10 if i % 2 else "T" * 10, Q30, "U") for i in range(9)]
    res, recs = call(meth_opts(EM, max_reads=4), [REF_C], [reads])
    t = _tags(recs[0])
    assert res["count"] == 1 and all(u + c == 4 for u, c in zip(t["cu"], t["ct"])) and t["cD"] == 4


def test_simplex_indel_family_maps_through_the_anchor_cigar():
    # 4M2D6M: query 4.. map to reference 105.. ; 3M1I6M: the inserted base has no reference position
    ref = b"N" * 99 + b"CCCCAACCCCCC" + b"N" * 10
    reads = [bamutil.make_record(f"r{i}", "CTCCCTCCCC", Q30, flag=0, ref_id=0, pos=99, cigar="4M2D6M", tags=[("MI", "Z", "U")]) for i in range(2)]
    _, recs = call(meth_opts(EM), [ref], [reads])
    t = _tags(recs[0])
    assert recs[0]["seq"] == "CCCCCCCCCC" and t["cu"] == [2, 0, 2, 2, 2, 0, 2, 2, 2, 2] and t["ct"] == [0, 2, 0, 0, 0, 2, 0, 0, 0, 0]
    reads = [bamutil.make_record(f"r{i}", "CTCTTCCCCC", Q30, flag=0, ref_id=0, pos=99, cigar="3M1I6M", tags=[("MI", "Z", "U")]) for i in range(2)]
    _, recs = call(meth_opts(EM), [ref], [reads])
    t = _tags(recs[0])
    # reference under the read: C C C | (ins) | C A A C C C → query 3 (inserted T) stays T, query 4 maps to reference 102 (C): T → C,
    # query 5, 6 map to A A: untouched
    assert recs[0]["seq"] == "CCCTCCCCCC" and t["ct"] == [0, 2, 0, 0, 2, 0, 0, 0, 0, 0] and t["cu"] == [2, 0, 2, 0, 0, 0, 0, 2, 2, 2]


# ---- the duplex caller (duplex_caller.rs:6741-7170) --------------------------------------------------------------------------

def duplex_opts(min_reads, mode, **kw):
    mr = list(min_reads) + [min_reads[-1]] * (3 - len(min_reads))
    base = dict(kind=1, overlapping_consensus=0, read_name_prefix=b"consensus", read_group_id=b"RG1", cell_tag=b"\0\0", min_input_base_quality=0,
                produce_per_base_tags=0, trim=0, track_rejects=0, error_rate_pre_umi=45, error_rate_post_umi=40, methylation_mode=mode)
    base.update(kw)
    o = fgx_opts.defaults(**base)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    o._kw = dict(base, duplex_min_reads=tuple(mr))
    return o


def _drec(name, seq, flag, pos, mi):
    return bamutil.make_record(name, seq, Q30, flag=flag, ref_id=0, pos=pos, mapq=0, cigar="10M", mate_ref=0, mate_pos=99, tags=[("MI", "Z", mi), ("RG", "Z", "A")])


def test_duplex_ba_only_methylation_tags_use_bottom_strand():  # duplex_caller.rs:7090-7170
    ref = b"N" * 99 + b"GGGGGGGGGG"
    reads = []
    for i in (1, 2, 3):
        reads += [_drec(f"q{i}", "G" * 10, F_PAIRED | F_FIRST | F_REVERSE, 99, "foo/B"), _drec(f"q{i}", "C" * 10, F_PAIRED | F_LAST | F_MATE_REVERSE, 99, "foo/B")]
    res, recs = call(duplex_opts([1, 1, 0], EM), [ref], [reads], batch_groups=100)
    assert res["count"] == 2
    for r in recs:
        t = _tags(r)
        assert "bu" in t and "bt" in t and "au" not in t and "at" not in t


def test_duplex_both_strands_conversion_is_not_a_disagreement():
    # duplex_consensus's conversion-artifact rule (duplex_caller.rs:988-1034) reached through whole molecules: the A strand reads T
    # at the reference C's its own strand converted, the B strand (same orientation after the R1/R2 swap) reads C; normalisation
    # makes each single-strand consensus C already, so the duplex agrees; the per-strand counts keep the evidence.
    ref = b"N" * 99 + b"ACGTCCGTAC" + b"N" * 90 + b"GGCATCGTCA" + b"N" * 50
    left, right = "ACGTCCGTAC", "GGCATCGTCA"
    conv = lambda s: s.replace("C", "T")
    reads = []
    for i in range(2):
        reads += [bamutil.make_record(f"a{i}", conv(left), Q30, flag=F_PAIRED | F_FIRST | F_MATE_REVERSE, ref_id=0, pos=99, cigar="10M", mate_ref=0, mate_pos=199, tags=[("MI", "Z", "m/A")]),
                  bamutil.make_record(f"a{i}", right, Q30, flag=F_PAIRED | F_LAST | F_REVERSE, ref_id=0, pos=199, cigar="10M", mate_ref=0, mate_pos=99, tags=[("MI", "Z", "m/A")]),
                  bamutil.make_record(f"b{i}", right, Q30, flag=F_PAIRED | F_FIRST | F_REVERSE, ref_id=0, pos=199, cigar="10M", mate_ref=0, mate_pos=99, tags=[("MI", "Z", "m/B")]),
                  bamutil.make_record(f"b{i}", left, Q30, flag=F_PAIRED | F_LAST | F_MATE_REVERSE, ref_id=0, pos=99, cigar="10M", mate_ref=0, mate_pos=199, tags=[("MI", "Z", "m/B")])]
    res, recs = call(duplex_opts([1, 1, 1], EM, produce_per_base_tags=1), [ref], [reads], batch_groups=100)
    assert res["count"] == 2
    t = _tags(recs[0])
    assert recs[0]["seq"] == left                      # C everywhere the A strand converted
    assert t["au"] == [0, 0, 0, 0, 0, 0, 0, 0, 0, 0] and t["at"] == [0, 2, 0, 0, 2, 2, 0, 0, 0, 2]      # AB-R1: forward R1 = top strand
    # BA-R2 (forward R2) is the bottom strand: it counts G / A against reference G
    assert t["bu"] == [0, 0, 2, 0, 0, 0, 2, 0, 0, 0] and t["bt"] == [0] * 10
    assert t["cu"] == [0, 0, 2, 0, 0, 0, 2, 0, 0, 0] and t["ct"] == [0, 2, 0, 0, 2, 2, 0, 0, 0, 2]
    assert t["am"] == "C+m,0,0,0,0;" and t["bm"] == "G-m,0,0;"
    assert t["MM"] == "C+m,0,0,0,0;" and t["ML"] == [0, 0, 0, 0]
    order = recs[0]["tag_order"]
    assert order[-10:] == ["am", "au", "at", "bm", "bu", "bt", "MM", "ML", "cu", "ct"]


def replay_cases():
    """(option keywords, reference contigs, MI groups) of every caller-level case above — the GPU suite sends the same inputs
    through the HIP path (tests/test_gpu_methylation.py)."""
    if not _REPLAY:
        mod = globals()
        for name, fn in sorted(mod.items()):
            if not (name.startswith("test_") and callable(fn)) or getattr(fn, "__test__", True) is False:
                continue
            marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
            argsets = [()] if not marks else [a if isinstance(a, tuple) else (a,) for a in marks[0].args[1]]
            for a in argsets:
                try:
                    fn(*a)
                except Exception:
                    pass
    seen, out = set(), []
    for kw, ref, groups in _REPLAY:
        key = (tuple(sorted((k, v) for k, v in kw.items())), tuple(ref), tuple(tuple(x) for x in groups))
        if key not in seen:
            seen.add(key)
            out.append((kw, ref, groups))
    return out
