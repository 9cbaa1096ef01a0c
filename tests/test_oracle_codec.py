"""CODEC caller oracle pinned on the reference's own tests (crates/fgumi-consensus/src/codec_caller.rs:2613-4790,
6867-6982): the fgbio-CAPTURED saturation expectations plus the behavioural cases (orientation, indels, soft clips,
FR classification, thresholds, masking).  The fixture builder restates `create_fr_pair` (:2429-2598) over our own
reference sequence — every assertion is relative to the sequence the reads were cut from."""
import random

import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads, split_records

_rng = random.Random(20260921)
REF = "T" + "".join(_rng.choice("ACGT") for _ in range(1199))
PLACEHOLDER = "A"
COMP = str.maketrans("ACGTN", "TGCAN")


def revcomp(s):
    return s.translate(COMP)[::-1]


def _ops(c):
    return [(o >> 4, "MIDNSHP=X"[o & 15]) for o in bamutil.cigar_ops(c)]


def fr_pair(name, start1, start2, q, cigar1, cigar2, mi="hi", rx="ACC-TGA", rev1=False, rev2=True, ref=REF, extra=()):
    def reflen(c):
        return sum(n for n, k in _ops(c) if k in "M=XDN")

    def seq(start, c):
        s, p = "", start - 1
        for n, k in _ops(c):
            if k in "M=X":
                e = min(p + n, len(ref))
                if p < len(ref):
                    s += ref[p:e] + PLACEHOLDER * (p + n - e)
                p += n
            elif k in "IS":
                s += PLACEHOLDER * n
            elif k in "DN":
                p += n
        return s

    l1, l2 = reflen(cigar1), reflen(cigar2)
    tlen = (start2 + l2 - start1) if start1 <= start2 else -(start1 + l1 - start2)
    s1, s2 = seq(start1, cigar1), seq(start2, cigar2)
    tags = [("MI", "Z", mi)] + ([("RX", "Z", rx)] if rx else []) + list(extra)
    f1 = 0x1 | 0x2 | 0x40 | (0x10 if rev1 else 0) | (0x20 if rev2 else 0)
    f2 = 0x1 | 0x2 | 0x80 | (0x10 if rev2 else 0) | (0x20 if rev1 else 0)
    r1 = bamutil.make_record(name, s1, [q] * len(s1), flag=f1, pos=start1 - 1, cigar=cigar1, mate_ref=0, mate_pos=start2 - 1, tlen=tlen, tags=tags)
    r2 = bamutil.make_record(name, s2, [q] * len(s2), flag=f2, pos=start2 - 1, cigar=cigar2, mate_ref=0, mate_pos=start1 - 1, tlen=-tlen, tags=tags)
    return [r1, r2]


def run(recs, groups=None, **kw):
    base = dict(read_name_prefix=b"codec", read_group_id=b"RG1", cell_tag=b"\0\0", produce_per_base_tags=0, overlapping_consensus=0)
    base.update(kw)
    o = fgx_opts.defaults(kind=2, **base)
    g = GroupedReads.from_groups(groups if groups is not None else [recs])
    return orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=1000)


def one(res):
    recs = split_records(res["data"])
    assert res["count"] == 1 and len(recs) == 1
    return bamutil.parse(recs[0])


REJ = {n: 3 + i for i, n in enumerate(
    "FragmentRead InsufficientReads QualityTooLow Unmapped Mapped TooManyNs MinorityAlignment SecondaryOrSupplementary FailedQC MissingUmi "
    "QualityTrimmed ZeroLengthAfterTrimming InsufficientOverlap OrphanConsensus IndelErrorBetweenStrands ClipOverlapFailed "
    "HighDuplexDisagreement PotentialCollision NotPrimaryFrPair Downsampled Other".split())}


def only_rejection(res, reason, n):
    st = res["stats"]
    assert st[REJ[reason]] == n
    assert sum(st[3:24]) == n and st[2] == n


def test_wholly_overlapping_family_reproduces_reference():  # :2613-2650
    p = one(run(fr_pair("read1", 200, 200, 35, "30M", "30M", rx=None)))
    assert p["seq"] == REF[199:229] and "N" not in p["seq"]
    assert p["flag"] == 4 and p["name"] == "codec:hi"


def test_simple_reads():  # :2669-2713
    p = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M")))
    assert p["seq"] == REF[0:40]
    assert p["tags"]["RX"][1] == "ACC-TGA"
    assert p["tag_order"] == ["RG", "MI", "cD", "cM", "cE", "aD", "aM", "aE", "bD", "bM", "bE", "RX"]
    # 20 duplex positions carry both strands' summed quality, tails the single-strand quality
    q = p["quals"]
    assert len(set(q[10:30])) == 1 and q[10] == 2 * q[0] and len(set(q[:10] + q[30:])) == 1


def test_r1_deletion():  # :2790-2838
    p = one(run(fr_pair("read1", 1, 13, 35, "5M2D25M", "30M")))
    assert p["seq"] == REF[0:5] + REF[7:42] != REF[0:40]


def test_r2_deletion():  # :4369-4412
    p = one(run(fr_pair("read1", 1, 11, 35, "30M", "25M5D5M")))
    assert p["seq"] == REF[0:35] + REF[40:45]


def test_rf_pair_rejected():  # :2842-2880
    res = run(fr_pair("read1", 100, 135, 35, "30M", "30M", rev1=True, rev2=False))
    assert res["count"] == 0
    only_rejection(res, "NotPrimaryFrPair", 2)


def test_dovetail_fr_pair_kept():  # :2886-2923
    res = run(fr_pair("dt", 100, 95, 35, "50M", "50M"))
    assert res["stats"][REJ["NotPrimaryFrPair"]] == 0 and res["count"] == 1


def test_insufficient_reads():  # :2927-2985
    res = run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_min_reads_per_strand=2)
    assert res["count"] == 0
    only_rejection(res, "InsufficientReads", 2)
    assert run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_min_reads_per_strand=1)["count"] == 1


def test_insufficient_overlap():  # :2989-3057
    assert run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_min_duplex_length=20)["count"] == 1
    res = run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_min_duplex_length=21)
    assert res["count"] == 0
    only_rejection(res, "InsufficientOverlap", 2)


def test_unmapped_mate():  # :3114-3146
    r = fr_pair("read1", 1, 11, 35, "30M", "30M")
    p = bamutil.parse(r[1])
    b = bytearray(r[1])
    b[14:16] = (p["flag"] | 0x4).to_bytes(2, "little")
    res = run([r[0], bytes(b)])
    assert res["count"] == 0
    only_rejection(res, "NotPrimaryFrPair", 2)


def test_r1_orientation():  # :3171-3262
    fwd = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M", rev1=False, rev2=True)))
    rev = one(run(fr_pair("read1", 11, 1, 35, "30M", "30M", rev1=True, rev2=False)))
    assert fwd["seq"] == REF[0:40]
    assert rev["seq"] == revcomp(fwd["seq"]) and rev["quals"] == fwd["quals"][::-1]
    assert fwd["flag"] == 4 and rev["flag"] == 4


def disagreement_fixture(n):  # :3271-3320
    r = fr_pair("read1", 1, 11, 35, "30M", "30M")
    p = bamutil.parse(r[1])
    seq = list(p["seq"])
    for i in range(n):
        seq[i] = "C" if REF[10 + i] == "A" else "A"
    tags = [("MI", "Z", "hi"), ("RX", "Z", "ACC-TGA")]
    r2 = bamutil.make_record("read1", "".join(seq), p["quals"], flag=p["flag"], pos=p["pos"], cigar="30M", mate_ref=0, mate_pos=p["mate_pos"],
                             tlen=p["tlen"], tags=tags)
    return [r[0], r2]


def test_high_disagreement():  # :3323-3370, 3538-3580
    res = run(disagreement_fixture(6), codec_max_duplex_disagreements=100)
    assert res["count"] == 1
    # equal qualities: each disagreement becomes an N at MIN_PHRED; the rest of the duplex stretch agrees
    p = one(res)
    assert p["seq"][10:16] == "NNNNNN" and p["quals"][10:16] == [2] * 6 and p["seq"][16:30] == REF[16:30]
    assert res["stats"][25] == 20 and res["stats"][26] == 6 and res["stats"][24] == 40 and res["stats"][27] == 0
    res = run(disagreement_fixture(6), codec_max_duplex_disagreements=5, codec_max_duplex_disagreement_rate=0.05, track_rejects=1)
    assert res["count"] == 0 and res["data"] == b""
    only_rejection(res, "HighDuplexDisagreement", 2)
    assert res["stats"][27] == 1 and res["stats"][24] == 0 and res["stats"][25] == 0 and res["stats"][26] == 0
    assert res["n_rejects"] == 2
    # rate threshold alone (6/20 = 0.3 > 0.25)
    res = run(disagreement_fixture(6), codec_max_duplex_disagreement_rate=0.25)
    assert res["count"] == 0 and res["stats"][27] == 1
    res = run(disagreement_fixture(6), codec_max_duplex_disagreement_rate=0.3)
    assert res["count"] == 1


def test_soft_clipping():  # :4433-4500
    p = one(run(fr_pair("read1", 1, 11, 35, "5S25M", "25M5S")))
    assert p["seq"] == PLACEHOLDER * 5 + REF[0:35] + PLACEHOLDER * 5


def test_both_soft_clipped_same_end():  # :4504-4583
    p = one(run(fr_pair("read1", 1, 1, 35, "5S25M", "5S25M")))
    assert p["seq"] == PLACEHOLDER * 5 + REF[0:25]
    assert p["quals"] == [66] * 30


def test_chimeric_pair():  # :4587-4655
    r = fr_pair("read1", 1, 11, 35, "30M", "30M")
    a, b = bytearray(r[0]), bytearray(r[1])
    a[0:4] = (2).to_bytes(4, "little")
    b[20:24] = (2).to_bytes(4, "little")
    res = run([bytes(a), bytes(b)])
    assert res["count"] == 0
    only_rejection(res, "NotPrimaryFrPair", 2)


def test_r1_end_in_indel():  # :4659-4703
    res = run(fr_pair("read1", 1, 11, 35, "30M", "19M2D11M"))
    assert res["count"] == 0
    only_rejection(res, "IndelErrorBetweenStrands", 2)


def test_mask_end_qualities():  # :4707-4781
    base = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M")))
    p = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_outer_bases_length=7, codec_outer_bases_qual=5, codec_has_outer_bases_qual=1))
    assert p["quals"][:7] == [5] * 7 and p["quals"][-7:] == [5] * 7 and p["quals"][7:-7] == base["quals"][7:-7]
    assert p["seq"] == base["seq"]


def test_mask_single_stranded_regions():  # :4785-4866
    base = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M")))
    p = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_single_strand_qual=4, codec_has_single_strand_qual=1))
    assert p["quals"][:10] == [4] * 10 and p["quals"][30:] == [4] * 10 and p["quals"][10:30] == base["quals"][10:30]
    # outer first, then single-strand wins where both apply (:4908-4932)
    p = one(run(fr_pair("read1", 1, 11, 35, "30M", "30M"), codec_single_strand_qual=4, codec_has_single_strand_qual=1,
                codec_outer_bases_length=12, codec_outer_bases_qual=5, codec_has_outer_bases_qual=1))
    assert p["quals"][:10] == [4] * 10 and p["quals"][10:12] == [5] * 2 and p["quals"][28:30] == [5] * 2 and p["quals"][30:] == [4] * 10


def codec_fixture(n_pairs, bases, variant=None, variant_count=0):  # build_codec_fixture :6867-6902
    recs = []
    for i in range(n_pairs):
        b1 = variant if (variant and i < variant_count) else bases
        recs += list(bamutil.pair2(f"p{i:07d}", b1, 40, bases, 40, "mol1", 100, 100, rev1=False, rev2=True, rx="ACC"))
    return recs


def check_tags(p, cd, cm, ce, ad, am, ae, bd, bm, be, ad_b, bd_b, ae_b, be_b):
    t = {k: v[1] for k, v in p["tags"].items()}
    assert (t["cD"], t["cM"], t["aD"], t["aM"], t["bD"], t["bM"]) == (cd, cm, ad, am, bd, bm)
    for got, want in ((t["cE"], ce), (t["aE"], ae), (t["bE"], be)):
        assert abs(got - want) < 1e-6, (got, want)
    assert t["ad"] == ad_b and t["bd"] == bd_b and t["ae"] == ae_b and t["be"] == be_b


def test_depth_saturation_fgbio():  # :6947-6956 (captured from fgbio 4.0.1)
    p = one(run(codec_fixture(33000, "ACGT"), produce_per_base_tags=1, read_group_id=b"A"))
    check_tags(p, 65534, 65534, 0.0, 32767, 32767, 0.0, 32767, 32767, 0.0, [32767] * 4, [32767] * 4, [0] * 4, [0] * 4)
    assert p["tag_order"] == ["RG", "MI", "cD", "cM", "cE", "aD", "aM", "aE", "bD", "bM", "bE", "ad", "bd", "ae", "be", "ac", "bc", "aq", "bq", "RX"]


def test_error_saturation_fgbio():  # :6961-6970
    p = one(run(codec_fixture(73000, "ACGTACGT", "CCGTACGT", 33000), produce_per_base_tags=1, read_group_id=b"A"))
    check_tags(p, 65534, 65534, 0.0625, 32767, 32767, 0.125, 32767, 32767, 0.0, [32767] * 8, [32767] * 8, [32767] + [0] * 7, [0] * 8)


def test_fragment_and_minority_rejects_tracked():  # :3630-3777
    fam = []
    for i in range(3):
        fam += fr_pair(f"t{i}", 1, 11, 35, "30M", "30M")
    fam += fr_pair("odd", 1, 11, 35, "10M2D20M", "30M")
    frag = bamutil.frag("solo", REF[:30], 35, "hi")
    res = run(fam + [frag], track_rejects=1)
    assert res["count"] == 1
    st = res["stats"]
    assert st[REJ["FragmentRead"]] == 1 and st[REJ["MinorityAlignment"]] == 1 and st[2] == 2 and st[0] == 9
    rej = split_records(res["rejects"])
    assert len(rej) == 2 and rej[0] == fam[6] and rej[1] == frag        # input order: odd/R1 then the fragment


def test_downsampling_per_strand():  # :6074-6290
    fam = []
    for i in range(6):
        fam += fr_pair(f"t{i}", 1, 11, 35, "30M", "30M")
    res = run(fam, codec_max_reads_per_strand=2, produce_per_base_tags=1)
    p = one(res)
    assert res["stats"][REJ["Downsampled"]] == 8 and p["tags"]["aD"][1] == 2 and p["tags"]["bD"][1] == 2
    res = run(fam, codec_max_reads_per_strand=0)
    assert res["count"] == 0
    only_rejection(res, "InsufficientReads", 12)


def test_counter_names_without_mi():  # write_read_name :1568-1579
    def no_mi(recs):
        out = []
        for r in recs:
            p = bamutil.parse(r)
            out.append(bamutil.make_record(p["name"], p["seq"], p["quals"], flag=p["flag"], pos=p["pos"], cigar="30M", mate_ref=0,
                                           mate_pos=p["mate_pos"], tlen=p["tlen"], tags=[("RX", "Z", "ACC")]))
        return out
    g1, g2 = no_mi(fr_pair("a", 1, 11, 35, "30M", "30M")), no_mi(fr_pair("b", 5, 15, 35, "30M", "30M"))
    res = run(None, groups=[g1, g2])
    names = [bamutil.parse(r)["name"] for r in split_records(res["data"])]
    assert names == ["codec:1", "codec:2"]
    assert "MI" not in bamutil.parse(split_records(res["data"])[0])["tags"]


def test_nocall_and_low_quality_inputs_are_not_masked():
    """to_source_read_for_codec_raw (:503-570) does no quality masking: a Q5 base still votes."""
    r = fr_pair("read1", 1, 1, 35, "30M", "30M")
    p = bamutil.parse(r[0])
    q = list(p["quals"])
    q[3] = 5
    r1 = bamutil.make_record("read1", p["seq"], q, flag=p["flag"], pos=0, cigar="30M", mate_ref=0, mate_pos=0, tlen=p["tlen"],
                             tags=[("MI", "Z", "hi")])
    out = one(run([r1, r[1]]))
    assert out["seq"] == REF[:30]
    assert out["quals"][3] < out["quals"][4]


# ---- overlap geometry with indels and dovetails (codec_caller.rs:3911-4330): EXACT consensus bases --------------------------
# These cases assert the emitted bases against the reference's own test sequence (`REF_BASES`, codec_caller.rs:2392 — fixture data
# of the reference's tests, copied as a golden vector), built by the same `create_fr_pair` rules as above (placeholder 'A').
REF_BASES = (
    "TGGGTGTTGTTTGGGTTCCTGCTTAAAGAGCTACTGTTCTTCACAGAAACTTCCAACTCACCCAGACTGAGATTTGTACTGAGACTACGATCCACATGTTCAATATCTGATATCTGATGG"
    "GAAATAGGCTTTACTGAATTATCCATTTGGGCTGTAATTAATTTCAGTGATGAGCGGGAGATGTTGTTAGTTGTGCTCAGTAACTTTTTGATAGTAGCGGGAGTAGGAGTAAATCTTGTA"
    "CTAATTAGTGAATATTCTGTTGATGGTGGCTGAAAATTTATAGCTACACAACCAAAAAAATAAAAAACGTTAGTCAATAGCATTTATAAATAGTCTTCTCTACCTGAAATATTTTACATT"
    "AAGTAATTCATTCCTTCATTTAGTATCTACACATGTCTAACATTGTAGTAGGAGCTGTGTACTAACAAGAAATCATGACACTGTTTCTGCCTTCAAGGAGCTTATAATCTTTTGGGGTAC"
    "ACAAGATAACCCAGAATGTTAAATAGTATAAAAGTCAAAGTACAATAATTTATTTCATTAAGATTTTGAAATGGCTAACAAACACCTGTTGATCACCTCATACACATGAGCCTCAAAACA"
    "AAGGAAAGCACAGCCCCTATGCCTGAGCAATTTAGAATATTGTCAAGGATAGAGACATGTGAGCCATTCACTATGAAACAATCATTGAGAACTACTACAAGAGTGATAAATATAAAATGA"
    "AACCTACAGAAACACAGAAGAGTAAGTAATTTTCCCTATAAAGAAGACAGGAACTAAATGTATAAGCAAAAATTGGGAAATTATATAAATGCTATTTTATATGAGAGGCAAAGAACCACA"
    "GGTCTAATAATTTTACAAATGTGATAAAATCAGATTTTATGTCCCCATCTTTCTTGACTGCTCAGCTAGAAATTAAAACATTTTTACACATCTTTTTGGCGGGGGCGGGGGGGATCATTA"
    "TTTATTTCACCTGCCAAAATACTTCATTTCCTTATTGCACTTTTTTACTTCTTTGGTATGGAAAAATCTAACGGGTTTTAGAGTATGAACACATTTTAAGCAGTGATTAGATACGTTTTT"
    "CTTGTTATGCTTTCTATTGCAAATTTAGGATTTGATTTTGCACTGTCTTCATGCAAAGCTCTTCTCAAAGGTCTTAAAATATAAAAAACACTTAATGCTTCTCAAAGCATTAAGATTTTA"
    "TGTAAATCAAACCAAAACCAGAAAAAGACAGAAGAAAATGAACCAAAAACAACAAAAATAATCCTTAACATAGTTGGCAACAAGTGCAATGAAAGATTTTT"
)


def _codec_template(pos_start, pos_cigar, neg_start, neg_cigar, name="t0"):  # codec_template :3983-4001
    return fr_pair(name, pos_start, neg_start, 35, pos_cigar, neg_cigar, mi="mi", rx="ACC-TGA", ref=REF_BASES)


def _call_family(recs, min_duplex_length=1):  # call_codec_family :4003-4024
    return run(recs, codec_min_reads_per_strand=1, codec_min_duplex_length=min_duplex_length, min_consensus_base_quality=0)


def _rejected_whole_family(res, n, reason):  # assert_rejected_whole_family :4218-4242
    assert res["count"] == 0 and res["stats"][1] == 0 and res["stats"][0] == n
    only_rejection(res, reason, n)


def test_indel_at_overlap_boundary_still_calls_a_consensus():  # :3960-3981 (fgumi#752): 2S124M1D3M at 200 against 3S125M at 200, two templates
    recs = []
    for i in range(2):
        recs += fr_pair(f"t{i}", 200, 200, 35, "2S124M1D3M", "3S125M", mi="mi", rx="ACC-TGA", ref=REF_BASES)
    res = _call_family(recs)
    c = one(res)
    assert len(c["seq"]) == 127
    assert c["seq"] == "AAGTAACTTTTTGATAGTAGCGGGAGTAGGAGTAAATCTTGTACTAATTAGTGAATATTCTGTTGATGGTGGCTGAAAATTTATAGCTACACAACCAAAAAAATAAAAAACGTTAGTCAATAGCATT"
    assert sum(res["stats"][3:24]) == 0 and res["stats"][2] == 0 and res["stats"][0] == len(recs)


_DOVETAIL = (201, "2S126M", 200, "3S127M")       # dovetailed_start_template :4030-4037


def test_dovetailed_starts_without_an_indel_call_a_consensus():  # :4051-4078 (fgumi#761)
    res = _call_family(_codec_template(*_DOVETAIL))
    c = one(res)
    assert sum(res["stats"][3:24]) == 0 and len(c["seq"]) == 128
    assert c["seq"] == "ANTAACTTTTTGATAGTAGCGGGAGTAGGAGTAAATCTTGTACTAATTAGTGAATATTCTGTTGATGGTGGCTGAAAATTTATAGCTACACAACCAAAAAAATAAAAAACGTTAGTCAATAGCATTTA"


def test_min_duplex_length_is_measured_over_the_shared_region():  # :4089-4106: 126 shared positions pass 126, fail 127
    assert _call_family(_codec_template(*_DOVETAIL), 126)["count"] == 1
    _rejected_whole_family(_call_family(_codec_template(*_DOVETAIL), 127), 2, "InsufficientOverlap")


def test_terminal_indel_outside_the_shared_region_calls_a_consensus():  # :4126-4158
    res = _call_family(_codec_template(201, "2S124M1D3M", 200, "3S124M2S"))
    c = one(res)
    assert sum(res["stats"][3:24]) == 0 and len(c["seq"]) == 127
    assert c["seq"] == "ANTAACTTTTTGATAGTAGCGGGAGTAGGAGTAAATCTTGTACTAATTAGTGAATATTCTGTTGATGGTGGCTGAAAATTTATAGCTACACAACCAAAAAAATAAAAAACGTTAGTCAATAGCATNA"


def test_indel_inside_the_shared_region_is_still_rejected():  # :4171-4188
    _rejected_whole_family(_call_family(_codec_template(201, "2S60M1D67M", 200, "3S124M2S")), 2, "IndelErrorBetweenStrands")


def _window_end_past_r2_fixture():  # :4254-4284: tA R1 100M at 100 / R2 40M at 160; tB R1 30M at 120 / R2 50M at 120
    return (fr_pair("tA", 100, 160, 35, "100M", "40M", mi="mi", rx="ACC-TGA", ref=REF_BASES) +
            fr_pair("tB", 120, 120, 35, "30M", "50M", mi="mi", rx="ACC-TGA", ref=REF_BASES))


def test_r1_running_past_r2_is_not_an_indel_error():  # :4306-4322, and test_clip_overlap_failed_counted_and_labeled :3806-3841
    res = _call_family(_window_end_past_r2_fixture())
    assert res["stats"][REJ["IndelErrorBetweenStrands"]] == 0
    _rejected_whole_family(res, 4, "ClipOverlapFailed")


def test_clip_overlap_failed_attribution_matches_fgbio():  # :3875-3909: tA 50M / 50M at 200, tB R1 40M / R2 102M at 199
    recs = (fr_pair("tA", 200, 200, 35, "50M", "50M", mi="mi", rx="ACC-TGA", ref=REF_BASES) +
            fr_pair("tB", 199, 199, 35, "40M", "102M", mi="mi", rx="ACC-TGA", ref=REF_BASES))
    res = _call_family(recs)
    assert res["stats"][REJ["IndelErrorBetweenStrands"]] == 0
    _rejected_whole_family(res, 4, "ClipOverlapFailed")


def test_consensus_bases_emitted_totals_emitted_consensus_bases():  # :3393-3415 (permissive thresholds, one built-in disagreement)
    res = run(disagreement_fixture(1), codec_min_reads_per_strand=1, codec_min_duplex_length=1)
    p = one(res)
    assert len(p["seq"]) > 0 and res["stats"][24] == len(p["seq"])


@pytest.mark.parametrize("max_dis,max_rate", [(0, 1.0), (0xFFFFFFFF, 0.0)])
def test_rejected_molecules_contribute_no_emitted_bases(max_dis, max_rate):  # :3418-3457, and the typed errors :3460-3512 (1 disagreement over 20 duplex positions)
    res = run(disagreement_fixture(1), codec_min_reads_per_strand=1, codec_min_duplex_length=1, codec_max_duplex_disagreements=max_dis,
              codec_max_duplex_disagreement_rate=max_rate)
    assert res["count"] == 0 and res["stats"][24] == 0 and res["stats"][25] == 0 and res["stats"][26] == 0 and res["stats"][27] == 1
    # the thresholds are exactly where the reference puts them: one disagreement / a rate of 1/20 passes
    ok = run(disagreement_fixture(1), codec_min_reads_per_strand=1, codec_min_duplex_length=1, codec_max_duplex_disagreements=1, codec_max_duplex_disagreement_rate=1.0 / 20.0)
    assert ok["count"] == 1 and ok["stats"][26] == 1 and ok["stats"][25] == 20
