"""Oracle pins: the caller-level unit tests of the reference's `VanillaUmiConsensusCaller`
(`crates/fgumi-consensus/src/vanilla_caller.rs`, `#[cfg(test)] mod tests`), transcribed case by case — same reads, same
options (the LIBRARY defaults of `VanillaUmiConsensusOptions::default()`, `:330-347`: min_reads 2, min consensus base quality 40,
pre / post UMI 45 / 40, min input base quality 10, per-base tags on; no overlapping-consensus pre-correction, which is a step of the
commands, not of the caller), same assertions.  Each test names the reference test it restates.  The GPU parity tests compare the
HIP path with this oracle byte for byte; these cases tie the oracle to what the reference's own tests expect."""
import math

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import GroupedReads
from fgumi_amd.caller import split_records

F_PAIRED, F_UNMAPPED, F_REVERSE, F_MATE_REVERSE, F_FIRST, F_LAST = 0x1, 0x4, 0x10, 0x20, 0x40, 0x80
ST_TOTAL, ST_CONSENSUS, ST_FILTERED, ST_REASON0 = 0, 1, 2, 3


def ref_defaults(**kw):
    """`VanillaUmiConsensusOptions::default()` (vanilla_caller.rs:330-347) behind the oracle's option block."""
    base = dict(min_reads=2, min_consensus_base_quality=40, overlapping_consensus=0, cell_tag=b"\0\0", read_name_prefix=b"consensus")
    base.update(kw)
    o = fgx_opts.defaults(**base)
    o._kw = dict(base)              # (remembered for replay_cases)
    return o


def test_read(name, bases, quals, umi):
    """`create_consensus_test_read` (vanilla_caller.rs:3054-3066): unpaired, ref 0, pos 99, one M op, MI tag."""
    return bamutil.make_record(name, bases, list(quals), flag=0, ref_id=0, pos=99, tags=[("MI", "Z", umi)])


test_read.__test__ = False


_REPLAY = []   # (option keywords, reads) of every run below that succeeded: see replay_cases()


def call_groups(opts, groups, replay=True):
    g = GroupedReads.from_groups(groups)
    res = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first)
    if replay and hasattr(opts, "_kw"):
        _REPLAY.append((dict(opts._kw), [list(x) for x in groups]))
    return res, [bamutil.parse(r) for r in split_records(res["data"])]


def call(opts, reads):
    return call_groups(opts, [reads])


def expected_consensus_quality(q, n):
    """`expected_consensus_quality` (vanilla_caller.rs:4550-4562), fgbio's expectedConsensusQuality."""
    p = 10.0 ** (q / -10.0)
    ok, err = 1.0 - p, p / 3.0
    num = ok ** n
    den = num + (err ** 2) * 3.0
    return int(math.floor(-10.0 * math.log10(1.0 - num / den)))


def test_consensus_from_two_reads():  # vanilla_caller.rs:3070-3098
    res, recs = call(ref_defaults(min_reads=1, min_consensus_base_quality=0), [test_read("r1", "GATTACA", [10] * 7, "UMI1"), test_read("r2", "GATTACA", [10] * 7, "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"] == "GATTACA"
    assert all(q > 10 for q in recs[0]["quals"])


def test_consensus_with_one_disagreement():  # vanilla_caller.rs:3102-3139
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=93)
    res, recs = call(o, [test_read("r1", "GATTACA", [10] * 7, "UMI1"), test_read("r2", "GATTACA", [10] * 7, "UMI1"), test_read("r3", "GATTTCA", [10] * 7, "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"] == "GATTACA"
    assert recs[0]["quals"][4] < recs[0]["quals"][0]


def test_shortened_consensus_different_lengths():  # vanilla_caller.rs:3143-3174
    res, recs = call(ref_defaults(min_reads=2, min_consensus_base_quality=0), [test_read("r1", "GATTACA", [10] * 7, "UMI1"), test_read("r2", "GATTAC", [10] * 6, "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"] == "GATTAC"


def test_consensus_truncates_when_below_minreads():  # vanilla_caller.rs:3178-3209
    res, recs = call(ref_defaults(min_reads=2, min_consensus_base_quality=10), [test_read("r1", "A" * 10, [30] * 10, "UMI1"), test_read("r2", "A" * 20, [30] * 20, "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"] == "A" * 10


def test_mask_low_quality_consensus_bases():  # vanilla_caller.rs:3213-3241
    o = ref_defaults(min_reads=1, min_consensus_base_quality=10, min_input_base_quality=2, error_rate_pre_umi=93)
    res, recs = call(o, [test_read("r1", "GATTACA", [10, 10, 10, 10, 10, 10, 5], "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"][6] == "N" and recs[0]["quals"][6] == 2


def test_pre_umi_error_rate_zero_probability():  # vanilla_caller.rs:3245-3276
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=93, error_rate_post_umi=93)
    res, recs = call(o, [test_read("r1", "GATTACA", [10] * 7, "UMI1")])
    assert res["count"] == 1 and all(abs(q - 10) <= 1 for q in recs[0]["quals"])


@pytest.mark.parametrize("pre,post", [(10, 93), (93, 10)])
def test_error_rates_with_positive_probability_lower_the_quality(pre, post):  # vanilla_caller.rs:3280-3310, 3314-3343
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=pre, error_rate_post_umi=post)
    res, recs = call(o, [test_read("r1", "GATTACA", [10] * 7, "UMI1")])
    assert res["count"] == 1 and all(q < 10 for q in recs[0]["quals"])


def test_min_input_base_quality():  # vanilla_caller.rs:3347-3381
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, min_input_base_quality=30, error_rate_pre_umi=93, error_rate_post_umi=93)
    res, recs = call(o, [test_read("r1", "GATTACA", [20] * 7, "UMI1"), test_read("r2", "GATTACA", [30] * 7, "UMI1")])
    assert res["count"] == 1 and recs[0]["seq"] == "GATTACA" and all(q <= 35 for q in recs[0]["quals"])


def test_per_read_and_per_base_tags():  # vanilla_caller.rs:3385-3450
    o = ref_defaults(min_reads=1, min_input_base_quality=2, produce_per_base_tags=1)
    reads = [test_read(f"r{i}", "A" * 10, [30] * 10, "UMI1") for i in (1, 2, 3)] + [test_read("r4", "AAAAACAAAA", [30] * 10, "UMI1")]
    res, recs = call(o, reads)
    c = recs[0]
    assert res["count"] == 1 and c["seq"] == "A" * 10
    assert c["tags"]["cD"][1] == 4 and c["tags"]["cM"][1] == 4
    assert abs(c["tags"]["cE"][1] - 0.025) < 0.01
    assert c["tags"]["cd"][1] == [4] * 10
    assert c["tags"]["ce"][1] == [0, 0, 0, 0, 0, 1, 0, 0, 0, 0]


def test_consensus_depth_saturates_at_i16_max_like_fgbio():  # vanilla_caller.rs:3456-3511
    depth, L = 40000, 4
    reads = [test_read(f"r{i}", ("C" if i == 0 else "A") + "A" * (L - 1), [30] * L, "UMI1") for i in range(depth)]
    res, recs = call(ref_defaults(produce_per_base_tags=1), reads)
    c = recs[0]
    assert res["count"] == 1 and c["seq"] == "A" * L
    assert c["tags"]["cD"][1] == 32767 and c["tags"]["cM"][1] == 32767
    assert c["tags"]["cd"][1] == [32767] * L
    expected_ce = 1.0 / 131068.0
    assert abs(c["tags"]["cE"][1] - expected_ce) < expected_ce * 0.01


def test_errors_relative_to_consensus():  # vanilla_caller.rs:3515-3552
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, produce_per_base_tags=1)
    res, recs = call(o, [test_read("r1", "GATNACAG", [20] * 8, "UMI1"), test_read("r2", "GATGACAG", [20] * 8, "UMI1"),
                         test_read("r3", "GATGACAG", [20] * 8, "UMI1"), test_read("r4", "GATTACAG", [20] * 8, "UMI1")])
    c = recs[0]
    assert res["count"] == 1 and c["seq"][3] == "G"
    assert len(c["tags"]["cd"][1]) == 8 and c["tags"]["cd"][1][3] == 3 and c["tags"]["ce"][1][3] == 1


def test_consensus_ns_when_all_inputs_masked():  # vanilla_caller.rs:3556-3599
    o = ref_defaults(min_reads=1, min_input_base_quality=30, min_consensus_base_quality=40, error_rate_pre_umi=93, error_rate_post_umi=93, produce_per_base_tags=1)
    reads = [test_read(f"r{i}", "GATTACA", [20] * 7, "UMI1") for i in (1, 2, 3)] + [test_read("r4", "CTAATGT", [30] * 7, "UMI1")]
    res, recs = call(o, reads)
    c = recs[0]
    assert res["count"] == 1 and c["seq"] == "N" * 7 and c["quals"] == [2] * 7 and c["tags"]["cd"][1] == [1] * 7


def test_no_per_base_tags_when_disabled():  # vanilla_caller.rs:3603-3646
    res, recs = call(ref_defaults(min_reads=1, produce_per_base_tags=0), [test_read("r1", "A" * 10, [30] * 10, "UMI1"), test_read("r2", "A" * 10, [30] * 10, "UMI1")])
    t = recs[0]["tags"]
    assert res["count"] == 1 and "cD" in t and "cM" in t and "cE" in t and "cd" not in t and "ce" not in t


def test_consensus_from_two_reads_exact_quality():  # vanilla_caller.rs:4567-4605
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=93, error_rate_post_umi=93)
    res, recs = call(o, [test_read("r1", "GATTACA", [10] * 7, "UMI1"), test_read("r2", "GATTACA", [10] * 7, "UMI1")])
    want = expected_consensus_quality(10, 2)
    assert res["count"] == 1 and recs[0]["seq"] == "GATTACA" and all(abs(q - want) <= 1 for q in recs[0]["quals"])


def test_consensus_from_three_reads_with_disagreement_errors():  # vanilla_caller.rs:4610-4655
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=93, error_rate_post_umi=93, produce_per_base_tags=1)
    res, recs = call(o, [test_read("r1", "GATTACA", [10] * 7, "UMI1"), test_read("r2", "GATTACA", [10] * 7, "UMI1"), test_read("r3", "GATTTCA", [10] * 7, "UMI1")])
    c = recs[0]
    assert res["count"] == 1 and c["seq"] == "GATTACA" and c["tags"]["ce"][1] == [0, 0, 0, 0, 1, 0, 0] and c["quals"][4] < c["quals"][0]


def frag_with_umi(name, umi, bases, quals):
    """`create_fragment_read_with_umi` (vanilla_caller.rs:4793-4809): ref 0, pos 0."""
    return bamutil.make_record(name, bases, list(quals), flag=0, ref_id=0, pos=0, tags=[("MI", "Z", umi)])


def paired_with_umi(name, umi, bases, quals, start1, start2):
    """`create_paired_reads_with_umi` (vanilla_caller.rs:4815-4853): both forward, mate fields set, no MC tag."""
    r1 = bamutil.make_record(name, bases, list(quals), flag=F_PAIRED | F_FIRST, ref_id=0, pos=start1 - 1, mate_ref=0, mate_pos=start2 - 1, tags=[("MI", "Z", umi)])
    r2 = bamutil.make_record(name, bases, list(quals), flag=F_PAIRED | F_LAST, ref_id=0, pos=start2 - 1, mate_ref=0, mate_pos=start1 - 1, tags=[("MI", "Z", umi)])
    return r1, r2


def test_two_consensus_for_two_umi_groups():  # vanilla_caller.rs:4857-4892
    o = ref_defaults(min_reads=1, error_rate_pre_umi=93, error_rate_post_umi=93)
    for umi, names in (("GATTACA", ("READ1", "READ2")), ("ACATTAG", ("READ3", "READ4"))):
        res, _ = call(o, [frag_with_umi(n, umi, "A" * 50, [60] * 50) for n in names])
        assert res["count"] == 1


def test_two_consensus_for_read_pair():  # vanilla_caller.rs:4896-4926
    o = ref_defaults(min_reads=1, error_rate_pre_umi=93, error_rate_post_umi=93, read_name_prefix=b"c")
    res, recs = call(o, list(paired_with_umi("READ1", "GATTACA", "A" * 100, [60] * 100, 1, 1000)))
    assert res["count"] == 2 and all(r["flag"] & F_PAIRED for r in recs)
    assert recs[0]["flag"] & F_FIRST and recs[1]["flag"] & F_LAST and recs[0]["name"] == recs[1]["name"]


def test_four_consensus_for_two_pairs_different_groups():  # vanilla_caller.rs:4930-4985
    names = []
    for prefix, read, umi in ((b"c1", "READ1", "GATTACA"), (b"c2", "READ2", "ACATTAG")):
        o = ref_defaults(min_reads=1, error_rate_pre_umi=93, error_rate_post_umi=93, read_name_prefix=prefix)
        res, recs = call(o, list(paired_with_umi(read, umi, "A" * 100, [60] * 100, 1, 1000)))
        assert res["count"] == 2 and all(r["flag"] & F_PAIRED for r in recs)
        assert recs[0]["flag"] & F_FIRST and recs[1]["flag"] & F_LAST and recs[0]["name"] == recs[1]["name"]
        names.append(recs[0]["name"])
    assert names[0] != names[1]


def test_quality_trim_and_mask_combined():  # vanilla_caller.rs:4989-5018 (create_source_read: trim to 3 bases "AGC"; seen through the caller)
    rec = bamutil.make_record("test", "AGCACGACGT", [30, 30, 30, 2, 5, 2, 3, 20, 2, 6], flag=0, ref_id=0, pos=0, tags=[("MI", "Z", "UMI1")])
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, min_input_base_quality=15, trim=1, error_rate_pre_umi=93, error_rate_post_umi=93)
    res, recs = call(o, [rec])
    assert res["count"] == 1 and recs[0]["seq"] == "AGC"


def test_absent_base_qualities_abort_consensus():  # vanilla_caller.rs:5022-5063 (both: the source read errors, the run aborts)
    rec = bamutil.make_record("test", "A" * 10, None, flag=F_PAIRED | F_FIRST, ref_id=0, pos=0, tags=[("MI", "Z", "UMI1")])
    g = GroupedReads.from_groups([[rec]])
    with pytest.raises(RuntimeError):
        orc.process(ref_defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)


def test_zero_length_record_is_not_treated_as_absent_qualities():  # vanilla_caller.rs:5067-5090
    rec = bamutil.make_record("empty", "", [], flag=F_UNMAPPED, ref_id=-1, pos=-1, cigar="", tags=[("MI", "Z", "UMI1")])
    ok = test_read("r1", "ACGT", [30] * 4, "UMI1")
    res, _ = call(ref_defaults(min_reads=1, track_rejects=1, min_consensus_base_quality=2), [rec, ok])      # must not raise
    assert res["count"] == 1


def test_mate_cigar_handling():  # vanilla_caller.rs:5094-5140: a pair WITHOUT MC tags still gives two consensus reads
    r1 = bamutil.make_record("READ1", "A" * 10, [30] * 10, flag=F_PAIRED | F_FIRST | F_MATE_REVERSE, ref_id=0, pos=0, mate_ref=0, mate_pos=99, tlen=109,
                             tags=[("MI", "Z", "GATTACA")])
    r2 = bamutil.make_record("READ1", "A" * 10, [30] * 10, flag=F_PAIRED | F_LAST | F_REVERSE, ref_id=0, pos=99, mate_ref=0, mate_pos=0, tlen=-109,
                             tags=[("MI", "Z", "GATTACA")])
    res, _ = call(ref_defaults(min_reads=1, min_input_base_quality=2, read_name_prefix=b"c"), [r1, r2])
    assert res["count"] == 2


def test_consensus_umi_on_filtered_reads_only():  # vanilla_caller.rs:5144-5209: RX from the reads that survive the alignment filter
    def rd(name, cigar, rx):
        return bamutil.make_record(name, "A" * 10, [30] * 10, flag=0, ref_id=0, pos=0, cigar=cigar, tags=[("MI", "Z", "AAA"), ("RX", "Z", rx)])
    reads = [rd("READ1", "10M", "TTT"), rd("READ2", "5M5D5M", "ATT"), rd("READ3", "10M", "TAT"), rd("READ4", "4M2I4M", "TTA")]
    res, recs = call(ref_defaults(min_reads=1, min_input_base_quality=2, read_name_prefix=b"c"), reads)
    assert res["count"] == 1 and recs[0]["tags"]["RX"][1] == "TNT"


@pytest.mark.parametrize("umi_len,expect_ok", [(244, True), (245, False), (4096, False)])
def test_over_long_umi_errors_instead_of_panicking(umi_len, expect_ok):  # vanilla_caller.rs:5341-5387
    umi = "A" * umi_len
    r1 = bamutil.make_record("read1", "ACGT", [30] * 4, flag=F_PAIRED | F_FIRST, ref_id=0, pos=0, tags=[("MI", "Z", umi)])
    r2 = bamutil.make_record("read1", "ACGT", [30] * 4, flag=F_PAIRED | F_LAST, ref_id=0, pos=0, tags=[("MI", "Z", umi)])
    g = GroupedReads.from_groups([[r1, r2]])
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0)
    if expect_ok:
        assert call(o, [r1, r2])[0]["count"] == 2
    else:
        with pytest.raises(RuntimeError, match="read name too long"):
            orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first)


def test_stats_no_double_counting():  # vanilla_caller.rs:5391-5440: two pair groups through ONE caller: 4 reads in, 4 consensus reads
    def rd(name, first, umi):
        return bamutil.make_record(name, "ACGT", [30] * 4, flag=F_PAIRED | (F_FIRST if first else F_LAST), ref_id=0, pos=0, tags=[("MI", "Z", umi)])
    res, _ = call_groups(ref_defaults(min_reads=1, read_name_prefix=b"test"), [[rd("read1", True, "1"), rd("read1", False, "1")], [rd("read2", True, "2"), rd("read2", False, "2")]])
    assert res["count"] == 4 and res["stats"][ST_TOTAL] == 4 and res["stats"][ST_CONSENSUS] == 4


def test_orphan_consensus_no_double_count_r1_succeeds_r2_fails():  # vanilla_caller.rs:5444-5599
    good, bad = [30] * 50, [2] * 50
    def rd(name, first, quals, cigar="50M"):
        return bamutil.make_record(name, "A" * 50, quals, flag=F_PAIRED | (F_FIRST if first else F_LAST), ref_id=0, pos=99 if first else 999, cigar=cigar,
                                   mate_ref=0, mate_pos=999 if first else 99, tags=[("MI", "Z", "UMI1")])
    reads = [rd("read_a", True, good), rd("read_b", True, good), rd("read_c", True, good, "25M25I"),
             rd("read_a", False, bad), rd("read_b", False, bad), rd("read_c", False, bad)]
    res, _ = call(ref_defaults(min_reads=2, read_name_prefix=b"test"), reads)
    st = res["stats"]
    assert res["count"] == 0 and st[ST_TOTAL] == 6 and st[ST_FILTERED] <= st[ST_TOTAL]
    names = _reason_names()
    assert st[ST_REASON0 + names["MinorityAlignment"]] == 1
    assert st[ST_REASON0 + names["ZeroLengthAfterTrimming"]] == 3
    assert st[ST_REASON0 + names["OrphanConsensus"]] == 2


def test_orphan_consensus_no_double_count_r1_fails_r2_succeeds():  # vanilla_caller.rs:5603-5760 (the mirror image)
    good, bad = [30] * 50, [2] * 50
    def rd(name, first, quals, cigar="50M"):
        return bamutil.make_record(name, "A" * 50, quals, flag=F_PAIRED | (F_FIRST if first else F_LAST), ref_id=0, pos=99 if first else 999, cigar=cigar,
                                   mate_ref=0, mate_pos=999 if first else 99, tags=[("MI", "Z", "UMI1")])
    reads = [rd("read_a", True, bad), rd("read_b", True, bad), rd("read_c", True, bad),
             rd("read_a", False, good), rd("read_b", False, good), rd("read_c", False, good, "25M25I")]
    res, _ = call(ref_defaults(min_reads=2, read_name_prefix=b"test"), reads)
    st = res["stats"]
    names = _reason_names()
    assert res["count"] == 0 and st[ST_TOTAL] == 6 and st[ST_FILTERED] <= st[ST_TOTAL]
    assert st[ST_REASON0 + names["MinorityAlignment"]] == 1
    assert st[ST_REASON0 + names["ZeroLengthAfterTrimming"]] == 3
    assert st[ST_REASON0 + names["OrphanConsensus"]] == 2


def _reason_names():
    """Index of each RejectionReason in the statistics block, read from the C header's enum (the order of caller.rs:401-446)."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "fgumi_amd.h")).read()
    body = hdr[hdr.index("FGX_REJ_FRAGMENT_READ"):hdr.index("FGX_N_REJECTION")]
    names = re.findall(r"FGX_REJ_([A-Z_]+)", body)
    return {"".join(w.capitalize() for w in n.split("_")): i for i, n in enumerate(names)}


def replay_cases():
    """(oracle option keywords, MI groups) of every transcribed case that is a valid run — the GPU parity suite sends the
    same inputs through the HIP path (tests/test_gpu_parity.py::test_reference_caller_unit_test_inputs).  Collected by running the
    cases above once."""
    if not _REPLAY:
        import inspect
        import itertools
        mod = globals()
        for name, fn in sorted(mod.items()):
            if not (name.startswith("test_") and callable(fn)) or getattr(fn, "__test__", True) is False:
                continue
            marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
            if not marks:
                argsets = [()]
            else:
                argsets = [a if isinstance(a, tuple) else (a,) for a in marks[0].args[1]]
            for a in argsets:
                try:
                    fn(*a)
                except Exception:      # the cases that expect an error raise inside pytest.raises and record nothing
                    pass
    seen, out = set(), []
    for kw, reads in _REPLAY:
        key = (tuple(sorted((k, v) for k, v in kw.items())), tuple(tuple(x) for x in reads))
        if key not in seen:
            seen.add(key)
            out.append((kw, reads))
    return out


# ---- the alignment filter (select_most_common_alignment_group, vanilla_caller.rs:48-120) ----------------------------------------
# The reference tests it on bare `SourceRead`s (`create_source_read_with_cigar`, :3713-3730: 'A' x query length at Q30); here the
# same CIGAR mixes go through the caller as fragment records, and what the filter kept shows as the consensus depth (cD) and
# the MinorityAlignment count.

def _cigar_family(cigars):
    reads = []
    for i, c in enumerate(cigars):
        qlen = sum(o >> 4 for o in bamutil.cigar_ops(c) if (o & 15) in (0, 1, 4, 7, 8))
        reads.append(bamutil.make_record(f"r{i:02d}", "A" * qlen, [30] * qlen, flag=0, ref_id=0, pos=99, cigar=c, tags=[("MI", "Z", "UMI1")]))
    return reads


def _kept(cigars):
    # (not handed to the GPU replay: added after the round's last GPU run; tests/test_gpu_indels.py covers the device filter)
    res, recs = call_groups(ref_defaults(min_reads=1, min_consensus_base_quality=0, track_rejects=1), [_cigar_family(cigars)], replay=False)
    names = _reason_names()
    rej = int(res["stats"][ST_REASON0 + names["MinorityAlignment"]])
    assert res["count"] == 1 and res["n_rejects"] == rej and int(res["stats"][ST_FILTERED]) == rej
    return recs[0]["tags"]["cD"][1], rej


def test_filter_all_reads_same_cigar():  # vanilla_caller.rs:3760-3777 (50M), :3781-3798 (10M5D10M5I20M5S), :3884-3897 (a single read)
    assert _kept(["50M"] * 10) == (10, 0)
    assert _kept(["10M5D10M5I20M5S"] * 10) == (10, 0)
    assert _kept(["50M"]) == (1, 0)


def test_filter_keeps_most_common_alignment():  # vanilla_caller.rs:3802-3838
    assert _kept(["25M1D25M"] * 3 + ["50M"] * 10 + ["25M2I23M"] * 3) == (10, 6)


def test_filter_compatible_with_deletion():  # vanilla_caller.rs:3842-3880: clips fold into M, 11 of 17 are compatible with 25M1D25M
    good = ["25M1D25M"] * 5 + ["5S20M1D25M"] * 2 + ["5S20M1D20M5H"] * 2 + ["25M1D20M5S"] * 2
    other = ["25M2D25M"] * 2 + ["25M1I24M"] * 2 + ["20M1D5M1D25M"] * 2
    assert _kept(good + other) == (11, 6)


def test_filter_compatible_with_2bp_deletion():  # vanilla_caller.rs:4262-4314: prefixes of the 25M2D... pattern stay, other deletions go
    assert _kept(["25M2D75M", "25M2D65M", "25M2D50M5S", "25M", "24M", "10M"] + ["30M", "25M1D25M", "25M4D25M"]) == (6, 3)


def test_filter_reads_added_to_multiple_cigar_groups():  # vanilla_caller.rs:4664-4699: the 40M read joins the 40M1I9M group, which then wins 4 : 3
    assert _kept(["50M"] * 2 + ["40M1I9M"] * 3 + ["40M"]) == (4, 2)


def test_filter_preserves_input_order():  # vanilla_caller.rs:4703-4740: prefix-compatible lengths are all kept
    assert _kept(["100M", "80M", "90M", "70M", "85M"]) == (5, 0)


# ---- create_source_read (vanilla_caller.rs:1080-1160) seen through a single-read family ------------------------------------------
# With min_reads 1, no error-rate adjustment and no consensus-quality floor, the consensus of ONE fragment is its source read: the
# reference's `to_source_read` unit tests (masking, trailing trims, strand) read off the emitted record.

def _source_read(seq, quals, flag=0, cigar=None, min_input_base_quality=20):
    rec = bamutil.make_record("test", seq, list(quals), flag=flag, ref_id=0, pos=0, cigar=cigar, tags=[("MI", "Z", "UMI1")])
    o = ref_defaults(min_reads=1, min_consensus_base_quality=0, min_input_base_quality=min_input_base_quality, error_rate_pre_umi=93, error_rate_post_umi=93, track_rejects=1)
    return call_groups(o, [[rec]], replay=False)


def test_to_source_read_masks_low_quality_bases():  # vanilla_caller.rs:3901-3929
    res, recs = _source_read("AAAAAAAAAA", [2, 30, 19, 21, 18, 20, 0, 30, 2, 30])
    assert res["count"] == 1 and recs[0]["seq"] == "NANANANANA" and recs[0]["tags"]["cd"][1] == [0, 1] * 5


def test_to_source_read_trims_trailing_low_quality_and_ns():  # vanilla_caller.rs:3957-3980, 3984-4007
    assert _source_read("AAAAAAAAAA", [30] * 6 + [2] * 4)[1][0]["seq"] == "AAAAAA"
    assert _source_read("AAAAAANNNN", [30] * 10)[1][0]["seq"] == "AAAAAA"


def test_to_source_read_trims_ns_on_negative_strand():  # vanilla_caller.rs:4011-4040: reverse-complemented first, then trimmed at ITS end
    assert _source_read("NNNNAAAAAA", [30] * 10, flag=F_REVERSE, cigar="4S1M1D5M")[1][0]["seq"] == "TTTTTT"


def test_to_source_read_returns_none_for_all_low_quality():  # vanilla_caller.rs:4044-4064
    res, recs = _source_read("NANANANANA", [30, 2] * 5)
    names = _reason_names()
    assert res["count"] == 0 and res["stats"][ST_REASON0 + names["ZeroLengthAfterTrimming"]] == 1 and res["n_rejects"] == 1
