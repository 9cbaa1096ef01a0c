"""Oracle pins for `fgumi filter`: the reference's own unit tests (src/lib/commands/filter.rs tests :1680-1900, :4428-4700;
crates/fgumi-consensus/src/filter.rs tests :2180-2400) replayed against oracle/oracle_filter.hpp, plus stream-level
behaviour (template mode, rejects, ordering) stated by src/lib/commands/filter.rs:653-731 and src/lib/template.rs:243-352."""
import struct

import numpy as np
import pytest

import bamutil
import orc

UNMAPPED = 0x4


def arr(tag, vals, ty="S"):
    fmt = {"S": "H", "s": "h", "C": "B", "c": "b", "i": "i", "I": "I"}[ty]
    return (tag, "raw", b"B" + ty.encode() + struct.pack("<I", len(vals)) + struct.pack(f"<{len(vals)}{fmt}", *vals))


def rec(seq, quals, cD=None, cE=None, cd=None, ce=None, flag=UNMAPPED, name="q", extra=(), pos=-1):
    tags = []
    if cD is not None:
        tags.append(("cD", "i", cD))
    if cE is not None:
        tags.append(("cE", "f", cE))
    if cd is not None:
        tags.append(arr("cd", cd))
    if ce is not None:
        tags.append(arr("ce", ce))
    return bamutil.make_record(name, seq, quals, flag=flag, ref_id=-1 if flag & UNMAPPED else 0, pos=pos, tags=tags + list(extra))


def duplex_rec(seq, quals, aD=10, bD=8, aE=0.01, bE=0.01, cD=18, cE=0.01, ad=None, bd=None, ae=None, be=None, ac=None, bc=None, flag=UNMAPPED, name="q", extra=()):
    L = len(seq)
    tags = [("cD", "i", cD), ("cE", "f", cE), ("aD", "i", aD), ("bD", "i", bD), ("aE", "f", aE), ("bE", "f", bE), ("aM", "i", aD), ("bM", "i", bD)]
    tags += [arr("ad", ad if ad is not None else [aD] * L), arr("bd", bd if bd is not None else [bD] * L), arr("ae", ae if ae is not None else [0] * L),
             arr("be", be if be is not None else [0] * L)]
    if ac is not None:
        tags.append(("ac", "Z", ac))
    if bc is not None:
        tags.append(("bc", "Z", bc))
    return bamutil.make_record(name, seq, quals, flag=flag, ref_id=-1, pos=-1, tags=tags + list(extra))


def stream(records):
    blob = bytearray()
    off, ln = [], []
    for r in records:
        blob += struct.pack("<I", len(r))
        off.append(len(blob))
        ln.append(len(r))
        blob += r
    return np.frombuffer(bytes(blob) + b"\0" * 8, dtype=np.uint8).copy(), np.array(off, dtype=np.uint64), np.array(ln, dtype=np.uint32)


def seq_quals(r):
    p = bamutil.parse(r)
    return p["seq"], list(p["quals"])


# ---- src/lib/commands/filter.rs:1680-1760 ---------------------------------------------------------------------------
def test_mask_bases_low_quality():
    n, out = orc.filter_mask_bases(rec("ACGT", [10, 30, 5, 30], cd=[10] * 4, ce=[0] * 4), (1, 1.0, 1.0), 20)
    assert seq_quals(out) == ("NCNT", [2, 30, 2, 30]) and n == 2


def test_mask_bases_low_depth():
    n, out = orc.filter_mask_bases(rec("ACGT", [30] * 4, cd=[1, 10, 4, 10], ce=[0] * 4), (5, 1.0, 1.0), 10)
    assert seq_quals(out)[0] == "NCNT" and n == 2


def test_mask_bases_high_error_count():
    n, out = orc.filter_mask_bases(rec("ACGT", [30] * 4, cd=[10] * 4, ce=[1, 3, 2, 0]), (1, 1.0, 0.2), 10)
    assert seq_quals(out)[0] == "ANGT" and n == 1       # 2/10 == threshold is NOT masked


# ---- crates/fgumi-consensus/src/filter.rs:2248-2284 (FILT-04) ---------------------------------------------------------
@pytest.mark.parametrize("with_cd,with_ce", [(False, False), (True, False), (False, True)])
def test_mask_bases_depth_mask_requires_both_per_base_tags(with_cd, with_ce):
    r = rec("A" * 10, [40] * 5 + [10] * 5, cd=[1] * 10 if with_cd else None, ce=[100] * 10 if with_ce else None)
    n, _ = orc.filter_mask_bases(r, (5, 0.05, 0.1), 30)
    assert n == 5


def test_mask_bases_already_n_requalified_but_not_counted():
    # filter.rs:800-806: an N that fails a mask gets quality 2 again but is not counted
    n, out = orc.filter_mask_bases(rec("ANGT", [30, 30, 30, 30], cd=[10, 1, 10, 1], ce=[0] * 4), (5, 1.0, 1.0))
    assert n == 1 and seq_quals(out) == ("ANGN", [30, 2, 30, 2])


def test_mask_bases_signed_and_short_arrays():
    # array_tag_element_u16 (tags.rs:590-610): negative s/c elements clamp to 0, i/I/f sub-types read as 0, short arrays read 0 past the end
    r = bamutil.make_record("q", "ACGTAC", [30] * 6, flag=UNMAPPED, ref_id=-1, pos=-1, tags=[arr("cd", [-3, 5, 5], "s"), arr("ce", [0, 0, 0], "C")])
    n, out = orc.filter_mask_bases(r, (1, 1.0, 1.0))
    assert seq_quals(out)[0] == "NCGNNN" and n == 4
    r = bamutil.make_record("q", "ACG", [30] * 3, flag=UNMAPPED, ref_id=-1, pos=-1, tags=[arr("cd", [9, 9, 9], "i"), arr("ce", [0, 0, 0], "S")])
    n, out = orc.filter_mask_bases(r, (1, 1.0, 1.0))
    assert seq_quals(out)[0] == "NNN"


# ---- filter_read (src/lib/commands/filter.rs:1762-1845; crates filter.rs:2180-2210) -------------------------------------
def test_filter_read():
    thr = (5, 0.1, 0.2)
    assert orc.filter_read(rec("ACGT", [30] * 4, cD=10, cE=0.05), thr) == 0
    assert orc.filter_read(rec("ACGT", [30] * 4, cD=3, cE=0.05), thr) == 1
    assert orc.filter_read(rec("ACGT", [30] * 4, cD=10, cE=0.3), thr) == 2
    assert orc.filter_read(rec("ACGT", [30] * 4), thr) == -1
    assert b"cD/cE" in orc.lib.orc_last_error()


@pytest.mark.parametrize("with_cd,with_ce,want", [(False, False, -1), (True, False, -1), (False, True, -1), (True, True, 0)])
def test_filter_read_requires_consensus_tags(with_cd, with_ce, want):
    r = rec("ACGT", [30] * 4, cD=20 if with_cd else None, cE=0.0 if with_ce else None)
    assert orc.filter_read(r, (1, 0.05, 0.1)) == want


@pytest.mark.parametrize("with_ad,with_bd,want", [(False, False, False), (True, False, False), (False, True, False), (True, True, True)])
def test_is_duplex_requires_both(with_ad, with_bd, want):
    tags = ([("aD", "i", 10)] if with_ad else []) + ([("bD", "i", 8)] if with_bd else [])
    assert orc.filter_is_duplex(rec("ACGT", [30] * 4, extra=tags)) == want


def test_filter_duplex_read_tiers():
    cc, ab, ba = (10, 0.05, 0.1), (6, 0.02, 0.1), (3, 0.05, 0.1)
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=7, bD=3, cD=10), cc, ab, ba) == 0
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=3, bD=7, cD=10), cc, ab, ba) == 0      # best/worst are per-metric, not per-strand
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=5, bD=5, cD=10), cc, ab, ba) == 1      # best depth 5 < AB 6
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=8, bD=2, cD=10), cc, ab, ba) == 1      # worst depth 2 < BA 3
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=8, bD=4, cD=9), cc, ab, ba) == 1       # CC first
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=8, bD=4, cD=12, aE=0.03, bE=0.04), cc, ab, ba) == 2   # best error 0.03 > 0.02
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=8, bD=4, cD=12, aE=0.01, bE=0.06), cc, ab, ba) == 2   # worst error 0.06 > 0.05
    assert orc.filter_duplex_read(duplex_rec("ACGT", [30] * 4, aD=8, bD=4, cD=12, aE=0.04, bE=0.01), cc, ab, ba) == 0


# ---- check_*_filters_raw no-call modes (src/lib/commands/filter.rs:4428-4600) and process_record_raw (:4608-4700) ------------
@pytest.mark.parametrize("seq,frac,want", [("AANNTTGGCC", 0.2, True), ("AANNTTGGCC", 0.19, False), ("AANNNTTGGC", 5.0, True), ("AANNNTTGGC", 3.0, True),
                                           ("AANNNTTGGC", 2.0, False)])
def test_no_call_modes(seq, frac, want):
    o = orc.filter_options(min_reads=5, max_read_error_rate=0.1, max_base_error_rate=0.1, max_no_call_fraction=frac)
    masked, ok, _ = orc.filter_process_record(o, rec(seq, [30] * 10, cD=10, cE=0.01))
    assert ok == want and masked == 0


@pytest.mark.parametrize("frac,want", [(5.0, True), (2.0, False)])
def test_duplex_no_call_count_mode(frac, want):
    o = orc.filter_options(min_reads=5, max_read_error_rate=0.1, max_base_error_rate=0.1, max_no_call_fraction=frac)
    _, ok, _ = orc.filter_process_record(o, duplex_rec("AANNNTTGGC", [30] * 10, cD=10))
    assert ok == want


def test_process_record_raw_no_reference():
    o = orc.filter_options(min_reads=1)
    masked, ok, _ = orc.filter_process_record(o, rec("ACGTACGT", [35] * 8, cD=10, cE=0.0, cd=[10] * 8, ce=[0] * 8, name="unmapped_read"))
    assert (masked, ok) == (0, True)
    with pytest.raises(RuntimeError, match="--ref is required"):
        orc.filter_process_record(o, rec("ACGTACGT", [35] * 8, cd=[10] * 8, ce=[0] * 8, flag=0, pos=99, name="mapped_read"))


def test_mean_quality_is_pre_mask_full_length():
    # filter.rs:2216-2235: (5*40 + 5*2)/10 = 21.0 over the full read, Ns included; evaluated before masking
    r = rec("AAAAANNNNN", [40] * 5 + [2] * 5, cD=5, cE=0.0)
    assert orc.filter_process_record(orc.filter_options(min_mean_base_quality=21.0, max_no_call_fraction=5.0), r)[1] is True
    assert orc.filter_process_record(orc.filter_options(min_mean_base_quality=21.01, max_no_call_fraction=5.0), r)[1] is False
    r = rec("AAAA", [40, 40, 10, 10], cD=5, cE=0.0)   # masking lowers two quals to 2, the mean stays 25
    assert orc.filter_process_record(orc.filter_options(min_base_quality=20, min_mean_base_quality=25.0, max_no_call_fraction=0.5), r)[:2] == (2, True)


def test_duplex_masking_and_strand_agreement():
    cc, ab, ba = (6, 1.0, 0.2), (4, 1.0, 0.2), (2, 1.0, 0.3)
    r = duplex_rec("ACGTACGT", [30] * 8, ad=[5, 5, 3, 5, 5, 5, 0, 5], bd=[3, 0, 3, 1, 3, 3, 3, 3], ae=[0, 0, 0, 0, 2, 0, 0, 0], be=[0, 0, 0, 0, 0, 1, 0, 0],
                   ac="ACGTACGA", bc="ACGTACGT")
    n, out = orc.filter_mask_duplex_bases(r, cc, ab, ba)
    # pos1: worst depth 0 < 2 and total 5 < 6; pos2: best 3 < 4; pos3: worst 1 < 2; pos4: best rate .4 > .2; pos5: worst rate 1/3 > .3; pos6: ab depth 0
    assert seq_quals(out)[0] == "ANNNNNNT" and n == 6
    n, out = orc.filter_mask_duplex_bases(r, (1, 1.0, 1.0), (1, 1.0, 1.0), (0, 1.0, 1.0), None, True)
    assert seq_quals(out)[0] == "ACGTACGN"      # only the last position has both strands present and ac != bc
    short = duplex_rec("ACGT", [30] * 4, ad=[5] * 4, bd=[5] * 4, ac="AC", bc="ACGT")
    assert seq_quals(orc.filter_mask_duplex_bases(short, (1, 1.0, 1.0), (1, 1.0, 1.0), (1, 1.0, 1.0), None, True)[1])[0] == "ACNN"   # missing ac → 'N' != base


def test_reverse_per_base_tags():
    tags = [arr("cd", [1, 2, 3, 4]), arr("ce", [0, 0, 1, 1], "C"), ("aq", "Z", "ABCD"), ("ac", "Z", "AACG"), ("bc", "Z", "ARGT"), arr("ad", [7, 8, 9, 10], "i")]
    r = rec("ACGT", [30] * 4, cD=4, cE=0.0, flag=UNMAPPED | 0x10, extra=tags)
    _, _, out = orc.filter_process_record(orc.filter_options(reverse_per_base_tags=True), r)
    t = {k: v[1] for k, v in bamutil.parse(out)["tags"].items()}
    assert list(t["cd"]) == [4, 3, 2, 1] and list(t["ce"]) == [1, 1, 0, 0] and t["aq"] == "DCBA" and t["ac"] == "CGTT" and t["bc"] == "ACYT"
    assert list(t["ad"]) == [10, 9, 8, 7]
    fwd = rec("ACGT", [30] * 4, cD=4, cE=0.0, extra=tags)
    same = lambda a, b: bamutil.parse(a)["tags"] == bamutil.parse(b)["tags"]
    assert same(orc.filter_process_record(orc.filter_options(reverse_per_base_tags=True), fwd)[2], fwd)     # forward reads are left alone
    assert same(orc.filter_process_record(orc.filter_options(), r)[2], r)


# ---- src/lib/tag_reversal.rs tests :130-260 ------------------------------------------------------------------------------
@pytest.mark.parametrize("ty,fmt", [("c", "b"), ("C", "B"), ("s", "h"), ("S", "H"), ("i", "i"), ("I", "I"), ("f", "f")])
def test_reverse_every_array_type(ty, fmt):
    raw = b"B" + ty.encode() + struct.pack("<I", 3) + struct.pack("<3" + fmt, 1, 2, 3)
    r = bamutil.make_record("q", "ACGT", [30] * 4, flag=UNMAPPED | 0x10, ref_id=-1, pos=-1, tags=[("cd", "raw", raw), ("zz", "raw", raw)])
    t = bamutil.parse(orc.filter_reverse_tags(r))["tags"]
    assert list(t["cd"][1]) == [3, 2, 1] and list(t["zz"][1]) == [1, 2, 3]          # only the per-base consensus tags are touched


def test_reverse_reference_string_and_array_pins():
    r = bamutil.make_record("q", "ACGT", [30] * 4, flag=UNMAPPED | 0x10, ref_id=-1, pos=-1,
                            tags=[("aq", "Z", "IIHG"), ("ac", "Z", "ACGA"), ("bc", "Z", "ACGA"), arr("cd", [1, 2, 3, 4]), ("bq", "Z", ""), ("ad", "i", 7)])
    t = {k: v[1] for k, v in bamutil.parse(orc.filter_reverse_tags(r))["tags"].items()}
    assert t["aq"] == "GHII" and t["ac"] == "TCGT" and t["bc"] == "TCGT" and list(t["cd"]) == [4, 3, 2, 1] and t["bq"] == "" and t["ad"] == 7
    fwd = bamutil.make_record("q", "ACGT", [30] * 4, flag=UNMAPPED, ref_id=-1, pos=-1, tags=[("aq", "Z", "IIHG"), arr("cd", [1, 2, 3, 4])])
    assert orc.filter_reverse_tags(fwd) == fwd                                      # positive strand: Ok(false), untouched


# ---- stream level ------------------------------------------------------------------------------------------------------
def P(name, ok1=True, ok2=True, n1=0, n2=0):
    """R1/R2 consensus pair; okX False → cD below min-reads 3; nX = bases masked by per-base depth."""
    def one(flag, ok, nm, base):
        cd = [1] * nm + [9] * (12 - nm)
        return rec(base * 12, [30] * 12, cD=9 if ok else 2, cE=0.0, cd=cd, ce=[0] * 12, flag=flag, name=name)
    return [one(0x4D, ok1, n1, "A"), one(0x8D, ok2, n2, "C")]


def test_stream_template_mode_and_rejects():
    recs = P("t1") + P("t2", ok2=False, n1=2) + P("t3", n1=1, n2=2) + P("t4", ok1=False)
    blob, off, ln = stream(recs)
    o = orc.filter_options(min_reads=3, track_rejects=True)
    res = orc.filter_records(o, blob, off, ln)
    kept, rej = bamutil_split(res["data"]), bamutil_split(res["rejects"])
    assert [bamutil.parse(r)["name"] for r in kept] == ["t1", "t1", "t3", "t3"] and [bamutil.parse(r)["name"] for r in rej] == ["t2", "t2", "t4", "t4"]
    assert (res["records"], res["passed"], res["rejected"], res["masked"]) == (8, 4, 4, 3)     # masked bases of retained primaries only
    assert bamutil.parse(kept[2])["seq"] == "N" + "A" * 11                                       # rejected and kept records are written masked
    assert bamutil.parse(rej[0])["seq"] == "NN" + "A" * 10
    o = orc.filter_options(min_reads=3, filter_by_template=False)
    res = orc.filter_records(o, blob, off, ln)
    assert (res["records"], res["passed"], res["rejected"], res["masked"]) == (8, 6, 0, 5) and res["rejects"] == b""


def bamutil_split(data):
    out, p = [], 0
    while p < len(data):
        n = struct.unpack_from("<I", data, p)[0]
        out.append(data[p + 4:p + 4 + n])
        p += 4 + n
    return out


def test_stream_template_ordering_and_secondary_rules():
    # Template::from_records: R1, R2, then R1 supplementaries, R2 supplementaries, R1 secondaries, R2 secondaries — each list reversed
    def r(flag, tag, ok=True):
        return rec("ACGTACGTAC", [30] * 10, cD=9 if ok else 1, cE=0.0, flag=flag | UNMAPPED, name="tpl", extra=[("xx", "Z", tag)])
    recs = [r(0x881, "r2supA"), r(0x81, "r2"), r(0x841, "r1supA"), r(0x141, "r1secA", ok=False), r(0x41, "r1"), r(0x841, "r1supB"), r(0x181, "r2sec")]
    blob, off, ln = stream(recs + P("next"))
    res = orc.filter_records(orc.filter_options(min_reads=3, track_rejects=True), blob, off, ln)
    kept = [bamutil.parse(x)["tags"].get("xx", ("Z", "-"))[1] for x in bamutil_split(res["data"])]
    assert kept == ["r1", "r2", "r1supB", "r1supA", "r2supA", "r2sec", "-", "-"]
    assert [bamutil.parse(x)["tags"]["xx"][1] for x in bamutil_split(res["rejects"])] == ["r1secA"]     # a failing secondary is dropped on its own
    blob, off, ln = stream([r(0x41, "a"), r(0x41, "b")])
    with pytest.raises(RuntimeError, match="Multiple non-secondary"):
        orc.filter_records(orc.filter_options(), blob, off, ln)
    blob, off, ln = stream([r(0x841, "only-supp")])        # no primary → the template fails (filter.rs:394)
    res = orc.filter_records(orc.filter_options(track_rejects=True), blob, off, ln)
    assert res["passed"] == 0 and res["rejected"] == 1
    blob, off, ln = stream([r(0x81, "r2first"), r(0x41, "r1second")])   # 2-record fast path swaps R2,R1 → R1,R2
    res = orc.filter_records(orc.filter_options(), blob, off, ln)
    assert [bamutil.parse(x)["tags"]["xx"][1] for x in bamutil_split(res["data"])] == ["r1second", "r2first"]


def test_stream_empty():
    blob, off, ln = stream([])
    res = orc.filter_records(orc.filter_options(), blob, off, ln)
    assert res["data"] == b"" and res["records"] == 0


# ---- src/lib/template.rs tests :1851-1945 (Template::from_records ordering) through the template-mode stream ---------------------------
def _ordered(names_flags):
    recs = [rec("ACGTACGTAC", [30] * 10, cD=9, cE=0.0, flag=f | UNMAPPED, name="read1", extra=[("xx", "Z", n)]) for n, f in names_flags]
    blob, off, ln = stream(recs)
    res = orc.filter_records(orc.filter_options(), blob, off, ln)
    return [bamutil.parse(x)["tags"]["xx"][1] for x in bamutil_split(res["data"])]


def test_template_record_ordering_reference_pins():
    P1, P2, SUP, SEC = 0x41, 0x81, 0x800, 0x100
    got = _ordered([("r2_sec1", P2 | SEC), ("r1_supp2", P1 | SUP), ("r1", P1), ("r2_supp", P2 | SUP), ("r1_sec", P1 | SEC), ("r2", P2), ("r2_sec2", P2 | SEC),
                    ("r1_supp1", P1 | SUP)])
    assert got == ["r1", "r2", "r1_supp1", "r1_supp2", "r2_supp", "r1_sec", "r2_sec2", "r2_sec1"]
    got = _ordered([("r1", P1), ("r2", P2), ("s1", P1 | SUP), ("s2", P1 | SUP), ("s3", P1 | SUP), ("t1", P2 | SUP), ("t2", P2 | SUP), ("c1", P1 | SEC), ("c2", P1 | SEC)])
    assert got == ["r1", "r2", "s3", "s2", "s1", "t2", "t1", "c2", "c1"]       # supplementaries / secondaries come out in reverse input order
