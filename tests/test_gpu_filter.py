"""GPU parity: `fgumi filter` on the device (fgx_filter_records / _device / fgx_filter_last_output_device) vs the oracle
restatement — kept bytes, rejected bytes and the FilterProcessedBatchRaw counters identical, for crafted records (every tag
type the reference's finders accept), for the consensus callers' own output, and for the device-resident consensus → filter
hand-over."""
import random
import struct

import numpy as np
import pytest

import bamutil
import orc
import test_oracle_filter as tof
from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, ConsensusFilter, DuplexConsensusCaller, FilterConfig, VanillaUmiConsensusCaller,
                       VanillaUmiConsensusOptions, record_offsets, simulate_grouped_reads)

pytestmark = pytest.mark.gpu


def _cfg(kw):
    return FilterConfig.new(kw.get("min_reads", [1]), kw.get("max_read_error_rate", [0.025]), kw.get("max_base_error_rate", [0.1]), kw.get("min_base_quality"),
                            kw.get("min_mean_base_quality"), kw.get("max_no_call_fraction", 0.2))


def _flags(kw):
    return {k: kw[k] for k in ("filter_by_template", "require_single_strand_agreement", "reverse_per_base_tags", "track_rejects") if k in kw}


def _same(records, **kw):
    blob, off, ln = tof.stream(records)
    want = orc.filter_records(orc.filter_options(**kw), blob, off, ln)
    f = ConsensusFilter(_cfg(kw), **_flags(kw))
    got = f.filter_stream(blob, off, ln)
    f.close()
    if got.data != want["data"]:
        a, b = tof.bamutil_split(got.data), tof.bamutil_split(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"kept record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError(f"kept count differs: {len(a)} vs {len(b)}")
    assert got.rejects == want["rejects"]
    assert (got.records_count, got.passed_count, got.bases_masked, got.rejected_count) == (want["records"], want["passed"], want["masked"], want["rejected"])
    return got


def crafted():
    rng = random.Random(7)
    recs = []
    for t in range(60):
        L = rng.choice([1, 2, 7, 30, 63, 64, 65, 129, 151])
        name = "t%03d" % t + "x" * rng.randrange(0, 9)          # varying name lengths: every alignment of the sequence / aux block
        for flag, base in ((0x4D, "ACGT"), (0x8D, "TGCA")):
            seq = "".join(rng.choice(base + "N" if rng.random() < 0.03 else base) for _ in range(L))
            q = [rng.choice([2, 5, 12, 25, 30, 40]) if rng.random() < 0.1 else 35 for _ in range(L)]
            cd = [rng.choice([0, 1, 2, 3, 9, 200]) if rng.random() < 0.08 else 9 for _ in range(rng.choice([L, L, L, max(0, L - 1), L + 2]))]
            ce = [rng.choice([1, 2]) if rng.random() < 0.05 else 0 for _ in range(L)]
            ty = rng.choice(["S", "s", "C", "c"])
            if ty in "cC":
                cd = [min(v, 100) for v in cd]
            if ty in "sc" and cd:
                cd[0] = -1
            tags = [("cD", "i", rng.choice([1, 2, 5, 9, 9, 300, 70000])), ("cM", "i", 1), ("cE", "f", rng.choice([0.0, 0.0, 0.01, 0.01, 0.03, 0.2]))]
            if rng.random() < 0.9:
                tags.append(tof.arr("cd", cd, ty))
            if rng.random() < 0.9:
                tags.append(tof.arr("ce", ce, rng.choice(["S", "C"])))
            tags.insert(rng.randrange(len(tags) + 1), ("RX", "Z", "ACGT-TTGA"))
            tags.insert(rng.randrange(len(tags) + 1), ("MI", "Z", str(t)))
            recs.append(bamutil.make_record(name, seq, q, flag=flag, ref_id=-1, pos=-1, tags=tags))
    # duplex records with every per-base tag, in shuffled tag order
    for t in range(60):
        L = rng.choice([3, 8, 50, 100, 150])
        name = "d%03d" % t
        for flag in (0x4D, 0x8D):
            seq = "".join(rng.choice("ACGTN" if rng.random() < 0.05 else "ACGT") for _ in range(L))
            q = [rng.choice([2, 10, 30]) if rng.random() < 0.06 else 60 for _ in range(L)]
            ad = [rng.choice([0, 1, 2, 5]) if rng.random() < 0.06 else 9 for _ in range(L)]
            bd = [rng.choice([0, 1, 2, 5]) if rng.random() < 0.06 else 7 for _ in range(L)]
            ae = [1 if rng.random() < 0.04 else 0 for _ in range(L)]
            be = [rng.choice([1, 3]) if rng.random() < 0.04 else 0 for _ in range(L)]
            ac = "".join(rng.choice("ACGT") for _ in range(rng.choice([L, L, L - 1])))
            bc = "".join(a if rng.random() < 0.97 else "T" for a in ac.ljust(L, "A"))
            tags = [("cD", "i", max(ad) + max(bd)), ("cE", "f", 0.01), ("aD", "i", max(ad)), ("bD", "i", max(bd)), ("aE", "f", rng.choice([0.0, 0.0, 0.02, 0.04])),
                    ("bE", "f", rng.choice([0.0, 0.0, 0.02, 0.06])), ("aM", "i", min(ad)), ("bM", "i", min(bd)), tof.arr("ad", ad), tof.arr("bd", bd, rng.choice(["S", "C"])),
                    tof.arr("ae", ae), tof.arr("be", be), ("aq", "Z", "I" * L), ("bq", "Z", "5" * L), ("MI", "Z", str(t)), ("RG", "Z", "A")]
            if rng.random() < 0.8:
                tags.append(("ac", "Z", ac))
            if rng.random() < 0.8:
                tags.append(("bc", "Z", bc) if rng.random() < 0.7 else tof.arr("bc", [ord(x) for x in bc], "C"))
            rng.shuffle(tags)
            if rng.random() < 0.1:
                tags = [x for x in tags if x[0] != "bD"]           # aD without bD → simplex rules
            recs.append(bamutil.make_record(name, seq, q, flag=flag, ref_id=-1, pos=-1, tags=tags))
    return recs


OPTION_SETS = [dict(), dict(min_reads=[3], track_rejects=True), dict(min_reads=[5, 3, 2], max_read_error_rate=[0.03, 0.02, 0.05], max_base_error_rate=[0.3, 0.2, 0.4],
                                                                       min_base_quality=12, track_rejects=True),
               dict(min_reads=[2], filter_by_template=False, track_rejects=True, min_base_quality=10), dict(min_mean_base_quality=24.0, max_no_call_fraction=0.5),
               dict(max_no_call_fraction=3.0, min_base_quality=13, track_rejects=True), dict(require_single_strand_agreement=True, min_reads=[2, 1, 1]),
               dict(min_reads=[1], max_base_error_rate=[1.0], max_read_error_rate=[1.0], max_no_call_fraction=1000.0)]


@pytest.mark.parametrize("kw", OPTION_SETS)
def test_filter_crafted(kw):
    _same(crafted(), **kw)


def test_filter_reference_unit_cases_on_device():
    # src/lib/commands/filter.rs:1680-1760 mask_bases cases, through the whole process_record_raw
    got = _same([tof.rec("ACGT", [10, 30, 5, 30], cD=10, cE=0.0, cd=[10] * 4, ce=[0] * 4)], min_base_quality=20, max_no_call_fraction=4.0)
    assert tof.seq_quals(tof.bamutil_split(got.data)[0]) == ("NCNT", [2, 30, 2, 30]) and got.bases_masked == 2
    got = _same([tof.rec("ACGT", [30] * 4, cD=10, cE=0.0, cd=[10] * 4, ce=[1, 3, 2, 0])], max_base_error_rate=[0.2], max_no_call_fraction=4.0)
    assert tof.seq_quals(tof.bamutil_split(got.data)[0])[0] == "ANGT"


def test_filter_template_ordering_and_secondaries():
    def r(flag, tag, ok=True):
        return tof.rec("ACGTACGTAC", [30] * 10, cD=9 if ok else 1, cE=0.0, flag=flag | 4, name="tpl", extra=[("xx", "Z", tag)])
    recs = [r(0x881, "r2supA"), r(0x81, "r2"), r(0x841, "r1supA"), r(0x141, "r1secA", ok=False), r(0x41, "r1"), r(0x841, "r1supB"), r(0x181, "r2sec")]
    got = _same(recs + tof.P("next") + [r(0x841, "only-supp").replace(b"tpl\0", b"zzz\0")], min_reads=[3], track_rejects=True)
    assert [bamutil.parse(x)["tags"].get("xx", ("Z", "-"))[1] for x in tof.bamutil_split(got.data)] == ["r1", "r2", "r1supB", "r1supA", "r2supA", "r2sec", "-", "-"]
    _same(recs + tof.P("next"), min_reads=[3], filter_by_template=False, track_rejects=True)
    _same([r(0x81, "r2first"), r(0x41, "r1second")])


def test_filter_reverse_tags_and_large_records():
    rng = random.Random(3)
    recs = []
    for t, L in enumerate([4, 33, 150, 151, 3000, 7000]):        # the last two exceed the LDS slice: read from HBM
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        cd = [rng.choice([1, 4, 9]) for _ in range(L)]
        ce = [rng.choice([0, 0, 1]) for _ in range(L)]
        ac = "".join(rng.choice("ACGTRYN") for _ in range(L))
        tags = [tof.arr("cd", cd), tof.arr("ce", ce, "C"), ("aq", "Z", "".join(chr(33 + (i % 60)) for i in range(L))), ("ac", "Z", ac), ("bc", "Z", ac[::-1]),
                tof.arr("ad", list(range(L)), "i"), tof.arr("ct", [i % 100 for i in range(L)], "s"), ("bq", "Z", "")]
        for flag in (0x4 | 0x10, 0x4):
            recs.append(tof.rec(seq, [rng.choice([5, 30]) for _ in range(L)], cD=9, cE=0.0, flag=flag, name=f"rev{t}_{flag}", extra=tags))
    _same(recs, reverse_per_base_tags=True, min_reads=[3], min_base_quality=10, max_no_call_fraction=0.9, filter_by_template=False, track_rejects=True)
    _same(recs, reverse_per_base_tags=False, min_reads=[3], max_no_call_fraction=0.9, filter_by_template=False)


def test_filter_errors_are_fatal():
    f = ConsensusFilter(FilterConfig.new([1]))
    ok1, ok2 = tof.rec("ACGT", [30] * 4, cD=5, cE=0.0, name="a"), tof.rec("ACGT", [30] * 4, cD=5, cE=0.0, name="c")
    for bad, msg in ((tof.rec("ACGT", [30] * 4, cD=5, cE=0.0, flag=0, pos=10, name="b"), "--ref is required"), (tof.rec("ACGT", [30] * 4, cD=5, name="b"), "cD/cE"),
                     (tof.rec("ACGT", [30] * 4, cE=0.1, name="b"), "cD/cE")):
        with pytest.raises(RuntimeError, match=msg):
            f.filter_stream(*tof.stream([ok1, bad, ok2]))
        with pytest.raises(RuntimeError):
            orc.filter_records(orc.filter_options(), *tof.stream([ok1, bad, ok2]))
    two_r1 = [tof.rec("ACGT", [30] * 4, cD=5, cE=0.0, flag=0x45, name="same"), tof.rec("ACGT", [30] * 4, cD=5, cE=0.0, flag=0x45, name="same")]
    with pytest.raises(RuntimeError, match="Multiple non-secondary"):
        f.filter_stream(*tof.stream(two_r1))
    assert f.filter_stream(*tof.stream([])).records_count == 0
    f.close()
    with pytest.raises(ValueError):
        FilterConfig.new([1, 2])                      # AB > duplex
    with pytest.raises(ValueError):
        FilterConfig.new([1], max_no_call_fraction=1.5)


def _consensus_records(kind, n):
    if kind == "simplex":
        g = simulate_grouped_reads(n, family_size=2, family_size_max=12, error_rate_ppm=20000)
        c = VanillaUmiConsensusCaller("c", "A", VanillaUmiConsensusOptions(min_reads=1, produce_per_base_tags=True, min_consensus_base_quality=2))
    elif kind == "duplex":
        g = simulate_grouped_reads(n, family_size=4, family_size_max=16, duplex=1, error_rate_ppm=20000)
        c = DuplexConsensusCaller("d", "A", [1], produce_per_base_tags=True)
    else:
        g = simulate_grouped_reads(n, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1, error_rate_ppm=20000)
        c = CodecConsensusCaller("x", "A", CodecConsensusOptions(produce_per_base_tags=True))
    out = c.process_batch(g)
    c.close()
    return out.data


@pytest.mark.parametrize("kind", ["simplex", "duplex", "codec"])
def test_filter_consensus_caller_output(kind):
    data = _consensus_records(kind, 600)
    off, ln = record_offsets(data)
    assert len(off) > 500
    blob = np.frombuffer(data + b"\0" * 8, dtype=np.uint8)
    passed = []
    for kw in (dict(min_reads=[3], track_rejects=True), dict(min_reads=[6, 3, 2], max_base_error_rate=[0.05], min_base_quality=30, track_rejects=True,
                                                              require_single_strand_agreement=True),
               dict(min_reads=[2], min_mean_base_quality=35.0, filter_by_template=False, max_no_call_fraction=0.05)):
        want = orc.filter_records(orc.filter_options(**kw), blob, off, ln)
        f = ConsensusFilter(_cfg(kw), **_flags(kw))
        got = f.filter_records(data)
        f.close()
        assert got.data == want["data"] and got.rejects == want["rejects"]
        assert (got.records_count, got.passed_count, got.bases_masked, got.rejected_count) == (want["records"], want["passed"], want["masked"], want["rejected"])
        passed.append(got.passed_count)
    assert any(0 < p < len(off) for p in passed)


@pytest.mark.parametrize("kind", ["simplex", "duplex", "codec"])
def test_consensus_then_filter_without_leaving_the_device(kind):
    """process_batch_device → filter_last_output_device equals (oracle consensus → oracle filter) on the same simulated molecules."""
    import fgx_opts
    if kind == "simplex":
        sim = dict(n_families=1500, family_size=2, family_size_max=12, error_rate_ppm=20000)
        c = VanillaUmiConsensusCaller("c", "A", VanillaUmiConsensusOptions(min_reads=1, produce_per_base_tags=True, min_consensus_base_quality=2))
        o = fgx_opts.defaults(kind=0, read_name_prefix=b"c", min_reads=1, produce_per_base_tags=1, overlapping_consensus=0)
    elif kind == "duplex":
        sim = dict(n_families=800, family_size=4, family_size_max=16, duplex=1, error_rate_ppm=20000)
        c = DuplexConsensusCaller("d", "A", [1], produce_per_base_tags=True)
        o = fgx_opts.defaults(kind=1, read_name_prefix=b"d", produce_per_base_tags=1, overlapping_consensus=0)
    else:
        sim = dict(n_families=800, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1, error_rate_ppm=20000)
        c = CodecConsensusCaller("x", "A", CodecConsensusOptions(produce_per_base_tags=True))
        o = fgx_opts.defaults(kind=2, read_name_prefix=b"x", overlapping_consensus=0, produce_per_base_tags=1)
    dg = c.simulate_on_device(**sim)
    out = c.process_batch_device(dg)
    assert out.n_deferred == 0
    kw = dict(min_reads=[3, 2, 1], max_base_error_rate=[0.2], min_base_quality=20, track_rejects=True)
    f = ConsensusFilter.on_caller(c, _cfg(kw), **_flags(kw))
    got = f.filter_last_output_device().to_host()
    g = simulate_grouped_reads(sim["n_families"], **{k: v for k, v in sim.items() if k != "n_families"})
    cons = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=g.n_grp)["data"]
    off, ln = record_offsets(cons)
    want = orc.filter_records(orc.filter_options(**kw), np.frombuffer(cons + b"\0" * 8, dtype=np.uint8), off, ln)
    assert got.data == want["data"] and got.rejects == want["rejects"]
    assert (got.records_count, got.passed_count, got.bases_masked) == (want["records"], want["passed"], want["masked"])
    assert got.passed_count > 0 and got.rejected_count > 0
    c.close()


def test_filter_odd_records_and_tag_encodings():
    """Corners of the raw-tag readers (crates/fgumi-raw-bam/src/tags.rs): every integer width for cD, duplicate tags (first occurrence
    wins), A / H / B:f / B:i entries to walk over, an unterminated Z string that ends the walk, empty reads, a read without aux data."""
    import struct
    recs = []
    widths = [("c", "<b", 7), ("C", "<B", 200), ("s", "<h", 300), ("S", "<H", 40000), ("i", "<i", 70000), ("I", "<I", 3000000000), ("c", "<b", -3)]
    for k, (ty, fmt, val) in enumerate(widths):
        tags = [("cD", "raw", ty.encode() + struct.pack(fmt, val)), ("XA", "raw", b"Aq"), ("XH", "raw", b"H1AF0\0"), tof.arr("XF", [1, 2], "i"), ("cE", "f", 0.01),
                tof.arr("cd", [9] * 10), tof.arr("ce", [0] * 10, "C"), ("cD", "i", 1), ("cE", "f", 0.9)]       # the later duplicates are ignored
        recs.append(bamutil.make_record(f"w{k}", "ACGTACGTAC", [30] * 10, flag=4, ref_id=-1, pos=-1, tags=tags))
    # B:f array named like a per-base tag: find_array_tag accepts it, every element reads as 0
    recs.append(bamutil.make_record("bf", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1,
                                    tags=[("cD", "i", 9), ("cE", "f", 0.0), ("cd", "raw", b"Bf" + struct.pack("<I4f", 4, 1.0, 2.0, 3.0, 4.0)), tof.arr("ce", [0] * 4)]))
    # unterminated Z after the consensus tags: the walk ends there, what came before still counts
    recs.append(bamutil.make_record("unterminated", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1,
                                    tags=[("cD", "i", 9), ("cE", "f", 0.0), ("ZZ", "raw", b"Zno-nul-here")]))
    # ... and before them: cD / cE are never seen, which is the reference's fatal error — checked separately below
    recs.append(bamutil.make_record("empty", "", [], flag=4, ref_id=-1, pos=-1, tags=[("cD", "i", 9), ("cE", "f", 0.0)]))
    recs.append(bamutil.make_record("odd", "ACG", [30, 2, 30], flag=4, ref_id=-1, pos=-1, tags=[("cD", "i", 9), ("cE", "f", 0.0), tof.arr("cd", [9, 9]), tof.arr("ce", [0, 0, 5])]))
    for kw in (dict(min_reads=[2], filter_by_template=False, track_rejects=True, min_base_quality=10), dict(min_reads=[250], max_no_call_fraction=1000.0, track_rejects=True),
               dict(min_reads=[1], max_no_call_fraction=0.0, min_mean_base_quality=20.5, track_rejects=True)):
        _same(recs, **kw)
    f = ConsensusFilter(FilterConfig.new([1]))
    bad = bamutil.make_record("hidden", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1, tags=[("ZZ", "raw", b"Zno-nul"), ("cD", "i", 9), ("cE", "f", 0.0)])
    with pytest.raises(RuntimeError, match="cD/cE"):
        f.filter_stream(*tof.stream([bad]))
    with pytest.raises(RuntimeError):
        orc.filter_records(orc.filter_options(), *tof.stream([bad]))
    no_aux = bamutil.make_record("noaux", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1)
    with pytest.raises(RuntimeError, match="cD/cE"):
        f.filter_stream(*tof.stream([no_aux]))
    wrong_type = bamutil.make_record("ce_int", "ACGT", [30] * 4, flag=4, ref_id=-1, pos=-1, tags=[("cD", "i", 9), ("cE", "i", 0)])
    with pytest.raises(RuntimeError, match="cD/cE"):
        f.filter_stream(*tof.stream([wrong_type]))
    f.close()


# ---- `fgumi filter --ref`: mapped consensus records, NM / UQ / MD regenerated after the masking (filter.rs:115-118, 888-890) ---------------------
def _mapped_consensus_records(rng, contigs, n_templates):
    """Consensus-shaped records as they look after alignment: mapped pairs (FR, reverse mates with reversed per-base tags left to the filter),
    CIGARs with clips / indels, stale NM / UQ / MD of several encodings among the consensus tags, a few unmapped mates and a few fragments."""
    recs = []
    for t in range(n_templates):
        L = rng.choice([20, 45, 80, 151])
        rid = rng.randrange(len(contigs))
        contig = contigs[rid]
        paired = rng.random() < 0.85
        for mate in ((1, 2) if paired else (0,)):
            lead, trail = rng.choice([0, 0, 3]), rng.choice([0, 0, 5])
            core = L - lead - trail
            a = max(1, core // 2 - 1)
            if rng.random() < 0.25 and core > 12:
                ops, span = f"{lead}S" * (lead > 0) + f"{a}M2I{core - a - 2}M" + f"{trail}S" * (trail > 0), core - 2
            elif rng.random() < 0.25 and core > 12:
                ops, span = f"{lead}S" * (lead > 0) + f"{a}M3D{core - a}M" + f"{trail}S" * (trail > 0), core + 3
            else:
                ops, span = f"{lead}S" * (lead > 0) + f"{core}M" + f"{trail}S" * (trail > 0), core
            start = rng.randrange(0, len(contig) - span - 1)
            seq = "".join((chr(contig[(start + i) % len(contig)]).upper() if rng.random() < 0.95 else rng.choice("ACGT")) for i in range(L))
            q = [rng.choice([5, 12, 25]) if rng.random() < 0.1 else 40 for _ in range(L)]
            cd = [rng.choice([1, 2]) if rng.random() < 0.1 else 8 for _ in range(L)]
            ce = [1 if rng.random() < 0.03 else 0 for _ in range(L)]
            unmapped = rng.random() < 0.06
            flag = (0x1 | (0x40 if mate == 1 else 0x80) if mate else 0) | (0x10 if (mate == 2 and not unmapped) else 0) | (0x4 if unmapped else 0)
            tags = [("RG", "Z", "A"), ("cD", "i", max(cd)), ("cM", "i", min(cd)), ("cE", "f", sum(ce) / max(1, sum(cd))),
                    ("cd", "raw", b"Bs" + struct.pack("<I", L) + struct.pack(f"<{L}h", *cd)), ("ce", "raw", b"Bs" + struct.pack("<I", L) + struct.pack(f"<{L}h", *ce)),
                    ("MI", "Z", str(t))]
            if rng.random() < 0.7:
                tags.insert(rng.randrange(len(tags) + 1), ("NM", "raw", rng.choice([b"c\x02", b"i\x02\x00\x00\x00", b"S\x10\x27"])))
            if rng.random() < 0.7:
                tags.insert(rng.randrange(len(tags) + 1), ("MD", "Z", rng.choice([str(L), "10A5^AC20", "0"])))
            if rng.random() < 0.5:
                tags.insert(rng.randrange(len(tags) + 1), ("UQ", "raw", rng.choice([b"C\x2d", b"I\x2d\x00\x00\x00"])))
            recs.append(bamutil.make_record(f"tmpl{t:05d}", seq, q, flag=flag, ref_id=-1 if unmapped else rid, pos=-1 if unmapped else start, cigar=None if unmapped else ops, tags=tags))
    return recs


@pytest.mark.parametrize("kw", [dict(min_reads=[3], min_base_quality=20, track_rejects=True), dict(min_reads=[1], filter_by_template=False, reverse_per_base_tags=True, track_rejects=True),
                                dict(min_reads=[8], max_no_call_fraction=0.3, track_rejects=True)])
def test_filter_with_a_reference_regenerates_alignment_tags(kw):
    rng = random.Random(99)
    contigs = [bytes(rng.choice(b"ACGTacgtN") for _ in range(L)) for L in (3000, 800)]
    recs = _mapped_consensus_records(rng, contigs, 700)
    blob, off, ln = tof.stream(recs)
    orc.set_reference(contigs)
    try:
        want = orc.filter_records(orc.filter_options(regenerate_alignment_tags=True, **kw), blob, off, ln)
    finally:
        orc.set_reference(None)
    f = ConsensusFilter(_cfg(kw), **_flags(kw))
    f.set_reference({f"chr{i}": s for i, s in enumerate(contigs)}, [f"chr{i}" for i in range(len(contigs))])
    got = f.filter_stream(blob, off, ln)
    if got.data != want["data"]:
        a, b = tof.bamutil_split(got.data), tof.bamutil_split(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"kept record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError(f"kept count differs: {len(a)} vs {len(b)}")
    assert got.rejects == want["rejects"] and got.rejected_count == want["rejected"] > 0
    assert (got.records_count, got.passed_count, got.bases_masked) == (want["records"], want["passed"], want["masked"]) and got.passed_count > 50, got.passed_count
    assert b"MD" in got.data
    # without the reference the first mapped record is the reference's fatal error
    f.set_reference(None, [])
    with pytest.raises(RuntimeError, match="--ref is required"):
        f.filter_stream(blob, off, ln)
    # and an alignment that leaves its contig is fatal with it
    f.set_reference({"chr0": contigs[0][:50], "chr1": contigs[1]}, ["chr0", "chr1"])
    with pytest.raises(RuntimeError, match="leaves its reference"):
        f.filter_stream(blob, off, ln)
    # a header contig the FASTA lacks is fatal only when a mapped record lies on it (the reference fails lazily: "Reference not found: <contig>",
    # fgumi-sam alignment_tags.rs:482): an extra header contig with no reads changes nothing, a contig with reads raises before anything is filtered
    f.set_reference({f"chr{i}": s for i, s in enumerate(contigs)}, ["chr0", "chr1", "chrUn_decoy"])
    assert f.filter_stream(blob, off, ln).data == want["data"]
    f.set_reference({"chr0": contigs[0]}, ["chr0", "chr1"])
    with pytest.raises(KeyError, match="Reference not found: chr1"):
        f.filter_stream(blob, off, ln)
    f.close()


# ---- the methylation (EM-Seq / TAPs) filters (src/lib/commands/filter.rs:181-206, 833-937; crates/fgumi-consensus/src/filter.rs:925-1340) -----------
def _methylation_records(rng, contigs, n_templates):
    """Mapped simplex and duplex consensus records with the methylation count arrays of the methylation-aware callers (cu ct, duplex: au at bu bt)
    in several encodings and with gaps (a tag missing, an array shorter than the read), CpG-rich contigs, strands that disagree at some CpGs."""
    def arr(tag, v):
        ty = rng.choice("sSC")
        v = [min(x, 255) for x in v] if ty == "C" else v
        return (tag, "raw", b"B" + ty.encode() + struct.pack("<I", len(v)) + struct.pack(f"<{len(v)}{'h' if ty == 's' else 'H' if ty == 'S' else 'B'}", *v))
    recs = []
    for t in range(n_templates):
        L = rng.choice([9, 20, 45, 80, 151])
        rid = rng.randrange(len(contigs))
        contig = contigs[rid]
        duplex = rng.random() < 0.5
        for mate in (1, 2):
            lead, trail = rng.choice([0, 0, 3]), rng.choice([0, 0, 4])
            core = L - lead - trail
            a = max(1, core // 2 - 1)
            pick = rng.random()
            if pick < 0.2 and core > 12:
                ops, span = f"{lead}S" * (lead > 0) + f"{a}M2I{core - a - 2}M" + f"{trail}S" * (trail > 0), core - 2
            elif pick < 0.4 and core > 12:
                ops, span = f"{lead}S" * (lead > 0) + f"{a}M3D{core - a}M" + f"{trail}S" * (trail > 0), core + 3
            elif pick < 0.5 and core > 12:
                ops, span = f"{lead}S" * (lead > 0) + f"{a}M7N{core - a}M" + f"{trail}S" * (trail > 0), core + 7
            else:
                ops, span = f"{lead}S" * (lead > 0) + f"{core}M" + f"{trail}S" * (trail > 0), core
            start = rng.randrange(0, len(contig) - span - 1)
            seq = "".join((chr(contig[(start + i) % len(contig)]).upper() if rng.random() < 0.93 else rng.choice("ACGTN")) for i in range(L))
            q = [rng.choice([5, 25]) if rng.random() < 0.05 else 40 for _ in range(L)]
            unmapped = rng.random() < 0.05
            flag = 0x1 | (0x40 if mate == 1 else 0x80) | (0x10 if (mate == 2 and not unmapped) else 0) | (0x4 if unmapped else 0)
            cd = [rng.choice([1, 2]) if rng.random() < 0.04 else 8 for _ in range(L)]
            ce = [0] * L
            tags = [("RG", "Z", "A"), ("cD", "i", max(cd)), ("cM", "i", min(cd)), ("cE", "f", 0.0), arr("cd", cd), arr("ce", ce), ("MI", "Z", str(t))]
            if duplex:
                tags += [("aD", "i", 4), ("bD", "i", 4), ("aE", "f", 0.0), ("bE", "f", 0.0), arr("ad", [4] * L), arr("bd", [4] * L), arr("ae", ce), arr("be", ce)]
            # counts: a converted (methylation-insensitive) library converts most non-CpG Cs; one template in five is badly converted
            bad = rng.random() < 0.2
            def counts(total):
                u = [0] * L
                c = [0] * L
                for i in range(L):
                    n = rng.choice([0, 1, 2]) if rng.random() < 0.08 else total
                    k = sum(rng.random() < (0.6 if bad else 0.05) for _ in range(n)) if rng.random() < 0.8 else rng.choice([0, n])
                    u[i], c[i] = k, n - k
                return u, c
            au, at = counts(4)
            bu, bt = counts(4)
            cu, ct = [x + y for x, y in zip(au, bu)], [x + y for x, y in zip(at, bt)]
            short = rng.random() < 0.1
            meth = [arr("cu", cu[:L - 2] if short else cu), arr("ct", ct)]
            if duplex:
                meth += [arr("au", au), arr("at", at), arr("bu", bu), arr("bt", bt[:max(0, L - 3)] if short else bt)]
            for k in range(len(meth) - 1, -1, -1):
                if rng.random() < 0.06:
                    del meth[k]
            if rng.random() < 0.05:
                meth = []
            for m in meth:
                tags.insert(rng.randrange(len(tags) + 1), m)
            if rng.random() < 0.5:
                tags.insert(rng.randrange(len(tags) + 1), ("NM", "raw", b"c\x02"))
            recs.append(bamutil.make_record(f"tmpl{t:05d}", seq, q, flag=flag, ref_id=-1 if unmapped else rid, pos=-1 if unmapped else start, cigar=None if unmapped else ops, tags=tags))
    return recs


METHYLATION_OPTION_SETS = [
    dict(min_reads=[1], min_methylation_depth=[6], track_rejects=True),
    dict(min_reads=[1], min_methylation_depth=[7, 4, 3], max_no_call_fraction=0.5, filter_by_template=False, track_rejects=True),
    dict(min_reads=[1], require_strand_methylation_agreement=True, track_rejects=True, max_no_call_fraction=0.4),
    dict(min_reads=[1], min_conversion_fraction=0.8, methylation_mode="em-seq", track_rejects=True, max_no_call_fraction=1000.0),
    dict(min_reads=[1], min_conversion_fraction=0.3, methylation_mode="taps", track_rejects=True, filter_by_template=False, max_no_call_fraction=1000.0),
    dict(min_reads=[3], min_base_quality=20, min_methylation_depth=[5, 2], require_strand_methylation_agreement=True, min_conversion_fraction=0.75,
         methylation_mode="em-seq", reverse_per_base_tags=True, track_rejects=True, max_no_call_fraction=0.6),
]


@pytest.mark.parametrize("kw", METHYLATION_OPTION_SETS)
def test_filter_methylation_filters(kw):
    rng = random.Random(2024)
    contigs = [bytes(rng.choice(b"ACGCGTacgN" if rng.random() < 0.9 else b"CG") for _ in range(n)) for n in (4000, 900)]
    recs = _methylation_records(rng, contigs, 600)
    blob, off, ln = tof.stream(recs)
    okw = dict(kw)
    okw["methylation_mode"] = {"em-seq": 1, "taps": 2, None: 0}[kw.get("methylation_mode")]
    orc.set_reference(contigs)
    try:
        want = orc.filter_records(orc.filter_options(regenerate_alignment_tags=True, **okw), blob, off, ln)
        plain = orc.filter_records(orc.filter_options(regenerate_alignment_tags=True, **{k: v for k, v in okw.items() if "methylation" not in k and "conversion" not in k}), blob, off, ln)
    finally:
        orc.set_reference(None)
    assert (want["data"], want["masked"], want["passed"]) != (plain["data"], plain["masked"], plain["passed"])       # the filters under test do something here
    f = ConsensusFilter(_cfg(kw), **_flags(kw), **{k: kw[k] for k in ("min_methylation_depth", "require_strand_methylation_agreement", "min_conversion_fraction", "methylation_mode") if k in kw})
    f.set_reference({f"chr{i}": s for i, s in enumerate(contigs)}, [f"chr{i}" for i in range(len(contigs))])
    got = f.filter_stream(blob, off, ln)
    if got.data != want["data"]:
        a, b = tof.bamutil_split(got.data), tof.bamutil_split(want["data"])
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                raise AssertionError(f"kept record {i} differs:\n got {bamutil.parse(x)}\nwant {bamutil.parse(y)}")
        raise AssertionError(f"kept count differs: {len(a)} vs {len(b)}")
    assert got.rejects == want["rejects"]
    assert (got.records_count, got.passed_count, got.bases_masked, got.rejected_count) == (want["records"], want["passed"], want["masked"], want["rejected"])
    assert 0 < got.passed_count < got.records_count
    f.close()


def test_filter_methylation_option_checks():
    """Filter::validate (src/lib/commands/filter.rs:1110-1154): the value order of --min-methylation-depth, the reference both reference-dependent
    filters need, the mode the conversion fraction needs."""
    blob, off, ln = tof.stream([tof.rec("ACGT", [30] * 4, cD=10, cE=0.0, cd=[10] * 4, ce=[0] * 4)])
    for kw, msg in ((dict(min_methylation_depth=[2, 5]), "high to low"), (dict(min_methylation_depth=[5, 2, 3]), "high to low"),
                    (dict(require_strand_methylation_agreement=True), "requires --ref"), (dict(min_conversion_fraction=0.5, methylation_mode="taps"), "requires --ref"),
                    (dict(min_conversion_fraction=1.5, methylation_mode="taps"), "between 0.0 and 1.0")):
        f = ConsensusFilter(FilterConfig.new([1]), **kw)
        with pytest.raises(RuntimeError, match=msg):
            f.filter_stream(blob, off, ln)
        f.close()
    f = ConsensusFilter(FilterConfig.new([1]), min_conversion_fraction=0.5)
    f.set_reference({"chr0": b"ACGT"}, ["chr0"])
    with pytest.raises(RuntimeError, match="--methylation-mode"):
        f.filter_stream(blob, off, ln)
    f.close()
    with pytest.raises(ValueError, match="1-3 values"):
        ConsensusFilter(FilterConfig.new([1]), min_methylation_depth=[4, 3, 2, 1])
