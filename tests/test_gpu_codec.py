"""GPU parity: CODEC caller (general path: host geometry, both single-strand column sets on the device) vs the
oracle — byte-identical records, stats (incl. the CODEC-only counters) and rejects."""
import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions, GroupedReads, simulate_grouped_reads, split_records
import test_oracle_codec as toc

pytestmark = pytest.mark.gpu

GENERAL_ONLY = False


@pytest.fixture(autouse=True, params=["device", "general"])
def path_mode(request):
    """Every case runs through the device-resident CODEC pipeline (deferred molecules fall back to the general path) and
    through the general host-orchestrated path alone."""
    global GENERAL_ONLY
    GENERAL_ONLY = request.param == "general"
    yield
    GENERAL_ONLY = False


def _same(g, track_rejects=False, prefix="codec", rg="A", **kw):
    v = CodecConsensusOptions(**kw)
    okw = dict(read_name_prefix=prefix.encode(), read_group_id=rg.encode(), overlapping_consensus=0, track_rejects=int(track_rejects),
               cell_tag=(v.cell_tag.encode() if v.cell_tag else b"\0\0"), produce_per_base_tags=int(v.produce_per_base_tags),
               min_input_base_quality=v.min_input_base_quality, error_rate_pre_umi=v.error_rate_pre_umi, error_rate_post_umi=v.error_rate_post_umi,
               codec_min_reads_per_strand=v.min_reads_per_strand, codec_max_reads_per_strand=-1 if v.max_reads_per_strand is None else v.max_reads_per_strand,
               codec_min_duplex_length=v.min_duplex_length, codec_has_single_strand_qual=int(v.single_strand_qual is not None),
               codec_single_strand_qual=v.single_strand_qual or 0, codec_has_outer_bases_qual=int(v.outer_bases_qual is not None),
               codec_outer_bases_qual=v.outer_bases_qual or 0, codec_outer_bases_length=v.outer_bases_length,
               codec_max_duplex_disagreements=0xFFFFFFFF if v.max_duplex_disagreements is None else v.max_duplex_disagreements,
               codec_max_duplex_disagreement_rate=v.max_duplex_disagreement_rate, tie_rule=v.tie_rule)
    o = fgx_opts.defaults(kind=2, **okw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=max(1, g.n_grp))
    c = CodecConsensusCaller(prefix, rg, v, track_rejects=track_rejects)
    c.set_general_only(GENERAL_ONLY)
    out = c.process_batch(g)
    st = c.last_batch_statistics()
    cst = c.codec_statistics()
    rej = c.take_rejected_reads()
    c.close()
    assert out.count == want["count"]
    if out.data != want["data"]:
        for i, (a, b) in enumerate(zip(split_records(out.data), split_records(want["data"]))):
            if a != b:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(a)}\nwant {bamutil.parse(b)}")
        raise AssertionError("record count/length differs")
    arr = np.zeros(28, dtype=np.uint64)
    arr[0], arr[1], arr[2] = st.total_reads, st.consensus_reads, st.filtered_reads
    for r, n in st.rejection_reasons.items():
        arr[3 + int(r)] = n
    arr[24:28] = [cst.consensus_bases_emitted, cst.consensus_duplex_bases_emitted, cst.duplex_disagreement_base_count, cst.consensus_reads_rejected_hdd]
    assert np.array_equal(arr, want["stats"]), (arr.tolist(), want["stats"].tolist())
    if track_rejects:
        assert rej == split_records(want["rejects"])
    return out, want


def test_reference_codec_geometry_inputs():
    """The overlap-geometry fixtures of the reference's own CODEC unit tests (indel at the overlap boundary, dovetailed starts,
    terminal indel outside / indel inside the shared region, R1 running past R2, cross-template overlap; codec_caller.rs:3875-4322 —
    tests/test_oracle_codec.py pins their exact consensus bases on the oracle) through the HIP path: byte-identical."""
    fams = [sum((toc.fr_pair(f"t{i}", 200, 200, 35, "2S124M1D3M", "3S125M", mi="mi", rx="ACC-TGA", ref=toc.REF_BASES) for i in range(2)), []),
            toc._codec_template(*toc._DOVETAIL), toc._codec_template(201, "2S124M1D3M", 200, "3S124M2S"),
            toc._codec_template(201, "2S60M1D67M", 200, "3S124M2S"), toc._window_end_past_r2_fixture(),
            toc.fr_pair("tA", 200, 200, 35, "50M", "50M", mi="mi", rx="ACC-TGA", ref=toc.REF_BASES) +
            toc.fr_pair("tB", 199, 199, 35, "40M", "102M", mi="mi", rx="ACC-TGA", ref=toc.REF_BASES)]
    for recs in fams:
        _same(GroupedReads.from_groups([recs]), rg="RG1", min_reads_per_strand=1, min_duplex_length=1)
    for mdl in (126, 127):
        _same(GroupedReads.from_groups([toc._codec_template(*toc._DOVETAIL)]), rg="RG1", min_reads_per_strand=1, min_duplex_length=mdl)
    _same(GroupedReads.from_groups(fams), rg="RG1", track_rejects=True, min_reads_per_strand=1, min_duplex_length=1)   # one batch, six molecules


def crafted_groups():
    P = toc.fr_pair
    groups = [
        P("a", 200, 200, 35, "30M", "30M", rx=None), P("b", 1, 11, 35, "30M", "30M"), P("c", 1, 13, 35, "5M2D25M", "30M"),
        P("d", 1, 11, 35, "30M", "25M5D5M"), P("e", 100, 135, 35, "30M", "30M", rev1=True, rev2=False), P("f", 100, 95, 35, "50M", "50M"),
        P("g", 11, 1, 35, "30M", "30M", rev1=True, rev2=False), P("h", 1, 11, 35, "5S25M", "25M5S"), P("i", 1, 1, 35, "5S25M", "5S25M"),
        P("j", 1, 11, 35, "30M", "19M2D11M"), toc.disagreement_fixture(6), toc.disagreement_fixture(1), P("k", 1, 11, 35, "3H27M", "30M2H"),
        P("l", 1, 11, 35, "10M3I17M", "30M"), P("m", 5, 5, 35, "40M", "2S38M"), P("n", 1, 60, 35, "30M", "30M"), [],
        [bamutil.frag("solo", toc.REF[:30], 35, "hi")],
    ]
    fam = []
    for i in range(5):
        fam += P(f"t{i}", 1, 11, 30 + i, "30M", "30M", mi="fam", extra=[("CB", "Z", "CELL1")])
    fam += P("odd", 1, 11, 35, "10M2D20M", "30M", mi="fam") + [bamutil.frag("solo", toc.REF[:30], 35, "fam")]
    groups.append(fam)
    deep = []
    for i in range(40):
        deep += P(f"d{i:03d}", 20, 40, 20 + (i % 20), "60M", "60M", mi="deep")
    groups.append(deep)
    return groups


@pytest.mark.parametrize("kw", [dict(), dict(track_rejects=True, produce_per_base_tags=True), dict(min_reads_per_strand=2, track_rejects=True),
                                dict(min_duplex_length=21, track_rejects=True), dict(max_duplex_disagreements=5, track_rejects=True),
                                dict(max_duplex_disagreement_rate=0.04), dict(single_strand_qual=4, outer_bases_qual=5, outer_bases_length=7),
                                dict(max_reads_per_strand=3, track_rejects=True, produce_per_base_tags=True), dict(max_reads_per_strand=0),
                                dict(cell_tag="CB", produce_per_base_tags=True), dict(tie_rule=1), dict(error_rate_pre_umi=30, error_rate_post_umi=25)])
def test_codec_crafted(kw):
    _same(GroupedReads.from_groups(crafted_groups()), **kw)


def test_codec_fgbio_saturation_fixtures():
    out, _ = _same(GroupedReads.from_groups([toc.codec_fixture(33000, "ACGT")]), produce_per_base_tags=True)
    p = bamutil.parse(split_records(out.data)[0])
    toc.check_tags(p, 65534, 65534, 0.0, 32767, 32767, 0.0, 32767, 32767, 0.0, [32767] * 4, [32767] * 4, [0] * 4, [0] * 4)
    out, _ = _same(GroupedReads.from_groups([toc.codec_fixture(73000, "ACGTACGT", "CCGTACGT", 33000)]), produce_per_base_tags=True)
    p = bamutil.parse(split_records(out.data)[0])
    toc.check_tags(p, 65534, 65534, 0.0625, 32767, 32767, 0.125, 32767, 32767, 0.0, [32767] * 8, [32767] * 8, [32767] + [0] * 7, [0] * 8)


@pytest.mark.parametrize("codec", [1, 0])
def test_codec_simulated_config5_shape(codec):
    """BASELINE.json configs[4] shape (2x300 bp, insert ~N(350,60), one MI per molecule) at a size the oracle finishes in seconds."""
    g = simulate_grouped_reads(1200, family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=codec)
    out, _ = _same(g, produce_per_base_tags=True, track_rejects=True)
    assert out.count > 1000
    if codec:
        # agreeing strands: the duplex stretch is called, not masked
        p = bamutil.parse(split_records(out.data)[0])
        assert p["seq"].count("N") < len(p["seq"]) // 10


def test_codec_simulated_150bp_and_counter_names():
    g = simulate_grouped_reads(800, family_size=2, read_length=150, insert_mean=200, insert_sd=30, codec=1)
    _same(g, max_reads_per_strand=1, min_duplex_length=30, track_rejects=True)


def test_codec_device_resident_matches_oracle():
    """Inputs generated in HBM, outputs left in HBM: nothing deferred on simulate-shaped molecules, bytes and the CODEC
    counters equal the oracle's."""
    if GENERAL_ONLY:
        pytest.skip("device-resident entry only")
    for kw, opt in ((dict(n_families=3000, family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1), dict(produce_per_base_tags=True)),
                    (dict(n_families=2000, family_size=1, read_length=150, insert_mean=200, insert_sd=40, codec=1), dict(cell_tag="CB")),
                    (dict(n_families=2000, family_size=6, read_length=150, insert_mean=220, insert_sd=60, codec=0, error_rate_ppm=20000),
                     dict(min_reads_per_strand=2, min_duplex_length=40, single_strand_qual=7, outer_bases_qual=9, outer_bases_length=4))):
        v = CodecConsensusOptions(**opt)
        c = CodecConsensusCaller("codec", "A", v)
        dg = c.simulate_on_device(**kw)
        out = c.process_batch_device(dg)
        data = out.to_host()
        st = c.codec_statistics()
        g = simulate_grouped_reads(kw["n_families"], **{k: x for k, x in kw.items() if k != "n_families"})
        o = fgx_opts.defaults(kind=2, read_name_prefix=b"codec", overlapping_consensus=0, cell_tag=(v.cell_tag.encode() if v.cell_tag else b"\0\0"),
                              produce_per_base_tags=int(v.produce_per_base_tags), codec_min_reads_per_strand=v.min_reads_per_strand,
                              codec_min_duplex_length=v.min_duplex_length, codec_has_single_strand_qual=int(v.single_strand_qual is not None),
                              codec_single_strand_qual=v.single_strand_qual or 0, codec_has_outer_bases_qual=int(v.outer_bases_qual is not None),
                              codec_outer_bases_qual=v.outer_bases_qual or 0, codec_outer_bases_length=v.outer_bases_length)
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=g.n_grp)
        assert out.n_deferred == 0
        assert out.count == want["count"] and data == want["data"]
        got = [st.total_input_reads, st.consensus_reads_generated, st.reads_filtered, st.consensus_bases_emitted, st.consensus_duplex_bases_emitted,
               st.duplex_disagreement_base_count, st.consensus_reads_rejected_hdd]
        ws = want["stats"]
        assert got == [int(ws[0]), int(ws[1]), int(ws[2]), int(ws[24]), int(ws[25]), int(ws[26]), int(ws[27])]
        c.close()


def test_long_read_name_prefix_takes_the_per_field_writer():
    """(round 6) k_emit_codec is launched only when the fast writer counted records it refuses (and the base counters are reduced again behind it)."""
    from fgumi_amd import simulate_grouped_reads
    _same(simulate_grouped_reads(200, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1), prefix="c" * 70)
