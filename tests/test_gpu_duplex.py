"""GPU parity: duplex caller (general path: host orchestration, single-strand columns on the device) vs the
oracle — byte-identical records, stats and rejects."""
import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
from fgumi_amd import DuplexConsensusCaller, GroupedReads, simulate_grouped_reads, split_records
from test_oracle_duplex import duplex_fixture

pytestmark = pytest.mark.gpu

GENERAL_ONLY = False


@pytest.fixture(autouse=True, params=["device", "general"])
def path_mode(request):
    """Every case runs through the device-resident duplex pipeline (deferred molecules fall back to the general path)
    and through the general host-orchestrated path alone."""
    global GENERAL_ONLY
    GENERAL_ONLY = request.param == "general"
    yield
    GENERAL_ONLY = False


def _same(g, min_reads=(1,), overlapping=True, track_rejects=False, cell_tag="CB", prefix="", **kw):
    mr = list(min_reads)
    total, xy, yx = mr[0], (mr[1] if len(mr) > 1 else mr[-1]), (mr[2] if len(mr) > 2 else mr[-1])
    okw = dict(overlapping_consensus=int(overlapping), track_rejects=int(track_rejects), read_name_prefix=prefix.encode(),
               cell_tag=(cell_tag.encode() if cell_tag else b"\0\0"))
    for k, v in kw.items():
        if k == "max_reads_per_strand":
            okw["duplex_max_reads_per_strand"] = -1 if v is None else v
        elif k in ("trim", "produce_per_base_tags"):
            okw[k] = int(v)
        else:
            okw[k] = v
    o = fgx_opts.defaults(kind=1, **okw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = total, xy, yx
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)
    c = DuplexConsensusCaller(prefix, "A", mr, cell_tag=cell_tag, track_rejects=track_rejects, overlapping_consensus=overlapping, **kw)
    c.set_general_only(GENERAL_ONLY)
    out = c.process_batch(g)
    st = c.last_batch_statistics()
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
    for r, v in st.rejection_reasons.items():
        arr[3 + int(r)] = v
    arr[24:28] = [st.overlapping[k] for k in ("overlapping_bases", "bases_agreeing", "bases_disagreeing", "bases_corrected")]
    assert np.array_equal(arr, want["stats"]), (arr.tolist(), want["stats"].tolist())
    if track_rejects:
        assert rej == split_records(want["rejects"])
    return out


def test_reference_duplex_unit_test_inputs():
    """The inputs of the reference's own process-level duplex unit tests (duplex_caller.rs `mod tests`; their assertions are replayed
    on the oracle in tests/test_oracle_duplex_pins.py) through the HIP path: byte-identical records, statistics and rejects —
    minority alignments, zero-length-after-trimming reads, whole-group rejections, stray fragments, lone strands, absent UMI halves."""
    import test_oracle_duplex_pins as pins
    cases = pins.replay_cases()
    assert len(cases) >= 15
    for kw, groups in cases:
        _same(GroupedReads.from_groups(groups), min_reads=kw["duplex_min_reads"], overlapping=False, track_rejects=bool(kw["track_rejects"]),
              cell_tag=(kw["cell_tag"].decode() if kw["cell_tag"] != b"\0\0" else None), prefix="consensus", trim=bool(kw["trim"]),
              produce_per_base_tags=bool(kw["produce_per_base_tags"]), min_input_base_quality=kw["min_input_base_quality"],
              error_rate_pre_umi=kw["error_rate_pre_umi"], error_rate_post_umi=kw["error_rate_post_umi"])


def test_duplex_simulated_config3_shape():
    """BASELINE.json configs[2] shape (duplex A/B, 6+6 pairs, 150 bp) at a size the oracle finishes in seconds."""
    out = _same(simulate_grouped_reads(1500, family_size=12, duplex=1), min_reads=(1,))
    assert out.count == 3000


@pytest.mark.parametrize("kw", [dict(min_reads=(3, 2, 1)), dict(min_reads=(2, 1, 0)), dict(min_reads=(1, 1, 0), track_rejects=True),
                                dict(min_reads=(6, 3, 3), track_rejects=True), dict(overlapping=False), dict(trim=True),
                                dict(max_reads_per_strand=2), dict(produce_per_base_tags=False), dict(min_input_base_quality=30, track_rejects=True),
                                dict(error_rate_pre_umi=30, error_rate_post_umi=25), dict(tie_rule=1)])
def test_duplex_option_matrix(kw):
    _same(simulate_grouped_reads(400, family_size=5, duplex=1, error_rate_ppm=20000), **kw)


def test_duplex_fgbio_fixtures_and_edge_cases():
    _same(duplex_fixture(3, 2, "ACGTACGT", "CCGTACGT", 1), cell_tag=None, prefix="duplex", overlapping=False)
    _same(duplex_fixture(1, 1, "ACGTACGT"), cell_tag=None, prefix="duplex", overlapping=False)
    _same(duplex_fixture(2, 0, "ACGTACGT"), cell_tag=None, prefix="duplex", overlapping=False, track_rejects=True)
    _same(duplex_fixture(2, 0, "ACGTACGT"), min_reads=(1, 1, 0), cell_tag=None, prefix="duplex", overlapping=True)
    _same(duplex_fixture(0, 3, "ACGTACGT"), min_reads=(1, 1, 0), cell_tag=None, prefix="duplex")
    _same(duplex_fixture(2000, 900, "ACGTACGTAC", "CCGTACGTAC", 700), cell_tag=None, overlapping=False)   # deep family
    frag = bamutil.frag("f", "ACGTACGT", 30, "mol/A")
    coll = list(bamutil.pair2("a0", "ACGTACGT", 40, "ACGTACGT", 40, "m2/A", 100, 200)) + list(bamutil.pair2("b0", "ACGTACGT", 40, "ACGTACGT", 40, "m2/B", 100, 200, rev1=False, rev2=True))
    g = GroupedReads.from_groups([[frag] + list(bamutil.pair2("a0", "ACGTACGT", 40, "ACGTTCGT", 35, "mol/A", 100, 104, rx="AAA-CCC", extra=[("CB", "Z", "cell9")])) +
                                  list(bamutil.pair2("b0", "ACGTACGT", 38, "ACGTACGT", 40, "mol/B", 100, 104, rev1=True, rev2=False, rx="CCC-AAA")), coll,
                                  list(bamutil.pair2("x0", "ACGTACGTAA", 40, "ACGTACGTAA", 40, "m3/A", 100, 300, cigar1="4M2I4M")) +
                                  list(bamutil.pair2("x1", "ACGTACGTAA", 40, "ACGTACGTAA", 40, "m3/A", 100, 300)) +
                                  list(bamutil.pair2("x2", "ACGTACGTAA", 40, "ACGTACGTAA", 40, "m3/A", 100, 300)) +
                                  list(bamutil.pair2("y0", "ACGTACGTAA", 40, "ACGTACGTAA", 40, "m3/B", 100, 300, rev1=True, rev2=False))])
    for tr in (False, True):
        _same(g, min_reads=(1,), track_rejects=tr)
        _same(g, min_reads=(2, 1, 1), track_rejects=tr, overlapping=False)


def test_duplex_fatal_errors():
    c = DuplexConsensusCaller("", "A", [1])
    with pytest.raises(RuntimeError, match="suffix"):
        c.consensus_reads(list(bamutil.pair2("a0", "ACGT", 40, "ACGT", 40, "mol", 100, 200)))
    with pytest.raises(RuntimeError, match="missing MI"):
        c.consensus_reads([bamutil.make_record("x", "ACGT", [30] * 4, flag=0x41)])
    c.close()
    with pytest.raises(ValueError):
        DuplexConsensusCaller("", "A", [1, 2])
    with pytest.raises(ValueError):
        DuplexConsensusCaller("", "A", [])


def test_duplex_device_resident_matches_oracle():
    """Inputs generated in HBM, outputs left in HBM: nothing deferred on simulate-shaped molecules, bytes equal the oracle's."""
    if GENERAL_ONLY:
        pytest.skip("device-resident entry only")
    for kw, mr in ((dict(n_families=3000, family_size=12, duplex=1), (1, 1, 1)), (dict(n_families=2000, family_size=5, duplex=1, error_rate_ppm=20000), (2, 1, 0)),
                   (dict(n_families=1500, family_size=3, duplex=1, read_length=100, insert_mean=150, insert_sd=40), (1, 1, 0))):
        c = DuplexConsensusCaller("", "A", list(mr), cell_tag="CB", overlapping_consensus=True)
        dg = c.simulate_on_device(**kw)
        out = c.process_batch_device(dg)
        data = out.to_host()
        g = simulate_grouped_reads(kw["n_families"], **{k: v for k, v in kw.items() if k != "n_families"})
        o = fgx_opts.defaults(kind=1, overlapping_consensus=1, cell_tag=b"CB")
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
        want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100)
        assert out.n_deferred == 0
        assert out.count == want["count"] and data == want["data"]
        c.close()


def test_long_read_name_prefix_takes_the_per_field_writer():
    """(round 6) k_emit_duplex — the per-field record writer — is launched only when the fast writer counted records it refuses: a 70-character
    read-name prefix makes it refuse every record."""
    from fgumi_amd import simulate_grouped_reads
    _same(simulate_grouped_reads(200, family_size=6, duplex=1), prefix="d" * 70)
