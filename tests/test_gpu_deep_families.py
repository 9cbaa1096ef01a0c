"""GPU parity: deep simplex families on the record / column split pipeline (round 4).  An end of more than 16 reads used to leave the split
pipeline (a disagreeing column travels to k_call_full as an item of 16 observations); now such a column takes ceil(reads / 16) consecutive
items (FULL_ITEM_CONT, fgumi_amd/csrc/fastpath.h) and every family of up to 64 records in the common shape stays — and a family of more
than 64 records goes straight to the workgroup-per-family kernel instead of walking the wavefront-per-family chain first.  Each test
compares the device-resident output with the oracle byte for byte and asserts the path taken."""
import ctypes as C

import numpy as np
import pytest

import fgx_opts
import orc
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, lib, simulate_grouped_reads

pytestmark = pytest.mark.gpu

for _f in ("fgx_debug_last_big_families", "fgx_debug_last_deep_families", "fgx_debug_last_routed", "fgx_debug_last_split_chunks"):
    getattr(lib, _f).restype = C.c_uint32
    getattr(lib, _f).argtypes = [C.c_void_p]


def _run(g, vo=None, okw=None):
    want = orc.process(fgx_opts.defaults(**dict(dict(min_reads=1), **(okw or {}))), g.blob, g.rec_off, g.rec_len, g.grp_first)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(**dict(dict(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), **(vo or {}))),
                                  overlapping_consensus=True)
    out = c.process_batch_device(g.to_device())
    got = out.to_host()
    path = dict(big=lib.fgx_debug_last_big_families(c._h), deep=lib.fgx_debug_last_deep_families(c._h), routed=lib.fgx_debug_last_routed(c._h), chunks=lib.fgx_debug_last_split_chunks(c._h),
                deferred=int(out.n_deferred))
    stats = np.array(c.last_stats_array, dtype=np.uint64)
    c.close()
    assert got == want["data"], f"output differs from the oracle ({len(got)} vs {len(want['data'])} bytes; path {path})"
    assert np.array_equal(stats, want["stats"])
    return path


def _records_per_family(g):
    return np.diff(np.asarray(g.grp_first, dtype=np.int64))


@pytest.mark.parametrize("pairs,err", [(20, 0), (24, 10000), (32, 30000), (17, 30000)])
def test_ends_of_17_to_32_reads_stay_in_the_split_pipeline(pairs, err):
    """Every family has `pairs` pairs (34 .. 64 records): nothing is routed down the chain, nothing deferred; with errors most columns of a
    family take the multi-item way (3 %: a column of 32 reads shows a second base with probability 0.62)."""
    g = simulate_grouped_reads(1500, family_size=pairs, error_rate_ppm=err)
    path = _run(g)
    # (routed: a family whose items outgrow the largest LDS slice — at 3 % errors and 32 pairs two columns in three take several items)
    assert path["chunks"] >= 1 and path["routed"] <= 30 and path["big"] == 0 and path["deferred"] == 0, path


def test_long_tail_sizes_split_takes_up_to_64_records_and_the_rest_goes_straight_to_the_workgroup_kernel():
    g = simulate_grouped_reads(6000, family_size=2, family_size_max=50, error_rate_ppm=10000)
    n = _records_per_family(g)
    path = _run(g)
    assert (n > 64).sum() > 50
    assert path["chunks"] >= 1 and path["big"] == path["deep"] == int((n > 64).sum()) and path["routed"] == 0 and path["deferred"] == 0, path


def test_long_tail_sizes_without_the_split_pipeline(monkeypatch):
    """FGX_SPLIT=0: the k_simplex_wave2 chain is the head; it, too, hands families of more than 64 records straight to the workgroup kernel."""
    monkeypatch.setenv("FGX_SPLIT", "0")
    g = simulate_grouped_reads(3000, family_size=2, family_size_max=50, error_rate_ppm=10000)
    n = _records_per_family(g)
    path = _run(g)
    assert path["chunks"] == 0 and path["big"] == path["deep"] == int((n > 64).sum()) > 20 and path["deferred"] == 0, path


def test_deep_ends_with_min_reads_and_quality_options():
    g = simulate_grouped_reads(1200, family_size=18, family_size_max=32, error_rate_ppm=20000)
    path = _run(g, dict(min_reads=3, min_input_base_quality=30), dict(min_reads=3, min_input_base_quality=30))
    assert path["routed"] == 0 and path["deferred"] == 0, path
    path = _run(g, dict(produce_per_base_tags=False), dict(produce_per_base_tags=0))
    assert path["routed"] == 0 and path["deferred"] == 0, path


def test_multi_item_columns_when_the_item_lists_are_tiny(monkeypatch):
    """A pool far too small (test knobs): lists of a few items.  A column of several consecutive items is never cut at the end of a list —
    the family's items move to a list that holds them all, the tail of the one that did not is padded — and when every list is full the
    batch runs again with more room; the result still equals the oracle."""
    monkeypatch.setenv("FGX_POOL_DIV", "4096")
    monkeypatch.setenv("FGX_POOL_SLACK", "1")
    g = simulate_grouped_reads(2500, family_size=20, family_size_max=32, error_rate_ppm=5000)
    path = _run(g)
    assert path["routed"] == 0, path
    monkeypatch.setenv("FGX_POOL_DIV", "64")
    monkeypatch.setenv("FGX_POOL_SLACK", "3")
    path = _run(simulate_grouped_reads(2500, family_size=20, family_size_max=32, error_rate_ppm=5000))
    assert path["routed"] == 0, path


def test_direct_records_keep_the_16_read_limit(monkeypatch):
    """FGX_DIRECT=1 (opt-in): families with a deeper end are not taken by the direct path's column kernel (its packed depth sums are 16 bits
    per end) — they come out of the chain below it and the merge; the bytes are the oracle's all the same."""
    monkeypatch.setenv("FGX_DIRECT", "1")
    g = simulate_grouped_reads(2000, family_size=4, family_size_max=30, error_rate_ppm=10000)
    path = _run(g)
    assert path["deferred"] == 0 and path["routed"] > 0, path


@pytest.mark.parametrize("kw", [dict(family_size=33, family_size_max=60, error_rate_ppm=10000), dict(family_size=40, family_size_max=150, error_rate_ppm=3000),
                                dict(family_size=70, family_size_max=127, error_rate_ppm=30000)])
def test_families_of_more_than_64_records_on_the_streaming_kernels(kw):
    """k_deep_parse + k_deep_cols (simplex_deep.inc): every family here has more than 64 records — up to 300, beyond what the workgroup kernel
    (128) ever took — and none leaves the device."""
    g = simulate_grouped_reads(400, **kw)
    n = _records_per_family(g)
    assert n.min() > 64
    path = _run(g)
    assert path["big"] == path["deep"] == len(n) and path["deferred"] == 0, path
    path = _run(g, dict(min_reads=5, min_input_base_quality=25), dict(min_reads=5, min_input_base_quality=25))
    assert path["deep"] == len(n) and path["deferred"] == 0, path


def test_deep_families_without_the_streaming_kernels(monkeypatch):
    """FGX_DEEP=0: the workgroup-per-family kernel takes the big list as in round 3 (up to 128 records; the rest is the host's)."""
    monkeypatch.setenv("FGX_DEEP", "0")
    g = simulate_grouped_reads(300, family_size=33, family_size_max=45, error_rate_ppm=10000)   # (up to 90 records: what fits that kernel's 64 KB of LDS)
    path = _run(g)
    assert path["deep"] == 0 and path["big"] == 300 and path["deferred"] == 0, path


def test_an_end_of_more_than_255_reads_leaves_the_streaming_kernels():
    """The per-chain observation counts of a k_call_full item are bytes: an end of more than 255 retained reads is not the streaming kernels'
    (k_family passes it on, the host entry finishes it on the general path); the bytes are the oracle's all the same."""
    g = simulate_grouped_reads(12, family_size=260, family_size_max=300, error_rate_ppm=5000)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    out = c.process_batch(g)
    assert out.data == want["data"]
    assert lib.fgx_debug_last_big_families(c._h) == 12 and lib.fgx_debug_last_deep_families(c._h) == 0
    c.close()


def _with_low_qualities(g, every=37, values=(0, 1, 2, 0, 5)):
    """The batch with every `every`-th quality byte of every record replaced by one of `values` (in place of the simulator's 30 - 40)."""
    import dataclasses
    blob = np.array(g.blob, copy=True)
    k = 0
    for o in np.asarray(g.rec_off, dtype=np.int64):          # (rec_off: the record's body, behind its block_size)
        l_name, n_cig, l_seq = int(blob[o + 8]), int(blob[o + 12]) | (int(blob[o + 13]) << 8), int(blob[o + 16:o + 20].view(np.uint32)[0])
        q0 = o + 32 + l_name + 4 * n_cig + (l_seq + 1) // 2
        for i in range(every - 1 - (k % 7), l_seq, every):      # (never 0xFF in the first byte: that would mean "no qualities")
            blob[q0 + i] = values[k % len(values)]
            k += 1
    assert k > 0
    return dataclasses.replace(g, blob=blob)


@pytest.mark.parametrize("min_bq", [0, 1, 3])
def test_quality_zero_observations_when_the_quality_floor_admits_them(min_bq):
    """`correct[0]` is ln 0 = -inf (phred 0: the base is wrong with certainty).  With --min-input-base-quality 0 such a base IS an
    observation: the reference's sums become -inf and call_full decides.  The split pipeline's f32 table holds a finite stand-in for
    that entry (its sums are built with multiply-adds: 0 x inf would poison every column that merely SEES a quality 0 below the floor);
    a column that adds it fails every gate and is recomputed exactly by k_call_full.  Floors 1 and 3: the same bytes are below the floor."""
    g = _with_low_qualities(simulate_grouped_reads(1500, family_size=2, family_size_max=12, error_rate_ppm=2000))
    path = _run(g, dict(min_input_base_quality=min_bq), dict(min_input_base_quality=min_bq))
    assert path["chunks"] >= 1 and path["deferred"] == 0, path


# ---- round 6: more shapes of families above 64 records (k_deep_parse's two-wavefront build, k_deep_cols) -----------------------------------------------
@pytest.mark.parametrize("sim,kw", [(dict(n_families=1500, family_size=35, family_size_max=120), {}),
                                    (dict(n_families=800, family_size=40, family_size_max=100, error_rate_ppm=20000, read_length=151, insert_mean=170, insert_sd=40), dict(min_reads=3)),
                                    (dict(n_families=600, family_size=70, family_size_max=127), dict(min_input_base_quality=30))])
def test_families_above_64_records_equal_the_oracle(sim, kw):
    g = simulate_grouped_reads(**sim)
    n = _records_per_family(g)
    path = _run(g, kw, kw)
    assert path["deep"] == int((n > 64).sum()) > 0 and path["deferred"] == 0, path
