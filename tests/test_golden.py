"""Frozen fixtures under tests/golden/ (see make_golden.py): the oracle must keep reproducing them (CPU), and the HIP paths —
device pipeline and general path — must reproduce them on the GPU without consulting a freshly built oracle."""
import json
import os

import numpy as np
import pytest

import fgx_opts
from fgumi_amd import GroupedReads

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["simplex_crafted", "simplex_sim_depth3", "duplex_sim", "duplex_fgbio_fixture", "codec_crafted", "codec_sim", "simplex_indels", "simplex_indels_max_reads"]
FILTER_NAMES = ["filter_crafted_default", "filter_crafted_duplex_tiers", "filter_crafted_single_read"]


def _load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return GroupedReads(z["blob"], z["rec_off"], z["rec_len"], z["grp_first"]), bytes(z["data"]), int(z["count"]), z["stats"]


def _options(name):
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    for n, _, o, batch in make_golden.inputs():
        if n == name:
            return o, batch
    raise KeyError(name)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    import orc
    g, data, count, stats = _load(name)
    o, batch = _options(name)
    res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch)
    assert res["count"] == count and res["data"] == data
    assert np.array_equal(res["stats"], stats)


def test_reference_pins_file_is_consistent():
    pins = json.load(open(os.path.join(GOLD, "reference_pins.json")))
    assert all("source" in p and "expect" in p for p in pins["pins"])


@pytest.mark.gpu
@pytest.mark.parametrize("general_only", [False, True])
@pytest.mark.parametrize("name", NAMES)
def test_hip_paths_reproduce_golden(name, general_only):
    import ctypes as C
    from fgumi_amd import lib
    from fgumi_amd._lib import Options, Output
    g, data, count, stats = _load(name)
    o, _ = _options(name)
    h = lib.fgx_create(C.cast(C.pointer(o), C.POINTER(Options)))     # tests/fgx_opts.Options has the same layout
    assert h, lib.fgx_global_error()
    lib.fgx_set_general_only(h, int(general_only))
    out = Output()
    rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data,
                               g.n_grp, C.byref(out))
    assert rc == 0, lib.fgx_last_error(h)
    got = C.string_at(out.data, out.data_len) if out.data_len else b""
    assert out.count == count and got == data
    assert [int(v) for v in out.stats] == [int(v) for v in stats]
    lib.fgx_destroy(h)


def _filter_case(name):
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    for n, _, kw in make_golden.filter_inputs():
        if n == name:
            z = np.load(os.path.join(GOLD, name + ".npz"))
            return z, kw
    raise KeyError(name)


@pytest.mark.parametrize("name", FILTER_NAMES)
def test_oracle_filter_reproduces_golden(name):
    import orc
    z, kw = _filter_case(name)
    res = orc.filter_records(orc.filter_options(**kw), z["blob"], z["rec_off"], z["rec_len"])
    assert res["data"] == bytes(z["data"]) and res["rejects"] == bytes(z["rejects"])
    assert [res["records"], res["passed"], res["masked"], res["rejected"]] == [int(v) for v in z["counts"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", FILTER_NAMES)
def test_hip_filter_reproduces_golden(name):
    import test_gpu_filter as tgf
    from fgumi_amd import ConsensusFilter
    z, kw = _filter_case(name)
    f = ConsensusFilter(tgf._cfg(kw), **tgf._flags(kw))
    got = f.filter_stream(z["blob"], z["rec_off"], z["rec_len"])
    f.close()
    assert got.data == bytes(z["data"]) and got.rejects == bytes(z["rejects"])
    assert [got.records_count, got.passed_count, got.bases_masked, got.rejected_count] == [int(v) for v in z["counts"]]
