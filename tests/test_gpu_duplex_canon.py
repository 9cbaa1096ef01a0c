"""GPU parity of the canonical second pass (default since round 4; FGX_DUPLEX_CANON=0 opts out; fgumi_amd/csrc/canon_core.h + api.cpp): duplex molecules with
indel / skip / pad CIGARs, which the device pipeline defers, are rewritten into their canonical form and decided by the device pipeline
in a second pass instead of by the general path — byte-identical to the oracle, and the diagnostics show the second pass took them."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import bamutil
import fgx_opts
import orc
import test_canon_core as tc
from fgumi_amd import GroupedReads, simulate_grouped_reads, split_records
from fgumi_amd._lib import Options, Output, lib

pytestmark = pytest.mark.gpu


def product(o, g):
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        out = Output()
        rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
        assert rc == 0, lib.fgx_last_error(h).decode()
        d = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d)
        return dict(data=C.string_at(out.data, out.data_len) if out.data_len else b"", count=int(out.count), stats=np.array(list(out.stats), dtype=np.uint64),
                    deferred=int(d[0]), canon=int(d[1]))
    finally:
        lib.fgx_destroy(h)


@pytest.fixture
def canon_on():
    old = os.environ.get("FGX_DUPLEX_CANON")
    os.environ["FGX_DUPLEX_CANON"] = "1"
    yield
    if old is None:
        os.environ.pop("FGX_DUPLEX_CANON", None)
    else:
        os.environ["FGX_DUPLEX_CANON"] = old


@pytest.mark.parametrize("kw,mr", [(dict(overlapping_consensus=1), (1, 1, 0)), (dict(overlapping_consensus=0, min_input_base_quality=20), (2, 1, 1)),
                                   (dict(overlapping_consensus=1, cell_tag=b"\0\0", produce_per_base_tags=0), (1, 1, 1))])
def test_indel_molecules_take_the_canonical_second_pass(canon_on, kw, mr):
    rng = random.Random(31)
    groups = []
    sim = simulate_grouped_reads(120, family_size=4, duplex=1)          # regular molecules between the indel ones: decided by the first pass
    for g in range(360):
        if g % 3 == 0:
            groups.append(sim.records(g // 3))
        else:
            m = tc.duplex_indel_molecule(rng, 1000 + g)
            if m:
                groups.append(m)
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=1, **kw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    want = orc.process(o, gr.blob, gr.rec_off, gr.rec_len, gr.grp_first, batch_groups=100)
    got = product(o, gr)
    assert got["deferred"] > 100 and got["canon"] > 0.6 * got["deferred"], (got["deferred"], got["canon"])
    if got["data"] != want["data"]:
        for i, (a, b) in enumerate(zip(split_records(got["data"]), split_records(want["data"]))):
            if a != b:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(a)}\nwant {bamutil.parse(b)}")
        raise AssertionError("record count / length differs")
    assert got["count"] == want["count"] and np.array_equal(got["stats"], want["stats"]), (got["stats"].tolist(), want["stats"].tolist())


def check_second_pass_can_be_switched_off():
    """FGX_DUPLEX_CANON=0: the round-3 behaviour — every deferred molecule takes the general path — and the same bytes."""
    os.environ["FGX_DUPLEX_CANON"] = "0"
    rng = random.Random(32)
    groups = [m for m in (tc.duplex_indel_molecule(rng, g) for g in range(80)) if m]
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=1)
    want = orc.process(o, gr.blob, gr.rec_off, gr.rec_len, gr.grp_first, batch_groups=100)
    try:
        got = product(o, gr)
        if os.environ.get("FGX_OPT_IN_ALL") == "0":
            os.environ["FGX_DUPLEX_CANON"] = "1"     # (a run with every path switched off: its own switch wins)
        else:
            del os.environ["FGX_DUPLEX_CANON"]       # the default: the second pass takes them
        again = product(o, gr)
    finally:
        os.environ.pop("FGX_DUPLEX_CANON", None)
    assert got["deferred"] > 0 and got["canon"] == 0 and got["data"] == want["data"] and np.array_equal(got["stats"], want["stats"])
    assert again["canon"] > 0.6 * again["deferred"] and again["data"] == want["data"] and np.array_equal(again["stats"], want["stats"])


def test_second_pass_can_be_switched_off():
    check_second_pass_can_be_switched_off()
