"""CPU-side checks of the product library (no GPU): it loads, exports the C ABI, its host-side
table builders and glibc-compatible libm agree bit for bit with the oracle / the box's libm,
and the synthetic generator is well-formed and deterministic."""
import ctypes as C
import math
import os
import re
import struct

import numpy as np
import pytest

import fgx_opts
import orc
from fgumi_amd import _lib, simulate_grouped_reads, split_records
from fgumi_amd._lib import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "fgumi_amd.h")).read()
    declared = set(re.findall(r"\b(fgx_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"fgx_options", "fgx_output", "fgx_caller", "fgx_sim_params", "fgx_group_options", "fgx_filter_options", "fgx_filter_output"}
    assert declared, "header parse failed"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/fgumi_amd.h but not exported"
    assert set(_lib.EXPORTS) <= declared | set(_lib.EXPORTS)


def test_options_struct_layout_matches_header():
    o = _lib.default_options()
    assert o.struct_size == C.sizeof(_lib.Options) == C.sizeof(fgx_opts.Options)
    assert o.tag == b"MI" and o.cell_tag == b"CB"
    assert (o.error_rate_pre_umi, o.error_rate_post_umi, o.min_input_base_quality, o.min_consensus_base_quality) == (45, 40, 10, 2)
    assert o.overlapping_consensus == 1 and o.max_reads == -1 and o.read_group_id == b"A"


def test_filter_options_layout_defaults_and_host_mirror():
    """fgx_filter_options: same layout in the product mirror and the oracle's, CLI defaults (src/lib/commands/filter.rs:118-170), and the
    FilterConfig host logic: 1-3 values expand from the last (filter.rs:20-27), ordering violations are errors (filter.rs:284-318)."""
    import struct as st
    from fgumi_amd import FilterConfig, FilterThresholds, record_offsets
    o = _lib.FilterOptions()
    lib.fgx_filter_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_lib.FilterOptions) == C.sizeof(orc.FilterOptions) == 112
    assert [f[0] for f in _lib.FilterOptions._fields_] == [f[0] for f in orc.FilterOptions._fields_]
    assert _lib.FilterOptions.min_methylation_depth.offset == 92 and _lib.FilterOptions.min_conversion_fraction.offset == 104
    assert not (o.has_min_methylation_depth or o.require_strand_methylation_agreement or o.has_min_conversion_fraction or o.methylation_mode)   # filter.rs:181-206
    assert list(o.min_reads) == [1, 1, 1] and list(o.max_read_error_rate) == [0.025] * 3 and list(o.max_base_error_rate) == [0.1] * 3
    assert o.max_no_call_fraction == 0.2 and o.filter_by_template == 1 and not o.has_min_base_quality and not o.track_rejects
    c = FilterConfig.new([5, 3], [0.05], [0.2, 0.1, 0.3], min_base_quality=20)
    assert (c.duplex.min_reads, c.ab.min_reads, c.ba.min_reads) == (5, 3, 3) and (c.ab.max_base_error_rate, c.ba.max_base_error_rate) == (0.1, 0.3)
    for bad in (dict(min_reads=[1, 2]), dict(min_reads=[3, 2, 3]), dict(min_reads=[3], max_read_error_rate=[0.1, 0.2, 0.1]),
                dict(min_reads=[3], max_base_error_rate=[0.1, 0.3, 0.2]), dict(min_reads=[3], max_no_call_fraction=2.5),
                dict(min_reads=[3], max_read_error_rate=[1.5]), dict(min_reads=[1, 1, 1, 1]), dict(min_reads=[])):
        with pytest.raises(ValueError):
            FilterConfig.new(**bad)
    assert FilterConfig.for_duplex(FilterThresholds(4), FilterThresholds(2)).ba.min_reads == 2
    data = st.pack("<I", 3) + b"abc" + st.pack("<I", 0) + st.pack("<I", 2) + b"zz"
    off, ln = record_offsets(data)
    assert off.tolist() == [4, 11, 15] and ln.tolist() == [3, 0, 2]
    with pytest.raises(ValueError):
        record_offsets(data[:-1])


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    o = _lib.default_options()
    h = lib.fgx_create(C.byref(o))
    assert not h
    assert b"no CPU fallback" in lib.fgx_global_error() or b"HIP" in lib.fgx_global_error()


@pytest.mark.parametrize("op,fn,lo,hi", [(0, math.exp, -40.0, 20.0), (0, math.exp, -745.0, -700.0), (1, math.log, 1e-9, 10.0),
                                         (1, math.log, 0.93, 1.07), (2, math.log1p, -0.9999, 2.0), (3, math.expm1, -0.7, 0.7),
                                         (3, math.expm1, -40.0, 5.0)])
def test_glibc_port_matches_box_libm(op, fn, lo, hi):
    rng = np.random.default_rng(op * 7 + 1)
    x = rng.uniform(lo, hi, 200000)
    y = np.zeros_like(x)
    lib.fgx_host_libm_array(op, x.ctypes.data, y.ctypes.data, x.size)
    ref = np.array([fn(v) for v in x])
    assert np.array_equal(y.view(np.uint64), ref.view(np.uint64))


def test_libm_self_check_of_fgx_create_agrees_on_this_box():
    # what fgx_create runs before handing out a caller: port vs the process's own libm (this image: glibc 2.35)
    msg = C.create_string_buffer(512)
    rc = lib.fgx_libm_self_check(msg, 512)
    assert rc == 0 and msg.value == b"", msg.value


@pytest.mark.parametrize("pre,post", [(45, 40), (90, 90), (93, 93), (2, 2), (20, 10), (70, 5), (0, 0), (45, 255)])
def test_host_table_builders_match_oracle_bitwise(pre, post):
    b = orc.Builder(pre, post)
    for which in range(4):
        want, cap = b.table(which)
        got = np.zeros(94)
        gcap = C.c_uint32()
        single = np.zeros(94, dtype=np.uint8)
        lib.fgx_build_tables_host(pre, post, which, got.ctypes.data, C.byref(gcap), single.ctypes.data)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), f"table {which}"
        assert gcap.value == cap
    o = fgx_opts.defaults(error_rate_pre_umi=pre, error_rate_post_umi=post)
    want = np.zeros(94, dtype=np.uint8)
    orc.lib.orc_single_input_quals(C.addressof(o), orc.ptr(want))
    assert np.array_equal(single, want)


def test_simulated_reads_shape_and_determinism():
    g = simulate_grouped_reads(50, family_size=3)
    g2 = simulate_grouped_reads(50, family_size=3)
    assert np.array_equal(g.blob, g2.blob) and g.n_rec == 300 and g.n_grp == 50
    recs = g.records(7)
    assert len(recs) == 6
    names = set()
    for i, r in enumerate(recs):
        l_name, n_cig, flag, l_seq = r[8], struct.unpack_from("<H", r, 12)[0], struct.unpack_from("<H", r, 14)[0], struct.unpack_from("<I", r, 16)[0]
        name = r[32:32 + l_name - 1].decode()
        names.add(name)
        assert re.fullmatch(r"mol\d{8}_read\d{4}", name) and n_cig == 1 and l_seq == 150 and r[9] == 60
        assert flag & 0x3 == 0x3 and bool(flag & 0x40) != bool(flag & 0x80) and bool(flag & 0x10) != bool(flag & 0x20)
        cig = struct.unpack_from("<I", r, 32 + l_name)[0]
        assert cig == (150 << 4)
        aux = r[32 + l_name + 4 + 75 + 150:]
        assert aux[:3] == b"RXZ" and aux[12:15] == b"MIZ" and b"MCZ150M\0" in aux and aux.endswith(b"MQc<")
        q = np.frombuffer(r, dtype=np.uint8, count=150, offset=32 + l_name + 4 + 75)
        assert q.min() >= 2 and q.max() <= 41
    assert len(names) == 3
    # block_size prefixes make the blob a legal BAM record stream
    p = 0
    for r in range(g.n_rec):
        assert int(g.rec_off[r]) == p + 4 and int.from_bytes(bytes(g.blob[p:p + 4]), "little") == int(g.rec_len[r])
        p += 4 + int(g.rec_len[r])
    assert p == g.blob.size
    # a different shard start yields the same molecules
    gs = simulate_grouped_reads(10, family_size=3, first_family=40)
    assert gs.records(0) == g.records(40)


def test_oracle_runs_on_simulated_reads():
    g = simulate_grouped_reads(200, family_size=3)
    o = fgx_opts.defaults(min_reads=1)
    res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first)
    assert res["count"] == 400 and res["stats"][0] == 1200 and res["stats"][1] == 400
    recs = split_records(res["data"])
    assert len(recs) == 400
    r = recs[0]
    assert r[:8] == b"\xff" * 8 and r[32:34] == b":0" and struct.unpack_from("<H", r, 14)[0] == 0x4D
    # multi-threaded batches of 50 concatenate to the same bytes
    res2 = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50, threads=4)
    assert res2["data"] == res["data"] and np.array_equal(res2["stats"], res["stats"])


def test_oracle_handles_crafted_edge_cases():
    import bamutil
    import cases
    from fgumi_amd import GroupedReads
    g = GroupedReads.from_groups(cases.crafted_groups())
    for mr in (1, 2):
        for ov in (0, 1):
            o = fgx_opts.defaults(min_reads=mr, overlapping_consensus=ov, track_rejects=1, read_name_prefix=b"lib1")
            res = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first)
            recs = [bamutil.parse(r) for r in split_records(res["data"])]
            assert res["count"] == len(recs) > 5
            assert all(r["name"].startswith("lib1:") and r["flag"] & 0x4 and r["tag_order"][:4] == ["RG", "cD", "cM", "cE"] for r in recs)
            assert res["stats"][0] == g.n_rec
            # every input read is either used, filtered, or part of a consensus: rejects are whole records
            assert res["n_rejects"] == len(split_records(res["rejects"]))
    # GATTACA ×3 with one disagreement (vanilla_caller.rs:3100-3139): consensus GATTACA, lower qual at the conflict
    o = fgx_opts.defaults(min_reads=1, min_consensus_base_quality=0, error_rate_pre_umi=93)
    first = GroupedReads.from_groups([cases.crafted_groups()[0]])
    r = bamutil.parse(split_records(orc.process(o, first.blob, first.rec_off, first.rec_len, first.grp_first)["data"])[0])
    assert r["seq"] == "GATTACA" and r["quals"][4] < r["quals"][0] and r["tags"]["cD"][1] == 3 and r["tags"]["ce"][1][4] == 1
