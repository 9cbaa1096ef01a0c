"""GPU parity of `--rejects` for the simplex caller WITHOUT the general path (default since round 4, FGX_REJECTS_DEVICE=0 opts out;
fgumi_amd/csrc/reject_device.hip runs reject_core.h's decision function a lane per MI group beside the unchanged device pipeline): consensus
records, counters AND the rejects stream byte-identical to the oracle, through the host entry and through the device-resident entry, and the
deferral diagnostics show that the batch did NOT go to the general path."""
import ctypes as C
import random

import numpy as np
import pytest

import fgx_opts
import orc
from isolated import run_isolated

pytestmark = pytest.mark.gpu

KWS = [dict(min_reads=1), dict(min_reads=2, max_reads=3), dict(min_reads=3, overlapping_consensus=0, min_input_base_quality=30), dict(min_reads=2, trim=1, min_input_base_quality=25)]


def batch(seed):
    """Simulated families of mixed sizes (some below --min-reads), read-through pairs (the pre-correction changes the rejected bytes) and
    families whose reads disagree on the alignment (minority alignments are rejects)."""
    import test_canon_core as tc
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    rng = random.Random(seed)
    groups = []
    for sim in (simulate_grouped_reads(300, family_size=1, family_size_max=9, seed=seed), simulate_grouped_reads(150, family_size=4, read_length=151, insert_mean=120, insert_sd=30, seed=seed + 1)):
        groups += [sim.records(i) for i in range(sim.n_grp)]
    for g in range(120):
        m = tc.duplex_indel_molecule(rng, 9000 + g)
        if m:
            groups.append(m)
    rng.shuffle(groups)
    return GroupedReads.from_groups(groups)


def check_host_entry(kw, seed):
    from fgumi_amd._lib import Options, Output, lib
    g = batch(seed)
    o = fgx_opts.defaults(kind=0, track_rejects=1, **kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    assert want["n_rejects"] > 0
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        out = Output()
        rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert (C.string_at(out.data, out.data_len) if out.data_len else b"") == want["data"]
        assert int(out.count) == want["count"] and np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"])
        assert int(out.n_rejects) == want["n_rejects"]
        assert (C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"") == want["rejects"]
        # the path taken: with the side kernels the device pipeline decides the batch (before round 4 `--rejects` sent EVERY group to the
        # general path); FGX_REJECTS_DEVICE=0 restores that
        d = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d)
        import os
        if os.environ.get("FGX_REJECTS_DEVICE") == "0":
            assert int(d[0]) == g.n_grp, (int(d[0]), g.n_grp)
        elif os.environ.get("APIEMU_DEFER"):          # (tests/apiemu: the stand-in for the device pipeline defers groups by rule)
            assert int(d[0]) < g.n_grp
        else:
            assert int(d[0]) == 0, f"{int(d[0])} of {g.n_grp} groups were deferred to the general path"
    finally:
        lib.fgx_destroy(h)


def check_device_entry(kw, seed):
    import torch
    from fgumi_amd._lib import Options, Output, hip_memcpy_d2h, lib
    g = batch(seed)
    o = fgx_opts.defaults(kind=0, track_rejects=1, **kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    assert want["n_rejects"] > 0
    po = Options.from_buffer_copy(bytes(o))
    po.device = 0
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        dg = g.to_device(0)
        torch.cuda.synchronize()
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, dg.blob.data_ptr(), dg.blob_len, dg.rec_off.data_ptr(), dg.rec_len.data_ptr(), dg.n_rec, dg.grp_first.data_ptr(), dg.n_grp,
                                          C.byref(out), C.byref(nd), C.byref(dp))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert int(out.n_rejects) == want["n_rejects"]
        assert hip_memcpy_d2h(out.rejects, int(out.rejects_len)) == want["rejects"]     # (covers every group, the deferred ones included)
        assert nd.value == 0, f"{nd.value} groups deferred by the device entry"
        if nd.value == 0:
            assert hip_memcpy_d2h(out.data, int(out.data_len)) == want["data"] and int(out.count) == want["count"]
            assert np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"])
    finally:
        lib.fgx_destroy(h)


@pytest.mark.parametrize("kw", KWS)
def test_host_entry_rejects_come_from_the_side_kernels(kw):
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry", kw, 11)


def test_host_entry_rejects_opt_out_takes_the_general_path():
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry", KWS[1], 11, env={"FGX_REJECTS_DEVICE": "0"})


@pytest.mark.parametrize("kw", KWS[:2])
def test_device_entry_accepts_track_rejects(kw):
    run_isolated("test_gpu_zz_rejects_device", "check_device_entry", kw, 12)


# ---- duplex / CODEC callers (round 6): the rejects come from side kernels run AFTER the device pipeline — whether a molecule gave
# ---- its consensus is read from the pipeline's own output slots (reject_device.hip strand_rejects_device, reject_core.h duplex_reject_codes / codec_reject_mask)

STRAND_KWS = {"duplex": [dict(), dict(duplex_min_reads=(3, 2, 1)), dict(duplex_min_reads=(4, 2, 1), overlapping_consensus=0), dict(duplex_min_reads=(2, 1, 0), trim=1, min_input_base_quality=25)],
              "codec": [dict(), dict(codec_min_reads_per_strand=3), dict(codec_min_duplex_length=120)]}


def strand_options(kind, kw):
    kw = dict(kw)
    mr = kw.pop("duplex_min_reads", None)
    o = fgx_opts.defaults(kind=1 if kind == "duplex" else 2, track_rejects=1, **{k: v for k, v in kw.items() if not k.startswith("codec_")})
    for k, v in kw.items():
        if k.startswith("codec_"):
            setattr(o, k, v)
    if mr:
        o.duplex_min_reads = (C.c_uint32 * 3)(*mr)
    return o


def strand_batch(kind, seed, hostile=False):
    """`simulate`-shaped molecules of mixed depth (shallow ones fall below the min-reads settings; read-through inserts: the duplex pre-step changes the
    rejected bytes), some reads with their qualities pushed under --min-input-base-quality (zero length after masking: rejects of a KEPT molecule);
    hostile = molecules whose reads disagree on the alignment, and fragment reads (molecules the device pipeline defers: the batch falls back)."""
    import test_canon_core as tc
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    rng = random.Random(seed)
    groups = []
    if kind == "duplex":
        sims = [simulate_grouped_reads(250, family_size=1, family_size_max=6, duplex=1, seed=seed), simulate_grouped_reads(150, family_size=4, duplex=1, read_length=151, insert_mean=120, insert_sd=30, seed=seed + 1, first_family=100000)]
    else:
        sims = [simulate_grouped_reads(250, family_size=1, family_size_max=5, read_length=150, insert_mean=200, insert_sd=40, codec=1, seed=seed),
                simulate_grouped_reads(100, family_size=3, read_length=150, insert_mean=260, insert_sd=30, codec=1, seed=seed + 1, first_family=100000)]      # (first_family: MI values of their own — in a FILE two neighbours with one MI are one group)
    for sim in sims:
        for i in range(sim.n_grp):
            m = sim.records(i)
            if kind == "duplex" and len(m) >= 6 and rng.random() < 0.3:          # one read's qualities all 2
                k = rng.randrange(len(m))
                r = bytearray(m[k])
                l_name, n_cig, l_seq = r[8], int.from_bytes(r[12:14], "little"), int.from_bytes(r[16:20], "little")
                q0 = 32 + l_name + 4 * n_cig + (l_seq + 1) // 2
                r[q0:q0 + l_seq] = bytes([2]) * l_seq
                m = m[:k] + [bytes(r)] + m[k + 1:]
            groups.append(m)
    if hostile:
        for g in range(60):
            m = tc.duplex_indel_molecule(rng, 9000 + g)
            if m:
                groups.append(m)
    rng.shuffle(groups)
    return GroupedReads.from_groups(groups)


def check_host_entry_strand(kind, kw, seed, hostile=False):
    from fgumi_amd._lib import Options, Output, lib
    import os
    g = strand_batch(kind, seed, hostile)
    o = strand_options(kind, kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100000)
    assert want["n_rejects"] > 0 and want["count"] > 0
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        out = Output()
        rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert (C.string_at(out.data, out.data_len) if out.data_len else b"") == want["data"]
        assert int(out.count) == want["count"] and np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"])
        assert int(out.n_rejects) == want["n_rejects"]
        assert (C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"") == want["rejects"]
        d = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d)
        if os.environ.get("FGX_REJECTS_DEVICE") == "0" or hostile or os.environ.get("APIEMU_DEFER") not in (None, "none", "indel"):
            pass                                             # (the general path decided the batch, or part of it: same answer)
        else:
            assert int(d[0]) == 0, f"{int(d[0])} of {g.n_grp} molecules were deferred to the general path"
    finally:
        lib.fgx_destroy(h)


def check_device_entry_strand(kind, kw, seed, on_gpu=True):
    from fgumi_amd._lib import Options, Output, lib
    g = strand_batch(kind, seed)
    o = strand_options(kind, kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=100000)
    assert want["n_rejects"] > 0 and want["count"] > 0
    po = Options.from_buffer_copy(bytes(o))
    po.device = 0 if on_gpu else -1
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        if on_gpu:
            import torch
            from fgumi_amd._lib import hip_memcpy_d2h
            dg = g.to_device(0)
            torch.cuda.synchronize()
            args = (dg.blob.data_ptr(), dg.blob_len, dg.rec_off.data_ptr(), dg.rec_len.data_ptr(), dg.n_rec, dg.grp_first.data_ptr(), dg.n_grp)
            fetch = lambda p, n: hip_memcpy_d2h(p, int(n)) if n else b""
        else:                                                # tests/apiemu: device memory is host memory
            blob = np.concatenate([g.blob, np.zeros(16, dtype=np.uint8)])
            args = (blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp)
            fetch = lambda p, n: C.string_at(p, int(n)) if n else b""
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, *args, C.byref(out), C.byref(nd), C.byref(dp))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert nd.value == 0, f"{nd.value} molecules deferred by the device entry"
        assert int(out.n_rejects) == want["n_rejects"]
        assert fetch(out.rejects, out.rejects_len) == want["rejects"]
        assert fetch(out.data, out.data_len) == want["data"] and int(out.count) == want["count"]
        assert np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"])
    finally:
        lib.fgx_destroy(h)


@pytest.mark.parametrize("kind,kw", [(k, kw) for k in ("duplex", "codec") for kw in STRAND_KWS[k]])
def test_duplex_codec_host_entry_rejects_come_from_the_side_kernels(kind, kw):
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry_strand", kind, kw, 21)


def test_duplex_rejects_of_a_batch_with_deferred_molecules_fall_back_whole():
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry_strand", "duplex", STRAND_KWS["duplex"][1], 22, True)


@pytest.mark.parametrize("kind", ["duplex", "codec"])
def test_duplex_codec_device_entry_accepts_track_rejects(kind):
    run_isolated("test_gpu_zz_rejects_device", "check_device_entry_strand", kind, STRAND_KWS[kind][1], 23)


def test_duplex_codec_rejects_opt_out_takes_the_general_path():
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry_strand", "duplex", STRAND_KWS["duplex"][1], 21, env={"FGX_REJECTS_DEVICE": "0"})
