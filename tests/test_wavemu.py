"""The device-resident pipeline's WAVEFRONT kernels, executed on the CPU in 64-lane lock-step (the wave-level host emulator).

tests/wavemu compiles fgumi_amd/csrc/fastpath.hip — FastPath::run_once and every kernel it launches (k_col_bound, k_split_parse, k_split_cols in
its builds, k_split_finish, k_simplex_seg, k_simplex_wave2, k_family_wave, k_deep_*, k_family, k_call_full, k_emit*) — for the host under
tests/wavemu/simt.h: a fiber per thread, a rendezvous per cross-lane operation (__ballot, __shfl*, readlane, DPP, ds_bpermute, wave barriers,
__syncthreads) evaluated with the hardware's semantics, LDS as arrays; everything else of the product's host side is tests/apiemu's (api.cpp
unmodified on a fake HIP runtime).  `fgx_process_batch_device` on host arrays then runs the product's real launch chain and real kernel sources,
and its bytes and counters are compared with the oracle — child interpreters: FGX_LIB is read when fgumi_amd is imported."""
import ctypes as C

import numpy as np
import pytest

import fgx_opts
import orc
from isolated import run_isolated


def env(**flags):
    import wavemu
    e = {"FGX_LIB": wavemu.build(), "FGX_ALLOW_LIBM_MISMATCH": "1"}
    e.update({k: str(v) for k, v in flags.items()})
    return e


def check_device_entry(kind, n_families, sim, opts, want_path=None):
    """One batch through fgx_process_batch_device of the emulation library, host arrays standing in for the tensors in HBM, against the oracle."""
    from fgumi_amd import simulate_grouped_reads
    from fgumi_amd._lib import Options, Output, lib
    g = simulate_grouped_reads(n_families, **sim)
    o = fgx_opts.defaults(kind=kind, **opts)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups={0: 50, 1: 100, 2: 1000}[kind])
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        blob = np.concatenate([g.blob, np.zeros(64, dtype=np.uint8)])
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp,
                                          C.byref(out), C.byref(nd), C.byref(dp))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert nd.value == 0, nd.value                  # nothing deferred: every family was decided by the emulated kernels
        got = C.string_at(out.data, out.data_len) if out.data_len else b""
        assert int(out.count) == want["count"] and got == want["data"]
        stats = np.ctypeslib.as_array(out.stats, shape=(len(want["stats"]),)) if hasattr(out, "stats") else None
        if stats is not None:
            assert np.array_equal(np.array(stats, dtype=np.uint64), want["stats"])
        if want_path:
            lib.fgx_debug_last_split_builds.restype = None
            lib.fgx_debug_last_split_builds.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            b = (C.c_uint64 * 4)()
            lib.fgx_debug_last_split_builds(h, b)
            if want_path == "packed":
                assert int(b[2]) == 1 and int(b[0]) >= 0.9 * n_families, list(b)
            elif want_path == "pair":
                assert int(b[2]) == 2 and int(b[0]) > 0 and int(b[1]) > 0, list(b)
    finally:
        lib.fgx_destroy(h)


def test_depth8_families_through_the_packed_build():
    """The headline shape: k_split_parse + k_split_cols<.., 1> (the packed pass, its tables, its items) + k_split_finish + k_call_full + k_emit's pair
    writer, on 10 000 families of 8 pairs — the kernels the bench times, lane by lane."""
    run_isolated("test_wavemu", "check_device_entry", 0, 10000, dict(family_size=8), dict(min_reads=1), "packed", env=env(), timeout=1500)


@pytest.mark.parametrize("case", ["depth3_seg4", "long_tail_pair_and_deep", "noisy_depth8", "depth8_classic_build", "min_reads_2_no_overlap"])
def test_other_simplex_shapes(case):
    if case == "depth3_seg4":            # four families per wavefront: k_simplex_seg<4>, then k_simplex_wave2 for what it hands on
        run_isolated("test_wavemu", "check_device_entry", 0, 1500, dict(family_size=3), dict(min_reads=1), env=env(), timeout=900)
    elif case == "long_tail_pair_and_deep":   # the launch pair <1> + <2>, the larger-slice stages, the streaming kernels for families above 64 records
        run_isolated("test_wavemu", "check_device_entry", 0, 1200, dict(family_size=2, family_size_max=50), dict(min_reads=1), "pair", env=env(), timeout=1500)
    elif case == "noisy_depth8":         # 2 % errors: many columns for k_call_full, items that outgrow the first slice
        run_isolated("test_wavemu", "check_device_entry", 0, 1500, dict(family_size=8, error_rate_ppm=20000), dict(min_reads=1), env=env(), timeout=900)
    elif case == "depth8_classic_build":      # FGX_S2_PACKED=0: run_cols (lane = column, f32 sums behind the error-inflated gate)
        run_isolated("test_wavemu", "check_device_entry", 0, 1500, dict(family_size=8), dict(min_reads=1), env=env(FGX_S2_PACKED=0), timeout=900)
    else:
        run_isolated("test_wavemu", "check_device_entry", 0, 1500, dict(family_size=6), dict(min_reads=2, overlapping_consensus=0), env=env(), timeout=900)


@pytest.mark.parametrize("kind", [0, 1, 2], ids=["simplex", "duplex", "codec"])
def test_per_field_record_writers(kind):
    """A 70-character read-name prefix: the one-store field writers refuse every record — emit_generic / k_emit_duplex / k_emit_codec (the latter two launched
    only because the fast writers counted what they refused)."""
    sim = [dict(family_size=5), dict(family_size=6, duplex=1), dict(family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1)][kind]
    opts = dict(read_name_prefix=b"n" * 70, **(dict(overlapping_consensus=0) if kind == 2 else dict(min_reads=1)))
    run_isolated("test_wavemu", "check_device_entry", kind, 150, sim, opts, env=env(), timeout=900)


@pytest.mark.parametrize("kind,sim", [(1, dict(family_size=12, duplex=1)), (2, dict(family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1))], ids=["duplex", "codec"])
def test_duplex_and_codec_wavefront_kernels(kind, sim):
    """k_family_wave<1> / <2> and their record writers (k_emit_duplex_fast / k_emit_codec_fast) on 400 molecules."""
    opts = dict(min_reads=1) if kind == 1 else dict(overlapping_consensus=0)
    run_isolated("test_wavemu", "check_device_entry", kind, 400, sim, opts, env=env(), timeout=1500)


def test_deep_families_streaming_kernels():
    """Families above 64 records: k_deep_parse (its two-wavefront build for 65 - 128 records, the workgroup build above) + k_deep_cols, which streams the rows
    from global memory: plain; noisy with read-through inserts (the pre-correction at the load) and a deeper --min-reads; a quality floor that masks observations."""
    e = env()
    run_isolated("test_wavemu", "check_device_entry", 0, 60, dict(family_size=35, family_size_max=60), dict(min_reads=1), env=e, timeout=1500)
    run_isolated("test_wavemu", "check_device_entry", 0, 50, dict(family_size=40, family_size_max=100, error_rate_ppm=20000, read_length=151, insert_mean=170, insert_sd=40),
                 dict(min_reads=3), env=e, timeout=1500)
    run_isolated("test_wavemu", "check_device_entry", 0, 30, dict(family_size=70, family_size_max=120), dict(min_reads=1, min_input_base_quality=25), env=e, timeout=1500)
