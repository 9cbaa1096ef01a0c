"""The product's host side, executed on the CPU: tests/apiemu links api.cpp, the general path, bgzf_host.cpp and pipeline.cpp — unmodified —
against a fake HIP runtime, the lane-per-item kernel sources compiled for the host (kernels.hip, boundaries.hip, grouping.hip,
reject_device.hip, canon_device.hip), zlib for the BGZF kernels and a stand-in for the device-resident pipeline (groups deferred by rule,
the rest decided through the general path); `FGX_LIB` makes the ordinary ctypes binding load that library.  What runs here, in child
interpreters, against the oracle:
  * the BODIES of the GPU tests of every opt-in path: canonical second pass (form made on the host / by the kernel / inside the device
    entry), `--rejects` side kernels through both entries, fgx_run_bam resubmitting only the deferred groups — and their combinations;
  * the EXISTING host-entry GPU tests as they are (parity, duplex, CODEC, methylation, streaming pipeline), with the opt-in paths off and on;
  * the hostile fuzz groups through the host entry; the sharded general path; two ranks over gloo; FindBoundaries / grouper / simulator
    kernels against their host twins.
APIEMU_DEFER=mod3 makes the stand-in defer every third group as well, which puts first-pass, second-pass and general-path records next to
each other in every order the splices have to handle.  What this cannot show is the real kernels running on an MI355X."""
import ctypes as C
import os

import numpy as np
import pytest

import apiemu
import fgx_opts
import orc
from isolated import run_isolated


def env(**flags):
    e = {"FGX_LIB": apiemu.build(), "FGX_ALLOW_LIBM_MISMATCH": "1"}
    e.update({k: str(v) for k, v in flags.items()})
    return e


def check_plain_hybrid(kind, defer):
    """No opt-in flag: the splice of device-pipeline records with general-path records, as it has always run on the GPU."""
    import random
    import test_canon_codec as tcc
    import test_canon_core as tc
    import test_gpu_duplex_canon as tg
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    rng = random.Random(5)
    if kind == 2:
        sim = simulate_grouped_reads(60, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1)
        groups = [sim.records(g // 2) if g % 2 == 0 else tcc.codec_molecule(rng, 100 + g) for g in range(120)]
        o = fgx_opts.defaults(kind=2, overlapping_consensus=0)
    else:
        sim = simulate_grouped_reads(60, family_size=4, duplex=int(kind == 1))
        groups = []
        for g in range(120):
            m = tc.duplex_indel_molecule(rng, 100 + g) if g % 2 else None
            groups.append(m if m else sim.records(g // 2))
        o = fgx_opts.defaults(kind=kind, min_reads=1)
    gr = GroupedReads.from_groups(groups)
    want = orc.process(o, gr.blob, gr.rec_off, gr.rec_len, gr.grp_first, batch_groups={0: 50, 1: 100, 2: 1000}[kind])
    got = tg.product(o, gr)
    assert got["data"] == want["data"] and got["count"] == want["count"] and np.array_equal(got["stats"], want["stats"]), (got["stats"].tolist(), want["stats"].tolist())
    if defer != "none":
        assert got["deferred"] > 20


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("defer", ["indel", "mod3", "none"])
def test_plain_hybrid_splice(kind, defer):
    if defer == "none" and kind != 0:
        pytest.skip("the stand-in decides through the general path, which takes everything: nothing to splice")
    run_isolated("test_apiemu", "check_plain_hybrid", kind, defer, env=env(APIEMU_DEFER=defer))


DUPLEX = [(dict(overlapping_consensus=1), (1, 1, 0)), (dict(overlapping_consensus=0, min_input_base_quality=20), (2, 1, 1)),
          (dict(overlapping_consensus=1, cell_tag=b"\0\0", produce_per_base_tags=0), (1, 1, 1))]


@pytest.mark.parametrize("on_device", [0, 1])
@pytest.mark.parametrize("defer", ["indel", "mod3"])
@pytest.mark.parametrize("kw,mr", DUPLEX)
def test_duplex_canonical_second_pass(kw, mr, defer, on_device):
    run_isolated("test_gpu_duplex_canon", "test_indel_molecules_take_the_canonical_second_pass", None, kw, mr,
                 env=env(FGX_DUPLEX_CANON=1, FGX_CANON_DEVICE=on_device, APIEMU_DEFER=defer))


@pytest.mark.parametrize("on_device", [0, 1])
@pytest.mark.parametrize("defer", ["indel", "mod3"])
@pytest.mark.parametrize("kw", [dict(), dict(min_input_base_quality=20, produce_per_base_tags=1), dict(codec_min_reads_per_strand=2, cell_tag=b"\0\0"),
                                dict(codec_outer_bases_length=5, codec_has_outer_bases_qual=1, codec_outer_bases_qual=7, codec_min_duplex_length=10)])
def test_codec_canonical_second_pass(kw, defer, on_device):
    run_isolated("test_gpu_zz_codec_canon", "check_codec_indel_molecules", kw, env=env(FGX_CODEC_CANON=1, FGX_CANON_DEVICE=on_device, APIEMU_DEFER=defer))


def test_canonical_pass_can_be_switched_off():
    run_isolated("test_gpu_duplex_canon", "check_second_pass_can_be_switched_off", env=env())
    run_isolated("test_gpu_zz_codec_canon", "check_switched_off", env=env(FGX_CODEC_CANON=0))


def check_device_entry_emu(kw, seed):
    """tests/test_gpu_zz_rejects_device.check_device_entry with host arrays standing in for the tensors in HBM."""
    import test_gpu_zz_rejects_device as tgr
    from fgumi_amd._lib import Options, Output, lib
    g = tgr.batch(seed)
    o = fgx_opts.defaults(kind=0, track_rejects=1, **kw)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        blob = np.concatenate([g.blob, np.zeros(16, dtype=np.uint8)])
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp,
                                          C.byref(out), C.byref(nd), C.byref(dp))
        assert rc == 0, lib.fgx_last_error(h).decode()
        assert int(out.n_rejects) == want["n_rejects"] > 0
        assert (C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"") == want["rejects"]
        if nd.value == 0:
            assert (C.string_at(out.data, out.data_len) if out.data_len else b"") == want["data"] and int(out.count) == want["count"]
    finally:
        lib.fgx_destroy(h)


def check_device_entry_refuses_without_the_flag():
    import test_gpu_zz_rejects_device as tgr
    from fgumi_amd._lib import Options, Output, lib
    g = tgr.batch(3)
    o = fgx_opts.defaults(kind=0, track_rejects=1, min_reads=1)
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    try:
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp,
                                          C.byref(out), C.byref(nd), C.byref(dp))
        assert rc != 0 and b"--rejects" in lib.fgx_last_error(h)
    finally:
        lib.fgx_destroy(h)


@pytest.mark.parametrize("defer", ["indel", "mod3", "none"])
@pytest.mark.parametrize("kw", [dict(min_reads=1), dict(min_reads=2, max_reads=3), dict(min_reads=3, overlapping_consensus=0, min_input_base_quality=30),
                                dict(min_reads=2, trim=1, min_input_base_quality=25)])
def test_rejects_side_kernels_host_entry(kw, defer):
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry", kw, 11, env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER=defer))


@pytest.mark.parametrize("kw", [dict(min_reads=1), dict(min_reads=2, max_reads=3)])
def test_rejects_side_kernels_device_entry(kw):
    run_isolated("test_apiemu", "check_device_entry_emu", kw, 12, env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER="none"))
    run_isolated("test_apiemu", "check_device_entry_emu", kw, 12, env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER="mod3"))


@pytest.mark.parametrize("defer", ["none", "mod3"])
@pytest.mark.parametrize("kind", ["duplex", "codec"])
def test_duplex_codec_rejects_side_kernels_host_entry(kind, defer):
    """Round 6: the duplex / CODEC callers' `--rejects` through api.cpp's new branch (device pipeline first, then the side kernels with the batch's own
    output slots); with deferred molecules (mod3) the whole batch goes to the general path — same bytes either way."""
    import test_gpu_zz_rejects_device as tgr
    for kw in tgr.STRAND_KWS[kind][:3]:
        run_isolated("test_gpu_zz_rejects_device", "check_host_entry_strand", kind, kw, 21, env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER=defer))
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry_strand", kind, tgr.STRAND_KWS[kind][1], 22, kind == "duplex", env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER="indel"))


@pytest.mark.parametrize("kind", ["duplex", "codec"])
def test_duplex_codec_rejects_side_kernels_device_entry(kind):
    import test_gpu_zz_rejects_device as tgr
    run_isolated("test_gpu_zz_rejects_device", "check_device_entry_strand", kind, tgr.STRAND_KWS[kind][1], 23, False, env=env(FGX_REJECTS_DEVICE=1, APIEMU_DEFER="none"))


def test_rejects_without_the_flag_take_the_general_path():
    run_isolated("test_gpu_zz_rejects_device", "check_host_entry", dict(min_reads=2), 11, env=env(FGX_REJECTS_DEVICE=0))          # (whole batch on the general path: same answer)
    run_isolated("test_apiemu", "check_device_entry_refuses_without_the_flag", env=env(FGX_REJECTS_DEVICE=0))


def check_resident_pass(kind, kw, mr, on_gpu=False):
    """fgx_process_batch_device with the canonical second pass inside it (FGX_CANON_RESIDENT=1): the records it returns are those of every
    group it did not leave in the deferred list, in group order, counters included; the deferred list shrinks to what the canonical form
    cannot express (or the second pass deferred again)."""
    import random
    import test_canon_codec as tcc
    import test_canon_core as tc
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    from fgumi_amd._lib import Options, Output, lib
    rng = random.Random(77)
    if kind == 2:
        sim = simulate_grouped_reads(60, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1)
        groups = [sim.records(g // 3) if g % 3 == 0 else tcc.codec_molecule(rng, 500 + g) for g in range(180)]
        o = fgx_opts.defaults(kind=2, overlapping_consensus=0, **kw)
    else:
        sim = simulate_grouped_reads(60, family_size=4, duplex=1)
        groups = []
        for g in range(180):
            m = tc.duplex_indel_molecule(rng, 500 + g) if g % 3 else None
            groups.append(m if m else sim.records(g // 3))
        o = fgx_opts.defaults(kind=1, **kw)
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    per_group = []
    for x in groups:
        g1 = GroupedReads.from_groups([x])
        per_group.append(orc.process(o, g1.blob, g1.rec_off, g1.rec_len, g1.grp_first, batch_groups=1000))
    g = GroupedReads.from_groups(groups)
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        if on_gpu:                                                  # tests/test_gpu_zz_canon_device.py: the same check on an MI355X
            import torch
            from fgumi_amd._lib import hip_memcpy_d2h as fetch
            dg = g.to_device(0)
            torch.cuda.synchronize()
            ptrs = (dg.blob.data_ptr(), dg.blob_len, dg.rec_off.data_ptr(), dg.rec_len.data_ptr(), dg.n_rec, dg.grp_first.data_ptr(), dg.n_grp)
        else:                                                       # tests/apiemu: device memory is host memory
            def fetch(p, n):
                return C.string_at(p, n) if n else b""
            blob = np.concatenate([g.blob, np.zeros(16, dtype=np.uint8)])
            ptrs = (blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp)
        rc = lib.fgx_process_batch_device(h, *ptrs, C.byref(out), C.byref(nd), C.byref(dp))
        assert rc == 0, lib.fgx_last_error(h).decode()
        left = set(np.frombuffer(fetch(dp.value, 4 * nd.value), dtype=np.uint32).tolist()) if nd.value else set()
        d = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d)
        first_pass_deferred, canon = int(d[0]), int(d[1])
        assert first_pass_deferred > 60 and canon > 0.2 * first_pass_deferred and first_pass_deferred - canon == len(left), (first_pass_deferred, canon, len(left))
        want = b"".join(per_group[i]["data"] for i in range(len(groups)) if i not in left)
        stats = sum((per_group[i]["stats"] for i in range(len(groups)) if i not in left), np.zeros(28, dtype=np.uint64))
        got = fetch(out.data, int(out.data_len))
        assert got == want
        assert int(out.count) == sum(per_group[i]["count"] for i in range(len(groups)) if i not in left)
        assert np.array_equal(np.array(list(out.stats), dtype=np.uint64), stats), (list(out.stats), stats.tolist())
    finally:
        lib.fgx_destroy(h)


@pytest.mark.parametrize("defer", ["indel", "mod3"])
@pytest.mark.parametrize("kind,kw,mr", [(1, dict(overlapping_consensus=1), (1, 1, 0)), (1, dict(overlapping_consensus=0, min_input_base_quality=20), (2, 1, 1)),
                                        (2, dict(), None), (2, dict(codec_min_reads_per_strand=2, cell_tag=b"\0\0"), None)])
def test_canonical_pass_inside_the_device_entry(kind, kw, mr, defer):
    run_isolated("test_apiemu", "check_resident_pass", kind, kw, mr, env=env(FGX_DUPLEX_CANON=1, FGX_CODEC_CANON=1, FGX_CANON_RESIDENT=1, APIEMU_DEFER=defer))


@pytest.mark.parametrize("all_on", [0, 1])
def test_host_entry_gpu_tests_run_against_the_emulation(all_on):
    """The GPU tests that go through the HOST entry (simplex / duplex / CODEC parity batches, the reference's unit-test inputs, the
    methylation-aware mode, the duplex canonical pass) run as they are against tests/apiemu: the host side they exercise — validation,
    hybrid splice, general path, record assembly — is then covered on the CPU as well, by the very assertions the hardware run makes.
    The per-base work of the general path is done by the REAL kernels.hip sources compiled for the host (k_column_jobs, k_meth_annotate,
    the device libm self-test).  Tests that need a real device (the device-resident entry with torch tensors, kernels with wavefront
    intrinsics and no host form) are left out."""
    import subprocess
    import sys
    files = ["tests/test_gpu_parity.py", "tests/test_gpu_duplex.py", "tests/test_gpu_codec.py", "tests/test_gpu_methylation.py", "tests/test_gpu_duplex_canon.py",
             "tests/test_gpu_pipeline.py"]            # (fgx_run_bam: BAM file -> consensus BAM file == the oracle; boundaries.hip / grouping.hip are the real sources)
    skip = "not device_resident and not full_size and not stay_on_the_device and not noisy_batch and not device_deflate and not device_boundaries"
    e = dict(os.environ)
    e.update(env(FGX_OPT_IN_ALL=all_on))      # all_on: the rehearsal of "the whole suite with every opt-in path switched on" (HISTORY.md §15, step 2: done in round 4, the paths are defaults now)
    p = subprocess.run([sys.executable, "-m", "pytest"] + files + ["-m", "gpu", "-q", "-x", "-k", skip, "-p", "no:cacheprovider", "-n", "4"],
                       env=e, cwd=apiemu.ROOT, capture_output=True, text=True, timeout=1800)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail + p.stderr[-3000:]
    import re
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 186, tail


def check_hybrid_fuzz(kind, seed0, n_seeds):
    """The hostile groups of the general-path fuzz through the HOST ENTRY (validation, device-pipeline stand-in, splice, opt-in paths as the
    environment says): records, counters and rejects as the oracle gives them, or an error where the reference refuses the batch."""
    import random
    import test_general_path_fuzz as fuzz
    from fgumi_amd import GroupedReads
    from fgumi_amd._lib import Options, Output, lib
    name = ["simplex", "duplex", "codec"][kind]
    done = errors = 0
    for seed in range(seed0, seed0 + n_seeds):
        rng = random.Random(seed)
        exotic = rng.random() < 0.6
        groups = [x for x in (fuzz.random_group(rng, g, name, exotic) for g in range(60)) if x]
        if not groups:
            continue
        o = fuzz.random_options(rng, name)
        g = GroupedReads.from_groups(groups)
        try:
            want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=max({0: 50, 1: 100, 2: 1000}[kind], len(groups)))
        except RuntimeError:
            want = None
        po = Options.from_buffer_copy(bytes(o))
        h = lib.fgx_create(C.byref(po))
        assert h, lib.fgx_global_error().decode()
        try:
            out = Output()
            rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
            if want is None:
                assert rc != 0, (name, seed)
                errors += 1
                continue
            assert rc == 0, (name, seed, lib.fgx_last_error(h).decode())
            assert (C.string_at(out.data, out.data_len) if out.data_len else b"") == want["data"], (name, seed)
            assert int(out.count) == want["count"] and np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"]), (name, seed)
            if o.track_rejects:
                assert int(out.n_rejects) == want["n_rejects"] and (C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"") == want["rejects"], (name, seed)
            done += 1
        finally:
            lib.fgx_destroy(h)
    assert done >= n_seeds // 2, (done, errors)


@pytest.mark.parametrize("defer", ["indel", "mod3"])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_hostile_groups_through_the_host_entry(kind, defer):
    run_isolated("test_apiemu", "check_hybrid_fuzz", kind, 40000 + 100 * kind, 12,
                 env=env(APIEMU_DEFER=defer, FGX_REJECTS_DEVICE=1, FGX_DUPLEX_CANON=1, FGX_CODEC_CANON=1, FGX_CANON_DEVICE=1))


def test_hostile_groups_through_the_host_entry_default_switches():
    for kind in (0, 1, 2):
        run_isolated("test_apiemu", "check_hybrid_fuzz", kind, 41000 + 100 * kind, 8, env=env(APIEMU_DEFER="mod3"))


def check_sharded_general_path_with_side_rejects():
    """Enough deferred groups (> 1024) for the general path to shard them over helper callers on threads while the rejects come from the
    side kernels: the helpers must not track rejects during that call and must track them again afterwards."""
    from fgumi_amd import GroupedReads, simulate_grouped_reads
    from fgumi_amd._lib import Options, Output, lib
    sim = simulate_grouped_reads(3300, family_size=1, family_size_max=5, seed=21)
    g = GroupedReads.from_groups([sim.records(i) for i in range(sim.n_grp)])
    o = fgx_opts.defaults(kind=0, track_rejects=1, min_reads=2, max_reads=3)
    want = orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=50)
    assert want["n_rejects"] > 100
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        for flag in ("1", "0", "1"):              # side kernels, then the whole batch on the (sharded) general path with tracking, then side kernels again
            os.environ["FGX_REJECTS_DEVICE"] = flag
            out = Output()
            rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
            assert rc == 0, lib.fgx_last_error(h).decode()
            assert (C.string_at(out.data, out.data_len) if out.data_len else b"") == want["data"], flag
            assert np.array_equal(np.array(list(out.stats), dtype=np.uint64), want["stats"]), flag
            assert int(out.n_rejects) == want["n_rejects"] and (C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"") == want["rejects"], flag
    finally:
        lib.fgx_destroy(h)


def test_sharded_general_path_with_side_rejects():
    run_isolated("test_apiemu", "check_sharded_general_path_with_side_rejects", env=env(APIEMU_DEFER="mod3"))


def check_boundaries_and_grouping_kernels():
    """boundaries.hip (segment guesses, walks, mutual check, repair rounds) and grouping.hip (keys, compaction, group bounds) are
    lane-per-item kernels around scans: tests/apiemu compiles the REAL sources for the host.  The GPU tests' own bodies run here with
    host arrays standing in for the tensors in HBM."""
    import test_gpu_pipeline as tp
    import test_grouping as tg
    from fgumi_amd import VanillaUmiConsensusCaller
    from fgumi_amd._lib import lib

    def boundaries(c, stream, start):
        buf = np.frombuffer(bytes(stream) + bytes(64), dtype=np.uint8).copy()
        n, used = C.c_uint64(), C.c_uint64()
        rc = lib.fgx_record_boundaries_device(c._h, buf.ctypes.data, len(stream), start, None, None, 0, C.byref(n), C.byref(used))
        assert rc == 0, lib.fgx_last_error(c._h)
        off = np.zeros(max(1, n.value), dtype=np.uint64)
        ln = np.zeros(max(1, n.value), dtype=np.uint32)
        n2 = C.c_uint64()
        rc = lib.fgx_record_boundaries_device(c._h, buf.ctypes.data, len(stream), start, off.ctypes.data, ln.ctypes.data, n.value, C.byref(n2), C.byref(used))
        assert rc == 0 and n2.value == n.value
        return off[:n.value], ln[:n.value], used.value
    tp._device_boundaries = boundaries
    tp.test_device_boundaries_equal_the_sequential_chain()
    tp.test_device_boundaries_with_records_longer_than_a_segment_and_record_like_payloads()
    c = VanillaUmiConsensusCaller("", "A")
    for kw in (dict(cell_tag="CB"), dict(cell_tag=None), dict(cell_tag="CB", strip_strand_suffix=True), dict(cell_tag=None, strip_strand_suffix=True, allow_unmapped=True)):
        okw = dict(kw)
        okw["cell_tag"] = kw["cell_tag"].encode() if kw["cell_tag"] else None
        for blob, off, ln in (tg._mixed_stream(), tg._stream([]), tg._stream([tg._rec()]), tg._stream([tg._rec("7", "X")])):
            want = orc.group_records(blob, off, ln, **okw)
            got = c.group_records(blob, off, ln, **kw)
            assert np.array_equal(got.rec_off, want[0]) and np.array_equal(got.rec_len, want[1]) and np.array_equal(got.grp_first, want[2])
    c.close()


def test_boundaries_and_grouping_kernels_on_the_host():
    run_isolated("test_apiemu", "check_boundaries_and_grouping_kernels", env=env())


@pytest.mark.parametrize("resident", [0, 1])
@pytest.mark.parametrize("defer", ["mod3", "indel"])
def test_pipeline_resubmits_only_the_deferred_groups(defer, resident):
    """FGX_PIPE_SUBSET=1: fgx_run_bam sends only the groups the device entry deferred (copies of their records) through the general path and
    merges on the host, instead of the whole batch through the host entry.  The file-to-file tests, with the stand-in deferring groups in
    every batch: the consensus BAM equals the oracle's either way."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env(FGX_PIPE_SUBSET=1, APIEMU_DEFER=defer, FGX_PIPE_DEBUG=1))
    if resident:            # with the canonical second pass inside the device entry: the merged stream's group offsets serve the resubmission
        e.update(FGX_DUPLEX_CANON="1", FGX_CODEC_CANON="1", FGX_CANON_RESIDENT="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_pipeline.py", "-m", "gpu", "-q", "-x", "-s", "-k", "not device_deflate and not device_boundaries",
                        "-p", "no:cacheprovider"], env=e, cwd=apiemu.ROOT, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    trace = p.stdout + p.stderr
    assert "12 passed" in trace
    assert trace.count("decided alone") > (20 if defer == "mod3" else 0) and "the whole batch through the host entry" not in trace


def check_pipeline_with_indel_duplex_molecules(chunk=1 << 17):
    """fgx_run_bam over a duplex BAM in which two molecules in three carry indels, every combination of the opt-in switches the environment
    holds: the consensus BAM equals the oracle's."""
    import random
    import tempfile
    import pathlib
    import test_canon_core as tc
    import test_gpu_pipeline as tp
    from fgumi_amd import DuplexConsensusCaller, GroupedReads, simulate_grouped_reads
    rng = random.Random(91)
    sim = simulate_grouped_reads(450, family_size=4, duplex=1)
    groups, used = [], 0
    for g in range(450):
        m = tc.duplex_indel_molecule(rng, 7000 + g) if g % 3 else None
        if not m:                                   # (every group its own MI: the pipeline regroups the stream by MI)
            m, used = sim.records(used), used + 1
        groups.append(m)
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=1)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 0
    c = DuplexConsensusCaller("", "A", [1, 1, 0], cell_tag="CB", overlapping_consensus=True)
    with tempfile.TemporaryDirectory() as d:
        st = tp._run_and_compare(pathlib.Path(d), c, o, gr, 100, chunk, strip_strand_suffix=True, cell_tag=None)     # (the generator varies CB inside a molecule: group by MI alone)
    c.close()
    assert st["deferred_groups"] > 0


@pytest.mark.parametrize("flags", [dict(), dict(FGX_PIPE_SUBSET=1), dict(FGX_DUPLEX_CANON=1, FGX_CANON_RESIDENT=1), dict(FGX_PIPE_SUBSET=1, FGX_DUPLEX_CANON=1, FGX_CANON_RESIDENT=1),
                                   dict(FGX_PIPE_SUBSET=1, FGX_DUPLEX_CANON=1, FGX_CANON_DEVICE=1)])
def test_pipeline_with_indel_duplex_molecules(flags):
    run_isolated("test_apiemu", "check_pipeline_with_indel_duplex_molecules", env=env(**flags))
    if flags.get("FGX_PIPE_SUBSET"):           # one chunk: hundreds of deferred groups in one batch (whole tables in one copy, joined spans)
        run_isolated("test_apiemu", "check_pipeline_with_indel_duplex_molecules", 0, env=env(FGX_PIPE_DEBUG=1, **flags))


def check_device_simulator_equals_the_host_one():
    """k_sim_generate (kernels.hip: the synthetic input of bench.py, written straight into HBM) against the host generator of the same
    source (simgen.h): the same bytes, offsets and group boundaries — simplex, duplex and CODEC shapes, a family-size range."""
    from fgumi_amd import VanillaUmiConsensusCaller, simulate_grouped_reads
    from fgumi_amd._lib import SimParams, lib
    c = VanillaUmiConsensusCaller("", "A")
    try:
        for kw in (dict(family_size=3), dict(family_size=2, family_size_max=9), dict(family_size=6, duplex=1), dict(family_size=3, read_length=300, insert_mean=350, insert_sd=60, codec=1)):
            want = simulate_grouped_reads(400, seed=7, **kw)
            p = SimParams()
            p.seed, p.n_families, p.read_length, p.family_size = 7, 400, kw.get("read_length", 150), kw["family_size"]
            p.insert_mean, p.insert_sd, p.error_rate_ppm = 300, 50, 1000
            for k, v in kw.items():
                if k not in ("family_size", "read_length"):
                    setattr(p, k, v)
            bl, nr = C.c_uint64(), C.c_uint64()
            assert lib.fgx_sim_sizes(C.byref(p), C.byref(bl), C.byref(nr)) == 0
            assert bl.value == want.blob.size and nr.value == want.n_rec
            blob = np.zeros(bl.value + 16, dtype=np.uint8)
            off = np.zeros(max(1, nr.value), dtype=np.uint64)
            ln = np.zeros(max(1, nr.value), dtype=np.uint32)
            grp = np.zeros(401, dtype=np.uint32)
            rc = lib.fgx_sim_generate_device(c._h, C.byref(p), blob.ctypes.data, off.ctypes.data, ln.ctypes.data, grp.ctypes.data)
            assert rc == 0, lib.fgx_last_error(c._h)
            assert np.array_equal(blob[:bl.value], want.blob) and np.array_equal(off[:nr.value], want.rec_off) and np.array_equal(ln[:nr.value], want.rec_len)
            assert np.array_equal(grp, want.grp_first)
    finally:
        c.close()


def test_device_simulator_equals_the_host_one():
    run_isolated("test_apiemu", "check_device_simulator_equals_the_host_one", env=env())


def check_switch_semantics():
    """A path is on by default; its own switch set to 0 turns it off; FGX_OPT_IN_ALL=0 turns every one off unless its own switch says 1."""
    import random
    import test_canon_core as tc
    import test_gpu_duplex_canon as tg
    from fgumi_amd import GroupedReads
    rng = random.Random(8)
    gr = GroupedReads.from_groups([m for m in (tc.duplex_indel_molecule(rng, g) for g in range(60)) if m])
    o = fgx_opts.defaults(kind=1)
    for all_on, own, expect in ((None, None, True), ("0", None, False), ("1", "0", False), (None, "0", False), ("0", "1", True), ("1", None, True)):
        for k, v in (("FGX_OPT_IN_ALL", all_on), ("FGX_DUPLEX_CANON", own)):
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        assert (tg.product(o, gr)["canon"] > 0) == expect, (all_on, own)


def test_switch_semantics():
    run_isolated("test_apiemu", "check_switch_semantics", env=env())


def check_pipeline_with_indel_codec_molecules():
    import random
    import tempfile
    import pathlib
    import test_canon_codec as tcc
    import test_gpu_pipeline as tp
    from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions, GroupedReads, simulate_grouped_reads
    rng = random.Random(92)
    sim = simulate_grouped_reads(300, family_size=3, read_length=150, insert_mean=200, insert_sd=30, codec=1)
    groups = [tcc.codec_molecule(rng, 9000 + g) if g % 3 else sim.records(g // 3) for g in range(300)]
    gr = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(kind=2, overlapping_consensus=0, cell_tag=b"\0\0", produce_per_base_tags=1)
    c = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True))
    with tempfile.TemporaryDirectory() as d:
        st = tp._run_and_compare(pathlib.Path(d), c, o, gr, 1000, 1 << 17, cell_tag=None)
    c.close()
    assert st["deferred_groups"] > 0


@pytest.mark.parametrize("flags", [dict(), dict(FGX_OPT_IN_ALL=1)])
def test_pipeline_with_indel_codec_molecules(flags):
    run_isolated("test_apiemu", "check_pipeline_with_indel_codec_molecules", env=env(**flags))


def _rank_worker(rank, world, port, outdir, shape):
    import torch
    torch.cuda.set_device = lambda *a, **k: None          # (the emulation has no device to select)
    import test_gpu_distributed as tgd
    tgd._worker(rank, world, port, outdir, shape)


def check_two_ranks_run_the_product():
    """tests/test_gpu_distributed.py's body with the emulation on every rank: two processes over gloo, each runs its contiguous shard of the
    family stream through the product's host entry, the payloads gathered in rank order and the summed counters equal the oracle's output
    of the whole stream."""
    import tempfile
    import pathlib
    import torch.multiprocessing as mp
    import test_gpu_distributed as tgd
    from fgumi_amd import simulate_grouped_reads
    for shape in (dict(family_size=3), dict(family_size=2, family_size_max=50)):
        with tempfile.TemporaryDirectory() as d:
            world = 2
            mp.spawn(_rank_worker, args=(world, tgd._free_port(), d, shape), nprocs=world, join=True)
            g = simulate_grouped_reads(world * tgd.F_PER_RANK, **shape)
            want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
            tmp = pathlib.Path(d)
            assert np.load(tmp / "payload.npy").tobytes() == want["data"]
            sizes, stats = np.load(tmp / "sizes.npy"), np.load(tmp / "stats.npy")
            assert sizes[:, 1].sum() == want["count"] and sizes[:, 2].sum() == g.n_rec
            assert np.array_equal(stats[:len(want["stats"])], want["stats"].astype(np.int64))


def test_two_ranks_run_the_product_over_gloo():
    run_isolated("test_apiemu", "check_two_ranks_run_the_product", env=env(APIEMU_DEFER="mod3"), timeout=600)


def check_pipeline_async(rounds, chunk):
    """fgx_run_bam (the form the environment selects) on a fake runtime whose streams are worker threads (APIEMU_ASYNC=1): uploads, the BGZF
    inflate stand-in and event records run beside the device stage and beside one another, with random pauses.  The consensus BAM must be
    the oracle's every time: a missing ordering in the host logic shows as a mismatch."""
    import pathlib
    import tempfile
    import test_gpu_pipeline as tp
    from fgumi_amd import simulate_grouped_reads
    for r in range(rounds):
        g = simulate_grouped_reads(1800, family_size=2, family_size_max=30, seed=100 + r)
        c = tp._caller()
        with tempfile.TemporaryDirectory() as d:
            st = tp._run_and_compare(pathlib.Path(d), c, fgx_opts.defaults(min_reads=1), g, 50, chunk)
            assert st["chunks"] > 8, st["chunks"]
            st = tp._run_and_compare(pathlib.Path(d), c, fgx_opts.defaults(min_reads=1), g, 50, chunk * 3)     # (the same caller: buffers and events reused)
        c.close()


@pytest.mark.parametrize("flags", [dict(), dict(FGX_FRONT_PAD=256), dict(APIEMU_D2D_LATE=1, FGX_FRONT_PAD=256)])   # (+ device-to-device hipMemcpy that returns before the bytes move)
def test_pipeline_under_asynchronous_streams(flags):
    """fgx_run_bam with truly asynchronous streams in the emulation: whether the HOST logic orders what it must (uploads, the inflate stand-in
    and event records run beside the device stage, with random pauses)."""
    run_isolated("test_apiemu", "check_pipeline_async", 4, 1 << 16, env=env(APIEMU_ASYNC=1, **flags))
