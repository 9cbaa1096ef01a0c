"""N>1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous shard of the family stream, runs the
per-shard caller (here the oracle stands in for the GPU path — no GPU in this container), and the shard
payloads gathered in rank order must equal the single-process result byte for byte."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

F_PER_RANK = 150


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    from fgumi_amd.distributed import gather_payload_to_root, gather_sizes, max_over_ranks, sum_over_ranks
    g = simulate_grouped_reads(F_PER_RANK, family_size=3, first_family=rank * F_PER_RANK)      # weak-scaling shard
    res = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    local = torch.frombuffer(bytearray(res["data"]), dtype=torch.uint8)
    sizes = gather_sizes([local.numel(), res["count"], g.n_rec], "cpu")
    stats = torch.tensor(sum_over_ranks(res["stats"].tolist(), "cpu"), dtype=torch.int64)     # additive counters (bench.py's all-reduce)
    payload = gather_payload_to_root(local, root=0)
    t = max_over_ranks(0.001 * (rank + 1), "cpu")
    if rank == 0:
        np.save(os.path.join(outdir, "payload.npy"), payload.numpy())
        np.save(os.path.join(outdir, "sizes.npy"), sizes.numpy())
        np.save(os.path.join(outdir, "stats.npy"), stats.numpy())
        assert abs(t - 0.001 * world) < 1e-12
    else:
        assert payload is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_concatenate_to_single_process_output(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    g = simulate_grouped_reads(world * F_PER_RANK, family_size=3)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    payload = np.load(tmp_path / "payload.npy").tobytes()
    sizes = np.load(tmp_path / "sizes.npy")
    stats = np.load(tmp_path / "stats.npy")
    assert payload == want["data"]
    assert sizes[:, 1].sum() == want["count"] and sizes[:, 2].sum() == g.n_rec and sizes[:, 0].sum() == len(want["data"])
    assert np.array_equal(stats, want["stats"].astype(np.int64))


def test_shard_helpers():
    from fgumi_amd.distributed import balanced_shards, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    w = [1] * 90 + [50] * 10            # long tail at the end
    sh = balanced_shards(w, 4)
    assert sh[0][0] == 0 and sh[-1][1] == len(w) and all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
    tot = [sum(w[a:b]) for a, b in sh]
    assert max(tot) <= 1.5 * sum(w) / 4 + 50


# ---- strong scaling: ONE long-tail family stream cut into shards of equal record bytes ------------------------------
F_TOTAL = 400
LONG_TAIL = dict(family_size=1, family_size_max=20)


def _strong_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads, simulated_family_bytes
    from fgumi_amd.distributed import balanced_shards, gather_payload_to_root, gather_sizes
    w = simulated_family_bytes(F_TOTAL, **LONG_TAIL)                       # every rank derives the same cut points
    lo, hi = balanced_shards(w, world)[rank]
    g = simulate_grouped_reads(hi - lo, first_family=lo, **LONG_TAIL)      # this rank's contiguous shard of the one stream
    assert int(w[lo:hi].sum()) == len(g.blob)
    res = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    local = torch.frombuffer(bytearray(res["data"]) or bytearray(1), dtype=torch.uint8)[:len(res["data"])]
    sizes = gather_sizes([local.numel(), res["count"], g.n_rec, hi - lo, len(g.blob)], "cpu")
    payload = gather_payload_to_root(local, root=0)
    if rank == 0:
        np.save(os.path.join(outdir, "payload.npy"), payload.numpy())
        np.save(os.path.join(outdir, "sizes.npy"), sizes.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_strong_scaling_weighted_shards_reassemble_to_the_whole_stream(tmp_path):
    world = 2
    mp.spawn(_strong_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    g = simulate_grouped_reads(F_TOTAL, **LONG_TAIL)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    payload = np.load(tmp_path / "payload.npy").tobytes()
    sizes = np.load(tmp_path / "sizes.npy")
    assert payload == want["data"]
    assert sizes[:, 3].sum() == F_TOTAL and sizes[:, 2].sum() == g.n_rec and sizes[:, 4].sum() == len(g.blob)
    # equal work, not equal family counts: the byte split is within one (largest) family of even
    assert abs(int(sizes[0, 4]) - int(sizes[1, 4])) <= 2 * 40 * 340
    assert sizes[0, 3] != sizes[1, 3]


def test_c_abi_balanced_shards_equal_the_python_mirror():
    """fgx_balanced_shards (the helper a non-Python host binds) cuts a weighted family stream exactly where
    fgumi_amd.distributed.balanced_shards does — constant weights, long-tail weights, fewer families than ranks, empty streams."""
    import ctypes as C
    import numpy as np
    from fgumi_amd import lib, simulated_family_bytes
    from fgumi_amd.distributed import balanced_shards
    lib.fgx_balanced_shards.restype = C.c_int
    lib.fgx_balanced_shards.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(3)
    cases = [np.full(1000, 5280, dtype=np.uint64), simulated_family_bytes(5000, family_size=2, family_size_max=50), rng.integers(1, 10**6, size=37).astype(np.uint64),
             np.array([7], dtype=np.uint64), np.zeros(0, dtype=np.uint64), np.array([1, 1, 1000000, 1, 1], dtype=np.uint64)]
    for w in cases:
        for world in (1, 2, 3, 4, 8):
            cuts = np.zeros(world + 1, dtype=np.uint32)
            assert lib.fgx_balanced_shards(w.ctypes.data if w.size else None, w.size, world, cuts.ctypes.data) == 0
            want = balanced_shards(w, world)
            assert [(int(cuts[k]), int(cuts[k + 1])) for k in range(world)] == [(int(a), int(b)) for a, b in want], (w.size, world)
