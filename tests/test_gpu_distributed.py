"""N>1 path with the PRODUCT on every rank (tests/test_distributed_cpu.py runs the oracle per rank: there is no GPU where the CPU suite
runs).  Two processes share the test box's one MI355X — gloo for the rendezvous, RCCL refuses two ranks on one device — each runs its
contiguous shard of the family stream through the HIP engine behind the C ABI, and the shard payloads gathered in rank order, with the
summed counters, must equal the oracle's output for the whole stream, byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

F_PER_RANK = 400


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, shape):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, simulate_grouped_reads
    from fgumi_amd.distributed import gather_payload_to_root, gather_sizes, sum_over_ranks
    g = simulate_grouped_reads(F_PER_RANK, first_family=rank * F_PER_RANK, **shape)              # this rank's shard of the stream
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    out = c.process_batch(g)                                                                     # the HIP engine; it has no CPU fallback
    local = torch.frombuffer(bytearray(out.data), dtype=torch.uint8)
    sizes = gather_sizes([local.numel(), out.count, g.n_rec], "cpu")
    stats = torch.tensor(sum_over_ranks(c.last_stats_array, "cpu"), dtype=torch.int64)
    payload = gather_payload_to_root(local, root=0)
    c.close()
    if rank == 0:
        np.save(os.path.join(outdir, "payload.npy"), payload.numpy())
        np.save(os.path.join(outdir, "sizes.npy"), sizes.numpy())
        np.save(os.path.join(outdir, "stats.npy"), stats.numpy())
    else:
        assert payload is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("shape", [dict(family_size=3), dict(family_size=2, family_size_max=50)], ids=["depth3", "long_tail_2_to_50"])
def test_two_ranks_on_one_gpu_concatenate_to_the_oracle_output_of_the_whole_stream(tmp_path, shape):
    """depth3: the smallest BASELINE shape; long_tail_2_to_50: the family-size distribution of configs[3] (the 8-GPU long-tail run), where
    the ranks' shards differ in bytes and the wave2 / workgroup kernels decide most families."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), shape), nprocs=world, join=True)
    import fgx_opts
    import orc
    from fgumi_amd import simulate_grouped_reads
    g = simulate_grouped_reads(world * F_PER_RANK, **shape)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    payload = np.load(tmp_path / "payload.npy").tobytes()
    sizes = np.load(tmp_path / "sizes.npy")
    stats = np.load(tmp_path / "stats.npy")
    assert payload == want["data"]
    assert sizes[:, 1].sum() == want["count"] and sizes[:, 2].sum() == g.n_rec and sizes[:, 0].sum() == len(want["data"])
    n = len(want["stats"])
    assert np.array_equal(stats[:n], want["stats"].astype(np.int64))


# ---- bench.py beyond one rank: the whole line first, the gather to rank 0 last and under a watchdog -----------------------------
def _bench_two_ranks(extra_env, timeout_s):
    import json
    import subprocess
    env = dict(os.environ, FGX_BENCH_TEST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--families", "20000", "--gather-timeout", str(timeout_s)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_line_with_the_gather_beside():
    d = _bench_two_ranks({}, 120)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["reassemble_error"] is None
    assert d["config"]["value_with_reassembly_on_root"] and d["config"]["reassembled_bytes_on_rank0"] == d["config"]["output_bytes"]
    assert d["strong_scaling"]["value"] > 0 and d["strong_scaling"]["value_with_reassembly_on_root"]


@pytest.mark.timeout(900)
def test_bench_survives_a_rank_that_fails_in_the_gather():
    """Rank 1 throws inside the gather loop: rank 0 is blocked in its receive and no exception reaches it — the watchdog prints the line as
    it stood before the gather and every rank leaves with exit code 0."""
    d = _bench_two_ranks({"FGX_BENCH_TEST_GATHER_FAIL": "1"}, 15)
    assert d["value"] > 0 and d["config"]["reassemble_error"] and d["config"]["value_with_reassembly_on_root"] is None
    assert d["strong_scaling"]["value"] > 0
