"""BASELINE.json configs[2] (duplex) and configs[4] (CODEC) at full size, through properties that do not need the oracle at that size:
sharding invariance (the concatenated shard outputs ARE the whole-batch output, counters add up — a checksum of checksums) and
idempotence over the same resident input.  Small-scale parity against the oracle is in test_gpu_duplex.py / test_gpu_codec.py; the
simplex configs[1] counterpart is test_gpu_parity.py::test_full_size_config2_properties."""
import os

import pytest

from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["duplex", "codec"])
def test_full_size_sharding_invariance_and_idempotence(kind):
    if os.environ.get("FGX_SKIP_FULL_SIZE"):
        pytest.skip("FGX_SKIP_FULL_SIZE set")
    import torch
    if torch.cuda.mem_get_info()[1] < 120 * 2**30:
        pytest.skip("needs a 288 GB-class GPU")
    if kind == "duplex":
        c = DuplexConsensusCaller("", "A", [1], cell_tag="CB", overlapping_consensus=True)
        n, shard, sim = 2_000_000, 500_000, dict(family_size=12, duplex=1)
        reads_per = 24
    else:
        c = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True, cell_tag="CB"))
        n, shard, sim = 1_000_000, 250_000, dict(family_size=4, read_length=300, insert_mean=350, insert_sd=60, codec=1)
        reads_per = 8
    dg = c.simulate_on_device(n, **sim)
    out = c.process_batch_device(dg)
    assert out.n_deferred == 0 and out.count > 0
    st = c.last_batch_statistics()
    assert st.total_reads == reads_per * n
    full = out.to_host()
    again = c.process_batch_device(dg)
    assert again.to_host() == full
    del dg, again
    torch.cuda.empty_cache()
    off = count = reads = 0
    for k in range(n // shard):
        dk = c.simulate_on_device(shard, first_family=k * shard, **sim)
        ok = c.process_batch_device(dk)
        part = ok.to_host()
        assert full[off:off + len(part)] == part, f"shard {k} differs from its slice of the whole batch"
        off += len(part)
        count += ok.count
        reads += c.last_batch_statistics().total_reads
        del dk
    assert off == len(full) and count == out.count and reads == st.total_reads
    c.close()


def test_noisy_million_family_batch_stays_on_the_device():
    """1 M depth-8 families with 3 % substitution errors: a fifth of the columns needs `call_full`, more than the default pool (1/8 of the
    column bound) holds.  The pool must grow and the batch be run again on the device: nothing deferred, every family two records, and
    a second pass (with the pool it has learned) identical."""
    if os.environ.get("FGX_SKIP_FULL_SIZE"):
        pytest.skip("FGX_SKIP_FULL_SIZE set")
    import torch
    from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions
    if torch.cuda.mem_get_info()[1] < 120 * 2**30:
        pytest.skip("needs a 288 GB-class GPU")
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)
    n = 1_000_000
    dg = c.simulate_on_device(n, family_size=8, error_rate_ppm=30000)
    out = c.process_batch_device(dg)
    assert out.n_deferred == 0 and out.count == 2 * n
    assert c.last_batch_statistics().total_reads == 16 * n
    first = out.to_host()
    again = c.process_batch_device(dg)
    assert again.n_deferred == 0 and again.to_host() == first
    c.close()

