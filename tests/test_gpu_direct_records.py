"""Direct records (FGX_DIRECT=1, opt-in; fastpath.h / simplex_split.inc): the split pipeline's column kernel writes the consensus records
itself — sizes predicted by the record kernel, offsets by a per-chunk scan, k_call_full patching its columns in place, the merge with the
records of the families that left the pipeline — against the oracle, byte for byte and counter for counter, and with the diagnostics
saying that the direct path (1) or the direct path + merge (2) was really taken.  Child interpreters: the switch is an environment
variable.  (tools/direct_check.py runs the same cases beside the default path, with field-level diffs.)"""
import os
import sys

import pytest

from isolated import run_isolated

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check(case, families, expect):
    import ctypes as C
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import direct_check as dc
    import fgx_opts
    import orc
    from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, lib, simulate_grouped_reads
    assert os.environ.get("FGX_DIRECT") == "1"
    sim, vo, okw, _ = dc.CASES[case]
    g = simulate_grouped_reads(families, **sim)
    okw = dict(dict(min_reads=1), **okw)
    want = orc.process(fgx_opts.defaults(**okw), g.blob, g.rec_off, g.rec_len, g.grp_first)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(**dict(dict(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), **vo)),
                                  overlapping_consensus=okw.get("overlapping_consensus", 1) != 0)
    dg = g.to_device()
    for _ in range(2):                                  # twice: the second batch reuses every buffer of the first
        out = c.process_batch_device(dg)
        got = out.to_host()
        assert out.n_deferred == 0
        assert got == want["data"], dc.first_diff(got, want["data"])
        assert np.array_equal(np.array(c.last_stats_array, dtype=np.uint64), want["stats"])
        assert lib.fgx_debug_last_direct(c._h) == expect, lib.fgx_debug_last_direct(c._h)
    c.close()


@pytest.mark.parametrize("case", ["depth 8", "depth 8, 3 % errors", "depth 4..9, min_reads 6 (orphans)", "depth 8, no per-base tags",
                                  "depth 6, read-through 151 / insert 120", "depth 8, min input q 38 (strips tails)", "depth 8, no cell tag"])
def test_direct_records_equal_the_oracle(case):
    """Every family stays in the split pipeline (noisy families retry in larger LDS slices): last_direct == 1, no merge."""
    run_isolated("test_gpu_direct_records", "check", case, 2000, 1, env={"FGX_DIRECT": "1"})


def test_direct_records_merge_with_families_that_left_the_pipeline():
    run_isolated("test_gpu_direct_records", "check", "long tail 2..50, split forced (merge)", 1200, 2, env={"FGX_DIRECT": "1", "FGX_SPLIT": "2"})


def test_direct_records_over_forced_chunks():
    run_isolated("test_gpu_direct_records", "check", "depth 8", 3000, 1, env={"FGX_DIRECT": "1", "FGX_SPLIT_CHUNKS": "8"})
