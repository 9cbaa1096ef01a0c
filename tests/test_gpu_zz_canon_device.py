"""GPU parity of the canonical second pass with the canonical form computed ON THE DEVICE (the default since round 4; FGX_CANON_DEVICE=0 computes
it on the host's cores; fgumi_amd/csrc/canon_device.hip: a lane per deferred molecule runs the scalar source of canon_core.h over the records
already uploaded).  Same inputs and assertions as the host-canonicalised twins (tests/test_gpu_duplex_canon.py, test_gpu_zz_codec_canon.py):
byte-identical to the oracle, counters included; the resident test asserts that the device entry's deferred list IS the out-of-scope remainder."""
import pytest

from isolated import run_isolated

pytestmark = pytest.mark.gpu


FLAGS = {"FGX_CANON_DEVICE": "1", "FGX_DUPLEX_CANON": "1", "FGX_CODEC_CANON": "1"}


@pytest.mark.parametrize("kw,mr", [(dict(overlapping_consensus=1), (1, 1, 0)), (dict(overlapping_consensus=0, min_input_base_quality=20), (2, 1, 1))])
def test_duplex_indel_molecules_canonicalised_on_the_device(kw, mr):
    run_isolated("test_gpu_duplex_canon", "test_indel_molecules_take_the_canonical_second_pass", None, kw, mr, env=FLAGS)


@pytest.mark.parametrize("kw", [dict(), dict(min_input_base_quality=20, produce_per_base_tags=1)])
def test_codec_molecules_canonicalised_on_the_device(kw):
    run_isolated("test_gpu_zz_codec_canon", "check_codec_indel_molecules", kw, env=FLAGS)


@pytest.mark.parametrize("kind,kw,mr", [(1, dict(overlapping_consensus=1), (1, 1, 0)), (2, dict(), None)])
def test_canonical_pass_inside_the_device_entry(kind, kw, mr):
    """FGX_CANON_RESIDENT=1: fgx_process_batch_device canonicalises, re-runs and merges on the device; its deferred list shrinks to the
    out-of-scope remainder (the CPU twin: tests/test_apiemu.py, same check function)."""
    run_isolated("test_apiemu", "check_resident_pass", kind, kw, mr, True, env=dict(FLAGS, FGX_CANON_RESIDENT="1"))
