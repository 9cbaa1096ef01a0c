"""GPU run of the streaming pipeline's way of handling deferred groups (default since round 4; FGX_PIPE_SUBSET=0 opts out: only the groups the
device entry deferred come back, as copies of their records; the general path decides them and the merged stream is assembled on the host —
api.cpp `resubmit_deferred`, pipeline.cpp): the file-to-file tests of tests/test_gpu_pipeline.py in a child pytest with the pipeline's trace on,
which must show the deferred groups decided alone and NO whole-chunk resubmission; and once more with the switch off (the round-3 way)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(flags):
    e = dict(os.environ)
    e.update(flags, FGX_PIPE_DEBUG="1")
    p = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_pipeline.py", "-m", "gpu", "-q", "-x", "-s", "-p", "no:cacheprovider"],
                       env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout + p.stderr


def test_pipeline_file_tests_with_subset_resubmission():
    """The default: every chunk with deferred groups (the 300-record family, what the canonical form does not cover) sends only those
    groups through the general path — the trace shows it, and shows no whole-chunk resubmission."""
    log = _child({})
    assert "decided alone" in log
    assert "the whole batch through the host entry" not in log


def test_pipeline_file_tests_with_whole_chunk_resubmission():
    """FGX_PIPE_SUBSET=0 keeps the round-3 way alive (same files, same bytes)."""
    log = _child({"FGX_PIPE_SUBSET": "0"})
    assert "the whole batch through the host entry" in log and "decided alone" not in log
