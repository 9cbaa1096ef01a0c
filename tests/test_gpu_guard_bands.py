"""the golden and parity files once more with every device buffer of the product between sentinel-filled guard bands that are
verified after every test (FGX_GUARD_BAND: the library allocates each DevBuf at exactly the size asked for, 4 KiB of 0xA5 in front and behind;
tests/guard_plugin.py checks them).  A kernel that stores outside its buffer — scratch columns, the k_call_full item pools, descriptors, the
output — fails the test that made it do so, instead of corrupting a neighbour's memory or faulting on some later box."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.timeout(1800)
def test_no_kernel_stores_outside_its_buffers():
    env = dict(os.environ, FGX_GUARD_BAND="4096", PYTHONPATH=HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "pytest", os.path.join(HERE, "test_golden.py"), os.path.join(HERE, "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "guard_plugin",
           "-p", "no:cacheprovider", "-s"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1700)
    tail = p.stdout[-6000:] + "\n--- stderr\n" + p.stderr[-3000:]
    assert p.returncode == 0, tail
    m = re.search(r"guard bands: self test (\d+), (\d+) tests checked, (\d+) guarded buffers verified", p.stdout)
    assert m, tail
    # the mechanism sees a byte stored right behind a buffer, it looked after every test, and the library's buffers really were guarded
    assert int(m.group(1)) == 1 and int(m.group(2)) >= 50 and int(m.group(3)) >= 500, m.group(0)


def check_rejects_and_deep_families_under_guard_bands():
    """(child interpreter, FGX_GUARD_BAND set) The round-6 side kernels of the duplex / CODEC `--rejects` (host and device entries), a simplex `--rejects` batch and deep
    families (the streaming kernels), each followed by a look at every guarded buffer."""
    import ctypes as C
    import test_gpu_zz_rejects_device as tgr
    import test_gpu_deep_families as tdf
    from fgumi_amd import lib, simulate_grouped_reads
    lib.fgx_debug_check_guard_bands.restype = C.c_int
    lib.fgx_debug_check_guard_bands.argtypes = [C.c_char_p, C.c_int]
    lib.fgx_debug_guarded_buffers.restype = C.c_int

    def look(what):
        msg = C.create_string_buffer(600)
        bad = lib.fgx_debug_check_guard_bands(msg, 600)
        assert bad == 0, f"{what}: {bad} device buffer(s) written outside their bounds: {msg.value.decode()}"
    for kind in ("duplex", "codec"):
        tgr.check_host_entry_strand(kind, tgr.STRAND_KWS[kind][1], 21)
        look(kind + " host entry")
        tgr.check_device_entry_strand(kind, tgr.STRAND_KWS[kind][1], 23)
        look(kind + " device entry")
    tgr.check_host_entry(tgr.KWS[1], 11)
    look("simplex rejects")
    tdf._run(simulate_grouped_reads(400, family_size=35, family_size_max=120))
    look("deep families")
    assert lib.fgx_debug_guarded_buffers() >= 100, lib.fgx_debug_guarded_buffers()


def test_reject_side_kernels_and_streaming_kernels_under_guard_bands():
    from isolated import run_isolated
    run_isolated("test_gpu_guard_bands", "check_rejects_and_deep_families_under_guard_bands", env={"FGX_GUARD_BAND": "4096"}, timeout=900)
