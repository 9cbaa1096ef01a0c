"""pytest plugin of tests/test_gpu_guard_bands.py (TEST INFRASTRUCTURE): after every test of the run, the sentinel bands around every live device
buffer of the product library are verified (fgx_debug_check_guard_bands; the library allocates them when FGX_GUARD_BAND is set)."""
import ctypes as C

import pytest

_checks = [0, 0]          # tests checked, guarded buffers verified so far (when freed, or alive at the last check)


def _lib():
    from fgumi_amd import lib
    lib.fgx_debug_check_guard_bands.restype = C.c_int
    lib.fgx_debug_check_guard_bands.argtypes = [C.c_char_p, C.c_int]
    lib.fgx_debug_guarded_buffers.restype = C.c_int
    lib.fgx_debug_guard_self_test.restype = C.c_int
    return lib


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    outcome = yield
    outcome.get_result()                       # (a failing test fails as itself)
    lib = _lib()
    msg = C.create_string_buffer(600)
    bad = lib.fgx_debug_check_guard_bands(msg, 600)
    _checks[0] += 1
    _checks[1] = lib.fgx_debug_guarded_buffers()
    if bad != 0:
        raise AssertionError(f"guard bands: {bad} device buffer(s) written outside their bounds after {item.nodeid}: {msg.value.decode()}")


def pytest_sessionfinish(session, exitstatus):
    lib = _lib()
    print(f"\nguard bands: self test {lib.fgx_debug_guard_self_test()}, {_checks[0]} tests checked, {_checks[1]} guarded buffers verified")
