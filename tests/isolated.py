"""Runs one check function in a child interpreter (TEST INFRASTRUCTURE).  Used by the GPU tests of code paths that have not run on
hardware yet: a device fault there ends the child, not the whole suite."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


_TIMED_OUT = []        # once a child has hung, no further child is started in this pytest process (a hung GPU would cost every later test its full timeout)


def run_isolated(module, func, *args, env=None, timeout=300):
    assert not _TIMED_OUT, f"not started: an earlier isolated test timed out ({_TIMED_OUT[0]})"
    code = (f"import sys; sys.path[:0] = [{ROOT!r}, {HERE!r}]\n"
            f"import {module} as m\n"
            f"m.{func}(*{args!r})\n")
    e = dict(os.environ)
    e.update(env or {})
    try:
        p = subprocess.run([sys.executable, "-c", code], env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        _TIMED_OUT.append(f"{module}.{func}")
        raise AssertionError(f"child did not finish within {timeout} s")
    assert p.returncode == 0, f"child exited with {p.returncode}\n--- stdout\n{p.stdout[-4000:]}\n--- stderr\n{p.stderr[-8000:]}"
