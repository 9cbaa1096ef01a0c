"""Runs one check function in a child interpreter (TEST INFRASTRUCTURE).  Used by the GPU tests of code paths that have not run on
hardware yet: a device fault there ends the child, not the whole suite."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def run_isolated(module, func, *args, env=None, timeout=900):
    code = (f"import sys; sys.path[:0] = [{ROOT!r}, {HERE!r}]\n"
            f"import {module} as m\n"
            f"m.{func}(*{args!r})\n")
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, "-c", code], env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"child exited with {p.returncode}\n--- stdout\n{p.stdout[-4000:]}\n--- stderr\n{p.stderr[-8000:]}"
