import os, sys, subprocess, pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: make sure its shared object exists (prebuilt on the GPU box)."""
    so = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    # the product library: prebuilt on the GPU box (it travels with the snapshot); in a fresh checkout with hipcc at hand it is built once
    # (cross-compilation needs no GPU) — without hipcc the tests that need it fail loudly through fgumi_amd._lib.LibraryMissing
    lib = os.path.join(ROOT, "fgumi_amd", "libfgumi_amd.so")
    if not os.path.exists(lib) and not os.environ.get("FGX_LIB") and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        subprocess.check_call([sys.executable, "-m", "fgumi_amd.build"], cwd=ROOT)
    yield
