"""Build the in-tree HIP shared library `fgumi_amd/libfgumi_amd.so` for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the f64 Kahan loop, the
margin gates and the glibc-compatible libm must not be contracted into FMAs (bit-exact contract).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfgumi_amd.so")
SOURCES = ["kernels.hip", "api.cpp", "simplex_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-pthread"]


def sources():
    extra = [f for f in ("fastpath.hip", "grouping.hip", "boundaries.hip", "bgzf_device.hip", "filter.hip", "canon_device.hip", "reject_device.hip", "duplex_host.cpp", "codec_host.cpp", "bgzf_host.cpp", "pipeline.cpp") if os.path.exists(os.path.join(CSRC, f))]
    return [os.path.join(CSRC, f) for f in SOURCES + extra]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "fgumi_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=OUT, extra_flags=()):
    """`out` / `extra_flags` build profiling variants (e.g. -DFGX_PHASE_TIMING=1) next to the product library."""
    if out == OUT and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags)
    if os.path.exists(os.path.join(CSRC, "codec_host.cpp")):
        cmd.append("-DFGX_HAVE_CODEC")
    for s in sources():
        cmd += ["-x", "hip", s]
    cmd += ["-lz", "-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    # python -m fgumi_amd.build [--force] [--variant NAME -DFOO=1 ...]  → fgumi_amd/variant_NAME.so
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build(force=True, out=os.path.join(HERE, f"variant_{sys.argv[i + 1]}.so"), extra_flags=sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
