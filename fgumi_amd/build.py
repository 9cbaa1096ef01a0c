"""Build the in-tree HIP shared library `fgumi_amd/libfgumi_amd.so` for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the f64 Kahan loop, the
margin gates and the glibc-compatible libm must not be contracted into FMAs (bit-exact contract).

Every source is its own translation unit (no relocatable device code: no kernel calls across files), so the sources are compiled
to objects side by side — one hipcc process each, objects under fgumi_amd/csrc/_obj/ keyed by the flags — and linked; an object is
rebuilt when its source, any header of csrc/ or include/, or the flags changed.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfgumi_amd.so")
SOURCES = ["kernels.hip", "api.cpp", "simplex_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-pthread"]
EXTRA = ("fastpath.hip", "grouping.hip", "boundaries.hip", "bgzf_device.hip", "filter.hip", "canon_device.hip", "reject_device.hip", "duplex_host.cpp",
         "codec_host.cpp", "bgzf_host.cpp", "pipeline.cpp")


def sources():
    extra = [f for f in EXTRA if os.path.exists(os.path.join(CSRC, f))]
    return [os.path.join(CSRC, f) for f in SOURCES + extra]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hpp"))]
    return hs + [os.path.join(HERE, "..", "include", "fgumi_amd.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("_")] + [os.path.join(HERE, "..", "include", "fgumi_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=OUT, extra_flags=()):
    """`out` / `extra_flags` build profiling variants (e.g. -DFGX_PHASE_TIMING=1) next to the product library."""
    if out == OUT and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = FLAGS + list(extra_flags)
    if os.path.exists(os.path.join(CSRC, "codec_host.cpp")):
        flags.append("-DFGX_HAVE_CODEC")
    key = hashlib.sha256(" ".join([hipcc] + flags).encode()).hexdigest()[:12]
    objdir = os.path.join(CSRC, "_obj", key)
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj
        cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, srcs))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-lz", "-o", out]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    return out


if __name__ == "__main__":
    # python -m fgumi_amd.build [--force] [--variant NAME -DFOO=1 ...]  → fgumi_amd/variant_NAME.so
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        build(force=True, out=os.path.join(HERE, f"variant_{sys.argv[i + 1]}.so"), extra_flags=sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
