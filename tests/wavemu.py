"""Builds tests/wavemu (TEST INFRASTRUCTURE): tests/apiemu's host side with the REAL device-resident pipeline — fgumi_amd/csrc/fastpath.hip: the
launch chain and every wavefront kernel — compiled for the host under a 64-lane lock-step shim (tests/wavemu/simt.h) in place of apiemu's stand-in.
`FGX_LIB=<this .so>` makes fgumi_amd._lib load it.  Never used by the product or by the GPU tests."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fgumi_amd", "csrc")
BUILD = os.path.join(ROOT, "tests", "hostemu", "_build")
OUT = os.path.join(BUILD, "libwavemu.so")
HOST = ["api.cpp", "simplex_host.cpp", "duplex_host.cpp", "codec_host.cpp", "bgzf_host.cpp", "pipeline.cpp"]
SRCS = [os.path.join(ROOT, "tests", "apiemu", "apiemu.cpp"), os.path.join(ROOT, "tests", "wavemu", "wavemu.cpp")]
CL = "/opt/rocm/lib/llvm/bin/clang++"


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SRCS + [os.path.join(ROOT, "tests", "wavemu", "simt.h"), os.path.join(ROOT, "tests", "devemu", "devemu.cpp"), os.path.join(ROOT, "include", "fgumi_amd.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.startswith("_")]
    return any(os.path.getmtime(d) > t for d in deps)


def build():
    if not _stale():
        return OUT
    os.makedirs(BUILD, exist_ok=True)
    tag = f"{os.getpid()}"
    flags = ["-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-DFGX_HAVE_CODEC", "-DAPIEMU_REAL_FASTPATH", "-gline-tables-only", "-I/opt/rocm/include", "-w",
             "-pthread"]
    srcs = [os.path.join(CSRC, f) for f in HOST] + SRCS
    first = os.path.join(BUILD, f"wavemu_nostub.{tag}.so")
    subprocess.check_call([CL] + flags + srcs + ["-o", first, "-lz", "-ldl"])
    # device entry points nothing here emulates (filter kernels, device deflate, ...): abort() when called
    und = subprocess.run(["ldd", "-r", first], capture_output=True, text=True)
    syms = sorted({ln.split()[2] for ln in (und.stdout + und.stderr).splitlines() if ln.startswith("undefined symbol: _ZN3fgx")})
    stubs = os.path.join(BUILD, f"wavemu_stubs.{tag}.S")
    with open(stubs, "w") as f:
        f.write(".text\n")
        for s in syms:
            body = "ret" if "release" in s else "jmp abort@PLT"
            f.write(f".globl {s}\n.type {s},@function\n{s}:\n  {body}\n")
    tmp = f"{OUT}.{tag}.tmp"
    subprocess.check_call([CL] + flags + srcs + [stubs, "-o", tmp, "-lz", "-ldl"])
    left = subprocess.run(["ldd", "-r", tmp], capture_output=True, text=True)
    missing = [ln for ln in (left.stdout + left.stderr).splitlines() if ln.startswith("undefined symbol")]
    assert not missing, missing
    os.replace(tmp, OUT)
    for p in (first, stubs):
        os.remove(p)
    return OUT


if __name__ == "__main__":
    print(build())
