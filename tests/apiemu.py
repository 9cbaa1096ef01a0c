"""Builds tests/apiemu (TEST INFRASTRUCTURE): the product's whole host side linked against a fake HIP runtime, host-compiled lane-per-item
kernels and a stand-in for the device-resident pipeline — `FGX_LIB=<this .so>` makes fgumi_amd._lib load it instead of libfgumi_amd.so, so
the CPU suite can run the bodies of the GPU tests of the opt-in paths.  Never used by the product or by the GPU tests."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fgumi_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "apiemu", "apiemu.cpp")
BUILD = os.path.join(ROOT, "tests", "hostemu", "_build")
TSAN = os.environ.get("HOSTEMU_SANITIZE") == "thread"    # ThreadSanitizer build (the sharded general path, the host threads of the canonical pass)
SANITIZE = os.environ.get("HOSTEMU_SANITIZE") == "1"     # tools/sanitize_host.sh: ASan + UBSan build, loaded under LD_PRELOAD of the ASan runtime (children inherit it)
OUT = os.path.join(BUILD, "libapiemu_tsan.so" if TSAN else "libapiemu_san.so" if SANITIZE else "libapiemu.so")
HOST = ["api.cpp", "simplex_host.cpp", "duplex_host.cpp", "codec_host.cpp", "bgzf_host.cpp", "pipeline.cpp"]
CL = "/opt/rocm/lib/llvm/bin/clang++"        # (inflate_core.h uses clang builtins)


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC, os.path.join(ROOT, "tests", "devemu", "devemu.cpp"), os.path.join(ROOT, "tests", "hostemu", "column_emu.h"), os.path.join(ROOT, "include", "fgumi_amd.h")]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build():
    if not _stale():
        return OUT
    os.makedirs(BUILD, exist_ok=True)
    tag = f"{os.getpid()}"
    flags = ["-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-DFGX_HAVE_CODEC", "-I/opt/rocm/include", "-w", "-pthread"]
    if TSAN:
        flags += ["-g", "-fsanitize=thread", "-shared-libsan", "-fno-omit-frame-pointer"]
    if SANITIZE:
        flags += ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libasan", "-fno-omit-frame-pointer"]
    srcs = [os.path.join(CSRC, f) for f in HOST] + [SRC]
    first = os.path.join(BUILD, f"apiemu_nostub.{tag}.so")
    subprocess.check_call([CL] + flags + srcs + ["-o", first, "-lz"])
    # device entry points nothing here emulates (BGZF / boundaries / grouping / filter kernels, the simulator, the libm self-test): abort() when called
    und = subprocess.run(["ldd", "-r", first], capture_output=True, text=True)
    syms = sorted({ln.split()[2] for ln in (und.stdout + und.stderr).splitlines() if ln.startswith("undefined symbol: _ZN3fgx")})
    stubs = os.path.join(BUILD, f"apiemu_stubs.{tag}.S")
    with open(stubs, "w") as f:
        f.write(".text\n")
        for s in syms:
            body = "ret" if "release" in s else "jmp abort@PLT"
            f.write(f".globl {s}\n.type {s},@function\n{s}:\n  {body}\n")
    tmp = f"{OUT}.{tag}.tmp"
    subprocess.check_call([CL] + flags + srcs + [stubs, "-o", tmp, "-lz"])
    left = subprocess.run(["ldd", "-r", tmp], capture_output=True, text=True)
    missing = [ln for ln in (left.stdout + left.stderr).splitlines() if ln.startswith("undefined symbol") and "asan" not in ln and "ubsan" not in ln and "tsan" not in ln]
    assert not missing, missing
    os.replace(tmp, OUT)
    for p in (first, stubs):
        os.remove(p)
    return OUT
