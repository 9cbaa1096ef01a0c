"""Builds and binds tests/hostemu/hostemu.cpp (TEST INFRASTRUCTURE): the general path's host orchestration with the kernels' work done
lane by lane on the host, so the CPU suite can drive it end to end against the oracle.  Never used by the product or the GPU tests."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")
SANITIZE = os.environ.get("HOSTEMU_SANITIZE") == "1"     # tools/sanitize_host.sh: clang -fsanitize=address,undefined build, loaded under LD_PRELOAD of the ASan runtime
OUT = os.path.join(ROOT, "tests", "hostemu", "_build", "libhostemu_san.so" if SANITIZE else "libhostemu.so")
CSRC = os.path.join(ROOT, "fgumi_amd", "csrc")
PARTS = [SRC, os.path.join(CSRC, "simplex_host.cpp"), os.path.join(CSRC, "duplex_host.cpp"), os.path.join(CSRC, "codec_host.cpp")]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = PARTS + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "fgumi_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build():
    if _stale():
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = f"{OUT}.{os.getpid()}.tmp"          # (xdist workers may build at the same time: each writes its own file, the rename is atomic)
        cc = ["/opt/rocm/lib/llvm/bin/clang++", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libasan", "-fno-omit-frame-pointer"] if SANITIZE else ["g++"]
        subprocess.check_call(cc + ["-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-Wno-unused-function", "-Wno-attributes", "-w"] + PARTS + ["-o", tmp, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
        os.replace(tmp, OUT)
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        VP, U32 = C.c_void_p, C.c_uint32
        L.hemu_create.argtypes = [VP]
        L.hemu_create.restype = VP
        L.hemu_destroy.argtypes = [VP]
        L.hemu_last_error.argtypes = [VP]
        L.hemu_last_error.restype = C.c_char_p
        L.hemu_set_reference.argtypes = [VP, U32, VP, VP]
        L.hemu_process_batch.argtypes = [VP, VP, VP, VP, U32, VP, U32, VP]
        _lib = L
    return _lib


def process(opts, contigs, g):
    """opts: tests/fgx_opts.Options; contigs: list of bytes or None; g: GroupedReads.  Returns the oracle's result layout."""
    from fgx_opts import Output
    L = lib()
    h = L.hemu_create(C.addressof(opts))
    assert h
    try:
        if contigs:
            bufs = [C.create_string_buffer(bytes(s), max(1, len(s))) for s in contigs]
            ptrs = (C.c_void_p * len(bufs))(*[C.cast(b, C.c_void_p).value for b in bufs])
            lens = (C.c_uint64 * len(bufs))(*[len(s) for s in contigs])
            L.hemu_set_reference(h, len(bufs), ptrs, lens)
        out = Output()
        rc = L.hemu_process_batch(h, g.blob.ctypes.data, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.addressof(out))
        if rc != 0:
            raise RuntimeError(L.hemu_last_error(h).decode())
        return dict(data=C.string_at(out.data, out.data_len) if out.data_len else b"", count=int(out.count), stats=np.array(list(out.stats), dtype=np.uint64),
                    rejects=C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"", n_rejects=int(out.n_rejects))
    finally:
        L.hemu_set_reference(h, 0, None, None)
        L.hemu_destroy(h)
