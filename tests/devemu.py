"""Builds and binds tests/devemu/devemu.cpp (TEST INFRASTRUCTURE): the lane-per-item kernels of reject_device.hip / canon_device.hip and
their launch code compiled for the host (a launch = a serial loop over blocks and threads).  Never used by the product or the GPU tests."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "devemu", "devemu.cpp")
SANITIZE = os.environ.get("HOSTEMU_SANITIZE") == "1"     # tools/sanitize_host.sh: ASan + UBSan build (the "device" buffers are malloc'd: a slab or offset overrun is reported)
OUT = os.path.join(ROOT, "tests", "hostemu", "_build", "libdevemu_san.so" if SANITIZE else "libdevemu.so")
CSRC = os.path.join(ROOT, "fgumi_amd", "csrc")


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [SRC, os.path.join(ROOT, "include", "fgumi_amd.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h") or f in ("reject_device.hip", "canon_device.hip")]
    return any(os.path.getmtime(d) > t for d in deps)


def build():
    if _stale():
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        tmp = f"{OUT}.{os.getpid()}.tmp"          # (xdist workers may build at the same time: each writes its own file, the rename is atomic)
        cc = ["/opt/rocm/lib/llvm/bin/clang++", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libasan", "-fno-omit-frame-pointer"] if SANITIZE else ["g++"]
        subprocess.check_call(cc + ["-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-w",
                               SRC, "-o", tmp, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
        os.replace(tmp, OUT)
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        VP, U32, U64 = C.c_void_p, C.c_uint32, C.c_uint64
        L.demu_simplex_rejects.argtypes = [VP, VP, U64, VP, VP, U32, VP, U32, VP, U64, VP, VP, VP]
        L.demu_canon.argtypes = [VP, C.c_int, VP, VP, VP, VP, VP, U32, VP, VP, VP, VP, VP, VP]
        L.demu_gates.argtypes = [C.c_uint8, C.c_uint8, U32, U32, VP, U32, VP, VP, VP, VP, VP, VP]
        L.demu_cap_depth.argtypes = [C.c_uint8, C.c_uint8, U32, U32, U32, VP]
        L.demu_cap_depth.restype = U32
        L.demu_packed_end.argtypes = [VP, VP, U32, U32, U32, U32, U32, C.c_int, U32, U32, U32, U32, U32, VP, VP, VP, VP]
        L.demu_t1.argtypes = [C.c_uint8, C.c_uint8, U32, VP]
        L.demu_t1.restype = None
        L.demu_t2.argtypes = [C.c_uint8, C.c_uint8, U32, VP]
        L.demu_t2.restype = None
        _lib = L
    return _lib


def simplex_rejects(o, g):
    """The rejects of batch `g` (GroupedReads) as the emulated device path produces them: (n_out_of_scope, bytes, count)."""
    L = lib()
    n, cnt, oos = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
    cap = int(g.blob.size) + 4 * int(g.n_rec) + 64
    out = np.zeros(cap, dtype=np.uint8)
    rc = L.demu_simplex_rejects(C.addressof(o), g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp,
                                out.ctypes.data, cap, C.addressof(n), C.addressof(cnt), C.addressof(oos))
    assert rc == 0, rc
    return oos.value, bytes(out[:n.value]) if oos.value == 0 else b"", cnt.value


def canon(o, codec, g, deferred):
    """Canonicalises the groups `deferred` (sorted indices) of batch `g` through the emulated kernel with api.cpp's slot layout.
    Returns per deferred group: (status, [canonical record bytes or None], [block_size prefix or None], delta5)."""
    L = lib()
    nd = len(deferred)
    d = np.array(deferred, dtype=np.uint32)
    first = np.zeros(nd + 1, dtype=np.uint64)
    for k, gi in enumerate(deferred):
        first[k + 1] = first[k] + (int(g.grp_first[gi + 1]) - int(g.grp_first[gi]))
    n_slots = int(first[nd])
    out_off = np.zeros(max(1, n_slots), dtype=np.uint64)
    b = 0
    for k, gi in enumerate(deferred):
        for i, r in enumerate(range(int(g.grp_first[gi]), int(g.grp_first[gi + 1]))):
            out_off[int(first[k]) + i] = b + 4
            b += 4 + int(g.rec_len[r])
    out = np.zeros(b + 16, dtype=np.uint8)
    out_len = np.zeros(max(1, n_slots), dtype=np.uint32)
    status = np.full(max(1, nd), -7, dtype=np.int32)
    delta = np.zeros(5 * max(1, nd), dtype=np.uint64)
    rc = L.demu_canon(C.addressof(o), int(codec), g.blob.ctypes.data, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.grp_first.ctypes.data, d.ctypes.data, nd,
                      first.ctypes.data, out.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, status.ctypes.data, delta.ctypes.data)
    assert rc == 0, rc
    res = []
    for k in range(nd):
        recs, pre = [], []
        for i in range(int(first[k]), int(first[k + 1])):
            o0, ln = int(out_off[i]), int(out_len[i])
            recs.append(bytes(out[o0:o0 + ln]) if ln else None)
            pre.append(int.from_bytes(bytes(out[o0 - 4:o0]), "little"))
        res.append((int(status[k]), recs, pre, [int(x) for x in delta[5 * k:5 * k + 5]]))
    return res


def gates(pre, post, quals, n, m=None, tie=0):
    """The three unanimous-column gates (gate_core.h) on single-base columns: quals = (count, stride) uint8, n = observations per column,
    m = the member count the pre-gate's bound is built for (>= n; default n).  Returns (q_exact, q_approx, q_pre, f32 sums): int32 arrays
    with -1 where a gate does not answer."""
    L = lib()
    quals = np.ascontiguousarray(quals, dtype=np.uint8)
    n = np.ascontiguousarray(n, dtype=np.uint32)
    m = n if m is None else np.ascontiguousarray(m, dtype=np.uint32)
    cnt = quals.shape[0]
    qe, qa, qp = (np.zeros(cnt, dtype=np.int32) for _ in range(3))
    sums = np.zeros((cnt, 2), dtype=np.float32)
    rc = L.demu_gates(pre, post, tie, cnt, quals.ctypes.data, quals.shape[1], n.ctypes.data, m.ctypes.data, qe.ctypes.data, qa.ctypes.data, qp.ctypes.data, sums.ctypes.data)
    assert rc == 0
    return qe, qa, qp, sums


def cap_depth(pre, post, min_bq, n_max=64, tie=0):
    """unanimous_cap_depth (gate_core.h): (n_safe or None when the table never allows the shortcut, cap)."""
    cap = C.c_uint32(0)
    n = lib().demu_cap_depth(pre, post, tie, min_bq, n_max, C.addressof(cap))
    return (None if n == 0xFFFFFFFF else int(n)), int(cap.value)


def packed_end(seq, qual, m, len_e, cnt_e, rev, min_bq, nsafe, cap, min_cons_bq=2, min_reads=1):
    """One end of a family through k_split_cols's packed column pass (packed_core.h) on the host.  seq: (rows, ss) uint8 packed codes, qual:
    (rows, qs) uint8.  Returns (code, qual, depth, flagged) per consensus column."""
    L = lib()
    seq = np.ascontiguousarray(seq, dtype=np.uint8); qual = np.ascontiguousarray(qual, dtype=np.uint8)
    code = np.zeros(cnt_e, dtype=np.uint8); qo = np.zeros(cnt_e, dtype=np.uint8); dep = np.zeros(cnt_e, dtype=np.uint16); fl = np.zeros(cnt_e, dtype=np.uint8)
    rc = L.demu_packed_end(seq.ctypes.data, qual.ctypes.data, qual.shape[1], seq.shape[1], m, len_e, cnt_e, 1 if rev else 0, min_bq, nsafe, cap, min_cons_bq, min_reads,
                           code.ctypes.data, qo.ctypes.data, dep.ctypes.data, fl.ctypes.data)
    assert rc == 0, rc
    return code, qo, dep, fl


def t1_table(pre, post, tie=0):
    """S2Lds::t1: the consensus quality of a column that holds ONE observation, by its quality (0xFF: not answered from the table)."""
    t = np.zeros(96, dtype=np.uint8)
    lib().demu_t1(pre, post, tie, t.ctypes.data)
    return t


def t2_table(pre, post, tie=0):
    """S2Image::t2: the consensus quality of a column that holds TWO observations of one base, [q1 * 94 + q2] in file order (0xFF: no table answer)."""
    t = np.zeros(94 * 94, dtype=np.uint8)
    lib().demu_t2(pre, post, tie, t.ctypes.data)
    return t.reshape(94, 94)
