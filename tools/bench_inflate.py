#!/usr/bin/env python3
"""The device's BGZF inflate (+ CRC-32) alone on a simulated grouped BAM stream (fgx_bgzf_inflate_device_bench): level-1 BGZF blocks made by
the library's own compressor or by zlib, uploaded once, inflated `--reps` times; the inflated bytes are compared with the input.
usage: python tools/bench_inflate.py [--families 250000] [--depth 8] [--reps 5] [--zlib LEVEL]"""
import argparse
import ctypes as C
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", type=int, default=250000)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--zlib", type=int, default=None, help="compress the blocks with zlib at this level instead of the library's level-1 compressor")
    a = ap.parse_args()
    import numpy as np
    from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, bgzf, lib, simulate_grouped_reads
    g = simulate_grouped_reads(a.families, family_size=a.depth)
    blob = np.ascontiguousarray(g.blob)
    if a.zlib is None:
        raw = np.frombuffer(memoryview(bgzf.native_deflate(blob, 1, 32, with_eof=False)[0]), dtype=np.uint8)
    else:
        parts = []
        data = blob.tobytes()
        for o in range(0, len(data), 0xFF00):
            d = data[o:o + 0xFF00]
            co = zlib.compressobj(a.zlib, zlib.DEFLATED, -15)
            comp = co.compress(d) + co.flush()
            bsize = 18 + len(comp) + 8 - 1
            parts.append(bytes([0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 66, 67, 2, 0, bsize & 0xFF, bsize >> 8]) + comp +
                         (zlib.crc32(d) & 0xFFFFFFFF).to_bytes(4, "little") + len(d).to_bytes(4, "little"))
        raw = np.frombuffer(b"".join(parts), dtype=np.uint8)
    raw = np.concatenate([raw, np.zeros(64, dtype=np.uint8)])
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1))
    lib.fgx_bgzf_inflate_device_bench.restype = C.c_int
    lib.fgx_bgzf_inflate_device_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_void_p, C.c_uint64]
    ms, n = C.c_double(), C.c_uint64()
    out = np.zeros(blob.size + 64, dtype=np.uint8)
    rc = lib.fgx_bgzf_inflate_device_bench(c._h, raw.ctypes.data, raw.size - 64, a.reps, C.byref(ms), C.byref(n), out.ctypes.data, out.size)
    if rc != 0:
        raise SystemExit("inflate failed: " + lib.fgx_last_error(c._h).decode())
    same = n.value == blob.size and bool(np.array_equal(out[:n.value], blob))
    print(json.dumps({"inflated_bytes": int(n.value), "compressed_bytes": int(raw.size - 64), "blocks": (int(blob.size) + 0xFEFF) // 0xFF00, "ms_per_pass": ms.value,
                      "inflated_GBs": n.value / ms.value / 1e6, "compressor": "zlib level %d" % a.zlib if a.zlib is not None else "library level 1",
                      "equal_to_input": same, "lanes": os.environ.get("FGX_INFL_LANES", "default")}))
    c.close()
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
