"""The device's DEFLATE compressor (csrc/deflate_core.h: one GPU lane per BGZF block, greedy LZ77 + one dynamic Huffman code per
block) run on the host through fgx_deflate_block_host — the same source: whatever it writes must inflate (zlib) to the input, for
every size around the token / slice boundaries and every kind of content; on consensus records it must not be larger than zlib level 1
by more than a few per cent (it is smaller)."""
import ctypes as C
import random
import zlib

import fgx_opts
import orc
from fgumi_amd import lib, simulate_grouped_reads


def _deflate(d: bytes, cap: int = 65000) -> bytes:
    lib.fgx_deflate_block_host.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32]
    lib.fgx_deflate_block_host.restype = C.c_uint32
    out = C.create_string_buffer(cap + 16)
    n = lib.fgx_deflate_block_host(d + bytes(16), len(d), out, cap)
    return out.raw[:n]


def test_deflate_core_round_trips_through_zlib():
    rng = random.Random(3)
    blob = bytes(simulate_grouped_reads(300, family_size=4).blob)
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 100, 258, 259, 1000, 4096, 30000, 63000, 65280):
        for kind in range(6):
            if kind == 0:
                d = bytes(rng.randrange(256) for _ in range(n))
            elif kind == 1:
                d = bytes([65]) * n
            elif kind == 2:
                o = rng.randrange(0, max(1, len(blob) - n))
                d = blob[o:o + n]
            elif kind == 3:
                d = (b"abcabcabd" * 8000)[:n]
            elif kind == 4:
                d = bytes(rng.choice(b"ACGT") for _ in range(n))
            else:
                d = bytes((i * i) & 0xFF for i in range(n))
            c = _deflate(d)
            if not c:                                   # "does not fit": only what cannot be compressed into the slot
                assert kind == 0 and n > 60000 or n > 63000, (n, kind)
                continue
            assert zlib.decompress(c, -15) == d, (n, kind)
    assert _deflate(b"x" * 1000, cap=4) == b""          # a capacity that cannot hold the stream is reported, not overrun


def test_deflate_core_on_consensus_records_is_not_larger_than_zlib_level_1():
    g = simulate_grouped_reads(1500, family_size=8)
    data = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)["data"]
    mine = ref = 0
    for i in range(0, len(data), 65280):
        d = data[i:i + 65280]
        c = _deflate(d)
        assert c and zlib.decompress(c, -15) == d
        mine += len(c)
        ref += len(zlib.compress(d, 1)) - 6
    assert mine <= 1.03 * ref, (mine, ref)
