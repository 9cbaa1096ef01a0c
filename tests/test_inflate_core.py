"""The device's DEFLATE decoder (csrc/inflate_core.h: one GPU lane per BGZF block) run on the host through
fgx_inflate_block_host — the same source — against zlib: every level and strategy (stored, fixed, dynamic, Huffman-only, RLE), sizes
around the lane-slice boundaries of the 64-lane CRC-32 fold, BAM record bytes, and corrupted streams (which must be refused)."""
import ctypes as C
import random
import zlib

import pytest

from fgumi_amd import lib, simulate_grouped_reads


def _inflate(comp: bytes, n: int):
    out = C.create_string_buffer(n + 16)
    crc = C.c_uint32()
    lib.fgx_inflate_block_host.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    st = lib.fgx_inflate_block_host(comp + bytes(16), len(comp), out, n, C.byref(crc))
    return st, out.raw[:n], crc.value


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_inflate_core_equals_zlib(level):
    rng = random.Random(5 + level)
    blob = bytes(simulate_grouped_reads(300, family_size=4).blob)
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
        for n in (0, 1, 2, 63, 64, 65, 100, 1000, 4096, 65280, 65535):
            for kind in range(4):
                if kind == 0:
                    d = bytes(rng.randrange(256) for _ in range(n))
                elif kind == 1:
                    d = bytes([rng.choice(b"ACGT")]) * n
                elif kind == 2:
                    o = rng.randrange(0, max(1, len(blob) - n))
                    d = blob[o:o + n]
                else:
                    d = (b"abcabcabd" * 8000)[:n]
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                comp = co.compress(d) + co.flush()
                st, out, crc = _inflate(comp, len(d))
                assert st == 0 and out == d and crc == (zlib.crc32(d) & 0xFFFFFFFF), (level, strategy, n, kind, st)


def test_inflate_core_refuses_corrupted_streams():
    rng = random.Random(9)
    data = bytes(simulate_grouped_reads(200, family_size=4).blob)[:60000]
    comp = zlib.compress(data, 1)[2:-4]
    want = zlib.crc32(data) & 0xFFFFFFFF
    for _ in range(300):
        b = bytearray(comp)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        st, out, crc = _inflate(bytes(b), len(data))
        assert st != 0 or crc != want          # a flipped bit is caught by the decoder or by the CRC-32
    st, _, _ = _inflate(comp, len(data) - 1)   # ISIZE smaller than what the stream holds
    assert st != 0
    st, _, _ = _inflate(comp, len(data) + 1)   # ... or larger
    assert st != 0


def check_truncated_payloads_never_read_past_the_slack():
    """a truncated / corrupt payload kept decoding garbage until the output overflowed and could read ~120 KB past the input.
    The payload sits at the END of a mapping whose next page is PROT_NONE, 8 bytes of slack behind it (what the host entry documents): a
    load past `in_len + 8` is a segmentation fault of this child process."""
    import mmap
    libc = C.CDLL(None, use_errno=True)
    libc.mprotect.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    PAGE = mmap.PAGESIZE
    rng = random.Random(17)
    data = bytes(simulate_grouped_reads(200, family_size=4).blob)[:65000]
    lib.fgx_inflate_block_host.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    out = C.create_string_buffer(65536 + 64)
    n_pages = 24
    m = mmap.mmap(-1, (n_pages + 1) * PAGE)
    base = C.addressof(C.c_char.from_buffer(m))
    assert libc.mprotect(base + n_pages * PAGE, PAGE, 0) == 0, C.get_errno()
    refused = 0
    for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (1, zlib.Z_HUFFMAN_ONLY), (0, zlib.Z_DEFAULT_STRATEGY)):
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = co.compress(data) + co.flush()
        for trial in range(60):
            cut = rng.randrange(1, len(comp)) if trial else len(comp)          # trial 0: the whole payload, which must decode
            piece = bytearray(comp[:cut])
            if trial % 3 == 2:                                                   # ... and bit flips on top
                piece[rng.randrange(len(piece))] ^= 1 << rng.randrange(8)
            assert len(piece) + 8 <= n_pages * PAGE
            at = n_pages * PAGE - 8 - len(piece)
            m[at:at + len(piece)] = bytes(piece)
            m[at + len(piece):n_pages * PAGE] = b"\0" * 8
            st = lib.fgx_inflate_block_host(base + at, len(piece), out, len(data), None)
            if trial == 0:
                assert st == 0 and out.raw[:len(data)] == data
            else:
                refused += st != 0
    assert refused > 200


def test_truncated_payloads_never_read_past_the_slack():
    from isolated import run_isolated
    run_isolated("test_inflate_core", "check_truncated_payloads_never_read_past_the_slack")


# ---- round 5: the two-phase form (k_bgzf_tokenize + k_bgzf_resolve): the decoder leaves literals and a list of match entries, a second pass
#      plays the list — in order, and in the kernel's schedule (batches of 64 entries, the frontier rule) emulated lane by lane -----------------
def _inflate2(comp: bytes, n: int, mode: int):
    out = C.create_string_buffer(n + 16)
    ne, rounds = C.c_uint32(), C.c_uint32()
    lib.fgx_inflate_block_two_phase_host.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    st = lib.fgx_inflate_block_two_phase_host(comp + bytes(16), len(comp), out, n, mode, C.byref(ne), C.byref(rounds))
    return st, out.raw[:n], ne.value, rounds.value


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_two_phase_inflate_equals_zlib(level, mode):
    rng = random.Random(50 + level)
    blob = bytes(simulate_grouped_reads(300, family_size=4).blob)
    for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
        for n in (0, 1, 2, 63, 64, 65, 100, 1000, 4096, 65280, 65535):
            for kind in range(5):
                if kind == 0:
                    d = bytes(rng.randrange(256) for _ in range(n))
                elif kind == 1:
                    d = bytes([rng.choice(b"ACGT")]) * n                      # one run: matches that repeat their own beginning (distance 1)
                elif kind == 2:
                    o = rng.randrange(0, max(1, len(blob) - n))
                    d = blob[o:o + n]
                elif kind == 3:
                    d = (b"abcabcabd" * 8000)[:n]                             # periods of 3 and 9: distance < length, not a power of two
                else:
                    unit = bytes(rng.randrange(256) for _ in range(37))       # near copies 37 bytes apart: sources INSIDE a batch of 64 entries
                    d = b"".join(unit[:k % 37] + bytes([k & 255]) for k in range(n // 19 + 2))[:n]
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
                comp = co.compress(d) + co.flush()
                st, out, ne, rounds = _inflate2(comp, len(d), mode)
                assert st == 0 and out == d, (level, strategy, n, kind, st, mode)
                assert ne <= n // 3 + n // 255 + 2 and (n == 0 or ne >= 1)          # (bgzf_entry_cap, engine.h: the device sizes a block's list from its ISIZE)


def test_two_phase_inflate_takes_few_rounds_on_bam_records():
    """The schedule's point: on BAM records (matches one record stride back) a batch of 64 entries needs a handful of rounds, not 64."""
    data = bytes(simulate_grouped_reads(400, family_size=8).blob)[:65280]
    comp = zlib.compress(data, 1)[2:-4]
    st, out, ne, rounds = _inflate2(comp, len(data), 1)
    assert st == 0 and out == data
    batches = (ne + 63) // 64
    assert ne > 3000 and rounds <= 12 * batches, (ne, rounds, batches)


def test_two_phase_inflate_refuses_corrupted_streams():
    rng = random.Random(19)
    data = bytes(simulate_grouped_reads(200, family_size=4).blob)[:60000]
    comp = zlib.compress(data, 1)[2:-4]
    refused = 0
    for _ in range(300):
        b = bytearray(comp)
        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        for mode in (0, 1):
            st, out, _, _ = _inflate2(bytes(b), len(data), mode)
            assert st != 0 or (zlib.crc32(out) & 0xFFFFFFFF) != (zlib.crc32(data) & 0xFFFFFFFF) or out == data      # (caught by the decoder, or left to the CRC-32)
            refused += st != 0
    assert refused > 100
    for mode in (0, 1):
        assert _inflate2(comp, len(data) - 1, mode)[0] != 0
        assert _inflate2(comp, len(data) + 1, mode)[0] != 0
