"""BGZF / BAM container I/O around the engine's record streams (host side).

The engine consumes and produces *uncompressed BAM record streams* (`[block_size u32][record]…`).  This module is the
container layer a `fgumi simplex` drop-in needs on both sides of it:

  * `write_bam`   record stream (+ header) → BGZF-compressed BAM, blocks deflated by a thread pool (zlib releases the GIL),
                  block layout as `crates/fgumi-bgzf/src/writer.rs` emits it: ≤ 0xff00 payload bytes per block, gzip member
                  with the `BC` extra subfield, CRC32 + ISIZE trailer, 28-byte EOF marker block
  * `read_bam`    BGZF BAM → header text, references, the record stream and its record boundaries (`rec_off`, `rec_len`:
                  body offsets / lengths, exactly what `fgx_process_batch` / `fgx_group_records` take); blocks are located by
                  walking the BSIZE chain and inflated by the pool (reference: `crates/fgumi-bgzf/src/reader.rs`,
                  `src/lib/unified_pipeline/bam.rs` FindBoundaries)
  * `record_boundaries`  the `block_size` chain of a record stream (a sequential length chain: host work)
  * `consensus_header`   the header of unmapped consensus output (`create_unmapped_consensus_header`,
                  src/lib/commands/consensus_runner.rs:130-173): `@HD SO:unsorted GO:query`, one `@RG`, the comment line

Compression level 1 is the reference's default for consensus output.  Python is the host language here
because the image has no Rust; the byte formats are the BAM / BGZF specifications', so a Rust host reads and writes the
same files.
"""
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np

BGZF_MAX_PAYLOAD = 0xFF00           # uncompressed bytes per block (htslib / noodles / fgumi-bgzf all use 0xff00)
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_HDR = struct.Struct("<4BI2BH2BHH")  # gzip header + extra field up to BSIZE


def _deflate_block(payload: bytes, level: int) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    data = c.compress(payload) + c.flush()
    if len(data) + 26 > 0x10000:     # incompressible payload: stored blocks always fit
        c = zlib.compressobj(0, zlib.DEFLATED, -15)
        data = c.compress(payload) + c.flush()
    bsize = len(data) + 25           # total block size - 1
    return (_HDR.pack(0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, 0x42, 0x43, 2, bsize) + data +
            struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def _pool(threads: Optional[int]) -> ThreadPoolExecutor:
    return ThreadPoolExecutor(max_workers=max(1, threads or (os.cpu_count() or 1)))


class _NativeBuffer:
    """A malloc'ed result of the library's BGZF entries, exposed as a uint8 numpy view and released with the view."""

    def __init__(self, L, ptr, n):
        self._L, self._ptr, self.n = L, ptr, n

    def array(self) -> np.ndarray:
        import ctypes as C
        if self.n == 0:
            return np.zeros(0, dtype=np.uint8)
        buf = (C.c_uint8 * self.n).from_address(self._ptr.value)
        buf._owner = self                      # the view keeps the allocation alive (numpy holds `buf`, `buf` holds this object)
        a = np.frombuffer(buf, dtype=np.uint8)
        a.flags.writeable = False
        return a

    def __del__(self):
        try:
            self._L.fgx_bgzf_free(self._ptr)
        except Exception:
            pass


def _native():
    try:
        from ._lib import load
        L = load()
        return L if hasattr(L, "fgx_bgzf_inflate") else None
    except (ImportError, OSError, RuntimeError):
        return None


def native_inflate(raw, threads: Optional[int] = None):
    """Whole BGZF file image → uncompressed bytes through the library's block-parallel zlib inflate (CRC32 / ISIZE checked).
    Returns (uint8 array, owner) — keep `owner` alive as long as the array is used — or None when the library is absent."""
    L = _native()
    if L is None:
        return None
    import ctypes as C
    buf = np.frombuffer(raw, dtype=np.uint8)
    out, n = C.c_void_p(), C.c_uint64()
    if L.fgx_bgzf_inflate(buf.ctypes.data, buf.size, threads or 0, C.byref(out), C.byref(n)) != 0:
        raise ValueError(L.fgx_bgzf_last_error().decode())
    own = _NativeBuffer(L, out, n.value)
    return own.array(), own


def recompress_file(in_path: str, out_path: str, level: int = 1, threads: Optional[int] = None, chunk_raw_bytes: int = 0) -> int:
    """The host stages of the streaming pipeline (read, block-parallel inflate, deflate, write) around a copy: re-blocks a BGZF
    file.  Returns the number of uncompressed bytes that went through."""
    import ctypes as C
    from ._lib import load
    L = load()
    n = C.c_uint64()
    if L.fgx_bgzf_recompress_file(in_path.encode(), out_path.encode(), threads or 0, level, chunk_raw_bytes, C.byref(n)) != 0:
        raise ValueError(L.fgx_pipeline_last_error().decode())
    return int(n.value)


WRITE_PIECE_BLOCKS = 4096      # write_bam deflates 4096 blocks (≈ 267 MB of records) per native call


def native_deflate(stream, level: int = 1, threads: Optional[int] = None, with_eof: bool = False):
    L = _native()
    if L is None:
        return None
    import ctypes as C
    buf = np.frombuffer(stream, dtype=np.uint8)
    out, n = C.c_void_p(), C.c_uint64()
    if L.fgx_bgzf_deflate(buf.ctypes.data if buf.size else None, buf.size, level, threads or 0, 1 if with_eof else 0, C.byref(out), C.byref(n)) != 0:
        raise ValueError(L.fgx_bgzf_last_error().decode())
    own = _NativeBuffer(L, out, n.value)
    return own.array(), own


def bgzf_compress(stream, level: int = 1, threads: Optional[int] = None) -> List[bytes]:
    """Cuts `stream` (bytes-like) into BGZF blocks; returns them in order (EOF marker not included)."""
    mv = memoryview(stream)
    n = len(mv)
    chunks = [mv[i:i + BGZF_MAX_PAYLOAD] for i in range(0, n, BGZF_MAX_PAYLOAD)]
    if not chunks:
        return []
    with _pool(threads) as ex:
        return list(ex.map(lambda c: _deflate_block(bytes(c), level), chunks, chunksize=max(1, len(chunks) // (8 * (threads or os.cpu_count() or 1)) or 1)))


def bam_header_bytes(text: str, refs: Sequence[Tuple[str, int]]) -> bytes:
    t = text.encode()
    out = [b"BAM\x01", struct.pack("<i", len(t)), t, struct.pack("<i", len(refs))]
    for name, length in refs:
        nb = name.encode() + b"\0"
        out += [struct.pack("<i", len(nb)), nb, struct.pack("<i", length)]
    return b"".join(out)


def consensus_header(read_group_id: str = "A", comment_prefix: str = "Read group", n_input_read_groups: int = 0,
                     command_line: str = "", rg_attrs: Sequence[Tuple[str, str]] = (), program_version: str = "0.0.0") -> str:
    """`create_unmapped_consensus_header` (consensus_runner.rs:130-173): sort order unsorted, group order query, one read
    group with the collapsed attributes of the input read groups, a @PG record (ID / PN / VN / CL, before the comment as the
    reference's writer orders them), the comment line.  `program_version` is the reference build's version string: a whole-file
    comparison against a reference binary needs its VN here; tools/ref_pin.sh compares records only."""
    rg = "@RG\tID:" + read_group_id + "".join(f"\t{k}:{v}" for k, v in rg_attrs)
    lines = ["@HD\tVN:1.6\tSO:unsorted\tGO:query", rg]
    if command_line:      # the noodles writer serialises @PG before @CO; the record carries VN (add_pg_to_builder)
        lines.append(f"@PG\tID:fgumi\tPN:fgumi\tVN:{program_version}\tCL:" + command_line)
    lines.append(f"@CO\t{comment_prefix} {read_group_id} contains consensus reads generated from {n_input_read_groups} input read groups.")
    return "\n".join(lines) + "\n"


def grouped_input_header(refs: Sequence[Tuple[str, int]]) -> str:
    """Header of a grouped (MI-tagged, template-coordinate) input BAM as `fgumi simulate grouped-reads` / `fgumi group` write it."""
    return ("@HD\tVN:1.6\tSO:unsorted\tGO:query\tSS:template-coordinate\n" +
            "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs))


def write_bam(path: str, header_text: str, refs: Sequence[Tuple[str, int]], records, level: int = 1, threads: Optional[int] = None) -> int:
    """Writes header + `records` (a record stream with block_size prefixes) as a BGZF BAM.  The header gets its own blocks,
    the record stream is cut every 0xff00 bytes (records may straddle blocks, as in any BAM).  Returns the file size."""
    hdr_blocks = bgzf_compress(bam_header_bytes(header_text, refs), level, threads)
    if _native() is not None:                 # the library's block-parallel deflate (same framing, checked by tests/test_bgzf.py)
        # in bounded pieces of whole blocks (a piece and its compressed form are all that is held at a time: a 25 GB stream in one call
        # would need twice that in host memory); cutting at multiples of the payload size gives the same blocks as one call
        mv = memoryview(records).cast("B") if not isinstance(records, (bytes, bytearray)) else memoryview(records)
        n, piece = len(mv), WRITE_PIECE_BLOCKS * BGZF_MAX_PAYLOAD
        size = 0
        with open(path, "wb") as f:
            for b in hdr_blocks:
                f.write(b); size += len(b)
            for at in range(0, max(n, 1), piece):
                last = at + piece >= n
                arr, own = native_deflate(mv[at:at + piece], level, threads, with_eof=last)
                f.write(memoryview(arr)); size += int(arr.size)
                del arr, own
        return size
    rec_blocks = bgzf_compress(records, level, threads)
    size = 0
    with open(path, "wb") as f:
        for b in hdr_blocks:
            f.write(b); size += len(b)
        for b in rec_blocks:
            f.write(b); size += len(b)
        f.write(BGZF_EOF); size += len(BGZF_EOF)
    return size


def bgzf_block_table(raw) -> List[Tuple[int, int]]:
    """(offset, total size) of every BGZF block of a file image, by walking the BSIZE chain."""
    mv = memoryview(raw)
    out, p, n = [], 0, len(mv)
    while p < n:
        if n - p < 18 or mv[p] != 0x1F or mv[p + 1] != 0x8B or mv[p + 2] != 8 or not (mv[p + 3] & 4):
            raise ValueError(f"not a BGZF block at offset {p}")
        xlen = mv[p + 10] | (mv[p + 11] << 8)
        q, end, bsize = p + 12, p + 12 + xlen, None
        while q + 4 <= end:
            slen = mv[q + 2] | (mv[q + 3] << 8)
            if q + 4 + slen > end:                      # a subfield that overruns XLEN
                break
            if mv[q] == 0x42 and mv[q + 1] == 0x43 and slen == 2:
                bsize = (mv[q + 4] | (mv[q + 5] << 8)) + 1
            q += 4 + slen
        if bsize is None or p + bsize > n:
            raise ValueError(f"BGZF block at offset {p} has no BC subfield or is truncated")
        out.append((p, bsize))
        p += bsize
    return out


def _inflate_block(mv, off: int, size: int) -> bytes:
    xlen = mv[off + 10] | (mv[off + 11] << 8)
    data = zlib.decompress(bytes(mv[off + 12 + xlen: off + size - 8]), -15)
    crc, isize = struct.unpack_from("<II", mv, off + size - 8)
    if len(data) != isize or (zlib.crc32(data) & 0xFFFFFFFF) != crc:
        raise ValueError(f"BGZF block at offset {off}: CRC32 / ISIZE mismatch")
    return data


def bgzf_decompress(raw, threads: Optional[int] = None) -> bytes:
    mv = memoryview(raw)
    blocks = bgzf_block_table(mv)
    with _pool(threads) as ex:
        parts = list(ex.map(lambda b: _inflate_block(mv, b[0], b[1]), blocks, chunksize=max(1, len(blocks) // (8 * (threads or os.cpu_count() or 1)) or 1)))
    return b"".join(parts)


def record_boundaries(stream, start: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Walks the `block_size` chain of a record stream from `start`: returns (rec_off, rec_len) with rec_off = BODY offsets
    (past the 4-byte prefix), what the engine's entries take.  A sequential chain — the cheap part of FindBoundaries."""
    try:                                  # the library's native chain walk (80 M records in well under a second)
        import ctypes as C
        from ._lib import load
        L = load()
        buf = np.frombuffer(stream, dtype=np.uint8)
        n = C.c_uint64()
        rc = L.fgx_record_boundaries(buf.ctypes.data, buf.size, start, None, None, 0, C.byref(n))
        if rc != 0:
            raise ValueError("record stream ends inside a record")
        off = np.empty(n.value, dtype=np.uint64)
        ln = np.empty(n.value, dtype=np.uint32)
        L.fgx_record_boundaries(buf.ctypes.data, buf.size, start, off.ctypes.data, ln.ctypes.data, n.value, C.byref(n))
        return off, ln
    except (ImportError, OSError, RuntimeError, AttributeError):   # no library (LibraryMissing is a RuntimeError) or a stale one without this entry
        pass
    mv = memoryview(stream)
    n = len(mv)
    offs, lens, p = [], [], start
    unpack = struct.Struct("<I").unpack_from
    while p < n:
        if p + 4 > n:
            raise ValueError("record stream ends inside a block_size prefix")
        (ln,) = unpack(mv, p)
        if p + 4 + ln > n:
            raise ValueError("record stream ends inside a record")
        offs.append(p + 4); lens.append(ln)
        p += 4 + ln
    return np.asarray(offs, dtype=np.uint64), np.asarray(lens, dtype=np.uint32)


def read_bam(path: str, threads: Optional[int] = None):
    """Returns (header_text, refs, stream, rec_off, rec_len): `stream` is the whole uncompressed BAM body; records are
    addressed by (rec_off, rec_len) inside it (no copy of the record bytes)."""
    with open(path, "rb") as f:
        raw = f.read()
    nat = native_inflate(raw, threads)
    data = nat[0].tobytes() if nat is not None else bgzf_decompress(raw, threads)
    if data[:4] != b"BAM\x01":
        raise ValueError("not a BAM file")
    (l_text,) = struct.unpack_from("<i", data, 4)
    text = data[8:8 + l_text].decode()
    p = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, p)
    p += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, p)
        name = data[p + 4:p + 4 + l_name - 1].decode()
        (l_ref,) = struct.unpack_from("<i", data, p + 4 + l_name)
        refs.append((name, l_ref))
        p += 8 + l_name
    rec_off, rec_len = record_boundaries(data, p)
    return text, refs, data, rec_off, rec_len
