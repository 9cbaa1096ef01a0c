"""BGZF / BAM container layer (fgumi_amd/bgzf.py) and tools/export_bam.py: what the whole-BAM pin path writes must be valid
BGZF BAM that a minimal independent reader (gzip members via zlib, BAM spec parsing) takes apart to the same bytes."""
import gzip
import io
import json
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fgumi_amd import bgzf, simulate_grouped_reads  # noqa: E402


def _gunzip_members(raw):
    """Independent reader: a BGZF file is a series of gzip members (RFC 1952); Python's gzip module concatenates them."""
    return gzip.GzipFile(fileobj=io.BytesIO(raw)).read()


def test_bgzf_blocks_are_gzip_members_with_bc_field_and_eof_marker(tmp_path):
    rng = np.random.default_rng(3)
    payload = bytes(rng.integers(0, 4, 300000, dtype=np.uint8))          # compressible, several blocks
    blocks = bgzf.bgzf_compress(payload, level=1, threads=3)
    assert len(blocks) == -(-len(payload) // bgzf.BGZF_MAX_PAYLOAD)
    for b in blocks:
        assert b[:4] == b"\x1f\x8b\x08\x04" and b[12:16] == b"BC\x02\x00"
        assert struct.unpack_from("<H", b, 16)[0] + 1 == len(b)           # BSIZE = total block size - 1
        assert struct.unpack_from("<I", b, len(b) - 4)[0] <= bgzf.BGZF_MAX_PAYLOAD
    raw = b"".join(blocks) + bgzf.BGZF_EOF
    assert _gunzip_members(raw) == payload
    assert bgzf.bgzf_decompress(raw, threads=2) == payload
    assert [s for _, s in bgzf.bgzf_block_table(raw)][-1] == 28 and len(bgzf.BGZF_EOF) == 28
    incompressible = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))  # stored-block fallback keeps every block < 64 KiB
    assert all(len(b) <= 0x10000 for b in bgzf.bgzf_compress(incompressible))
    assert bgzf.bgzf_decompress(b"".join(bgzf.bgzf_compress(incompressible)) + bgzf.BGZF_EOF) == incompressible
    corrupt = bytearray(raw)
    corrupt[40] ^= 0xFF
    with pytest.raises((ValueError, zlib.error)):
        bgzf.bgzf_decompress(bytes(corrupt))


def test_grouped_input_round_trips_through_a_bam_file(tmp_path):
    g = simulate_grouped_reads(300, family_size=1, family_size_max=9)
    refs = [(f"chr{i + 1}", 2147483647) for i in range(24)]
    path = str(tmp_path / "grouped.bam")
    size = bgzf.write_bam(path, bgzf.grouped_input_header(refs), refs, g.blob, threads=2)
    assert size == os.path.getsize(path)
    text, refs2, stream, rec_off, rec_len = bgzf.read_bam(path, threads=2)
    assert refs2 == refs and text.startswith("@HD\tVN:1.6\tSO:unsorted\tGO:query") and text.count("@SQ") == 24
    assert np.array_equal(rec_len, g.rec_len)
    base = int(rec_off[0]) - int(g.rec_off[0])                            # the header shifts every record by the same amount
    assert np.array_equal(rec_off - np.uint64(base), g.rec_off)
    assert stream[int(rec_off[0]) - 4:] == g.blob.tobytes()
    # independent check of the container: plain gzip members, then the BAM magic and n_ref
    raw = open(path, "rb").read()
    body = _gunzip_members(raw)
    assert body == stream and body[:4] == b"BAM\x01" and raw.endswith(bgzf.BGZF_EOF)
    # the chain walk agrees with a pure-Python walk of the block_size prefixes
    p, offs = int(rec_off[0]) - 4, []
    while p < len(body):
        (ln,) = struct.unpack_from("<I", body, p)
        offs.append(p + 4)
        p += 4 + ln
    assert offs == rec_off.tolist()
    with pytest.raises(ValueError):
        bgzf.record_boundaries(body[:-3], int(rec_off[0]) - 4)


def test_consensus_header_is_the_reference_shape():
    h = bgzf.consensus_header("A", "Read group", 2, "fgumi simplex -i g.bam -o o.bam --min-reads 1")
    lines = h.strip().split("\n")
    assert lines[0] == "@HD\tVN:1.6\tSO:unsorted\tGO:query"               # consensus_runner.rs:156-161
    assert lines[1] == "@RG\tID:A"
    # the noodles header writer serialises @HD, @SQ, @RG, @PG, @CO in that order; the @PG record carries ID / PN / VN / CL
    assert lines[2].startswith("@PG\tID:fgumi\tPN:fgumi\tVN:") and "\tCL:fgumi simplex" in lines[2]
    assert lines[3] == "@CO\tRead group A contains consensus reads generated from 2 input read groups."


def test_export_bam_tool_writes_the_pin_input(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_bam.py"), "--families", "200", "--depth", "3", "--input-only",
                          "--out-dir", str(tmp_path), "--threads", "2"], check=True, capture_output=True, text=True).stdout
    rep = json.loads(out.strip().splitlines()[-1])
    text, refs, stream, rec_off, rec_len = bgzf.read_bam(rep["grouped_bam"])
    assert rep["grouped_records"] == len(rec_off) == 1200 and len(refs) == 24
    g = simulate_grouped_reads(200, family_size=3)
    assert stream[int(rec_off[0]) - 4:] == g.blob.tobytes()


def test_native_bgzf_entries_match_the_python_container_code():
    """The library's block-parallel zlib inflate / deflate (csrc/bgzf_host.cpp) against the pure-Python framing above."""
    rng = np.random.default_rng(11)
    payload = bytes(rng.integers(0, 6, 500000, dtype=np.uint8)) + bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    nat = bgzf.native_deflate(payload, level=1, threads=3, with_eof=True)
    assert nat is not None, "libfgumi_amd.so is built in-tree: the native entries must be there"
    blob = nat[0].tobytes()
    assert blob.endswith(bgzf.BGZF_EOF) and _gunzip_members(blob) == payload
    table = bgzf.bgzf_block_table(blob)
    assert len(table) == -(-len(payload) // bgzf.BGZF_MAX_PAYLOAD) + 1 and all(sz <= 0x10000 for _, sz in table)
    assert bgzf.bgzf_decompress(blob, threads=2) == payload                       # Python reader takes the native writer's file
    py_blocks = b"".join(bgzf.bgzf_compress(payload, level=1, threads=2)) + bgzf.BGZF_EOF
    back = bgzf.native_inflate(py_blocks, threads=4)                              # native reader takes the Python writer's file
    assert back[0].tobytes() == payload
    assert bgzf.native_inflate(bgzf.BGZF_EOF)[0].size == 0 and bgzf.native_deflate(b"", with_eof=True)[0].tobytes() == bgzf.BGZF_EOF
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55
    with pytest.raises(ValueError):
        bgzf.native_inflate(bytes(bad))
    with pytest.raises(ValueError):
        bgzf.native_inflate(blob[:-40] + blob[-28:])


def test_record_boundaries_falls_back_without_the_library(monkeypatch):
    """`load()` raises LibraryMissing (a RuntimeError) when libfgumi_amd.so is absent, AttributeError when it is stale: both
    must reach the pure-Python chain walk."""
    from fgumi_amd import _lib
    g = simulate_grouped_reads(20, family_size=2)
    want_off, want_len = np.asarray(g.rec_off, dtype=np.uint64), np.asarray(g.rec_len, dtype=np.uint32)
    stream = bytes(g.blob[:int(want_off[-1]) + int(want_len[-1])])
    for exc in (_lib.LibraryMissing("no library"), AttributeError("fgx_record_boundaries")):
        def boom(*a, _e=exc, **k):
            raise _e
        monkeypatch.setattr(_lib, "load", boom)
        off, ln = bgzf.record_boundaries(stream, 0)
        assert np.array_equal(off, want_off) and np.array_equal(ln, want_len)


def test_block_table_refuses_a_subfield_that_overruns_xlen():
    blk = bytearray(bgzf.bgzf_compress(b"x" * 100, 1, 1)[0])
    # BC subfield header moved so that its payload would lie past XLEN: SI1 SI2 SLEN=2 at end-4 .. end
    xlen = blk[10] | (blk[11] << 8)
    assert xlen == 6
    blk[10] = 4                       # XLEN now ends right behind the subfield HEADER
    with pytest.raises(ValueError):
        bgzf.bgzf_block_table(bytes(blk))


def test_streaming_host_stages_reblock_a_bam_file(tmp_path):
    """fgx_bgzf_recompress_file = the reader / inflate / deflate / writer stages of the streaming pipeline around a copy: what
    comes out must inflate to the same bytes, for chunk sizes that cut the file inside BGZF blocks."""
    g = simulate_grouped_reads(300, family_size=4)
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    refs = [("chr1", 1000000)]
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob)
    want = bgzf.bgzf_decompress(open(src, "rb").read())
    for chunk in (0, 1 << 16):
        n = bgzf.recompress_file(src, dst, level=1, threads=3, chunk_raw_bytes=chunk)
        raw = open(dst, "rb").read()
        assert raw[-28:] == bgzf.BGZF_EOF and n == len(want)
        assert bgzf.bgzf_decompress(raw) == want
    with open(src, "rb") as f:
        cut = f.read()[:-40]
    open(src, "wb").write(cut)
    with pytest.raises(ValueError):
        bgzf.recompress_file(src, dst)


def test_write_bam_deflates_in_bounded_pieces(tmp_path):
    """write_bam hands the native deflate whole-block pieces (so that a 25 GB stream never sits twice in host memory): the file does not
    depend on the piece size."""
    import gzip
    rng = np.random.default_rng(3)
    recs = (rng.integers(0, 4, size=700_000, dtype=np.uint8) + 65).tobytes()
    old = bgzf.WRITE_PIECE_BLOCKS
    try:
        files = []
        for blocks in (4096, 3, 1):
            bgzf.WRITE_PIECE_BLOCKS = blocks
            p = str(tmp_path / f"p{blocks}.bam")
            bgzf.write_bam(p, "@HD\tVN:1.6\n", [], recs)
            files.append(open(p, "rb").read())
        assert files[0] == files[1] == files[2]
        assert gzip.decompress(files[0]).endswith(recs) and files[0].endswith(bgzf.BGZF_EOF)
    finally:
        bgzf.WRITE_PIECE_BLOCKS = old
