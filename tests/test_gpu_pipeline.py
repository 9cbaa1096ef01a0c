"""GPU parity of the input side: FindBoundaries on the device against the sequential chain walk, and the
streaming pipeline BAM file -> consensus BAM file against the oracle run over the same records."""
import ctypes as C
import os
import random

import numpy as np
import pytest
import torch

import bamutil
import fgx_opts
import orc
from fgumi_amd import (CodecConsensusCaller, CodecConsensusOptions, DuplexConsensusCaller, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, bgzf,
                       lib, simulate_grouped_reads)

pytestmark = pytest.mark.gpu


def _caller():
    return VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB"), overlapping_consensus=True)


def _device_boundaries(c, stream: bytes, start: int):
    """fgx_record_boundaries_device over `stream` uploaded to HBM: (rec_off, rec_len, consumed, repair rounds are in the caller)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    d = torch.frombuffer(bytearray(stream) + bytearray(64), dtype=torch.uint8).to(dev)
    n, used = C.c_uint64(), C.c_uint64()
    rc = lib.fgx_record_boundaries_device(c._h, d.data_ptr(), len(stream), start, None, None, 0, C.byref(n), C.byref(used))   # count
    assert rc == 0, lib.fgx_last_error(c._h)
    off = torch.empty(max(1, n.value), dtype=torch.int64, device=dev)
    ln = torch.empty(max(1, n.value), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    n2 = C.c_uint64()
    rc = lib.fgx_record_boundaries_device(c._h, d.data_ptr(), len(stream), start, off.data_ptr(), ln.data_ptr(), n.value, C.byref(n2), C.byref(used))
    assert rc == 0 and n2.value == n.value
    return off[:n.value].cpu().numpy().astype(np.uint64), ln[:n.value].cpu().numpy().astype(np.uint32), used.value


def _host_chain(stream: bytes, start: int):
    offs, lens, p = [], [], start
    while p + 4 <= len(stream):
        bs = int.from_bytes(stream[p:p + 4], "little")
        if p + 4 + bs > len(stream):
            break
        offs.append(p + 4); lens.append(bs)
        p += 4 + bs
    return np.asarray(offs, dtype=np.uint64), np.asarray(lens, dtype=np.uint32), p


def test_device_boundaries_equal_the_sequential_chain():
    c = _caller()
    g = simulate_grouped_reads(3000, family_size=2, family_size_max=9)
    stream = bytes(g.blob[:int(g.rec_off[-1]) + int(g.rec_len[-1])])
    # the whole stream; behind a "header" of 1237 bytes; cut inside a record, inside a block_size prefix, and right at a record's end
    for prefix, cut in ((0, 0), (1237, 0), (0, 100), (0, int(g.rec_len[-1]) + 2), (5, int(g.rec_len[-1]) + 4)):
        s = bytes(prefix) + (stream[:len(stream) - cut] if cut else stream)
        off, ln, used = _device_boundaries(c, s, prefix)
        woff, wln, wused = _host_chain(s, prefix)
        assert np.array_equal(off, woff) and np.array_equal(ln, wln) and used == wused
    c.close()


def test_device_boundaries_with_records_longer_than_a_segment_and_record_like_payloads():
    """Segments without any record start (records of 40 KB), and payload bytes that look like record headers: the walks are checked
    against each other, so a wrong guess costs a repair round and nothing else."""
    rng = random.Random(11)
    recs = []
    fake = bamutil.make_record("fake", "ACGT" * 10, [30] * 40, flag=0x4, ref_id=-1, pos=-1)
    fake = len(fake).to_bytes(4, "little") + fake
    for i in range(400):
        L = 30000 if i % 37 == 5 else rng.choice([50, 150, 151, 300])
        seq = "".join(rng.choice("ACGT") for _ in range(L))
        tags = [("MI", "Z", str(i // 4))]
        if i % 11 == 3:                                   # a B:C array whose bytes are three whole fake records
            tags.append(("zz", "raw", b"BC" + (3 * len(fake)).to_bytes(4, "little") + fake * 3))
        recs.append(bamutil.make_record("r%05d" % i, seq, [rng.randrange(2, 41) for _ in range(L)], flag=0x4D, ref_id=0, pos=100 + i, tags=tags))
    stream = b"".join(len(r).to_bytes(4, "little") + r for r in recs)
    c = _caller()
    off, ln, used = _device_boundaries(c, stream, 0)
    woff, wln, wused = _host_chain(stream, 0)
    assert np.array_equal(off, woff) and np.array_equal(ln, wln) and used == wused == len(stream)
    c.close()


def _run_and_compare(tmp_path, caller, opts, g, batch_groups, chunk, **run_kw):
    refs = [("chr%d" % (i + 1), 2147483647) for i in range(24)]
    src, dst = str(tmp_path / "grouped.bam"), str(tmp_path / "consensus.bam")
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob)
    st = caller.run_bam(src, dst, chunk_raw_bytes=chunk, threads=8, **run_kw)
    want = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    text, orefs, stream, off, ln = bgzf.read_bam(dst)
    got = b"".join(bytes(stream[int(o) - 4:int(o) + int(l)]) for o, l in zip(off, ln))
    assert text.startswith("@HD\tVN:1.6\tSO:unsorted\tGO:query") and orefs == []
    assert st["consensus_records"] == want["count"] == len(off)
    assert got == want["data"], "the consensus BAM's records differ from the oracle's"
    assert st["stats"][:len(want["stats"])] == [int(v) for v in want["stats"]]
    assert st["groups"] == len(g.grp_first) - 1 and st["kept_records"] == g.n_rec
    return st


@pytest.mark.parametrize("host_inflate", [False, True])
def test_bam_file_to_consensus_bam_file_simplex(tmp_path, host_inflate):
    """BGZF inflate on the device (one lane per block, CRC-32 checked there) and on the host cores: the same records either way."""
    g = simulate_grouped_reads(6000, family_size=2, family_size_max=12)
    c = _caller()
    one = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 0, host_inflate=host_inflate)               # one chunk
    many = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 16, host_inflate=host_inflate)        # 64 KiB of compressed bytes per chunk: groups cross chunks
    assert one["chunks"] == 1 and many["chunks"] > 20
    assert one["device_inflate"] == (0 if host_inflate else 1)
    c.close()


@pytest.mark.parametrize("host_inflate", [False, True])
def test_a_leftover_larger_than_the_front_pad_widens_it(tmp_path, monkeypatch, host_inflate):
    """The next chunk is uploaded and inflated behind a front pad while this one is worked on; what this chunk leaves over (its last MI
    group) goes into that pad.  With a pad of 256 bytes every leftover is larger than it: the stream already in place moves behind a
    wider pad, and the records stay the oracle's."""
    monkeypatch.setenv("FGX_FRONT_PAD", "256")
    g = simulate_grouped_reads(4000, family_size=3, family_size_max=40)
    c = _caller()                                             # (a fresh caller: the pad belongs to its pipeline state)
    st = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 16, host_inflate=host_inflate)
    assert st["chunks"] > 10
    again = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 17, host_inflate=host_inflate)   # the same state, other chunk sizes
    assert again["chunks"] > 5
    c.close()


def test_bam_file_to_consensus_bam_file_with_device_deflate(tmp_path):
    """The output side on the device as well (deflate_core.h, a lane per BGZF block, CRC-32 by a wavefront per block): the file must be a
    valid BGZF BAM whose records equal the oracle's."""
    g = simulate_grouped_reads(6000, family_size=2, family_size_max=12)
    c = _caller()
    st = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 20, device_deflate=True)
    assert st["device_deflate"] == 1 and st["chunks"] > 1
    raw = open(str(tmp_path / "consensus.bam"), "rb").read()
    assert raw[-28:] == bgzf.BGZF_EOF
    blocks = bgzf.bgzf_block_table(raw)
    assert len(blocks) > 5 and all(size <= 65536 for _, size in blocks)
    c.close()


def test_device_inflate_refuses_a_corrupted_block(tmp_path):
    g = simulate_grouped_reads(500, family_size=3)
    refs = [("chr1", 1000000)]
    src = str(tmp_path / "in.bam")
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob)
    raw = bytearray(open(src, "rb").read())
    blocks = bgzf.bgzf_block_table(bytes(raw))
    off, size = blocks[len(blocks) // 2]
    raw[off + 18 + (size - 26) // 2] ^= 0x10                     # one bit inside the DEFLATE payload of a middle block
    open(src, "wb").write(raw)
    c = _caller()
    with pytest.raises(RuntimeError, match="failed to inflate on the device"):
        c.run_bam(src, str(tmp_path / "out.bam"))
    c.close()


def test_bam_file_to_consensus_bam_file_duplex_and_codec(tmp_path):
    g = simulate_grouped_reads(1500, family_size=6, duplex=1)
    o = fgx_opts.defaults(kind=1)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = 1, 1, 0
    c = DuplexConsensusCaller("", "A", [1, 1, 0], cell_tag="CB", overlapping_consensus=True)
    _run_and_compare(tmp_path, c, o, g, 100, 1 << 17, strip_strand_suffix=True)
    c.close()
    g = simulate_grouped_reads(800, family_size=3, read_length=300, insert_mean=350, insert_sd=60, codec=1)
    o = fgx_opts.defaults(kind=2, overlapping_consensus=0, cell_tag=b"\0\0", produce_per_base_tags=1)
    c = CodecConsensusCaller("", "A", CodecConsensusOptions(produce_per_base_tags=True))
    _run_and_compare(tmp_path, c, o, g, 1000, 1 << 17, cell_tag=None)
    c.close()


def test_run_bam_reports_a_truncated_file(tmp_path):
    g = simulate_grouped_reads(200, family_size=3)
    refs = [("chr1", 1000000)]
    src = str(tmp_path / "in.bam")
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob[:len(g.blob) - 37])    # the stream ends inside the last record
    c = _caller()
    with pytest.raises(RuntimeError, match="ends inside a record"):
        c.run_bam(src, str(tmp_path / "out.bam"))
    c.close()


def test_run_bam_edge_files(tmp_path):
    """A header-only file; one family of 600 records (300 per end: more than any device kernel takes — the streaming kernels of
    simplex_deep.inc stop at 255 retained reads per end — so the batch goes through the host entry, and with 64 KiB chunks the group is the
    leftover of chunk after chunk until the file ends); records without the group tag in between."""
    refs = [("chr1", 1000000)]
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    c = _caller()
    # header only
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, b"")
    st = c.run_bam(src, dst)
    text, orefs, stream, off, ln = bgzf.read_bam(dst)
    assert st["consensus_records"] == 0 and len(off) == 0 and text.startswith("@HD")
    # one big family
    g = simulate_grouped_reads(1, family_size=300)
    st = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 16)
    assert st["chunks"] >= 1 and st["deferred_groups"] == 1
    # ... and one of 300 records: the device's since round 4
    g = simulate_grouped_reads(1, family_size=150)
    st = _run_and_compare(tmp_path, c, fgx_opts.defaults(min_reads=1), g, 50, 1 << 16)
    assert st["chunks"] >= 1 and st["deferred_groups"] == 0
    # untagged records between the families: MiGrouper skips them (mi_group.rs:285-290)
    g = simulate_grouped_reads(400, family_size=3)
    recs = [bytes(g.blob[int(o) - 4:int(o) + int(l)]) for o, l in zip(g.rec_off, g.rec_len)]
    plain = bamutil.make_record("untagged", "ACGT" * 10, [30] * 40, flag=0, ref_id=0, pos=5)
    plain = len(plain).to_bytes(4, "little") + plain
    mixed = b"".join(r + (plain if i % 7 == 3 else b"") for i, r in enumerate(recs))
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, mixed)
    st = c.run_bam(src, dst, chunk_raw_bytes=1 << 16)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    text, orefs, stream, off, ln = bgzf.read_bam(dst)
    got = b"".join(bytes(stream[int(o) - 4:int(o) + int(l)]) for o, l in zip(off, ln))
    assert got == want["data"] and st["kept_records"] == g.n_rec
    c.close()


def test_chunks_that_keep_no_record_at_all(tmp_path):
    """Whole chunks of records the grouper drops (unmapped, untagged): the first chunk holds the BAM header and nothing kept, later chunks
    hold dropped records only.  Nothing but the partial record behind the last whole one may be carried into the next chunk — not the
    header, not the alignment bytes in front of the leftover (`batch_end = 0` carried the whole chunk)."""
    refs = [("chr1", 1000000)]
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    g = simulate_grouped_reads(300, family_size=3)
    recs = [bytes(g.blob[int(o) - 4:int(o) + int(l)]) for o, l in zip(g.rec_off, g.rec_len)]
    rng = random.Random(5)

    def dropped(i):
        seq = "".join(rng.choice("ACGT") for _ in range(150))
        r = bamutil.make_record(f"drop{i:06d}", seq, [rng.randrange(2, 41) for _ in range(150)], flag=4 if i % 2 else 0, ref_id=-1 if i % 2 else 0, pos=-1 if i % 2 else 7,
                                tags=() if i % 3 else (("RX", "Z", "ACGT"),))
        return len(r).to_bytes(4, "little") + r
    run = b"".join(dropped(i) for i in range(2500))                   # ~ 600 KB of records that all go away: several 32 KiB chunks (random bases: they do not compress away)
    mixed = run + b"".join(recs[:len(recs) // 2]) + run + b"".join(recs[len(recs) // 2:]) + run
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, mixed)
    c = _caller()
    st = c.run_bam(src, dst, chunk_raw_bytes=1 << 15)
    want = orc.process(fgx_opts.defaults(min_reads=1), g.blob, g.rec_off, g.rec_len, g.grp_first)
    text, orefs, stream, off, ln = bgzf.read_bam(dst)
    got = b"".join(bytes(stream[int(o) - 4:int(o) + int(l)]) for o, l in zip(off, ln))
    assert st["chunks"] > 12
    assert got == want["data"] and st["kept_records"] == g.n_rec and st["groups"] == g.n_grp
    c.close()


def _rejects_case(tmp_path, caller, opts, g, batch_groups, chunk, **run_kw):
    """fgx_run_bam_rejects: consensus BAM == the oracle's records, rejects BAM == the oracle's rejects (batch-input order) under the INPUT header."""
    refs = [("chr%d" % (i + 1), 2147483647) for i in range(24)]
    src, dst, rej = str(tmp_path / "grouped.bam"), str(tmp_path / "consensus.bam"), str(tmp_path / "rejects.bam")
    in_text = bgzf.grouped_input_header(refs)
    bgzf.write_bam(src, in_text, refs, g.blob)
    st = caller.run_bam(src, dst, chunk_raw_bytes=chunk, threads=8, rejects_path=rej, **run_kw)
    want = orc.process(opts, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    text, orefs, stream, off, ln = bgzf.read_bam(dst)
    got = b"".join(bytes(stream[int(o) - 4:int(o) + int(l)]) for o, l in zip(off, ln))
    assert got == want["data"], "the consensus BAM's records differ from the oracle's"
    rtext, rrefs, rstream, roff, rln = bgzf.read_bam(rej)
    rgot = b"".join(bytes(rstream[int(o) - 4:int(o) + int(l)]) for o, l in zip(roff, rln))
    assert rtext == in_text and [n for n, _ in rrefs] == [n for n, _ in refs], "the rejects BAM must advertise the input header"
    assert want["n_rejects"] > 0 and len(roff) == want["n_rejects"] == st["rejected_records"]
    assert rgot == want["rejects"], "the rejects BAM's records differ from the oracle's rejects"
    assert st["stats"][:len(want["stats"])] == [int(v) for v in want["stats"]]
    return st


def test_run_bam_with_rejects_simplex(tmp_path):
    """`--rejects` through the streaming pipeline (simplex.rs:7-12, 260-285): groups below --min-reads leave their input records, the caller's
    rejects are the overlap-corrected copies; one chunk and many (groups cross chunks, rejects of successive batches keep input order)."""
    g = simulate_grouped_reads(3000, family_size=1, family_size_max=9)
    for kw in (dict(min_reads=2), dict(min_reads=3, max_reads=4)):
        c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_consensus_base_quality=2, cell_tag="CB", **kw), track_rejects=True, overlapping_consensus=True)
        o = fgx_opts.defaults(track_rejects=1, **kw)
        one = _rejects_case(tmp_path, c, o, g, 50, 0)
        many = _rejects_case(tmp_path, c, o, g, 50, 1 << 16)
        assert one["chunks"] == 1 and many["chunks"] > 10
        c.close()


def test_run_bam_with_rejects_takes_the_host_entry_when_the_side_kernels_refuse(tmp_path):
    """A family of 600 records is outside the reject side kernels' scope (and the device pipeline's): that batch goes through the host entry
    in one piece, rejects included; and the duplex caller, whose `--rejects` the device entry does not serve, does so for every batch."""
    big = simulate_grouped_reads(1, family_size=300)
    small = simulate_grouped_reads(300, family_size=1, family_size_max=5, seed=7)
    recs = [small.records(i) for i in range(150)] + [big.records(0)] + [small.records(i) for i in range(150, 300)]
    from fgumi_amd import GroupedReads
    g = GroupedReads.from_groups(recs)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=2, min_consensus_base_quality=2, cell_tag="CB"), track_rejects=True, overlapping_consensus=True)
    _rejects_case(tmp_path, c, fgx_opts.defaults(track_rejects=1, min_reads=2), g, 50, 1 << 16)
    c.close()
    gd = simulate_grouped_reads(400, family_size=1, family_size_max=6, duplex=1)
    od = fgx_opts.defaults(kind=1, track_rejects=1)
    od.duplex_min_reads[0], od.duplex_min_reads[1], od.duplex_min_reads[2] = 3, 2, 1
    cd = DuplexConsensusCaller("", "A", [3, 2, 1], cell_tag="CB", overlapping_consensus=True, track_rejects=True)
    _rejects_case(tmp_path, cd, od, gd, 100, 1 << 16, strip_strand_suffix=True)
    cd.close()


def test_run_bam_with_rejects_duplex_and_codec_on_the_device(tmp_path):
    """VERDICT r5 item 7 (the --rejects half): `fgumi duplex --rejects` / `fgumi codec --rejects` through the streaming pipeline WITHOUT the general path —
    the device entry decides every batch (no deferred group, no batch through the host entry) and its side kernels write the rejects: shallow molecules whole
    (/A records, then /B), the zero-length reads of kept ones; both files equal the oracle's."""
    import test_gpu_zz_rejects_device as tgr
    from fgumi_amd import CodecConsensusCaller, CodecConsensusOptions
    # (the opt-outs send every batch through the host entry, and so does tests/apiemu's stand-in pipeline when it defers groups by rule: same files)
    served = os.environ.get("FGX_REJECTS_DEVICE") != "0" and os.environ.get("FGX_OPT_IN_ALL") != "0" and os.environ.get("APIEMU_DEFER") in (None, "none", "indel")
    gd = tgr.strand_batch("duplex", 31)
    od = tgr.strand_options("duplex", dict(duplex_min_reads=(3, 2, 1)))
    cd = DuplexConsensusCaller("", "A", [3, 2, 1], cell_tag="CB", overlapping_consensus=True, track_rejects=True)
    for chunk in (0, 1 << 16):
        st = _rejects_case(tmp_path, cd, od, gd, 100000, chunk, strip_strand_suffix=True)
        assert st["deferred_groups"] == 0 and (st["host_entry_batches"] == 0 or not served), (st["deferred_groups"], st["host_entry_batches"])
    cd.close()
    gc = tgr.strand_batch("codec", 32)
    oc = tgr.strand_options("codec", dict(codec_min_reads_per_strand=2))
    cc = CodecConsensusCaller("", "A", CodecConsensusOptions(min_reads_per_strand=2, produce_per_base_tags=True, cell_tag="CB"), track_rejects=True)
    st = _rejects_case(tmp_path, cc, oc, gc, 100000, 0)
    assert st["deferred_groups"] == 0 and (st["host_entry_batches"] == 0 or not served), (st["deferred_groups"], st["host_entry_batches"])
    cc.close()
