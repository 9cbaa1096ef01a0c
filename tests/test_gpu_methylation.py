"""GPU parity: the methylation-aware mode (EM-Seq / TAPs) of the simplex and duplex callers through the C ABI — genome resident in
HBM (`fgx_set_reference`), annotation + normalisation kernel (`k_meth_annotate`) ahead of the column kernel, MM / ML / cu / ct (and the
duplex per-strand am/au/at, bm/bu/bt) assembled with the records — against the oracle, byte for byte: the inputs of the reference's own
unit tests (tests/test_oracle_methylation_pins.py) and seeded EM-Seq-like batches (tests/methsim.py) large enough for the sharded
general path."""
import ctypes as C

import numpy as np
import pytest

import bamutil
import fgx_opts
import methsim
import orc
from fgumi_amd import DuplexConsensusCaller, GroupedReads, MethylationMode, VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, split_records
from fgumi_amd._lib import Options, Output, lib

pytestmark = pytest.mark.gpu


LAST = {}
lib.fgx_debug_last_meth_device.restype = C.c_uint32
lib.fgx_debug_last_meth_device.argtypes = [C.c_void_p]
lib.fgx_debug_last_deferral.restype = None
lib.fgx_debug_last_deferral.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]


def oracle(o, contigs, g, batch_groups=50):
    orc.set_reference(contigs)
    try:
        return orc.process(o, g.blob, g.rec_off, g.rec_len, g.grp_first, batch_groups=batch_groups)
    finally:
        orc.set_reference(None)


def set_reference(h, contigs):
    if not contigs:
        assert lib.fgx_set_reference(h, 0, None, None) == 0
        return
    bufs = [C.create_string_buffer(bytes(s), max(1, len(s))) for s in contigs]
    ptrs = (C.c_void_p * len(bufs))(*[C.cast(b, C.c_void_p).value for b in bufs])
    lens = (C.c_uint64 * len(bufs))(*[len(s) for s in contigs])
    rc = lib.fgx_set_reference(h, len(bufs), ptrs, lens)
    assert rc == 0, lib.fgx_last_error(h).decode()


def product(o, contigs, g, general_only=False):
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h, lib.fgx_global_error().decode()
    try:
        set_reference(h, contigs)
        if general_only:
            lib.fgx_set_general_only(h, 1)
        out = Output()
        rc = lib.fgx_process_batch(h, g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp, C.byref(out))
        assert rc == 0, lib.fgx_last_error(h).decode()
        d2 = (C.c_uint64 * 2)()
        lib.fgx_debug_last_deferral(h, d2)
        LAST.update(meth_device=int(lib.fgx_debug_last_meth_device(h)), deferred=int(d2[0]), groups=int(g.n_grp))   # which way the (last) batch went
        return dict(data=C.string_at(out.data, out.data_len) if out.data_len else b"", count=int(out.count), stats=np.array(list(out.stats), dtype=np.uint64),
                    rejects=C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"", n_rejects=int(out.n_rejects))
    finally:
        lib.fgx_destroy(h)


def same(o, contigs, groups, batch_groups=50, **kw):
    g = groups if isinstance(groups, GroupedReads) else GroupedReads.from_groups(groups)
    want = oracle(o, contigs, g, batch_groups)
    got = product(o, contigs, g, **kw)
    assert got["count"] == want["count"]
    if got["data"] != want["data"]:
        for i, (a, b) in enumerate(zip(split_records(got["data"]), split_records(want["data"]))):
            if a != b:
                raise AssertionError(f"record {i} differs:\n got {bamutil.parse(a)}\nwant {bamutil.parse(b)}")
        raise AssertionError("record count / length differs")
    assert np.array_equal(got["stats"], want["stats"]), (got["stats"].tolist(), want["stats"].tolist())
    if o.track_rejects:
        assert got["rejects"] == want["rejects"] and got["n_rejects"] == want["n_rejects"]
    return want


def opts_from(kw):
    kw = dict(kw)
    mr = kw.pop("duplex_min_reads", None)
    o = fgx_opts.defaults(**kw)
    if mr:
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = mr
    return o


def test_reference_unit_test_inputs():
    """Every caller-level case of the reference's methylation tests (simplex EM-Seq / TAPs, longest-read anchor, non-C reference,
    mode off, BA-only duplex molecules) and of this repo's crafted additions (reverse fragments, pairs, indel anchors, no reference,
    contig outside the header) through the HIP path."""
    import test_oracle_methylation_pins as pins
    cases = pins.replay_cases()
    assert len(cases) >= 18
    n_tagged = 0
    for kw, contigs, groups in cases:
        o = opts_from(kw)
        want = same(o, contigs, groups, batch_groups=100 if kw.get("kind") == 1 else 50)
        n_tagged += sum("cu" in bamutil.parse(r)["tags"] for r in split_records(want["data"]))
    assert n_tagged >= 15


@pytest.mark.parametrize("mode,kw", [
    (1, dict(min_reads=1)), (2, dict(min_reads=1)), (1, dict(min_reads=2, track_rejects=1)), (1, dict(min_reads=1, max_reads=3)),
    (2, dict(min_reads=2, overlapping_consensus=0, produce_per_base_tags=0)), (1, dict(min_reads=1, min_input_base_quality=25, trim=1)),
])
def test_simplex_em_seq_like_batches(mode, kw):
    """1 500 families (fragments of both orientations, pairs, overlapping mates, indel and soft-clipped anchors, reads off the end of a
    contig, a contig outside the header): enough groups for the general path's shards to run on several helper callers, which share
    the one genome in HBM."""
    rng = methsim.seeded(40 + mode)
    contigs = methsim.genome(rng)
    groups = methsim.simplex_groups(rng, contigs, 1500)
    want = same(fgx_opts.defaults(methylation_mode=mode, **kw), contigs, groups)
    recs = [bamutil.parse(r) for r in split_records(want["data"])]
    assert sum("MM" in r["tags"] for r in recs) > 200 and sum("cu" in r["tags"] for r in recs) > 800
    # round 4: the simplex caller's mode runs in the device-resident pipeline (the streaming kernels of simplex_deep.inc) unless --trim or
    # --rejects ask for the general path; families outside their shape (indel / clipped anchors: a third of these groups) are deferred to it
    on_device = not kw.get("trim") and not kw.get("track_rejects")
    assert (LAST["meth_device"] == LAST["groups"] and 0 < LAST["deferred"] < LAST["groups"]) if on_device else LAST["meth_device"] == 0, LAST


def test_simplex_mode_on_the_device_equals_the_general_path(monkeypatch):
    """FGX_METH_DEVICE=0 sends the same batch through the general path: the same bytes, the oracle's."""
    rng = methsim.seeded(401)
    contigs = methsim.genome(rng)
    groups = methsim.simplex_groups(rng, contigs, 900, depth=(1, 40), read_len=(30, 160))
    o = fgx_opts.defaults(methylation_mode=1, min_reads=1)
    same(o, contigs, groups)
    assert LAST["meth_device"] == LAST["groups"], LAST
    monkeypatch.setenv("FGX_METH_DEVICE", "0")
    same(o, contigs, groups)
    assert LAST["meth_device"] == 0, LAST


def test_simplex_mode_in_the_device_resident_entry():
    """fgx_process_batch_device with the mode on (simplex, a reference set): records in HBM in, records with MM / ML / cu / ct in HBM out; the
    families the streaming kernels defer are named in the deferred list (here: none — plain reads only)."""
    import random
    rng = random.Random(5)
    contig = bytes(rng.choice(b"ACGT") for _ in range(3000))
    groups = []
    for gi in range(400):
        pos, L, n = rng.randrange(0, 2700), rng.choice([50, 100, 150]), rng.randrange(1, 12)
        rev = gi % 3 == 0
        reads = []
        for i in range(n):
            seq = bytearray(contig[pos:pos + L])
            for j in range(L):
                if seq[j] == (ord("G") if rev else ord("C")) and rng.random() < 0.7:
                    seq[j] = ord("A") if rev else ord("T")                     # conversion on the read's own strand
                if rng.random() < 0.01:
                    seq[j] = rng.choice(b"ACGT")
            reads.append(bamutil.make_record(f"g{gi}_{i}", seq.decode(), [rng.choice([8, 25, 37]) for _ in range(L)], flag=0x10 if rev else 0, ref_id=0, pos=pos,
                                             tags=[("MI", "Z", str(gi)), ("RX", "Z", "ACGTAC")]))
        groups.append(reads)
    g = GroupedReads.from_groups(groups)
    o = fgx_opts.defaults(min_reads=1, methylation_mode=1)
    want = oracle(o, [contig], g)
    assert b"MM" in want["data"]
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB", methylation_mode=MethylationMode.EmSeq),
                                  overlapping_consensus=True)
    c.set_reference({"chr1": contig}, ["chr1"])
    out = c.process_batch_device(g.to_device())
    assert out.n_deferred == 0 and out.to_host() == want["data"]
    assert np.array_equal(np.array(c.last_stats_array, dtype=np.uint64), want["stats"])
    assert lib.fgx_debug_last_meth_device(c._h) == g.n_grp
    c.close()


@pytest.mark.parametrize("mode,min_reads,kw", [
    (1, (1, 1, 0), {}), (2, (1, 1, 0), {}), (1, (2, 1, 1), dict(track_rejects=1)), (1, (1, 1, 0), dict(duplex_max_reads_per_strand=2)),
    (1, (3, 2, 1), dict(produce_per_base_tags=0, overlapping_consensus=0)),
])
def test_duplex_em_seq_like_batches(mode, min_reads, kw):
    """1 200 duplex molecules: A-only, B-only and two-strand molecules, conversion on the A strand's C's and the B strand's G's (the
    conversion-artifact rule of duplex_consensus), deletions and soft clips in the anchors, a per-strand cap that bites (annotation over
    all reads, consensus over the capped ones)."""
    rng = methsim.seeded(70 + mode)
    contigs = methsim.genome(rng)
    groups = methsim.duplex_groups(rng, contigs, 1200)
    o = fgx_opts.defaults(kind=1, methylation_mode=mode, **kw)
    o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = min_reads
    want = same(o, contigs, groups, batch_groups=100)
    recs = [bamutil.parse(r) for r in split_records(want["data"])]
    assert sum("au" in r["tags"] and "bu" in r["tags"] for r in recs) > 100 and sum("MM" in r["tags"] for r in recs) > 200


def test_small_batch_runs_inline_and_general_only_agrees():
    rng = methsim.seeded(9)
    contigs = methsim.genome(rng, n_contigs=2, length=1500)
    groups = methsim.simplex_groups(rng, contigs, 60)
    o = fgx_opts.defaults(min_reads=1, methylation_mode=1)
    same(o, contigs, groups)
    same(o, contigs, groups, general_only=True)


def test_mode_without_a_reference_and_dropping_the_reference():
    rng = methsim.seeded(21)
    contigs = methsim.genome(rng, n_contigs=1, length=1200)
    groups = methsim.simplex_groups(rng, contigs, 80)
    o = fgx_opts.defaults(min_reads=1, methylation_mode=1)
    want = same(o, None, groups)                 # annotate_and_normalize returns None without a reference: the plain consensus
    assert not any("cu" in bamutil.parse(r)["tags"] for r in split_records(want["data"]))
    g = GroupedReads.from_groups(groups)
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    try:
        out = Output()
        args = (g.blob.ctypes.data, g.blob.size, g.rec_off.ctypes.data, g.rec_len.ctypes.data, g.n_rec, g.grp_first.ctypes.data, g.n_grp)
        set_reference(h, contigs)
        assert lib.fgx_process_batch(h, *args, C.byref(out)) == 0
        with_ref = C.string_at(out.data, out.data_len)
        set_reference(h, None)
        assert lib.fgx_process_batch(h, *args, C.byref(out)) == 0
        assert C.string_at(out.data, out.data_len) == want["data"] and with_ref == oracle(o, contigs, g)["data"] != want["data"]
    finally:
        lib.fgx_destroy(h)


def test_device_entry_and_codec_refuse_the_mode():
    o = fgx_opts.defaults(min_reads=1, methylation_mode=1)
    po = Options.from_buffer_copy(bytes(o))
    h = lib.fgx_create(C.byref(po))
    assert h
    try:
        out, nd, dp = Output(), C.c_uint32(), C.c_void_p()
        rc = lib.fgx_process_batch_device(h, None, 0, None, None, 0, None, 0, C.byref(out), C.byref(nd), C.byref(dp))
        assert rc != 0 and b"methylation" in lib.fgx_last_error(h)
    finally:
        lib.fgx_destroy(h)
    bad = Options.from_buffer_copy(bytes(fgx_opts.defaults(kind=2, methylation_mode=1)))
    assert not lib.fgx_create(C.byref(bad)) and b"CODEC" in lib.fgx_global_error()
    bad = Options.from_buffer_copy(bytes(fgx_opts.defaults(methylation_mode=7)))
    assert not lib.fgx_create(C.byref(bad))


def test_python_mirror_of_set_reference():
    """`VanillaUmiConsensusCaller::set_reference(reference, ref_names)` and `DuplexConsensusCaller::set_reference(reference,
    ref_names, methylation_mode)` as the host mirror spells them."""
    ref = {"chr1": b"N" * 99 + b"CCCCCCCCCC"}
    reads = [bamutil.make_record(f"r{i}", s, [30] * 10, flag=0, ref_id=0, pos=99, tags=[("MI", "Z", "UMI1")]) for i, s in enumerate(["C" * 10, "C" * 10, "T" * 10])]
    c = VanillaUmiConsensusCaller("consensus", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=0, methylation_mode=MethylationMode.EmSeq))
    c.set_reference(ref, ["chr1"])
    rec = bamutil.parse(split_records(c.process_batch(GroupedReads.from_groups([reads])).data)[0])
    c.close()
    assert rec["seq"] == "C" * 10 and rec["tags"]["ML"][1] == [170] * 10 and rec["tags"]["cu"][1] == [2] * 10 and rec["tags"]["ct"][1] == [1] * 10
    ref = {"chr1": b"N" * 99 + b"GGGGGGGGGG"}
    F_PAIRED, F_REVERSE, F_MATE_REVERSE, F_FIRST, F_LAST = 0x1, 0x10, 0x20, 0x40, 0x80
    mol = []
    for i in (1, 2, 3):
        for seq, flag in (("G" * 10, F_PAIRED | F_FIRST | F_REVERSE), ("C" * 10, F_PAIRED | F_LAST | F_MATE_REVERSE)):
            mol.append(bamutil.make_record(f"q{i}", seq, [30] * 10, flag=flag, ref_id=0, pos=99, mapq=0, cigar="10M", mate_ref=0, mate_pos=99, tags=[("MI", "Z", "foo/B"), ("RG", "Z", "A")]))
    d = DuplexConsensusCaller("consensus", "RG1", [1, 1, 0], min_input_base_quality=0, produce_per_base_tags=False)
    d.set_reference(ref, ["chr1"], MethylationMode.EmSeq)
    out = d.process_batch(GroupedReads.from_groups([mol]))
    d.close()
    assert out.count == 2
    for r in map(bamutil.parse, split_records(out.data)):
        assert "bu" in r["tags"] and "bt" in r["tags"] and "au" not in r["tags"]


def test_run_bam_with_the_methylation_aware_mode(tmp_path):
    """BAM file in, consensus BAM file out with `--methylation-mode em-seq` (simplex.rs:240-245): since round 4 the streaming pipeline keeps
    such a caller's batches on the device (the streaming kernels of simplex_deep.inc; the families they defer are resubmitted to the general
    path) — same records as the oracle's, MM / ML / cu / ct included, over one chunk and over many."""
    from fgumi_amd import bgzf
    rng = methsim.seeded(77)
    contigs = methsim.genome(rng, n_contigs=3, length=4000)
    names = [f"chr{i + 1}" for i in range(len(contigs))]
    g = GroupedReads.from_groups(methsim.simplex_groups(rng, contigs, 600))
    o = fgx_opts.defaults(min_reads=1, methylation_mode=int(MethylationMode.EmSeq))
    want = oracle(o, contigs, g, 50)
    assert b"MM" in want["data"] and b"cu" in want["data"]
    refs = [(n, len(s)) for n, s in zip(names, contigs)]
    src, dst = str(tmp_path / "grouped.bam"), str(tmp_path / "consensus.bam")
    bgzf.write_bam(src, bgzf.grouped_input_header(refs), refs, g.blob)
    c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1, min_consensus_base_quality=2, cell_tag="CB", methylation_mode=MethylationMode.EmSeq),
                                  overlapping_consensus=True)
    c.set_reference({n: bytes(s) for n, s in zip(names, contigs)}, names)
    for chunk in (0, 1 << 16):
        st = c.run_bam(src, dst, chunk_raw_bytes=chunk, threads=8)
        text, orefs, stream, off, ln = bgzf.read_bam(dst)
        got = b"".join(bytes(stream[int(o_) - 4:int(o_) + int(l)]) for o_, l in zip(off, ln))
        assert got == want["data"] and st["consensus_records"] == want["count"]
        assert st["stats"][:len(want["stats"])] == [int(v) for v in want["stats"]]
    c.close()
