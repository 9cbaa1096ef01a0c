"""ctypes binding of the ORACLE (test infrastructure only; never imported by the product)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(_SO)
    D, U8, U32, I32, U64, VP, CP = C.c_double, C.c_uint8, C.c_uint32, C.c_int32, C.c_uint64, C.c_void_p, C.c_char_p
    P = C.POINTER
    sig = {
        "orc_last_error": (CP, []),
        "orc_process": (VP, [VP, VP, VP, VP, U32, VP, U32, U32, U32]),
        "orc_result_data": (VP, [VP]), "orc_result_len": (U64, [VP]), "orc_result_count": (U64, [VP]),
        "orc_result_stats": (None, [VP, VP]), "orc_result_rejects": (VP, [VP]), "orc_result_rejects_len": (U64, [VP]),
        "orc_result_n_rejects": (U64, [VP]), "orc_result_free": (None, [VP]), "orc_result_seconds": (C.c_double, [VP]),
        "orc_phred_to_ln_error_prob": (D, [U8]), "orc_phred_to_ln_correct_prob": (D, [U8]), "orc_ln_prob_to_phred": (U8, [D]),
        "orc_log1pexp": (D, [D]), "orc_ln_sum_exp": (D, [D, D]), "orc_ln_sum_exp_array": (D, [VP, U32]), "orc_ln_not": (D, [D]),
        "orc_ln_error_prob_two_trials": (D, [D, D]), "orc_ln_a_minus_b": (C.c_int, [D, D, P(D)]),
        "orc_fgbio_unique_max_index": (C.c_int, [VP]), "orc_unique_max_index": (C.c_int, [VP]),
        "orc_consensus_error": (D, [D]), "orc_unanimous_quality_from_gap": (U8, [D, U8]), "orc_unanimous_margin": (D, [D, D, D]),
        "orc_builder_new": (VP, [U8, U8, C.c_int]), "orc_builder_free": (None, [VP]), "orc_builder_reset": (None, [VP]),
        "orc_builder_add": (None, [VP, U8, U8]), "orc_builder_add_n": (None, [VP, U8, U8, U32]),
        "orc_builder_call": (None, [VP, P(U8), P(U8)]), "orc_builder_call_full": (None, [VP, P(U8), P(U8)]),
        "orc_builder_fast_path": (C.c_int, [VP, P(U8), P(U8)]), "orc_builder_contributions": (U32, [VP]),
        "orc_builder_observations_for_base": (U32, [VP, U8]), "orc_builder_likelihoods": (None, [VP, VP]),
        "orc_builder_set_likelihoods": (None, [VP, VP, VP]), "orc_builder_table": (None, [VP, C.c_int, VP, P(U32)]),
        "orc_call_columns": (None, [U8, U8, C.c_int, VP, VP, U32, U32, VP, VP, VP, VP]),
        "orc_single_input_quals": (None, [VP, VP]), "orc_read_name_rank": (I32, [CP, U32]), "orc_mate_clip": (U64, [VP, U32]),
        "orc_mate_clip_ops": (U64, [C.c_int, I32, VP, U32, I32, VP, U32]), "orc_parse_mc": (C.c_int, [CP, VP]),
        "orc_quality_trim_point": (U32, [VP, U32, U8]), "orc_consensus_umis": (C.c_int, [CP, VP, U32]),
        "orc_overlap_pair": (C.c_int, [VP, U32, VP, U32, VP]),
        "orc_apply_overlapping": (None, [VP, VP, VP, U32, VP]),
        "orc_clip_cigar_ops": (U32, [VP, U32, U32, C.c_int, VP, U32, VP]), "orc_read_pos_at_ref_pos": (U64, [VP, U32, U64, U64, C.c_int]),
        "orc_duplex_consensus": (C.c_int, [VP, VP, VP, VP, U32, VP, VP, VP, VP, U32, VP, VP, VP, U32, P(U32)]), "orc_duplex_cap_quality": (U8, [I32]),
        "orc_sweep_fast_vs_full": (U64, [C.c_int, P(U64)]),
        "orc_group_records": (U32, [VP, VP, VP, VP, U32, VP, VP, VP, P(U32)]),
        "orc_filter_records": (VP, [VP, VP, U64, VP, VP, U32]), "orc_filter_data": (VP, [VP]), "orc_filter_data_len": (U64, [VP]),
        "orc_filter_rejects": (VP, [VP]), "orc_filter_rejects_len": (U64, [VP]), "orc_filter_counts": (None, [VP, VP]), "orc_filter_free": (None, [VP]),
        "orc_filter_mask_bases": (C.c_int64, [VP, U32, VP, C.c_int, U8]), "orc_filter_mask_duplex_bases": (C.c_int64, [VP, U32, VP, VP, VP, C.c_int, U8, C.c_int]),
        "orc_filter_read": (C.c_int, [VP, U32, VP]), "orc_filter_duplex_read": (C.c_int, [VP, U32, VP, VP, VP]), "orc_filter_is_duplex": (C.c_int, [VP, U32]),
        "orc_filter_process_record": (C.c_int, [VP, VP, U32, P(U64), P(C.c_int)]), "orc_filter_reverse_tags": (None, [VP, U32]),
        "orc_set_reference": (None, [U32, VP, VP]),
        "orc_meth_query_to_ref_positions": (U32, [VP, U32, C.c_int64, C.c_int, VP, U32, VP, U32]), "orc_meth_is_cpg_context": (C.c_int, [CP, U64, U64, C.c_int]),
        "orc_meth_is_top_strand": (C.c_int, [C.c_uint16]), "orc_meth_annotate": (None, [U32, VP, VP, U32, VP, U32, C.c_int, VP, VP, VP]),
        "orc_meth_build_mm_ml": (C.c_int, [VP, U32, U32, VP, VP, VP, C.c_int, C.c_int, VP, U32, VP, U32]),
        "orc_meth_combine": (None, [U32, VP, VP, VP, U32, VP, VP, VP, U32, VP, VP, VP]),
    }
    for name, (res, args) in sig.items():
        if hasattr(lib, name):
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
    return lib


lib = _load()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Builder:
    def __init__(self, pre, post, tie=0):
        self.h = lib.orc_builder_new(pre, post, tie)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_builder_free(self.h)
            self.h = None

    def reset(self):
        lib.orc_builder_reset(self.h)

    def add(self, base, qual, n=1):
        b = ord(base) if isinstance(base, str) else base
        lib.orc_builder_add_n(self.h, b, qual, n)

    def _bq(self, fn):
        b, q = C.c_uint8(), C.c_uint8()
        r = fn(self.h, C.byref(b), C.byref(q))
        return r, chr(b.value), q.value

    def call(self):
        _, b, q = self._bq(lib.orc_builder_call)
        return b, q

    def call_full(self):
        _, b, q = self._bq(lib.orc_builder_call_full)
        return b, q

    def fast_path(self):
        r, b, q = self._bq(lib.orc_builder_fast_path)
        return (b, q) if r else None

    def contributions(self):
        return lib.orc_builder_contributions(self.h)

    def observations_for_base(self, base):
        return lib.orc_builder_observations_for_base(self.h, ord(base))

    def likelihoods(self):
        a = np.zeros(4)
        lib.orc_builder_likelihoods(self.h, ptr(a))
        return a

    def table(self, which):
        a = np.zeros(94)
        cap = C.c_uint32()
        lib.orc_builder_table(self.h, which, ptr(a), C.byref(cap))
        return a, cap.value


def _bytes_at(address, n):
    """`n` bytes at `address` (ctypes.string_at takes its size as a C int: lengths of 2 GiB and more come out truncated)."""
    if not n:
        return b""
    return C.string_at(address, n) if n < (1 << 31) else bytes((C.c_char * n).from_address(address))


def process(opts, blob, rec_off, rec_len, grp_first, batch_groups=50, threads=1):
    """Run the oracle over a whole input. Returns dict(data=bytes, count, stats=np.uint64[28], rejects, n_rejects)."""
    n_rec = len(rec_off)
    n_grp = len(grp_first) - 1
    h = lib.orc_process(C.addressof(opts), ptr(blob), ptr(rec_off), ptr(rec_len), n_rec, ptr(grp_first), n_grp, batch_groups, threads)
    if not h:
        raise RuntimeError("oracle error: " + lib.orc_last_error().decode())
    try:
        n = lib.orc_result_len(h)
        data = _bytes_at(lib.orc_result_data(h), n)
        stats = np.zeros(28, dtype=np.uint64)
        lib.orc_result_stats(h, ptr(stats))
        rn = lib.orc_result_rejects_len(h)
        rej = _bytes_at(lib.orc_result_rejects(h), rn)
        return dict(data=data, count=lib.orc_result_count(h), stats=stats, rejects=rej, n_rejects=lib.orc_result_n_rejects(h),
                    seconds_workers=lib.orc_result_seconds(h))
    finally:
        lib.orc_result_free(h)


class GroupOptions(C.Structure):
    _fields_ = [("tag", C.c_char * 2), ("cell_tag", C.c_char * 2), ("strip_strand_suffix", C.c_uint8), ("allow_unmapped", C.c_uint8), ("_pad", C.c_uint8 * 2)]


def group_records(blob, rec_off, rec_len, tag=b"MI", cell_tag=b"CB", strip_strand_suffix=False, allow_unmapped=False):
    """MiGrouper restatement: returns (kept rec_off, kept rec_len, grp_first)."""
    o = GroupOptions(tag, cell_tag or b"\0\0", int(strip_strand_suffix), int(allow_unmapped))
    n = len(rec_off)
    out_off = np.zeros(max(1, n), dtype=np.uint64)
    out_len = np.zeros(max(1, n), dtype=np.uint32)
    grp = np.zeros(n + 1, dtype=np.uint32)
    nk = C.c_uint32()
    ng = lib.orc_group_records(C.addressof(o), ptr(blob), ptr(rec_off), ptr(rec_len), n, ptr(out_off), ptr(out_len), ptr(grp), C.byref(nk))
    return out_off[:nk.value].copy(), out_len[:nk.value].copy(), grp[:ng + 1].copy()


class FilterOptions(C.Structure):
    """include/fgumi_amd.h fgx_filter_options."""
    _fields_ = [("struct_size", C.c_uint32), ("min_reads", C.c_uint32 * 3), ("max_read_error_rate", C.c_double * 3), ("max_base_error_rate", C.c_double * 3),
                ("min_mean_base_quality", C.c_double), ("max_no_call_fraction", C.c_double), ("has_min_base_quality", C.c_uint8), ("min_base_quality", C.c_uint8),
                ("has_min_mean_base_quality", C.c_uint8), ("require_single_strand_agreement", C.c_uint8), ("reverse_per_base_tags", C.c_uint8),
                ("filter_by_template", C.c_uint8), ("track_rejects", C.c_uint8), ("regenerate_alignment_tags", C.c_uint8),
                ("has_min_methylation_depth", C.c_uint8), ("require_strand_methylation_agreement", C.c_uint8), ("has_min_conversion_fraction", C.c_uint8),
                ("methylation_mode", C.c_uint8), ("min_methylation_depth", C.c_uint32 * 3), ("min_conversion_fraction", C.c_double)]


def _three(v):
    v = list(v) if isinstance(v, (list, tuple)) else [v]
    return (v + [v[-1]] * 3)[:3]          # expand_three_from_last (filter.rs:20-27)


def filter_options(min_reads=1, max_read_error_rate=0.025, max_base_error_rate=0.1, min_base_quality=None, min_mean_base_quality=None,
                   max_no_call_fraction=0.2, require_single_strand_agreement=False, reverse_per_base_tags=False, filter_by_template=True,
                   track_rejects=False, regenerate_alignment_tags=False, min_methylation_depth=None, require_strand_methylation_agreement=False,
                   min_conversion_fraction=None, methylation_mode=0):
    o = FilterOptions()
    o.struct_size = C.sizeof(FilterOptions)
    o.min_reads[:] = _three(min_reads)
    o.max_read_error_rate[:] = _three(max_read_error_rate)
    o.max_base_error_rate[:] = _three(max_base_error_rate)
    o.has_min_base_quality, o.min_base_quality = int(min_base_quality is not None), int(min_base_quality or 0)
    o.has_min_mean_base_quality, o.min_mean_base_quality = int(min_mean_base_quality is not None), float(min_mean_base_quality or 0.0)
    o.max_no_call_fraction = max_no_call_fraction
    o.require_single_strand_agreement, o.reverse_per_base_tags = int(require_single_strand_agreement), int(reverse_per_base_tags)
    o.filter_by_template, o.track_rejects = int(filter_by_template), int(track_rejects)
    o.regenerate_alignment_tags = int(regenerate_alignment_tags)
    o.has_min_methylation_depth = int(min_methylation_depth is not None)
    if min_methylation_depth is not None:
        o.min_methylation_depth[:] = _three(min_methylation_depth)      # MethylationDepthThresholds::from_values (filter.rs:944-954)
    o.require_strand_methylation_agreement = int(require_strand_methylation_agreement)
    o.has_min_conversion_fraction, o.min_conversion_fraction = int(min_conversion_fraction is not None), float(min_conversion_fraction or 0.0)
    o.methylation_mode = int(methylation_mode)                           # FGX_METHYLATION_*: 0 disabled, 1 em-seq, 2 taps
    return o


def _thr(t):
    return (C.c_double * 3)(float(t[0]), float(t[1]), float(t[2]))


def filter_records(o, blob, rec_off, rec_len):
    """`fgumi filter` restatement over a record stream: dict(data, rejects, records, passed, masked, rejected)."""
    r = lib.orc_filter_records(C.addressof(o), ptr(blob), len(blob), ptr(rec_off), ptr(rec_len), len(rec_off))
    if not r:
        raise RuntimeError(lib.orc_last_error().decode())
    try:
        cnt = (C.c_uint64 * 4)()
        lib.orc_filter_counts(r, cnt)
        return dict(data=C.string_at(lib.orc_filter_data(r), lib.orc_filter_data_len(r)), rejects=C.string_at(lib.orc_filter_rejects(r), lib.orc_filter_rejects_len(r)),
                    records=cnt[0], passed=cnt[1], masked=cnt[2], rejected=cnt[3])
    finally:
        lib.orc_filter_free(r)


def filter_mask_bases(rec, thr, min_base_quality=None):
    buf = bytearray(rec)
    a = (C.c_uint8 * len(buf)).from_buffer(buf)
    n = lib.orc_filter_mask_bases(a, len(buf), _thr(thr), int(min_base_quality is not None), int(min_base_quality or 0))
    return n, bytes(buf)


def filter_mask_duplex_bases(rec, cc, ab, ba, min_base_quality=None, ss_agreement=False):
    buf = bytearray(rec)
    a = (C.c_uint8 * len(buf)).from_buffer(buf)
    n = lib.orc_filter_mask_duplex_bases(a, len(buf), _thr(cc), _thr(ab), _thr(ba), int(min_base_quality is not None), int(min_base_quality or 0), int(ss_agreement))
    return n, bytes(buf)


def filter_read(rec, thr):
    return lib.orc_filter_read(rec, len(rec), _thr(thr))


def filter_duplex_read(rec, cc, ab, ba):
    return lib.orc_filter_duplex_read(rec, len(rec), _thr(cc), _thr(ab), _thr(ba))


def filter_is_duplex(rec):
    return bool(lib.orc_filter_is_duplex(rec, len(rec)))


def filter_process_record(o, rec):
    buf = bytearray(rec)
    a = (C.c_uint8 * len(buf)).from_buffer(buf)
    masked, ok = C.c_uint64(), C.c_int()
    if lib.orc_filter_process_record(C.addressof(o), a, len(buf), C.byref(masked), C.byref(ok)) != 0:
        raise RuntimeError(lib.orc_last_error().decode())
    return masked.value, bool(ok.value), bytes(buf)


def filter_reverse_tags(rec):
    buf = bytearray(rec)
    a = (C.c_uint8 * len(buf)).from_buffer(buf)
    lib.orc_filter_reverse_tags(a, len(buf))
    return bytes(buf)


# ---- methylation-aware mode (oracle/oracle_methylation.hpp) -----------------------------------------------------------------
NO_REF_POS = -(1 << 63)
METH_DISABLED, METH_EM_SEQ, METH_TAPS = 0, 1, 2


def set_reference(seqs):
    """`set_reference(reference, ref_names)`: seqs[i] = the bases of header contig i (bytes); None / [] clears it."""
    seqs = list(seqs or [])
    if not seqs:
        lib.orc_set_reference(0, None, None)
        return
    bufs = [C.create_string_buffer(bytes(s), len(s)) for s in seqs]
    ptrs = (C.c_void_p * len(seqs))(*[C.cast(b, C.c_void_p).value for b in bufs])
    lens = (C.c_uint64 * len(seqs))(*[len(s) for s in seqs])
    lib.orc_set_reference(len(seqs), ptrs, lens)          # (the oracle copies the sequences)


def _ops(cigar):
    import bamutil
    return np.array(bamutil.cigar_ops(cigar), dtype=np.uint32)


def meth_query_to_ref_positions(simplified, alignment_start, is_reverse, original):
    """CIGAR strings of simplified ops (M / I / D / N / P); None for insertions."""
    s, o = _ops(simplified), _ops(original)
    out = np.zeros(int(sum(x >> 4 for x in s)) + 1, dtype=np.int64)
    n = lib.orc_meth_query_to_ref_positions(ptr(s), len(s), alignment_start, int(is_reverse), ptr(o), len(o), ptr(out), len(out))
    return [None if v == NO_REF_POS else int(v) for v in out[:n]]


def meth_annotate(length, reads, ref_bases, top):
    """reads: list of base strings; ref_bases: list of one-character strings or None.  Returns (is_ref_c, unconverted, converted)."""
    cat = np.frombuffer("".join(reads).encode(), dtype=np.uint8) if reads else np.zeros(0, np.uint8)
    cat = np.ascontiguousarray(cat) if len(cat) else np.zeros(1, np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    rb = np.array([0 if b is None else ord(b) for b in ref_bases], dtype=np.uint8)
    is_c, u, t = np.zeros(max(1, length), np.uint8), np.zeros(max(1, length), np.uint32), np.zeros(max(1, length), np.uint32)
    lib.orc_meth_annotate(length, ptr(cat), ptr(lens) if len(lens) else None, len(reads), ptr(rb) if len(rb) else None, len(rb), int(top), ptr(is_c), ptr(u), ptr(t))
    return [bool(x) for x in is_c[:length]], [int(x) for x in u[:length]], [int(x) for x in t[:length]]


def _ev(evidence):
    n = len(evidence)
    return (np.array([int(e[0]) for e in evidence] + [0], dtype=np.uint8), np.array([e[1] for e in evidence] + [0], dtype=np.uint32),
            np.array([e[2] for e in evidence] + [0], dtype=np.uint32), n)


def meth_build_mm_ml(bases, evidence, top, mode):
    """evidence: list of (is_ref_c, unconverted, converted).  Returns (MM string, ML list) or None; raises where the reference panics."""
    b = np.frombuffer(bases.encode() + b"\0", dtype=np.uint8).copy()
    c, u, t, n = _ev(evidence)
    mm = C.create_string_buffer(16 + 12 * (len(bases) + 1))
    ml = np.zeros(len(bases) + 1, dtype=np.uint8)
    r = lib.orc_meth_build_mm_ml(ptr(b), len(bases), n, ptr(c), ptr(u), ptr(t), int(top), mode, mm, len(mm), ptr(ml), len(ml))
    if r == -1:
        return None
    if r < 0:
        raise RuntimeError("build_mm_ml_tags: the reference panics (length mismatch)" if r == -2 else "buffer")
    return mm.value.decode(), [int(x) for x in ml[:r]]


def meth_combine(ab, ba, length):
    ac, au, at, na = _ev(ab)
    bc, bu, bt, nb = _ev(ba)
    oc, ou, ot = np.zeros(length + 1, np.uint8), np.zeros(length + 1, np.uint32), np.zeros(length + 1, np.uint32)
    lib.orc_meth_combine(na, ptr(ac), ptr(au), ptr(at), nb, ptr(bc), ptr(bu), ptr(bt), length, ptr(oc), ptr(ou), ptr(ot))
    return [(bool(oc[i]), int(ou[i]), int(ot[i])) for i in range(length)]
