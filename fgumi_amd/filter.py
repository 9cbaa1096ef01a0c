"""Python host mirror of `fgumi filter` over the C ABI (fgx_filter_*): unmapped consensus records, and — with a reference (`--ref`,
`ConsensusFilter.set_reference`) — mapped ones, whose NM / UQ / MD tags are regenerated after the masking.

Mirrors (names, argument meaning, error behaviour):
  * `FilterThresholds`, `FilterConfig::{new, for_single_strand, for_duplex, for_duplex_asymmetric}`
                                         crates/fgumi-consensus/src/filter.rs:29-329 (1-3 values expand from the last; the
                                         ordering asserts of :284-318 raise ValueError here)
  * `Filter::validate_parameters`        src/lib/commands/filter.rs:1021-1107
  * the Process closures                 src/lib/commands/filter.rs:581-625 (single read), :653-731 (template) — `filter_records`
  * `FilterProcessedBatchRaw`            src/lib/commands/filter.rs:226-238 — `FilterResult`

Masking, thresholds, template decisions and the record copies all run in the HIP library; nothing here filters on the CPU.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from ._lib import FilterOptions, FilterOutput, default_options, lib


@dataclass
class FilterThresholds:
    min_reads: int
    max_read_error_rate: float = 0.025
    max_base_error_rate: float = 0.1


def _three(v):
    v = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
    if not v:
        raise ValueError("at least one value required")
    if len(v) > 3:
        raise ValueError(f"must have 1-3 values, got {len(v)}")
    return (v + [v[-1]] * 3)[:3]


@dataclass
class FilterConfig:
    """[duplex (CC), AB, BA] thresholds + the read-level options (filter.rs:52-75)."""
    duplex: FilterThresholds
    ab: FilterThresholds
    ba: FilterThresholds
    min_base_quality: Optional[int] = None
    min_mean_base_quality: Optional[float] = None
    max_no_call_fraction: float = 0.2

    def __post_init__(self):
        cc, ab, ba = self.duplex, self.ab, self.ba
        for t in (cc, ab, ba):
            if not 0.0 <= t.max_read_error_rate <= 1.0:
                raise ValueError(f"--max-read-error-rate must be between 0.0 and 1.0, got {t.max_read_error_rate}")
            if not 0.0 <= t.max_base_error_rate <= 1.0:
                raise ValueError(f"--max-base-error-rate must be between 0.0 and 1.0, got {t.max_base_error_rate}")
        if ab.min_reads > cc.min_reads:
            raise ValueError(f"min-reads values must be specified high to low: AB ({ab.min_reads}) > duplex ({cc.min_reads})")
        if ba.min_reads > ab.min_reads:
            raise ValueError(f"min-reads values must be specified high to low: BA ({ba.min_reads}) > AB ({ab.min_reads})")
        if ab.max_read_error_rate > ba.max_read_error_rate:
            raise ValueError(f"max-read-error-rate for AB ({ab.max_read_error_rate}) must be <= BA ({ba.max_read_error_rate})")
        if ab.max_base_error_rate > ba.max_base_error_rate:
            raise ValueError(f"max-base-error-rate for AB ({ab.max_base_error_rate}) must be <= BA ({ba.max_base_error_rate})")
        if self.max_no_call_fraction < 0.0:
            raise ValueError(f"--max-no-call-fraction must be >= 0.0, got {self.max_no_call_fraction}")
        if self.max_no_call_fraction >= 1.0 and self.max_no_call_fraction != int(self.max_no_call_fraction):
            raise ValueError(f"--max-no-call-fraction >= 1.0 must be an integer (count of bases), got {self.max_no_call_fraction}")

    @classmethod
    def new(cls, min_reads: Sequence[int], max_read_error_rate: Sequence[float] = (0.025,), max_base_error_rate: Sequence[float] = (0.1,),
            min_base_quality: Optional[int] = None, min_mean_base_quality: Optional[float] = None, max_no_call_fraction: float = 0.2) -> "FilterConfig":
        r, e, b = _three(min_reads), _three(max_read_error_rate), _three(max_base_error_rate)
        t = [FilterThresholds(int(r[i]), float(e[i]), float(b[i])) for i in range(3)]
        return cls(t[0], t[1], t[2], min_base_quality, min_mean_base_quality, max_no_call_fraction)

    @classmethod
    def for_single_strand(cls, thresholds: FilterThresholds, min_base_quality=None, min_mean_base_quality=None, max_no_call_fraction=0.2):
        return cls(thresholds, thresholds, thresholds, min_base_quality, min_mean_base_quality, max_no_call_fraction)

    @classmethod
    def for_duplex(cls, duplex: FilterThresholds, strand: FilterThresholds, min_base_quality=None, min_mean_base_quality=None, max_no_call_fraction=0.2):
        return cls(duplex, strand, strand, min_base_quality, min_mean_base_quality, max_no_call_fraction)

    @classmethod
    def for_duplex_asymmetric(cls, duplex, ab, ba, min_base_quality=None, min_mean_base_quality=None, max_no_call_fraction=0.2):
        return cls(duplex, ab, ba, min_base_quality, min_mean_base_quality, max_no_call_fraction)


@dataclass
class FilterResult:
    """`FilterProcessedBatchRaw`: kept / rejected records as block_size-prefixed streams + the counters."""
    data: bytes
    rejects: bytes
    records_count: int
    passed_count: int
    bases_masked: int
    rejected_count: int


@dataclass
class DeviceFilterResult:
    data_ptr: int
    data_len: int
    rejects_ptr: int
    rejects_len: int
    records_count: int
    passed_count: int
    bases_masked: int
    rejected_count: int

    def to_host(self) -> FilterResult:
        import torch  # noqa: F401
        from ._lib import hip_memcpy_d2h
        d = hip_memcpy_d2h(self.data_ptr, self.data_len) if self.data_len else b""
        r = hip_memcpy_d2h(self.rejects_ptr, self.rejects_len) if self.rejects_len else b""
        return FilterResult(d, r, self.records_count, self.passed_count, self.bases_masked, self.rejected_count)


def record_offsets(data: bytes):
    """Record boundaries of a block_size-prefixed stream (the sequential length chain; host work, like the reader's FindBoundaries)."""
    off, ln = [], []
    p, n = 0, len(data)
    mv = memoryview(data)
    while p + 4 <= n:
        l = int.from_bytes(mv[p:p + 4], "little")
        off.append(p + 4)
        ln.append(l)
        p += 4 + l
    if p != n:
        raise ValueError("truncated record stream")
    return np.array(off, dtype=np.uint64), np.array(ln, dtype=np.uint32)


class ConsensusFilter:
    """`fgumi filter`: consensus records in, kept (and optionally rejected) records out, masked like the reference.  Without a reference a mapped
    record is the reference's fatal error ("--ref is required ..."); after `set_reference` (the command's `--ref`, filter.rs:115-118) mapped
    records are accepted and NM / UQ / MD are regenerated after the masking (unmapped records lose the three tags), as
    `regenerate_alignment_tags_raw` does (crates/fgumi-sam/src/alignment_tags.rs:259-433)."""

    def __init__(self, config: FilterConfig, filter_by_template: bool = True, require_single_strand_agreement: bool = False,
                 reverse_per_base_tags: bool = False, track_rejects: bool = False, device: int = -1, handle=None,
                 min_methylation_depth: Optional[Sequence[int]] = None, require_strand_methylation_agreement: bool = False,
                 min_conversion_fraction: Optional[float] = None, methylation_mode: Optional[str] = None):
        self.config = config
        o = FilterOptions()
        lib.fgx_filter_options_default(C.byref(o))
        for i, t in enumerate((config.duplex, config.ab, config.ba)):
            o.min_reads[i], o.max_read_error_rate[i], o.max_base_error_rate[i] = t.min_reads, t.max_read_error_rate, t.max_base_error_rate
        o.has_min_base_quality, o.min_base_quality = int(config.min_base_quality is not None), int(config.min_base_quality or 0)
        o.has_min_mean_base_quality, o.min_mean_base_quality = int(config.min_mean_base_quality is not None), float(config.min_mean_base_quality or 0.0)
        o.max_no_call_fraction = config.max_no_call_fraction
        o.require_single_strand_agreement, o.reverse_per_base_tags = int(require_single_strand_agreement), int(reverse_per_base_tags)
        o.filter_by_template, o.track_rejects = int(filter_by_template), int(track_rejects)
        # the methylation filters (--min-methylation-depth 1-3 values, --require-strand-methylation-agreement, --min-conversion-fraction with
        # --methylation-mode em-seq|taps: src/lib/commands/filter.rs:181-206; the last two need set_reference, checked by the library at the call)
        if min_methylation_depth is not None:
            d = list(min_methylation_depth) if isinstance(min_methylation_depth, (list, tuple)) else [int(min_methylation_depth)]
            if not 1 <= len(d) <= 3:
                raise ValueError(f"--min-methylation-depth must have 1-3 values, got {len(d)}")
            d = (d + [d[-1]] * 3)[:3]
            o.has_min_methylation_depth = 1
            o.min_methylation_depth[:] = d
        o.require_strand_methylation_agreement = int(require_strand_methylation_agreement)
        if min_conversion_fraction is not None:
            o.has_min_conversion_fraction, o.min_conversion_fraction = 1, float(min_conversion_fraction)
        if methylation_mode is not None:
            modes = {"disabled": 0, "em-seq": 1, "emseq": 1, "taps": 2}
            if str(methylation_mode).lower() not in modes:
                raise ValueError(f"methylation_mode must be 'em-seq' or 'taps', got {methylation_mode!r}")
            o.methylation_mode = modes[str(methylation_mode).lower()]
        self._o = o
        self._own = handle is None
        if handle is None:
            opts = default_options(device=device)
            handle = lib.fgx_create(C.byref(opts))
            if not handle:
                raise RuntimeError(lib.fgx_global_error().decode())
        self._h = handle

    def close(self):
        if self._own and self._h:
            lib.fgx_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())

    def set_reference(self, reference, ref_names: Sequence[str]):
        """`--ref <fasta>`: `reference` maps a contig name to its bases, `ref_names[i]` = the name of contig i of the BAM header (a record's
        reference id).  A header contig that `reference` lacks is NOT an error here: the reference fails lazily, with
        "Reference not found: <contig>", at the first mapped record that needs the contig (fgumi-sam alignment_tags.rs:272-300, 482) — a BAM
        whose header lists decoy / alt contigs with no reads on them filters fine there, and does here.  The missing contigs are
        remembered, get an empty stand-in on the device, and `filter_stream` / `filter_device` raise `KeyError("Reference not found: <contig>")`
        before anything is filtered when a mapped record of the batch lies on one.  `set_reference(None, [])` drops the reference again."""
        names = list(ref_names or [])
        self._missing = {}
        if reference is None or not names:
            self._check(lib.fgx_set_reference(self._h, 0, None, None))
            self._o.regenerate_alignment_tags = 0
            return
        self._missing = {i: n for i, n in enumerate(names) if n not in reference}
        seqs = [b"" if i in self._missing else bytes(reference[n]) for i, n in enumerate(names)]
        bufs = [C.create_string_buffer(s, max(1, len(s))) for s in seqs]
        ptrs = (C.c_void_p * len(seqs))(*[C.cast(b, C.c_void_p).value for b in bufs])
        lens = (C.c_uint64 * len(seqs))(*[len(s) for s in seqs])
        self._check(lib.fgx_set_reference(self._h, len(seqs), ptrs, lens))
        self._o.regenerate_alignment_tags = 1

    def _refuse_missing_contigs(self, ref_id, flag):
        """`ref_id` / `flag` of the batch's records (numpy arrays): the reference's lazy "Reference not found" for a mapped record on a contig the FASTA lacks."""
        missing = getattr(self, "_missing", None)
        if not missing or not self._o.regenerate_alignment_tags:
            return
        mapped = (flag & 0x4) == 0
        hit = mapped & np.isin(ref_id, np.fromiter(missing.keys(), dtype=np.int64))
        if hit.any():
            raise KeyError(f"Reference not found: {missing[int(ref_id[np.argmax(hit)])]}")

    def filter_stream(self, blob: np.ndarray, rec_off: np.ndarray, rec_len: np.ndarray) -> FilterResult:
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        rec_len = np.ascontiguousarray(rec_len, dtype=np.uint32)
        if getattr(self, "_missing", None) and len(rec_off):
            o = rec_off.astype(np.int64)
            rid = (blob[o].astype(np.int64) | (blob[o + 1].astype(np.int64) << 8) | (blob[o + 2].astype(np.int64) << 16) | (blob[o + 3].astype(np.int64) << 24))
            rid = np.where(rid >= 1 << 31, rid - (1 << 32), rid)
            self._refuse_missing_contigs(rid, blob[o + 14].astype(np.int64) | (blob[o + 15].astype(np.int64) << 8))
        out = FilterOutput()
        self._check(lib.fgx_filter_records(self._h, C.byref(self._o), blob.ctypes.data, blob.size, rec_off.ctypes.data, rec_len.ctypes.data, len(rec_off),
                                           C.byref(out)))
        return FilterResult(C.string_at(out.data, out.data_len) if out.data_len else b"", C.string_at(out.rejects, out.rejects_len) if out.rejects_len else b"",
                            int(out.records_count), int(out.passed_count), int(out.bases_masked), int(out.rejected_count))

    def filter_records(self, data: bytes) -> FilterResult:
        """`data` = `ConsensusOutput.data` (records with block_size prefixes), e.g. what a consensus caller returned."""
        off, ln = record_offsets(data)
        return self.filter_stream(np.frombuffer(data, dtype=np.uint8), off, ln)

    def filter_device(self, blob, blob_len: int, rec_off, rec_len, n_rec: int) -> DeviceFilterResult:
        """torch tensors resident in HBM; `blob` is masked in place, outputs stay in HBM."""
        import torch
        torch.cuda.synchronize(blob.device)
        if getattr(self, "_missing", None) and n_rec:
            o = rec_off[:n_rec].view(torch.int64)
            b = blob.view(torch.uint8)
            rid = (b[o].long() | (b[o + 1].long() << 8) | (b[o + 2].long() << 16) | (b[o + 3].long() << 24)).cpu().numpy()
            rid = np.where(rid >= 1 << 31, rid - (1 << 32), rid)
            self._refuse_missing_contigs(rid, (b[o + 14].long() | (b[o + 15].long() << 8)).cpu().numpy())
        out = FilterOutput()
        self._check(lib.fgx_filter_records_device(self._h, C.byref(self._o), blob.data_ptr(), blob_len, rec_off.data_ptr(), rec_len.data_ptr(), n_rec, C.byref(out)))
        return DeviceFilterResult(out.data or 0, int(out.data_len), out.rejects or 0, int(out.rejects_len), int(out.records_count), int(out.passed_count),
                                  int(out.bases_masked), int(out.rejected_count))

    @classmethod
    def on_caller(cls, caller, config: FilterConfig, **kw) -> "ConsensusFilter":
        """A filter bound to a consensus caller's handle, for `filter_last_output_device`."""
        return cls(config, handle=caller._h, **kw)

    def filter_last_output_device(self) -> DeviceFilterResult:
        """Filters, in HBM, the records the bound caller's last `process_batch_device` produced."""
        out = FilterOutput()
        self._check(lib.fgx_filter_last_output_device(self._h, C.byref(self._o), C.byref(out)))
        return DeviceFilterResult(out.data or 0, int(out.data_len), out.rejects or 0, int(out.rejects_len), int(out.records_count), int(out.passed_count),
                                  int(out.bases_masked), int(out.rejected_count))
