"""Python host mirror of the reference's consensus-caller interface, over the C ABI.

Mirrors (names, argument meaning, error behaviour):
  * `ConsensusCaller` trait            crates/fgumi-consensus/src/caller.rs:220-252
  * `ConsensusOutput`                  caller.rs:172-177
  * `ConsensusCallingStats`            caller.rs:256-321, `RejectionReason` caller.rs:401-446
  * `VanillaUmiConsensusOptions`       vanilla_caller.rs:292-353 (library defaults, NOT the CLI's)
  * `VanillaUmiConsensusCaller::new_with_rejects_tracking`   vanilla_caller.rs:432-463
  * the Process-step closure           src/lib/commands/simplex.rs:637-718  (`process_batch`)

All arithmetic runs in the HIP library; nothing here computes consensus on the CPU.
"""
import ctypes as C
import enum
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from ._lib import GroupOptions, Options, Output, SimParams, lib, load


class RejectionReason(enum.IntEnum):
    FragmentRead = 0
    InsufficientReads = 1
    QualityTooLow = 2
    Unmapped = 3
    Mapped = 4
    TooManyNs = 5
    MinorityAlignment = 6
    SecondaryOrSupplementary = 7
    FailedQC = 8
    MissingUmi = 9
    QualityTrimmed = 10
    ZeroLengthAfterTrimming = 11
    InsufficientOverlap = 12
    OrphanConsensus = 13
    IndelErrorBetweenStrands = 14
    ClipOverlapFailed = 15
    HighDuplexDisagreement = 16
    PotentialCollision = 17
    NotPrimaryFrPair = 18
    Downsampled = 19
    Other = 20


@dataclass
class ConsensusOutput:
    """Concatenated BAM records, each prefixed by its LE u32 block_size (caller.rs:172-177)."""
    data: bytes = b""
    count: int = 0

    def extend(self, other: "ConsensusOutput"):
        self.data += other.data
        self.count += other.count


@dataclass
class ConsensusCallingStats:
    total_reads: int = 0
    consensus_reads: int = 0
    filtered_reads: int = 0
    rejection_reasons: Dict[RejectionReason, int] = field(default_factory=dict)
    overlapping: Dict[str, int] = field(default_factory=dict)

    @classmethod
    def from_array(cls, a):
        s = cls(int(a[0]), int(a[1]), int(a[2]))
        for r in RejectionReason:
            if a[3 + r]:
                s.rejection_reasons[r] = int(a[3 + r])
        s.overlapping = dict(overlapping_bases=int(a[24]), bases_agreeing=int(a[25]), bases_disagreeing=int(a[26]),
                             bases_corrected=int(a[27]))
        return s

    def merge(self, o: "ConsensusCallingStats"):
        self.total_reads += o.total_reads
        self.consensus_reads += o.consensus_reads
        self.filtered_reads += o.filtered_reads
        for k, v in o.rejection_reasons.items():
            self.rejection_reasons[k] = self.rejection_reasons.get(k, 0) + v
        for k, v in o.overlapping.items():
            self.overlapping[k] = self.overlapping.get(k, 0) + v


@dataclass
class VanillaUmiConsensusOptions:
    """vanilla_caller.rs:292-353 (same defaults as `impl Default`)."""
    tag: str = "MI"
    error_rate_pre_umi: int = 45
    error_rate_post_umi: int = 40
    min_input_base_quality: int = 10
    min_reads: int = 2
    max_reads: Optional[int] = None
    produce_per_base_tags: bool = True
    trim: bool = False
    min_consensus_base_quality: int = 40
    cell_tag: Optional[str] = None
    tie_rule: int = 0  # 0 FgbioCompat (default), 1 UlpRelative
    methylation_mode: int = 0  # MethylationMode (lib.rs:45-68): 0 Disabled, 1 EmSeq, 2 Taps; needs set_reference()


@dataclass
class GroupedReads:
    """A batch of MI groups = `MiGroupBatch` (src/lib/mi_group.rs:22-57) as flat arrays."""
    blob: np.ndarray       # uint8: BAM record stream
    rec_off: np.ndarray    # uint64[n_rec]: offset of each record body
    rec_len: np.ndarray    # uint32[n_rec]
    grp_first: np.ndarray  # uint32[n_grp+1]

    @property
    def n_rec(self):
        return len(self.rec_off)

    @property
    def n_grp(self):
        return len(self.grp_first) - 1

    @classmethod
    def from_groups(cls, groups: Sequence[Sequence[bytes]]) -> "GroupedReads":
        chunks, off, lens, first = [], [], [], [0]
        pos = 0
        for g in groups:
            for r in g:
                chunks.append(len(r).to_bytes(4, "little"))
                chunks.append(bytes(r))
                off.append(pos + 4)
                lens.append(len(r))
                pos += 4 + len(r)
            first.append(len(off))
        blob = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8).copy()
        return cls(blob, np.array(off, dtype=np.uint64), np.array(lens, dtype=np.uint32), np.array(first, dtype=np.uint32))

    def records(self, g: int) -> List[bytes]:
        a, b = int(self.grp_first[g]), int(self.grp_first[g + 1])
        return [bytes(self.blob[int(self.rec_off[r]): int(self.rec_off[r]) + int(self.rec_len[r])]) for r in range(a, b)]

    def to_device(self, device=None) -> "DeviceGroupedReads":
        """Upload to HBM (torch tensors own the memory) for `process_batch_device`."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        pad = np.zeros(16, dtype=np.uint8)
        blob = torch.from_numpy(np.concatenate([self.blob, pad])).to(dev)
        return DeviceGroupedReads(blob, int(self.blob.size), torch.from_numpy(self.rec_off.astype(np.int64)).to(dev),
                                  torch.from_numpy(self.rec_len.astype(np.int32)).to(dev), torch.from_numpy(self.grp_first.astype(np.int32)).to(dev),
                                  self.n_rec, self.n_grp)

    def subset(self, g0: int, g1: int) -> "GroupedReads":
        return GroupedReads.from_groups([self.records(g) for g in range(g0, g1)])


@dataclass
class DeviceGroupedReads:
    """A batch of MI groups resident in HBM (torch tensors own the memory; plumbing only)."""
    blob: object
    blob_len: int
    rec_off: object
    rec_len: object
    grp_first: object
    n_rec: int
    n_grp: int


@dataclass
class DeviceOutput:
    """`ConsensusOutput` left in HBM: `data_ptr` is a device pointer owned by the caller object."""
    data_ptr: int
    data_len: int
    count: int
    n_deferred: int
    deferred_ptr: Optional[int]
    rejects_ptr: Optional[int] = None      # track_rejects with FGX_REJECTS_DEVICE=1: the rejected input records, block_size-prefixed, in HBM
    rejects_len: int = 0
    n_rejects: int = 0

    def rejects_to_host(self) -> bytes:
        import torch  # noqa: F401
        from ._lib import hip_memcpy_d2h
        return hip_memcpy_d2h(self.rejects_ptr, self.rejects_len) if self.rejects_len else b""

    def to_host(self) -> bytes:
        import torch  # noqa: F401  (ensures the HIP runtime is initialised in this process)
        from ._lib import hip_memcpy_d2h
        return hip_memcpy_d2h(self.data_ptr, self.data_len)

    def as_tensor(self, device=None):
        """Zero-copy uint8 torch view of the records in HBM (valid until the caller's next call): what a collective takes."""
        import torch

        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (int(self.data_len),), "typestr": "|u1", "data": (int(self.data_ptr), False), "version": 2}
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        if self.data_len == 0:
            return torch.empty(0, dtype=torch.uint8, device=dev)
        return torch.as_tensor(v, device=dev)


def split_records(data: bytes) -> List[bytes]:
    """Split `ConsensusOutput.data` into record bodies (without the block_size prefixes)."""
    out, p = [], 0
    while p < len(data):
        n = int.from_bytes(data[p:p + 4], "little")
        out.append(data[p + 4:p + 4 + n])
        p += 4 + n
    return out


def simulate_grouped_reads(n_families, family_size=3, read_length=150, seed=42, **kw) -> GroupedReads:
    """Host generation of `fgumi simulate grouped-reads`-shaped input (see csrc/simgen.h)."""
    p = SimParams()
    p.seed, p.n_families, p.read_length, p.family_size = seed, n_families, read_length, family_size
    p.insert_mean, p.insert_sd, p.error_rate_ppm = 300, 50, 1000
    for k, v in kw.items():
        setattr(p, k, v)
    bl, nr = C.c_uint64(), C.c_uint64()
    if lib.fgx_sim_sizes(C.byref(p), C.byref(bl), C.byref(nr)) != 0:
        raise ValueError("simulated input too large for 32-bit record indices")
    blob = np.zeros(max(1, bl.value), dtype=np.uint8)
    rec_off = np.zeros(nr.value, dtype=np.uint64)
    rec_len = np.zeros(nr.value, dtype=np.uint32)
    grp_first = np.zeros(n_families + 1, dtype=np.uint32)
    lib.fgx_sim_generate_host(C.byref(p), blob.ctypes.data, rec_off.ctypes.data, rec_len.ctypes.data, grp_first.ctypes.data)
    return GroupedReads(blob, rec_off, rec_len, grp_first)


def simulated_family_bytes(n_families, family_size=3, read_length=150, seed=42, **kw) -> np.ndarray:
    """Record bytes of each simulated family (uint64[n_families]) without generating the records: the weights a reader
    cuts the family stream by when it shards a FIXED total over several GPUs (`distributed.balanced_shards`)."""
    p = SimParams()
    p.seed, p.n_families, p.read_length, p.family_size = seed, n_families, read_length, family_size
    p.insert_mean, p.insert_sd, p.error_rate_ppm = 300, 50, 1000
    for k, v in kw.items():
        setattr(p, k, v)
    out = np.zeros(n_families, dtype=np.uint64)
    if lib.fgx_sim_family_bytes(C.byref(p), out.ctypes.data) != 0:
        raise ValueError("fgx_sim_family_bytes failed")
    return out


class ConsensusCaller:
    """The `ConsensusCaller` trait (caller.rs:220-252)."""

    def consensus_reads(self, records: Sequence[bytes]) -> ConsensusOutput:
        raise NotImplementedError

    def total_reads(self) -> int:
        return self._stats.total_reads

    def total_filtered(self) -> int:
        return self._stats.filtered_reads

    def consensus_reads_constructed(self) -> int:
        return self._stats.consensus_reads

    def statistics(self) -> ConsensusCallingStats:
        return self._stats

    def log_statistics(self):
        import logging
        log = logging.getLogger("fgumi_amd")
        log.info("Consensus Calling Statistics:")
        log.info("  Total input reads: %d", self._stats.total_reads)
        log.info("  Consensus reads generated: %d", self._stats.consensus_reads)
        log.info("  Reads filtered: %d", self._stats.filtered_reads)
        for reason, count in self._stats.rejection_reasons.items():
            log.info("    %s: %d", reason.name, count)


class _HandleCaller(ConsensusCaller):
    kind = 0

    def __init__(self, opts: Options, read_name_prefix: str, read_group_id: str):
        load()
        self._prefix = read_name_prefix.encode()
        self._rg = read_group_id.encode()
        opts.read_name_prefix = self._prefix
        opts.read_group_id = self._rg
        self._opts = opts
        self._h = lib.fgx_create(C.byref(opts))
        if not self._h:
            raise RuntimeError(lib.fgx_global_error().decode())
        self._stats = ConsensusCallingStats()
        self._rejected: List[bytes] = []
        self.last_timing = {}

    def close(self):
        if getattr(self, "_h", None):
            lib.fgx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batch level: the process_fn closure -------------------------------------------------
    def process_batch(self, grouped: GroupedReads) -> ConsensusOutput:
        out = Output()
        rc = lib.fgx_process_batch(self._h, grouped.blob.ctypes.data, grouped.blob.size, grouped.rec_off.ctypes.data,
                                   grouped.rec_len.ctypes.data, grouped.n_rec, grouped.grp_first.ctypes.data, grouped.n_grp, C.byref(out))
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())  # every error is fatal to the run (simplex.rs:699-701)
        self._last_stats = ConsensusCallingStats.from_array(out.stats)
        self._stats.merge(self._last_stats)
        self.last_stats_array = [int(v) for v in out.stats]          # the raw counters of this batch (fgx_output.stats order)
        if out.n_rejects:
            self._rejected.extend(split_records(C.string_at(out.rejects, out.rejects_len)))
        self.last_timing = dict(host_prep=out.ms_host_prep, h2d=out.ms_h2d, kernels=out.ms_kernels, d2h=out.ms_d2h, emit=out.ms_emit)
        data = C.string_at(out.data, out.data_len) if out.data_len else b""
        return ConsensusOutput(data, int(out.count))

    def last_batch_statistics(self) -> ConsensusCallingStats:
        return self._last_stats

    # ---- methylation-aware mode ---------------------------------------------------------------------
    def set_reference(self, reference, ref_names: Sequence[str]):
        """`set_reference(reference, ref_names)` (vanilla_caller.rs:512-522): `reference` maps a contig name to its bases (the
        in-memory `RefBaseProvider`), `ref_names[i]` = the name of header contig i (a record's ref_id).  The sequences go to HBM
        once.  A name the mapping lacks is an empty contig: every base of it is unknown, as with the reference's per-base
        lookups.  `set_reference(None, [])` drops the reference."""
        names = list(ref_names or [])
        if reference is None or not names:
            rc = lib.fgx_set_reference(self._h, 0, None, None)
        else:
            seqs = [bytes(reference.get(n, b"")) for n in names]
            bufs = [C.create_string_buffer(s, max(1, len(s))) for s in seqs]
            ptrs = (C.c_void_p * len(seqs))(*[C.cast(b, C.c_void_p).value for b in bufs])
            lens = (C.c_uint64 * len(seqs))(*[len(s) for s in seqs])
            rc = lib.fgx_set_reference(self._h, len(seqs), ptrs, lens)
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())

    def set_general_only(self, on: bool = True):
        """Route every family through the general host-orchestrated path (default: device-resident fast
        path, general path only for the families it defers)."""
        lib.fgx_set_general_only(self._h, int(on))

    # ---- MI grouping (mi_group.rs MiGrouper + the commands' pre-group record filter) ---------------
    def group_records(self, blob: np.ndarray, rec_off: np.ndarray, rec_len: np.ndarray, tag: str = "MI", cell_tag: Optional[str] = "CB",
                      strip_strand_suffix: bool = False, allow_unmapped: bool = False) -> GroupedReads:
        """Which records a consensus command keeps and where the MI groups start, computed on the device from host
        buffers: drop-in for `MiGrouper::add_records`; the result feeds `process_batch` directly."""
        o = GroupOptions(tag.encode(), cell_tag.encode() if cell_tag else b"\0\0", int(strip_strand_suffix), int(allow_unmapped))
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        rec_len = np.ascontiguousarray(rec_len, dtype=np.uint32)
        n = len(rec_off)
        out_off = np.zeros(max(1, n), dtype=np.uint64)
        out_len = np.zeros(max(1, n), dtype=np.uint32)
        grp = np.zeros(n + 1, dtype=np.uint32)
        nk, ng = C.c_uint32(), C.c_uint32()
        rc = lib.fgx_group_records(self._h, C.byref(o), blob.ctypes.data, blob.size, rec_off.ctypes.data, rec_len.ctypes.data, n,
                                   out_off.ctypes.data, out_len.ctypes.data, grp.ctypes.data, C.byref(nk), C.byref(ng))
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())
        return GroupedReads(blob, out_off[:nk.value].copy(), out_len[:nk.value].copy(), grp[:ng.value + 1].copy())

    def group_records_device(self, dg: "DeviceGroupedReads", tag: str = "MI", cell_tag: Optional[str] = "CB", strip_strand_suffix: bool = False,
                             allow_unmapped: bool = False) -> "DeviceGroupedReads":
        """Same on a record stream resident in HBM (`dg.grp_first` is ignored); returns a new DeviceGroupedReads over the same blob."""
        import torch
        o = GroupOptions(tag.encode(), cell_tag.encode() if cell_tag else b"\0\0", int(strip_strand_suffix), int(allow_unmapped))
        dev = dg.blob.device
        n = dg.n_rec
        out_off = torch.empty(max(1, n), dtype=torch.int64, device=dev)
        out_len = torch.empty(max(1, n), dtype=torch.int32, device=dev)
        grp = torch.empty(n + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        nk, ng = C.c_uint32(), C.c_uint32()
        rc = lib.fgx_group_records_device(self._h, C.byref(o), dg.blob.data_ptr(), dg.blob_len, dg.rec_off.data_ptr(), dg.rec_len.data_ptr(), n,
                                          out_off.data_ptr(), out_len.data_ptr(), grp.data_ptr(), C.byref(nk), C.byref(ng))
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())
        return DeviceGroupedReads(dg.blob, dg.blob_len, out_off, out_len, grp, nk.value, ng.value)

    # ---- device-resident batch (inputs and outputs stay in HBM) ---------------------------------
    def process_batch_device(self, dg: "DeviceGroupedReads"):
        import torch
        # the library runs on its own non-blocking stream: whatever still produces `dg` on torch's stream must have finished
        torch.cuda.synchronize(dg.blob.device)
        out = Output()
        n_def = C.c_uint32()
        d_def = C.c_void_p()
        rc = lib.fgx_process_batch_device(self._h, dg.blob.data_ptr(), dg.blob_len, dg.rec_off.data_ptr(), dg.rec_len.data_ptr(), dg.n_rec,
                                          dg.grp_first.data_ptr(), dg.n_grp, C.byref(out), C.byref(n_def), C.byref(d_def))
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())
        self._last_stats = ConsensusCallingStats.from_array(out.stats)
        self._stats.merge(self._last_stats)
        self.last_stats_array = [int(v) for v in out.stats]          # the raw counters of this batch (fgx_output.stats order)
        self.last_timing = dict(kernels=out.ms_kernels, k_family=out.ms_k_family, k_emit=out.ms_k_emit, full_columns=int(out.ms_emit))
        return DeviceOutput(out.data, int(out.data_len), int(out.count), int(n_def.value), d_def.value, out.rejects, int(out.rejects_len), int(out.n_rejects))

    def simulate_on_device(self, n_families, family_size=3, read_length=150, seed=42, **kw) -> "DeviceGroupedReads":
        import torch
        p = SimParams()
        p.seed, p.n_families, p.read_length, p.family_size = seed, n_families, read_length, family_size
        p.insert_mean, p.insert_sd, p.error_rate_ppm = 300, 50, 1000
        for k, v in kw.items():
            setattr(p, k, v)
        bl, nr = C.c_uint64(), C.c_uint64()
        if lib.fgx_sim_sizes(C.byref(p), C.byref(bl), C.byref(nr)) != 0:
            raise ValueError("simulated input too large for 32-bit record indices")
        dev = torch.device("cuda", self._opts.device if self._opts.device >= 0 else torch.cuda.current_device())
        blob = torch.empty(max(16, bl.value) + 16, dtype=torch.uint8, device=dev)   # (the kernels read whole 16-byte pieces: see fgx_process_batch_device)
        rec_off = torch.empty(max(1, nr.value), dtype=torch.int64, device=dev)
        rec_len = torch.empty(max(1, nr.value), dtype=torch.int32, device=dev)
        grp_first = torch.empty(n_families + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        rc = lib.fgx_sim_generate_device(self._h, C.byref(p), blob.data_ptr(), rec_off.data_ptr(), rec_len.data_ptr(), grp_first.data_ptr())
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())
        return DeviceGroupedReads(blob, bl.value, rec_off, rec_len, grp_first, nr.value, n_families)

    # ---- a BAM file in, a consensus BAM file out (the streaming pipeline of csrc/pipeline.cpp) ---------
    def run_bam(self, in_path: str, out_path: str, header_text: Optional[str] = None, level: int = 1, threads: Optional[int] = None,
                chunk_raw_bytes: int = 0, tag: str = "MI", cell_tag: Optional[str] = "CB", strip_strand_suffix: bool = False,
                allow_unmapped: bool = False, host_inflate: bool = False, device_deflate: bool = False, rejects_path: Optional[str] = None) -> dict:
        """Reads the grouped BAM `in_path` chunk by chunk (BGZF inflate on the host cores, record boundaries + MI grouping + the
        consensus batch on the device — and the BGZF inflate itself, unless `host_inflate` — BGZF deflate on the host cores) and writes the consensus BAM `out_path`; the stages of
        successive chunks overlap.  Returns the pipeline's counters and stage times."""
        from . import bgzf
        from ._lib import BamRunStats
        if header_text is None:
            header_text = bgzf.consensus_header(self._rg.decode())
        hdr = bgzf.bam_header_bytes(header_text, [])
        hb = np.frombuffer(hdr, dtype=np.uint8)
        o = GroupOptions(tag.encode(), cell_tag.encode() if cell_tag else b"\0\0", int(strip_strand_suffix), int(allow_unmapped))
        st = BamRunStats()
        n_rej = C.c_uint64(0)
        # `rejects_path` = the reference's `--rejects <file>` (simplex.rs:260-285): the input header + the rejected input records, batch-input order
        rc = lib.fgx_run_bam_rejects(self._h, in_path.encode(), out_path.encode(), rejects_path.encode() if rejects_path else None, hb.ctypes.data, hb.size,
                                     C.byref(o), threads or 0, level, chunk_raw_bytes, (1 if host_inflate else 0) | (2 if device_deflate else 0), C.byref(st),
                                     C.byref(n_rej))
        if rc != 0:
            raise RuntimeError(lib.fgx_last_error(self._h).decode())
        self._last_stats = ConsensusCallingStats.from_array(st.stats)
        self._stats.merge(self._last_stats)
        out = {k: getattr(st, k) for k, _ in BamRunStats._fields_ if k not in ("stats", "_pad")}
        out["stats"] = [int(v) for v in st.stats]
        out["rejected_records"] = int(n_rej.value)
        return out

    # ---- the trait -----------------------------------------------------------------------------
    def consensus_reads(self, records: Sequence[bytes]) -> ConsensusOutput:
        if not records:
            return ConsensusOutput()
        return self.process_batch(GroupedReads.from_groups([list(records)]))

    def rejected_reads(self) -> List[bytes]:
        return self._rejected

    def take_rejected_reads(self) -> List[bytes]:
        r, self._rejected = self._rejected, []
        return r

    def clear(self):
        self._stats = ConsensusCallingStats()
        self._rejected = []


def _fill_vanilla(o: Options, v: VanillaUmiConsensusOptions):
    if len(v.tag) != 2:
        raise ValueError(f"Tag '{v.tag}' must be exactly 2 characters")  # vanilla_caller.rs:1892-1894
    o.tag = v.tag.encode()
    o.cell_tag = (v.cell_tag.encode() if v.cell_tag else b"\0\0")
    o.error_rate_pre_umi, o.error_rate_post_umi = v.error_rate_pre_umi, v.error_rate_post_umi
    o.min_input_base_quality, o.min_consensus_base_quality = v.min_input_base_quality, v.min_consensus_base_quality
    o.min_reads = v.min_reads
    o.max_reads = -1 if v.max_reads is None else v.max_reads
    o.produce_per_base_tags, o.trim, o.tie_rule = int(v.produce_per_base_tags), int(v.trim), v.tie_rule
    o.methylation_mode = int(v.methylation_mode)


class MethylationMode(enum.IntEnum):
    """crates/fgumi-consensus/src/lib.rs:45-68."""
    Disabled = 0
    EmSeq = 1
    Taps = 2


class VanillaUmiConsensusCaller(_HandleCaller):
    """`VanillaUmiConsensusCaller::new_with_rejects_tracking(read_name_prefix, read_group_id, options, track_rejects)`.

    `overlapping_consensus` switches on the R1/R2 pre-correction the commands apply before the caller
    (simplex.rs:688-694); it is off for the bare trait object, as in the reference."""

    def __init__(self, read_name_prefix: str, read_group_id: str, options: Optional[VanillaUmiConsensusOptions] = None,
                 track_rejects: bool = False, overlapping_consensus: bool = False, device: int = -1):
        from ._lib import default_options
        options = options or VanillaUmiConsensusOptions()
        o = default_options()
        o.caller_kind = 0
        _fill_vanilla(o, options)
        o.track_rejects = int(track_rejects)
        o.overlapping_consensus = int(overlapping_consensus)
        o.device = device
        self.options = options
        super().__init__(o, read_name_prefix, read_group_id)


class DuplexConsensusCaller(_HandleCaller):
    """`DuplexConsensusCaller::new(read_name_prefix, read_group_id, min_reads, min_input_base_quality,
    produce_per_base_tags, trim, max_reads_per_strand, cell_tag, track_rejects, error_rate_pre_umi,
    error_rate_post_umi)` (crates/fgumi-consensus/src/duplex_caller.rs:403-513).  `min_reads` = 1-3 values
    [total, XY, YX], missing slots repeat the last one (fgbio `padTo(3, last)`), must be high to low."""

    def __init__(self, read_name_prefix: str, read_group_id: str, min_reads: Sequence[int], min_input_base_quality: int = 10,
                 produce_per_base_tags: bool = True, trim: bool = False, max_reads_per_strand: Optional[int] = None,
                 cell_tag: Optional[str] = None, track_rejects: bool = False, error_rate_pre_umi: int = 45, error_rate_post_umi: int = 40,
                 tie_rule: int = 0, overlapping_consensus: bool = False, device: int = -1, methylation_mode: int = 0):
        from ._lib import default_options
        mr = list(min_reads)
        if not mr:
            raise ValueError("min_reads parameter must have at least 1 value")           # duplex_caller.rs:369-371
        if len(mr) > 3:
            raise ValueError(f"min_reads parameter must have 1-3 values (total, [XY, [YX]]), got {len(mr)} values")
        total, xy, yx = mr[0], (mr[1] if len(mr) > 1 else mr[-1]), (mr[2] if len(mr) > 2 else mr[-1])
        if xy > total:
            raise ValueError("min-reads values must be specified high to low (total >= XY)")
        if yx > xy:
            raise ValueError("min-reads values must be specified high to low (XY >= YX)")
        o = default_options()
        o.caller_kind = 1
        o.duplex_min_reads[0], o.duplex_min_reads[1], o.duplex_min_reads[2] = total, xy, yx
        o.min_input_base_quality = min_input_base_quality
        o.produce_per_base_tags, o.trim, o.tie_rule = int(produce_per_base_tags), int(trim), tie_rule
        o.duplex_max_reads_per_strand = -1 if max_reads_per_strand is None else max_reads_per_strand
        o.cell_tag = cell_tag.encode() if cell_tag else b"\0\0"
        o.error_rate_pre_umi, o.error_rate_post_umi = error_rate_pre_umi, error_rate_post_umi
        o.track_rejects, o.overlapping_consensus, o.device = int(track_rejects), int(overlapping_consensus), device
        o.methylation_mode = int(methylation_mode)
        super().__init__(o, read_name_prefix, read_group_id)

    def set_reference(self, reference, ref_names: Sequence[str], methylation_mode: Optional[int] = None):
        """`DuplexConsensusCaller::set_reference(reference, ref_names, methylation_mode)` (duplex_caller.rs:524-536): the mode
        goes to the single-strand caller's options (a new engine handle when it differs from the constructor's)."""
        if methylation_mode is not None and int(methylation_mode) != self._opts.methylation_mode:
            self.close()
            self._opts.methylation_mode = int(methylation_mode)
            self._h = lib.fgx_create(C.byref(self._opts))
            if not self._h:
                raise RuntimeError(lib.fgx_global_error().decode())
        super().set_reference(reference, ref_names)


@dataclass
class CodecConsensusOptions:
    """codec_caller.rs:176-262 (same defaults as `impl Default`)."""
    min_input_base_quality: int = 10
    error_rate_pre_umi: int = 45
    error_rate_post_umi: int = 40
    min_reads_per_strand: int = 1
    max_reads_per_strand: Optional[int] = None
    min_duplex_length: int = 1
    single_strand_qual: Optional[int] = None
    outer_bases_qual: Optional[int] = None
    outer_bases_length: int = 5
    max_duplex_disagreements: Optional[int] = None       # None = usize::MAX
    max_duplex_disagreement_rate: float = 1.0
    cell_tag: Optional[str] = None
    produce_per_base_tags: bool = False
    tie_rule: int = 0


@dataclass
class CodecConsensusStats:
    """codec_caller.rs:264-310."""
    total_input_reads: int = 0
    consensus_reads_generated: int = 0
    reads_filtered: int = 0
    consensus_bases_emitted: int = 0
    consensus_duplex_bases_emitted: int = 0
    duplex_disagreement_base_count: int = 0
    consensus_reads_rejected_hdd: int = 0
    rejection_reasons: Dict[RejectionReason, int] = field(default_factory=dict)

    def duplex_disagreement_rate(self) -> float:
        d = self.consensus_duplex_bases_emitted
        return self.duplex_disagreement_base_count / d if d else 0.0


class CodecConsensusCaller(_HandleCaller):
    """`CodecConsensusCaller::new_with_rejects_tracking(read_name_prefix, read_group_id, options, track_rejects)`
    (crates/fgumi-consensus/src/codec_caller.rs:361-445).  A molecule over the duplex-disagreement thresholds is a
    counted, recoverable reject exactly as in `fgumi codec`'s process_fn (commands/codec.rs:745-775): it emits
    nothing, is counted under HighDuplexDisagreement and the batch carries on."""

    def __init__(self, read_name_prefix: str, read_group_id: str, options: Optional[CodecConsensusOptions] = None,
                 track_rejects: bool = False, device: int = -1):
        from ._lib import default_options
        v = options or CodecConsensusOptions()
        o = default_options()
        o.caller_kind = 2
        o.min_input_base_quality = v.min_input_base_quality
        o.error_rate_pre_umi, o.error_rate_post_umi = v.error_rate_pre_umi, v.error_rate_post_umi
        o.codec_min_reads_per_strand = v.min_reads_per_strand
        o.codec_max_reads_per_strand = -1 if v.max_reads_per_strand is None else v.max_reads_per_strand
        o.codec_min_duplex_length = v.min_duplex_length
        o.codec_has_single_strand_qual, o.codec_single_strand_qual = int(v.single_strand_qual is not None), v.single_strand_qual or 0
        o.codec_has_outer_bases_qual, o.codec_outer_bases_qual = int(v.outer_bases_qual is not None), v.outer_bases_qual or 0
        o.codec_outer_bases_length = v.outer_bases_length
        o.codec_max_duplex_disagreements = 0xFFFFFFFF if v.max_duplex_disagreements is None else min(v.max_duplex_disagreements, 0xFFFFFFFE)
        o.codec_max_duplex_disagreement_rate = v.max_duplex_disagreement_rate
        o.cell_tag = v.cell_tag.encode() if v.cell_tag else b"\0\0"
        o.produce_per_base_tags, o.tie_rule = int(v.produce_per_base_tags), v.tie_rule
        o.track_rejects, o.overlapping_consensus, o.device = int(track_rejects), 0, device
        self.options = v
        super().__init__(o, read_name_prefix, read_group_id)

    def codec_statistics(self) -> CodecConsensusStats:
        """Counters of the last batch in the reference's `CodecConsensusStats` shape."""
        s = self.last_batch_statistics()
        ov = s.overlapping
        return CodecConsensusStats(s.total_reads, s.consensus_reads, s.filtered_reads, ov.get("overlapping_bases", 0), ov.get("bases_agreeing", 0),
                                   ov.get("bases_disagreeing", 0), ov.get("bases_corrected", 0), dict(s.rejection_reasons))
