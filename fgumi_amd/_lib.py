"""ctypes binding of libfgumi_amd.so (include/fgumi_amd.h).  There is no fallback: if the HIP
library is missing the import of any product entry point fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("FGX_LIB") or os.path.join(HERE, "libfgumi_amd.so")


class LibraryMissing(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("caller_kind", C.c_uint32), ("tag", C.c_char * 2), ("cell_tag", C.c_char * 2),
        ("error_rate_pre_umi", C.c_uint8), ("error_rate_post_umi", C.c_uint8), ("min_input_base_quality", C.c_uint8),
        ("min_consensus_base_quality", C.c_uint8), ("produce_per_base_tags", C.c_uint8), ("trim", C.c_uint8),
        ("tie_rule", C.c_uint8), ("overlapping_consensus", C.c_uint8), ("track_rejects", C.c_uint8), ("methylation_mode", C.c_uint8), ("_pad0", C.c_uint8 * 2),
        ("min_reads", C.c_uint32), ("max_reads", C.c_int64), ("read_name_prefix", C.c_char_p), ("read_group_id", C.c_char_p),
        ("duplex_min_reads", C.c_uint32 * 3), ("duplex_max_reads_per_strand", C.c_int64),
        ("codec_min_reads_per_strand", C.c_uint32), ("codec_max_reads_per_strand", C.c_int64), ("codec_min_duplex_length", C.c_uint32),
        ("codec_single_strand_qual", C.c_uint8), ("codec_outer_bases_qual", C.c_uint8), ("codec_has_single_strand_qual", C.c_uint8),
        ("codec_has_outer_bases_qual", C.c_uint8), ("codec_outer_bases_length", C.c_uint32),
        ("codec_max_duplex_disagreements", C.c_uint32), ("codec_max_duplex_disagreement_rate", C.c_double),
        ("device", C.c_int32), ("_pad1", C.c_uint32),
    ]


class Output(C.Structure):
    _fields_ = [
        ("data", C.c_void_p), ("data_len", C.c_uint64), ("count", C.c_uint64), ("stats", C.c_uint64 * 28),
        ("rejects", C.c_void_p), ("rejects_len", C.c_uint64), ("n_rejects", C.c_uint64),
        ("ms_host_prep", C.c_double), ("ms_h2d", C.c_double), ("ms_kernels", C.c_double), ("ms_d2h", C.c_double), ("ms_emit", C.c_double),
        ("ms_k_family", C.c_double), ("ms_k_emit", C.c_double),
    ]


class GroupOptions(C.Structure):
    """`fgx_group_options` (include/fgumi_amd.h)."""
    _fields_ = [("tag", C.c_char * 2), ("cell_tag", C.c_char * 2), ("strip_strand_suffix", C.c_uint8), ("allow_unmapped", C.c_uint8),
                ("_pad", C.c_uint8 * 2)]


class FilterOptions(C.Structure):
    """`fgx_filter_options` (include/fgumi_amd.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("min_reads", C.c_uint32 * 3), ("max_read_error_rate", C.c_double * 3), ("max_base_error_rate", C.c_double * 3),
                ("min_mean_base_quality", C.c_double), ("max_no_call_fraction", C.c_double), ("has_min_base_quality", C.c_uint8), ("min_base_quality", C.c_uint8),
                ("has_min_mean_base_quality", C.c_uint8), ("require_single_strand_agreement", C.c_uint8), ("reverse_per_base_tags", C.c_uint8),
                ("filter_by_template", C.c_uint8), ("track_rejects", C.c_uint8), ("regenerate_alignment_tags", C.c_uint8),
                ("has_min_methylation_depth", C.c_uint8), ("require_strand_methylation_agreement", C.c_uint8), ("has_min_conversion_fraction", C.c_uint8),
                ("methylation_mode", C.c_uint8), ("min_methylation_depth", C.c_uint32 * 3), ("min_conversion_fraction", C.c_double)]


class FilterOutput(C.Structure):
    """`fgx_filter_output` (include/fgumi_amd.h)."""
    _fields_ = [("data", C.c_void_p), ("data_len", C.c_uint64), ("rejects", C.c_void_p), ("rejects_len", C.c_uint64), ("records_count", C.c_uint64),
                ("passed_count", C.c_uint64), ("bases_masked", C.c_uint64), ("rejected_count", C.c_uint64)]


class SimParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_families", C.c_uint32), ("read_length", C.c_uint32), ("family_size", C.c_uint32),
        ("family_size_max", C.c_uint32), ("duplex", C.c_uint32), ("insert_mean", C.c_uint32), ("insert_sd", C.c_uint32),
        ("error_rate_ppm", C.c_uint32), ("first_family", C.c_uint32), ("codec", C.c_uint32),
    ]


# every symbol include/fgumi_amd.h declares
class BamRunStats(C.Structure):
    """include/fgumi_amd.h fgx_bam_run_stats."""
    _fields_ = [("kept_records", C.c_uint64), ("groups", C.c_uint64), ("consensus_records", C.c_uint64), ("deferred_groups", C.c_uint64),
                ("chunks", C.c_uint64), ("in_bytes", C.c_uint64), ("inflated_bytes", C.c_uint64), ("out_bytes", C.c_uint64),
                ("out_file_bytes", C.c_uint64), ("stats", C.c_uint64 * 28), ("seconds_total", C.c_double),
                ("seconds_read", C.c_double), ("seconds_inflate", C.c_double), ("seconds_device", C.c_double), ("seconds_deflate", C.c_double),
                ("seconds_write", C.c_double), ("seconds_h2d", C.c_double), ("seconds_boundaries", C.c_double), ("seconds_grouping", C.c_double),
                ("seconds_consensus", C.c_double), ("seconds_d2h", C.c_double), ("seconds_device_inflate", C.c_double),
                ("boundary_repair_rounds", C.c_uint32), ("device_inflate", C.c_uint32), ("seconds_device_deflate", C.c_double),
                ("device_deflate", C.c_uint32), ("host_entry_batches", C.c_uint32)]


EXPORTS = ["fgx_options_default", "fgx_create", "fgx_destroy", "fgx_last_error", "fgx_global_error", "fgx_process_batch",
           "fgx_process_batch_device", "fgx_call_columns", "fgx_device_libm", "fgx_get_table", "fgx_sim_sizes", "fgx_sim_family_bytes", "fgx_libm_self_check", "fgx_record_boundaries",
           "fgx_bgzf_inflate", "fgx_bgzf_deflate", "fgx_bgzf_free", "fgx_bgzf_last_error",
           "fgx_sim_generate_host", "fgx_sim_generate_device", "fgx_group_records", "fgx_group_records_device", "fgx_filter_options_default",
           "fgx_filter_records", "fgx_filter_records_device", "fgx_filter_last_output_device",
           "fgx_record_boundaries_device", "fgx_inflate_block_host", "fgx_inflate_block_two_phase_host", "fgx_deflate_block_host", "fgx_run_bam", "fgx_run_bam_rejects", "fgx_bgzf_inflate_device_bench", "fgx_bgzf_recompress_file", "fgx_pipeline_last_error",
           "fgx_set_reference", "fgx_methylation_annotate_host", "fgx_methylation_runs_host", "fgx_methylation_mm_ml_host", "fgx_canon_duplex_host", "fgx_canon_codec_host", "fgx_simplex_rejects_host", "fgx_strand_rejects_host", "fgx_balanced_shards", "fgx_regenerate_alignment_tags_host"]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise LibraryMissing(f"{SO} is missing: build it with `python -m fgumi_amd.build` (hipcc, gfx950). "
                             "There is no CPU fallback for the consensus path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; when torch is present,
    # let it load (and initialise) first so this library binds to the same runtime instance.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    L = C.CDLL(SO)
    VP, U8, U32, U64, I = C.c_void_p, C.c_uint8, C.c_uint32, C.c_uint64, C.c_int
    P = C.POINTER
    L.fgx_options_default.argtypes = [P(Options)]
    L.fgx_options_default.restype = None
    L.fgx_create.argtypes = [P(Options)]
    L.fgx_create.restype = VP
    L.fgx_destroy.argtypes = [VP]
    L.fgx_destroy.restype = None
    L.fgx_last_error.argtypes = [VP]
    L.fgx_last_error.restype = C.c_char_p
    L.fgx_global_error.argtypes = []
    L.fgx_global_error.restype = C.c_char_p
    L.fgx_process_batch.argtypes = [VP, VP, U64, VP, VP, U32, VP, U32, P(Output)]
    L.fgx_process_batch.restype = I
    L.fgx_process_batch_device.argtypes = [VP, VP, U64, VP, VP, U32, VP, U32, P(Output), P(U32), P(VP)]
    L.fgx_process_batch_device.restype = I
    L.fgx_call_columns.argtypes = [VP, VP, VP, U32, U32, VP, VP, VP, VP]
    L.fgx_call_columns.restype = I
    L.fgx_device_libm.argtypes = [VP, I, VP, VP, U64]
    L.fgx_device_libm.restype = I
    L.fgx_get_table.argtypes = [VP, I, VP, P(U32)]
    L.fgx_get_table.restype = I
    L.fgx_sim_sizes.argtypes = [P(SimParams), P(U64), P(U64)]
    L.fgx_sim_sizes.restype = I
    L.fgx_bgzf_inflate.argtypes = [VP, U64, U32, P(VP), P(U64)]
    L.fgx_bgzf_inflate.restype = I
    L.fgx_bgzf_deflate.argtypes = [VP, U64, I, U32, I, P(VP), P(U64)]
    L.fgx_bgzf_deflate.restype = I
    L.fgx_bgzf_free.argtypes = [VP]
    L.fgx_bgzf_free.restype = None
    L.fgx_bgzf_last_error.argtypes = []
    L.fgx_bgzf_last_error.restype = C.c_char_p
    L.fgx_record_boundaries.argtypes = [VP, U64, U64, VP, VP, U64, P(U64)]
    L.fgx_record_boundaries.restype = I
    L.fgx_libm_self_check.argtypes = [C.c_char_p, U64]
    L.fgx_libm_self_check.restype = I
    L.fgx_sim_family_bytes.argtypes = [P(SimParams), VP]
    L.fgx_sim_family_bytes.restype = I
    L.fgx_sim_generate_host.argtypes = [P(SimParams), VP, VP, VP, VP]
    L.fgx_sim_generate_host.restype = I
    L.fgx_sim_generate_device.argtypes = [VP, P(SimParams), VP, VP, VP, VP]
    L.fgx_sim_generate_device.restype = I
    L.fgx_group_records.argtypes = [VP, P(GroupOptions), VP, U64, VP, VP, U32, VP, VP, VP, P(U32), P(U32)]
    L.fgx_group_records.restype = I
    L.fgx_group_records_device.argtypes = [VP, P(GroupOptions), VP, U64, VP, VP, U32, VP, VP, VP, P(U32), P(U32)]
    L.fgx_group_records_device.restype = I
    L.fgx_record_boundaries_device.argtypes = [VP, VP, U64, U64, VP, VP, U64, P(U64), P(U64)]
    L.fgx_record_boundaries_device.restype = I
    L.fgx_run_bam.argtypes = [VP, C.c_char_p, C.c_char_p, VP, U64, P(GroupOptions), U32, I, U64, U32, P(BamRunStats)]
    L.fgx_run_bam.restype = I
    L.fgx_run_bam_rejects.argtypes = [VP, C.c_char_p, C.c_char_p, C.c_char_p, VP, U64, P(GroupOptions), U32, I, U64, U32, P(BamRunStats), P(U64)]
    L.fgx_run_bam_rejects.restype = I
    L.fgx_bgzf_recompress_file.argtypes = [C.c_char_p, C.c_char_p, U32, I, U64, P(U64)]
    L.fgx_bgzf_recompress_file.restype = I
    L.fgx_pipeline_last_error.argtypes = []
    L.fgx_pipeline_last_error.restype = C.c_char_p
    L.fgx_filter_options_default.argtypes = [P(FilterOptions)]
    L.fgx_filter_options_default.restype = None
    L.fgx_filter_records.argtypes = [VP, P(FilterOptions), VP, U64, VP, VP, U32, P(FilterOutput)]
    L.fgx_filter_records.restype = I
    L.fgx_filter_records_device.argtypes = [VP, P(FilterOptions), VP, U64, VP, VP, U32, P(FilterOutput)]
    L.fgx_filter_records_device.restype = I
    L.fgx_filter_last_output_device.argtypes = [VP, P(FilterOptions), P(FilterOutput)]
    L.fgx_filter_last_output_device.restype = I
    L.fgx_set_reference.argtypes = [VP, U32, VP, VP]
    L.fgx_set_reference.restype = I
    L.fgx_methylation_annotate_host.argtypes = [VP, VP, VP, U32, VP, U32, VP, U64, I, U32, VP, VP, VP]
    L.fgx_methylation_annotate_host.restype = I
    L.fgx_methylation_runs_host.argtypes = [VP, U32, C.c_int64, I, VP, U32, VP, U32]
    L.fgx_methylation_runs_host.restype = U32
    L.fgx_methylation_mm_ml_host.argtypes = [VP, U32, VP, VP, VP, I, I, VP, U32, VP, U32]
    L.fgx_methylation_mm_ml_host.restype = I
    L.fgx_canon_duplex_host.argtypes = [VP, VP, VP, VP, U32, VP, VP, VP]
    L.fgx_canon_duplex_host.restype = I
    L.fgx_canon_codec_host.argtypes = [VP, VP, VP, VP, U32, VP, VP]
    L.fgx_canon_codec_host.restype = I
    L.fgx_simplex_rejects_host.argtypes = [VP, VP, VP, VP, VP, U32, VP, U64, VP, VP]
    L.fgx_simplex_rejects_host.restype = I
    L.fgx_strand_rejects_host.argtypes = [VP, VP, VP, VP, VP, U32, VP, VP, U64, VP, VP]
    L.fgx_strand_rejects_host.restype = I
    # host-only helpers (not part of the public header; used by CPU-side tests)
    L.fgx_build_tables_host.argtypes = [U8, U8, I, VP, P(U32), VP]
    L.fgx_build_tables_host.restype = I
    L.fgx_host_libm_array.argtypes = [I, VP, VP, U64]
    L.fgx_host_libm_array.restype = None
    L.fgx_debug_last_deferral.argtypes = [VP, VP]
    L.fgx_debug_last_deferral.restype = None
    L.fgx_set_general_only.argtypes = [VP, I]
    L.fgx_set_general_only.restype = None
    L.fgx_set_fast_lds_bytes.argtypes = [VP, U32]
    L.fgx_set_fast_lds_bytes.restype = None
    _lib = L
    return L


class _Lazy:
    def __getattr__(self, name):
        return getattr(load(), name)


lib = _Lazy()


def default_options(**kw):
    o = Options()
    load().fgx_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def hip_memcpy_d2h(dev_ptr: int, n: int) -> bytes:
    """Copy `n` bytes from a device pointer to host (libamdhip64 hipMemcpy)."""
    if n == 0:
        return b""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    buf = C.create_string_buffer(n)
    rc = hip.hipMemcpy(buf, C.c_void_p(dev_ptr), n, 2)  # hipMemcpyDeviceToHost
    if rc != 0:
        raise RuntimeError(f"hipMemcpy D2H failed: {rc}")
    return buf.raw
