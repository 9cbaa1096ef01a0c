"""ctypes mirror of `fgx_options` / `fgx_output` / `fgx_sim_params` (include/fgumi_amd.h)."""
import ctypes as C


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


class SimParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_families", C.c_uint32), ("read_length", C.c_uint32), ("family_size", C.c_uint32),
        ("family_size_max", C.c_uint32), ("duplex", C.c_uint32), ("insert_mean", C.c_uint32), ("insert_sd", C.c_uint32),
        ("error_rate_ppm", C.c_uint32), ("first_family", C.c_uint32), ("codec", C.c_uint32),
    ]


def defaults(kind=0, min_reads=1, **kw):
    """Reference CLI defaults (src/lib/commands/common.rs:749-806): pre 45, post 40, min-input-bq 10,
    min-consensus-bq 2, per-base tags on, fgbio tie rule, overlapping consensus on."""
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.caller_kind = kind
    o.tag = b"MI"
    o.cell_tag = b"CB"
    o.error_rate_pre_umi, o.error_rate_post_umi = 45, 40
    o.min_input_base_quality, o.min_consensus_base_quality = 10, 2
    o.produce_per_base_tags, o.trim, o.tie_rule, o.overlapping_consensus, o.track_rejects = 1, 0, 0, 1, 0
    o.min_reads = min_reads
    o.max_reads = -1
    o.read_name_prefix = b""
    o.read_group_id = b"A"
    o.duplex_min_reads = (C.c_uint32 * 3)(1, 1, 0)
    o.duplex_max_reads_per_strand = -1
    o.codec_min_reads_per_strand = 1
    o.codec_max_reads_per_strand = -1
    o.codec_min_duplex_length = 1
    o.codec_outer_bases_length = 5
    o.codec_max_duplex_disagreements = 0xFFFFFFFF
    o.codec_max_duplex_disagreement_rate = 1.0
    o.device = -1
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def sim_params(n_families, family_size=3, read_length=150, seed=42, **kw):
    p = SimParams()
    p.seed, p.n_families, p.read_length, p.family_size = seed, n_families, read_length, family_size
    p.family_size_max, p.duplex, p.insert_mean, p.insert_sd, p.error_rate_ppm, p.first_family, p.codec = 0, 0, 300, 50, 1000, 0, 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p
