/* fgumi_amd.h — C ABI of the MI355X-native consensus engine (drop-in for fgumi's
 * per-UMI-family consensus hot path).
 *
 * The reference has no FFI surface (100% Rust, `#![deny(unsafe_code)]`), so every entry
 * point below replaces a *Rust* interface; the citation names it (paths relative to the
 * reference checkout).  INTEGRATION.md shows the `extern "C"` binding a maintainer would add
 * on the Rust side.
 *
 * Seam: the BATCH level, i.e. the `process_fn: Fn(MiGroupBatch) -> io::Result<ProcessedBatch>`
 * closure of `fgumi simplex|duplex|codec`
 *   (src/lib/commands/simplex.rs:637-718, duplex.rs:742ff, codec.rs:722ff),
 * which wraps `ConsensusCaller::consensus_reads` (crates/fgumi-consensus/src/caller.rs:220-252)
 * and `apply_overlapping_consensus` (crates/fgumi-consensus/src/overlapping.rs:627-684).
 * A per-family call would mean one kernel launch per molecule; a GPU wants >= 1e5 families per
 * call, so a caller should aggregate many reference-sized batches (50/100/1000 groups).
 *
 * Plain pointers and sizes only; no torch / HIP types.  Device pointers are `void*`.
 */
#ifndef FGUMI_AMD_H
#define FGUMI_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RejectionReason, in declaration order (crates/fgumi-consensus/src/caller.rs:401-446). */
enum {
  FGX_REJ_FRAGMENT_READ = 0, FGX_REJ_INSUFFICIENT_READS, FGX_REJ_QUALITY_TOO_LOW, FGX_REJ_UNMAPPED,
  FGX_REJ_MAPPED, FGX_REJ_TOO_MANY_NS, FGX_REJ_MINORITY_ALIGNMENT, FGX_REJ_SECONDARY_OR_SUPPLEMENTARY,
  FGX_REJ_FAILED_QC, FGX_REJ_MISSING_UMI, FGX_REJ_QUALITY_TRIMMED, FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING,
  FGX_REJ_INSUFFICIENT_OVERLAP, FGX_REJ_ORPHAN_CONSENSUS, FGX_REJ_INDEL_ERROR_BETWEEN_STRANDS,
  FGX_REJ_CLIP_OVERLAP_FAILED, FGX_REJ_HIGH_DUPLEX_DISAGREEMENT, FGX_REJ_POTENTIAL_COLLISION,
  FGX_REJ_NOT_PRIMARY_FR_PAIR, FGX_REJ_DOWNSAMPLED, FGX_REJ_OTHER, FGX_N_REJECTION
};

/* stats[] layout returned with every batch:
 *   [0] total_reads  [1] consensus_reads  [2] filtered_reads      (ConsensusCallingStats, caller.rs:256-321)
 *   [3 .. 3+21)      per-RejectionReason counters, enum order above
 *   [24..28)         simplex / duplex: CorrectionStats of the overlapping pre-step: overlapping_bases,
 *                    bases_agreeing, bases_disagreeing, bases_corrected   (overlapping.rs:51-60)
 *                    CODEC: consensus_bases_emitted, consensus_duplex_bases_emitted,
 *                    duplex_disagreement_base_count, consensus_reads_rejected_hdd (codec_caller.rs:264-310) */
#define FGX_STATS_LEN 28

enum { FGX_CALLER_SIMPLEX = 0, FGX_CALLER_DUPLEX = 1, FGX_CALLER_CODEC = 2 };
enum { FGX_TIE_FGBIO_COMPAT = 0, FGX_TIE_ULP_RELATIVE = 1 };  /* base_builder.rs:418-438 */
enum { FGX_METHYLATION_DISABLED = 0, FGX_METHYLATION_EM_SEQ = 1, FGX_METHYLATION_TAPS = 2 };  /* MethylationMode, crates/fgumi-consensus/src/lib.rs:45-68 */

/* Options = VanillaUmiConsensusOptions (vanilla_caller.rs:292-334) + the command-level knobs the
 * process_fn closure captures (read_name_prefix, read_group_id, overlapping on/off, rejects
 * tracking; simplex.rs:590-611) + duplex (duplex_caller.rs:344-513) and CODEC
 * (codec_caller.rs:145-223) option blocks.  Zero-initialise, then call fgx_options_default(). */
typedef struct fgx_options {
  uint32_t struct_size;                 /* = sizeof(fgx_options); ABI guard */
  uint32_t caller_kind;                 /* FGX_CALLER_* */
  char     tag[2];                      /* "MI" */
  char     cell_tag[2];                 /* "CB" as the commands pass it; {0,0} = None */
  uint8_t  error_rate_pre_umi;          /* 45 */
  uint8_t  error_rate_post_umi;         /* 40 */
  uint8_t  min_input_base_quality;      /* 10 */
  uint8_t  min_consensus_base_quality;  /* CLI default 2 (common.rs:749-806) */
  uint8_t  produce_per_base_tags;       /* 1 */
  uint8_t  trim;                        /* 0 */
  uint8_t  tie_rule;                    /* FGX_TIE_FGBIO_COMPAT */
  uint8_t  overlapping_consensus;       /* 1: simplex/duplex default on */
  uint8_t  track_rejects;               /* 0 */
  uint8_t  methylation_mode;            /* FGX_METHYLATION_* (vanilla_caller.rs:323-326; --methylation-mode, simplex.rs:240-245); needs fgx_set_reference.
                                           simplex and duplex only — CODEC has no methylation mode (codec_caller.rs:425) */
  uint8_t  _pad0[2];
  uint32_t min_reads;                   /* simplex --min-reads (required by the CLI) */
  int64_t  max_reads;                   /* -1 = None */
  const char* read_name_prefix;         /* "" when the header has no @RG (caller.rs:612-645) */
  const char* read_group_id;            /* "A" */
  /* duplex (duplex_caller.rs:465-489) */
  uint32_t duplex_min_reads[3];         /* [total, XY, YX] */
  int64_t  duplex_max_reads_per_strand; /* -1 = None */
  /* codec (codec_caller.rs:145-200) */
  uint32_t codec_min_reads_per_strand;  /* 1 */
  int64_t  codec_max_reads_per_strand;  /* -1 = None */
  uint32_t codec_min_duplex_length;     /* 1 */
  uint8_t  codec_single_strand_qual;    /* 0 = None */
  uint8_t  codec_outer_bases_qual;      /* 0 = None */
  uint8_t  codec_has_single_strand_qual;
  uint8_t  codec_has_outer_bases_qual;
  uint32_t codec_outer_bases_length;    /* 5 */
  uint32_t codec_max_duplex_disagreements;   /* usize::MAX in the reference => 0xFFFFFFFF */
  double   codec_max_duplex_disagreement_rate; /* 1.0 */
  int32_t  device;                      /* HIP device ordinal; -1 = current */
  uint32_t _pad1;
} fgx_options;

/* What one batch returns: `ConsensusOutput { data, count }` (caller.rs:172-177) + the stats the
 * closure merges per batch (simplex.rs:711-717) + `take_rejected_reads` (vanilla_caller.rs:529).
 * All pointers stay owned by the caller object and are valid until its next call / destroy. */
typedef struct fgx_output {
  const uint8_t* data;       /* concatenated BAM records, each prefixed by LE u32 block_size, input group order */
  uint64_t data_len;
  uint64_t count;            /* number of consensus records in `data` */
  uint64_t stats[FGX_STATS_LEN];
  const uint8_t* rejects;    /* when track_rejects: rejected input records, each with block_size prefix */
  uint64_t rejects_len;
  uint64_t n_rejects;
  /* timing of the last call, milliseconds (0 when not measured) */
  double ms_host_prep, ms_h2d, ms_kernels, ms_d2h, ms_emit;
  /* device-resident pipeline: HIP-event time of its two kernels on the caller's stream */
  double ms_k_family, ms_k_emit;
} fgx_output;

typedef struct fgx_caller fgx_caller;

/* Fill `o` with the reference defaults (src/lib/commands/common.rs:749-806, 894-904). */
void fgx_options_default(fgx_options* o);

/* Replaces `VanillaUmiConsensusCaller::new_with_rejects_tracking` (vanilla_caller.rs:432-463),
 * `DuplexConsensusCaller::new` and `CodecConsensusCaller::new`.  Builds the Phred tables
 * (base_builder.rs:349-370, 615-656, 743-754; vanilla_caller.rs:469-501), uploads them, and
 * creates the HIP stream.  Returns NULL on error (see fgx_global_error()).  Fails loudly when
 * no HIP device is usable — there is no CPU fallback. */
fgx_caller* fgx_create(const fgx_options* opts);
void fgx_destroy(fgx_caller* c);
/* The check fgx_create runs first: the engine's bit-exact libm port (glibc 2.35 exp / log / log1p / expm1, what a reference
 * build links on this image — phred.rs:158-187 through Rust std) against THIS process's libm on 7168 fixed points.
 * 0 = identical; 1 = they differ, `msg` names the glibc version and the first differing point, and fgx_create refuses to
 * hand out a caller unless FGX_ALLOW_LIBM_MISMATCH=1 (results would no longer equal a reference build on this box). */
int fgx_libm_self_check(char* msg, uint64_t msg_cap);

/* Replaces `VanillaUmiConsensusCaller::set_reference(reference, ref_names)` (vanilla_caller.rs:512-522) and
 * `DuplexConsensusCaller::set_reference` (duplex_caller.rs:524-536) after `load_methylation_reference` (src/lib/commands/common.rs:108-144):
 * the reference genome of the methylation-aware mode.  Contig i of the BAM header (= a record's ref_id) is seqs[i], lens[i] bases as the
 * FASTA holds them (any case).  seqs[i] == NULL stands for a contig the FASTA does not hold: an empty contig, every base unknown — the
 * caller-level `set_reference` accepts any provider; failing fast on missing contigs is the CLI loader's job (common.rs:131-141).  The sequences are copied into
 * HBM once (one byte per base; a human genome is 3.1 GB of the 288) and every batch's annotation kernel reads them there.  n_ref = 0
 * drops the reference.  With a methylation mode set and no reference, reads are called without annotation or tags, as the reference
 * does (annotate_and_normalize, vanilla_caller.rs:792-797).  Returns 0, or non-zero with fgx_last_error(c). */
int fgx_set_reference(fgx_caller* c, uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens);
const char* fgx_last_error(const fgx_caller* c);
const char* fgx_global_error(void);

/* Replaces the Process-step closure `process_fn(MiGroupBatch)`
 * (src/lib/commands/simplex.rs:637-718): for each MI group — min-reads short-circuit,
 * overlapping-bases pre-correction, `consensus_reads` — concatenating output in group order.
 *
 *   records   : blob holding the raw BAM records of the batch (host memory)
 *   rec_off   : n_rec byte offsets of each record body (the bytes AFTER its block_size prefix)
 *   rec_len   : n_rec record body lengths (= block_size)
 *   grp_first : n_grp+1 record indices; group g = records [grp_first[g], grp_first[g+1])
 *               (one group = one MiGroup, src/lib/mi_group.rs:22-57: consecutive records with the
 *               same MI — duplex: same MI with /A,/B stripped)
 * Records are not modified (the overlap pre-step works on device/host copies).
 * Returns 0 on success; non-zero is fatal for the run, like the reference's `anyhow::Error`
 * (simplex.rs:699-701).  Message via fgx_last_error(). */
int fgx_process_batch(fgx_caller* c, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                      const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp,
                      fgx_output* out);

/* Same contract with every input array already resident in HBM (device pointers) and the
 * consensus records left in HBM: the measured configuration of bench.py and the multi-GPU path.
 * `out->data` is a DEVICE pointer; `out->stats` is copied back (224 bytes).  Families the device
 * pipelines do not decide (reads with more than 6 CIGAR ops, unmapped reads, malformed records; for
 * duplex / CODEC also what the canonical form does not cover) are reported in
 * *n_deferred / d_deferred_groups and must be re-submitted through fgx_process_batch; 0 for
 * `simulate`-shaped input.  Duplex / CODEC molecules with indels or clips are canonicalised and decided in a second device pass
 * inside this entry (default since round 4; FGX_DUPLEX_CANON / FGX_CODEC_CANON / FGX_CANON_RESIDENT = 0 opt out).  `track_rejects`
 * is accepted for the simplex caller (side kernels, fgumi_amd/csrc/reject_device.hip; FGX_REJECTS_DEVICE=0 opts out and the entry then
 * refuses `track_rejects`): `out->rejects` is then a DEVICE pointer too and
 * covers every group, the deferred ones included.  The kernels stage a family's bytes in whole 16-byte pieces: `d_records` must be
 * READABLE for 16 bytes past `records_len` (any allocation larger than the stream by 16 bytes will do; the
 * bytes are never interpreted).  The host entry above pads its own device copy. */
int fgx_process_batch_device(fgx_caller* c, const void* d_records, uint64_t records_len, const void* d_rec_off,
                             const void* d_rec_len, uint32_t n_rec, const void* d_grp_first, uint32_t n_grp,
                             fgx_output* out, uint32_t* n_deferred, const void** d_deferred_groups);

/* ---- column-level entry (ConsensusBaseBuilder, base_builder.rs:775-1081) -----------------
 * Calls `n_cols` independent columns on the device: column j has observations
 * (bases[j*depth + i], quals[j*depth + i]) for i < depth, added in that order; bases are ASCII
 * ('N'/other = ignored, as `add` does).  Outputs the raw call() result (before thresholds),
 * contributions() and depth - observations_for_base(base).  Used by the parity tests that
 * mirror base_builder.rs's own unit tests. */
int fgx_call_columns(fgx_caller* c, const uint8_t* bases, const uint8_t* quals, uint32_t n_cols, uint32_t depth,
                     uint8_t* out_base, uint8_t* out_qual, uint32_t* out_depth, uint32_t* out_errors);

/* ---- methylation-aware mode, host-side test hooks (fgumi_amd/csrc/methylation_core.h) ---------------------------------------
 * The annotation kernel's per-position body (reference base lookup through the anchor read's aligned runs, unconverted / converted
 * counts over the source reads, in-place normalisation of converted bases: methylation.rs:116-178, 193-242; vanilla_caller.rs:838-852)
 * run on the host, lane by lane — the same source the device compiles, for tests without a device.
 *   stage        : the staged source reads (bases at [off, off+len) of each read), normalised IN PLACE
 *   read_off/len : n_reads staged reads
 *   runs         : n_runs aligned runs of the anchor, 4 int64 each: query start, length, reference position of the first base
 *                  (0-based, may lie outside the contig), step (+1 forward read, -1 reverse read)
 *   contig       : the anchor's contig (contig_len bases); top_strand as `is_top_strand` (methylation.rs:392-398)
 * Fills is_ref_c / unconverted / converted [n_pos]. */
int fgx_methylation_annotate_host(uint8_t* stage, const uint64_t* read_off, const uint32_t* read_len, uint32_t n_reads, const int64_t* runs, uint32_t n_runs,
                                  const uint8_t* contig, uint64_t contig_len, int top_strand, uint32_t n_pos, uint8_t* is_ref_c, uint32_t* unconverted,
                                  uint32_t* converted);
/* The aligned runs the host side derives from a SourceRead (query_to_ref_positions, methylation.rs:116-178): `simplified` = the
 * read's simplified CIGAR after reversal and truncation, `original` = before, both as BAM-encoded ops (len << 4 | code).  Writes up
 * to `cap` runs of 4 int64 and returns the number of runs. */
uint32_t fgx_methylation_runs_host(const uint32_t* simplified, uint32_t n_s, int64_t alignment_start, int is_reverse, const uint32_t* original, uint32_t n_o,
                                   int64_t* runs, uint32_t cap);
/* build_mm_ml_tags (methylation.rs:264-329) as the record assembly uses it: returns the ML length and the MM string (NUL-terminated),
 * or -1 when no tag is written. */
int fgx_methylation_mm_ml_host(const uint8_t* bases, uint32_t n, const uint8_t* is_ref_c, const uint32_t* unconverted, const uint32_t* converted, int top_strand,
                               int mode, char* mm, uint32_t mm_cap, uint8_t* ml, uint32_t ml_cap);

/* ---- canonical form of a duplex molecule with indel / skip / pad CIGARs (fgumi_amd/csrc/canon_core.h), host-side test hook ------
 * Rewrites ONE duplex molecule so that the device pipeline's one-aligned-block duplex kernels decide it exactly as the reference
 * decides the original: the R1/R2 overlap pre-correction (overlapping.rs:236-336; duplex.rs:786-795) is written into the bases and
 * qualities, the mate clip (raw-bam/overlap.rs:181-268) is applied by cutting the read, reads the alignment filter rejects
 * (vanilla_caller.rs:48-120, 1242-1296) are dropped and counted, the survivors get the CIGAR `<len>M` and lose their MC tag.
 * `out` receives the canonical records at the records' own offsets (out_len[i] bytes each; 0 = dropped); delta5 = {dropped reads
 * (MinorityAlignment), overlapping_bases, bases_agreeing, bases_disagreeing, bases_corrected} to add to the statistics of the
 * canonical molecule.  Returns 0; 1 = out of scope (the molecule stays on the general path); 2 = bad arguments.  The same
 * source compiles for the device (one lane per molecule). */
int fgx_canon_duplex_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out, uint32_t* out_len,
                          uint64_t* delta5);
/* The same for ONE CODEC molecule (codec_caller.rs:625-1262): every read cut by its virtual clip against the mate in hand
 * (raw-bam/cigar.rs:404-446), `<len>M`, and PLACED so that the one-M-op CODEC kernels recompute the original's overlap geometry — the
 * consensus length above all (query positions at the end of the shared window, cigar.rs:461-500).  Every record is kept (the consensus
 * UMI is called over all of them); a molecule the original would reject, or whose filter / cap would drop a read, is out of scope. */
int fgx_canon_codec_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out, uint32_t* out_len);

/* The `--rejects` stream of the simplex caller for a batch of MI groups, from the records alone (fgumi_amd/csrc/reject_core.h): every
 * rejection the vanilla caller makes is taken before the per-position arithmetic (vanilla_caller.rs:1329-1646: secondary / supplementary,
 * --min-reads at the group and subgroup level, zero length after trimming, unmapped among mapped, minority alignments, --max-reads
 * downsampling, the orphan R1 / R2 rule), and the rejected records are the group's records after the R1 / R2 overlap pre-correction
 * (simplex.rs:685-700), block_size-prefixed, in input order (:1430-1436).  This host entry runs the scalar source the device kernels of
 * reject_device.hip run lane per group; tests compare it with the reference restatement byte for byte.  out == NULL sizes (*out_len).
 * Returns 0; 1 = a group is out of scope (> 128 records, > 16 CIGAR ops, malformed records: the general path decides the batch);
 * 2 = bad arguments or `cap` too small. */
int fgx_simplex_rejects_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first, uint32_t n_grp,
                             uint8_t* out, uint64_t cap, uint64_t* out_len, uint64_t* n_rejects);

/* The same for the duplex (o->caller_kind 1) and CODEC (2) callers (reject_core.h duplex_reject_codes / codec_reject_mask).  ONE of their decisions
 * needs the per-position arithmetic — whether the molecule gave its consensus (the per-base read-count gate after the strand combine,
 * duplex_caller.rs:2087-2120; the CODEC strand combine, codec_caller.rs:1272-1512) — so it is an input: kept[g] != 0.  On the device the pipeline's
 * own output slots say so (fgx_process_batch_device with track_rejects).  Duplex: fragments, then every /A and /B record of a molecule without
 * consensus, or the zero-length and filtered reads of one with (duplex_caller.rs:1944-2120, 2545-2610; overlap-corrected copies when duplex.rs:786-795
 * ran the pre-step).  CODEC: a mask in input order (codec_caller.rs:1767-1834). */
int fgx_strand_rejects_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first, uint32_t n_grp,
                            const uint8_t* kept, uint8_t* out, uint64_t cap, uint64_t* out_len, uint64_t* n_rejects);

/* Multi-GPU from a host that is not Python (one process / one fgx_caller per GPU): contiguous shards of a weighted family
 * stream — weights[i] = record bytes of family i (fgx_sim_family_bytes for simulated input; Σ rec_len per group otherwise) — with roughly
 * equal total weight: shard k = families [cuts[k], cuts[k + 1]), cuts has world + 1 entries.  The same cuts as the Python mirror
 * (fgumi_amd/distributed.py balanced_shards) that bench.py --scaling strong uses.  Families are independent: no collective on the data path;
 * output is SO:unsorted in input order, so the shard payloads concatenated in rank order ARE the whole output. */
int fgx_balanced_shards(const uint64_t* weights, uint32_t n, uint32_t world, uint32_t* cuts);

/* Device self-test of the glibc-compatible libm: op 0 exp, 1 log, 2 log1p, 3 expm1. */
int fgx_device_libm(fgx_caller* c, int op, const double* x, double* y, uint64_t n);

/* Host copies of the tables the kernels use (94 entries each), for parity tests:
 * which: 0 correct[q], 1 error_per_alt[q], 2 gap thresholds, 3 cerr_min; *cap = pre-UMI cap. */
int fgx_get_table(const fgx_caller* c, int which, double* out94, uint32_t* cap);

/* fgumi_amd/csrc/aln_tags_core.h on the host (the scalar source a GPU lane per record runs for `fgumi filter --ref`): NM / UQ / MD of one record
 * regenerated against the given contigs as regenerate_alignment_tags_raw does (crates/fgumi-sam/src/alignment_tags.rs:259-433): 0 = regenerated,
 * 1 = tags removed (unmapped / no reference id), >= 2 = the reference's fatal errors (too short, reference id not in the header, invalid start,
 * alignment leaves the contig, truncated record, CIGAR longer than the sequence), -1 = `cap` too small (*out_len = bytes needed). */
int fgx_regenerate_alignment_tags_host(const uint8_t* rec, uint32_t len, uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens, uint8_t* out, uint32_t cap,
                                       uint32_t* out_len);

/* ---- synthetic grouped reads (`fgumi simulate grouped-reads` model) ------------------------
 * Restates the record SHAPE of src/lib/commands/simulate/grouped_reads.rs:666-1011 and the
 * quality model of src/lib/simulate/quality.rs:60-135 with a counter-based integer RNG so the
 * same molecule is generated bit-identically on host and device (the reference's ChaCha12
 * stream cannot be reproduced; seeds are this build's own). */
typedef struct fgx_sim_params {
  uint64_t seed;            /* 42 */
  uint32_t n_families;
  uint32_t read_length;     /* 150 */
  uint32_t family_size;     /* pairs per molecule when family_size_max == 0 */
  uint32_t family_size_max; /* >0: long-tail sizes in [family_size, family_size_max], count ~ size^-1.5 */
  uint32_t duplex;          /* 1: /A,/B strand split like --duplex */
  uint32_t insert_mean;     /* 300 */
  uint32_t insert_sd;       /* 50  */
  uint32_t error_rate_ppm;  /* per-base substitution probability * 1e6 (1000 = 0.001) */
  uint32_t first_family;    /* molecule id of family 0 (sharding) */
  uint32_t codec;           /* 1: the reverse mate's SEQ is stored in reference orientation (aligner convention) so overlapping
                               mates agree — the CODEC workload; 0 keeps the reference simulator's RC(template) bytes */
} fgx_sim_params;

/* MI grouping of a record stream — replaces `MiGrouper::add_records` / `MiGroupIterator` (src/lib/mi_group.rs:227-310, 414-520)
 * together with the pre-group record filter of the consensus commands (src/lib/commands/common.rs:384-397): records are
 * kept unless secondary / supplementary (always), unmapped (unless allow_unmapped) or without the group tag; a group is a
 * run of consecutive kept records with an equal key, key = tag value (cut at its last '/' when strip_strand_suffix is set:
 * duplex, crates/fgumi-umi/src/lib.rs:370-375) + '\t' + cell-tag value when cell_tag is configured.  Produces exactly the
 * arrays fgx_process_batch takes: the kept records' offsets / lengths (input order) and grp_first[n_grp + 1].
 * Output arrays must hold n_rec (+1 for grp_first) entries.  The *_device variant takes and fills device pointers. */
typedef struct fgx_group_options {
  char    tag[2];               /* "MI" */
  char    cell_tag[2];          /* "CB" for all three commands; {0,0} = none */
  uint8_t strip_strand_suffix;  /* 1 for duplex */
  uint8_t allow_unmapped;       /* --allow-unmapped */
  uint8_t _pad[2];
} fgx_group_options;
int fgx_group_records_device(fgx_caller* c, const fgx_group_options* g, const void* d_records, uint64_t records_len, const void* d_rec_off,
                             const void* d_rec_len, uint32_t n_rec, void* d_out_rec_off, void* d_out_rec_len, void* d_grp_first,
                             uint32_t* n_kept, uint32_t* n_grp);
int fgx_group_records(fgx_caller* c, const fgx_group_options* g, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                      const uint32_t* rec_len, uint32_t n_rec, uint64_t* out_rec_off, uint32_t* out_rec_len, uint32_t* grp_first,
                      uint32_t* n_kept, uint32_t* n_grp);

/* `fgumi filter` on a stream of unmapped consensus records — replaces the Process step of the filter command
 * (src/lib/commands/filter.rs:581-625 single-read mode, :653-731 template mode) with `Filter::process_record_raw`
 * (:762-940, no --ref, no methylation filters) and the fgumi-consensus filter functions it calls
 * (crates/fgumi-consensus/src/filter.rs: filter_read :523-551, filter_duplex_read :558-637, mask_bases :765-811,
 * mask_duplex_bases :824-923, mean_base_quality_full_length :688-705, compute_read_stats :646-671, template_passes
 * :371-395, retained_primary_masked_bases :419-442) plus `reverse_per_base_tags_raw` (src/lib/tag_reversal.rs:27-67).
 * Thresholds arrive already expanded to [duplex (CC), AB, BA] (FilterConfig::new, filter.rs:237-329; the single-strand
 * thresholds are index 0).  Records are masked IN PLACE in `records` (the reference mutates its RawRecords the same way),
 * then kept records are concatenated (each with its block_size prefix, template order R1,R2,supplementaries,secondaries as
 * Template::from_records builds it, src/lib/template.rs:170-352) into `data`, rejected ones into `rejects` when
 * track_rejects is set.  Errors are fatal like the reference's: a mapped record ("--ref is required ..."), a record
 * without cD/cE ("... consensus calling tags (cD/cE) ..."), two primary R1s or R2s in a template. */
typedef struct fgx_filter_options {
  uint32_t struct_size;                   /* = sizeof(fgx_filter_options) */
  uint32_t min_reads[3];                  /* --min-reads, expanded [CC, AB, BA] */
  double   max_read_error_rate[3];        /* --max-read-error-rate (default 0.025) */
  double   max_base_error_rate[3];        /* --max-base-error-rate (default 0.1) */
  double   min_mean_base_quality;         /* --min-mean-base-quality when has_min_mean_base_quality */
  double   max_no_call_fraction;          /* --max-no-call-fraction (default 0.2); >= 1.0 = absolute count */
  uint8_t  has_min_base_quality, min_base_quality;   /* --min-base-quality (optional) */
  uint8_t  has_min_mean_base_quality;
  uint8_t  require_single_strand_agreement;           /* -s */
  uint8_t  reverse_per_base_tags;                     /* -R */
  uint8_t  filter_by_template;                        /* --filter-by-template (default true) */
  uint8_t  track_rejects;                             /* --rejects given */
  uint8_t  regenerate_alignment_tags;                 /* --ref given (filter.rs:115-118, 888-890): mapped records are accepted and their NM / UQ / MD tags are
                                                         recomputed after the masking against the reference handed over with fgx_set_reference (contig i of the
                                                         BAM header = seqs[i]); on unmapped records the three tags are removed (regenerate_alignment_tags_raw,
                                                         crates/fgumi-sam/src/alignment_tags.rs:259-433).  0: a mapped record is the reference's fatal error */
  /* the methylation (EM-Seq / TAPs) filters (src/lib/commands/filter.rs:181-206, 833-937; crates/fgumi-consensus/src/filter.rs:925-1340) */
  uint8_t  has_min_methylation_depth;                 /* --min-methylation-depth given: a base whose cu+ct (duplex: also au+at, bu+bt) is below its threshold is masked */
  uint8_t  require_strand_methylation_agreement;      /* duplex records: both bases of a reference CpG are masked when the AB strand (au/at at the C) and the BA strand
                                                         (bu/bt at the G) call the methylation differently; needs the reference (regenerate_alignment_tags) */
  uint8_t  has_min_conversion_fraction;               /* --min-conversion-fraction given (needs the reference and methylation_mode) */
  uint8_t  methylation_mode;                          /* FGX_METHYLATION_*: EM-Seq counts ct, TAPs cu at non-CpG reference Cs; DISABLED = the check passes */
  uint32_t min_methylation_depth[3];                  /* [duplex, AB, BA], expanded like min_reads */
  double   min_conversion_fraction;
} fgx_filter_options;
void fgx_filter_options_default(fgx_filter_options* o);   /* min_reads {1,1,1}; everything else the CLI defaults */

typedef struct fgx_filter_output {
  const uint8_t* data;      uint64_t data_len;      /* kept records (host entry: host memory; device entry: device memory) */
  const uint8_t* rejects;   uint64_t rejects_len;   /* rejected records when track_rejects */
  uint64_t records_count, passed_count, bases_masked;   /* FilterProcessedBatchRaw counters (filter.rs:226-238) */
  uint64_t rejected_count;
} fgx_filter_output;
/* Host buffers in and out; `records` is NOT modified (the library masks its device copy).  Buffers in `out` stay valid
 * until the next call on the handle. */
int fgx_filter_records(fgx_caller* c, const fgx_filter_options* f, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                       const uint32_t* rec_len, uint32_t n_rec, fgx_filter_output* out);
/* Device buffers in (masked in place) and out (`out->data` / `out->rejects` are device pointers owned by the handle). */
int fgx_filter_records_device(fgx_caller* c, const fgx_filter_options* f, void* d_records, uint64_t records_len, const void* d_rec_off,
                              const void* d_rec_len, uint32_t n_rec, fgx_filter_output* out);

/* The consensus records the handle's last fgx_process_batch_device call left in HBM, filtered in place (consensus → filter
 * without leaving the device: the hand-over the reference does through a BAM file or a pipe). */
int fgx_filter_last_output_device(fgx_caller* c, const fgx_filter_options* f, fgx_filter_output* out);

/* ---- BGZF container on the host cores (crates/fgumi-bgzf/src/{reader,writer}.rs for this path) ------------------------------
 * Block-parallel over zlib: `threads` workers (0 = all cores) take the independent <= 64 KiB gzip members of a BGZF file image.
 * Both return 0 and a malloc'ed buffer the caller releases with fgx_bgzf_free, or 1 (fgx_bgzf_last_error, thread-local).
 * inflate verifies every block's CRC32 and ISIZE; deflate cuts the stream every 0xff00 bytes, level 1 being the reference's
 * default for consensus output, and appends the 28-byte EOF marker when `with_eof`. */
int fgx_bgzf_inflate(const uint8_t* raw, uint64_t raw_len, uint32_t threads, uint8_t** out, uint64_t* out_len);
int fgx_bgzf_deflate(const uint8_t* in, uint64_t len, int level, uint32_t threads, int with_eof, uint8_t** out, uint64_t* out_len);
void fgx_bgzf_free(uint8_t* p);
const char* fgx_bgzf_last_error(void);

/* FindBoundaries (src/lib/unified_pipeline/bam.rs): walks the `block_size` chain of an uncompressed BAM record stream from
 * byte `start`; fills rec_off (BODY offsets, past the 4-byte prefix) and rec_len for up to `cap` records and sets *n_rec to
 * the number of records in the stream (call with cap = 0 to count).  Returns 0, or 1 when the stream ends inside a record. */
int fgx_record_boundaries(const uint8_t* stream, uint64_t stream_len, uint64_t start, uint64_t* rec_off, uint32_t* rec_len,
                          uint64_t cap, uint64_t* n_rec);

/* FindBoundaries with the stream resident in HBM (fgumi_amd/csrc/boundaries.hip): the same chain, found by a thread per 16 KiB
 * segment — each proposes where its first record starts, walks on from there, and the walks are checked against each other
 * from the (given) first record on until the table is exactly the sequential walk's.  A record that does not end inside the
 * stream is NOT an error here (streaming: it continues in the next chunk): *consumed = the offset just past the last whole
 * record.  Returns 0; 1 = a record with block_size < 32 (fgx_last_error); 2 = `cap` too small (*n_rec = records present;
 * cap = 0 with null arrays just counts). */
int fgx_record_boundaries_device(fgx_caller* c, const void* d_stream, uint64_t stream_len, uint64_t start, void* d_rec_off,
                                 void* d_rec_len, uint64_t cap, uint64_t* n_rec, uint64_t* consumed);

/* A BAM file in, a consensus BAM file out (fgumi_amd/csrc/pipeline.cpp): read -> BGZF inflate (worker pool, pinned buffers) ->
 * upload -> record boundaries -> MI grouping -> consensus batch -> download -> BGZF deflate -> write, as five overlapping stages
 * over chunks of `chunk_raw_bytes` compressed bytes (0 = 512 MiB: the device inflate runs a lane per BGZF block, so a chunk should hold
 * tens of thousands of them).  Stands in, for this path, for the reader / FindBoundaries /
 * group / process / compress / write steps of src/lib/unified_pipeline/bam.rs around `process_fn`.  `out_header` = the
 * uncompressed BAM header of the output ("BAM\1", l_text, text, n_ref = 0): written as its own BGZF block(s).  A group that
 * reaches the end of a chunk waits for the next chunk (it may continue there).  `threads` = pool size (0 = all cores).
 * Returns 0, or non-zero with fgx_last_error(c). */
typedef struct fgx_bam_run_stats {
  uint64_t kept_records, groups, consensus_records, deferred_groups, chunks;
  uint64_t in_bytes, inflated_bytes, out_bytes, out_file_bytes;
  uint64_t stats[FGX_STATS_LEN];
  double seconds_total;
  double seconds_read, seconds_inflate, seconds_device, seconds_deflate, seconds_write;     /* busy time of the five stage threads */
  double seconds_h2d, seconds_boundaries, seconds_grouping, seconds_consensus, seconds_d2h; /* inside the device stage */
  double seconds_device_inflate;                                                            /* inside the device stage (0 with FGX_RUN_HOST_INFLATE); h2d and this one are sums of
                                                                                               every chunk's own upload / inflate time: chunks on their way in at once overlap */
  uint32_t boundary_repair_rounds, device_inflate;
  double seconds_device_deflate;                                                            /* inside the device stage (FGX_RUN_DEVICE_DEFLATE) */
  uint32_t device_deflate;
  uint32_t host_entry_batches;                                                              /* batches that went through the host entry in one piece (deferred families without the subset way; --rejects / methylation the device entry refused) */
} fgx_bam_run_stats;
#define FGX_RUN_HOST_INFLATE   1u   /* flags: inflate the BGZF blocks on the host cores (zlib) instead of on the device */
#define FGX_RUN_DEVICE_DEFLATE 2u   /* flags: compress the consensus records on the device too (level 1 only; fgumi_amd/csrc/deflate_core.h, a lane
                                       per block): an eighth of the bytes comes back over PCIe and the host only writes.  Off by default: with
                                       the same compressor on the host's cores the device stage is the one that bounds the pipeline */
int fgx_run_bam(fgx_caller* c, const char* in_path, const char* out_path, const uint8_t* out_header, uint64_t out_header_len,
                const fgx_group_options* g, uint32_t threads, int level, uint64_t chunk_raw_bytes, uint32_t flags, fgx_bam_run_stats* st);
/* fgx_run_bam with the reference's `--rejects <file>` (src/lib/commands/simplex.rs:7-12, 260-285, 613-720; duplex.rs / codec.rs alike): a second
 * BGZF BAM that advertises the INPUT header (its bytes as the input holds them: @RG / @PG / contigs preserved) and holds the rejected input
 * records in batch-input order — the records of MI groups below --min-reads as they stand, and the caller's rejects (overlap-corrected copies
 * where the pre-correction ran).  `c` must have been created with track_rejects; rejects_path = NULL is fgx_run_bam.  The simplex caller's
 * rejects come from the side kernels of the device entry (fgumi_amd/csrc/reject_device.hip; the duplex / CODEC callers' too since round 6: reject_core.h duplex_reject_codes / codec_reject_mask
 * with the batch's own output slots); a batch they refuse (st->host_entry_batches counts them) goes through the host entry in one piece.  *rejected_records (may be NULL) = records written to the rejects file. */
int fgx_run_bam_rejects(fgx_caller* c, const char* in_path, const char* out_path, const char* rejects_path, const uint8_t* out_header,
                        uint64_t out_header_len, const fgx_group_options* g, uint32_t threads, int level, uint64_t chunk_raw_bytes, uint32_t flags,
                        fgx_bam_run_stats* st, uint64_t* rejected_records);
/* The device's DEFLATE decoder (fgumi_amd/csrc/inflate_core.h, one GPU lane per BGZF block) run on the host — the same source, for
 * tests without a device.  `in` must be readable for 8 bytes past in_len; `out_len` = the block's ISIZE.  Returns the decoder's
 * status (0 = ok) and, in *crc_out, the CRC-32 of the output computed with the device's 64-slice fold. */
int fgx_inflate_block_host(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, uint32_t* crc_out);
/* The two-phase form of the device decoder (round 5: k_bgzf_tokenize + k_bgzf_resolve) on the host, same contract: the decoder leaves the
 * literals in place and a list of 32-bit match entries (*n_entries of them), a second pass plays the list — mode 0: in order
 * (inflate_resolve), mode 1: in k_bgzf_resolve's batches of 64 entries with its frontier rule, emulated lane by lane (*rounds = copy rounds). */
int fgx_inflate_block_two_phase_host(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, int mode, uint32_t* n_entries, uint32_t* rounds);
/* The device's DEFLATE compressor (fgumi_amd/csrc/deflate_core.h, one GPU lane per BGZF block: greedy LZ77 + one dynamic Huffman
 * code per block) run on the host, for tests without a device.  `in` readable for 8 bytes past n (n <= 65535).  Returns the bytes
 * written to `out`, or 0 when they do not fit `cap` (the caller stores the block). */
uint32_t fgx_deflate_block_host(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap);
/* The device's BGZF inflate (+ CRC-32 check) alone (fgumi_amd/csrc/bgzf_device.hip), for measurements and tests: the whole blocks of
 * raw[0 .. raw_len) go to the device once and are inflated `reps` times; *ms = device time of one pass, *inflated_len = bytes produced; with
 * `out` (inflated_cap bytes) the inflated stream comes back.  Returns 0, or non-zero with fgx_last_error(c) (a failing block is named). */
int fgx_bgzf_inflate_device_bench(fgx_caller* c, const uint8_t* raw, uint64_t raw_len, uint32_t reps, double* ms, uint64_t* inflated_len, uint8_t* out,
                                  uint64_t inflated_cap);
/* The host stages alone (read, inflate, deflate, write) around a copy: re-blocks a BGZF file; needs no device. */
int fgx_bgzf_recompress_file(const char* in_path, const char* out_path, uint32_t threads, int level, uint64_t chunk_raw_bytes,
                             uint64_t* inflated_bytes);
const char* fgx_pipeline_last_error(void);

/* Sizes for a parameter set: total blob bytes (records WITH block_size prefixes) and record count. */
int fgx_sim_sizes(const fgx_sim_params* p, uint64_t* blob_len, uint64_t* n_rec);
/* Record bytes of each of the p->n_families simulated families (what a reader would weigh a family by when it cuts the
 * family stream into shards of equal work: a family's bytes are proportional to reads x length). */
int fgx_sim_family_bytes(const fgx_sim_params* p, uint64_t* bytes_per_family);
/* Host generation into caller-provided arrays (blob_len bytes; n_rec offsets/lengths; n_families+1 firsts). */
int fgx_sim_generate_host(const fgx_sim_params* p, uint8_t* blob, uint64_t* rec_off, uint32_t* rec_len,
                          uint32_t* grp_first);
/* Device generation straight into HBM (same layout, device pointers). */
int fgx_sim_generate_device(fgx_caller* c, const fgx_sim_params* p, void* d_blob, void* d_rec_off, void* d_rec_len,
                            void* d_grp_first);

#ifdef __cplusplus
}
#endif
#endif /* FGUMI_AMD_H */
