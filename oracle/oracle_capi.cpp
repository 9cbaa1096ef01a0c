// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
// C entry points (ctypes) over the C++ restatement of the reference CPU caller.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <chrono>
#include <thread>
#include "../include/fgumi_amd.h"
#include "oracle_vanilla.hpp"
#ifdef ORC_WITH_DUPLEX
#include "oracle_duplex.hpp"
#endif
#ifdef ORC_WITH_CODEC
#include "oracle_codec.hpp"
#endif
#include "oracle_filter.hpp"

using namespace orc;

static thread_local std::string g_err;
static std::shared_ptr<const Reference> g_reference;   // orc_set_reference: what `set_reference` hands the callers (methylation mode)

struct OrcResult {
  Bytes data;
  uint64_t count = 0;
  uint64_t stats[FGX_STATS_LEN] = {0};
  Bytes rejects;
  uint64_t n_rejects = 0;
  double seconds_workers = 0.0;   // orc_process: wall time of the worker section alone (per-batch outputs ready; before they are joined into one buffer)
};

static VanillaOptions vanilla_options_from(const fgx_options* o) {
  VanillaOptions v;
  v.tag = std::string(o->tag, 2);
  v.error_rate_pre_umi = o->error_rate_pre_umi;
  v.error_rate_post_umi = o->error_rate_post_umi;
  v.min_input_base_quality = o->min_input_base_quality;
  v.min_reads = o->min_reads;
  v.has_max_reads = o->max_reads >= 0;
  v.max_reads = o->max_reads >= 0 ? (size_t)o->max_reads : 0;
  v.produce_per_base_tags = o->produce_per_base_tags;
  v.trim = o->trim;
  v.min_consensus_base_quality = o->min_consensus_base_quality;
  v.has_cell_tag = o->cell_tag[0] != 0;
  v.cell_tag[0] = o->cell_tag[0];
  v.cell_tag[1] = o->cell_tag[1];
  v.tie_rule = o->tie_rule == FGX_TIE_ULP_RELATIVE ? TieRule::UlpRelative : TieRule::FgbioCompat;
  v.methylation_mode = o->methylation_mode;
  return v;
}

static void stats_to_array(const Stats& s, const CorrectionStats& cs, uint64_t* out) {
  out[0] += s.total_reads; out[1] += s.consensus_reads; out[2] += s.filtered_reads;
  for (int i = 0; i < N_REJECTION; i++) out[3 + i] += s.rejection[i];
  out[24] += cs.overlapping_bases; out[25] += cs.bases_agreeing; out[26] += cs.bases_disagreeing; out[27] += cs.bases_corrected;
}

static void append_reject(OrcResult& r, const uint8_t* p, size_t n) {
  uint32_t bs = (uint32_t)n;
  for (int i = 0; i < 4; i++) r.rejects.push_back((bs >> (8 * i)) & 0xFF);
  r.rejects.insert(r.rejects.end(), p, p + n);
  r.n_rejects++;
}

// One Process-step batch over groups [g0, g1): restates process_fn (simplex.rs:637-718).
static void simplex_groups(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len,
                           const uint32_t* grp_first, uint32_t g0, uint32_t g1, OrcResult& res) {
  VanillaOptions vo = vanilla_options_from(o);
  VanillaCaller caller(o->read_name_prefix ? o->read_name_prefix : "", o->read_group_id ? o->read_group_id : "A", vo,
                       o->track_rejects != 0);
  if (vo.methylation_mode != MethDisabled) caller.reference = g_reference;   // simplex.rs:405-408
  Stats batch_stats;
  CorrectionStats batch_overlap;
  for (uint32_t g = g0; g < g1; g++) {
    caller.clear();
    uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
    size_t n = r1 - r0;
    if (n < vo.min_reads) {  // simplex.rs:673-683
      batch_stats.record_input(n);
      batch_stats.record_rejection(InsufficientReads, n);
      if (o->track_rejects) for (uint32_t r = r0; r < r1; r++) append_reject(res, blob + rec_off[r], rec_len[r]);
      continue;
    }
    std::vector<Bytes> recs;
    recs.reserve(n);
    for (uint32_t r = r0; r < r1; r++) recs.emplace_back(blob + rec_off[r], blob + rec_off[r] + rec_len[r]);
    if (o->overlapping_consensus) { CorrectionStats cs; apply_overlapping_consensus(recs, cs);
      batch_overlap.overlapping_bases += cs.overlapping_bases; batch_overlap.bases_agreeing += cs.bases_agreeing;
      batch_overlap.bases_disagreeing += cs.bases_disagreeing; batch_overlap.bases_corrected += cs.bases_corrected; }
    std::vector<std::pair<const uint8_t*, size_t>> ptrs;
    for (auto& b : recs) ptrs.push_back({b.data(), b.size()});
    ConsensusOutput out = caller.consensus_reads(ptrs);
    res.data.insert(res.data.end(), out.data.begin(), out.data.end());
    res.count += out.count;
    batch_stats.merge(caller.stats);
    if (o->track_rejects) for (auto& rj : caller.rejected_reads) append_reject(res, rj.data(), rj.size());
  }
  stats_to_array(batch_stats, batch_overlap, res.stats);
}

#ifdef ORC_WITH_DUPLEX
// process_fn of `fgumi duplex` (src/lib/commands/duplex.rs:742-830)
static void duplex_groups(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len,
                          const uint32_t* grp_first, uint32_t g0, uint32_t g1, OrcResult& res) {
  DuplexOptions d;
  d.min_total = o->duplex_min_reads[0]; d.min_xy = o->duplex_min_reads[1]; d.min_yx = o->duplex_min_reads[2];
  d.min_input_base_quality = o->min_input_base_quality; d.per_base_tags = o->produce_per_base_tags; d.trim = o->trim;
  d.has_max_reads = o->duplex_max_reads_per_strand >= 0; d.max_reads = d.has_max_reads ? (size_t)o->duplex_max_reads_per_strand : 0;
  d.has_cell_tag = o->cell_tag[0] != 0; d.cell_tag[0] = o->cell_tag[0]; d.cell_tag[1] = o->cell_tag[1];
  d.pre = o->error_rate_pre_umi; d.post = o->error_rate_post_umi;
  d.tie_rule = o->tie_rule == FGX_TIE_ULP_RELATIVE ? TieRule::UlpRelative : TieRule::FgbioCompat;
  d.methylation_mode = o->methylation_mode;
  if (d.min_xy > d.min_total || d.min_yx > d.min_xy) throw OracleError{"min-reads values must be specified high to low"};
  DuplexCaller caller(o->read_name_prefix ? o->read_name_prefix : "", o->read_group_id ? o->read_group_id : "A", d, o->track_rejects != 0);
  if (d.methylation_mode != MethDisabled) caller.ss.reference = g_reference;   // duplex.rs:465-471
  const bool single_strand_allowed = d.min_yx == 0;
  Stats batch_stats;
  CorrectionStats batch_overlap;
  for (uint32_t g = g0; g < g1; g++) {
    caller.clear();
    uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
    std::vector<Bytes> recs;
    for (uint32_t r = r0; r < r1; r++) recs.emplace_back(blob + rec_off[r], blob + rec_off[r] + rec_len[r]);
    if (o->overlapping_consensus && (single_strand_allowed || has_both_strands_raw(recs))) {
      CorrectionStats cs;
      apply_overlapping_consensus(recs, cs);
      batch_overlap.overlapping_bases += cs.overlapping_bases; batch_overlap.bases_agreeing += cs.bases_agreeing;
      batch_overlap.bases_disagreeing += cs.bases_disagreeing; batch_overlap.bases_corrected += cs.bases_corrected;
    }
    std::vector<DuplexCaller::Rec> ptrs;
    for (auto& b : recs) ptrs.push_back({b.data(), b.size()});
    ConsensusOutput out = caller.consensus_reads(ptrs);
    res.data.insert(res.data.end(), out.data.begin(), out.data.end());
    res.count += out.count;
    batch_stats.merge(caller.stats);
    if (o->track_rejects) for (auto& rj : caller.rejected) append_reject(res, rj.data(), rj.size());
  }
  stats_to_array(batch_stats, batch_overlap, res.stats);
}
#endif

#ifdef ORC_WITH_CODEC
// process_fn of `fgumi codec` (src/lib/commands/codec.rs:722-790).  stats[24..28) carry the CODEC-only
// counters: consensus_bases_emitted, consensus_duplex_bases_emitted, duplex_disagreement_base_count,
// consensus_reads_rejected_hdd (codec_caller.rs:264-310).
static void codec_groups(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len,
                         const uint32_t* grp_first, uint32_t g0, uint32_t g1, OrcResult& res) {
  CodecOptions c;
  c.min_input_base_quality = o->min_input_base_quality; c.pre = o->error_rate_pre_umi; c.post = o->error_rate_post_umi;
  c.min_reads_per_strand = o->codec_min_reads_per_strand;
  c.has_max_reads = o->codec_max_reads_per_strand >= 0; c.max_reads_per_strand = c.has_max_reads ? (size_t)o->codec_max_reads_per_strand : 0;
  c.min_duplex_length = o->codec_min_duplex_length;
  c.has_ss_qual = o->codec_has_single_strand_qual; c.ss_qual = o->codec_single_strand_qual;
  c.has_outer_qual = o->codec_has_outer_bases_qual; c.outer_qual = o->codec_outer_bases_qual;
  c.outer_bases_length = o->codec_outer_bases_length;
  c.max_duplex_disagreements = o->codec_max_duplex_disagreements == 0xFFFFFFFFu ? UINT64_MAX : o->codec_max_duplex_disagreements;
  c.max_duplex_disagreement_rate = o->codec_max_duplex_disagreement_rate;
  c.has_cell_tag = o->cell_tag[0] != 0; c.cell_tag[0] = o->cell_tag[0]; c.cell_tag[1] = o->cell_tag[1];
  c.per_base_tags = o->produce_per_base_tags;
  c.tie_rule = o->tie_rule == FGX_TIE_ULP_RELATIVE ? TieRule::UlpRelative : TieRule::FgbioCompat;
  CodecCaller caller(o->read_name_prefix ? o->read_name_prefix : "", o->read_group_id ? o->read_group_id : "A", c, o->track_rejects != 0);
  CodecStats batch;
  for (uint32_t g = g0; g < g1; g++) {
    caller.clear();
    uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
    std::vector<CodecCaller::Rec> ptrs;
    for (uint32_t r = r0; r < r1; r++) ptrs.push_back({blob + rec_off[r], rec_len[r]});
    ConsensusOutput out;
    caller.consensus_reads(ptrs, out);   // a duplex-disagreement error is recoverable: stats and rejects are kept
    res.data.insert(res.data.end(), out.data.begin(), out.data.end());
    res.count += out.count;
    batch.merge(caller.stats);
    if (o->track_rejects) for (auto& rj : caller.rejected) append_reject(res, rj.data(), rj.size());
  }
  CorrectionStats none;
  stats_to_array(batch, none, res.stats);
  res.stats[24] += batch.consensus_bases_emitted; res.stats[25] += batch.duplex_bases_emitted;
  res.stats[26] += batch.disagreement_bases; res.stats[27] += batch.rejected_hdd;
}
#endif

typedef void (*groups_fn)(const fgx_options*, const uint8_t*, const uint64_t*, const uint32_t*, const uint32_t*, uint32_t,
                          uint32_t, OrcResult&);

static groups_fn pick(const fgx_options* o) {
  switch (o->caller_kind) {
    case FGX_CALLER_SIMPLEX: return simplex_groups;
#ifdef ORC_WITH_DUPLEX
    case FGX_CALLER_DUPLEX: return duplex_groups;
#endif
#ifdef ORC_WITH_CODEC
    case FGX_CALLER_CODEC: return codec_groups;
#endif
    default: return nullptr;
  }
}

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- methylation-aware mode (oracle_methylation.hpp) ------------------------------------------------------------------------
// `set_reference(reference, ref_names)`: contig i of the BAM header = seqs[i]; n_ref == 0 clears it.  Applies to the callers
// orc_process creates afterwards when the options carry a methylation mode.
void orc_set_reference(uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens) {
  if (n_ref == 0) { g_reference.reset(); return; }
  auto r = std::make_shared<Reference>();
  for (uint32_t i = 0; i < n_ref; i++) r->seqs.emplace_back(seqs[i], seqs[i] + lens[i]);
  g_reference = r;
}
// the single functions, for replaying the reference's unit tests
static SimpCigar simp_from(const uint32_t* ops, uint32_t n) { SimpCigar c; for (uint32_t i = 0; i < n; i++) c.push_back({(uint8_t)(ops[i] & 0xF), (size_t)(ops[i] >> 4)}); return c; }
// ops as BAM-encoded (len << 4 | code) simplified CIGAR ops; out[i] = ref position or INT64_MIN for None; returns the count
uint32_t orc_meth_query_to_ref_positions(const uint32_t* simplified, uint32_t n_s, int64_t alignment_start, int is_reverse, const uint32_t* original, uint32_t n_o,
                                         int64_t* out, uint32_t cap) {
  std::vector<int64_t> p = query_to_ref_positions(simp_from(simplified, n_s), alignment_start, is_reverse != 0, simp_from(original, n_o));
  for (size_t i = 0; i < p.size() && i < cap; i++) out[i] = p[i];
  return (uint32_t)p.size();
}
int orc_meth_is_cpg_context(const uint8_t* ref, uint64_t n, uint64_t pos, int top) { return is_cpg_context(ref, n, pos, top != 0) ? 1 : 0; }
int orc_meth_is_top_strand(uint16_t flg) { return is_top_strand(flg) ? 1 : 0; }
// reads: n_reads strings of read_lens[r] bases, concatenated; ref_bases[i] == 0 = None.  Fills is_ref_c / unconverted / converted[len].
void orc_meth_annotate(uint32_t len, const uint8_t* reads, const uint32_t* read_lens, uint32_t n_reads, const uint8_t* ref_bases, uint32_t n_ref_bases, int top,
                       uint8_t* is_ref_c, uint32_t* unconverted, uint32_t* converted) {
  std::vector<Bytes> rb(n_reads);
  size_t off = 0;
  for (uint32_t r = 0; r < n_reads; r++) { rb[r].assign(reads + off, reads + off + read_lens[r]); off += read_lens[r]; }
  std::vector<const Bytes*> ptr;
  for (auto& b : rb) ptr.push_back(&b);
  MethylationAnnotation a = annotate_simplex_methylation(len, ptr, Bytes(ref_bases, ref_bases + n_ref_bases), top != 0);
  for (uint32_t i = 0; i < len; i++) { is_ref_c[i] = a.evidence[i].is_ref_c; unconverted[i] = a.evidence[i].unconverted; converted[i] = a.evidence[i].converted; }
}
static MethylationAnnotation annot_from(uint32_t n, const uint8_t* is_ref_c, const uint32_t* unconverted, const uint32_t* converted) {
  MethylationAnnotation a;
  a.evidence.resize(n);
  for (uint32_t i = 0; i < n; i++) { a.evidence[i].is_ref_c = is_ref_c[i] != 0; a.evidence[i].unconverted = unconverted[i]; a.evidence[i].converted = converted[i]; }
  return a;
}
// returns the ML length (MM written NUL-terminated into mm), -1 for None, -2 when the reference would panic (length mismatch)
int orc_meth_build_mm_ml(const uint8_t* bases, uint32_t n_bases, uint32_t n_ev, const uint8_t* is_ref_c, const uint32_t* unconverted, const uint32_t* converted, int top, int mode,
                         char* mm, uint32_t mm_cap, uint8_t* ml, uint32_t ml_cap) {
  std::string s; Bytes m;
  try { if (!build_mm_ml_tags(Bytes(bases, bases + n_bases), annot_from(n_ev, is_ref_c, unconverted, converted), top != 0, mode, s, m)) return -1; }
  catch (const OracleError&) { return -2; }
  if (s.size() + 1 > mm_cap || m.size() > ml_cap) return -3;
  memcpy(mm, s.c_str(), s.size() + 1);
  memcpy(ml, m.data(), m.size());
  return (int)m.size();
}
void orc_meth_combine(uint32_t n_ab, const uint8_t* a_ref, const uint32_t* a_u, const uint32_t* a_t, uint32_t n_ba, const uint8_t* b_ref, const uint32_t* b_u, const uint32_t* b_t,
                      uint32_t len, uint8_t* o_ref, uint32_t* o_u, uint32_t* o_t) {
  MethylationAnnotation c = combine_methylation_annotations(annot_from(n_ab, a_ref, a_u, a_t), annot_from(n_ba, b_ref, b_u, b_t), len);
  for (uint32_t i = 0; i < len; i++) { o_ref[i] = c.evidence[i].is_ref_c; o_u[i] = c.evidence[i].unconverted; o_t[i] = c.evidence[i].converted; }
}

// MiGrouper::add_records with the consensus commands' record filter (src/lib/mi_group.rs:227-310;
// src/lib/commands/common.rs:384-397; crates/fgumi-umi/src/lib.rs:370-375).  Fills the kept records' offsets / lengths and
// grp_first[n_grp + 1]; returns n_grp, *n_kept = kept records.
uint32_t orc_group_records(const fgx_group_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                           uint64_t* out_off, uint32_t* out_len, uint32_t* grp_first, uint32_t* n_kept) {
  bool have = false;
  std::string current;
  uint32_t nk = 0, ng = 0;
  for (uint32_t r = 0; r < n_rec; r++) {
    RecView v(blob + rec_off[r], rec_len[r]);
    if (rec_len[r] < 32) continue;
    uint16_t f = v.flags();
    if ((f & flags::SECONDARY) || (f & flags::SUPPLEMENTARY)) continue;          // consensus_pregroup_keep_flags
    if (!o->allow_unmapped && (f & flags::UNMAPPED)) continue;
    Slice mi = find_string_tag(v.aux(), o->tag);
    if (!mi.some) continue;                                                      // records without the tag are skipped
    std::string key = mi.str();
    if (o->strip_strand_suffix) { size_t sl = key.rfind('/'); if (sl != std::string::npos && sl > 0) key.resize(sl); }   // extract_mi_base
    if (o->cell_tag[0]) { key.push_back('\t'); Slice cb = find_string_tag(v.aux(), o->cell_tag); if (cb.some) key += cb.str(); }
    if (!have || key != current) { grp_first[ng++] = nk; current = key; have = true; }
    out_off[nk] = rec_off[r]; out_len[nk] = rec_len[r]; nk++;
  }
  grp_first[ng] = nk;
  *n_kept = nk;
  return ng;
}

// ---- `fgumi filter` restatement (oracle_filter.hpp) ----------------------------------------------------------------
// Whole stream; the blob is copied first because masking is in place.  Returns a result handle or nullptr (orc_last_error).
struct OrcFilterResult { orc_filter::BatchResult r; Bytes blob; };
void* orc_filter_records(const fgx_filter_options* o, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec) {
  OrcFilterResult* res = new OrcFilterResult();
  res->blob.assign(blob, blob + blob_len);
  // --ref (fgx_filter_options.regenerate_alignment_tags): the reference set with orc_set_reference; a missing one = no contigs at all
  static const Reference no_contigs;
  const Reference* ref = o->regenerate_alignment_tags ? (g_reference ? g_reference.get() : &no_contigs) : nullptr;
  try { orc_filter::filter_stream(o, res->blob.data(), rec_off, rec_len, n_rec, res->r, ref); }
  catch (const OracleError& e) { g_err = e.what; delete res; return nullptr; }
  return res;
}
const uint8_t* orc_filter_data(void* r) { return ((OrcFilterResult*)r)->r.data.data(); }
uint64_t orc_filter_data_len(void* r) { return ((OrcFilterResult*)r)->r.data.size(); }
const uint8_t* orc_filter_rejects(void* r) { return ((OrcFilterResult*)r)->r.rejects.data(); }
uint64_t orc_filter_rejects_len(void* r) { return ((OrcFilterResult*)r)->r.rejects.size(); }
void orc_filter_counts(void* r, uint64_t* out4) {
  auto& b = ((OrcFilterResult*)r)->r;
  out4[0] = b.records_count; out4[1] = b.passed_count; out4[2] = b.bases_masked; out4[3] = b.rejected_count;
}
void orc_filter_free(void* r) { delete (OrcFilterResult*)r; }
// single functions, for replaying the reference's unit tests; thresholds as (min_reads, max_read_error_rate, max_base_error_rate)
static orc_filter::Thr thr_of(const double* t) { return orc_filter::Thr{(uint64_t)t[0], t[1], t[2]}; }
int64_t orc_filter_mask_bases(uint8_t* rec, uint32_t len, const double* thr, int has_minq, uint8_t minq) {
  return (int64_t)orc_filter::mask_bases(rec, len, thr_of(thr), has_minq != 0, minq);
}
int64_t orc_filter_mask_duplex_bases(uint8_t* rec, uint32_t len, const double* cc, const double* ab, const double* ba, int has_minq, uint8_t minq, int ss) {
  return (int64_t)orc_filter::mask_duplex_bases(rec, len, thr_of(cc), thr_of(ab), thr_of(ba), has_minq != 0, minq, ss != 0);
}
int orc_filter_read(const uint8_t* rec, uint32_t len, const double* thr) {   // 0 pass, 1 insufficient reads, 2 excessive error rate, -1 error
  try { return (int)orc_filter::filter_read(RecView(rec, len).aux(), thr_of(thr)); } catch (const OracleError& e) { g_err = e.what; return -1; }
}
int orc_filter_duplex_read(const uint8_t* rec, uint32_t len, const double* cc, const double* ab, const double* ba) {
  try { return (int)orc_filter::filter_duplex_read(RecView(rec, len).aux(), thr_of(cc), thr_of(ab), thr_of(ba)); }
  catch (const OracleError& e) { g_err = e.what; return -1; }
}
void orc_filter_reverse_tags(uint8_t* rec, uint32_t len) { orc_filter::reverse_per_base_tags(rec, len); }   // tag_reversal.rs:27-67 alone
// regenerate_alignment_tags_raw alone (alignment_tags.rs:259-433) against the reference of orc_set_reference: the new record into out
// (cap bytes), *out_len its length.  Returns 1 regenerated, 0 tags removed, -1 error (orc_last_error), -2 cap too small.
int orc_regenerate_alignment_tags(const uint8_t* rec, uint32_t len, uint8_t* out, uint32_t cap, uint32_t* out_len) {
  static const Reference no_contigs;
  Bytes v(rec, rec + len);
  bool r;
  try { r = orc_filter::regenerate_alignment_tags_raw(v, g_reference ? *g_reference : no_contigs); } catch (const OracleError& e) { g_err = e.what; return -1; }
  *out_len = (uint32_t)v.size();
  if (v.size() > cap) return -2;
  memcpy(out, v.data(), v.size());
  return r ? 1 : 0;
}
// the methylation filters alone (filter.rs:925-1340), for replaying the reference's unit tests; the reference of orc_set_reference
int64_t orc_filter_mask_methylation_depth(uint8_t* rec, uint32_t len, int duplex, const uint32_t* thr3) {
  try {
    auto t = orc_filter::methylation_tags_from_record(rec, len);
    return (int64_t)(duplex ? orc_filter::mask_methylation_depth_duplex(rec, len, thr3, t) : orc_filter::mask_methylation_depth_simplex(rec, len, thr3[0], t));
  } catch (const OracleError& e) { g_err = e.what; return -1; }
}
// out[i] = upper-cased reference base of query position i or -1; returns 1 (a map; *n = l_seq) or 0 (None)
int orc_filter_resolve_ref_bases(const uint8_t* rec, uint32_t len, int16_t* out, uint32_t cap, uint32_t* n) {
  static const Reference no_contigs;
  std::vector<int16_t> m;
  if (!orc_filter::resolve_ref_bases(rec, len, g_reference ? *g_reference : no_contigs, m)) { *n = 0; return 0; }
  *n = (uint32_t)m.size();
  for (size_t i = 0; i < m.size() && i < cap; i++) out[i] = m[i];
  return 1;
}
int64_t orc_filter_mask_strand_methylation_agreement(uint8_t* rec, uint32_t len) {
  static const Reference no_contigs;
  try {
    auto t = orc_filter::methylation_tags_from_record(rec, len);
    std::vector<int16_t> m;
    bool has = orc_filter::resolve_ref_bases(rec, len, g_reference ? *g_reference : no_contigs, m);
    return (int64_t)orc_filter::mask_strand_methylation_agreement(rec, len, has ? &m : nullptr, t);
  } catch (const OracleError& e) { g_err = e.what; return -1; }
}
int orc_filter_check_conversion_fraction(const uint8_t* rec, uint32_t len, double min_fraction, int mode) {
  static const Reference no_contigs;
  auto t = orc_filter::methylation_tags_from_record(rec, len);
  std::vector<int16_t> m;
  bool has = orc_filter::resolve_ref_bases(rec, len, g_reference ? *g_reference : no_contigs, m);
  return orc_filter::check_conversion_fraction(rec, len, min_fraction, has ? &m : nullptr, t, mode) ? 1 : 0;
}
int orc_filter_is_duplex(const uint8_t* rec, uint32_t len) { return orc_filter::is_duplex_consensus(RecView(rec, len).aux()); }
int orc_filter_process_record(const fgx_filter_options* o, uint8_t* rec, uint32_t len, uint64_t* masked, int* pass) {
  bool p = false;
  try { orc_filter::process_record_raw(rec, len, o, *masked, p); } catch (const OracleError& e) { g_err = e.what; return -1; }
  *pass = p;
  return 0;
}

// Whole input, mirroring `--threads T`: batches of `batch_groups` MI groups (50 simplex / 100
// duplex / 1000 codec in the reference), one caller object per batch, batches pulled by T worker
// threads, output concatenated in input order.  threads <= 1 runs inline.
void* orc_process(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len,
                  uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp, uint32_t batch_groups, uint32_t threads) {
  (void)n_rec;
  groups_fn fn = pick(o);
  if (!fn) { g_err = "unsupported caller kind in this oracle build"; return nullptr; }
  if (batch_groups == 0) batch_groups = 50;
  uint32_t n_batches = (n_grp + batch_groups - 1) / batch_groups;
  std::vector<OrcResult> parts(n_batches);
  std::atomic<uint32_t> next(0);
  std::atomic<bool> failed(false);
  std::string err;
  auto worker = [&]() {
    for (;;) {
      uint32_t b = next.fetch_add(1);
      if (b >= n_batches || failed.load()) return;
      uint32_t g0 = b * batch_groups, g1 = std::min(n_grp, g0 + batch_groups);
      try { fn(o, blob, rec_off, rec_len, grp_first, g0, g1, parts[b]); }
      catch (const OracleError& e) { if (!failed.exchange(true)) err = e.what; return; }
    }
  };
  const auto t_begin = std::chrono::steady_clock::now();
  if (threads <= 1) worker();
  else {
    std::vector<std::thread> ts;
    for (uint32_t t = 0; t < threads; t++) ts.emplace_back(worker);
    for (auto& t : ts) t.join();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  if (failed.load()) { g_err = err; return nullptr; }
  OrcResult* res = new OrcResult();
  res->seconds_workers = secs;
  size_t total = 0, rtotal = 0;
  for (auto& p : parts) { total += p.data.size(); rtotal += p.rejects.size(); }
  res->data.reserve(total);
  res->rejects.reserve(rtotal);
  for (auto& p : parts) {
    res->data.insert(res->data.end(), p.data.begin(), p.data.end());
    res->rejects.insert(res->rejects.end(), p.rejects.begin(), p.rejects.end());
    res->count += p.count;
    res->n_rejects += p.n_rejects;
    for (int i = 0; i < FGX_STATS_LEN; i++) res->stats[i] += p.stats[i];
  }
  return res;
}
const uint8_t* orc_result_data(void* r) { return ((OrcResult*)r)->data.data(); }
uint64_t orc_result_len(void* r) { return ((OrcResult*)r)->data.size(); }
uint64_t orc_result_count(void* r) { return ((OrcResult*)r)->count; }
void orc_result_stats(void* r, uint64_t* out) { memcpy(out, ((OrcResult*)r)->stats, sizeof(uint64_t) * FGX_STATS_LEN); }
const uint8_t* orc_result_rejects(void* r) { return ((OrcResult*)r)->rejects.data(); }
uint64_t orc_result_rejects_len(void* r) { return ((OrcResult*)r)->rejects.size(); }
uint64_t orc_result_n_rejects(void* r) { return ((OrcResult*)r)->n_rejects; }
// the worker section of orc_process alone: what the reference's Process step does (each batch's ConsensusOutput handed to the
// writer as it is); joining the batches into ONE buffer afterwards is this harness's convenience, single-threaded
double orc_result_seconds(void* r) { return ((OrcResult*)r)->seconds_workers; }
void orc_result_free(void* r) { delete (OrcResult*)r; }

// ---- scalar / column level, for the known-answer tests ---------------------------------------
double orc_phred_to_ln_error_prob(uint8_t q) { return phred_to_ln_error_prob(q); }
double orc_phred_to_ln_correct_prob(uint8_t q) { return phred_to_ln_correct_prob(q); }
uint8_t orc_ln_prob_to_phred(double x) { return ln_prob_to_phred(x); }
double orc_log1pexp(double x) { return log1pexp(x); }
double orc_ln_sum_exp(double a, double b) { return ln_sum_exp(a, b); }
double orc_ln_sum_exp_array(const double* v, uint32_t n) { return ln_sum_exp_array(v, n); }
double orc_ln_not(double x) { return ln_not(x); }
double orc_ln_error_prob_two_trials(double a, double b) { try { return ln_error_prob_two_trials(a, b); } catch (const OracleError&) { return NAN; } }
// returns 1 when the reference would panic (a < b by >= EPSILON)
int orc_ln_a_minus_b(double a, double b, double* out) { try { *out = ln_a_minus_b(a, b); return 0; } catch (const OracleError&) { return 1; } }
int orc_fgbio_unique_max_index(const double* ll) { return fgbio_unique_max_index(ll); }
int orc_unique_max_index(const double* ll) { return unique_max_index(ll); }
double orc_consensus_error(double gap) { return consensus_error(gap); }
uint8_t orc_unanimous_quality_from_gap(double gap, uint8_t pre) { return unanimous_quality_from_gap(gap, phred_to_ln_error_prob(pre)); }
double orc_unanimous_margin(double w, double l, double c) { return unanimous_margin(w, l, c); }

void* orc_builder_new(uint8_t pre, uint8_t post, int tie_rule) {
  auto* b = new ConsensusBaseBuilder(pre, post);
  b->tie_rule = tie_rule == FGX_TIE_ULP_RELATIVE ? TieRule::UlpRelative : TieRule::FgbioCompat;
  return b;
}
void orc_builder_free(void* b) { delete (ConsensusBaseBuilder*)b; }
void orc_builder_reset(void* b) { ((ConsensusBaseBuilder*)b)->reset(); }
void orc_builder_add(void* b, uint8_t base, uint8_t qual) { ((ConsensusBaseBuilder*)b)->add(base, qual); }
void orc_builder_add_n(void* b, uint8_t base, uint8_t qual, uint32_t n) { for (uint32_t i = 0; i < n; i++) ((ConsensusBaseBuilder*)b)->add(base, qual); }
void orc_builder_call(void* b, uint8_t* base, uint8_t* qual) { ((ConsensusBaseBuilder*)b)->call(*base, *qual); }
void orc_builder_call_full(void* b, uint8_t* base, uint8_t* qual) { ((ConsensusBaseBuilder*)b)->call_full(*base, *qual); }
int orc_builder_fast_path(void* b, uint8_t* base, uint8_t* qual) { return ((ConsensusBaseBuilder*)b)->try_unanimous_fast_path(*base, *qual) ? 1 : 0; }
uint32_t orc_builder_contributions(void* b) { return ((ConsensusBaseBuilder*)b)->contributions(); }
uint32_t orc_builder_observations_for_base(void* b, uint8_t base) { return ((ConsensusBaseBuilder*)b)->observations_for_base(base); }
void orc_builder_likelihoods(void* b, double* out4) { memcpy(out4, ((ConsensusBaseBuilder*)b)->likelihoods, 32); }
void orc_builder_set_likelihoods(void* b, const double* ll, const uint32_t* obs) {
  auto* B = (ConsensusBaseBuilder*)b;
  for (int i = 0; i < 4; i++) { B->likelihoods[i] = ll[i]; B->compensations[i] = 0.0; B->observations[i] = obs[i]; }
}
// which: 0 correct, 1 error_per_alt, 2 thresholds, 3 cerr_min
void orc_builder_table(void* b, int which, double* out94, uint32_t* cap) {
  auto* B = (ConsensusBaseBuilder*)b;
  const double* src = which == 0 ? B->adj.correct : which == 1 ? B->adj.error_per_alt : which == 2 ? B->gap.thresholds : B->gap.cerr_min;
  memcpy(out94, src, 94 * sizeof(double));
  if (cap) *cap = (uint32_t)B->gap.cap;
}
// Column batch with the same contract as fgx_call_columns.
void orc_call_columns(uint8_t pre, uint8_t post, int tie_rule, const uint8_t* bases, const uint8_t* quals, uint32_t n_cols,
                      uint32_t depth, uint8_t* out_base, uint8_t* out_qual, uint32_t* out_depth, uint32_t* out_errors) {
  ConsensusBaseBuilder b(pre, post);
  b.tie_rule = tie_rule == FGX_TIE_ULP_RELATIVE ? TieRule::UlpRelative : TieRule::FgbioCompat;
  for (uint32_t j = 0; j < n_cols; j++) {
    b.reset();
    for (uint32_t i = 0; i < depth; i++) { uint8_t base = bases[(size_t)j * depth + i]; if (base != NO_CALL_BASE) b.add(base, quals[(size_t)j * depth + i]); }
    uint8_t cb, cq;
    b.call(cb, cq);
    out_base[j] = cb; out_qual[j] = cq;
    out_depth[j] = b.contributions();
    out_errors[j] = b.contributions() - b.observations_for_base(cb);
  }
}
void orc_single_input_quals(const fgx_options* o, uint8_t* out94) {
  VanillaCaller c("", "A", vanilla_options_from(o));
  memcpy(out94, c.single_input_quals, 94);
}
int32_t orc_read_name_rank(const uint8_t* name, uint32_t n) { return fgbio_read_name_rank(name, n); }
uint64_t orc_mate_clip(const uint8_t* rec, uint32_t n) { return num_bases_extending_past_mate_raw(RecView(rec, n)); }
uint64_t orc_mate_clip_ops(int is_reverse, int32_t this_pos1, const uint32_t* this_ops, uint32_t n_this, int32_t mate_pos1,
                           const uint32_t* mate_ops, uint32_t n_mate) {
  return bases_extending_past_mate_ops(is_reverse != 0, this_pos1, std::vector<uint32_t>(this_ops, this_ops + n_this), mate_pos1,
                                       std::vector<uint32_t>(mate_ops, mate_ops + n_mate));
}
// returns 1 and (lead soft, ref len, trail soft) when the MC CIGAR parses, else 0
int orc_parse_mc(const char* s, int32_t* out3) {
  std::vector<uint32_t> ops;
  if (!parse_mc_cigar_ops((const uint8_t*)s, strlen(s), ops)) return 0;
  size_t ls = leading_soft_clip(ops), ts = trailing_soft_clip(ops);
  out3[0] = ls > (size_t)INT32_MAX ? INT32_MAX : (int32_t)ls;
  out3[1] = saturating_reference_length(ops);
  out3[2] = ts > (size_t)INT32_MAX ? INT32_MAX : (int32_t)ts;
  return 1;
}
uint32_t orc_quality_trim_point(const uint8_t* q, uint32_t n, uint8_t trim_qual) { return (uint32_t)VanillaCaller::find_quality_trim_point(Bytes(q, q + n), trim_qual); }
// consensus_umis over '\n'-separated UMIs; returns length written (or -1 on panic)
int orc_consensus_umis(const char* joined, char* out, uint32_t cap) {
  std::vector<std::string> umis;
  std::string cur;
  for (const char* p = joined;; p++) { if (*p == '\n' || *p == 0) { umis.push_back(cur); cur.clear(); if (*p == 0) break; } else cur.push_back(*p); }
  try { std::string r = consensus_umis(umis); if (r.size() + 1 > cap) return -1; memcpy(out, r.c_str(), r.size() + 1); return (int)r.size(); }
  catch (const OracleError&) { return -1; }
}
// Overlap pre-step on one pair (records modified in place). Returns 1 if call() returned true.
int orc_overlap_pair(uint8_t* r1, uint32_t n1, uint8_t* r2, uint32_t n2, uint64_t* stats4) {
  CorrectionStats cs;
  bool ok = overlapping_call(r1, n1, r2, n2, cs);
  stats4[0] = cs.overlapping_bases; stats4[1] = cs.bases_agreeing; stats4[2] = cs.bases_disagreeing; stats4[3] = cs.bases_corrected;
  return ok ? 1 : 0;
}

// apply_overlapping_consensus (overlapping.rs:627-684) over one group's records, modified in place (lengths do not change).
void orc_apply_overlapping(uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint64_t* stats4) {
  std::vector<Bytes> recs(n);
  for (uint32_t i = 0; i < n; i++) recs[i].assign(blob + rec_off[i], blob + rec_off[i] + rec_len[i]);
  CorrectionStats cs;
  apply_overlapping_consensus(recs, cs);
  for (uint32_t i = 0; i < n; i++) memcpy(blob + rec_off[i], recs[i].data(), rec_len[i]);
  stats4[0] = cs.overlapping_bases; stats4[1] = cs.bases_agreeing; stats4[2] = cs.bases_disagreeing; stats4[3] = cs.bases_corrected;
}

#ifdef ORC_WITH_CODEC
// clip_cigar_ops_raw (raw-bam/cigar.rs:404-446): the virtual hard clip of the CODEC caller.  Returns the number of ops written.
uint32_t orc_clip_cigar_ops(const uint32_t* ops, uint32_t n, uint32_t clip, int from_start, uint32_t* out, uint32_t cap, uint64_t* ref_consumed) {
  size_t rc = 0;
  std::vector<uint32_t> r = clip_cigar_ops_raw(std::vector<uint32_t>(ops, ops + n), clip, from_start != 0, rc);
  *ref_consumed = rc;
  for (size_t i = 0; i < r.size() && i < cap; i++) out[i] = r[i];
  return (uint32_t)r.size();
}
// read_pos_at_ref_pos_raw (raw-bam/cigar.rs:461-500): 1-based read position, 0 for None.
uint64_t orc_read_pos_at_ref_pos(const uint32_t* ops, uint32_t n, uint64_t aln_start, uint64_t ref_pos, int last_if_deleted) {
  size_t out = 0;
  return read_pos_at_ref_pos_raw(std::vector<uint32_t>(ops, ops + n), aln_start, ref_pos, last_if_deleted != 0, out) ? out : 0;
}
#endif

#ifdef ORC_WITH_DUPLEX
// duplex_consensus (duplex_caller.rs:931-1108) on two single-strand consensus reads given as plain arrays (n == 0: that strand is
// absent); no source reads (the approximate error recount).  Returns the duplex length, -1 when no duplex read comes out;
// flags: bit 0 = a BA consensus is attached, bit 1 = the lone strand was BA.
int orc_duplex_consensus(const uint8_t* ab_b, const uint8_t* ab_q, const uint16_t* ab_d, const uint16_t* ab_e, uint32_t n_ab,
                         const uint8_t* ba_b, const uint8_t* ba_q, const uint16_t* ba_d, const uint16_t* ba_e, uint32_t n_ba,
                         uint8_t* out_b, uint8_t* out_q, uint16_t* out_e, uint32_t cap, uint32_t* flags) {
  auto mk = [](const uint8_t* b, const uint8_t* q, const uint16_t* d, const uint16_t* e, uint32_t n) {
    VanillaConsensusRead v;
    v.id = "UMI123";
    v.bases.assign(b, b + n); v.quals.assign(q, q + n); v.depths.assign(d, d + n); v.errors.assign(e, e + n);
    return v;
  };
  VanillaConsensusRead a, b;
  if (n_ab) a = mk(ab_b, ab_q, ab_d, ab_e, n_ab);
  if (n_ba) b = mk(ba_b, ba_q, ba_d, ba_e, n_ba);
  DuplexConsensusRead out;
  if (!duplex_consensus(n_ab ? &a : nullptr, n_ba ? &b : nullptr, nullptr, out)) return -1;
  if (out.len() > cap) return -2;
  memcpy(out_b, out.bases.data(), out.len()); memcpy(out_q, out.quals.data(), out.len());
  for (size_t i = 0; i < out.len(); i++) out_e[i] = out.errors[i];
  *flags = (out.has_ba ? 1u : 0u) | (out.is_ba_only ? 2u : 0u);
  return (int)out.len();
}
uint8_t orc_duplex_cap_quality(int32_t s) { return cap_quality(s); }
#endif

// Replays the reference's fast-path ≡ call_full sweeps (base_builder.rs:1986-2012, 2042-2083,
// 2092-2123).  which: 0 broad (30 720 cases), 1 dense contiguous-depth, 2 deep cap region.
// Returns the number of mismatches; *n_cases = cases evaluated.
uint64_t orc_sweep_fast_vs_full(int which, uint64_t* n_cases) {
  uint64_t bad = 0, n = 0;
  const uint8_t bases[4] = {'A', 'C', 'G', 'T'};
  auto same = [&](ConsensusBaseBuilder& b) {
    uint8_t b1, q1, b2, q2;
    b.call(b1, q1); b.call_full(b2, q2);
    n++;
    if (b1 != b2 || q1 != q2) bad++;
  };
  if (which == 0) {
    const uint8_t pres[] = {0, 2, 20, 45, 69, 70, 90, 93}, posts[] = {0, 1, 2, 5, 10, 20, 40, 93};
    const uint32_t depths[] = {1, 2, 3, 5, 8, 12, 20, 35, 50, 100, 400, 1000};
    const uint8_t quals[] = {0, 1, 2, 5, 10, 20, 30, 40, 60, 93};
    for (uint8_t pre : pres) for (uint8_t post : posts) for (uint8_t base : bases) {
      ConsensusBaseBuilder b(pre, post);
      for (uint32_t d : depths) for (uint8_t q : quals) { b.reset(); for (uint32_t i = 0; i < d; i++) b.add(base, q); same(b); }
    }
  } else if (which == 1) {
    const uint8_t pres[] = {45, 70, 90};
    for (uint8_t pre : pres) for (uint8_t base : bases) for (int post = 0; post <= 12; post++) {
      ConsensusBaseBuilder b(pre, (uint8_t)post);
      double cap_thr = b.gap.thresholds[b.gap.cap];
      int idx = base_to_index(base);
      for (int obs = 0; obs <= 12; obs++) {
        b.reset();
        for (uint32_t d = 1; d <= 4000; d++) {
          b.add(base, (uint8_t)obs);
          same(b);
          double g = b.likelihoods[idx] - b.likelihoods[(idx + 1) % 4];
          if (std::isfinite(g) && g > cap_thr + 50.0) break;
        }
      }
    }
  } else {
    const int cfg[3][5] = {{90, 93, 93, 1450, 1560}, {93, 93, 40, 1550, 1660}, {90, 90, 40, 3150, 3260}};
    for (auto& c : cfg) {
      ConsensusBaseBuilder b((uint8_t)c[0], (uint8_t)c[1]);
      for (int i = 0; i < c[3] - 1; i++) b.add('A', (uint8_t)c[2]);
      for (int d = c[3]; d <= c[4]; d++) { b.add('A', (uint8_t)c[2]); same(b); }
    }
  }
  if (n_cases) *n_cases = n;
  return bad;
}

}  // extern "C"
