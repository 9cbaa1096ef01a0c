// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Restatement of the simplex caller and its pre-step:
//   crates/fgumi-consensus/src/vanilla_caller.rs:48-120, 292-353, 432-501, 706-779, 862-1062,
//       1080-1326, 1329-1909
//   crates/fgumi-consensus/src/caller.rs:172-213, 256-321, 401-446, 560-566, 665-674
//   crates/fgumi-consensus/src/simple_umi.rs:9-131, 236-245
//   crates/fgumi-consensus/src/overlapping.rs:14-17, 111-336, 382-684
//   src/lib/commands/simplex.rs:637-718 (process_fn: per-group min-reads skip + overlap pre-step)
// Methylation-aware mode (vanilla_caller.rs:715-724, 781-860, 1612-1625, 1853-1876) over oracle_methylation.hpp.
#pragma once
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include "oracle_bam.hpp"
#include "oracle_phred.hpp"
#include "oracle_methylation.hpp"

namespace orc {

// caller.rs:401-446 enum order
enum Rejection : int {
  FragmentRead = 0, InsufficientReads, QualityTooLow, Unmapped, Mapped, TooManyNs, MinorityAlignment,
  SecondaryOrSupplementary, FailedQC, MissingUmi, QualityTrimmed, ZeroLengthAfterTrimming, InsufficientOverlap,
  OrphanConsensus, IndelErrorBetweenStrands, ClipOverlapFailed, HighDuplexDisagreement, PotentialCollision,
  NotPrimaryFrPair, Downsampled, Other, N_REJECTION
};

struct Stats {  // caller.rs:256-321
  uint64_t total_reads = 0, consensus_reads = 0, filtered_reads = 0;
  uint64_t rejection[N_REJECTION] = {0};
  void record_rejection(Rejection r, size_t n) { filtered_reads += n; rejection[r] += n; }
  void record_input(size_t n) { total_reads += n; }
  void merge(const Stats& o) {
    total_reads += o.total_reads; consensus_reads += o.consensus_reads; filtered_reads += o.filtered_reads;
    for (int i = 0; i < N_REJECTION; i++) rejection[i] += o.rejection[i];
  }
};

struct CorrectionStats {  // overlapping.rs:51-60
  uint64_t overlapping_bases = 0, bases_agreeing = 0, bases_disagreeing = 0, bases_corrected = 0;
};

struct ConsensusOutput {  // caller.rs:172-177
  Bytes data;
  size_t count = 0;
  void extend(const ConsensusOutput& o) { data.insert(data.end(), o.data.begin(), o.data.end()); count += o.count; }
};

struct VanillaOptions {  // vanilla_caller.rs:292-353
  std::string tag = "MI";
  uint8_t error_rate_pre_umi = 45, error_rate_post_umi = 40, min_input_base_quality = 10;
  size_t min_reads = 2;
  bool has_max_reads = false;
  size_t max_reads = 0;
  bool produce_per_base_tags = true, trim = false;
  uint8_t min_consensus_base_quality = 40;
  bool has_cell_tag = false;
  char cell_tag[2] = {'C', 'B'};
  TieRule tie_rule = TieRule::FgbioCompat;
  int methylation_mode = MethDisabled;   // :323-326
};

struct SourceRead {  // vanilla_caller.rs:129-154
  size_t original_idx;
  Bytes bases, quals;
  SimpCigar simplified_cigar;
  uint16_t flags;
  int32_t name_hash;
  int32_t ref_id = -1;           // methylation annotation: reference sequence id, 0-based alignment start and the
  int64_t alignment_start = -1;  // simplified CIGAR before reversal / truncation (:1176-1190)
  SimpCigar original_cigar;
};

struct VanillaConsensusRead {
  std::string id;
  Bytes bases, quals;
  std::vector<uint16_t> depths, errors;
  std::vector<SourceRead> source_reads;
  bool has_methylation = false;
  MethylationAnnotation methylation;
  uint16_t max_depth() const { uint16_t m = 0; for (auto d : depths) m = std::max(m, d); return m; }
  uint16_t min_depth() const { if (depths.empty()) return 0; uint16_t m = 0xFFFF; for (auto d : depths) m = std::min(m, d); return m; }
};

// simple_umi.rs:236-245 + SimpleConsensusCaller::call_consensus :46-117
inline std::string consensus_umis(const std::vector<std::string>& umis) {
  if (umis.empty()) return "";
  if (umis.size() == 1) return umis[0];
  static thread_local std::unique_ptr<ConsensusBaseBuilder> b;
  if (!b) b.reset(new ConsensusBaseBuilder(90, 90));
  const std::string& first = umis[0];
  size_t L = first.size();
  for (auto& s : umis) if (s.size() != L) throw OracleError{"Sequences must all have the same length"};
  auto is_dna = [](uint8_t c) { uint8_t u = (c >= 'a' && c <= 'z') ? c - 32 : c; return u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N'; };
  std::string result;
  for (size_t i = 0; i < L; i++) {
    b->reset();
    size_t non_dna = 0;
    uint8_t fc = (uint8_t)first[i];
    for (auto& s : umis) {
      uint8_t c = (uint8_t)s[i];
      if (is_dna(c)) b->add(c, 20);
      else { non_dna++; if (fc != c) throw OracleError{"Sequences must have same non-DNA character"}; }
    }
    if (non_dna == 0) { uint8_t base, q; b->call(base, q); result.push_back((char)base); }
    else if (non_dna == umis.size()) result.push_back((char)fc);
    else throw OracleError{"mix of DNA and non-DNA characters"};
  }
  return result;
}

// select_lowest_ranking caller.rs:665-674
inline std::vector<size_t> select_lowest_ranking(const std::vector<int32_t>& ranks, size_t max_reads) {
  std::vector<size_t> idx(ranks.size());
  for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
  if (ranks.size() <= max_reads) return idx;
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ranks[a] < ranks[b]; });
  idx.resize(max_reads);
  std::sort(idx.begin(), idx.end());
  return idx;
}

// select_most_common_alignment_group vanilla_caller.rs:48-120
struct IndexedSR { size_t idx; size_t len; SimpCigar cigar; };
inline int cmp_cigar(const SimpCigar& a, const SimpCigar& b) {
  size_t n = std::min(a.size(), b.size());
  for (size_t i = 0; i < n; i++) {
    if (a[i].second != b[i].second) return a[i].second < b[i].second ? -1 : 1;
    if (a[i].first != b[i].first) return a[i].first < b[i].first ? -1 : 1;   // kind_ord == BAM op code
  }
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  return 0;
}
inline std::vector<size_t> select_most_common_alignment_group(const std::vector<IndexedSR>& indexed) {
  std::vector<size_t> out;
  if (indexed.size() < 2) { for (auto& e : indexed) out.push_back(e.idx); return out; }
  std::vector<std::pair<SimpCigar, std::vector<size_t>>> groups;
  for (auto& e : indexed) {
    bool found = false;
    for (auto& g : groups) if (is_cigar_prefix(e.cigar, g.first)) { g.second.push_back(e.idx); found = true; }
    if (!found) groups.push_back({e.cigar, {e.idx}});
  }
  // Iterator::max_by returns the LAST maximal element.
  size_t best = 0;
  for (size_t i = 1; i < groups.size(); i++) {
    auto& a = groups[best]; auto& b = groups[i];
    int c;  // compare(best, i)
    if (a.second.size() != b.second.size()) c = a.second.size() < b.second.size() ? -1 : 1;
    else c = cmp_cigar(b.first, a.first);
    if (c <= 0) best = i;
  }
  return groups[best].second;
}

enum ReadType { Fragment = 0, R1 = 1, R2 = 2 };

struct Positioned { size_t pos; const uint8_t* p; size_t n; };

class VanillaCaller {
 public:
  std::string read_name_prefix, read_group_id;
  VanillaOptions opt;
  Stats stats;
  ConsensusBaseBuilder builder;
  std::vector<Bytes> rejected_reads;
  bool track_rejects;
  uint8_t single_input_quals[94];
  std::shared_ptr<const Reference> reference;   // set_reference :512-522 (contig i of the header = reference->seqs[i])

  VanillaCaller(std::string prefix, std::string rg, VanillaOptions o, bool track = false)
      : read_name_prefix(std::move(prefix)), read_group_id(std::move(rg)), opt(std::move(o)),
        builder(opt.error_rate_pre_umi, opt.error_rate_post_umi), track_rejects(track) {
    builder.tie_rule = opt.tie_rule;
    // compute_single_input_consensus_quals :469-501
    uint8_t lab = std::min(opt.error_rate_pre_umi, opt.error_rate_post_umi);
    double ln_lab = phred_to_ln_error_prob(lab);
    for (int q = 0; q <= MAX_PHRED; q++) {
      double ln_seq = phred_to_ln_error_prob((uint8_t)q);
      uint8_t adj = ln_prob_to_phred(ln_error_prob_two_trials(ln_seq, ln_lab));
      single_input_quals[q] = std::min(adj, MAX_PHRED);
    }
  }
  void clear() { stats = Stats(); rejected_reads.clear(); }

  // find_quality_trim_point :992-1016
  static size_t find_quality_trim_point(const Bytes& quals, uint8_t trim_qual) {
    size_t length = quals.size();
    if (trim_qual < 1 || length == 0) return 0;
    int32_t score = 0, max_score = 0;
    size_t trim_point = length;
    for (size_t i = length; i-- > 0;) {
      score += (int32_t)trim_qual - (int32_t)quals[i];
      if (score < 0) break;
      if (score > max_score) { max_score = score; trim_point = i; }
    }
    return trim_point;
  }
  // truncate_simplified_cigar :1028-1062
  static SimpCigar truncate_simplified_cigar(const SimpCigar& c, size_t query_length) {
    SimpCigar r;
    size_t remaining = query_length;
    for (auto& op : c) {
      if (remaining == 0) break;
      bool cq = (op.first == 0 || op.first == 1 || op.first == 4 || op.first == 7 || op.first == 8);
      if (cq) { size_t take = std::min(op.second, remaining); r.push_back({op.first, take}); remaining -= take; }
      else r.push_back(op);
    }
    return r;
  }
  // create_source_read :1080-1190 ; returns false for Ok(None)
  bool create_source_read(const uint8_t* raw, size_t n, size_t original_idx, size_t mate_clip, SourceRead& out,
                          bool mask = true, bool strip_n = true) const {
    RecView v(raw, n);
    uint16_t flg = v.flags();
    bool neg = flg & flags::REVERSE;
    uint8_t min_bq = opt.min_input_base_quality;
    Bytes bases = v.sequence_vec();
    Bytes quals = v.quality_vec();
    size_t read_len = bases.size();
    if (read_len == 0) return false;
    if (quals.empty() || quals.size() != read_len) throw OracleError{"input read has invalid base qualities"};
    bool all_ff = true;
    for (auto q : quals) if (q != 0xFF) { all_ff = false; break; }
    if (all_ff) throw OracleError{"input read is missing base qualities"};
    if (neg) {
      Bytes rc(read_len);
      for (size_t i = 0; i < read_len; i++) rc[i] = complement_base(bases[read_len - 1 - i]);
      bases.swap(rc);
      std::reverse(quals.begin(), quals.end());
    }
    size_t trim_to = opt.trim ? find_quality_trim_point(quals, min_bq) : read_len;
    if (mask)
      for (size_t i = 0; i < trim_to; i++) if (quals[i] < min_bq) { bases[i] = NO_CALL_BASE; quals[i] = MIN_PHRED; }
    size_t clip_position = read_len > mate_clip ? read_len - mate_clip : 0;
    size_t final_len = std::min(clip_position, trim_to);
    if (strip_n) while (final_len > 0 && bases[final_len - 1] == NO_CALL_BASE) final_len--;
    if (final_len == 0) return false;
    bases.resize(final_len);
    quals.resize(final_len);
    SimpCigar orig = simplify_cigar_from_raw(v.cigar_ops());
    SimpCigar simp = orig;
    if (neg) std::reverse(simp.begin(), simp.end());
    simp = truncate_simplified_cigar(simp, final_len);
    out.original_idx = original_idx;
    out.bases.swap(bases);
    out.quals.swap(quals);
    out.simplified_cigar.swap(simp);
    out.flags = flg;
    out.ref_id = v.ref_id();
    out.alignment_start = (int64_t)v.pos();
    out.original_cigar.swap(orig);
    Slice nm = v.read_name();
    out.name_hash = opt.has_max_reads ? fgbio_read_name_rank(nm.p, nm.n) : 0;
    return true;
  }

  // drop_unmapped_if_any_mapped :1217-1232 + filter_source_reads_by_alignment :1242-1296
  std::vector<SourceRead> filter_source_reads_by_alignment(std::vector<SourceRead> srs, std::vector<size_t>& rejected_orig) {
    bool any_unmapped = false, all_unmapped = true;
    for (auto& s : srs) { bool u = s.flags & flags::UNMAPPED; any_unmapped |= u; all_unmapped &= u; }
    if (any_unmapped && !all_unmapped) {
      std::vector<SourceRead> kept;
      size_t dropped = 0;
      for (auto& s : srs) { if (s.flags & flags::UNMAPPED) { rejected_orig.push_back(s.original_idx); dropped++; } else kept.push_back(std::move(s)); }
      stats.record_rejection(Unmapped, dropped);
      srs.swap(kept);
    }
    if (srs.size() < 2) return srs;
    std::vector<IndexedSR> indexed;
    for (size_t i = 0; i < srs.size(); i++) indexed.push_back({i, srs[i].bases.size(), srs[i].simplified_cigar});
    std::stable_sort(indexed.begin(), indexed.end(), [](const IndexedSR& a, const IndexedSR& b) { return a.len > b.len; });
    std::vector<size_t> keep = select_most_common_alignment_group(indexed);
    std::vector<bool> mask(srs.size(), false);
    for (size_t k : keep) mask[k] = true;
    size_t rejected_count = srs.size() > keep.size() ? srs.size() - keep.size() : 0;
    for (size_t i = 0; i < srs.size(); i++) if (!mask[i]) rejected_orig.push_back(srs[i].original_idx);
    if (rejected_count > 0) stats.record_rejection(MinorityAlignment, rejected_count);
    std::vector<SourceRead> filtered;
    for (size_t i = 0; i < srs.size(); i++) if (mask[i]) filtered.push_back(std::move(srs[i]));
    return filtered;
  }

  // downsample_source_reads :970-980 (clone-based; used by consensus_call)
  std::vector<SourceRead> downsample_source_reads(const std::vector<SourceRead>& srs) const {
    if (opt.has_max_reads && srs.size() > opt.max_reads) {
      std::vector<size_t> idx(srs.size());
      for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
      std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return srs[a].name_hash < srs[b].name_hash; });
      idx.resize(opt.max_reads);
      std::vector<SourceRead> out;
      for (size_t i : idx) out.push_back(srs[i]);
      return out;
    }
    return srs;
  }

  // create_consensus_from_source_reads :1652-1755
  void create_consensus_from_source_reads(const std::vector<SourceRead>& srs, Bytes& cb, Bytes& cq,
                                          std::vector<uint16_t>& depths, std::vector<uint16_t>& errors) {
    if (srs.empty()) throw OracleError{"Cannot create consensus from empty source reads"};
    std::vector<size_t> lengths;
    for (auto& s : srs) lengths.push_back(s.bases.size());
    std::sort(lengths.begin(), lengths.end(), [](size_t a, size_t b) { return a > b; });
    size_t min_reads = opt.min_reads;
    size_t consensus_len = lengths[min_reads - 1];
    cb.clear(); cq.clear(); depths.clear(); errors.clear();
    if (srs.size() == 1) {
      const SourceRead& sr = srs[0];
      for (size_t pos = 0; pos < consensus_len; pos++) {
        uint8_t raw_base = sr.bases[pos];
        size_t qi = sr.quals[pos];
        uint8_t adj = qi < 94 ? single_input_quals[qi] : 0;
        if (adj < opt.min_consensus_base_quality) { cb.push_back(NO_CALL_BASE); cq.push_back(MIN_PHRED); }
        else { cb.push_back(raw_base); cq.push_back(adj); }
        depths.push_back(raw_base != NO_CALL_BASE ? 1 : 0);
        errors.push_back(0);
      }
      return;
    }
    for (size_t pos = 0; pos < consensus_len; pos++) {
      builder.reset();
      for (auto& sr : srs) {
        if (pos < sr.bases.size()) {
          uint8_t base = sr.bases[pos];
          if (base != NO_CALL_BASE) builder.add(base, sr.quals[pos]);
        }
      }
      uint8_t base, qual;
      builder.call(base, qual);
      uint32_t depth = builder.contributions();
      const uint32_t max_short = 32767;
      depths.push_back((uint16_t)std::min<uint32_t>(std::min<uint32_t>(depth, 0xFFFF), max_short));
      uint32_t err = depth - builder.observations_for_base(base);
      errors.push_back((uint16_t)std::min<uint32_t>(std::min<uint32_t>(err, 0xFFFF), max_short));
      if ((size_t)depth < min_reads) { cb.push_back(NO_CALL_BASE); cq.push_back(0); }
      else if (qual < opt.min_consensus_base_quality) { cb.push_back(NO_CALL_BASE); cq.push_back(MIN_PHRED); }
      else { cb.push_back(base); cq.push_back(qual); }
    }
  }

  // annotate_and_normalize :781-860.  Returns false for a None annotation (no reference, unplaced anchor, ref_id outside the header).
  bool annotate_and_normalize(std::vector<SourceRead>& srs, MethylationAnnotation& annot) const {
    if (!reference) return false;
    if (srs.empty()) return false;
    size_t anchor_idx = 0;   // Iterator::max_by_key returns the LAST maximal element
    for (size_t i = 1; i < srs.size(); i++) if (srs[i].bases.size() >= srs[anchor_idx].bases.size()) anchor_idx = i;
    const SourceRead& anchor = srs[anchor_idx];
    if (anchor.ref_id < 0 || anchor.alignment_start < 0) return false;
    if ((size_t)anchor.ref_id >= reference->seqs.size()) return false;
    const bool top = is_top_strand(anchor.flags);
    std::vector<int64_t> ref_positions = query_to_ref_positions(anchor.simplified_cigar, anchor.alignment_start, (anchor.flags & flags::REVERSE) != 0, anchor.original_cigar);
    Bytes ref_bases = fetch_ref_bases_at_positions(ref_positions, reference->seqs[(size_t)anchor.ref_id]);
    std::vector<const Bytes*> rb;
    for (auto& s : srs) rb.push_back(&s.bases);
    annot = annotate_simplex_methylation(anchor.bases.size(), rb, ref_bases, top);
    const uint8_t unconv = top ? 'C' : 'G', conv = top ? 'T' : 'A';
    for (auto& s : srs)
      for (size_t i = 0; i < annot.evidence.size(); i++)
        if (annot.evidence[i].is_ref_c && i < s.bases.size() && upper(s.bases[i]) == conv) s.bases[i] = unconv;
    return true;
  }

  // consensus_call :706-779 (used by duplex / codec). Returns false for None.
  bool consensus_call(const std::string& umi, std::vector<SourceRead> srs, VanillaConsensusRead& out) {
    if (srs.empty() || srs.size() < opt.min_reads) return false;
    MethylationAnnotation annot;
    bool has_annot = opt.methylation_mode != MethDisabled && annotate_and_normalize(srs, annot);
    std::vector<SourceRead> capped;
    const std::vector<SourceRead>* use = &srs;
    if (opt.has_max_reads && srs.size() > opt.max_reads) { capped = downsample_source_reads(srs); use = &capped; }
    if (use->size() < opt.min_reads) return false;
    out.id = umi;
    create_consensus_from_source_reads(*use, out.bases, out.quals, out.depths, out.errors);
    out.has_methylation = has_annot;
    out.methylation = has_annot ? annot.truncate(out.bases.size()) : MethylationAnnotation();
    out.source_reads = std::move(srs);
    return true;
  }

  // build_consensus_record_into :1767-1881
  void build_consensus_record_into(ConsensusOutput& output, const std::string& umi, ReadType rt,
                                   const std::vector<RecView>& original_raws, const Bytes& bases, const Bytes& quals,
                                   const std::vector<uint16_t>& depths, const std::vector<uint16_t>& errors,
                                   const MethylationAnnotation* methylation = nullptr) {
    std::string name = read_name_prefix + ":" + umi;  // write_consensus_read_name caller.rs:560-566
    uint16_t flag = flags::UNMAPPED;
    if (rt == R1) flag |= flags::PAIRED | flags::FIRST_SEGMENT | flags::MATE_UNMAPPED;
    else if (rt == R2) flag |= flags::PAIRED | flags::LAST_SEGMENT | flags::MATE_UNMAPPED;
    Bytes rec;
    if (!build_unmapped_record(rec, (const uint8_t*)name.data(), name.size(), flag, bases.data(), quals.data(), bases.size()))
      throw OracleError{"could not write the consensus record: read name too long"};
    append_string_tag(rec, "RG", (const uint8_t*)read_group_id.data(), read_group_id.size());
    int32_t max_depth = 0, min_depth = 0;
    if (!depths.empty()) { max_depth = *std::max_element(depths.begin(), depths.end()); min_depth = *std::min_element(depths.begin(), depths.end()); }
    uint64_t total_errors = 0, total_depth = 0;
    for (auto e : errors) total_errors += e;
    for (auto d : depths) total_depth += d;
    float error_rate = total_depth > 0 ? (float)total_errors / (float)total_depth : 0.0f;
    append_int_tag(rec, "cD", max_depth);
    append_int_tag(rec, "cM", min_depth);
    append_float_tag(rec, "cE", error_rate);
    if (opt.produce_per_base_tags) {
      std::vector<int16_t> d(depths.size()), e(errors.size());
      for (size_t i = 0; i < depths.size(); i++) d[i] = (int16_t)std::min<uint16_t>(depths[i], 32767);
      for (size_t i = 0; i < errors.size(); i++) e[i] = (int16_t)std::min<uint16_t>(errors[i], 32767);
      append_i16_array_tag(rec, "cd", d.data(), d.size());
      append_i16_array_tag(rec, "ce", e.data(), e.size());
    }
    append_string_tag(rec, "MI", (const uint8_t*)umi.data(), umi.size());
    if (opt.has_cell_tag && !original_raws.empty()) {
      Slice cb = find_string_tag(original_raws[0].aux(), opt.cell_tag);
      if (cb.some) append_string_tag(rec, opt.cell_tag, cb.p, cb.n);
    }
    std::vector<std::string> umis;
    for (auto& r : original_raws) { Slice rx = find_string_tag(r.aux(), "RX"); if (rx.some) umis.push_back(rx.str()); }
    if (!umis.empty()) { std::string cu = consensus_umis(umis); append_string_tag(rec, "RX", (const uint8_t*)cu.data(), cu.size()); }
    if (methylation) {   // :1853-1876
      bool top = original_raws.empty() ? true : is_top_strand(original_raws[0].flags());
      std::string mm; Bytes ml;
      if (build_mm_ml_tags(bases, *methylation, top, opt.methylation_mode, mm, ml)) {
        append_string_tag(rec, "MM", (const uint8_t*)mm.data(), mm.size());
        append_u8_array_tag(rec, "ML", ml.data(), ml.size());
      }
      std::vector<int16_t> cu = methylation->unconverted_counts(), ct = methylation->converted_counts();
      append_i16_array_tag(rec, "cu", cu.data(), cu.size());
      append_i16_array_tag(rec, "ct", ct.data(), ct.size());
    }
    write_with_block_size(rec, output.data);
    output.count += 1;
  }

  using Rejects = std::vector<std::pair<size_t, Bytes>>;

  // process_subgroup :1454-1646 ; returns ok, sets surviving
  bool process_subgroup(ConsensusOutput& output, const std::string& umi, ReadType rt, const std::vector<Positioned>& group_reads,
                        Rejects& group_rejects, size_t& surviving_count, Rejects& surviving_reads) {
    surviving_count = 0;
    surviving_reads.clear();
    if (group_reads.empty()) return false;
    auto push_rej = [&](size_t idx) { if (track_rejects) group_rejects.push_back({group_reads[idx].pos, Bytes(group_reads[idx].p, group_reads[idx].p + group_reads[idx].n)}); };
    if (group_reads.size() < opt.min_reads) {
      stats.record_rejection(InsufficientReads, group_reads.size());
      for (size_t i = 0; i < group_reads.size(); i++) push_rej(i);
      return false;
    }
    std::vector<SourceRead> srs;
    std::vector<size_t> zero_len;
    for (size_t idx = 0; idx < group_reads.size(); idx++) {
      RecView v(group_reads[idx].p, group_reads[idx].n);
      size_t clip = num_bases_extending_past_mate_raw(v);
      SourceRead sr;
      if (create_source_read(group_reads[idx].p, group_reads[idx].n, idx, clip, sr)) srs.push_back(std::move(sr));
      else zero_len.push_back(idx);
    }
    if (!zero_len.empty()) {
      stats.record_rejection(ZeroLengthAfterTrimming, zero_len.size());
      for (size_t idx : zero_len) push_rej(idx);
    }
    if (srs.size() < opt.min_reads) {
      if (!srs.empty()) { stats.record_rejection(InsufficientReads, srs.size()); for (auto& s : srs) push_rej(s.original_idx); }
      return false;
    }
    std::vector<size_t> rejected_idx;
    std::vector<SourceRead> filtered = filter_source_reads_by_alignment(std::move(srs), rejected_idx);
    // (HashSet iteration order is unspecified in the reference; rejects are re-sorted by position later.)
    for (size_t idx : rejected_idx) push_rej(idx);
    if (filtered.size() < opt.min_reads) {
      if (!filtered.empty()) { stats.record_rejection(InsufficientReads, filtered.size()); for (auto& s : filtered) push_rej(s.original_idx); }
      return false;
    }
    // downsample_filtered_source_reads :902-932
    if (opt.has_max_reads && filtered.size() > opt.max_reads) {
      std::vector<int32_t> ranks;
      for (auto& s : filtered) ranks.push_back(s.name_hash);
      std::vector<size_t> keep = select_lowest_ranking(ranks, opt.max_reads);
      std::vector<bool> km(filtered.size(), false);
      for (size_t k : keep) km[k] = true;
      std::vector<SourceRead> kept;
      size_t ndrop = 0;
      for (size_t i = 0; i < filtered.size(); i++) {
        if (km[i]) kept.push_back(std::move(filtered[i]));
        else { ndrop++; push_rej(filtered[i].original_idx); }
      }
      if (ndrop) stats.record_rejection(Downsampled, ndrop);
      filtered.swap(kept);
    }
    if (filtered.size() < opt.min_reads) {
      if (!filtered.empty()) { stats.record_rejection(InsufficientReads, filtered.size()); for (auto& s : filtered) push_rej(s.original_idx); }
      return false;
    }
    surviving_count = filtered.size();
    if (track_rejects)
      for (auto& s : filtered) surviving_reads.push_back({group_reads[s.original_idx].pos, Bytes(group_reads[s.original_idx].p, group_reads[s.original_idx].p + group_reads[s.original_idx].n)});
    MethylationAnnotation annot;   // :1612-1625
    bool has_annot = opt.methylation_mode != MethDisabled && annotate_and_normalize(filtered, annot);
    Bytes cb, cq;
    std::vector<uint16_t> depths, errors;
    create_consensus_from_source_reads(filtered, cb, cq, depths, errors);
    if (has_annot) annot = annot.truncate(cb.size());
    std::vector<RecView> raws;
    for (auto& s : filtered) raws.push_back(RecView(group_reads[s.original_idx].p, group_reads[s.original_idx].n));
    build_consensus_record_into(output, umi, rt, raws, cb, cq, depths, errors, has_annot ? &annot : nullptr);
    return true;
  }

  void flush_group_rejects(Rejects& r) {
    if (!track_rejects) return;
    std::stable_sort(r.begin(), r.end(), [](const std::pair<size_t, Bytes>& a, const std::pair<size_t, Bytes>& b) { return a.first < b.first; });
    for (auto& e : r) rejected_reads.push_back(std::move(e.second));
  }

  // process_group :1329-1422
  ConsensusOutput process_group(const std::string& umi, const std::vector<std::pair<const uint8_t*, size_t>>& records) {
    stats.record_input(records.size());
    Rejects group_rejects;
    std::vector<Positioned> reads;
    size_t filtered_count = 0;
    for (size_t i = 0; i < records.size(); i++) {
      uint16_t f = RecView(records[i].first, records[i].second).flags();
      if ((f & flags::SECONDARY) == 0 && (f & flags::SUPPLEMENTARY) == 0) reads.push_back({i, records[i].first, records[i].second});
      else { filtered_count++; if (track_rejects) group_rejects.push_back({i, Bytes(records[i].first, records[i].first + records[i].second)}); }
    }
    if (filtered_count > 0) stats.record_rejection(SecondaryOrSupplementary, filtered_count);
    if (reads.empty()) { flush_group_rejects(group_rejects); return ConsensusOutput(); }
    if (reads.size() < opt.min_reads) {
      stats.record_rejection(InsufficientReads, reads.size());
      if (track_rejects) for (auto& r : reads) group_rejects.push_back({r.pos, Bytes(r.p, r.p + r.n)});
      flush_group_rejects(group_rejects);
      return ConsensusOutput();
    }
    std::vector<Positioned> frag, r1, r2;
    for (auto& r : reads) {
      uint16_t f = RecView(r.p, r.n).flags();
      if (!(f & flags::PAIRED)) frag.push_back(r);
      else if (f & flags::FIRST_SEGMENT) r1.push_back(r);
      else if (f & flags::LAST_SEGMENT) r2.push_back(r);
    }
    ConsensusOutput output;
    size_t sc; Rejects sr;
    if (process_subgroup(output, umi, Fragment, frag, group_rejects, sc, sr)) stats.consensus_reads += 1;
    ConsensusOutput r1r2;
    size_t r1c, r2c; Rejects r1s, r2s;
    bool r1_ok = process_subgroup(r1r2, umi, R1, r1, group_rejects, r1c, r1s);
    bool r2_ok = process_subgroup(r1r2, umi, R2, r2, group_rejects, r2c, r2s);
    if (r1_ok && r2_ok) { stats.consensus_reads += 2; output.extend(r1r2); }
    else if (r1_ok) { stats.record_rejection(OrphanConsensus, r1c); if (track_rejects) for (auto& e : r1s) group_rejects.push_back(std::move(e)); }
    else if (r2_ok) { stats.record_rejection(OrphanConsensus, r2c); if (track_rejects) for (auto& e : r2s) group_rejects.push_back(std::move(e)); }
    flush_group_rejects(group_rejects);
    return output;
  }

  // ConsensusCaller::consensus_reads :1885-1909
  ConsensusOutput consensus_reads(const std::vector<std::pair<const uint8_t*, size_t>>& records) {
    if (records.empty()) return ConsensusOutput();
    if (opt.tag.size() != 2) throw OracleError{"Tag must be exactly 2 characters"};
    RecView first(records[0].first, records[0].second);
    Slice tv = find_string_tag(first.aux(), opt.tag.c_str());
    if (!tv.some) throw OracleError{"Missing UMI tag"};
    return process_group(tv.str(), records);
  }
};

// ------------------------------------------------------------------------------------
// overlapping.rs — Consensus/Consensus strategies only (hard-wired by all three commands,
// src/lib/commands/simplex.rs:444-447, 658-661)
// ------------------------------------------------------------------------------------
inline bool is_no_call(uint8_t b) { return b == 'N' || b == 'n' || b == '.'; }

struct ReadAndRefPosIter {  // overlapping.rs:382-540
  int32_t cur_read_pos, cur_ref_pos;
  std::vector<std::pair<uint8_t, size_t>> ops;
  size_t element_index = 0, in_elem_offset = 0;
  int32_t start_ref_pos, end_ref_pos, start_read_pos, end_read_pos;

  ReadAndRefPosIter(const RecView& v, size_t rec_start, size_t rec_end, size_t mate_start, size_t mate_end) {
    int32_t rs = (int32_t)rec_start, re = (int32_t)rec_end;
    int32_t rec_len = (int32_t)v.l_seq();
    int32_t min_ref = std::max(rs, (int32_t)mate_start), max_ref = std::min(re, (int32_t)mate_end);
    for (uint32_t op : v.cigar_ops()) ops.push_back({(uint8_t)(op & 0xF), (size_t)(op >> 4)});
    start_read_pos = 1; end_read_pos = rec_len;
    start_ref_pos = std::max(rs, min_ref); end_ref_pos = std::min(re, max_ref);
    cur_read_pos = 1; cur_ref_pos = rs;
    skip_to_start();
  }
  size_t len_on_target() const { if (element_index >= ops.size()) return 0; auto t = ops[element_index].first; return (t == 0 || t == 2 || t == 3 || t == 7 || t == 8) ? ops[element_index].second : 0; }
  size_t len_on_query() const { if (element_index >= ops.size()) return 0; auto t = ops[element_index].first; return (t == 0 || t == 1 || t == 4 || t == 7 || t == 8) ? ops[element_index].second : 0; }
  bool is_alignment() const { if (element_index >= ops.size()) return false; auto t = ops[element_index].first; return t == 0 || t == 7 || t == 8; }
  void skip_to_start() {
    while (element_index < ops.size()) {
      int32_t cur_ref_end = cur_ref_pos + (int32_t)len_on_target() - 1;
      int32_t cur_read_end = cur_read_pos + (int32_t)len_on_query() - 1;
      if (cur_ref_end >= start_ref_pos && cur_read_end >= start_read_pos) break;
      cur_ref_pos += (int32_t)len_on_target();
      cur_read_pos += (int32_t)len_on_query();
      element_index++;
    }
    skip_non_aligned();
  }
  void skip_non_aligned() {
    in_elem_offset = 0;
    while (element_index < ops.size() && !is_alignment()) {
      cur_ref_pos += (int32_t)len_on_target();
      cur_read_pos += (int32_t)len_on_query();
      element_index++;
    }
    if (element_index < ops.size() && (cur_ref_pos < start_ref_pos || cur_read_pos < start_read_pos)) {
      int32_t off = std::max(start_ref_pos - cur_ref_pos, start_read_pos - cur_read_pos);
      in_elem_offset = (size_t)off;
      cur_ref_pos += off;
      cur_read_pos += off;
    }
  }
  bool next(size_t& read_offset, int32_t& ref_pos) {
    if (element_index < ops.size()) {
      if (in_elem_offset >= ops[element_index].second) { element_index++; skip_non_aligned(); }
    }
    if (element_index >= ops.size() || cur_read_pos > end_read_pos || cur_ref_pos > end_ref_pos) return false;
    read_offset = (size_t)(cur_read_pos - 1);
    ref_pos = cur_ref_pos;
    cur_read_pos++; cur_ref_pos++; in_elem_offset++;
    return true;
  }
};

// OverlappingBasesConsensusCaller::call :236-336 (Consensus / Consensus)
inline bool overlapping_call(uint8_t* r1, size_t n1, uint8_t* r2, size_t n2, CorrectionStats& st) {
  RecView v1(r1, n1), v2(r2, n2);
  if ((v1.flags() & flags::UNMAPPED) || (v2.flags() & flags::UNMAPPED)) return false;
  if (v1.ref_id() != v2.ref_id()) return false;
  auto astart = [](const RecView& v, size_t& out) { int32_t p = v.pos(); if (p < 0) return false; out = (size_t)(uint32_t)(p + 1); return true; };
  auto aend = [](const RecView& v, size_t& out) {
    int32_t p = v.pos(); if (p < 0) return false;
    int32_t rl = reference_length_from_raw_bam(v); if (rl == 0) return false;
    out = (size_t)(uint32_t)(p + rl); return true; };
  size_t s1, e1, s2, e2;
  if (!astart(v1, s1) || !aend(v1, e1) || !astart(v2, s2) || !aend(v2, e2)) return false;
  ReadAndRefPosIter it1(v1, s1, e1, s2, e2), it2(v2, s2, e2, s1, e1);
  std::vector<std::pair<size_t, size_t>> positions;
  {
    size_t o1, o2; int32_t p1, p2;
    bool h1 = it1.next(o1, p1), h2 = it2.next(o2, p2);
    while (h1 && h2) {
      if (p1 < p2) h1 = it1.next(o1, p1);
      else if (p1 > p2) h2 = it2.next(o2, p2);
      else { positions.push_back({o1, o2}); h1 = it1.next(o1, p1); h2 = it2.next(o2, p2); }
    }
  }
  if (positions.empty()) return false;
  Bytes seq1 = v1.sequence_vec(), seq2 = v2.sequence_vec();
  Bytes q1 = v1.quality_vec(), q2 = v2.quality_vec();
  bool modified = false;
  for (auto& pp : positions) {
    uint8_t b1 = seq1[pp.first], b2 = seq2[pp.second];
    if (is_no_call(b1) || is_no_call(b2)) continue;
    st.overlapping_bases++;
    uint8_t qa = q1[pp.first], qb = q2[pp.second];
    if (b1 == b2) {
      st.bases_agreeing++;
      uint8_t nq = (uint8_t)std::min<unsigned>((unsigned)qa + (unsigned)qb, 93);
      q1[pp.first] = nq; q2[pp.second] = nq;
      if (nq != qa || nq != qb) { st.bases_corrected++; modified = true; }
    } else {
      st.bases_disagreeing++;
      uint8_t cb, cq;
      if (qa == qb) { cb = NO_CALL_BASE; cq = MIN_PHRED; }
      else if (qa > qb) { cb = b1; cq = std::max<uint8_t>((uint8_t)(qa - qb), MIN_PHRED); }
      else { cb = b2; cq = std::max<uint8_t>((uint8_t)(qb - qa), MIN_PHRED); }
      seq1[pp.first] = cb; seq2[pp.second] = cb; q1[pp.first] = cq; q2[pp.second] = cq;
      st.bases_corrected += 2;
      modified = true;
    }
  }
  if (modified) {
    size_t so1 = v1.seq_offset(), so2 = v2.seq_offset();
    for (size_t i = 0; i < seq1.size(); i++) set_base(r1, so1, i, seq1[i]);
    for (size_t i = 0; i < seq2.size(); i++) set_base(r2, so2, i, seq2[i]);
    memcpy(r1 + v1.qual_offset(), q1.data(), q1.size());
    memcpy(r2 + v2.qual_offset(), q2.data(), q2.size());
  }
  return true;
}

// apply_overlapping_consensus :627-684.  Records are mutable copies owned by the caller.
inline void apply_overlapping_consensus(std::vector<Bytes>& records, CorrectionStats& st) {
  std::map<std::string, std::pair<long, long>> pairs;  // pairs are disjoint ⇒ iteration order is unobservable
  for (size_t idx = 0; idx < records.size(); idx++) {
    RecView v(records[idx].data(), records[idx].size());
    uint16_t f = v.flags();
    if (f & (flags::SECONDARY | flags::SUPPLEMENTARY)) continue;
    std::string name = v.read_name().str();
    if (f & flags::FIRST_SEGMENT) {
      auto it = pairs.find(name);
      if (it != pairs.end()) it->second.first = (long)idx; else pairs[name] = {(long)idx, -1};
    } else if (f & flags::LAST_SEGMENT) {
      auto it = pairs.find(name);
      if (it != pairs.end()) it->second.second = (long)idx; else pairs[name] = {-1, (long)idx};
    }
  }
  for (auto& kv : pairs) {
    long i1 = kv.second.first, i2 = kv.second.second;
    if (i1 >= 0 && i2 >= 0)
      overlapping_call(records[i1].data(), records[i1].size(), records[i2].data(), records[i2].size(), st);
  }
}

}  // namespace orc
