// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Restatement of the CODEC caller:
//   crates/fgumi-consensus/src/codec_caller.rs:100-310 (options, stats), 316-325 (capped_short_array),
//       455-620 (source-read conversion, strand helpers), 625-1004 (consensus_reads_raw), 1005-1128
//       (clipped info, per-strand cap), 1130-1270 (alignment filter, phase check, consensus length),
//       1272-1314 (pad_consensus), 1331-1512 (build_duplex_consensus_from_padded), 1526-1561
//       (mask_consensus_quals_query_based), 1563-1757 (record emission), 1759-1830 (reject mask)
//   crates/fgumi-raw-bam/src/cigar.rs:137-150, 404-500, 669-922 (virtual hard clip, read_pos_at_ref_pos)
//   crates/fgumi-raw-bam/src/overlap.rs:83-108, 223-230 (pair classification, clip vs the mate in hand)
//   src/lib/commands/codec.rs:722-790 (process_fn: duplex-disagreement errors are recoverable)
#pragma once
#include "oracle_duplex.hpp"

namespace orc {

inline bool consumes_read(uint32_t t) { return t == 0 || t == 1 || t == 7 || t == 8; }   // cigar.rs:75-77
inline uint32_t enc_op(uint32_t t, size_t len) { return ((uint32_t)len << 4) | t; }

// reference_length_from_cigar cigar.rs:137-150 (i32, wrapping)
inline int32_t reference_length_from_cigar(const std::vector<uint32_t>& ops) {
  uint32_t n = 0;
  for (uint32_t op : ops) if (consumes_ref(op & 0xF)) n += op >> 4;
  return (int32_t)n;
}

// upgrade_clipping_raw cigar.rs:669-745
inline std::vector<uint32_t> upgrade_clipping_raw(const std::vector<uint32_t>& ops, size_t clip, bool from_start) {
  size_t hard = 0, soft = 0, skip = 0, n = ops.size();
  std::vector<uint32_t> out;
  if (from_start) {
    while (skip < n && (ops[skip] & 0xF) == 5) { hard += ops[skip] >> 4; skip++; }
    while (skip < n && (ops[skip] & 0xF) == 4) { soft += ops[skip] >> 4; skip++; }
    size_t up = std::min(soft, clip > hard ? clip - hard : 0);
    out.push_back(enc_op(5, hard + up));
    if (soft - up > 0) out.push_back(enc_op(4, soft - up));
    out.insert(out.end(), ops.begin() + skip, ops.end());
  } else {
    while (skip < n && (ops[n - 1 - skip] & 0xF) == 5) { hard += ops[n - 1 - skip] >> 4; skip++; }
    while (skip < n && (ops[n - 1 - skip] & 0xF) == 4) { soft += ops[n - 1 - skip] >> 4; skip++; }
    size_t up = std::min(soft, clip > hard ? clip - hard : 0);
    out.assign(ops.begin(), ops.begin() + (n - skip));
    if (soft - up > 0) out.push_back(enc_op(4, soft - up));
    out.push_back(enc_op(5, hard + up));
  }
  return out;
}

// clip_cigar_start_raw cigar.rs:748-849
inline std::vector<uint32_t> clip_cigar_start_raw(const std::vector<uint32_t>& ops, size_t clip, size_t& ref_clipped) {
  size_t hard = 0, soft = 0, skip = 0, n = ops.size();
  while (skip < n && (ops[skip] & 0xF) == 5) { hard += ops[skip] >> 4; skip++; }
  while (skip < n && (ops[skip] & 0xF) == 4) { soft += ops[skip] >> 4; skip++; }
  size_t read_clipped = 0;
  ref_clipped = 0;
  std::vector<uint32_t> kept;
  size_t idx = skip;
  while (idx < n) {
    uint32_t t = ops[idx] & 0xF;
    size_t len = ops[idx] >> 4;
    if (read_clipped == clip && kept.empty() && t == 2) { ref_clipped += len; idx++; continue; }
    if (read_clipped >= clip) break;
    bool is_read = consumes_read(t), is_ref = consumes_ref(t);
    if (is_read && len > clip - read_clipped) {
      if (t == 1) read_clipped += len;
      else {
        size_t rem_clip = clip - read_clipped;
        read_clipped += rem_clip;
        if (is_ref) ref_clipped += rem_clip;
        kept.push_back(enc_op(t, len - rem_clip));
      }
    } else {
      if (is_read) read_clipped += len;
      if (is_ref) ref_clipped += len;
    }
    idx++;
  }
  kept.insert(kept.end(), ops.begin() + idx, ops.end());
  std::vector<uint32_t> out;
  out.push_back(enc_op(5, hard + soft + read_clipped));
  out.insert(out.end(), kept.begin(), kept.end());
  return out;
}

// clip_cigar_end_raw cigar.rs:852-922
inline std::vector<uint32_t> clip_cigar_end_raw(const std::vector<uint32_t>& ops, size_t clip) {
  size_t hard = 0, soft = 0, skip = 0, n = ops.size();
  while (skip < n && (ops[n - 1 - skip] & 0xF) == 5) { hard += ops[n - 1 - skip] >> 4; skip++; }
  while (skip < n && (ops[n - 1 - skip] & 0xF) == 4) { soft += ops[n - 1 - skip] >> 4; skip++; }
  size_t read_clipped = 0;
  std::vector<uint32_t> kept_rev;
  size_t idx = n - skip;
  while (idx > 0) {
    uint32_t t = ops[idx - 1] & 0xF;
    size_t len = ops[idx - 1] >> 4;
    if (read_clipped == clip && kept_rev.empty() && t == 2) { idx--; continue; }
    if (read_clipped >= clip) break;
    bool is_read = consumes_read(t);
    if (is_read && len > clip - read_clipped) {
      if (t == 1) read_clipped += len;
      else {
        size_t rem_clip = clip - read_clipped;
        read_clipped += rem_clip;
        kept_rev.push_back(enc_op(t, len - rem_clip));
      }
    } else if (is_read) read_clipped += len;
    idx--;
  }
  std::vector<uint32_t> out(ops.begin(), ops.begin() + idx);
  out.insert(out.end(), kept_rev.rbegin(), kept_rev.rend());
  out.push_back(enc_op(5, hard + soft + read_clipped));
  return out;
}

// clip_cigar_ops_raw cigar.rs:404-446
inline std::vector<uint32_t> clip_cigar_ops_raw(const std::vector<uint32_t>& ops, size_t clip, bool from_start, size_t& ref_consumed) {
  ref_consumed = 0;
  if (clip == 0 || ops.empty()) return ops;
  size_t existing = 0, n = ops.size();
  if (from_start) { for (size_t i = 0; i < n && ((ops[i] & 0xF) == 4 || (ops[i] & 0xF) == 5); i++) existing += ops[i] >> 4; }
  else { for (size_t i = n; i > 0 && ((ops[i - 1] & 0xF) == 4 || (ops[i - 1] & 0xF) == 5); i--) existing += ops[i - 1] >> 4; }
  if (clip <= existing) return upgrade_clipping_raw(ops, clip, from_start);
  if (from_start) return clip_cigar_start_raw(ops, clip - existing, ref_consumed);
  return clip_cigar_end_raw(ops, clip - existing);
}

// read_pos_at_ref_pos_raw cigar.rs:461-500 ; returns false for None
inline bool read_pos_at_ref_pos_raw(const std::vector<uint32_t>& ops, size_t aln_start, size_t ref_pos, bool last_if_deleted, size_t& out) {
  if (ref_pos < aln_start) return false;
  size_t ref_off = 0, q_off = 0;
  for (uint32_t op : ops) {
    uint32_t t = op & 0xF;
    size_t len = op >> 4;
    size_t s = aln_start + ref_off;
    if (consumes_ref(t)) {
      size_t e = s + len - 1;   // (len == 0 wraps exactly as usize does)
      if (ref_pos >= s && ref_pos <= e) {
        if (consumes_query(t)) { out = q_off + (ref_pos - s) + 1; return true; }
        if (last_if_deleted) { out = q_off > 0 ? q_off : 1; return true; }
        return false;
      }
      ref_off += len;
    }
    if (consumes_query(t)) q_off += len;
  }
  return false;
}

struct CodecOptions {   // codec_caller.rs:176-262
  uint8_t min_input_base_quality = 10, pre = 45, post = 40;
  size_t min_reads_per_strand = 1;
  bool has_max_reads = false; size_t max_reads_per_strand = 0;
  size_t min_duplex_length = 1;
  bool has_ss_qual = false; uint8_t ss_qual = 0;
  bool has_outer_qual = false; uint8_t outer_qual = 0;
  size_t outer_bases_length = 5;
  uint64_t max_duplex_disagreements = UINT64_MAX;
  double max_duplex_disagreement_rate = 1.0;
  bool has_cell_tag = false; char cell_tag[2] = {0, 0};
  bool per_base_tags = false;
  TieRule tie_rule = TieRule::FgbioCompat;
};

struct CodecStats : Stats {   // codec_caller.rs:264-310 (first three alias total_input_reads / generated / filtered)
  uint64_t consensus_bases_emitted = 0, duplex_bases_emitted = 0, disagreement_bases = 0, rejected_hdd = 0;
  void merge(const CodecStats& o) {
    Stats::merge(o);
    consensus_bases_emitted += o.consensus_bases_emitted; duplex_bases_emitted += o.duplex_bases_emitted;
    disagreement_bases += o.disagreement_bases; rejected_hdd += o.rejected_hdd;
  }
};

struct SingleStrand {   // SingleStrandConsensus :283-300 (fields with observable effect)
  Bytes bases, quals;
  std::vector<uint16_t> depths, errors;
};

struct ClippedInfo {   // :323-336
  size_t raw_idx, clip_amount;
  bool clip_from_start;
  size_t clipped_seq_len;
  std::vector<uint32_t> clipped_cigar;
  size_t adjusted_pos;
  uint16_t flags;
};

class CodecCaller {
 public:
  using Rec = std::pair<const uint8_t*, size_t>;
  std::string prefix, rg;
  CodecOptions o;
  CodecStats stats;
  uint64_t consensus_counter = 0;
  VanillaCaller ss;
  bool track;
  std::vector<Bytes> rejected;
  std::vector<bool> mask;

  static VanillaOptions ss_options(const CodecOptions& c) {   // :374-397
    VanillaOptions v;
    v.error_rate_pre_umi = c.pre; v.error_rate_post_umi = c.post; v.min_input_base_quality = c.min_input_base_quality;
    v.min_reads = 1; v.has_max_reads = false; v.produce_per_base_tags = true; v.trim = false; v.min_consensus_base_quality = 0;
    v.has_cell_tag = false; v.tie_rule = c.tie_rule;
    return v;
  }
  CodecCaller(std::string p, std::string r, CodecOptions c, bool track_rejects)
      : prefix(std::move(p)), rg(std::move(r)), o(c), ss("x", rg, ss_options(c)), track(track_rejects) {}
  void clear() { stats = CodecStats(); ss.clear(); rejected.clear(); mask.clear(); }   // :476-481 (counter survives)

  void reject_count(size_t n, Rejection r) { stats.record_rejection(r, n); }
  void reject_at(const std::vector<size_t>& idx, Rejection r) {   // :1767-1790
    for (size_t i : idx) if (i < mask.size()) mask[i] = true;
    reject_count(idx.size(), r);
  }
  static std::vector<size_t> strand_idx(const std::vector<ClippedInfo>& a, const std::vector<ClippedInfo>& b) {
    std::vector<size_t> v;
    for (auto& i : a) v.push_back(i.raw_idx);
    for (auto& i : b) v.push_back(i.raw_idx);
    return v;
  }

  static ClippedInfo build_clipped_info(const Rec& r, size_t raw_idx, size_t clip) {   // :1006-1040
    RecView v(r.first, r.second);
    ClippedInfo ci;
    ci.raw_idx = raw_idx; ci.clip_amount = clip; ci.flags = v.flags();
    ci.clip_from_start = ci.flags & flags::REVERSE;
    size_t ref_consumed = 0;
    ci.clipped_cigar = clip_cigar_ops_raw(v.cigar_ops(), clip, ci.clip_from_start, ref_consumed);
    size_t l = v.l_seq();
    ci.clipped_seq_len = l > clip ? l - clip : 0;
    size_t p1 = (size_t)(int64_t)(v.pos() + 1);
    ci.adjusted_pos = ci.clip_from_start ? p1 + ref_consumed : p1;
    return ci;
  }

  std::vector<ClippedInfo> filter_most_common(std::vector<ClippedInfo> infos) {   // :1130-1174
    if (infos.size() < 2) return infos;
    std::vector<IndexedSR> indexed;
    for (size_t i = 0; i < infos.size(); i++) {
      SimpCigar c = simplify_cigar_from_raw(infos[i].clipped_cigar);
      if (infos[i].flags & flags::REVERSE) std::reverse(c.begin(), c.end());
      indexed.push_back({i, infos[i].clipped_seq_len, std::move(c)});
    }
    std::stable_sort(indexed.begin(), indexed.end(), [](const IndexedSR& a, const IndexedSR& b) { return a.len > b.len; });
    std::vector<size_t> best = select_most_common_alignment_group(indexed);
    std::vector<bool> keep(infos.size(), false);
    for (size_t i : best) keep[i] = true;
    std::vector<size_t> rej;
    for (size_t i = 0; i < infos.size(); i++) if (!keep[i]) rej.push_back(infos[i].raw_idx);
    if (!rej.empty()) reject_at(rej, MinorityAlignment);
    std::vector<ClippedInfo> out;
    for (size_t i = 0; i < infos.size(); i++) if (keep[i]) out.push_back(std::move(infos[i]));
    return out;
  }

  static size_t cap_lowest_ranking(const std::vector<Rec>& recs, std::vector<ClippedInfo>& infos, size_t max_reads) {   // :1096-1113
    if (infos.size() <= max_reads) return 0;
    size_t dropped = infos.size() - max_reads;
    std::vector<int32_t> ranks;
    for (auto& i : infos) { Slice nm = RecView(recs[i.raw_idx].first, recs[i.raw_idx].second).read_name(); ranks.push_back(fgbio_read_name_rank(nm.p, nm.n)); }
    std::vector<size_t> keep = select_lowest_ranking(ranks, max_reads);
    std::vector<ClippedInfo> kept;
    for (size_t k : keep) kept.push_back(std::move(infos[k]));
    infos.swap(kept);
    return dropped;
  }

  static const ClippedInfo& longest(const std::vector<ClippedInfo>& v) {   // first maximum (rev().max_by_key)
    size_t best = 0;
    int32_t bl = reference_length_from_cigar(v[0].clipped_cigar);
    for (size_t i = 1; i < v.size(); i++) { int32_t l = reference_length_from_cigar(v[i].clipped_cigar); if (l > bl) { bl = l; best = i; } }
    return v[best];
  }

  static bool check_overlap_phase(const ClippedInfo& r1, const ClippedInfo& r2, size_t os, size_t oe) {   // :1176-1218
    auto at = [](const ClippedInfo& r, size_t p) -> int64_t { size_t q; return read_pos_at_ref_pos_raw(r.clipped_cigar, r.adjusted_pos, p, true, q) ? (int64_t)q : 0; };
    return (at(r1, os) - at(r2, os)) == (at(r1, oe) - at(r2, oe));
  }

  static SourceRead to_source_read(const Rec& r, size_t idx, const ClippedInfo& ci) {   // :503-570
    RecView v(r.first, r.second);
    SourceRead sr;
    sr.original_idx = idx;
    sr.bases = v.sequence_vec(); sr.quals = v.quality_vec(); sr.flags = v.flags();
    size_t clip = std::min(ci.clip_amount, sr.bases.size());
    if (clip > 0) {
      if (ci.clip_from_start) { sr.bases.erase(sr.bases.begin(), sr.bases.begin() + clip); sr.quals.erase(sr.quals.begin(), sr.quals.begin() + clip); }
      else { sr.bases.resize(sr.bases.size() - clip); sr.quals.resize(sr.quals.size() - clip); }
    }
    sr.simplified_cigar = simplify_cigar_from_raw(ci.clipped_cigar);
    if (sr.flags & flags::REVERSE) {
      std::reverse(sr.simplified_cigar.begin(), sr.simplified_cigar.end());
      std::reverse(sr.bases.begin(), sr.bases.end());
      for (auto& b : sr.bases) b = complement_base(b);
      std::reverse(sr.quals.begin(), sr.quals.end());
    }
    sr.name_hash = 0;
    return sr;
  }

  static SingleStrand rc(const SingleStrand& s) {   // :589-603
    SingleStrand o2;
    o2.bases.assign(s.bases.rbegin(), s.bases.rend());
    for (auto& b : o2.bases) b = complement_base(b);
    o2.quals.assign(s.quals.rbegin(), s.quals.rend());
    o2.depths.assign(s.depths.rbegin(), s.depths.rend());
    o2.errors.assign(s.errors.rbegin(), s.errors.rend());
    return o2;
  }
  static SingleStrand pad(const SingleStrand& s, size_t L, bool left) {   // :1272-1314
    size_t cur = s.bases.size();
    if (L <= cur) return s;
    size_t n = L - cur;
    SingleStrand o2;
    if (left) { o2.bases.assign(n, 'n'); o2.quals.assign(n, 0); o2.depths.assign(n, 0); o2.errors.assign(n, 0); }
    o2.bases.insert(o2.bases.end(), s.bases.begin(), s.bases.end()); o2.quals.insert(o2.quals.end(), s.quals.begin(), s.quals.end());
    o2.depths.insert(o2.depths.end(), s.depths.begin(), s.depths.end()); o2.errors.insert(o2.errors.end(), s.errors.begin(), s.errors.end());
    if (!left) { o2.bases.insert(o2.bases.end(), n, 'n'); o2.quals.insert(o2.quals.end(), n, 0); o2.depths.insert(o2.depths.end(), n, 0); o2.errors.insert(o2.errors.end(), n, 0); }
    return o2;
  }

  // build_duplex_consensus_from_padded :1331-1512 ; returns 0 ok / 1 count exceeded / 2 rate exceeded
  int build_duplex(const SingleStrand& a, const SingleStrand& b, SingleStrand& out, uint64_t& n_duplex, uint64_t& n_disagree) const {
    size_t len = a.bases.size();
    out.bases.assign(len, 'N'); out.quals.assign(len, MIN_PHRED); out.depths.assign(len, 0); out.errors.assign(len, 0);
    uint64_t dis = 0, dup = 0;
    for (size_t i = 0; i < len; i++) {
      uint8_t ba = a.bases[i], qa = a.quals[i], bb = b.bases[i], qb = b.quals[i];
      uint16_t da = a.depths[i], ea = a.errors[i], db = b.depths[i], eb = b.errors[i];
      bool ha = ba != 'N' && ba != 'n', hb = bb != 'N' && bb != 'n';
      uint8_t fb, fq;
      uint16_t depth, error;
      if (ha && hb) {
        dup++;
        uint8_t rb, rq;
        if (ba == bb) { rb = ba; rq = (uint8_t)std::min<uint16_t>(93, (uint16_t)qa + (uint16_t)qb); }
        else if (qa > qb) { dis++; rb = ba; rq = std::max<uint8_t>(MIN_PHRED, (uint8_t)(qa - qb)); }
        else if (qb > qa) { dis++; rb = bb; rq = std::max<uint8_t>(MIN_PHRED, (uint8_t)(qb - qa)); }
        else { dis++; rb = ba; rq = MIN_PHRED; }
        if (rq == MIN_PHRED) { fb = 'N'; fq = MIN_PHRED; } else { fb = rb; fq = rq; }
        int64_t de;
        if (ba == bb) de = (int64_t)ea + eb;
        else if (ba == rb) de = (int64_t)ea + (db > eb ? db - eb : 0);
        else de = (int64_t)eb + (da > ea ? da - ea : 0);
        error = clamp_combined_error(de);
        depth = (uint16_t)(clamp_per_base_short(da) + clamp_per_base_short(db));
      } else if (ha) {
        if (qa == MIN_PHRED) { fb = 'N'; fq = MIN_PHRED; } else { fb = ba; fq = qa; }
        depth = da; error = ea;
      } else if (hb) {
        if (qb == MIN_PHRED) { fb = 'N'; fq = MIN_PHRED; } else { fb = bb; fq = qb; }
        depth = db; error = eb;
      } else {
        fb = 'N'; fq = MIN_PHRED; depth = 0; error = clamp_combined_error((int64_t)ea + eb);
      }
      if (ba == 'N' || bb == 'N') { fb = 'N'; fq = MIN_PHRED; }
      out.bases[i] = fb; out.quals[i] = fq; out.depths[i] = depth; out.errors[i] = error;
    }
    n_duplex = 0; n_disagree = 0;
    if (dup > 0) {
      double rate = (double)dis / (double)dup;
      if (dis > o.max_duplex_disagreements) return 1;
      if (rate > o.max_duplex_disagreement_rate) return 2;
      n_duplex = dup; n_disagree = dis;
    }
    return 0;
  }

  void mask_quals(SingleStrand& c, const SingleStrand& p1, const SingleStrand& p2) const {   // :1526-1561
    size_t len = c.quals.size();
    if (o.outer_bases_length > 0 && o.has_outer_qual) {
      size_t last = len > 0 ? len - 1 : 0;
      for (size_t i = 0; i < std::min(o.outer_bases_length, len); i++) { c.quals[i] = o.outer_qual; c.quals[last - i] = o.outer_qual; }
    }
    if (o.has_ss_qual)
      for (size_t i = 0; i < len; i++) {
        uint8_t a = i < p1.bases.size() ? p1.bases[i] : 'N', b = i < p2.bases.size() ? p2.bases[i] : 'N';
        if (a == 'N' || a == 'n' || b == 'N' || b == 'n') c.quals[i] = o.ss_qual;
      }
  }

  static void strand_tags(Bytes& rec, const char* d, const char* m, const char* e, const std::vector<uint16_t>& depths, const std::vector<uint16_t>& errors) {
    int32_t mx = 0, mn = 0;
    uint64_t te = 0, tb = 0;
    bool first = true;
    for (auto dd : depths) { int32_t c = clamp_per_base_short(dd); if (first) { mx = mn = c; first = false; } mx = std::max(mx, c); mn = std::min(mn, c); tb += (uint64_t)c; }
    for (auto ee : errors) te += (uint64_t)clamp_per_base_short(ee);
    float rate = tb > 0 ? (float)te / (float)tb : 0.0f;
    append_int_tag(rec, d, mx); append_int_tag(rec, m, mn); append_float_tag(rec, e, rate);
  }

  // build_output_record_into :1590-1757
  void build_output(ConsensusOutput& out, const SingleStrand& c, const SingleStrand& a, const SingleStrand& b, bool has_umi,
                    const std::string& umi, const std::vector<Rec>& source_raws, const std::vector<Rec>& all) {
    consensus_counter++;
    std::string name = has_umi ? prefix + ":" + umi : prefix + ":" + std::to_string(consensus_counter);
    Bytes rec;
    if (!build_unmapped_record(rec, (const uint8_t*)name.data(), name.size(), flags::UNMAPPED, c.bases.data(), c.quals.data(), c.bases.size()))
      throw OracleError{"could not write the consensus record: read name too long"};
    append_string_tag(rec, "RG", (const uint8_t*)rg.data(), rg.size());
    if (has_umi) append_string_tag(rec, "MI", (const uint8_t*)umi.data(), umi.size());
    {
      int32_t mx = 0, mn = 0;
      uint64_t te = 0, tb = 0;
      size_t n = std::min(a.depths.size(), b.depths.size());
      for (size_t i = 0; i < n; i++) { int32_t t = clamp_per_base_short(a.depths[i]) + clamp_per_base_short(b.depths[i]); if (i == 0) mx = mn = t; mx = std::max(mx, t); mn = std::min(mn, t); tb += (uint64_t)t; }
      for (auto e : c.errors) te += (uint64_t)clamp_per_base_short(e);
      float rate = tb > 0 ? (float)te / (float)tb : 0.0f;
      append_int_tag(rec, "cD", mx); append_int_tag(rec, "cM", mn); append_float_tag(rec, "cE", rate);
    }
    strand_tags(rec, "aD", "aM", "aE", a.depths, a.errors);
    strand_tags(rec, "bD", "bM", "bE", b.depths, b.errors);
    if (o.per_base_tags) {
      auto capped = [](const std::vector<uint16_t>& v) { std::vector<int16_t> r(v.size()); for (size_t i = 0; i < v.size(); i++) r[i] = (int16_t)std::min<uint16_t>(v[i], 32767); return r; };
      auto ad = capped(a.depths), bd = capped(b.depths), ae = capped(a.errors), be = capped(b.errors);
      append_i16_array_tag(rec, "ad", ad.data(), ad.size()); append_i16_array_tag(rec, "bd", bd.data(), bd.size());
      append_i16_array_tag(rec, "ae", ae.data(), ae.size()); append_i16_array_tag(rec, "be", be.data(), be.size());
      append_string_tag(rec, "ac", a.bases.data(), a.bases.size()); append_string_tag(rec, "bc", b.bases.data(), b.bases.size());
      append_phred33_string_tag(rec, "aq", a.quals.data(), a.quals.size()); append_phred33_string_tag(rec, "bq", b.quals.data(), b.quals.size());
    }
    if (o.has_cell_tag)
      for (auto& r : source_raws) {
        Slice cb = find_string_tag(RecView(r.first, r.second).aux(), o.cell_tag);
        if (cb.some && cb.n > 0) { append_string_tag(rec, o.cell_tag, cb.p, cb.n); break; }
      }
    std::vector<std::string> umis;
    for (auto& r : all) { Slice rx = find_string_tag(RecView(r.first, r.second).aux(), "RX"); if (rx.some) umis.push_back(rx.str()); }
    if (!umis.empty()) { std::string cu = consensus_umis(umis); if (!cu.empty()) append_string_tag(rec, "RX", (const uint8_t*)cu.data(), cu.size()); }
    write_with_block_size(rec, out.data);
    out.count += 1;
  }

  // consensus_reads_raw :625-1004 ; returns false when a (recoverable) duplex-disagreement error is raised
  bool consensus_reads_raw(const std::vector<Rec>& recs, ConsensusOutput& out) {
    stats.total_reads += recs.size();
    mask.clear();
    if (track) mask.assign(recs.size(), false);
    if (recs.empty()) return true;
    Slice mi = find_string_tag(RecView(recs[0].first, recs[0].second).aux(), "MI");
    bool has_umi = mi.some;
    std::string umi = has_umi ? mi.str() : std::string();

    std::vector<size_t> paired, frags;
    for (size_t i = 0; i < recs.size(); i++) {
      uint16_t f = RecView(recs[i].first, recs[i].second).flags();
      if (!(f & flags::PAIRED)) { frags.push_back(i); continue; }
      if (f & (flags::SECONDARY | flags::SUPPLEMENTARY)) continue;
      paired.push_back(i);
    }
    if (!frags.empty()) reject_at(frags, FragmentRead);
    if (paired.empty()) return true;

    std::vector<std::string> order;
    std::unordered_map<std::string, std::vector<size_t>> by_name;
    for (size_t i : paired) {
      std::string nm = RecView(recs[i].first, recs[i].second).read_name().str();
      auto it = by_name.find(nm);
      if (it == by_name.end()) { order.push_back(nm); by_name[nm].push_back(i); } else it->second.push_back(i);
    }
    std::vector<ClippedInfo> r1s, r2s;
    for (auto& nm : order) {
      auto& idx = by_name[nm];
      bool fr = idx.size() == 2 && is_primary_fr_pair_raw(RecView(recs[idx[0]].first, recs[idx[0]].second), RecView(recs[idx[1]].first, recs[idx[1]].second));
      if (!fr) { reject_at(idx, NotPrimaryFrPair); continue; }
      size_t i1, i2;
      if (RecView(recs[idx[0]].first, recs[idx[0]].second).flags() & flags::FIRST_SEGMENT) { i1 = idx[0]; i2 = idx[1]; } else { i1 = idx[1]; i2 = idx[0]; }
      RecView v1(recs[i1].first, recs[i1].second), v2(recs[i2].first, recs[i2].second);
      size_t c1 = num_bases_extending_past_mate_vs_mate_raw(v1, v2), c2 = num_bases_extending_past_mate_vs_mate_raw(v2, v1);
      r1s.push_back(build_clipped_info(recs[i1], i1, c1));
      r2s.push_back(build_clipped_info(recs[i2], i2, c2));
    }
    if (r1s.empty()) return true;
    if (r1s.size() < o.min_reads_per_strand) { reject_at(strand_idx(r1s, r2s), InsufficientReads); return true; }
    r1s = filter_most_common(std::move(r1s));
    r2s = filter_most_common(std::move(r2s));
    if (r1s.empty() || r2s.empty()) return true;
    if (r1s.size() < o.min_reads_per_strand || r2s.size() < o.min_reads_per_strand) { reject_at(strand_idx(r1s, r2s), InsufficientReads); return true; }
    if (o.has_max_reads) {
      if (o.max_reads_per_strand == 0) { reject_at(strand_idx(r1s, r2s), InsufficientReads); return true; }
      size_t d = cap_lowest_ranking(recs, r1s, o.max_reads_per_strand) + cap_lowest_ranking(recs, r2s, o.max_reads_per_strand);
      if (d > 0) reject_count(d, Downsampled);
    }
    const ClippedInfo& l1 = longest(r1s);
    const ClippedInfo& l2 = longest(r2s);
    bool r1_neg = l1.flags & flags::REVERSE;
    const ClippedInfo& lpos = r1_neg ? l2 : l1;
    const ClippedInfo& lneg = r1_neg ? l1 : l2;
    size_t neg_start = lneg.adjusted_pos, pos_start = lpos.adjusted_pos;
    size_t pos_ref = (size_t)(int64_t)reference_length_from_cigar(lpos.clipped_cigar), neg_ref = (size_t)(int64_t)reference_length_from_cigar(lneg.clipped_cigar);
    size_t pos_end = pos_start + (pos_ref > 0 ? pos_ref - 1 : 0), neg_end = neg_start + (neg_ref > 0 ? neg_ref - 1 : 0);
    size_t os = std::max(neg_start, pos_start), oe = std::min(pos_end, neg_end);
    int64_t duplex_len = (int64_t)oe - (int64_t)os + 1;
    if (duplex_len < (int64_t)o.min_duplex_length) { reject_at(strand_idx(r1s, r2s), InsufficientOverlap); return true; }
    if (!check_overlap_phase(l1, l2, os, oe)) { reject_at(strand_idx(r1s, r2s), IndelErrorBetweenStrands); return true; }
    bool r2_neg = l2.flags & flags::REVERSE;
    size_t prp, nrp;
    if (!read_pos_at_ref_pos_raw(lpos.clipped_cigar, lpos.adjusted_pos, oe, false, prp) ||
        !read_pos_at_ref_pos_raw(lneg.clipped_cigar, lneg.adjusted_pos, oe, false, nrp)) {
      reject_at(strand_idx(r1s, r2s), IndelErrorBetweenStrands); return true;
    }
    if (prp + lneg.clipped_seq_len < nrp) throw OracleError{"codec consensus length underflow"};
    size_t cons_len = prp + lneg.clipped_seq_len - nrp;

    auto call_strand = [&](const std::vector<ClippedInfo>& infos, SingleStrand& s) {
      std::vector<SourceRead> srs;
      for (size_t i = 0; i < infos.size(); i++) srs.push_back(to_source_read(recs[infos[i].raw_idx], i, infos[i]));
      VanillaConsensusRead v;
      if (!ss.consensus_call(umi, std::move(srs), v)) return false;
      s.bases = std::move(v.bases); s.quals = std::move(v.quals); s.depths = std::move(v.depths); s.errors = std::move(v.errors);
      return true;
    };
    SingleStrand s1, s2;
    if (!call_strand(r1s, s1)) return true;
    if (!call_strand(r2s, s2)) return true;
    if (cons_len < s1.bases.size() || cons_len < s2.bases.size()) { reject_at(strand_idx(r1s, r2s), ClipOverlapFailed); return true; }
    SingleStrand o1 = r1_neg ? rc(s1) : s1, o2 = r1_neg ? s2 : rc(s2);
    SingleStrand p1 = pad(o1, cons_len, r1_neg), p2 = pad(o2, cons_len, r2_neg);
    SingleStrand cons;
    uint64_t n_dup = 0, n_dis = 0;
    if (build_duplex(p1, p2, cons, n_dup, n_dis) != 0) {
      reject_at(strand_idx(r1s, r2s), HighDuplexDisagreement);
      stats.rejected_hdd += 1;
      return false;
    }
    mask_quals(cons, p1, p2);
    if (r1_neg) cons = rc(cons);
    SingleStrand ac = r1_neg ? rc(p1) : p1, bc = r1_neg ? rc(p2) : p2;
    std::vector<Rec> srcs;
    for (auto& i : r1s) srcs.push_back(recs[i.raw_idx]);
    for (auto& i : r2s) srcs.push_back(recs[i.raw_idx]);
    build_output(out, cons, ac, bc, has_umi, umi, srcs, recs);
    stats.duplex_bases_emitted += n_dup; stats.disagreement_bases += n_dis;
    stats.consensus_reads += 1;
    stats.consensus_bases_emitted += cons.bases.size();
    return true;
  }

  // consensus_reads_typed :1807-1834
  bool consensus_reads(const std::vector<Rec>& recs, ConsensusOutput& out) {
    bool ok = consensus_reads_raw(recs, out);
    if (track) {
      for (size_t i = 0; i < recs.size(); i++) if (i < mask.size() && mask[i]) rejected.emplace_back(recs[i].first, recs[i].first + recs[i].second);
      mask.clear();
    }
    return ok;
  }
};

}  // namespace orc
