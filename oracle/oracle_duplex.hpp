// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Restatement of the duplex caller:
//   crates/fgumi-consensus/src/duplex_caller.rs:344-513 (options / min-reads triple), 590-725 (strand
//   partition), 804-930 (gates, helpers), 931-1108 (duplex_consensus), 1118-1405 (duplex_read_into),
//   1837-2540 (process_group), 2545-2624 (consensus_reads + stats re-attribution)
//   src/lib/commands/duplex.rs:742-830 (process_fn), 942-980 (has_both_strands_raw)
#pragma once
#include "oracle_vanilla.hpp"

namespace orc {

struct DuplexConsensusRead {
  std::string id;
  Bytes bases, quals;
  std::vector<uint16_t> errors;
  VanillaConsensusRead ab;
  bool has_ba = false;
  VanillaConsensusRead ba;
  bool is_ba_only = false;
  bool has_methylation = false;          // combined annotation (duplex_caller.rs:261-262, 1086-1094)
  MethylationAnnotation methylation;
  size_t len() const { return bases.size(); }
};

inline int32_t clamp_per_base_short(uint16_t d) { return (int32_t)std::min<uint16_t>(d, 32767); }   // caller.rs:355-357
inline uint16_t clamp_combined_error(int64_t e) { return (uint16_t)std::min<int64_t>(std::max<int64_t>(e, 0), 32767); }   // :363-368
inline uint8_t cap_quality(int32_t s) { return s < 2 ? 2 : s > 93 ? 93 : (uint8_t)s; }   // duplex_caller.rs:873-881

inline bool is_conversion_pair(uint8_t x, uint8_t y) {   // :897-903
  x = upper(x); y = upper(y);
  return (x == 'C' && y == 'T') || (x == 'T' && y == 'C') || (x == 'G' && y == 'A') || (x == 'A' && y == 'G');
}
inline uint8_t unconverted_base(uint8_t x, uint8_t y) {   // :907-913
  uint8_t a = upper(x), b = upper(y);
  if ((a == 'C' && b == 'T') || (a == 'T' && b == 'C')) return 'C';
  if ((a == 'G' && b == 'A') || (a == 'A' && b == 'G')) return 'G';
  return x;
}

// duplex_consensus :931-1108
inline bool duplex_consensus(const VanillaConsensusRead* ab, const VanillaConsensusRead* ba, const std::vector<SourceRead>* srcs,
                             DuplexConsensusRead& out) {
  size_t len = std::min(ab ? ab->bases.size() : SIZE_MAX, ba ? ba->bases.size() : SIZE_MAX);
  auto covered = [&](const VanillaConsensusRead* v) {
    if (!v) return false;
    for (size_t i = 0; i < std::min(len, v->depths.size()); i++) if (v->depths[i] > 0) return true;
    return false;
  };
  const VanillaConsensusRead* a = covered(ab) ? ab : nullptr;
  const VanillaConsensusRead* b = covered(ba) ? ba : nullptr;
  if (a && !b) { out = DuplexConsensusRead(); out.id = a->id; out.bases = a->bases; out.quals = a->quals; out.errors = a->errors; out.ab = *a; out.ab.source_reads.clear(); out.has_ba = false; out.is_ba_only = false; out.has_methylation = a->has_methylation; out.methylation = a->methylation; return true; }
  if (!a && b) { out = DuplexConsensusRead(); out.id = b->id; out.bases = b->bases; out.quals = b->quals; out.errors = b->errors; out.ab = *b; out.ab.source_reads.clear(); out.has_ba = false; out.is_ba_only = true; out.has_methylation = b->has_methylation; out.methylation = b->methylation; return true; }
  if (!a && !b) return false;
  out = DuplexConsensusRead();
  out.id = a->id;
  for (size_t i = 0; i < len; i++) {
    uint8_t ab_b = a->bases[i], ba_b = b->bases[i];
    int32_t aq = a->quals[i], bq = b->quals[i];
    uint8_t raw_base, raw_qual;
    // a C/T (G/A) disagreement at a reference cytosine of either strand is a conversion event, not an error (:988-1005)
    const bool is_ref_c = (a->has_methylation && i < a->methylation.evidence.size() && a->methylation.evidence[i].is_ref_c) ||
                          (b->has_methylation && i < b->methylation.evidence.size() && b->methylation.evidence[i].is_ref_c);
    const bool artifact = ab_b != ba_b && is_ref_c && is_conversion_pair(ab_b, ba_b);
    if (artifact) { raw_base = unconverted_base(ab_b, ba_b); raw_qual = cap_quality(aq + bq); }
    else if (ab_b == ba_b) { raw_base = ab_b; raw_qual = cap_quality(aq + bq); }
    else if (aq > bq) { raw_base = ab_b; raw_qual = cap_quality(aq - bq); }
    else if (bq > aq) { raw_base = ba_b; raw_qual = cap_quality(bq - aq); }
    else { raw_base = ab_b; raw_qual = MIN_PHRED; }
    if (ab_b == 'N' || ba_b == 'N' || raw_qual == MIN_PHRED) { out.bases.push_back('N'); out.quals.push_back(MIN_PHRED); }
    else { out.bases.push_back(raw_base); out.quals.push_back(raw_qual); }
    uint16_t ec;
    if (artifact) ec = 0;
    else if (srcs) {
      int32_t ne = 0;
      for (auto& sr : *srcs) if (sr.bases.size() > i && sr.bases[i] != 'N' && raw_base != 'N' && sr.bases[i] != raw_base) ne++;
      ec = clamp_combined_error(ne);
    } else {
      int32_t ae = a->errors[i], be = b->errors[i], ad = a->depths[i], bd = b->depths[i];
      int32_t err = (ab_b == ba_b) ? ae + be : (raw_base == ab_b) ? ae + (bd - be) : be + (ad - ae);
      ec = clamp_combined_error(err);
    }
    out.errors.push_back(ec);
  }
  auto trunc = [&](const VanillaConsensusRead& v) {
    VanillaConsensusRead t;
    t.id = v.id;
    t.bases.assign(v.bases.begin(), v.bases.begin() + len); t.quals.assign(v.quals.begin(), v.quals.begin() + len);
    t.depths.assign(v.depths.begin(), v.depths.begin() + len); t.errors.assign(v.errors.begin(), v.errors.begin() + len);
    t.has_methylation = v.has_methylation;
    if (v.has_methylation) t.methylation = v.methylation.truncate(len);
    return t;
  };
  out.ab = trunc(*a); out.has_ba = true; out.ba = trunc(*b); out.is_ba_only = false;
  if (a->has_methylation && b->has_methylation) { out.has_methylation = true; out.methylation = combine_methylation_annotations(a->methylation, b->methylation, len); }
  else if (a->has_methylation || b->has_methylation) { out.has_methylation = true; out.methylation = (a->has_methylation ? a->methylation : b->methylation).truncate(len); }
  return true;
}

struct DuplexOptions {
  size_t min_total = 1, min_xy = 1, min_yx = 0;
  uint8_t min_input_base_quality = 10;
  bool per_base_tags = true, trim = false;
  bool has_max_reads = false; size_t max_reads = 0;
  bool has_cell_tag = true; char cell_tag[2] = {'C', 'B'};
  uint8_t pre = 45, post = 40;
  TieRule tie_rule = TieRule::FgbioCompat;
  int methylation_mode = MethDisabled;   // set_reference :524-536
};

class DuplexCaller {
 public:
  std::string prefix, rg;
  DuplexOptions o;
  Stats stats;
  VanillaCaller ss;
  std::vector<Bytes> rejected;
  bool track;

  static VanillaOptions ss_options(const DuplexOptions& d) {   // :474-489
    VanillaOptions v;
    v.min_reads = 1; v.min_input_base_quality = d.min_input_base_quality; v.produce_per_base_tags = d.per_base_tags; v.trim = d.trim;
    v.has_max_reads = d.has_max_reads; v.max_reads = d.max_reads; v.error_rate_pre_umi = d.pre; v.error_rate_post_umi = d.post;
    v.min_consensus_base_quality = MIN_PHRED; v.has_cell_tag = d.has_cell_tag; v.cell_tag[0] = d.cell_tag[0]; v.cell_tag[1] = d.cell_tag[1];
    v.tie_rule = d.tie_rule;
    v.methylation_mode = d.methylation_mode;
    return v;
  }
  DuplexCaller(std::string p, std::string r, DuplexOptions d, bool track_rejects)
      : prefix(std::move(p)), rg(std::move(r)), o(d), ss(prefix, rg, ss_options(d), track_rejects), track(track_rejects) {}
  void clear() { stats = Stats(); rejected.clear(); ss.clear(); }

  using Rec = std::pair<const uint8_t*, size_t>;
  static uint16_t fl(const Rec& r) { return RecView(r.first, r.second).flags(); }
  static bool is_r1(const Rec& r) { uint16_t f = fl(r); return (f & flags::PAIRED) && (f & flags::FIRST_SEGMENT); }
  static bool is_r2(const Rec& r) { uint16_t f = fl(r); return (f & flags::PAIRED) && (f & flags::LAST_SEGMENT); }

  bool min_reads_ok(size_t na, size_t nb) const {
    size_t xy = std::max(na, nb), yx = std::min(na, nb);
    return o.min_total <= xy + yx && o.min_xy <= xy && o.min_yx <= yx;
  }
  bool consensus_min_reads(const DuplexConsensusRead& c) const { return min_reads_ok(c.ab.max_depth(), c.has_ba ? c.ba.max_depth() : 0); }

  // duplex_read_into :1118-1405
  void duplex_read_into(ConsensusOutput& out, const DuplexConsensusRead& c, ReadType rt, const std::string& umi,
                        const std::vector<Rec>& src_a, const std::vector<Rec>& src_b, bool first_of_pair, const Slice& cell_barcode) {
    uint16_t flag = flags::UNMAPPED;
    if (rt == R1) flag |= flags::PAIRED | flags::FIRST_SEGMENT | flags::MATE_UNMAPPED;
    else if (rt == R2) flag |= flags::PAIRED | flags::LAST_SEGMENT | flags::MATE_UNMAPPED;
    std::string name = prefix + ":" + umi;
    Bytes rec;
    if (!build_unmapped_record(rec, (const uint8_t*)name.data(), name.size(), flag, c.bases.data(), c.quals.data(), c.bases.size()))
      throw OracleError{"could not write the consensus record: read name too long"};
    append_string_tag(rec, "MI", (const uint8_t*)umi.data(), umi.size());
    if (o.has_cell_tag && cell_barcode.some) append_string_tag(rec, o.cell_tag, cell_barcode.p, cell_barcode.n);
    append_string_tag(rec, "RG", (const uint8_t*)rg.data(), rg.size());
    auto strand = [&](const VanillaConsensusRead& v, int32_t& dmax, int32_t& dmin, float& er) {
      dmax = 0; dmin = 0;
      int64_t td = 0, te = 0;
      for (size_t i = 0; i < v.depths.size(); i++) { int32_t d = clamp_per_base_short(v.depths[i]); if (i == 0) { dmax = d; dmin = d; } dmax = std::max(dmax, d); dmin = std::min(dmin, d); td += d; }
      for (auto e : v.errors) te += clamp_per_base_short(e);
      er = td > 0 ? (float)te / (float)td : 0.0f;
    };
    auto per_base = [&](const VanillaConsensusRead& v, const char* tc, const char* td_, const char* te_, const char* tq) {
      append_string_tag(rec, tc, v.bases.data(), v.bases.size());
      std::vector<int16_t> d(v.depths.size()), e(v.errors.size());
      for (size_t i = 0; i < d.size(); i++) d[i] = v.depths[i] > 32767 ? 32767 : (int16_t)v.depths[i];
      for (size_t i = 0; i < e.size(); i++) e[i] = v.errors[i] > 32767 ? 32767 : (int16_t)v.errors[i];
      append_i16_array_tag(rec, td_, d.data(), d.size());
      append_i16_array_tag(rec, te_, e.data(), e.size());
      append_phred33_string_tag(rec, tq, v.quals.data(), v.quals.size());
    };
    int32_t amax, amin; float aer;
    strand(c.ab, amax, amin, aer);
    append_int_tag(rec, "aD", amax); append_float_tag(rec, "aE", aer); append_int_tag(rec, "aM", amin);
    if (o.per_base_tags) per_base(c.ab, "ac", "ad", "ae", "aq");
    int32_t bmax = 0, bmin = 0; float ber = 0.0f;
    if (c.has_ba) strand(c.ba, bmax, bmin, ber);
    append_int_tag(rec, "bD", bmax); append_float_tag(rec, "bE", ber); append_int_tag(rec, "bM", bmin);
    if (o.per_base_tags && c.has_ba) per_base(c.ba, "bc", "bd", "be", "bq");
    int32_t cmax = 0, cmin = 0;
    int64_t td = 0, te = 0;
    for (size_t i = 0; i < c.len(); i++) {
      int32_t d = clamp_per_base_short(i < c.ab.depths.size() ? c.ab.depths[i] : 0) + clamp_per_base_short(c.has_ba && i < c.ba.depths.size() ? c.ba.depths[i] : 0);
      if (i == 0) { cmax = d; cmin = d; }
      cmax = std::max(cmax, d); cmin = std::min(cmin, d); td += d;
    }
    for (auto e : c.errors) te += clamp_per_base_short(e);
    float cer = td > 0 ? (float)te / (float)td : 0.0f;
    append_int_tag(rec, "cD", cmax); append_float_tag(rec, "cE", cer); append_int_tag(rec, "cM", cmin);
    std::vector<std::string> umis;
    auto add_umis = [&](const std::vector<Rec>& src) {
      for (auto& r : src) {
        RecView v(r.first, r.second);
        Slice rx = find_string_tag(v.aux(), "RX");
        if (!rx.some) continue;
        std::string s = rx.str();
        bool is_first = v.flags() & flags::FIRST_SEGMENT;
        if (is_first == first_of_pair) umis.push_back(s);
        else {   // rx.split('-').rev().join("-")
          std::vector<std::string> parts;
          size_t st = 0;
          for (;;) { size_t k = s.find('-', st); if (k == std::string::npos) { parts.push_back(s.substr(st)); break; } parts.push_back(s.substr(st, k - st)); st = k + 1; }
          std::string j;
          for (size_t i = parts.size(); i-- > 0;) { j += parts[i]; if (i) j += "-"; }
          umis.push_back(j);
        }
      }
    };
    add_umis(src_a); add_umis(src_b);
    if (!umis.empty()) { std::string cu = consensus_umis(umis); append_string_tag(rec, "RX", (const uint8_t*)cu.data(), cu.size()); }
    if (c.has_methylation) {   // :1338-1398
      const bool top = !c.is_ba_only;
      auto counts = [&](const MethylationAnnotation& m, const char* tu, const char* tt) {
        std::vector<int16_t> u = m.unconverted_counts(), t = m.converted_counts();
        append_i16_array_tag(rec, tu, u.data(), u.size());
        append_i16_array_tag(rec, tt, t.data(), t.size());
      };
      std::string mm; Bytes ml;
      if (c.ab.has_methylation) {
        if (build_mm_ml_tags(c.ab.bases, c.ab.methylation, top, o.methylation_mode, mm, ml)) append_string_tag(rec, top ? "am" : "bm", (const uint8_t*)mm.data(), mm.size());
        counts(c.ab.methylation, top ? "au" : "bu", top ? "at" : "bt");
      }
      if (c.has_ba && c.ba.has_methylation) {
        if (build_mm_ml_tags(c.ba.bases, c.ba.methylation, false, o.methylation_mode, mm, ml)) append_string_tag(rec, "bm", (const uint8_t*)mm.data(), mm.size());
        counts(c.ba.methylation, "bu", "bt");
      }
      if (build_mm_ml_tags(c.bases, c.methylation, top, o.methylation_mode, mm, ml)) {
        append_string_tag(rec, "MM", (const uint8_t*)mm.data(), mm.size());
        append_u8_array_tag(rec, "ML", ml.data(), ml.size());
      }
      counts(c.methylation, "cu", "ct");
    }
    write_with_block_size(rec, out.data);
    out.count += 1;
  }

  // process_group :1944-2540. Returns kept=false for a whole-group rejection (rejected_raw filled when tracking).
  ConsensusOutput process_group(const std::string& base_mi, const std::vector<Rec>& a, const std::vector<Rec>& b, Stats& gs, bool& kept,
                                std::vector<Bytes>& rejected_raw) {
    kept = true;
    ConsensusOutput output;
    auto reject_all = [&](Rejection why) {
      gs.record_rejection(why, a.size() + b.size());
      kept = false;
      if (track) { for (auto& r : a) rejected_raw.emplace_back(r.first, r.first + r.second); for (auto& r : b) rejected_raw.emplace_back(r.first, r.first + r.second); }
      return ConsensusOutput();
    };
    if (a.empty() && b.empty()) return output;
    size_t na = 0, nb = 0;
    for (auto& r : a) na += is_r1(r);
    for (auto& r : b) nb += is_r1(r);
    if (!min_reads_ok(na, nb)) return reject_all(InsufficientReads);
    Slice cell_barcode;
    if (o.has_cell_tag) { const Rec* f = !a.empty() ? &a[0] : (!b.empty() ? &b[0] : nullptr); if (f) cell_barcode = find_string_tag(RecView(f->first, f->second).aux(), o.cell_tag); }
    std::vector<Rec> ab_r1, ab_r2, ba_r1, ba_r2;
    for (auto& r : a) { if (is_r1(r)) ab_r1.push_back(r); if (is_r2(r)) ab_r2.push_back(r); }
    for (auto& r : b) { if (is_r1(r)) ba_r1.push_back(r); if (is_r2(r)) ba_r2.push_back(r); }
    if (!a.empty() && !b.empty()) {
      auto same_strand = [&](const std::vector<Rec>& p, const std::vector<Rec>& q) {
        bool have = false, first_rev = false;
        for (auto* v : {&p, &q}) for (auto& r : *v) { bool rv = fl(r) & flags::REVERSE; if (!have) { have = true; first_rev = rv; } else if (rv != first_rev) return false; }
        return true;
      };
      if (!same_strand(ab_r1, ba_r2)) return reject_all(PotentialCollision);
      if (!same_strand(ab_r2, ba_r1)) return reject_all(PotentialCollision);
    }
    std::vector<Rec> x_raws = ab_r1, y_raws = ab_r2;
    x_raws.insert(x_raws.end(), ba_r2.begin(), ba_r2.end());
    y_raws.insert(y_raws.end(), ba_r1.begin(), ba_r1.end());
    auto to_sources = [&](const std::vector<Rec>& raws, std::vector<SourceRead>& srcs, std::vector<size_t>& zero) {
      for (size_t i = 0; i < raws.size(); i++) {
        size_t clip = num_bases_extending_past_mate_raw(RecView(raws[i].first, raws[i].second));
        SourceRead sr;
        if (ss.create_source_read(raws[i].first, raws[i].second, i, clip, sr)) srcs.push_back(std::move(sr)); else zero.push_back(i);
      }
    };
    std::vector<SourceRead> xs, ys;
    std::vector<size_t> xz, yz;
    to_sources(x_raws, xs, xz);
    to_sources(y_raws, ys, yz);
    std::vector<size_t> x_rej, y_rej;
    std::vector<SourceRead> fx = ss.filter_source_reads_by_alignment(std::move(xs), x_rej);
    std::vector<SourceRead> fy = ss.filter_source_reads_by_alignment(std::move(ys), y_rej);
    // ordinals: position of each record in (a ++ b)
    auto ordinals = [&](bool a_r1_b_r2) {
      std::vector<size_t> o1;
      for (size_t i = 0; i < a.size(); i++) if (a_r1_b_r2 ? is_r1(a[i]) : is_r2(a[i])) o1.push_back(i);
      for (size_t j = 0; j < b.size(); j++) if (a_r1_b_r2 ? is_r2(b[j]) : is_r1(b[j])) o1.push_back(a.size() + j);
      return o1;
    };
    {
      size_t nz = xz.size() + yz.size();
      if (track) {
        std::vector<size_t> xo = ordinals(true), yo = ordinals(false);
        std::vector<std::pair<size_t, Rec>> zr, sr;
        for (size_t i : xz) zr.push_back({xo[i], x_raws[i]});
        for (size_t i : yz) zr.push_back({yo[i], y_raws[i]});
        std::stable_sort(zr.begin(), zr.end(), [](const auto& p, const auto& q) { return p.first < q.first; });
        if (nz) { ss.stats.record_rejection(ZeroLengthAfterTrimming, nz); for (auto& e : zr) ss.rejected_reads.emplace_back(e.second.first, e.second.first + e.second.second); }
        for (size_t i : x_rej) sr.push_back({xo[i], x_raws[i]});
        for (size_t i : y_rej) sr.push_back({yo[i], y_raws[i]});
        std::stable_sort(sr.begin(), sr.end(), [](const auto& p, const auto& q) { return p.first < q.first; });
        for (auto& e : sr) ss.rejected_reads.emplace_back(e.second.first, e.second.first + e.second.second);
      } else if (nz) ss.stats.record_rejection(ZeroLengthAfterTrimming, nz);
    }
    auto split = [&](std::vector<SourceRead>& f, std::vector<SourceRead>& first, std::vector<SourceRead>& notfirst) {
      for (auto& s : f) { if (s.flags & flags::FIRST_SEGMENT) first.push_back(std::move(s)); else notfirst.push_back(std::move(s)); }
    };
    std::vector<SourceRead> f_ab_r1, f_ba_r2, f_ba_r1, f_ab_r2;
    split(fx, f_ab_r1, f_ba_r2);
    split(fy, f_ba_r1, f_ab_r2);
    auto raws_of = [&](const std::vector<SourceRead>& v, const std::vector<Rec>& raws) { std::vector<Rec> r; for (auto& s : v) r.push_back(raws[s.original_idx]); return r; };
    std::vector<Rec> ab_r1_raws = raws_of(f_ab_r1, x_raws), ba_r2_raws = raws_of(f_ba_r2, x_raws), ab_r2_raws = raws_of(f_ab_r2, y_raws), ba_r1_raws = raws_of(f_ba_r1, y_raws);
    std::string ab_umi = base_mi + "/A", ba_umi = base_mi + "/B";
    VanillaConsensusRead c_ab_r1, c_ab_r2, c_ba_r1, c_ba_r2;
    bool h_ab_r1 = ss.consensus_call(ab_umi, std::move(f_ab_r1), c_ab_r1);
    bool h_ab_r2 = ss.consensus_call(ab_umi, std::move(f_ab_r2), c_ab_r2);
    bool h_ba_r1 = ss.consensus_call(ba_umi, std::move(f_ba_r1), c_ba_r1);
    bool h_ba_r2 = ss.consensus_call(ba_umi, std::move(f_ba_r2), c_ba_r2);
    std::vector<Rec> empty;
    if (h_ab_r1 && h_ab_r2 && h_ba_r1 && h_ba_r2) {
      std::vector<SourceRead> r1s = c_ab_r1.source_reads, r2s = c_ab_r2.source_reads;
      r1s.insert(r1s.end(), c_ba_r2.source_reads.begin(), c_ba_r2.source_reads.end());
      r2s.insert(r2s.end(), c_ba_r1.source_reads.begin(), c_ba_r1.source_reads.end());
      DuplexConsensusRead d1, d2;
      bool k1 = duplex_consensus(&c_ab_r1, &c_ba_r2, r1s.empty() ? nullptr : &r1s, d1);
      bool k2 = duplex_consensus(&c_ab_r2, &c_ba_r1, r2s.empty() ? nullptr : &r2s, d2);
      if (k1 && k2) {
        if (consensus_min_reads(d1) && consensus_min_reads(d2)) {
          duplex_read_into(output, d1, R1, base_mi, ab_r1_raws, ba_r2_raws, true, cell_barcode);
          duplex_read_into(output, d2, R2, base_mi, ab_r2_raws, ba_r1_raws, false, cell_barcode);
          gs.consensus_reads += 2;
          return output;
        }
        return reject_all(InsufficientReads);
      }
    } else if (h_ab_r1 && h_ab_r2 && !h_ba_r1 && !h_ba_r2) {
      if (o.min_yx == 0) {
        DuplexConsensusRead d1, d2;
        if (duplex_consensus(&c_ab_r1, nullptr, nullptr, d1) && duplex_consensus(&c_ab_r2, nullptr, nullptr, d2) && consensus_min_reads(d1) && consensus_min_reads(d2)) {
          duplex_read_into(output, d1, R1, base_mi, ab_r1_raws, empty, true, cell_barcode);
          duplex_read_into(output, d2, R2, base_mi, ab_r2_raws, empty, false, cell_barcode);
          gs.consensus_reads += 2;
          return output;
        }
      }
    } else if (!h_ab_r1 && !h_ab_r2 && h_ba_r1 && h_ba_r2) {
      if (o.min_yx == 0) {
        DuplexConsensusRead d1, d2;
        if (duplex_consensus(nullptr, &c_ba_r1, nullptr, d1) && duplex_consensus(nullptr, &c_ba_r2, nullptr, d2) && consensus_min_reads(d1) && consensus_min_reads(d2)) {
          duplex_read_into(output, d1, R1, base_mi, empty, ba_r1_raws, true, cell_barcode);
          duplex_read_into(output, d2, R2, base_mi, empty, ba_r2_raws, false, cell_barcode);
          gs.consensus_reads += 2;
          return output;
        }
      }
    }
    return reject_all(InsufficientReads);
  }

  // ConsensusCaller::consensus_reads :2545-2624
  ConsensusOutput consensus_reads(const std::vector<Rec>& records) {
    stats.record_input(records.size());
    std::vector<Rec> paired;
    size_t nfrag = 0;
    for (auto& r : records) { if (fl(r) & flags::PAIRED) paired.push_back(r); else { nfrag++; if (track) rejected.emplace_back(r.first, r.first + r.second); } }
    if (nfrag) stats.record_rejection(FragmentRead, nfrag);
    std::vector<Rec> a, b;
    bool have_mi = false;
    std::string base_mi;
    for (auto& r : paired) {
      Slice mi = find_string_tag(RecView(r.first, r.second).aux(), "MI");
      if (!mi.some) throw OracleError{"Read is missing MI tag"};
      if (!have_mi) { have_mi = true; base_mi = mi.n >= 2 ? std::string((const char*)mi.p, mi.n - 2) : mi.str(); }
      char strand = 0;
      if (mi.n >= 2 && mi.p[mi.n - 2] == '/') { if (mi.p[mi.n - 1] == 'A') strand = 'A'; else if (mi.p[mi.n - 1] == 'B') strand = 'B'; }
      if (strand == 'A') a.push_back(r); else if (strand == 'B') b.push_back(r); else throw OracleError{"Read has MI tag without /A or /B suffix"};
    }
    if (!have_mi) return ConsensusOutput();
    Stats gs;
    bool kept;
    std::vector<Bytes> dup_rej;
    ConsensusOutput out = process_group(base_mi, a, b, gs, kept, dup_rej);
    stats.merge(gs);
    Stats ss_stats = ss.stats;
    ss.stats = Stats();
    std::vector<Bytes> ss_rej;
    ss_rej.swap(ss.rejected_reads);
    if (kept) {
      stats.merge(ss_stats);
      if (track) for (auto& r : ss_rej) rejected.push_back(std::move(r));
    } else {
      // reattribute_single_strand_rejections :1894-1926
      uint64_t ssr = ss_stats.filtered_reads;
      if (ssr != 0) {
        int reason = -1;
        for (int i = 0; i < N_REJECTION; i++) if (gs.rejection[i]) { reason = i; break; }
        if (reason >= 0) {
          if (stats.rejection[reason]) stats.rejection[reason] = stats.rejection[reason] > ssr ? stats.rejection[reason] - ssr : 0;
          stats.filtered_reads = stats.filtered_reads > ssr ? stats.filtered_reads - ssr : 0;
        }
        stats.merge(ss_stats);
      }
      if (track) for (auto& r : dup_rej) rejected.push_back(std::move(r));
    }
    return out;
  }
};

inline bool has_both_strands_raw(const std::vector<Bytes>& recs) {   // duplex.rs:942-980
  if (recs.size() < 2) return false;
  bool ha = false, hb = false;
  for (auto& r : recs) {
    Slice mi = find_string_tag(RecView(r.data(), r.size()).aux(), "MI");
    if (!mi.some) continue;
    if (mi.n >= 2 && mi.p[mi.n - 2] == '/') {
      if (mi.p[mi.n - 1] == 'A') { ha = true; if (hb) return true; }
      else if (mi.p[mi.n - 1] == 'B') { hb = true; if (ha) return true; }
    }
  }
  return false;
}

}  // namespace orc
