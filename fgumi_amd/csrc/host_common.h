// host_common.h — host-side building blocks shared by the simplex / duplex / CODEC general paths.
// (Product code; independent of oracle/.)  Reference interfaces are cited per function.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/fgumi_amd.h"
#include "bamrec.h"
#include "consensus_math.h"

namespace fgx {

struct HostStats {  // ConsensusCallingStats (caller.rs:256-321)
  uint64_t total_reads = 0, consensus_reads = 0, filtered_reads = 0;
  uint64_t rej[FGX_N_REJECTION] = {0};
  void reject(int reason, size_t n) { filtered_reads += n; rej[reason] += n; }
  void to_array(uint64_t* out) const {
    out[0] = total_reads; out[1] = consensus_reads; out[2] = filtered_reads;
    for (int i = 0; i < FGX_N_REJECTION; i++) out[3 + i] = rej[i];
  }
};

inline void append_with_block_size(std::vector<uint8_t>& out, const uint8_t* p, uint32_t n) {  // builder.rs:287-293
  uint8_t h[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
  out.insert(out.end(), h, h + 4);
  out.insert(out.end(), p, p + n);
}

// ---- simplified CIGARs (fgumi-sam clipper.rs:1183-1241, raw-bam noodles_compat.rs:10-55) -------
using SimpCigar = std::vector<std::pair<uint8_t, uint64_t>>;  // (BAM op code after S,=,X,H→M ; length)
inline SimpCigar simplify_cigar(const bam::Rec& v) {
  SimpCigar out;
  uint32_t n = v.n_cigar();
  if ((uint64_t)v.cigar_off() + 4ull * n > v.len) return out;   // get_cigar_ops: out-of-bounds → empty
  for (uint32_t i = 0; i < n; i++) {
    uint32_t raw = v.cigar_op(i);
    uint32_t t = raw & 0xF;
    if (t > 8) continue;
    uint8_t k = (t == 4 || t == 5 || t == 7 || t == 8) ? 0 : (uint8_t)t;
    if (!out.empty() && out.back().first == k) out.back().second += raw >> 4;
    else out.push_back({k, raw >> 4});
  }
  return out;
}
inline SimpCigar truncate_cigar(const SimpCigar& c, uint64_t query_len) {  // vanilla_caller.rs:1028-1062
  SimpCigar r;
  uint64_t remaining = query_len;
  for (auto& op : c) {
    if (remaining == 0) break;
    if (op.first == 0 || op.first == 1) { uint64_t take = std::min(op.second, remaining); r.push_back({op.first, take}); remaining -= take; }
    else r.push_back(op);
  }
  return r;
}
inline bool cigar_is_prefix(const SimpCigar& a, const SimpCigar& b) {
  if (a.size() > b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) {
    if (a[i].first != b[i].first) return false;
    if (i + 1 == a.size()) { if (a[i].second > b[i].second) return false; }
    else if (a[i].second != b[i].second) return false;
  }
  return true;
}
inline int cigar_cmp(const SimpCigar& a, const SimpCigar& b) {
  size_t n = std::min(a.size(), b.size());
  for (size_t i = 0; i < n; i++) {
    if (a[i].second != b[i].second) return a[i].second < b[i].second ? -1 : 1;
    if (a[i].first != b[i].first) return a[i].first < b[i].first ? -1 : 1;
  }
  return a.size() == b.size() ? 0 : (a.size() < b.size() ? -1 : 1);
}
// select_most_common_alignment_group (vanilla_caller.rs:48-120). `cigs` are in descending-length
// order; returns positions (into cigs) of the winning group.
inline std::vector<uint32_t> most_common_alignment_group(const std::vector<const SimpCigar*>& cigs) {
  std::vector<uint32_t> all;
  if (cigs.size() < 2) { for (uint32_t i = 0; i < cigs.size(); i++) all.push_back(i); return all; }
  struct Group { const SimpCigar* cigar; std::vector<uint32_t> members; };
  std::vector<Group> groups;
  for (uint32_t i = 0; i < cigs.size(); i++) {
    bool found = false;
    for (auto& g : groups) if (cigar_is_prefix(*cigs[i], *g.cigar)) { g.members.push_back(i); found = true; }   // no break (fgbio)
    if (!found) groups.push_back(Group{cigs[i], {i}});
  }
  size_t best = 0;   // max_by: larger group wins, then smaller CIGAR; later element wins exact ties
  for (size_t i = 1; i < groups.size(); i++) {
    int c;
    if (groups[best].members.size() != groups[i].members.size()) c = groups[best].members.size() < groups[i].members.size() ? -1 : 1;
    else c = cigar_cmp(*groups[i].cigar, *groups[best].cigar);
    if (c <= 0) best = i;
  }
  return groups[best].members;
}

// select_lowest_ranking (caller.rs:665-674)
inline std::vector<uint32_t> lowest_ranking(const std::vector<int32_t>& ranks, size_t max_reads) {
  std::vector<uint32_t> idx(ranks.size());
  for (uint32_t i = 0; i < idx.size(); i++) idx[i] = i;
  if (ranks.size() <= max_reads) return idx;
  std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ranks[a] < ranks[b]; });
  idx.resize(max_reads);
  std::sort(idx.begin(), idx.end());
  return idx;
}

// find_quality_trim_point (vanilla_caller.rs:992-1016)
inline uint32_t quality_trim_point(const uint8_t* q, uint32_t n, uint8_t trim_qual) {
  if (trim_qual < 1 || n == 0) return 0;
  int32_t score = 0, max_score = 0;
  uint32_t point = n;
  for (uint32_t i = n; i-- > 0;) {
    score += (int32_t)trim_qual - (int32_t)q[i];
    if (score < 0) break;
    if (score > max_score) { max_score = score; point = i; }
  }
  return point;
}

// fgbio_read_name_rank: Murmur3_32 over UTF-16 code units, seed 42 (raw-bam/hash.rs:14-89)
inline int32_t read_name_rank(const uint8_t* name, uint32_t len) {
  auto rotl = [](uint32_t v, int r) { return (v << r) | (v >> (32 - r)); };
  auto mixk = [&](uint32_t k) { k *= 0xcc9e2d51u; k = rotl(k, 15); return k * 0x1b873593u; };
  uint32_t h = 42;
  for (uint32_t i = 1; i < len; i += 2) { h ^= mixk((uint32_t)name[i - 1] | ((uint32_t)name[i] << 16)); h = rotl(h, 13) * 5 + 0xe6546b64u; }
  if (len & 1) h ^= mixk(name[len - 1]);
  h ^= 2 * len;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return (int32_t)h;
}

// ---- mate-overlap clip (raw-bam/overlap.rs:21-207) -----------------------------------------------
inline std::vector<uint32_t> cigar_ops_vec(const bam::Rec& v) {
  std::vector<uint32_t> ops;
  uint32_t n = v.n_cigar();
  if ((uint64_t)v.cigar_off() + 4ull * n > v.len) return ops;
  for (uint32_t i = 0; i < n; i++) ops.push_back(v.cigar_op(i));
  return ops;
}
inline bool is_fr_pair_raw(const bam::Rec& v) {   // :21-69
  uint16_t f = v.flags();
  if (!(f & bam::F_PAIRED) || (f & bam::F_UNMAPPED) || (f & bam::F_MATE_UNMAPPED)) return false;
  if (v.ref_id() != v.mate_ref_id()) return false;
  bool rev = f & bam::F_REVERSE, mrev = f & bam::F_MATE_REVERSE;
  if (rev == mrev) return false;
  uint32_t astart = (uint32_t)v.pos() + 1u, mstart = (uint32_t)v.mate_pos() + 1u;
  int32_t p5, n5;
  if (rev) {
    std::vector<uint32_t> ops = cigar_ops_vec(v);
    int32_t rl = bam::ref_len_checked0(ops.data(), (uint32_t)ops.size());
    p5 = (int32_t)mstart;
    n5 = (int32_t)(astart + (uint32_t)std::max(rl - 1, 0));
  } else {
    p5 = (int32_t)astart;
    n5 = (int32_t)(astart + (uint32_t)v.tlen());
  }
  return p5 < n5;
}
inline bool is_primary_fr_pair_raw(const bam::Rec& a, const bam::Rec& b) {   // :83-108
  uint16_t fa = a.flags(), fb = b.flags();
  if ((fa | fb) & bam::F_UNMAPPED) return false;
  if ((fa | fb) & bam::F_MATE_UNMAPPED) return false;
  if (a.ref_id() != b.ref_id()) return false;
  bool ar = fa & bam::F_REVERSE, br = fb & bam::F_REVERSE;
  if (ar == br) return false;
  return is_fr_pair_raw(ar ? a : b);
}
inline uint64_t mate_clip_raw(const bam::Rec& v) {   // num_bases_extending_past_mate_raw :181-207
  uint32_t aux_n = v.len > v.aux_off() ? v.len - v.aux_off() : 0;
  uint32_t vl;
  int64_t off = bam::find_z_tag(v.b + v.aux_off(), aux_n, 'M', 'C', &vl);
  if (off < 0) return 0;
  std::vector<uint32_t> mops(vl + 1);
  int n = bam::parse_mc(v.b + v.aux_off() + off, vl, mops.data(), (uint32_t)mops.size());
  if (n <= 0) return 0;
  int32_t mate_ref_len = bam::sat_ref_len(mops.data(), (uint32_t)n);
  // is_fr_pair_with_mate_cigar_raw :129-161
  uint16_t f = v.flags();
  if (!(f & bam::F_PAIRED) || (f & bam::F_UNMAPPED) || (f & bam::F_MATE_UNMAPPED)) return 0;
  if (v.ref_id() != v.mate_ref_id()) return 0;
  bool rev = f & bam::F_REVERSE, mrev = f & bam::F_MATE_REVERSE;
  if (rev == mrev) return 0;
  int32_t this_pos1 = (int32_t)((uint32_t)v.pos() + 1u), mate_pos1 = (int32_t)((uint32_t)v.mate_pos() + 1u);
  if (rev) { if (!is_fr_pair_raw(v)) return 0; }
  else {
    int32_t mate_end = bam::sat_add(mate_pos1, std::max(mate_ref_len - 1, 0));
    if (!(this_pos1 < mate_end)) return 0;
  }
  std::vector<uint32_t> ops = cigar_ops_vec(v);
  return bam::past_mate_ops(rev, this_pos1, ops.data(), (uint32_t)ops.size(), mate_pos1, mops.data(), (uint32_t)n);
}
inline uint64_t mate_clip_vs_mate_raw(const bam::Rec& rec, const bam::Rec& mate) {   // :223-230
  if (!is_primary_fr_pair_raw(rec, mate)) return 0;
  std::vector<uint32_t> ops = cigar_ops_vec(rec), mops = cigar_ops_vec(mate);
  return bam::past_mate_ops(rec.flags() & bam::F_REVERSE, (int32_t)((uint32_t)rec.pos() + 1u), ops.data(), (uint32_t)ops.size(),
                            (int32_t)((uint32_t)mate.pos() + 1u), mops.data(), (uint32_t)mops.size());
}

// ---- overlapping-bases pre-correction (overlapping.rs:236-336, 382-684), Consensus/Consensus ----
struct MutRec { uint8_t* p; uint32_t n; };

// Aligned (query offset, ref pos) pairs of `v` restricted to ref window [lo, hi] (1-based, inclusive).
inline void aligned_positions(const bam::Rec& v, int64_t lo, int64_t hi, std::vector<std::pair<uint32_t, int64_t>>& out) {
  out.clear();
  int64_t ref = (int64_t)v.pos() + 1, q = 0, rec_len = v.l_seq();
  uint32_t n = v.n_cigar();
  for (uint32_t i = 0; i < n && ref <= hi && q < rec_len; i++) {
    uint32_t op = v.cigar_op(i), t = op & 0xF;
    int64_t len = op >> 4;
    if (t == 0 || t == 7 || t == 8) {
      int64_t k0 = std::max<int64_t>(0, lo - ref);
      for (int64_t k = k0; k < len; k++) {
        if (ref + k > hi || q + k >= rec_len) break;
        out.push_back({(uint32_t)(q + k), ref + k});
      }
      ref += len; q += len;
    } else if (t == 1 || t == 4) q += len;
    else if (t == 2 || t == 3) ref += len;
  }
}

inline bool overlap_call(MutRec a, MutRec b, uint64_t* st) {
  bam::Rec v1{a.p, a.n}, v2{b.p, b.n};
  if ((v1.flags() | v2.flags()) & bam::F_UNMAPPED) return false;
  if (v1.ref_id() != v2.ref_id()) return false;
  if (v1.pos() < 0 || v2.pos() < 0) return false;
  std::vector<uint32_t> o1 = cigar_ops_vec(v1), o2 = cigar_ops_vec(v2);
  int32_t rl1 = bam::ref_len_checked0(o1.data(), (uint32_t)o1.size()), rl2 = bam::ref_len_checked0(o2.data(), (uint32_t)o2.size());
  if (rl1 == 0 || rl2 == 0) return false;
  int64_t s1 = (int64_t)v1.pos() + 1, e1 = (int64_t)v1.pos() + rl1, s2 = (int64_t)v2.pos() + 1, e2 = (int64_t)v2.pos() + rl2;
  int64_t lo = std::max(s1, s2), hi = std::min(e1, e2);
  std::vector<std::pair<uint32_t, int64_t>> p1, p2;
  aligned_positions(v1, lo, hi, p1);
  aligned_positions(v2, lo, hi, p2);
  size_t i = 0, j = 0;
  uint8_t* q1 = a.p + v1.qual_off();
  uint8_t* q2 = b.p + v2.qual_off();
  uint32_t so1 = v1.seq_off(), so2 = v2.seq_off();
  bool any = false;
  while (i < p1.size() && j < p2.size()) {
    if (p1[i].second < p2[j].second) { i++; continue; }
    if (p1[i].second > p2[j].second) { j++; continue; }
    any = true;
    uint32_t x = p1[i].first, y = p2[j].first;
    i++; j++;
    uint8_t c1 = v1.base_code(x), c2 = v2.base_code(y);
    if (c1 == 15 || c2 == 15) continue;            // is_no_call: decoded 'N' (n and '.' cannot come out of a BAM nibble)
    st[0]++;
    uint8_t qa = q1[x], qb = q2[y];
    if (c1 == c2) {
      st[1]++;
      uint8_t nq = (uint8_t)std::min<unsigned>((unsigned)qa + qb, 93);
      q1[x] = nq; q2[y] = nq;
      if (nq != qa || nq != qb) st[3]++;
    } else {
      st[2]++;
      uint8_t cb, cq;
      if (qa == qb) { cb = 15; cq = FGX_MIN_PHRED; }
      else if (qa > qb) { cb = c1; cq = std::max<uint8_t>((uint8_t)(qa - qb), FGX_MIN_PHRED); }
      else { cb = c2; cq = std::max<uint8_t>((uint8_t)(qb - qa), FGX_MIN_PHRED); }
      auto setc = [](uint8_t* rec, uint32_t so, uint32_t pos, uint8_t code) {
        uint8_t& byte = rec[so + (pos >> 1)];
        byte = (pos & 1) ? (uint8_t)((byte & 0xF0) | code) : (uint8_t)((code << 4) | (byte & 0x0F));
      };
      setc(a.p, so1, x, cb);
      setc(b.p, so2, y, cb);
      q1[x] = cq; q2[y] = cq;
      st[3] += 2;
    }
  }
  return any;
}

// apply_overlapping_consensus (overlapping.rs:627-684): pair primaries by read name within the group.
inline void apply_overlapping_consensus(std::vector<MutRec>& recs, uint64_t* st) {
  struct Pair { int64_t r1 = -1, r2 = -1; };
  std::unordered_map<std::string, Pair> pairs;
  std::vector<std::string> order;   // pairs are disjoint, processing order is unobservable; keep it deterministic
  for (size_t i = 0; i < recs.size(); i++) {
    bam::Rec v{recs[i].p, recs[i].n};
    uint16_t f = v.flags();
    if (f & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) continue;
    if (!(f & (bam::F_FIRST | bam::F_LAST))) continue;
    std::string name((const char*)v.name(), v.name_len());
    auto it = pairs.find(name);
    if (it == pairs.end()) { it = pairs.emplace(name, Pair{}).first; order.push_back(name); }
    if (f & bam::F_FIRST) it->second.r1 = (int64_t)i; else it->second.r2 = (int64_t)i;
  }
  for (auto& nm : order) {
    Pair& p = pairs[nm];
    if (p.r1 >= 0 && p.r2 >= 0) overlap_call(recs[p.r1], recs[p.r2], st);
  }
}

// ---- BAM record assembly (raw-bam/builder.rs:122-301, tags.rs:650-760) ---------------------------
inline void put_le32(std::vector<uint8_t>& r, uint32_t v) { for (int i = 0; i < 4; i++) r.push_back((uint8_t)(v >> (8 * i))); }
inline void put_le16(std::vector<uint8_t>& r, uint16_t v) { r.push_back((uint8_t)v); r.push_back((uint8_t)(v >> 8)); }
inline bool build_unmapped_record(std::vector<uint8_t>& r, const std::string& name, uint16_t flag, const uint8_t* bases,
                                  const uint8_t* quals, uint32_t n) {
  r.clear();
  if (name.size() >= 255) return false;
  put_le32(r, 0xFFFFFFFFu); put_le32(r, 0xFFFFFFFFu);
  r.push_back((uint8_t)(name.size() + 1)); r.push_back(0);
  put_le16(r, 4680); put_le16(r, 0); put_le16(r, flag); put_le32(r, n);
  put_le32(r, 0xFFFFFFFFu); put_le32(r, 0xFFFFFFFFu); put_le32(r, 0);
  r.insert(r.end(), name.begin(), name.end()); r.push_back(0);
  for (uint32_t i = 0; i + 1 < n; i += 2) r.push_back((uint8_t)((bam::ascii_to_code(bases[i]) << 4) | bam::ascii_to_code(bases[i + 1])));
  if (n & 1) r.push_back((uint8_t)(bam::ascii_to_code(bases[n - 1]) << 4));
  r.insert(r.end(), quals, quals + n);
  return true;
}
inline void tag_z(std::vector<uint8_t>& r, const char* tag, const char* v, size_t n) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('Z');
  r.insert(r.end(), v, v + n); r.push_back(0);
}
inline void tag_int(std::vector<uint8_t>& r, const char* tag, int32_t v) {   // smallest type, signed first: c C S s i
  r.push_back(tag[0]); r.push_back(tag[1]);
  if (v >= -128 && v <= 127) { r.push_back('c'); r.push_back((uint8_t)(int8_t)v); }
  else if (v >= 0 && v <= 255) { r.push_back('C'); r.push_back((uint8_t)v); }
  else if (v >= 0 && v <= 65535) { r.push_back('S'); put_le16(r, (uint16_t)v); }
  else if (v >= -32768 && v <= 32767) { r.push_back('s'); put_le16(r, (uint16_t)(int16_t)v); }
  else { r.push_back('i'); put_le32(r, (uint32_t)v); }
}
inline void tag_float(std::vector<uint8_t>& r, const char* tag, float v) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('f');
  uint32_t u; memcpy(&u, &v, 4); put_le32(r, u);
}
inline void tag_i16_array(std::vector<uint8_t>& r, const char* tag, const uint16_t* v, uint32_t n) {   // values already <= 32767
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('B'); r.push_back('s');
  put_le32(r, n);
  for (uint32_t i = 0; i < n; i++) put_le16(r, v[i] < 32767 ? v[i] : 32767);
}
inline void tag_u8_array(std::vector<uint8_t>& r, const char* tag, const uint8_t* v, uint32_t n) {   // B:C (raw-bam/tags.rs:733-743)
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('B'); r.push_back('C');
  put_le32(r, n);
  r.insert(r.end(), v, v + n);
}
inline void tag_count_array(std::vector<uint8_t>& r, const char* tag, const uint32_t* v, uint32_t n) {   // B:s of u32 counts clamped to i16::MAX (methylation.rs:50-62)
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('B'); r.push_back('s');
  put_le32(r, n);
  for (uint32_t i = 0; i < n; i++) put_le16(r, (uint16_t)(v[i] < 32767u ? v[i] : 32767u));
}
inline void tag_phred33(std::vector<uint8_t>& r, const char* tag, const uint8_t* q, uint32_t n) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('Z');
  for (uint32_t i = 0; i < n; i++) { unsigned v = (unsigned)q[i] + 33; r.push_back((uint8_t)(v > 255 ? 255 : v)); }
  r.push_back(0);
}
// cD, cM, cE, [cd, ce] (vanilla_caller.rs:1800-1824)
inline void append_depth_error_tags(std::vector<uint8_t>& r, const uint16_t* depths, const uint16_t* errors, uint32_t n, bool per_base) {
  int32_t maxd = 0, mind = 0;
  uint64_t te = 0, td = 0;
  if (n) { maxd = depths[0]; mind = depths[0]; }
  for (uint32_t i = 0; i < n; i++) { maxd = std::max<int32_t>(maxd, depths[i]); mind = std::min<int32_t>(mind, depths[i]); te += errors[i]; td += depths[i]; }
  float er = td > 0 ? (float)te / (float)td : 0.0f;
  tag_int(r, "cD", maxd);
  tag_int(r, "cM", mind);
  tag_float(r, "cE", er);
  if (per_base) { tag_i16_array(r, "cd", depths, n); tag_i16_array(r, "ce", errors, n); }
}

// consensus_umis (simple_umi.rs:46-117, 236-245): per-character call at (Q90, Q90) with Q20 observations.
// Tiny host-side use of the same column code the kernels run.  Returns false where the reference panics.
inline bool consensus_umis(const ConsensusTables& T9090, const std::vector<std::string>& umis, std::string& out) {
  out.clear();
  if (umis.empty()) return true;
  if (umis.size() == 1) { out = umis[0]; return true; }
  size_t L = umis[0].size();
  for (auto& s : umis) if (s.size() != L) return false;
  auto is_dna = [](uint8_t c) { uint8_t u = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; return u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N'; };
  for (size_t i = 0; i < L; i++) {
    ColumnAcc acc;
    acc.reset();
    size_t non_dna = 0;
    uint8_t fc = (uint8_t)umis[0][i];
    for (auto& s : umis) {
      uint8_t ch = (uint8_t)s[i];
      if (is_dna(ch)) { int lane = bam::ascii_to_lane(ch); if (lane != 255) acc.add(lane, T9090.correct[20], T9090.error_per_alt[20]); }
      else { non_dna++; if (ch != fc) return false; }
    }
    if (non_dna == 0) { int bi; uint8_t q; column_call(T9090, acc.s, acc.obs, &bi, &q); out.push_back(bi >= 0 ? "ACGT"[bi] : 'N'); }
    else if (non_dna == umis.size()) out.push_back((char)fc);
    else return false;
  }
  return true;
}

}  // namespace fgx
