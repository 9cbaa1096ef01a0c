// codec_host.cpp — CODEC consensus, general path: host orchestration of one batch of MI groups (pairing,
// virtual overlap clip, alignment filter, overlap geometry), the two single-strand consensus calls per
// molecule on the device (k_column_jobs), then the integer strand combine and BAM record assembly.
//
// Mirrors, in batch form:
//   src/lib/commands/codec.rs:722-790                        process_fn (duplex-disagreement errors are recoverable)
//   crates/fgumi-consensus/src/codec_caller.rs:625-1004      consensus_reads_raw, 1006-1040 build_clipped_info,
//       1096-1113 per-strand cap, 1130-1174 alignment filter, 1200-1262 phase check / consensus length,
//       503-570 to_source_read_for_codec_raw, 1272-1314 pad, 1331-1512 strand combine, 1526-1561 quality
//       masking, 1590-1757 record emission, 1767-1834 reject mask
//   crates/fgumi-raw-bam/src/cigar.rs:404-500, 669-922       virtual hard clip, read_pos_at_ref_pos
// The per-position likelihood arithmetic (ss_caller.consensus_call → create_consensus_from_source_reads) never
// runs on the host: both strands' read sets are staged and called by the HIP kernel.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <unordered_map>
#include "bamrec.h"
#include "engine.h"
#include "host_common.h"

namespace fgx {

using bam::Rec;

namespace {

inline bool consumes_read(uint32_t t) { return t == 0 || t == 1 || t == 7 || t == 8; }
inline uint32_t enc(uint32_t t, uint64_t len) { return ((uint32_t)len << 4) | t; }
using Ops = std::vector<uint32_t>;

int32_t ref_len_wrapping(const Ops& ops) {   // reference_length_from_cigar cigar.rs:137-150
  uint32_t n = 0;
  for (uint32_t op : ops) if (bam::op_consumes_ref(op & 0xF)) n += op >> 4;
  return (int32_t)n;
}

// Leading (or trailing) run of H then S: hard / soft totals and how many ops it spans.
void edge_clips(const Ops& ops, bool from_start, uint64_t& hard, uint64_t& soft, size_t& skip) {
  hard = soft = 0; skip = 0;
  size_t n = ops.size();
  auto at = [&](size_t k) { return from_start ? ops[k] : ops[n - 1 - k]; };
  while (skip < n && (at(skip) & 0xF) == 5) { hard += at(skip) >> 4; skip++; }
  while (skip < n && (at(skip) & 0xF) == 4) { soft += at(skip) >> 4; skip++; }
}

// clip_cigar_ops_raw (cigar.rs:404-446) with its three helpers (:669-922): hard-clip `clip` query bases off one end.
Ops clip_cigar(const Ops& ops, uint64_t clip, bool from_start, uint64_t& ref_consumed) {
  ref_consumed = 0;
  if (clip == 0 || ops.empty()) return ops;
  size_t n = ops.size();
  uint64_t existing = 0;
  for (size_t k = 0; k < n; k++) { uint32_t op = from_start ? ops[k] : ops[n - 1 - k]; uint32_t t = op & 0xF; if (t != 4 && t != 5) break; existing += op >> 4; }
  uint64_t hard, soft; size_t skip;
  edge_clips(ops, from_start, hard, soft, skip);
  Ops out;
  if (clip <= existing) {   // upgrade_clipping_raw: soft → hard, alignment untouched
    uint64_t up = std::min<uint64_t>(soft, clip > hard ? clip - hard : 0);
    if (from_start) {
      out.push_back(enc(5, hard + up));
      if (soft - up) out.push_back(enc(4, soft - up));
      out.insert(out.end(), ops.begin() + skip, ops.end());
    } else {
      out.assign(ops.begin(), ops.begin() + (n - skip));
      if (soft - up) out.push_back(enc(4, soft - up));
      out.push_back(enc(5, hard + up));
    }
    return out;
  }
  uint64_t want = clip - existing, got = 0;
  Ops kept;   // ops that survive next to the clip (in walking order)
  size_t lo = from_start ? skip : 0, hi = from_start ? n : n - skip;   // the unclipped middle [lo, hi)
  size_t taken = 0;   // ops consumed from the clipped end
  while (lo + taken < hi) {
    uint32_t op = from_start ? ops[lo + taken] : ops[hi - 1 - taken];
    uint32_t t = op & 0xF;
    uint64_t len = op >> 4;
    if (got == want && kept.empty() && t == 2) { if (from_start) ref_consumed += len; taken++; continue; }   // deletion at the boundary
    if (got >= want) break;
    bool is_read = consumes_read(t), is_ref = bam::op_consumes_ref(t);
    if (is_read && len > want - got) {
      if (t == 1) got += len;   // an insertion at the boundary goes whole
      else {
        uint64_t part = want - got;
        got += part;
        if (is_ref && from_start) ref_consumed += part;
        kept.push_back(enc(t, len - part));
      }
    } else {
      if (is_read) got += len;
      if (is_ref && from_start) ref_consumed += len;
    }
    taken++;
  }
  uint64_t total_hard = hard + soft + got;
  if (from_start) {
    out.push_back(enc(5, total_hard));
    out.insert(out.end(), kept.begin(), kept.end());
    out.insert(out.end(), ops.begin() + lo + taken, ops.end());
  } else {
    out.assign(ops.begin(), ops.begin() + (hi - taken));
    out.insert(out.end(), kept.rbegin(), kept.rend());
    out.push_back(enc(5, total_hard));
  }
  return out;
}

// read_pos_at_ref_pos_raw (cigar.rs:461-500): 1-based query position at a 1-based reference position.
bool read_pos_at(const Ops& ops, uint64_t aln_start, uint64_t ref_pos, bool last_if_deleted, uint64_t& out) {
  if (ref_pos < aln_start) return false;
  uint64_t ref_off = 0, q_off = 0;
  for (uint32_t op : ops) {
    uint32_t t = op & 0xF;
    uint64_t len = op >> 4;
    if (bam::op_consumes_ref(t)) {
      uint64_t s = aln_start + ref_off, e = s + len - 1;
      if (ref_pos >= s && ref_pos <= e) {
        if (bam::op_consumes_query(t)) { out = q_off + (ref_pos - s) + 1; return true; }
        if (last_if_deleted) { out = q_off > 0 ? q_off : 1; return true; }
        return false;
      }
      ref_off += len;
    }
    if (bam::op_consumes_query(t)) q_off += len;
  }
  return false;
}

SimpCigar simplify_ops(const Ops& ops) {   // noodles_compat.rs:10-55
  SimpCigar out;
  for (uint32_t raw : ops) {
    uint32_t t = raw & 0xF;
    if (t > 8) continue;
    uint8_t k = (t == 4 || t == 5 || t == 7 || t == 8) ? 0 : (uint8_t)t;
    if (!out.empty() && out.back().first == k) out.back().second += raw >> 4;
    else out.push_back({k, raw >> 4});
  }
  return out;
}

struct Info {   // ClippedRecordInfo (codec_caller.rs:323-336)
  uint32_t raw_idx;
  uint64_t clip;
  bool from_start;
  uint64_t seq_len;
  Ops cigar;
  uint64_t adj_pos;
  uint16_t flags;
};

struct Strand {   // SingleStrandConsensus, the fields with an observable effect
  std::vector<uint8_t> b, q;
  std::vector<uint16_t> d, e;
};

inline uint8_t comp_ascii(uint8_t c) {
  if (c == 'n') return 'n';   // padding stays lower case; a lone read can put any IUPAC letter into its strand consensus
  return bam::code_to_ascii(bam::code_complement(bam::ascii_to_code(c)));
}
Strand rc(const Strand& s) {
  Strand o;
  o.b.assign(s.b.rbegin(), s.b.rend());
  for (auto& c : o.b) c = comp_ascii(c);
  o.q.assign(s.q.rbegin(), s.q.rend()); o.d.assign(s.d.rbegin(), s.d.rend()); o.e.assign(s.e.rbegin(), s.e.rend());
  return o;
}
Strand pad(const Strand& s, uint64_t L, bool left) {   // :1272-1314 (lower-case 'n', quality 0)
  if (L <= s.b.size()) return s;
  size_t n = (size_t)L - s.b.size();
  Strand o;
  o.b.reserve(L); o.q.reserve(L); o.d.reserve(L); o.e.reserve(L);
  if (left) { o.b.assign(n, 'n'); o.q.assign(n, 0); o.d.assign(n, 0); o.e.assign(n, 0); }
  o.b.insert(o.b.end(), s.b.begin(), s.b.end()); o.q.insert(o.q.end(), s.q.begin(), s.q.end());
  o.d.insert(o.d.end(), s.d.begin(), s.d.end()); o.e.insert(o.e.end(), s.e.begin(), s.e.end());
  if (!left) { o.b.insert(o.b.end(), n, 'n'); o.q.insert(o.q.end(), n, 0); o.d.insert(o.d.end(), n, 0); o.e.insert(o.e.end(), n, 0); }
  return o;
}
inline int32_t cap_short(uint16_t v) { return v > 32767 ? 32767 : v; }
inline uint16_t cap_err(int64_t v) { return (uint16_t)(v < 0 ? 0 : v > 32767 ? 32767 : v); }

struct Group {
  bool pending = false;          // two column jobs staged, decided after the device pass
  bool has_umi = false;
  std::string umi;
  HostStats st;
  std::vector<uint8_t> mask;     // reject mask over the group's records (track_rejects)
  std::vector<uint32_t> strand_idx;   // r1 infos ++ r2 infos raw indices (rejects, CB source order)
  uint64_t cons_len = 0;
  bool r1_neg = false, r2_neg = false;
  int64_t job1 = -1, job2 = -1;
};

}  // namespace

int codec_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                          const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  (void)n_rec;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  const fgx_options& o = c->opt;
  const bool track = o.track_rejects != 0;
  const bool has_max = o.codec_max_reads_per_strand >= 0;
  const uint64_t max_reads = has_max ? (uint64_t)o.codec_max_reads_per_strand : 0;
  const uint64_t max_dis = o.codec_max_duplex_disagreements == 0xFFFFFFFFu ? UINT64_MAX : o.codec_max_duplex_disagreements;
  ColumnBatch& B = c->batch;
  B.clear();
  c->out_data.clear(); c->out_rejects.clear();
  c->grp_out_end.assign(n_grp, 0);
  c->counter_names_used = false;
  c->err.clear();
  std::vector<Group> groups(n_grp);
  std::vector<uint8_t> tb, tq;

  for (uint32_t g = 0; g < n_grp; g++) {
    Group& G = groups[g];
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    auto rec = [&](uint32_t i) { return Rec{blob + rec_off[r0 + i], rec_len[r0 + i]}; };
    G.st.total_reads += n;
    if (track) G.mask.assign(n, 0);
    if (n == 0) continue;
    auto reject_at = [&](const std::vector<uint32_t>& idx, int reason) {
      if (track) for (uint32_t i : idx) G.mask[i] = 1;
      G.st.reject(reason, idx.size());
    };
    {
      Rec v = rec(0);
      uint32_t an = v.len > v.aux_off() ? v.len - v.aux_off() : 0, vl = 0;
      int64_t off = bam::find_z_tag(v.b + v.aux_off(), an, 'M', 'I', &vl);
      if (off >= 0) { G.has_umi = true; G.umi.assign((const char*)v.b + v.aux_off() + off, vl); }
    }
    // phase 1: paired primaries only
    std::vector<uint32_t> paired, frags;
    for (uint32_t i = 0; i < n; i++) {
      uint16_t f = rec(i).flags();
      if (!(f & bam::F_PAIRED)) { frags.push_back(i); continue; }
      if (f & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) continue;
      paired.push_back(i);
    }
    if (!frags.empty()) reject_at(frags, FGX_REJ_FRAGMENT_READ);
    if (paired.empty()) continue;
    // phase 2: templates in first-appearance order; exactly one primary FR pair each
    std::vector<std::vector<uint32_t>> templates;
    {
      std::unordered_map<std::string, uint32_t> by_name;
      for (uint32_t i : paired) {
        Rec v = rec(i);
        std::string nm((const char*)v.name(), v.name_len());
        auto it = by_name.find(nm);
        if (it == by_name.end()) { by_name.emplace(std::move(nm), (uint32_t)templates.size()); templates.push_back({i}); }
        else templates[it->second].push_back(i);
      }
    }
    std::vector<Info> r1s, r2s;
    auto make_info = [&](uint32_t i, uint64_t clip) {
      Rec v = rec(i);
      Info ci;
      ci.raw_idx = i; ci.clip = clip; ci.flags = v.flags(); ci.from_start = ci.flags & bam::F_REVERSE;
      uint64_t ref_consumed = 0;
      ci.cigar = clip_cigar(cigar_ops_vec(v), clip, ci.from_start, ref_consumed);
      uint64_t l = v.l_seq();
      ci.seq_len = l > clip ? l - clip : 0;
      uint64_t p1 = (uint64_t)(int64_t)(v.pos() + 1);
      ci.adj_pos = ci.from_start ? p1 + ref_consumed : p1;
      return ci;
    };
    for (auto& idx : templates) {
      bool fr = idx.size() == 2 && is_primary_fr_pair_raw(rec(idx[0]), rec(idx[1]));
      if (!fr) { reject_at(idx, FGX_REJ_NOT_PRIMARY_FR_PAIR); continue; }
      uint32_t i1 = idx[0], i2 = idx[1];
      if (!(rec(idx[0]).flags() & bam::F_FIRST)) std::swap(i1, i2);
      uint64_t c1 = mate_clip_vs_mate_raw(rec(i1), rec(i2)), c2 = mate_clip_vs_mate_raw(rec(i2), rec(i1));
      r1s.push_back(make_info(i1, c1));
      r2s.push_back(make_info(i2, c2));
    }
    if (r1s.empty()) continue;
    auto all_idx = [&]() { std::vector<uint32_t> v; for (auto& i : r1s) v.push_back(i.raw_idx); for (auto& i : r2s) v.push_back(i.raw_idx); return v; };
    if (r1s.size() < o.codec_min_reads_per_strand) { reject_at(all_idx(), FGX_REJ_INSUFFICIENT_READS); continue; }
    // phase 3: most common alignment per strand
    auto filter = [&](std::vector<Info>& infos) {
      if (infos.size() < 2) return;
      std::vector<SimpCigar> cig(infos.size());
      std::vector<uint32_t> order(infos.size());
      for (uint32_t i = 0; i < infos.size(); i++) {
        cig[i] = simplify_ops(infos[i].cigar);
        if (infos[i].flags & bam::F_REVERSE) std::reverse(cig[i].begin(), cig[i].end());
        order[i] = i;
      }
      std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return infos[a].seq_len > infos[b].seq_len; });
      std::vector<const SimpCigar*> sorted;
      for (uint32_t i : order) sorted.push_back(&cig[i]);
      std::vector<uint8_t> keep(infos.size(), 0);
      for (uint32_t k : most_common_alignment_group(sorted)) keep[order[k]] = 1;
      std::vector<uint32_t> rej;
      for (uint32_t i = 0; i < infos.size(); i++) if (!keep[i]) rej.push_back(infos[i].raw_idx);
      if (!rej.empty()) reject_at(rej, FGX_REJ_MINORITY_ALIGNMENT);
      std::vector<Info> kept;
      for (uint32_t i = 0; i < infos.size(); i++) if (keep[i]) kept.push_back(std::move(infos[i]));
      infos.swap(kept);
    };
    filter(r1s); filter(r2s);
    if (r1s.empty() || r2s.empty()) continue;
    if (r1s.size() < o.codec_min_reads_per_strand || r2s.size() < o.codec_min_reads_per_strand) { reject_at(all_idx(), FGX_REJ_INSUFFICIENT_READS); continue; }
    if (has_max) {
      if (max_reads == 0) { reject_at(all_idx(), FGX_REJ_INSUFFICIENT_READS); continue; }
      auto cap = [&](std::vector<Info>& infos) -> uint64_t {
        if (infos.size() <= max_reads) return 0;
        uint64_t dropped = infos.size() - max_reads;
        std::vector<int32_t> ranks;
        for (auto& i : infos) { Rec v = rec(i.raw_idx); ranks.push_back(read_name_rank(v.name(), v.name_len())); }
        std::vector<Info> kept;
        for (uint32_t k : lowest_ranking(ranks, (size_t)max_reads)) kept.push_back(std::move(infos[k]));
        infos.swap(kept);
        return dropped;
      };
      uint64_t d = cap(r1s) + cap(r2s);
      if (d) G.st.reject(FGX_REJ_DOWNSAMPLED, d);   // counted, not routed to the rejects output
    }
    // phase 4: overlap geometry from the longest alignment of each strand (first maximum)
    auto longest = [](const std::vector<Info>& v) -> const Info& {
      size_t best = 0; int32_t bl = ref_len_wrapping(v[0].cigar);
      for (size_t i = 1; i < v.size(); i++) { int32_t l = ref_len_wrapping(v[i].cigar); if (l > bl) { bl = l; best = i; } }
      return v[best];
    };
    const Info& l1 = longest(r1s);
    const Info& l2 = longest(r2s);
    G.r1_neg = l1.flags & bam::F_REVERSE;
    G.r2_neg = l2.flags & bam::F_REVERSE;
    const Info& lpos = G.r1_neg ? l2 : l1;
    const Info& lneg = G.r1_neg ? l1 : l2;
    uint64_t pos_ref = (uint64_t)(int64_t)ref_len_wrapping(lpos.cigar), neg_ref = (uint64_t)(int64_t)ref_len_wrapping(lneg.cigar);
    uint64_t pos_end = lpos.adj_pos + (pos_ref ? pos_ref - 1 : 0), neg_end = lneg.adj_pos + (neg_ref ? neg_ref - 1 : 0);
    uint64_t ov_s = std::max(lneg.adj_pos, lpos.adj_pos), ov_e = std::min(pos_end, neg_end);
    int64_t duplex_len = (int64_t)ov_e - (int64_t)ov_s + 1;
    if (duplex_len < (int64_t)o.codec_min_duplex_length) { reject_at(all_idx(), FGX_REJ_INSUFFICIENT_OVERLAP); continue; }
    auto at0 = [](const Info& r, uint64_t p) -> int64_t { uint64_t q; return read_pos_at(r.cigar, r.adj_pos, p, true, q) ? (int64_t)q : 0; };
    if ((at0(l1, ov_s) - at0(l2, ov_s)) != (at0(l1, ov_e) - at0(l2, ov_e))) { reject_at(all_idx(), FGX_REJ_INDEL_ERROR_BETWEEN_STRANDS); continue; }
    uint64_t prp, nrp;
    if (!read_pos_at(lpos.cigar, lpos.adj_pos, ov_e, false, prp) || !read_pos_at(lneg.cigar, lneg.adj_pos, ov_e, false, nrp)) {
      reject_at(all_idx(), FGX_REJ_INDEL_ERROR_BETWEEN_STRANDS); continue;
    }
    if (prp + lneg.seq_len < nrp) { c->err = "codec consensus length underflow"; return 2; }
    G.cons_len = prp + lneg.seq_len - nrp;
    // phase 5: stage the source reads of both strands (clip, orient; no quality masking) as two column jobs
    auto stage = [&](const std::vector<Info>& infos) -> int64_t {
      uint32_t rd0 = (uint32_t)B.reads.size(), longest_len = 0;
      for (auto& ci : infos) {
        Rec v = rec(ci.raw_idx);
        uint32_t l = v.l_seq();
        uint32_t clip = (uint32_t)std::min<uint64_t>(ci.clip, l), keep = l - clip, first = ci.from_start ? clip : 0;
        const uint8_t* q = v.b + v.qual_off();
        tb.resize(keep); tq.resize(keep);
        if (ci.flags & bam::F_REVERSE)
          for (uint32_t i = 0; i < keep; i++) { uint32_t s = first + keep - 1 - i; tb[i] = bam::code_to_ascii(bam::code_complement(v.base_code(s))); tq[i] = q[s]; }
        else
          for (uint32_t i = 0; i < keep; i++) { tb[i] = bam::code_to_ascii(v.base_code(first + i)); tq[i] = q[first + i]; }
        B.add_read(tb.data(), tq.data(), keep);
        longest_len = std::max(longest_len, keep);
      }
      return (int64_t)B.add_job(rd0, (uint32_t)infos.size(), longest_len);
    };
    G.job1 = stage(r1s);
    G.job2 = stage(r2s);
    G.strand_idx = all_idx();
    G.pending = true;
  }
  auto t1 = clk::now();

  // ss_caller settings (codec_caller.rs:374-397): min_reads 1, no cap, min consensus base quality 0
  double ms_k = c->run_columns(B, ColParams{1, 0});
  auto t2 = clk::now();

  HostStats batch;
  uint64_t extra[4] = {0, 0, 0, 0};   // consensus_bases_emitted, duplex bases, disagreement bases, rejected_hdd
  uint64_t n_rejects = 0, count = 0, counter = 0;
  for (uint32_t g = 0; g < n_grp; g++) {
    Group& G = groups[g];
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    auto reject_all = [&](int reason) {
      if (track) for (uint32_t i : G.strand_idx) G.mask[i] = 1;
      G.st.reject(reason, G.strand_idx.size());
    };
    if (G.pending) {
      auto load = [&](int64_t job, Strand& s) {
        const ColJob& j = B.jobs[(size_t)job];
        s.b.assign(B.ob.begin() + j.out_off, B.ob.begin() + j.out_off + j.cons_len); s.q.assign(B.oq.begin() + j.out_off, B.oq.begin() + j.out_off + j.cons_len);
        s.d.assign(B.od.begin() + j.out_off, B.od.begin() + j.out_off + j.cons_len); s.e.assign(B.oe.begin() + j.out_off, B.oe.begin() + j.out_off + j.cons_len);
      };
      Strand s1, s2;
      load(G.job1, s1); load(G.job2, s2);
      if (G.cons_len < s1.b.size() || G.cons_len < s2.b.size()) reject_all(FGX_REJ_CLIP_OVERLAP_FAILED);
      else {
        Strand p1 = pad(G.r1_neg ? rc(s1) : s1, G.cons_len, G.r1_neg), p2 = pad(G.r1_neg ? s2 : rc(s2), G.cons_len, G.r2_neg);
        // build_duplex_consensus_from_padded (:1331-1512)
        size_t L = p1.b.size();
        Strand cs;
        cs.b.assign(L, 'N'); cs.q.assign(L, FGX_MIN_PHRED); cs.d.assign(L, 0); cs.e.assign(L, 0);
        uint64_t dis = 0, dup = 0;
        for (size_t i = 0; i < L; i++) {
          uint8_t ba = p1.b[i], qa = p1.q[i], bb = p2.b[i], qb = p2.q[i];
          uint16_t da = p1.d[i], ea = p1.e[i], db = p2.d[i], eb = p2.e[i];
          bool ha = ba != 'N' && ba != 'n', hb = bb != 'N' && bb != 'n';
          uint8_t fb, fq; uint16_t depth, error;
          if (ha && hb) {
            dup++;
            uint8_t rb, rq;
            if (ba == bb) { rb = ba; rq = (uint8_t)std::min<uint32_t>(93, (uint32_t)qa + qb); }
            else if (qa > qb) { dis++; rb = ba; rq = std::max<uint8_t>(FGX_MIN_PHRED, (uint8_t)(qa - qb)); }
            else if (qb > qa) { dis++; rb = bb; rq = std::max<uint8_t>(FGX_MIN_PHRED, (uint8_t)(qb - qa)); }
            else { dis++; rb = ba; rq = FGX_MIN_PHRED; }
            if (rq == FGX_MIN_PHRED) { fb = 'N'; fq = FGX_MIN_PHRED; } else { fb = rb; fq = rq; }
            int64_t de = ba == bb ? (int64_t)ea + eb : ba == rb ? (int64_t)ea + (db > eb ? db - eb : 0) : (int64_t)eb + (da > ea ? da - ea : 0);
            error = cap_err(de);
            depth = (uint16_t)(cap_short(da) + cap_short(db));
          } else if (ha) { if (qa == FGX_MIN_PHRED) { fb = 'N'; fq = FGX_MIN_PHRED; } else { fb = ba; fq = qa; } depth = da; error = ea; }
          else if (hb) { if (qb == FGX_MIN_PHRED) { fb = 'N'; fq = FGX_MIN_PHRED; } else { fb = bb; fq = qb; } depth = db; error = eb; }
          else { fb = 'N'; fq = FGX_MIN_PHRED; depth = 0; error = cap_err((int64_t)ea + eb); }
          if (ba == 'N' || bb == 'N') { fb = 'N'; fq = FGX_MIN_PHRED; }
          cs.b[i] = fb; cs.q[i] = fq; cs.d[i] = depth; cs.e[i] = error;
        }
        bool hdd = false;
        if (dup > 0) {
          double rate = (double)dis / (double)dup;
          if (dis > max_dis || rate > o.codec_max_duplex_disagreement_rate) hdd = true;
        }
        if (hdd) { reject_all(FGX_REJ_HIGH_DUPLEX_DISAGREEMENT); extra[3] += 1; }
        else {
          // mask_consensus_quals_query_based (:1526-1561): outer bases first, then single-strand stretches
          if (o.codec_outer_bases_length > 0 && o.codec_has_outer_bases_qual) {
            size_t last = L ? L - 1 : 0;
            for (size_t i = 0; i < std::min<size_t>(o.codec_outer_bases_length, L); i++) { cs.q[i] = o.codec_outer_bases_qual; cs.q[last - i] = o.codec_outer_bases_qual; }
          }
          if (o.codec_has_single_strand_qual)
            for (size_t i = 0; i < L; i++) if (p1.b[i] == 'N' || p1.b[i] == 'n' || p2.b[i] == 'N' || p2.b[i] == 'n') cs.q[i] = o.codec_single_strand_qual;
          if (G.r1_neg) { cs = rc(cs); p1 = rc(p1); p2 = rc(p2); }
          // build_output_record_into (:1590-1757)
          counter++;
          std::string name = G.has_umi ? c->prefix + ":" + G.umi : c->prefix + ":" + std::to_string(counter);
          if (!G.has_umi) c->counter_names_used = true;
          std::vector<uint8_t> r;
          if (!build_unmapped_record(r, name, bam::F_UNMAPPED, cs.b.data(), cs.q.data(), (uint32_t)L)) {
            c->err = "could not write the consensus record for read '" + name + "': read name too long";
            return 2;
          }
          tag_z(r, "RG", c->rg.data(), c->rg.size());
          if (G.has_umi) tag_z(r, "MI", G.umi.data(), G.umi.size());
          {
            int32_t mx = 0, mn = 0; uint64_t te = 0, tbs = 0;
            for (size_t i = 0; i < L; i++) { int32_t t = cap_short(p1.d[i]) + cap_short(p2.d[i]); if (i == 0) mx = mn = t; mx = std::max(mx, t); mn = std::min(mn, t); tbs += (uint64_t)t; }
            for (auto e : cs.e) te += (uint64_t)cap_short(e);
            tag_int(r, "cD", mx); tag_int(r, "cM", mn); tag_float(r, "cE", tbs ? (float)te / (float)tbs : 0.0f);
          }
          auto strand_tags = [&](const char* td, const char* tm, const char* te_, const Strand& s) {
            int32_t mx = 0, mn = 0; uint64_t te = 0, tbs = 0;
            for (size_t i = 0; i < s.d.size(); i++) { int32_t t = cap_short(s.d[i]); if (i == 0) mx = mn = t; mx = std::max(mx, t); mn = std::min(mn, t); tbs += (uint64_t)t; }
            for (auto e : s.e) te += (uint64_t)cap_short(e);
            tag_int(r, td, mx); tag_int(r, tm, mn); tag_float(r, te_, tbs ? (float)te / (float)tbs : 0.0f);
          };
          strand_tags("aD", "aM", "aE", p1);
          strand_tags("bD", "bM", "bE", p2);
          if (o.produce_per_base_tags) {
            tag_i16_array(r, "ad", p1.d.data(), (uint32_t)L); tag_i16_array(r, "bd", p2.d.data(), (uint32_t)L);
            tag_i16_array(r, "ae", p1.e.data(), (uint32_t)L); tag_i16_array(r, "be", p2.e.data(), (uint32_t)L);
            tag_z(r, "ac", (const char*)p1.b.data(), L); tag_z(r, "bc", (const char*)p2.b.data(), L);
            tag_phred33(r, "aq", p1.q.data(), (uint32_t)L); tag_phred33(r, "bq", p2.q.data(), (uint32_t)L);
          }
          auto find = [&](uint32_t i, char t0, char t1, const char** v, uint32_t* vl) {
            Rec rv{blob + rec_off[r0 + i], rec_len[r0 + i]};
            uint32_t an = rv.len > rv.aux_off() ? rv.len - rv.aux_off() : 0;
            int64_t off = bam::find_z_tag(rv.b + rv.aux_off(), an, (uint8_t)t0, (uint8_t)t1, vl);
            if (off < 0) return false;
            *v = (const char*)rv.b + rv.aux_off() + off;
            return true;
          };
          if (o.cell_tag[0])
            for (uint32_t i : G.strand_idx) { const char* v; uint32_t vl; if (find(i, o.cell_tag[0], o.cell_tag[1], &v, &vl) && vl > 0) { tag_z(r, o.cell_tag, v, vl); break; } }
          std::vector<std::string> umis;
          for (uint32_t i = 0; i < n; i++) { const char* v; uint32_t vl; if (find(i, 'R', 'X', &v, &vl)) umis.emplace_back(v, vl); }
          if (!umis.empty()) {
            std::string cu;
            if (!consensus_umis(c->h_umi_tables.t, umis, cu)) { c->err = "consensus_umis: UMIs of unequal length or mixed DNA/non-DNA characters"; return 2; }
            if (!cu.empty()) tag_z(r, "RX", cu.data(), cu.size());
          }
          append_with_block_size(c->out_data, r.data(), (uint32_t)r.size());
          count++;
          G.st.consensus_reads += 1;
          extra[0] += L; extra[1] += dup; extra[2] += dis;
        }
      }
    }
    batch.total_reads += G.st.total_reads; batch.consensus_reads += G.st.consensus_reads; batch.filtered_reads += G.st.filtered_reads;
    for (int i = 0; i < FGX_N_REJECTION; i++) batch.rej[i] += G.st.rej[i];
    if (track)
      for (uint32_t i = 0; i < n; i++) if (G.mask[i]) { append_with_block_size(c->out_rejects, blob + rec_off[r0 + i], rec_len[r0 + i]); n_rejects++; }
    c->grp_out_end[g] = c->out_data.size();
  }
  auto t3 = clk::now();

  memset(out, 0, sizeof(*out));
  out->data = c->out_data.data(); out->data_len = c->out_data.size(); out->count = count;
  batch.to_array(out->stats);
  for (int i = 0; i < 4; i++) out->stats[24 + i] = extra[i];
  out->rejects = c->out_rejects.data(); out->rejects_len = c->out_rejects.size(); out->n_rejects = n_rejects;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  out->ms_host_prep = ms(t0, t1); out->ms_kernels = ms_k; out->ms_h2d = ms(t1, t2) - ms_k; out->ms_emit = ms(t2, t3);
  return 0;
}

}  // namespace fgx
