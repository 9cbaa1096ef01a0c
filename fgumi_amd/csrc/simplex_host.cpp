// simplex_host.cpp — GENERAL simplex path: host orchestration of one batch of MI groups, the
// per-position arithmetic on the device (k_column_jobs), BAM record assembly on the host.
//
// This is the path that accepts ANY input the reference accepts (indel CIGARs, unmapped reads,
// fragments, --max-reads, --trim …).  It mirrors, in batch form:
//   src/lib/commands/simplex.rs:637-718          process_fn (min-reads skip, overlap pre-step, rejects)
//   crates/fgumi-consensus/src/overlapping.rs:236-336, 627-684   R1/R2 overlap pre-correction
//   crates/fgumi-consensus/src/vanilla_caller.rs:1329-1422 process_group, 1454-1646 process_subgroup,
//       1080-1190 create_source_read, 1217-1296 alignment filter, 902-932 downsampling,
//       1767-1881 build_consensus_record_into
// The column arithmetic itself (vanilla_caller.rs:1652-1755 + base_builder.rs) never runs here:
// source reads are staged and every column is called by the HIP kernel.
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

struct SrcRead {
  uint32_t orig_idx;   // index into the subgroup's read list
  uint32_t rd;         // ReadDesc index in the batch
  uint32_t len;
  uint16_t flags;
  int32_t name_hash;
  SimpCigar cigar;
  int32_t ref_id = -1;         // methylation-aware mode: reference id, 0-based start and the simplified CIGAR before reversal /
  int64_t aln_start = -1;      // truncation (vanilla_caller.rs:1176-1190)
  SimpCigar orig_cigar;
};

enum ReadType { RT_FRAGMENT = 0, RT_R1 = 1, RT_R2 = 2 };

struct PendingRecord {   // one consensus record to assemble once the device returns the columns
  uint32_t job;
  uint8_t read_type;
  std::string umi;
  const uint8_t* first_raw; uint32_t first_len;    // for the cell-barcode tag
  std::vector<std::string> rx;                      // RX of every retained read
  int64_t meth_job = -1;                            // annotation job of the methylation-aware mode (-1: no annotation)
  bool meth_top = true;                             // is_top_strand of the FIRST retained read (vanilla_caller.rs:1855-1858)
};

struct Positioned { uint32_t pos; const uint8_t* p; uint32_t n; };

struct Ctx {
  fgx_caller* c;
  const fgx_options& o;
  HostStats stats;
  uint64_t ov[4] = {0, 0, 0, 0};
  std::vector<PendingRecord> pending;
  std::vector<std::pair<uint32_t, std::vector<uint8_t>>> group_rejects;
  std::vector<uint8_t>* rejects_out;
  uint64_t n_rejects = 0;
  std::string err;
  Ctx(fgx_caller* cc) : c(cc), o(cc->opt), rejects_out(&cc->out_rejects) {}

  void reject_now(const uint8_t* p, uint32_t n) { append_with_block_size(*rejects_out, p, n); n_rejects++; }
  void reject_pos(uint32_t pos, const uint8_t* p, uint32_t n) { if (o.track_rejects) group_rejects.push_back({pos, std::vector<uint8_t>(p, p + n)}); }
  void flush_group_rejects() {
    if (!o.track_rejects) { group_rejects.clear(); return; }
    std::stable_sort(group_rejects.begin(), group_rejects.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (auto& e : group_rejects) reject_now(e.second.data(), (uint32_t)e.second.size());
    group_rejects.clear();
  }
};

// create_source_read (vanilla_caller.rs:1080-1190): orient, (trim), mask, mate-clip, strip trailing N.
// Stages the transformed read into the batch.  Returns 0 = dropped (zero length), 1 = ok, -1 = fatal.
int create_source_read(Ctx& x, const Positioned& pr, uint32_t idx, uint64_t mate_clip, SrcRead& out, std::vector<uint8_t>& tb,
                       std::vector<uint8_t>& tq) {
  Rec v{pr.p, pr.n};
  uint16_t flg = v.flags();
  bool neg = flg & bam::F_REVERSE;
  uint8_t min_bq = x.o.min_input_base_quality;
  uint32_t read_len = v.l_seq();
  if (read_len == 0) return 0;
  if ((uint64_t)v.qual_off() + read_len > pr.n) { x.err = "input read has invalid base qualities (length does not match sequence length): " + std::string((const char*)v.name(), v.name_len()); return -1; }
  const uint8_t* q = pr.p + v.qual_off();
  bool all_ff = true;
  for (uint32_t i = 0; i < read_len; i++) if (q[i] != 0xFF) { all_ff = false; break; }
  if (all_ff) { x.err = "input read is missing base qualities (BAM QUAL is '*'): " + std::string((const char*)v.name(), v.name_len()); return -1; }
  tb.resize(read_len);
  tq.resize(read_len);
  if (neg) {
    for (uint32_t i = 0; i < read_len; i++) { uint32_t s = read_len - 1 - i; tb[i] = bam::code_to_ascii(bam::code_complement(v.base_code(s))); tq[i] = q[s]; }
  } else {
    for (uint32_t i = 0; i < read_len; i++) { tb[i] = bam::code_to_ascii(v.base_code(i)); tq[i] = q[i]; }
  }
  uint32_t trim_to = x.o.trim ? quality_trim_point(tq.data(), read_len, min_bq) : read_len;
  for (uint32_t i = 0; i < trim_to; i++) if (tq[i] < min_bq) { tb[i] = 'N'; tq[i] = FGX_MIN_PHRED; }
  uint64_t clip_position = read_len > mate_clip ? read_len - mate_clip : 0;
  uint32_t final_len = (uint32_t)std::min<uint64_t>(clip_position, trim_to);
  while (final_len > 0 && tb[final_len - 1] == 'N') final_len--;
  if (final_len == 0) return 0;
  out.orig_idx = idx;
  out.len = final_len;
  out.flags = flg;
  out.rd = x.c->batch.add_read(tb.data(), tq.data(), final_len);
  SimpCigar sc = simplify_cigar(v);
  if (x.o.methylation_mode != FGX_METHYLATION_DISABLED) { out.ref_id = v.ref_id(); out.aln_start = (int64_t)v.pos(); out.orig_cigar = sc; }
  if (neg) std::reverse(sc.begin(), sc.end());
  out.cigar = truncate_cigar(sc, final_len);
  out.name_hash = x.o.max_reads >= 0 ? read_name_rank(v.name(), v.name_len()) : 0;
  return 1;
}

// process_subgroup (vanilla_caller.rs:1454-1646). ok → a PendingRecord was queued.
int process_subgroup(Ctx& x, const std::string& umi, ReadType rt, const std::vector<Positioned>& reads, bool& ok, uint32_t& surviving,
                     std::vector<uint32_t>& surviving_idx, std::vector<PendingRecord>& out_pending) {
  ok = false;
  surviving = 0;
  surviving_idx.clear();
  const uint32_t min_reads = x.o.min_reads;
  if (reads.empty()) return 0;
  if (reads.size() < min_reads) {
    x.stats.reject(FGX_REJ_INSUFFICIENT_READS, reads.size());
    for (auto& r : reads) x.reject_pos(r.pos, r.p, r.n);
    return 0;
  }
  std::vector<SrcRead> srs;
  std::vector<uint32_t> zero_len;
  std::vector<uint8_t> tb, tq;
  for (uint32_t i = 0; i < reads.size(); i++) {
    uint64_t clip = mate_clip_raw(Rec{reads[i].p, reads[i].n});
    SrcRead sr;
    int rc = create_source_read(x, reads[i], i, clip, sr, tb, tq);
    if (rc < 0) return -1;
    if (rc == 1) srs.push_back(std::move(sr)); else zero_len.push_back(i);
  }
  if (!zero_len.empty()) {
    x.stats.reject(FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING, zero_len.size());
    for (uint32_t i : zero_len) x.reject_pos(reads[i].pos, reads[i].p, reads[i].n);
  }
  auto insufficient = [&](std::vector<SrcRead>& v) {
    if (!v.empty()) { x.stats.reject(FGX_REJ_INSUFFICIENT_READS, v.size()); for (auto& s : v) x.reject_pos(reads[s.orig_idx].pos, reads[s.orig_idx].p, reads[s.orig_idx].n); }
  };
  if (srs.size() < min_reads) { insufficient(srs); return 0; }

  // drop_unmapped_if_any_mapped (:1217-1232)
  {
    bool any_un = false, all_un = true;
    for (auto& s : srs) { bool u = s.flags & bam::F_UNMAPPED; any_un |= u; all_un &= u; }
    if (any_un && !all_un) {
      std::vector<SrcRead> kept;
      size_t dropped = 0;
      for (auto& s : srs) { if (s.flags & bam::F_UNMAPPED) { dropped++; x.reject_pos(reads[s.orig_idx].pos, reads[s.orig_idx].p, reads[s.orig_idx].n); } else kept.push_back(std::move(s)); }
      x.stats.reject(FGX_REJ_UNMAPPED, dropped);
      srs.swap(kept);
    }
  }
  // filter_source_reads_by_alignment (:1242-1296)
  if (srs.size() >= 2) {
    std::vector<uint32_t> order(srs.size());
    for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return srs[a].len > srs[b].len; });
    std::vector<const SimpCigar*> cigs;
    for (uint32_t i : order) cigs.push_back(&srs[i].cigar);
    std::vector<uint32_t> keep_sorted = most_common_alignment_group(cigs);   // positions in `order`
    std::vector<bool> keep(srs.size(), false);
    for (uint32_t k : keep_sorted) keep[order[k]] = true;
    size_t n_keep = 0;
    for (bool b : keep) n_keep += b;
    size_t rejected = srs.size() - n_keep;
    if (rejected) {
      x.stats.reject(FGX_REJ_MINORITY_ALIGNMENT, rejected);
      std::vector<SrcRead> kept;
      for (size_t i = 0; i < srs.size(); i++) { if (keep[i]) kept.push_back(std::move(srs[i])); else x.reject_pos(reads[srs[i].orig_idx].pos, reads[srs[i].orig_idx].p, reads[srs[i].orig_idx].n); }
      srs.swap(kept);
    }
  }
  if (srs.size() < min_reads) { insufficient(srs); return 0; }
  // downsample_filtered_source_reads (:902-932)
  if (x.o.max_reads >= 0 && srs.size() > (size_t)x.o.max_reads) {
    std::vector<int32_t> ranks;
    for (auto& s : srs) ranks.push_back(s.name_hash);
    std::vector<uint32_t> keep_idx = lowest_ranking(ranks, (size_t)x.o.max_reads);
    std::vector<bool> keep(srs.size(), false);
    for (uint32_t k : keep_idx) keep[k] = true;
    std::vector<SrcRead> kept;
    size_t dropped = 0;
    for (size_t i = 0; i < srs.size(); i++) { if (keep[i]) kept.push_back(std::move(srs[i])); else { dropped++; x.reject_pos(reads[srs[i].orig_idx].pos, reads[srs[i].orig_idx].p, reads[srs[i].orig_idx].n); } }
    if (dropped) x.stats.reject(FGX_REJ_DOWNSAMPLED, dropped);
    srs.swap(kept);
  }
  if (srs.size() < min_reads) { insufficient(srs); return 0; }

  surviving = (uint32_t)srs.size();
  for (auto& s : srs) surviving_idx.push_back(s.orig_idx);

  // consensus length = min_reads-th longest (:1661-1669); source reads must be contiguous ReadDescs
  // in retained order, so re-stage when filtering dropped some in between.
  std::vector<uint32_t> lens;
  for (auto& s : srs) lens.push_back(s.len);
  std::sort(lens.begin(), lens.end(), [](uint32_t a, uint32_t b) { return a > b; });
  uint32_t cons_len = lens[min_reads - 1];
  ColumnBatch& B = x.c->batch;
  bool contiguous = true;
  for (size_t i = 1; i < srs.size(); i++) if (srs[i].rd != srs[i - 1].rd + 1) { contiguous = false; break; }
  uint32_t rd0 = srs[0].rd;
  if (!contiguous) {
    rd0 = (uint32_t)B.reads.size();
    for (auto& s : srs) B.reads.push_back(B.reads[s.rd]);   // descriptors only; staged bytes are shared
  }
  PendingRecord pr;
  // annotate_and_normalize (vanilla_caller.rs:781-860) of the retained reads: the anchor is the LAST longest read (max_by_key); no
  // annotation without a reference, for an unplaced anchor or a reference id outside the header.  The job runs on the device before
  // the column job and rewrites the staged bases (methylation_core.h).
  if (x.o.methylation_mode != FGX_METHYLATION_DISABLED && x.c->genome) {
    size_t a = 0;
    for (size_t i = 1; i < srs.size(); i++) if (srs[i].len >= srs[a].len) a = i;
    const SrcRead& an = srs[a];
    const GenomeRef& G = *x.c->genome;
    if (an.ref_id >= 0 && an.aln_start >= 0 && (size_t)an.ref_id < G.len.size()) {
      std::vector<MethRun> runs;
      meth_runs(an.cigar, an.aln_start, (an.flags & bam::F_REVERSE) != 0, an.orig_cigar, runs);
      pr.meth_job = (int64_t)B.add_meth_job(rd0, (uint32_t)srs.size(), an.len, runs, meth_is_top_strand(an.flags), G.off[(size_t)an.ref_id], G.len[(size_t)an.ref_id]);
      pr.meth_top = meth_is_top_strand(srs[0].flags);
    }
  }
  pr.job = B.add_job(rd0, (uint32_t)srs.size(), cons_len);
  pr.read_type = (uint8_t)rt;
  pr.umi = umi;
  pr.first_raw = reads[srs[0].orig_idx].p;
  pr.first_len = reads[srs[0].orig_idx].n;
  for (auto& s : srs) {
    Rec v{reads[s.orig_idx].p, reads[s.orig_idx].n};
    uint32_t vl;
    int64_t off = bam::find_z_tag(v.b + v.aux_off(), v.len > v.aux_off() ? v.len - v.aux_off() : 0, 'R', 'X', &vl);
    if (off >= 0) pr.rx.emplace_back((const char*)v.b + v.aux_off() + off, vl);
  }
  out_pending.push_back(std::move(pr));
  ok = true;
  return 0;
}

// process_group (vanilla_caller.rs:1329-1422)
int process_group(Ctx& x, const std::string& umi, const std::vector<Positioned>& records) {
  x.stats.total_reads += records.size();
  std::vector<Positioned> reads;
  size_t filtered = 0;
  for (auto& r : records) {
    uint16_t f = Rec{r.p, r.n}.flags();
    if ((f & bam::F_SECONDARY) == 0 && (f & bam::F_SUPPLEMENTARY) == 0) reads.push_back(r);
    else { filtered++; x.reject_pos(r.pos, r.p, r.n); }
  }
  if (filtered) x.stats.reject(FGX_REJ_SECONDARY_OR_SUPPLEMENTARY, filtered);
  if (reads.empty()) { x.flush_group_rejects(); return 0; }
  if (reads.size() < x.o.min_reads) {
    x.stats.reject(FGX_REJ_INSUFFICIENT_READS, reads.size());
    for (auto& r : reads) x.reject_pos(r.pos, r.p, r.n);
    x.flush_group_rejects();
    return 0;
  }
  std::vector<Positioned> frag, r1, r2;
  for (auto& r : reads) {
    uint16_t f = Rec{r.p, r.n}.flags();
    if (!(f & bam::F_PAIRED)) frag.push_back(r);
    else if (f & bam::F_FIRST) r1.push_back(r);
    else if (f & bam::F_LAST) r2.push_back(r);
  }
  bool ok;
  uint32_t sc;
  std::vector<uint32_t> si;
  if (process_subgroup(x, umi, RT_FRAGMENT, frag, ok, sc, si, x.pending) < 0) return -1;
  if (ok) x.stats.consensus_reads += 1;
  std::vector<PendingRecord> pair_pending;
  bool ok1, ok2;
  uint32_t c1, c2;
  std::vector<uint32_t> s1, s2;
  if (process_subgroup(x, umi, RT_R1, r1, ok1, c1, s1, pair_pending) < 0) return -1;
  if (process_subgroup(x, umi, RT_R2, r2, ok2, c2, s2, pair_pending) < 0) return -1;
  if (ok1 && ok2) {
    x.stats.consensus_reads += 2;
    for (auto& p : pair_pending) x.pending.push_back(std::move(p));
  } else if (ok1) {
    x.stats.reject(FGX_REJ_ORPHAN_CONSENSUS, c1);
    for (uint32_t i : s1) x.reject_pos(r1[i].pos, r1[i].p, r1[i].n);
  } else if (ok2) {
    x.stats.reject(FGX_REJ_ORPHAN_CONSENSUS, c2);
    for (uint32_t i : s2) x.reject_pos(r2[i].pos, r2[i].p, r2[i].n);
  }
  x.flush_group_rejects();
  return 0;
}

}  // namespace

int simplex_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                            const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  (void)n_rec;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  Ctx x(c);
  c->batch.clear();
  c->out_data.clear();
  c->out_rejects.clear();
  c->grp_out_end.assign(n_grp, 0);
  std::vector<uint32_t> grp_pending_end(n_grp, 0);
  const fgx_options& o = c->opt;
  std::vector<std::vector<uint8_t>> scratch;   // per-group mutable copies for the overlap pre-step
  std::vector<Positioned> records;
  for (uint32_t g = 0; g < n_grp; g++) {
    uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
    uint32_t n = r1 - r0;
    if (n < o.min_reads) {   // simplex.rs:673-683
      x.stats.total_reads += n;
      x.stats.reject(FGX_REJ_INSUFFICIENT_READS, n);
      if (o.track_rejects) for (uint32_t r = r0; r < r1; r++) x.reject_now(blob + rec_off[r], rec_len[r]);
      grp_pending_end[g] = (uint32_t)x.pending.size();
      continue;
    }
    records.clear();
    if (o.overlapping_consensus) {
      // The pre-step rewrites bases/quals of overlapping mates; work on copies of the records.
      // Copies must outlive this loop (PendingRecord keeps pointers for tag lookup), so park them.
      size_t base = scratch.size();
      for (uint32_t r = r0; r < r1; r++) scratch.emplace_back(blob + rec_off[r], blob + rec_off[r] + rec_len[r]);
      std::vector<MutRec> mut;
      for (uint32_t i = 0; i < n; i++) mut.push_back(MutRec{scratch[base + i].data(), (uint32_t)scratch[base + i].size()});
      apply_overlapping_consensus(mut, x.ov);
      for (uint32_t i = 0; i < n; i++) records.push_back(Positioned{i, scratch[base + i].data(), (uint32_t)scratch[base + i].size()});
    } else {
      for (uint32_t r = r0; r < r1; r++) records.push_back(Positioned{r - r0, blob + rec_off[r], rec_len[r]});
    }
    // ConsensusCaller::consensus_reads (vanilla_caller.rs:1885-1909): UMI = MI of the first record
    Rec first{records[0].p, records[0].n};
    uint32_t vl;
    int64_t off = bam::find_z_tag(first.b + first.aux_off(), first.len > first.aux_off() ? first.len - first.aux_off() : 0,
                                  (uint8_t)o.tag[0], (uint8_t)o.tag[1], &vl);
    if (off < 0) {
      c->err = "Missing UMI tag '" + std::string(o.tag, 2) + "' for read '" + std::string((const char*)first.name(), first.name_len()) + "'";
      return 2;
    }
    std::string umi((const char*)first.b + first.aux_off() + off, vl);
    if (process_group(x, umi, records) < 0) { c->err = x.err; return 2; }
    grp_pending_end[g] = (uint32_t)x.pending.size();
  }
  auto t1 = clk::now();

  ColParams prm{o.min_reads, o.min_consensus_base_quality};
  double ms_k = c->run_columns(c->batch, prm);
  auto t2 = clk::now();

  // build_consensus_record_into (vanilla_caller.rs:1767-1881)
  ColumnBatch& B = c->batch;
  std::vector<uint8_t> rec;
  uint32_t g_cur = 0, pi = 0;
  for (auto& pr : x.pending) {
    while (g_cur < n_grp && grp_pending_end[g_cur] <= pi) { c->grp_out_end[g_cur] = c->out_data.size(); g_cur++; }
    pi++;
    const ColJob& j = B.jobs[pr.job];
    const uint8_t* bases = B.ob.data() + j.out_off;
    const uint8_t* quals = B.oq.data() + j.out_off;
    const uint16_t* depths = B.od.data() + j.out_off;
    const uint16_t* errors = B.oe.data() + j.out_off;
    std::string name = c->prefix + ":" + pr.umi;
    uint16_t flag = bam::F_UNMAPPED;
    if (pr.read_type == RT_R1) flag |= bam::F_PAIRED | bam::F_FIRST | bam::F_MATE_UNMAPPED;
    else if (pr.read_type == RT_R2) flag |= bam::F_PAIRED | bam::F_LAST | bam::F_MATE_UNMAPPED;
    if (!build_unmapped_record(rec, name, flag, bases, quals, j.cons_len)) {
      c->err = "could not write the consensus record for read '" + name + "': read name too long";
      return 2;
    }
    tag_z(rec, "RG", c->rg.data(), c->rg.size());
    append_depth_error_tags(rec, depths, errors, j.cons_len, o.produce_per_base_tags != 0);
    tag_z(rec, "MI", pr.umi.data(), pr.umi.size());
    if (o.cell_tag[0]) {
      Rec fr{pr.first_raw, pr.first_len};
      uint32_t vl;
      int64_t off = bam::find_z_tag(fr.b + fr.aux_off(), fr.len > fr.aux_off() ? fr.len - fr.aux_off() : 0, (uint8_t)o.cell_tag[0],
                                    (uint8_t)o.cell_tag[1], &vl);
      if (off >= 0) tag_z(rec, o.cell_tag, (const char*)fr.b + fr.aux_off() + off, vl);
    }
    if (!pr.rx.empty()) {
      std::string cu;
      if (!consensus_umis(c->h_umi_tables.t, pr.rx, cu)) { c->err = "consensus_umis: UMIs of unequal length or mixed DNA/non-DNA characters"; return 2; }
      tag_z(rec, "RX", cu.data(), cu.size());
    }
    if (pr.meth_job >= 0 && !B.mflag.empty()) {   // MM, ML, cu, ct (vanilla_caller.rs:1853-1876); the annotation is cut to the consensus length
      const MethJob& mj = B.mjobs[(size_t)pr.meth_job];
      const uint32_t n = std::min(mj.n_pos, j.cons_len);
      std::string mm;
      std::vector<uint8_t> ml;
      if (n == j.cons_len && meth_build_mm_ml(bases, n, B.mflag.data() + mj.out_off, B.mu.data() + mj.out_off, B.mt.data() + mj.out_off, pr.meth_top, o.methylation_mode, mm, ml)) {
        tag_z(rec, "MM", mm.data(), mm.size());
        tag_u8_array(rec, "ML", ml.data(), (uint32_t)ml.size());
      }
      tag_count_array(rec, "cu", B.mu.data() + mj.out_off, n);
      tag_count_array(rec, "ct", B.mt.data() + mj.out_off, n);
    }
    append_with_block_size(c->out_data, rec.data(), (uint32_t)rec.size());
  }
  for (; g_cur < n_grp; g_cur++) c->grp_out_end[g_cur] = c->out_data.size();
  auto t3 = clk::now();

  memset(out, 0, sizeof(*out));
  out->data = c->out_data.data();
  out->data_len = c->out_data.size();
  out->count = x.pending.size();
  x.stats.to_array(out->stats);
  for (int i = 0; i < 4; i++) out->stats[24 + i] = x.ov[i];
  out->rejects = c->out_rejects.data();
  out->rejects_len = c->out_rejects.size();
  out->n_rejects = x.n_rejects;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  out->ms_host_prep = ms(t0, t1);
  out->ms_kernels = ms_k;
  out->ms_h2d = ms(t1, t2) - ms_k;
  out->ms_emit = ms(t2, t3);
  return 0;
}

}  // namespace fgx
