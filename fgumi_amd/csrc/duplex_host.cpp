// duplex_host.cpp — duplex consensus, general path: host orchestration of one batch of MI groups, the four
// single-strand consensus calls per molecule on the device (k_column_jobs), the integer A/B strand
// combine and the BAM record assembly on the host.
//
// Mirrors, in batch form:
//   src/lib/commands/duplex.rs:742-830        process_fn (conditional overlap pre-step, rejects)
//   crates/fgumi-consensus/src/duplex_caller.rs:2545-2624 consensus_reads (+ stats re-attribution :1894-1926),
//       1944-2540 process_group, 931-1108 duplex_consensus, 1118-1405 duplex_read_into
// The per-position arithmetic (ss_caller.consensus_call → create_consensus_from_source_reads) never runs on
// the host: the four read sets are staged and called by the HIP kernel.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include "bamrec.h"
#include "engine.h"
#include "host_common.h"
#include "host_reads.h"

namespace fgx {

using bam::Rec;

namespace {

struct RawRef { const uint8_t* p; uint32_t n; };

struct SsCall {            // one ss_caller.consensus_call
  int64_t job = -1;        // column job (None when < 0)
  int64_t meth = -1;       // methylation-aware mode: the call's annotation job (-1: none)
  std::vector<uint32_t> src_rd;   // ReadDescs of ALL (uncapped) source reads, for the duplex error recount
  std::vector<RawRef> raws;       // alignment-filtered raw records (RX source)
};

struct Molecule {
  bool early = false;             // decided before the device pass (empty / rejected whole group)
  bool early_kept = true;
  std::string base_mi;
  std::vector<RawRef> a, b;       // strand records (whole-group rejects, counts)
  SsCall ab_r1, ab_r2, ba_r1, ba_r2;
  HostStats caller_stats;         // DuplexConsensusCaller::stats so far (input, FragmentRead)
  HostStats group_stats;          // process_group's stats
  HostStats ss_stats;             // ss_caller's stats for this molecule
  std::vector<std::vector<uint8_t>> frag_rejects, ss_rejects;
  bool has_cb = false;
  std::string cell_barcode;
};

struct View {              // arrays of one single-strand consensus
  const uint8_t* bases; const uint8_t* quals; const uint16_t* depths; const uint16_t* errors; uint32_t len;
  // its methylation annotation, cut to the consensus length (null: none) — VanillaConsensusRead::methylation
  const uint8_t* mflag = nullptr; const uint32_t* mu = nullptr; const uint32_t* mt = nullptr;
};

struct Annot {             // MethylationAnnotation
  bool some = false;
  std::vector<uint8_t> flag;
  std::vector<uint32_t> u, t;
  void from(const View& v, uint32_t n) { some = v.mflag != nullptr; if (some) { flag.assign(v.mflag, v.mflag + n); u.assign(v.mu, v.mu + n); t.assign(v.mt, v.mt + n); } }
};

struct Duplex {            // DuplexConsensusRead
  std::vector<uint8_t> bases, quals;
  std::vector<uint16_t> errors;
  std::vector<uint8_t> ab_b, ab_q, ba_b, ba_q;
  std::vector<uint16_t> ab_d, ab_e, ba_d, ba_e;
  bool has_ba = false;
  bool is_ba_only = false;
  Annot meth, ab_meth, ba_meth;      // combined and per-strand annotations (duplex_caller.rs:1086-1094)
  uint16_t ab_max() const { uint16_t m = 0; for (auto d : ab_d) m = std::max(m, d); return m; }
  uint16_t ba_max() const { uint16_t m = 0; for (auto d : ba_d) m = std::max(m, d); return m; }
};

inline uint8_t cap_quality(int32_t s) { return s < 2 ? 2 : s > 93 ? 93 : (uint8_t)s; }
inline int32_t clamp_short(uint16_t v) { return v > 32767 ? 32767 : v; }
inline bool is_conversion_pair(uint8_t x, uint8_t y) {   // duplex_caller.rs:897-903
  x = meth_upper(x); y = meth_upper(y);
  return (x == 'C' && y == 'T') || (x == 'T' && y == 'C') || (x == 'G' && y == 'A') || (x == 'A' && y == 'G');
}
inline uint8_t unconverted_base(uint8_t x, uint8_t y) {   // :907-913
  const uint8_t a = meth_upper(x), b = meth_upper(y);
  if ((a == 'C' && b == 'T') || (a == 'T' && b == 'C')) return 'C';
  if ((a == 'G' && b == 'A') || (a == 'A' && b == 'G')) return 'G';
  return x;
}
inline uint32_t sat_add_u32(uint32_t a, uint32_t b) { const uint64_t v = (uint64_t)a + b; return v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v; }

// duplex_consensus (duplex_caller.rs:931-1108).  srcs: ReadDescs of the source reads of both strands, or null.
bool duplex_consensus(const ColumnBatch& B, const View* ab, const View* ba, const std::vector<uint32_t>* srcs, Duplex& out) {
  uint32_t len = std::min(ab ? ab->len : 0xFFFFFFFFu, ba ? ba->len : 0xFFFFFFFFu);
  auto covered = [&](const View* v) { if (!v) return false; for (uint32_t i = 0; i < std::min(len, v->len); i++) if (v->depths[i] > 0) return true; return false; };
  const View* a = covered(ab) ? ab : nullptr;
  const View* b = covered(ba) ? ba : nullptr;
  auto copy_strand = [](const View& v, uint32_t n, std::vector<uint8_t>& bb, std::vector<uint8_t>& qq, std::vector<uint16_t>& dd, std::vector<uint16_t>& ee) {
    bb.assign(v.bases, v.bases + n); qq.assign(v.quals, v.quals + n); dd.assign(v.depths, v.depths + n); ee.assign(v.errors, v.errors + n);
  };
  out = Duplex();
  if (!a && !b) return false;
  if (!a || !b) {
    const View& s = a ? *a : *b;
    out.bases.assign(s.bases, s.bases + s.len); out.quals.assign(s.quals, s.quals + s.len); out.errors.assign(s.errors, s.errors + s.len);
    copy_strand(s, s.len, out.ab_b, out.ab_q, out.ab_d, out.ab_e);
    out.has_ba = false;
    out.is_ba_only = !a;
    out.ab_meth.from(s, s.len);
    out.meth = out.ab_meth;
    return true;
  }
  for (uint32_t i = 0; i < len; i++) {
    uint8_t ab_b = a->bases[i], ba_b = b->bases[i];
    int32_t aq = a->quals[i], bq = b->quals[i];
    uint8_t raw_base, raw_qual;
    // a C/T (G/A) disagreement at a reference cytosine of either strand is a conversion event, not an error (:988-1005)
    const bool is_ref_c = (a->mflag && a->mflag[i]) || (b->mflag && b->mflag[i]);
    const bool artifact = ab_b != ba_b && is_ref_c && is_conversion_pair(ab_b, ba_b);
    if (artifact) { raw_base = unconverted_base(ab_b, ba_b); raw_qual = cap_quality(aq + bq); }
    else if (ab_b == ba_b) { raw_base = ab_b; raw_qual = cap_quality(aq + bq); }
    else if (aq > bq) { raw_base = ab_b; raw_qual = cap_quality(aq - bq); }
    else if (bq > aq) { raw_base = ba_b; raw_qual = cap_quality(bq - aq); }
    else { raw_base = ab_b; raw_qual = FGX_MIN_PHRED; }
    if (ab_b == 'N' || ba_b == 'N' || raw_qual == FGX_MIN_PHRED) { out.bases.push_back('N'); out.quals.push_back(FGX_MIN_PHRED); }
    else { out.bases.push_back(raw_base); out.quals.push_back(raw_qual); }
    int64_t err;
    if (artifact) err = 0;
    else if (srcs) {
      err = 0;
      for (uint32_t rd : *srcs) {
        const ReadDesc& d = B.reads[rd];
        if (d.len > i) { uint8_t sb = B.stage[d.off + i]; if (sb != 'N' && raw_base != 'N' && sb != raw_base) err++; }
      }
    } else {
      int32_t ae = a->errors[i], be = b->errors[i], ad = a->depths[i], bd = b->depths[i];
      err = (ab_b == ba_b) ? ae + be : (raw_base == ab_b) ? ae + (bd - be) : be + (ad - ae);
    }
    out.errors.push_back((uint16_t)std::min<int64_t>(std::max<int64_t>(err, 0), 32767));
  }
  copy_strand(*a, len, out.ab_b, out.ab_q, out.ab_d, out.ab_e);
  copy_strand(*b, len, out.ba_b, out.ba_q, out.ba_d, out.ba_e);
  out.has_ba = true;
  out.ab_meth.from(*a, len);
  out.ba_meth.from(*b, len);
  if (out.ab_meth.some && out.ba_meth.some) {   // combine_methylation_annotations (methylation.rs:404-427)
    out.meth.some = true;
    for (uint32_t i = 0; i < len; i++) {
      out.meth.flag.push_back(out.ab_meth.flag[i] | out.ba_meth.flag[i]);
      out.meth.u.push_back(sat_add_u32(out.ab_meth.u[i], out.ba_meth.u[i]));
      out.meth.t.push_back(sat_add_u32(out.ab_meth.t[i], out.ba_meth.t[i]));
    }
  } else if (out.ab_meth.some) out.meth = out.ab_meth;
  else if (out.ba_meth.some) out.meth = out.ba_meth;
  return true;
}

struct Ctx {
  fgx_caller* c;
  const fgx_options& o;
  uint32_t min_total, min_xy, min_yx;
  std::string err;
  explicit Ctx(fgx_caller* cc) : c(cc), o(cc->opt), min_total(o.duplex_min_reads[0]), min_xy(o.duplex_min_reads[1]), min_yx(o.duplex_min_reads[2]) {}
  bool min_reads_ok(size_t na, size_t nb) const {
    size_t xy = std::max(na, nb), yx = std::min(na, nb);
    return min_total <= xy + yx && min_xy <= xy && min_yx <= yx;
  }
};

inline bool is_r1(const RawRef& r) { uint16_t f = Rec{r.p, r.n}.flags(); return (f & bam::F_PAIRED) && (f & bam::F_FIRST); }
inline bool is_r2(const RawRef& r) { uint16_t f = Rec{r.p, r.n}.flags(); return (f & bam::F_PAIRED) && (f & bam::F_LAST); }

inline bool find_tag(const RawRef& r, char t0, char t1, const char** v, uint32_t* n) {
  Rec rv{r.p, r.n};
  uint32_t an = rv.len > rv.aux_off() ? rv.len - rv.aux_off() : 0;
  int64_t off = bam::find_z_tag(rv.b + rv.aux_off(), an, (uint8_t)t0, (uint8_t)t1, n);
  if (off < 0) return false;
  *v = (const char*)rv.b + rv.aux_off() + off;
  return true;
}

// duplex_read_into (duplex_caller.rs:1118-1405)
bool duplex_read_into(Ctx& x, std::vector<uint8_t>& out, const Duplex& d, int read_type, const std::string& umi, const std::vector<RawRef>& src_a,
                      const std::vector<RawRef>& src_b, bool first_of_pair, const Molecule& m) {
  uint16_t flag = bam::F_UNMAPPED;
  if (read_type == 1) flag |= bam::F_PAIRED | bam::F_FIRST | bam::F_MATE_UNMAPPED;
  else if (read_type == 2) flag |= bam::F_PAIRED | bam::F_LAST | bam::F_MATE_UNMAPPED;
  std::string name = x.c->prefix + ":" + umi;
  std::vector<uint8_t> rec;
  if (!build_unmapped_record(rec, name, flag, d.bases.data(), d.quals.data(), (uint32_t)d.bases.size())) {
    x.err = "could not write the consensus record for read '" + name + "': read name too long";
    return false;
  }
  tag_z(rec, "MI", umi.data(), umi.size());
  if (x.o.cell_tag[0] && m.has_cb) tag_z(rec, x.o.cell_tag, m.cell_barcode.data(), m.cell_barcode.size());
  tag_z(rec, "RG", x.c->rg.data(), x.c->rg.size());
  auto strand = [&](const std::vector<uint16_t>& dep, const std::vector<uint16_t>& er, int32_t& dmax, int32_t& dmin, float& rate) {
    dmax = 0; dmin = 0;
    int64_t td = 0, te = 0;
    for (size_t i = 0; i < dep.size(); i++) { int32_t v = clamp_short(dep[i]); if (i == 0) { dmax = v; dmin = v; } dmax = std::max(dmax, v); dmin = std::min(dmin, v); td += v; }
    for (auto e : er) te += clamp_short(e);
    rate = td > 0 ? (float)te / (float)td : 0.0f;
  };
  auto per_base = [&](const char* tc, const char* td_, const char* te_, const char* tq, const std::vector<uint8_t>& bb, const std::vector<uint8_t>& qq,
                      const std::vector<uint16_t>& dd, const std::vector<uint16_t>& ee) {
    tag_z(rec, tc, (const char*)bb.data(), bb.size());
    tag_i16_array(rec, td_, dd.data(), (uint32_t)dd.size());
    tag_i16_array(rec, te_, ee.data(), (uint32_t)ee.size());
    tag_phred33(rec, tq, qq.data(), (uint32_t)qq.size());
  };
  int32_t amax, amin; float aer;
  strand(d.ab_d, d.ab_e, amax, amin, aer);
  tag_int(rec, "aD", amax); tag_float(rec, "aE", aer); tag_int(rec, "aM", amin);
  if (x.o.produce_per_base_tags) per_base("ac", "ad", "ae", "aq", d.ab_b, d.ab_q, d.ab_d, d.ab_e);
  int32_t bmax = 0, bmin = 0; float ber = 0.0f;
  if (d.has_ba) strand(d.ba_d, d.ba_e, bmax, bmin, ber);
  tag_int(rec, "bD", bmax); tag_float(rec, "bE", ber); tag_int(rec, "bM", bmin);
  if (x.o.produce_per_base_tags && d.has_ba) per_base("bc", "bd", "be", "bq", d.ba_b, d.ba_q, d.ba_d, d.ba_e);
  int32_t cmax = 0, cmin = 0;
  int64_t td = 0, te = 0;
  for (size_t i = 0; i < d.bases.size(); i++) {
    int32_t v = clamp_short(i < d.ab_d.size() ? d.ab_d[i] : 0) + clamp_short(d.has_ba && i < d.ba_d.size() ? d.ba_d[i] : 0);
    if (i == 0) { cmax = v; cmin = v; }
    cmax = std::max(cmax, v); cmin = std::min(cmin, v); td += v;
  }
  for (auto e : d.errors) te += clamp_short(e);
  tag_int(rec, "cD", cmax); tag_float(rec, "cE", td > 0 ? (float)te / (float)td : 0.0f); tag_int(rec, "cM", cmin);
  std::vector<std::string> umis;
  auto add = [&](const std::vector<RawRef>& src) {
    for (auto& r : src) {
      const char* v; uint32_t n;
      if (!find_tag(r, 'R', 'X', &v, &n)) continue;
      std::string s(v, n);
      bool is_first = Rec{r.p, r.n}.flags() & bam::F_FIRST;
      if (is_first == first_of_pair) umis.push_back(s);
      else {   // split('-').rev().join("-")
        std::vector<std::string> parts;
        size_t st = 0;
        for (;;) { size_t k = s.find('-', st); if (k == std::string::npos) { parts.push_back(s.substr(st)); break; } parts.push_back(s.substr(st, k - st)); st = k + 1; }
        std::string j;
        for (size_t i = parts.size(); i-- > 0;) { j += parts[i]; if (i) j += "-"; }
        umis.push_back(j);
      }
    }
  };
  add(src_a); add(src_b);
  if (!umis.empty()) {
    std::string cu;
    if (!consensus_umis(x.c->h_umi_tables.t, umis, cu)) { x.err = "consensus_umis: UMIs of unequal length or mixed DNA/non-DNA characters"; return false; }
    tag_z(rec, "RX", cu.data(), cu.size());
  }
  if (d.meth.some) {   // am/au/at, bm/bu/bt, MM/ML/cu/ct (duplex_caller.rs:1338-1398)
    const bool top = !d.is_ba_only;
    const int mode = x.o.methylation_mode;
    std::string mm;
    std::vector<uint8_t> ml;
    auto counts = [&](const Annot& a, const char* tu, const char* tt) { tag_count_array(rec, tu, a.u.data(), (uint32_t)a.u.size()); tag_count_array(rec, tt, a.t.data(), (uint32_t)a.t.size()); };
    if (d.ab_meth.some) {
      if (meth_build_mm_ml(d.ab_b.data(), (uint32_t)d.ab_b.size(), d.ab_meth.flag.data(), d.ab_meth.u.data(), d.ab_meth.t.data(), top, mode, mm, ml)) tag_z(rec, top ? "am" : "bm", mm.data(), mm.size());
      counts(d.ab_meth, top ? "au" : "bu", top ? "at" : "bt");
    }
    if (d.has_ba && d.ba_meth.some) {
      if (meth_build_mm_ml(d.ba_b.data(), (uint32_t)d.ba_b.size(), d.ba_meth.flag.data(), d.ba_meth.u.data(), d.ba_meth.t.data(), false, mode, mm, ml)) tag_z(rec, "bm", mm.data(), mm.size());
      counts(d.ba_meth, "bu", "bt");
    }
    if (meth_build_mm_ml(d.bases.data(), (uint32_t)d.bases.size(), d.meth.flag.data(), d.meth.u.data(), d.meth.t.data(), top, mode, mm, ml)) {
      tag_z(rec, "MM", mm.data(), mm.size());
      tag_u8_array(rec, "ML", ml.data(), (uint32_t)ml.size());
    }
    counts(d.meth, "cu", "ct");
  }
  append_with_block_size(out, rec.data(), (uint32_t)rec.size());
  return true;
}

}  // namespace

int duplex_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                           const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  (void)n_rec;
  using clk = std::chrono::steady_clock;
  auto t0 = clk::now();
  Ctx x(c);
  const fgx_options& o = c->opt;
  if (x.min_xy > x.min_total || x.min_yx > x.min_xy) { c->err = "min-reads values must be specified high to low (total >= XY >= YX)"; return 2; }
  ColumnBatch& B = c->batch;
  B.clear();
  c->out_data.clear();
  c->out_rejects.clear();
  c->grp_out_end.assign(n_grp, 0);
  const bool track = o.track_rejects;
  const bool single_strand_allowed = x.min_yx == 0;
  const bool meth_on = o.methylation_mode != FGX_METHYLATION_DISABLED && c->genome;
  const SrcParams sp{o.min_input_base_quality, o.trim != 0, o.duplex_max_reads_per_strand >= 0, meth_on};
  B.want_stage_back = meth_on;   // the duplex error recount reads the normalised source reads (consensus_call keeps them, vanilla_caller.rs:771)
  std::vector<std::vector<uint8_t>> scratch;
  std::vector<Molecule> mols(n_grp);
  uint64_t ov[4] = {0, 0, 0, 0};
  std::vector<uint8_t> tb, tq;

  for (uint32_t g = 0; g < n_grp; g++) {
    Molecule& m = mols[g];
    uint32_t r0 = grp_first[g], r1 = grp_first[g + 1], n = r1 - r0;
    std::vector<RawRef> records;
    // conditional overlap pre-step (duplex.rs:786-795)
    bool both = false;
    if (o.overlapping_consensus && !single_strand_allowed && n >= 2) {
      bool ha = false, hb = false;
      for (uint32_t r = r0; r < r1 && !(ha && hb); r++) {
        const char* v; uint32_t vl;
        if (find_tag(RawRef{blob + rec_off[r], rec_len[r]}, 'M', 'I', &v, &vl) && vl >= 2 && v[vl - 2] == '/') { if (v[vl - 1] == 'A') ha = true; else if (v[vl - 1] == 'B') hb = true; }
      }
      both = ha && hb;
    }
    if (o.overlapping_consensus && (single_strand_allowed || both)) {
      size_t base = scratch.size();
      for (uint32_t r = r0; r < r1; r++) scratch.emplace_back(blob + rec_off[r], blob + rec_off[r] + rec_len[r]);
      std::vector<MutRec> mut;
      for (uint32_t i = 0; i < n; i++) mut.push_back(MutRec{scratch[base + i].data(), (uint32_t)scratch[base + i].size()});
      apply_overlapping_consensus(mut, ov);
      for (uint32_t i = 0; i < n; i++) records.push_back(RawRef{scratch[base + i].data(), (uint32_t)scratch[base + i].size()});
    } else for (uint32_t r = r0; r < r1; r++) records.push_back(RawRef{blob + rec_off[r], rec_len[r]});

    // consensus_reads (duplex_caller.rs:2545-2570)
    m.caller_stats.total_reads += n;
    std::vector<RawRef> paired;
    size_t nfrag = 0;
    for (auto& r : records) {
      if (Rec{r.p, r.n}.flags() & bam::F_PAIRED) paired.push_back(r);
      else { nfrag++; if (track) m.frag_rejects.emplace_back(r.p, r.p + r.n); }
    }
    if (nfrag) m.caller_stats.reject(FGX_REJ_FRAGMENT_READ, nfrag);
    bool have_mi = false;
    for (auto& r : paired) {
      const char* v; uint32_t vl;
      if (!find_tag(r, 'M', 'I', &v, &vl)) {
        Rec rv{r.p, r.n};
        c->err = "Read '" + std::string((const char*)rv.name(), rv.name_len()) + "' is missing MI tag. The duplex command requires all reads to have MI tags.";
        return 2;
      }
      if (!have_mi) { have_mi = true; m.base_mi = vl >= 2 ? std::string(v, vl - 2) : std::string(v, vl); }
      char strand = 0;
      if (vl >= 2 && v[vl - 2] == '/') { if (v[vl - 1] == 'A') strand = 'A'; else if (v[vl - 1] == 'B') strand = 'B'; }
      if (strand == 'A') m.a.push_back(r);
      else if (strand == 'B') m.b.push_back(r);
      else { c->err = "Read has MI tag '" + std::string(v, vl) + "' without /A or /B suffix. The duplex command requires reads to be grouped using the 'paired' strategy."; return 2; }
    }
    if (!have_mi) { m.early = true; m.early_kept = true; continue; }

    // process_group (duplex_caller.rs:1944-2120)
    auto reject_all = [&](int why) { m.group_stats.reject(why, m.a.size() + m.b.size()); m.early = true; m.early_kept = false; };
    if (m.a.empty() && m.b.empty()) { m.early = true; m.early_kept = true; continue; }
    size_t na = 0, nb = 0;
    for (auto& r : m.a) na += is_r1(r);
    for (auto& r : m.b) nb += is_r1(r);
    if (!x.min_reads_ok(na, nb)) { reject_all(FGX_REJ_INSUFFICIENT_READS); continue; }
    if (o.cell_tag[0]) {
      const RawRef& f = !m.a.empty() ? m.a[0] : m.b[0];
      const char* v; uint32_t vl;
      if (find_tag(f, o.cell_tag[0], o.cell_tag[1], &v, &vl)) { m.has_cb = true; m.cell_barcode.assign(v, vl); }
    }
    std::vector<RawRef> ab_r1, ab_r2, ba_r1, ba_r2;
    for (auto& r : m.a) { if (is_r1(r)) ab_r1.push_back(r); if (is_r2(r)) ab_r2.push_back(r); }
    for (auto& r : m.b) { if (is_r1(r)) ba_r1.push_back(r); if (is_r2(r)) ba_r2.push_back(r); }
    if (!m.a.empty() && !m.b.empty()) {
      auto same_strand = [&](const std::vector<RawRef>& p, const std::vector<RawRef>& q) {
        bool have = false, first_rev = false;
        for (auto* v : {&p, &q}) for (auto& r : *v) { bool rv = Rec{r.p, r.n}.flags() & bam::F_REVERSE; if (!have) { have = true; first_rev = rv; } else if (rv != first_rev) return false; }
        return true;
      };
      if (!same_strand(ab_r1, ba_r2) || !same_strand(ab_r2, ba_r1)) { reject_all(FGX_REJ_POTENTIAL_COLLISION); continue; }
    }
    std::vector<RawRef> x_raws = ab_r1, y_raws = ab_r2;
    x_raws.insert(x_raws.end(), ba_r2.begin(), ba_r2.end());
    y_raws.insert(y_raws.end(), ba_r1.begin(), ba_r1.end());
    std::vector<SrcRead> xs, ys;
    std::vector<uint32_t> xz, yz;
    auto to_sources = [&](const std::vector<RawRef>& raws, std::vector<SrcRead>& srcs, std::vector<uint32_t>& zero) {
      for (uint32_t i = 0; i < raws.size(); i++) {
        uint64_t clip = mate_clip_raw(Rec{raws[i].p, raws[i].n});
        SrcRead sr;
        int rc = make_source_read(B, sp, raws[i].p, raws[i].n, i, clip, sr, tb, tq, x.err);
        if (rc < 0) return false;
        if (rc == 1) srcs.push_back(std::move(sr)); else zero.push_back(i);
      }
      return true;
    };
    if (!to_sources(x_raws, xs, xz) || !to_sources(y_raws, ys, yz)) { c->err = x.err; return 2; }
    std::vector<uint32_t> x_rej, y_rej;
    filter_by_alignment(xs, m.ss_stats, x_rej);
    filter_by_alignment(ys, m.ss_stats, y_rej);
    {
      size_t nz = xz.size() + yz.size();
      if (nz) m.ss_stats.reject(FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING, nz);
      if (track) {
        auto ordinals = [&](bool a_r1_b_r2) {
          std::vector<uint32_t> ord;
          for (uint32_t i = 0; i < m.a.size(); i++) if (a_r1_b_r2 ? is_r1(m.a[i]) : is_r2(m.a[i])) ord.push_back(i);
          for (uint32_t j = 0; j < m.b.size(); j++) if (a_r1_b_r2 ? is_r2(m.b[j]) : is_r1(m.b[j])) ord.push_back((uint32_t)m.a.size() + j);
          return ord;
        };
        std::vector<uint32_t> xo = ordinals(true), yo = ordinals(false);
        std::vector<std::pair<uint32_t, RawRef>> zr, sr;
        for (uint32_t i : xz) zr.push_back({xo[i], x_raws[i]});
        for (uint32_t i : yz) zr.push_back({yo[i], y_raws[i]});
        std::stable_sort(zr.begin(), zr.end(), [](const auto& p, const auto& q) { return p.first < q.first; });
        for (auto& e : zr) m.ss_rejects.emplace_back(e.second.p, e.second.p + e.second.n);
        for (uint32_t i : x_rej) sr.push_back({xo[i], x_raws[i]});
        for (uint32_t i : y_rej) sr.push_back({yo[i], y_raws[i]});
        std::stable_sort(sr.begin(), sr.end(), [](const auto& p, const auto& q) { return p.first < q.first; });
        for (auto& e : sr) m.ss_rejects.emplace_back(e.second.p, e.second.p + e.second.n);
      }
    }
    auto split = [&](std::vector<SrcRead>& f, std::vector<SrcRead>& first, std::vector<SrcRead>& rest) {
      for (auto& s : f) { if (s.flags & bam::F_FIRST) first.push_back(std::move(s)); else rest.push_back(std::move(s)); }
    };
    std::vector<SrcRead> f_ab_r1, f_ba_r2, f_ba_r1, f_ab_r2;
    split(xs, f_ab_r1, f_ba_r2);
    split(ys, f_ba_r1, f_ab_r2);
    auto call = [&](SsCall& sc, const std::vector<SrcRead>& srs, const std::vector<RawRef>& raws) {
      for (auto& s : srs) { sc.src_rd.push_back(s.rd); sc.raws.push_back(raws[s.orig_idx]); }
      sc.job = stage_consensus_call(B, srs, o.duplex_max_reads_per_strand, meth_on ? c->genome.get() : nullptr, &sc.meth);
    };
    call(m.ab_r1, f_ab_r1, x_raws); call(m.ab_r2, f_ab_r2, y_raws); call(m.ba_r1, f_ba_r1, y_raws); call(m.ba_r2, f_ba_r2, x_raws);
  }
  auto t1 = clk::now();

  // single-strand caller settings (duplex_caller.rs:474-489): min_reads 1, min consensus base quality 2
  double ms_k = c->run_columns(B, ColParams{1, FGX_MIN_PHRED});
  auto t2 = clk::now();

  HostStats batch;
  uint64_t n_rejects = 0, count = 0;
  auto reject_out = [&](const uint8_t* p, size_t n) { append_with_block_size(c->out_rejects, p, (uint32_t)n); n_rejects++; };
  for (uint32_t g = 0; g < n_grp; g++) {
    Molecule& m = mols[g];
    bool kept = true;
    if (m.early) kept = m.early_kept;
    else {
      auto view = [&](const SsCall& sc, View& v) {
        if (sc.job < 0) return false;
        const ColJob& j = B.jobs[(size_t)sc.job];
        v = View{B.ob.data() + j.out_off, B.oq.data() + j.out_off, B.od.data() + j.out_off, B.oe.data() + j.out_off, j.cons_len};
        if (sc.meth >= 0 && !B.mflag.empty()) {   // (the anchor is the longest read of the uncapped set: its annotation covers the consensus)
          const MethJob& mj = B.mjobs[(size_t)sc.meth];
          if (mj.n_pos >= j.cons_len) { v.mflag = B.mflag.data() + mj.out_off; v.mu = B.mu.data() + mj.out_off; v.mt = B.mt.data() + mj.out_off; }
        }
        return true;
      };
      View v_ab_r1, v_ab_r2, v_ba_r1, v_ba_r2;
      bool h1 = view(m.ab_r1, v_ab_r1), h2 = view(m.ab_r2, v_ab_r2), h3 = view(m.ba_r1, v_ba_r1), h4 = view(m.ba_r2, v_ba_r2);
      auto cons_ok = [&](const Duplex& d) { return x.min_reads_ok(d.ab_max(), d.has_ba ? d.ba_max() : 0); };
      bool emitted = false;
      std::vector<RawRef> empty;
      Duplex d1, d2;
      size_t mark = c->out_data.size();
      if (h1 && h2 && h3 && h4) {
        std::vector<uint32_t> r1s = m.ab_r1.src_rd, r2s = m.ab_r2.src_rd;
        r1s.insert(r1s.end(), m.ba_r2.src_rd.begin(), m.ba_r2.src_rd.end());
        r2s.insert(r2s.end(), m.ba_r1.src_rd.begin(), m.ba_r1.src_rd.end());
        bool k1 = duplex_consensus(B, &v_ab_r1, &v_ba_r2, r1s.empty() ? nullptr : &r1s, d1);
        bool k2 = duplex_consensus(B, &v_ab_r2, &v_ba_r1, r2s.empty() ? nullptr : &r2s, d2);
        if (k1 && k2 && cons_ok(d1) && cons_ok(d2)) {
          if (!duplex_read_into(x, c->out_data, d1, 1, m.base_mi, m.ab_r1.raws, m.ba_r2.raws, true, m) ||
              !duplex_read_into(x, c->out_data, d2, 2, m.base_mi, m.ab_r2.raws, m.ba_r1.raws, false, m)) { c->err = x.err; return 2; }
          emitted = true;
        }
      } else if (h1 && h2 && !h3 && !h4) {
        if (x.min_yx == 0 && duplex_consensus(B, &v_ab_r1, nullptr, nullptr, d1) && duplex_consensus(B, &v_ab_r2, nullptr, nullptr, d2) && cons_ok(d1) && cons_ok(d2)) {
          if (!duplex_read_into(x, c->out_data, d1, 1, m.base_mi, m.ab_r1.raws, empty, true, m) ||
              !duplex_read_into(x, c->out_data, d2, 2, m.base_mi, m.ab_r2.raws, empty, false, m)) { c->err = x.err; return 2; }
          emitted = true;
        }
      } else if (!h1 && !h2 && h3 && h4) {
        if (x.min_yx == 0 && duplex_consensus(B, nullptr, &v_ba_r1, nullptr, d1) && duplex_consensus(B, nullptr, &v_ba_r2, nullptr, d2) && cons_ok(d1) && cons_ok(d2)) {
          if (!duplex_read_into(x, c->out_data, d1, 1, m.base_mi, empty, m.ba_r1.raws, true, m) ||
              !duplex_read_into(x, c->out_data, d2, 2, m.base_mi, empty, m.ba_r2.raws, false, m)) { c->err = x.err; return 2; }
          emitted = true;
        }
      }
      (void)mark;
      if (emitted) { m.group_stats.consensus_reads += 2; count += 2; kept = true; }
      else { m.group_stats.reject(FGX_REJ_INSUFFICIENT_READS, m.a.size() + m.b.size()); kept = false; }
    }
    // fold the molecule's statistics (duplex_caller.rs:2587-2610, 1894-1926)
    HostStats s = m.caller_stats;
    auto merge = [](HostStats& dst, const HostStats& src) {
      dst.total_reads += src.total_reads; dst.consensus_reads += src.consensus_reads; dst.filtered_reads += src.filtered_reads;
      for (int i = 0; i < FGX_N_REJECTION; i++) dst.rej[i] += src.rej[i];
    };
    merge(s, m.group_stats);
    if (kept) merge(s, m.ss_stats);
    else {
      uint64_t ssr = m.ss_stats.filtered_reads;
      if (ssr != 0) {
        int reason = -1;
        for (int i = 0; i < FGX_N_REJECTION; i++) if (m.group_stats.rej[i]) { reason = i; break; }
        if (reason >= 0) { s.rej[reason] = s.rej[reason] > ssr ? s.rej[reason] - ssr : 0; s.filtered_reads = s.filtered_reads > ssr ? s.filtered_reads - ssr : 0; }
        merge(s, m.ss_stats);
      }
    }
    merge(batch, s);
    if (track) {
      for (auto& r : m.frag_rejects) reject_out(r.data(), r.size());
      if (kept) for (auto& r : m.ss_rejects) reject_out(r.data(), r.size());
      else { for (auto& r : m.a) reject_out(r.p, r.n); for (auto& r : m.b) reject_out(r.p, r.n); }
    }
    c->grp_out_end[g] = c->out_data.size();
  }
  auto t3 = clk::now();

  memset(out, 0, sizeof(*out));
  out->data = c->out_data.data(); out->data_len = c->out_data.size(); out->count = count;
  batch.to_array(out->stats);
  for (int i = 0; i < 4; i++) out->stats[24 + i] = ov[i];
  out->rejects = c->out_rejects.data(); out->rejects_len = c->out_rejects.size(); out->n_rejects = n_rejects;
  auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  out->ms_host_prep = ms(t0, t1); out->ms_kernels = ms_k; out->ms_h2d = ms(t1, t2) - ms_k; out->ms_emit = ms(t2, t3);
  return 0;
}

}  // namespace fgx
