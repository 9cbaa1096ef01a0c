// canon_core.h — canonical form of a duplex molecule whose reads carry indels / skips / pads, so that the device pipeline's duplex
// kernels (one aligned block per read: `k_family_wave<1>`, fastpath.hip) decide it instead of the host-orchestrated general path.
//
// Everything the reference derives from a read's CIGAR happens BEFORE the per-position arithmetic:
//   * the R1/R2 overlap pre-correction walks both CIGARs (`apply_overlapping_consensus`, overlapping.rs:236-336, 627-684;
//     src/lib/commands/duplex.rs:786-795 for when the duplex command applies it),
//   * the mate clip needs the read's CIGAR and the mate's (`num_bases_extending_past_mate_raw`, raw-bam/overlap.rs:181-268),
//   * the alignment filter compares simplified, reversed, truncated CIGARs (`filter_source_reads_by_alignment`,
//     vanilla_caller.rs:1242-1296; `select_most_common_alignment_group` :48-120).
// After these three, a source read is a string of bases and qualities of its final length and nothing else of its alignment is
// looked at (create_source_read :1080-1190, consensus_call :706-779, duplex_consensus duplex_caller.rs:931-1108).  So the molecule
// is rewritten, record by record, into one the fast kernels take and decide IDENTICALLY:
//   * bases / qualities carry the overlap correction; the pass is told not to correct again by moving R2 records to another
//     reference id (`overlap_call` returns before touching mates on different references) — consensus records are unmapped;
//   * the mate clip is applied physically (the clipped end of the read in sequencing orientation is cut: the tail of a forward
//     read's stored bases, the head of a reverse read's) and the MC tag is dropped, so the pass computes a clip of 0;
//   * reads the alignment filter rejects are dropped and counted here (MinorityAlignment); the survivors get the CIGAR `<len>M`,
//     under which the pass's own filter keeps all of them (one prefix-compatible group);
//   * everything else of the record (name: the downsampling rank; flags; MI / RX / cell tag) is copied.
// What the pass then reports for the canonical molecule, plus the `Delta` counted here, is what the reference reports for the
// original one (statistics included: see tests/test_canon_core.py, which checks exactly this through the oracle).
//
// Out of scope (status CANON_OUT_OF_SCOPE: the molecule stays on the general path): quality trimming (`--trim` looks at the clipped
// qualities), a per-strand cap that would bite, more than MAX_READS records / MAX_OPS CIGAR ops / MAX_GROUPS alignment groups,
// fragments, secondary / supplementary / unmapped records, a strand-orientation collision, records the reference refuses.
//
// Host + device source, scalar, no allocation: one GPU lane (or one host thread) per molecule.
#pragma once
#include <cstdint>
#include <cstring>
#include "bamrec.h"

#if defined(__HIPCC__)
#define CANON_HD __host__ __device__ inline
#else
#define CANON_HD inline
#endif

namespace fgx {
namespace canon {

constexpr uint32_t MAX_READS = 128;
constexpr uint32_t MAX_OPS = 16;
constexpr uint32_t MAX_GROUPS = 16;
constexpr int32_t OTHER_REF_XOR = 0x20000000;   // R2 records of the canonical molecule: ref_id ^ this

enum : int { CANON_OK = 0, CANON_OUT_OF_SCOPE = 1 };

struct Params {
  uint8_t min_bq;                 // min_input_base_quality
  uint8_t overlapping;            // overlapping_consensus option of the command
  uint8_t trim;                   // --trim: out of scope
  uint8_t _pad;
  char cell_tag[2];               // {0,0} = none
  uint8_t _pad2[2];
  uint32_t min_total, min_xy, min_yx;   // duplex min-reads triple
  int64_t max_reads_per_strand;   // -1 = none
};

struct SimpOp { uint8_t k; uint32_t len; };
struct ReadInfo {
  uint32_t len;          // l_seq
  uint32_t clip;         // bases extending past the mate
  uint32_t final_len;    // after masking, clip and trailing-N strip; 0 = zero length after trimming
  uint8_t strand;        // 0 = /A, 1 = /B
  uint8_t r2;            // 0 = R1, 1 = R2
  uint8_t keep;          // 0 = rejected by the alignment filter
  uint8_t n_simp;
  SimpOp simp[MAX_OPS];  // simplified CIGAR, reversed for reverse reads, truncated to final_len
};
struct Scratch {
  ReadInfo r[MAX_READS];
  uint32_t list[MAX_READS];      // one alignment-filter set, in source order, then sorted by length
  uint32_t mc_ops[MAX_OPS + 1];
  uint32_t ops[MAX_OPS];
};
struct Delta {
  uint64_t minority;             // reads dropped by the alignment filter (MinorityAlignment; also total_reads / filtered_reads)
  uint64_t ov[4];                // CorrectionStats of the overlap pre-step
};

CANON_HD void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
CANON_HD void wr16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }

// ---- the overlap pre-correction on two mutable records (host_common.h: aligned_positions + overlap_call, without the vectors) ----
struct AlnCursor {
  bam::Rec v;
  uint32_t n_ops, i;
  int64_t ref, q, k, lo, hi, rec_len;
  bool done;
  CANON_HD void init(const bam::Rec& rec, int64_t lo_, int64_t hi_) {
    v = rec; n_ops = v.n_cigar(); i = 0; ref = (int64_t)v.pos() + 1; q = 0; k = -1; lo = lo_; hi = hi_; rec_len = v.l_seq(); done = false;
  }
  // next aligned (query offset, reference position) inside [lo, hi]
  CANON_HD bool next(uint32_t* qoff, int64_t* rpos) {
    while (!done) {
      if (i >= n_ops || ref > hi || q >= rec_len) { done = true; return false; }
      const uint32_t op = v.cigar_op(i), t = op & 0xF;
      const int64_t len = op >> 4;
      if (t == 0 || t == 7 || t == 8) {
        if (k < 0) k = lo - ref > 0 ? lo - ref : 0;
        if (k < len && ref + k <= hi && q + k < rec_len) { *qoff = (uint32_t)(q + k); *rpos = ref + k; k++; return true; }
        ref += len; q += len;
      } else if (t == 1 || t == 4) q += len;
      else if (t == 2 || t == 3) ref += len;
      i++; k = -1;
    }
    return false;
  }
};

CANON_HD void set_code(uint8_t* rec, uint32_t seq_off, uint32_t pos, uint8_t code) {
  uint8_t& b = rec[seq_off + (pos >> 1)];
  b = (pos & 1) ? (uint8_t)((b & 0xF0) | code) : (uint8_t)((code << 4) | (b & 0x0F));
}

CANON_HD void overlap_pair(uint8_t* a, uint32_t an, uint8_t* b, uint32_t bn, uint32_t* ops_scratch, uint64_t* st) {
  bam::Rec v1{a, an}, v2{b, bn};
  if ((v1.flags() | v2.flags()) & bam::F_UNMAPPED) return;
  if (v1.ref_id() != v2.ref_id()) return;
  if (v1.pos() < 0 || v2.pos() < 0) return;
  auto ref_len = [&](const bam::Rec& v) -> int32_t {
    const uint32_t n = v.n_cigar();
    if ((uint64_t)v.cigar_off() + 4ull * n > v.len) return 0;          // cigar_ops_vec: out of bounds → empty
    for (uint32_t i = 0; i < n; i++) ops_scratch[i] = v.cigar_op(i);
    return bam::ref_len_checked0(ops_scratch, n);
  };
  const int32_t rl1 = ref_len(v1), rl2 = ref_len(v2);
  if (rl1 == 0 || rl2 == 0) return;
  const int64_t s1 = (int64_t)v1.pos() + 1, e1 = (int64_t)v1.pos() + rl1, s2 = (int64_t)v2.pos() + 1, e2 = (int64_t)v2.pos() + rl2;
  const int64_t lo = s1 > s2 ? s1 : s2, hi = e1 < e2 ? e1 : e2;
  AlnCursor c1, c2;
  c1.init(v1, lo, hi);
  c2.init(v2, lo, hi);
  uint8_t* q1 = a + v1.qual_off();
  uint8_t* q2 = b + v2.qual_off();
  const uint32_t so1 = v1.seq_off(), so2 = v2.seq_off();
  uint32_t x = 0, y = 0;
  int64_t p1 = 0, p2 = 0;
  bool h1 = c1.next(&x, &p1), h2 = c2.next(&y, &p2);
  while (h1 && h2) {
    if (p1 < p2) { h1 = c1.next(&x, &p1); continue; }
    if (p1 > p2) { h2 = c2.next(&y, &p2); continue; }
    const uint8_t k1 = v1.base_code(x), k2 = v2.base_code(y);
    if (k1 != 15 && k2 != 15) {
      st[0]++;
      const uint8_t qa = q1[x], qb = q2[y];
      if (k1 == k2) {
        st[1]++;
        const unsigned s = (unsigned)qa + qb;
        const uint8_t nq = (uint8_t)(s < 93 ? s : 93);
        q1[x] = nq; q2[y] = nq;
        if (nq != qa || nq != qb) st[3]++;
      } else {
        st[2]++;
        uint8_t cb, cq;
        if (qa == qb) { cb = 15; cq = 2; }
        else if (qa > qb) { cb = k1; cq = (uint8_t)(qa - qb) > 2 ? (uint8_t)(qa - qb) : (uint8_t)2; }
        else { cb = k2; cq = (uint8_t)(qb - qa) > 2 ? (uint8_t)(qb - qa) : (uint8_t)2; }
        set_code(a, so1, x, cb);
        set_code(b, so2, y, cb);
        q1[x] = cq; q2[y] = cq;
        st[3] += 2;
      }
    }
    h1 = c1.next(&x, &p1);
    h2 = c2.next(&y, &p2);
  }
}

CANON_HD bool names_equal(const bam::Rec& a, const bam::Rec& b) {
  const uint32_t n = a.name_len();
  if (n != b.name_len()) return false;
  for (uint32_t i = 0; i < n; i++) if (a.name()[i] != b.name()[i]) return false;
  return true;
}

// ---- simplified CIGARs (host_common.h: simplify_cigar / truncate_cigar / cigar_is_prefix / cigar_cmp) ------------------------------
CANON_HD bool simp_is_prefix(const SimpOp* a, uint32_t na, const SimpOp* b, uint32_t nb) {
  if (na > nb) return false;
  for (uint32_t i = 0; i < na; i++) {
    if (a[i].k != b[i].k) return false;
    if (i + 1 == na) { if (a[i].len > b[i].len) return false; }
    else if (a[i].len != b[i].len) return false;
  }
  return true;
}
CANON_HD int simp_cmp(const SimpOp* a, uint32_t na, const SimpOp* b, uint32_t nb) {
  const uint32_t n = na < nb ? na : nb;
  for (uint32_t i = 0; i < n; i++) {
    if (a[i].len != b[i].len) return a[i].len < b[i].len ? -1 : 1;
    if (a[i].k != b[i].k) return a[i].k < b[i].k ? -1 : 1;
  }
  return na == nb ? 0 : (na < nb ? -1 : 1);
}

// The alignment filter over S.list[0 .. n) (indices into S.r, source order).  Clears `keep` of the rejected reads; returns their
// count, or -1 when there are more than MAX_GROUPS groups.
CANON_HD int alignment_filter(Scratch& S, uint32_t n) {
  if (n < 2) return 0;
  // stable sort by length, longest first (insertion sort: n <= 128)
  for (uint32_t i = 1; i < n; i++) {
    const uint32_t v = S.list[i];
    uint32_t j = i;
    while (j > 0 && S.r[S.list[j - 1]].final_len < S.r[v].final_len) { S.list[j] = S.list[j - 1]; j--; }
    S.list[j] = v;
  }
  uint32_t g_rep[MAX_GROUPS], g_size[MAX_GROUPS];
  uint64_t g_mem[MAX_GROUPS][2];
  uint32_t ng = 0;
  for (uint32_t i = 0; i < n; i++) {
    const ReadInfo& R = S.r[S.list[i]];
    bool found = false;
    for (uint32_t g = 0; g < ng; g++) {
      const ReadInfo& G = S.r[S.list[g_rep[g]]];
      if (simp_is_prefix(R.simp, R.n_simp, G.simp, G.n_simp)) { g_mem[g][i >> 6] |= 1ull << (i & 63); g_size[g]++; found = true; }   // no break (fgbio)
    }
    if (!found) {
      if (ng == MAX_GROUPS) return -1;
      g_rep[ng] = i; g_size[ng] = 1; g_mem[ng][0] = g_mem[ng][1] = 0; g_mem[ng][i >> 6] |= 1ull << (i & 63);
      ng++;
    }
  }
  uint32_t best = 0;   // max_by: larger group, then smaller CIGAR; the later element wins exact ties
  for (uint32_t g = 1; g < ng; g++) {
    int c;
    if (g_size[best] != g_size[g]) c = g_size[best] < g_size[g] ? -1 : 1;
    else {
      const ReadInfo& A = S.r[S.list[g_rep[g]]];
      const ReadInfo& B = S.r[S.list[g_rep[best]]];
      c = simp_cmp(A.simp, A.n_simp, B.simp, B.n_simp);
    }
    if (c <= 0) best = g;
  }
  int rejected = 0;
  for (uint32_t i = 0; i < n; i++)
    if (!((g_mem[best][i >> 6] >> (i & 63)) & 1)) { S.r[S.list[i]].keep = 0; rejected++; }
  return rejected;
}

// MI strand of a record: 0 = /A, 1 = /B, -1 = none of them / no MI
CANON_HD int mi_strand(const bam::Rec& v) {
  const uint32_t an = v.len > v.aux_off() ? v.len - v.aux_off() : 0;
  uint32_t vl = 0;
  const int64_t off = bam::find_z_tag(v.b + v.aux_off(), an, 'M', 'I', &vl);
  if (off < 0 || vl < 2) return -1;
  const uint8_t* p = v.b + v.aux_off() + off;
  if (p[vl - 2] != '/') return -1;
  return p[vl - 1] == 'A' ? 0 : p[vl - 1] == 'B' ? 1 : -1;
}

// Rewrites record `rec` (length n, already overlap-corrected) in place into its canonical form; returns the new length.
CANON_HD uint32_t rewrite_record(uint8_t* rec, uint32_t n, uint32_t clip, bool move_ref) {
  bam::Rec v{rec, n};
  const uint32_t l = v.l_seq(), name_l = v.l_read_name();
  const bool rev = (v.flags() & bam::F_REVERSE) != 0;
  const uint32_t newl = l > clip ? l - clip : 0;
  const uint32_t s0 = rev ? (l - newl) : 0;                      // first stored base that stays
  const uint32_t old_seq = v.seq_off(), old_qual = v.qual_off(), old_aux = v.aux_off();
  const uint32_t new_cig = 32 + name_l, new_seq = new_cig + (newl ? 4u : 0u), new_qual = new_seq + (newl + 1) / 2, new_aux = new_qual + newl;
  if (move_ref) wr32(rec, (uint32_t)((int32_t)bam::rd32(rec) ^ OTHER_REF_XOR));
  // sequence first (reads run ahead of the writes: the new fields start no later than the old ones), then qualities, then aux
  for (uint32_t j = 0; j < (newl + 1) / 2; j++) {
    const uint32_t i0 = 2 * j, i1 = 2 * j + 1;
    auto code = [&](uint32_t i) -> uint8_t { const uint8_t b = rec[old_seq + ((s0 + i) >> 1)]; return ((s0 + i) & 1) ? (uint8_t)(b & 0xF) : (uint8_t)(b >> 4); };
    const uint8_t hi = code(i0), lo = i1 < newl ? code(i1) : (uint8_t)0;
    rec[new_seq + j] = (uint8_t)((hi << 4) | lo);
  }
  for (uint32_t i = 0; i < newl; i++) rec[new_qual + i] = rec[old_qual + s0 + i];
  // aux: every tag but MC
  uint32_t w = new_aux, p = old_aux;
  while (p + 3 <= n) {
    const uint8_t t0 = rec[p], t1 = rec[p + 1], ty = rec[p + 2];
    uint32_t sz;
    if (ty == 'Z' || ty == 'H') { const int64_t e = bam::find_nul(rec + p + 3, n - (p + 3)); if (e < 0) break; sz = 3 + (uint32_t)e + 1; }
    else if (ty == 'B') { if (p + 8 > n) break; const int es = bam::tag_fixed_size(rec[p + 3]); if (es <= 0) break; sz = 8 + (uint32_t)es * bam::rd32(rec + p + 4); }
    else { const int fs = bam::tag_fixed_size(ty); if (fs <= 0) break; sz = 3 + (uint32_t)fs; }
    if ((uint64_t)p + sz > n) break;
    if (!(t0 == 'M' && t1 == 'C')) { for (uint32_t i = 0; i < sz; i++) rec[w + i] = rec[p + i]; w += sz; }
    p += sz;
  }
  // (the header fields are written last: the loops above read the old layout through saved offsets only)
  if (newl) wr32(rec + new_cig, newl << 4);
  wr16(rec + 12, (uint16_t)(newl ? 1 : 0));
  wr32(rec + 16, newl);
  return w;
}

// One duplex molecule: records `n` at rec_off / rec_len in `blob` → canonical records in `out` at out_off[i] (room for rec_len[i]
// bytes each), out_len[i] = new length, 0 = the record is dropped.  Returns CANON_OK or CANON_OUT_OF_SCOPE (outputs undefined).
CANON_HD int canon_duplex_molecule(const Params& P, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out,
                                   const uint64_t* out_off, uint32_t* out_len, Scratch& S, Delta& D) {
  D.minority = 0; D.ov[0] = D.ov[1] = D.ov[2] = D.ov[3] = 0;
  if (n == 0 || n > MAX_READS || P.trim) return CANON_OUT_OF_SCOPE;
  if (P.min_xy > P.min_total || P.min_yx > P.min_xy) return CANON_OUT_OF_SCOPE;          // the reference refuses the options
  bool has_a = false, has_b = false;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t len = rec_len[i];
    if (len < 32) return CANON_OUT_OF_SCOPE;
    bam::Rec v{blob + rec_off[i], len};
    const uint16_t f = v.flags();
    const uint32_t nc = v.n_cigar(), l = v.l_seq();
    if (v.l_read_name() == 0 || (uint64_t)v.aux_off() > len) return CANON_OUT_OF_SCOPE;
    if (!(f & bam::F_PAIRED) || (f & (bam::F_UNMAPPED | bam::F_SECONDARY | bam::F_SUPPLEMENTARY))) return CANON_OUT_OF_SCOPE;
    if (((f & bam::F_FIRST) != 0) == ((f & bam::F_LAST) != 0)) return CANON_OUT_OF_SCOPE;
    if (nc == 0 || nc > MAX_OPS) return CANON_OUT_OF_SCOPE;
    uint64_t ql = 0;
    for (uint32_t k = 0; k < nc; k++) { const uint32_t op = v.cigar_op(k); if ((op & 0xF) > 8) return CANON_OUT_OF_SCOPE; if (bam::op_consumes_query(op & 0xF)) ql += op >> 4; }
    if (ql != l) return CANON_OUT_OF_SCOPE;
    if (l) { bool all_ff = true; const uint8_t* q = v.b + v.qual_off(); for (uint32_t k = 0; k < l; k++) if (q[k] != 0xFF) { all_ff = false; break; } if (all_ff) return CANON_OUT_OF_SCOPE; }
    const int st = mi_strand(v);
    if (st < 0) return CANON_OUT_OF_SCOPE;
    (st == 0 ? has_a : has_b) = true;
    ReadInfo& R = S.r[i];
    R.len = l; R.strand = (uint8_t)st; R.r2 = (f & bam::F_LAST) ? 1 : 0; R.keep = 1; R.clip = 0; R.final_len = 0; R.n_simp = 0;
    // working copy
    uint8_t* w = out + out_off[i];
    for (uint32_t k = 0; k < len; k++) w[k] = v.b[k];
    out_len[i] = len;
  }
  // strand-orientation collision (duplex_caller.rs:2007-2040): decided over ALL reads, so such molecules stay where they are
  if (has_a && has_b) {
    for (int set = 0; set < 2; set++) {          // set 0: AB-R1 ++ BA-R2, set 1: AB-R2 ++ BA-R1
      bool have = false, first_rev = false;
      for (int pass = 0; pass < 2; pass++)
        for (uint32_t i = 0; i < n; i++) {
          const ReadInfo& R = S.r[i];
          const bool in_set = (R.strand == 0) ? (R.r2 == set) : (R.r2 != set);
          if (!in_set || R.strand != pass) continue;
          const bool rv = (bam::Rec{blob + rec_off[i], rec_len[i]}.flags() & bam::F_REVERSE) != 0;
          if (!have) { have = true; first_rev = rv; } else if (rv != first_rev) return CANON_OUT_OF_SCOPE;
        }
    }
  }
  // the overlap pre-step (duplex.rs:786-795; apply_overlapping_consensus pairs the LAST R1 and the LAST R2 of a name)
  if (P.overlapping && (P.min_yx == 0 || (n >= 2 && has_a && has_b))) {
    for (uint32_t i = 0; i < n; i++) {
      bam::Rec vi{out + out_off[i], rec_len[i]};
      bool seen = false;
      for (uint32_t j = 0; j < i && !seen; j++) seen = names_equal(vi, bam::Rec{out + out_off[j], rec_len[j]});
      if (seen) continue;
      int64_t r1 = -1, r2 = -1;
      for (uint32_t j = i; j < n; j++) {
        bam::Rec vj{out + out_off[j], rec_len[j]};
        if (!names_equal(vi, vj)) continue;
        if (vj.flags() & bam::F_FIRST) r1 = j; else r2 = j;
      }
      if (r1 >= 0 && r2 >= 0) overlap_pair(out + out_off[r1], rec_len[r1], out + out_off[r2], rec_len[r2], S.ops, D.ov);
    }
  }
  // source-read lengths: mate clip, masking, trailing no-calls (create_source_read, vanilla_caller.rs:1080-1190)
  uint32_t na = 0, nb = 0;
  for (uint32_t i = 0; i < n; i++) {
    ReadInfo& R = S.r[i];
    bam::Rec v{out + out_off[i], rec_len[i]};
    if (!R.r2) (R.strand == 0 ? na : nb)++;
    const uint32_t nc = v.n_cigar();
    for (uint32_t k = 0; k < nc; k++) S.ops[k] = v.cigar_op(k);
    const uint32_t an = v.len > v.aux_off() ? v.len - v.aux_off() : 0;
    uint32_t mcl = 0;
    const int64_t mco = bam::find_z_tag(v.b + v.aux_off(), an, 'M', 'C', &mcl);
    bool overflow = false;
    const uint64_t clip = bam::mate_clip(v, S.ops, nc, mco >= 0 ? v.b + v.aux_off() + mco : nullptr, mcl, S.mc_ops, MAX_OPS + 1, &overflow);
    if (overflow) return CANON_OUT_OF_SCOPE;
    R.clip = clip > R.len ? R.len : (uint32_t)clip;
    const bool rev = (v.flags() & bam::F_REVERSE) != 0;
    uint32_t fl = R.len - R.clip;
    const uint8_t* q = v.b + v.qual_off();
    while (fl > 0) {                                   // oriented position fl-1 = stored position (rev ? len - fl : fl - 1)
      const uint32_t s = rev ? R.len - fl : fl - 1;
      if (v.base_code(s) == 15 || q[s] < P.min_bq) fl--; else break;
    }
    R.final_len = fl;
    // simplified CIGAR: S, H, =, X → M, adjacent ops merged; reversed for reverse reads; truncated to final_len
    SimpOp tmp[MAX_OPS];
    uint32_t nt = 0;
    for (uint32_t k = 0; k < nc; k++) {
      const uint32_t t = S.ops[k] & 0xF;
      const uint8_t kk = (t == 4 || t == 5 || t == 7 || t == 8) ? (uint8_t)0 : (uint8_t)t;
      if (nt && tmp[nt - 1].k == kk) tmp[nt - 1].len += S.ops[k] >> 4;
      else { tmp[nt].k = kk; tmp[nt].len = S.ops[k] >> 4; nt++; }
    }
    uint32_t remaining = fl;
    R.n_simp = 0;
    for (uint32_t k = 0; k < nt && remaining > 0; k++) {
      const SimpOp& op = tmp[rev ? nt - 1 - k : k];
      if (op.k == 0 || op.k == 1) { const uint32_t take = op.len < remaining ? op.len : remaining; R.simp[R.n_simp].k = op.k; R.simp[R.n_simp].len = take; R.n_simp++; remaining -= take; }
      else { R.simp[R.n_simp] = op; R.n_simp++; }
    }
  }
  // the alignment filter, per output end, when the molecule passes the read-count gate (a molecule that fails it is rejected whole
  // before any read is looked at: duplex_caller.rs:1985-1996)
  const uint64_t xy = na > nb ? na : nb, yx = na > nb ? nb : na;
  const bool gate = P.min_total <= xy + yx && P.min_xy <= xy && P.min_yx <= yx;
  if (gate) {
    for (int set = 0; set < 2; set++) {
      uint32_t m = 0;
      for (int pass = 0; pass < 2; pass++)             // AB reads first, then BA reads, each in input order
        for (uint32_t i = 0; i < n; i++) {
          const ReadInfo& R = S.r[i];
          const bool in_set = (R.strand == 0) ? (R.r2 == set) : (R.r2 != set);
          if (in_set && R.strand == pass && R.final_len > 0) S.list[m++] = i;
        }
      const int rej = alignment_filter(S, m);
      if (rej < 0) return CANON_OUT_OF_SCOPE;
      D.minority += (uint64_t)rej;
    }
    if (D.minority) {   // the canonical molecule must pass the read-count gate too: failing it there would hide the zero-length reads'
      uint64_t ka = 0, kb = 0;                         // own rejections (the molecule is rejected either way: leave it where it is)
      for (uint32_t i = 0; i < n; i++) if (S.r[i].keep && !S.r[i].r2) (S.r[i].strand == 0 ? ka : kb)++;
      const uint64_t kxy = ka > kb ? ka : kb, kyx = ka > kb ? kb : ka;
      if (!(P.min_total <= kxy + kyx && P.min_xy <= kxy && P.min_yx <= kyx)) return CANON_OUT_OF_SCOPE;
    }
    if (P.max_reads_per_strand >= 0) {                 // a cap that bites shapes the consensus but not the error recount: general path
      uint64_t cnt[2][2] = {{0, 0}, {0, 0}};
      for (uint32_t i = 0; i < n; i++) if (S.r[i].keep && S.r[i].final_len > 0) cnt[S.r[i].strand][S.r[i].r2]++;
      for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) if (cnt[a][b] > (uint64_t)P.max_reads_per_strand) return CANON_OUT_OF_SCOPE;
    }
  }
  // the cell barcode is read from the molecule's FIRST /A record (the first /B record when there is none: duplex_caller.rs:1998-2006):
  // when the filter drops that record, its successor must carry the same value
  if (P.cell_tag[0] && D.minority) {
    const uint8_t want_strand = has_a ? 0 : 1;
    int64_t first = -1, first_kept = -1;
    for (uint32_t i = 0; i < n; i++) if (S.r[i].strand == want_strand) { if (first < 0) first = i; if (first_kept < 0 && S.r[i].keep) first_kept = i; }
    if (first >= 0 && first != first_kept) {
      if (first_kept < 0) return CANON_OUT_OF_SCOPE;
      bam::Rec a{out + out_off[first], rec_len[first]}, b{out + out_off[first_kept], rec_len[first_kept]};
      uint32_t la = 0, lb = 0;
      const int64_t oa = bam::find_z_tag(a.b + a.aux_off(), a.len - a.aux_off(), (uint8_t)P.cell_tag[0], (uint8_t)P.cell_tag[1], &la);
      const int64_t ob = bam::find_z_tag(b.b + b.aux_off(), b.len - b.aux_off(), (uint8_t)P.cell_tag[0], (uint8_t)P.cell_tag[1], &lb);
      if ((oa < 0) != (ob < 0) || (oa >= 0 && la != lb)) return CANON_OUT_OF_SCOPE;
      if (oa >= 0) for (uint32_t k = 0; k < la; k++) if (a.b[a.aux_off() + oa + k] != b.b[b.aux_off() + ob + k]) return CANON_OUT_OF_SCOPE;
    }
  }
  // the canonical records
  for (uint32_t i = 0; i < n; i++) {
    if (!S.r[i].keep) { out_len[i] = 0; continue; }
    out_len[i] = rewrite_record(out + out_off[i], rec_len[i], S.r[i].clip, S.r[i].r2 != 0);
  }
  return CANON_OK;
}


// =====================================================================================================================================
// CODEC molecules (codec_caller.rs:625-1262).  The CODEC caller reads a CIGAR in four places: the virtual hard clip against the mate in
// hand (`clip_cigar_ops_raw`, raw-bam/cigar.rs:404-446, by `num_bases_extending_past_mate_vs_mate_raw`), the per-strand alignment
// filter, and the overlap GEOMETRY of the longest alignment of each strand — the shared reference window, the phase check
// (`read_pos_at_ref_pos`, cigar.rs:461-500) and the consensus length = query position of the window's end in the forward read + length
// of the reverse read - query position of that end in the reverse read.  The source reads themselves are the clipped, oriented bases.
// A canonical molecule keeps every record (the consensus UMI is called over ALL records of the group, so nothing may be dropped: a
// molecule whose filter rejects a read is out of scope), cuts each read by its clip, gives it `<len>M`, and PLACES the reads so that the
// pass recomputes the same geometry: each template's reverse read starts d >= max(0, len_fwd - len_rev) after its forward read (an FR
// pair with nothing extending past the mate: clip 0), and the two reads the pass will pick as longest sit `cons_len - len_rev` apart, so
// that the consensus length comes out the same; the phase check holds trivially for single-block reads.  Every decision the original
// takes before the per-position work is taken here first; anything but "emit" leaves the molecule where it is (the general path decides
// rejected molecules cheaply).
struct CodecParams {
  uint32_t min_reads_per_strand;
  uint32_t min_duplex_length;
  int64_t max_reads_per_strand;   // -1 = none
};
struct CodecInfo {                // ClippedRecordInfo (codec_caller.rs:323-336)
  uint32_t rec, mate;             // record index, its mate's
  uint32_t clip, keep;            // clipped bases, bases left
  uint32_t n_ops;
  uint32_t ops[MAX_OPS + 2];      // clipped CIGAR
  uint64_t adj_pos;               // 1-based start after the clip
  uint8_t reverse, first;
  uint32_t canon_pos;             // 1-based start of the canonical record
};
struct CodecScratch {
  CodecInfo r[MAX_READS];
  uint32_t r1[MAX_READS / 2], r2[MAX_READS / 2];   // per template: index into r of its R1 / R2
  Scratch filt;                   // the alignment filter's lists
};

CANON_HD bool cq_consumes_read(uint32_t t) { return t == 0 || t == 1 || t == 7 || t == 8; }   // (clip walk: S / H are handled apart)
CANON_HD uint32_t cq_enc(uint32_t t, uint64_t len) { return ((uint32_t)len << 4) | t; }

// clip_cigar_ops_raw (cigar.rs:404-446 with its helpers :669-922): hard-clip `clip` query bases off one end.  Returns the op count, or
// -1 when the result does not fit.
CANON_HD int clip_cigar(const uint32_t* ops, uint32_t n, uint64_t clip, bool from_start, uint32_t* out, uint32_t cap, uint64_t* ref_consumed) {
  *ref_consumed = 0;
  if (clip == 0 || n == 0) { if (n > cap) return -1; for (uint32_t i = 0; i < n; i++) out[i] = ops[i]; return (int)n; }
  auto at = [&](uint32_t k) { return from_start ? ops[k] : ops[n - 1 - k]; };
  uint64_t existing = 0;
  for (uint32_t k = 0; k < n; k++) { const uint32_t t = at(k) & 0xF; if (t != 4 && t != 5) break; existing += at(k) >> 4; }
  uint64_t hard = 0, soft = 0;
  uint32_t skip = 0;
  while (skip < n && (at(skip) & 0xF) == 5) { hard += at(skip) >> 4; skip++; }
  while (skip < n && (at(skip) & 0xF) == 4) { soft += at(skip) >> 4; skip++; }
  uint32_t m = 0;
  auto push = [&](uint32_t v) { if (m < cap) out[m] = v; m++; };
  if (clip <= existing) {   // upgrade_clipping_raw: soft -> hard, alignment untouched
    const uint64_t room = clip > hard ? clip - hard : 0, up = soft < room ? soft : room;
    if (from_start) {
      push(cq_enc(5, hard + up));
      if (soft - up) push(cq_enc(4, soft - up));
      for (uint32_t k = skip; k < n; k++) push(ops[k]);
    } else {
      for (uint32_t k = 0; k < n - skip; k++) push(ops[k]);
      if (soft - up) push(cq_enc(4, soft - up));
      push(cq_enc(5, hard + up));
    }
    return m <= cap ? (int)m : -1;
  }
  const uint64_t want = clip - existing;
  uint64_t got = 0;
  uint32_t kept[MAX_OPS + 2], nk = 0;                     // ops that survive next to the clip, in walking order
  const uint32_t lo = from_start ? skip : 0, hi = from_start ? n : n - skip;   // the unclipped middle [lo, hi)
  uint32_t taken = 0;
  while (lo + taken < hi) {
    const uint32_t op = from_start ? ops[lo + taken] : ops[hi - 1 - taken];
    const uint32_t t = op & 0xF;
    const uint64_t len = op >> 4;
    if (got == want && nk == 0 && t == 2) { if (from_start) *ref_consumed += len; taken++; continue; }   // a deletion at the boundary
    if (got >= want) break;
    const bool is_read = cq_consumes_read(t), is_ref = bam::op_consumes_ref(t);
    if (is_read && len > want - got) {
      if (t == 1) got += len;                               // an insertion at the boundary goes whole
      else {
        const uint64_t part = want - got;
        got += part;
        if (is_ref && from_start) *ref_consumed += part;
        if (nk < MAX_OPS + 2) kept[nk] = cq_enc(t, len - part);
        nk++;
      }
    } else {
      if (is_read) got += len;
      if (is_ref && from_start) *ref_consumed += len;
    }
    taken++;
  }
  if (nk > MAX_OPS + 2) return -1;
  const uint64_t total_hard = hard + soft + got;
  if (from_start) {
    push(cq_enc(5, total_hard));
    for (uint32_t k = 0; k < nk; k++) push(kept[k]);
    for (uint32_t k = lo + taken; k < n; k++) push(ops[k]);
  } else {
    for (uint32_t k = 0; k < hi - taken; k++) push(ops[k]);
    for (uint32_t k = nk; k-- > 0;) push(kept[k]);
    push(cq_enc(5, total_hard));
  }
  return m <= cap ? (int)m : -1;
}

CANON_HD int32_t ref_len_wrapping(const uint32_t* ops, uint32_t n) {   // reference_length_from_cigar cigar.rs:137-150
  uint32_t r = 0;
  for (uint32_t i = 0; i < n; i++) if (bam::op_consumes_ref(ops[i] & 0xF)) r += ops[i] >> 4;
  return (int32_t)r;
}

// read_pos_at_ref_pos_raw (cigar.rs:461-500): 1-based query position at a 1-based reference position
CANON_HD bool read_pos_at(const uint32_t* ops, uint32_t n, uint64_t aln_start, uint64_t ref_pos, bool last_if_deleted, uint64_t* out) {
  if (ref_pos < aln_start) return false;
  uint64_t ref_off = 0, q_off = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t t = ops[i] & 0xF;
    const uint64_t len = ops[i] >> 4;
    if (bam::op_consumes_ref(t)) {
      const uint64_t s = aln_start + ref_off, e = s + len - 1;
      if (ref_pos >= s && ref_pos <= e) {
        if (bam::op_consumes_query(t)) { *out = q_off + (ref_pos - s) + 1; return true; }
        if (last_if_deleted) { *out = q_off > 0 ? q_off : 1; return true; }
        return false;
      }
      ref_off += len;
    }
    if (bam::op_consumes_query(t)) q_off += len;
  }
  return false;
}

// is_primary_fr_pair_raw (raw-bam/overlap.rs:83-108) and num_bases_extending_past_mate_vs_mate_raw (:223-230)
CANON_HD bool primary_fr_pair(const bam::Rec& a, const uint32_t* a_ops, const bam::Rec& b, const uint32_t* b_ops) {
  const uint16_t fa = a.flags(), fb = b.flags();
  if ((fa | fb) & bam::F_UNMAPPED) return false;
  if ((fa | fb) & bam::F_MATE_UNMAPPED) return false;
  if (a.ref_id() != b.ref_id()) return false;
  const bool ar = (fa & bam::F_REVERSE) != 0, br = (fb & bam::F_REVERSE) != 0;
  if (ar == br) return false;
  return ar ? bam::is_fr_pair(a, a_ops, a.n_cigar()) : bam::is_fr_pair(b, b_ops, b.n_cigar());
}

// One CODEC molecule; same contract as canon_duplex_molecule (every record is kept: out_len[i] > 0 on CANON_OK).
CANON_HD int canon_codec_molecule(const CodecParams& P, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out,
                                  const uint64_t* out_off, uint32_t* out_len, CodecScratch& S) {
  if (n < 2 || n > MAX_READS || (n & 1)) return CANON_OUT_OF_SCOPE;
  // every record: paired, primary, mapped with a mapped mate, a CIGAR that fits and spans the read, qualities present; MI on the first
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t len = rec_len[i];
    if (len < 32) return CANON_OUT_OF_SCOPE;
    bam::Rec v{blob + rec_off[i], len};
    const uint16_t f = v.flags();
    const uint32_t nc = v.n_cigar(), l = v.l_seq();
    if (v.l_read_name() == 0 || (uint64_t)v.aux_off() > len) return CANON_OUT_OF_SCOPE;
    if (!(f & bam::F_PAIRED) || (f & (bam::F_UNMAPPED | bam::F_MATE_UNMAPPED | bam::F_SECONDARY | bam::F_SUPPLEMENTARY))) return CANON_OUT_OF_SCOPE;
    if (((f & bam::F_FIRST) != 0) == ((f & bam::F_LAST) != 0)) return CANON_OUT_OF_SCOPE;
    if (nc == 0 || nc > MAX_OPS || l == 0 || v.pos() < 0) return CANON_OUT_OF_SCOPE;
    uint64_t ql = 0;
    for (uint32_t k = 0; k < nc; k++) { const uint32_t op = v.cigar_op(k); if ((op & 0xF) > 8) return CANON_OUT_OF_SCOPE; if (bam::op_consumes_query(op & 0xF)) ql += op >> 4; }
    if (ql != l) return CANON_OUT_OF_SCOPE;
    S.r[i].rec = i; S.r[i].mate = 0xFFFFFFFFu;
  }
  {
    bam::Rec v0{blob + rec_off[0], rec_len[0]};
    uint32_t vl = 0;
    if (bam::find_z_tag(v0.b + v0.aux_off(), v0.len - v0.aux_off(), 'M', 'I', &vl) < 0) return CANON_OUT_OF_SCOPE;   // (named by a running counter otherwise)
  }
  // templates in first-appearance order: exactly one R1 and one R2 of a name, a primary FR pair
  uint32_t nt = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (S.r[i].mate != 0xFFFFFFFFu) continue;
    bam::Rec vi{blob + rec_off[i], rec_len[i]};
    uint32_t j_found = 0xFFFFFFFFu;
    for (uint32_t j = i + 1; j < n; j++) {
      if (!names_equal(vi, bam::Rec{blob + rec_off[j], rec_len[j]})) continue;
      if (j_found != 0xFFFFFFFFu || S.r[j].mate != 0xFFFFFFFFu) return CANON_OUT_OF_SCOPE;      // three records of a name
      j_found = j;
    }
    if (j_found == 0xFFFFFFFFu) return CANON_OUT_OF_SCOPE;
    S.r[i].mate = j_found; S.r[j_found].mate = i;
    const bool i_first = (vi.flags() & bam::F_FIRST) != 0;
    const bool j_first = (bam::Rec{blob + rec_off[j_found], rec_len[j_found]}.flags() & bam::F_FIRST) != 0;
    if (i_first == j_first) return CANON_OUT_OF_SCOPE;
    S.r1[nt] = i_first ? i : j_found; S.r2[nt] = i_first ? j_found : i;
    nt++;
  }
  if (nt < P.min_reads_per_strand) return CANON_OUT_OF_SCOPE;
  if (P.max_reads_per_strand >= 0 && (uint64_t)nt > (uint64_t)P.max_reads_per_strand) return CANON_OUT_OF_SCOPE;
  // the virtual clip of every read against its mate, and its ClippedRecordInfo
  uint32_t opsa[MAX_OPS], opsb[MAX_OPS];
  for (uint32_t t = 0; t < nt; t++) {
    const uint32_t i1 = S.r1[t], i2 = S.r2[t];
    bam::Rec a{blob + rec_off[i1], rec_len[i1]}, b{blob + rec_off[i2], rec_len[i2]};
    for (uint32_t k = 0; k < a.n_cigar(); k++) opsa[k] = a.cigar_op(k);
    for (uint32_t k = 0; k < b.n_cigar(); k++) opsb[k] = b.cigar_op(k);
    if (!primary_fr_pair(a, opsa, b, opsb)) return CANON_OUT_OF_SCOPE;
    for (int side = 0; side < 2; side++) {
      const bam::Rec& v = side ? b : a;
      const bam::Rec& mt = side ? a : b;
      const uint32_t* vo = side ? opsb : opsa;
      const uint32_t* mo = side ? opsa : opsb;
      CodecInfo& I = S.r[side ? i2 : i1];
      const bool rev = (v.flags() & bam::F_REVERSE) != 0;
      const uint64_t clip = bam::past_mate_ops(rev, (int32_t)((uint32_t)v.pos() + 1u), vo, v.n_cigar(), (int32_t)((uint32_t)mt.pos() + 1u), mo, mt.n_cigar());
      const uint32_t l = v.l_seq();
      I.clip = clip > l ? l : (uint32_t)clip;
      I.keep = l - I.clip;
      I.reverse = rev; I.first = (v.flags() & bam::F_FIRST) != 0;
      uint64_t ref_consumed = 0;
      const int no = clip_cigar(vo, v.n_cigar(), clip, rev, I.ops, MAX_OPS + 2, &ref_consumed);
      if (no < 0) return CANON_OUT_OF_SCOPE;
      I.n_ops = (uint32_t)no;
      const uint64_t p1 = (uint64_t)(int64_t)(v.pos() + 1);
      I.adj_pos = rev ? p1 + ref_consumed : p1;
      if (I.keep < 2) return CANON_OUT_OF_SCOPE;
    }
  }
  // one orientation per strand
  const bool r1_neg = S.r[S.r1[0]].reverse != 0;
  for (uint32_t t = 0; t < nt; t++) if ((S.r[S.r1[t]].reverse != 0) != r1_neg || (S.r[S.r2[t]].reverse != 0) == r1_neg) return CANON_OUT_OF_SCOPE;
  // the alignment filter of each strand must keep every read (codec_caller.rs:1130-1174): simplified CLIPPED CIGARs, reversed for
  // reverse reads, untruncated, ordered by the clipped length
  for (int strand = 0; strand < 2; strand++) {
    if (nt < 2) break;
    for (uint32_t t = 0; t < nt; t++) {
      const CodecInfo& I = S.r[strand ? S.r2[t] : S.r1[t]];
      ReadInfo& R = S.filt.r[t];
      R.final_len = I.keep; R.keep = 1; R.n_simp = 0;
      SimpOp tmp[MAX_OPS + 2];
      uint32_t m = 0;
      for (uint32_t k = 0; k < I.n_ops; k++) {
        const uint32_t ty = I.ops[k] & 0xF;
        const uint8_t kk = (ty == 4 || ty == 5 || ty == 7 || ty == 8) ? (uint8_t)0 : (uint8_t)ty;
        if (m && tmp[m - 1].k == kk) tmp[m - 1].len += I.ops[k] >> 4;
        else { tmp[m].k = kk; tmp[m].len = I.ops[k] >> 4; m++; }
      }
      if (m > MAX_OPS) return CANON_OUT_OF_SCOPE;
      for (uint32_t k = 0; k < m; k++) R.simp[k] = tmp[I.reverse ? m - 1 - k : k];
      R.n_simp = (uint8_t)m;
      S.filt.list[t] = t;
    }
    const int rej = alignment_filter(S.filt, nt);
    if (rej != 0) return CANON_OUT_OF_SCOPE;
  }
  // overlap geometry of the original (codec_caller.rs:1200-1262): longest alignment of each strand = first maximum of the reference length
  auto longest = [&](const uint32_t* idx, bool by_keep) -> uint32_t {
    uint32_t best = 0;
    int64_t bl = by_keep ? (int64_t)S.r[idx[0]].keep : (int64_t)ref_len_wrapping(S.r[idx[0]].ops, S.r[idx[0]].n_ops);
    for (uint32_t t = 1; t < nt; t++) {
      const int64_t l = by_keep ? (int64_t)S.r[idx[t]].keep : (int64_t)ref_len_wrapping(S.r[idx[t]].ops, S.r[idx[t]].n_ops);
      if (l > bl) { bl = l; best = t; }
    }
    return best;
  };
  const CodecInfo& l1 = S.r[S.r1[longest(S.r1, false)]];
  const CodecInfo& l2 = S.r[S.r2[longest(S.r2, false)]];
  const CodecInfo& lpos = r1_neg ? l2 : l1;
  const CodecInfo& lneg = r1_neg ? l1 : l2;
  const uint64_t pos_ref = (uint64_t)(int64_t)ref_len_wrapping(lpos.ops, lpos.n_ops), neg_ref = (uint64_t)(int64_t)ref_len_wrapping(lneg.ops, lneg.n_ops);
  const uint64_t pos_end = lpos.adj_pos + (pos_ref ? pos_ref - 1 : 0), neg_end = lneg.adj_pos + (neg_ref ? neg_ref - 1 : 0);
  const uint64_t ov_s = lneg.adj_pos > lpos.adj_pos ? lneg.adj_pos : lpos.adj_pos, ov_e = pos_end < neg_end ? pos_end : neg_end;
  const int64_t duplex_len = (int64_t)ov_e - (int64_t)ov_s + 1;
  if (duplex_len < (int64_t)P.min_duplex_length || duplex_len < 1) return CANON_OUT_OF_SCOPE;
  auto at0 = [&](const CodecInfo& r, uint64_t p) -> int64_t { uint64_t q; return read_pos_at(r.ops, r.n_ops, r.adj_pos, p, true, &q) ? (int64_t)q : 0; };
  if ((at0(l1, ov_s) - at0(l2, ov_s)) != (at0(l1, ov_e) - at0(l2, ov_e))) return CANON_OUT_OF_SCOPE;
  uint64_t prp = 0, nrp = 0;
  if (!read_pos_at(lpos.ops, lpos.n_ops, lpos.adj_pos, ov_e, false, &prp) || !read_pos_at(lneg.ops, lneg.n_ops, lneg.adj_pos, ov_e, false, &nrp)) return CANON_OUT_OF_SCOPE;
  if (prp + lneg.keep < nrp) return CANON_OUT_OF_SCOPE;
  const int64_t cons_len = (int64_t)(prp + lneg.keep - nrp);
  // the canonical placement: what the pass will pick as longest (first maximum of the kept length), `cons_len - len_rev` apart
  const uint32_t t1c = longest(S.r1, true), t2c = longest(S.r2, true);
  const uint32_t tA = r1_neg ? t2c : t1c;      // template of the forward read the pass picks
  const uint32_t tB = r1_neg ? t1c : t2c;      // template of the reverse read it picks
  auto fwd_of = [&](uint32_t t) -> CodecInfo& { return S.r[r1_neg ? S.r2[t] : S.r1[t]]; };
  auto rev_of = [&](uint32_t t) -> CodecInfo& { return S.r[r1_neg ? S.r1[t] : S.r2[t]]; };
  const int64_t a_c = fwd_of(tA).keep, b_c = rev_of(tB).keep;
  const int64_t D = cons_len - b_c;              // reverse start - forward start of the picked reads
  const int64_t BASE = 1000000;
  for (uint32_t t = 0; t < nt; t++) {
    const int64_t a = fwd_of(t).keep, b = rev_of(t).keep;
    int64_t d = a > b ? a - b : 0;
    int64_t Pf = BASE + (int64_t)t * 4096, Qr;
    if (tA == tB && t == tA) { if (D < d) return CANON_OUT_OF_SCOPE; d = D; Qr = Pf + d; }
    else if (t == tA) { Pf = BASE; Qr = Pf + d; }
    else if (t == tB) { Qr = BASE + D; Pf = Qr - d; }
    else Qr = Pf + d;
    if (Pf < 1 || Qr < 1) return CANON_OUT_OF_SCOPE;
    fwd_of(t).canon_pos = (uint32_t)Pf; rev_of(t).canon_pos = (uint32_t)Qr;
  }
  {
    // the pass's own view of the geometry: the window must stay non-empty and long enough
    const int64_t Pa = fwd_of(tA).canon_pos, Qb = rev_of(tB).canon_pos;
    const int64_t s2 = Qb > Pa ? Qb : Pa, e2 = (Pa + a_c - 1) < (Qb + b_c - 1) ? (Pa + a_c - 1) : (Qb + b_c - 1);
    const int64_t dl = e2 - s2 + 1;
    if (dl < 1 || dl < (int64_t)P.min_duplex_length) return CANON_OUT_OF_SCOPE;
    if ((e2 - Pa + 1) + b_c - (e2 - Qb + 1) != cons_len) return CANON_OUT_OF_SCOPE;
  }
  // the canonical records: cut, `<keep>M`, placed; the mate fields follow
  for (uint32_t i = 0; i < n; i++) {
    uint8_t* w = out + out_off[i];
    const uint32_t len = rec_len[i];
    const uint8_t* src = blob + rec_off[i];
    for (uint32_t k = 0; k < len; k++) w[k] = src[k];
    const CodecInfo& I = S.r[i];
    wr32(w + 4, I.canon_pos - 1);
    wr32(w + 24, S.r[I.mate].canon_pos - 1);
    out_len[i] = rewrite_record(w, len, I.clip, false);
  }
  return CANON_OK;
}

}  // namespace canon
}  // namespace fgx
