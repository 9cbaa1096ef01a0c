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

}  // namespace canon
}  // namespace fgx
