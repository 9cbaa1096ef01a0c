// bamrec.h — raw BAM record access and the integer geometry the consensus path needs, usable
// from host C++ and from HIP device code (no heap, fixed-capacity scratch).
//
// Interfaces replaced (reference, crates/fgumi-raw-bam/src/):
//   fields.rs:78-148,309-330,508-522 (RawRecordView accessors, tag_value_size, seq/qual offsets)
//   tags.rs:13-48 (find_tag_position / find_string_tag)
//   sequence.rs:9-52 (4-bit base codec)
//   overlap.rs:21-161 (FR-pair tests), 181-357 (bases extending past mate), 311-376 (MC parser)
//   cigar.rs:160-232 (reference/query length)
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define BAM_HD __host__ __device__ inline
#else
#define BAM_HD inline
#endif

namespace fgx {
namespace bam {

enum : uint16_t {
  F_PAIRED = 0x1, F_PROPER = 0x2, F_UNMAPPED = 0x4, F_MATE_UNMAPPED = 0x8, F_REVERSE = 0x10, F_MATE_REVERSE = 0x20,
  F_FIRST = 0x40, F_LAST = 0x80, F_SECONDARY = 0x100, F_QCFAIL = 0x200, F_DUP = 0x400, F_SUPPLEMENTARY = 0x800
};

BAM_HD uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | ((uint16_t)p[1] << 8)); }
BAM_HD uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct Rec {
  const uint8_t* b;
  uint32_t len;
  BAM_HD int32_t ref_id() const { return (int32_t)rd32(b); }
  BAM_HD int32_t pos() const { return (int32_t)rd32(b + 4); }
  BAM_HD uint32_t l_read_name() const { return b[8]; }
  BAM_HD uint32_t n_cigar() const { return rd16(b + 12); }
  BAM_HD uint16_t flags() const { return rd16(b + 14); }
  BAM_HD uint32_t l_seq() const { return rd32(b + 16); }
  BAM_HD int32_t mate_ref_id() const { return (int32_t)rd32(b + 20); }
  BAM_HD int32_t mate_pos() const { return (int32_t)rd32(b + 24); }
  BAM_HD int32_t tlen() const { return (int32_t)rd32(b + 28); }
  BAM_HD const uint8_t* name() const { return b + 32; }
  BAM_HD uint32_t name_len() const { uint32_t l = l_read_name(); return l ? l - 1 : 0; }
  BAM_HD uint32_t cigar_off() const { return 32 + l_read_name(); }
  BAM_HD uint32_t seq_off() const { return 32 + l_read_name() + 4 * n_cigar(); }
  BAM_HD uint32_t qual_off() const { return seq_off() + (l_seq() + 1) / 2; }
  BAM_HD uint32_t aux_off() const { return qual_off() + l_seq(); }
  BAM_HD uint32_t cigar_op(uint32_t i) const { return rd32(b + cigar_off() + 4 * i); }
  BAM_HD uint8_t base_code(uint32_t i) const { uint8_t v = b[seq_off() + (i >> 1)]; return (i & 1) ? (v & 0xF) : (v >> 4); }
};

// 4-bit code → ASCII ("=ACMGRSVTWYHKDBN") and the IUPAC complement in code space.
BAM_HD uint8_t code_to_ascii(uint8_t c) {
  const char* T = "=ACMGRSVTWYHKDBN";
  return (uint8_t)T[c & 15];
}
// complement of code: '='→'=', A↔T, C↔G, M↔K, R↔Y, S, W, V↔B, H↔D, N
BAM_HD uint8_t code_complement(uint8_t c) {
  // index:      =  A  C  M   G  R  S  V   T  W  Y   H   K  D   B  N
  const uint8_t T[16] = {0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15};
  return T[c & 15];
}
// BASE_TO_INDEX for codes: A(1)->0, C(2)->1, G(4)->2, T(8)->3, else 255
BAM_HD int code_to_lane(uint8_t c) { return c == 1 ? 0 : c == 2 ? 1 : c == 4 ? 2 : c == 8 ? 3 : 255; }
BAM_HD int ascii_to_lane(uint8_t b) {
  switch (b) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 255; }
}
BAM_HD uint8_t ascii_to_code(uint8_t b) {
  switch (b) {
    case '=': return 0; case 'A': case 'a': return 1; case 'C': case 'c': return 2; case 'M': case 'm': return 3;
    case 'G': case 'g': return 4; case 'R': case 'r': return 5; case 'S': case 's': return 6; case 'V': case 'v': return 7;
    case 'T': case 't': return 8; case 'W': case 'w': return 9; case 'Y': case 'y': return 10; case 'H': case 'h': return 11;
    case 'K': case 'k': return 12; case 'D': case 'd': return 13; case 'B': case 'b': return 14; default: return 15;
  }
}

// ---- aux tags -------------------------------------------------------------------------------
BAM_HD int tag_fixed_size(uint8_t t) {
  switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; }
}
BAM_HD int64_t find_nul(const uint8_t* p, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) if (p[i] == 0) return i;
  return -1;
}
// Finds a Z-typed tag; returns value offset relative to `aux` and its length (excluding NUL); -1 if absent / not Z.
BAM_HD int64_t find_z_tag(const uint8_t* aux, uint32_t n, uint8_t t0, uint8_t t1, uint32_t* vlen) {
  uint32_t p = 0;
  while (p + 3 <= n) {
    uint8_t vt = aux[p + 2];
    if (aux[p] == t0 && aux[p + 1] == t1) {
      if (vt != 'Z') return -1;
      int64_t e = find_nul(aux + p + 3, n - (p + 3));
      if (e < 0) return -1;
      *vlen = (uint32_t)e;
      return (int64_t)p + 3;
    }
    int fixed = tag_fixed_size(vt);
    uint32_t size;
    if (fixed > 0) size = (uint32_t)fixed;
    else if (vt == 'Z' || vt == 'H') {
      int64_t e = find_nul(aux + p + 3, n - (p + 3));
      if (e < 0) return -1;
      size = (uint32_t)e + 1;
    } else if (vt == 'B') {
      if (n - (p + 3) < 5) return -1;
      int es = tag_fixed_size(aux[p + 3]);
      if (es == 0) return -1;
      uint64_t cnt = rd32(aux + p + 4);
      uint64_t s = 5 + cnt * (uint64_t)es;
      if (s > 0xFFFFFFFFull) return -1;
      size = (uint32_t)s;
    } else return -1;
    uint64_t np = (uint64_t)p + 3 + size;
    if (np > n) { p = n; break; }
    p = (uint32_t)np;
  }
  return -1;
}

// ---- CIGAR arithmetic (ops are BAM-encoded u32: len<<4 | op) ---------------------------------
BAM_HD bool op_consumes_ref(uint32_t t) { return t == 0 || t == 2 || t == 3 || t == 7 || t == 8; }
BAM_HD bool op_consumes_query(uint32_t t) { return t == 0 || t == 1 || t == 4 || t == 7 || t == 8; }
BAM_HD int32_t sat_add(int32_t a, int32_t b) { int64_t s = (int64_t)a + b; return s > 2147483647LL ? 2147483647 : s < -2147483648LL ? (int32_t)(-2147483647 - 1) : (int32_t)s; }
BAM_HD int32_t sat_sub(int32_t a, int32_t b) { int64_t s = (int64_t)a - b; return s > 2147483647LL ? 2147483647 : s < -2147483648LL ? (int32_t)(-2147483647 - 1) : (int32_t)s; }
BAM_HD int32_t oplen_i32(uint32_t op) { uint32_t l = op >> 4; return l > 2147483647u ? 2147483647 : (int32_t)l; }
BAM_HD uint64_t sub0(uint64_t a, uint64_t b) { return a > b ? a - b : 0; }

BAM_HD int32_t sat_ref_len(const uint32_t* ops, uint32_t n) {
  int32_t r = 0;
  for (uint32_t i = 0; i < n; i++) if (op_consumes_ref(ops[i] & 0xF)) r = sat_add(r, oplen_i32(ops[i]));
  return r;
}
// reference_length_from_raw_bam (unchecked adds wrap → the checked variant returns None → 0)
BAM_HD int32_t ref_len_checked0(const uint32_t* ops, uint32_t n) {
  int64_t r = 0;
  for (uint32_t i = 0; i < n; i++) if (op_consumes_ref(ops[i] & 0xF)) { r += (int32_t)(ops[i] >> 4); if (r > 2147483647LL) return 0; }
  return (int32_t)r;
}
BAM_HD uint64_t query_len(const uint32_t* ops, uint32_t n) {
  uint64_t l = 0;
  for (uint32_t i = 0; i < n; i++) if (op_consumes_query(ops[i] & 0xF)) l += ops[i] >> 4;
  return l;
}
BAM_HD uint64_t lead_soft(const uint32_t* ops, uint32_t n) {
  uint64_t t = 0;
  for (uint32_t i = 0; i < n; i++) { uint32_t ty = ops[i] & 0xF; if (ty == 4) t += ops[i] >> 4; else if (ty != 5) break; }
  return t;
}
BAM_HD uint64_t trail_soft(const uint32_t* ops, uint32_t n) {
  uint64_t t = 0;
  for (uint32_t i = n; i-- > 0;) { uint32_t ty = ops[i] & 0xF; if (ty == 4) t += ops[i] >> 4; else if (ty != 5) break; }
  return t;
}
BAM_HD uint64_t qbases_up_to(const uint32_t* ops, uint32_t n, int32_t start1, int32_t target_pos, bool inclusive) {
  int64_t target = target_pos, incl = inclusive ? 1 : 0;
  int32_t ref_pos = start1;
  uint64_t q = 0;
  for (uint32_t i = 0; i < n; i++) {
    if ((int64_t)ref_pos > target) break;
    uint32_t ty = ops[i] & 0xF;
    uint64_t len = ops[i] >> 4;
    if (ty == 0 || ty == 7 || ty == 8) {
      int64_t span = target - (int64_t)ref_pos + incl;
      uint64_t s = span > 0 ? (uint64_t)span : 0;
      uint64_t take = len < s ? len : s;
      q += take;
      ref_pos = sat_add(ref_pos, oplen_i32(ops[i]));
      if (take < len) break;
    } else if (ty == 1 || ty == 4) q += len;
    else if (ty == 2 || ty == 3) ref_pos = sat_add(ref_pos, oplen_i32(ops[i]));
  }
  return q;
}
// bases_extending_past_mate_ops (overlap.rs:207-268)
BAM_HD uint64_t past_mate_ops(bool is_reverse, int32_t this_pos1, const uint32_t* t_ops, uint32_t tn, int32_t mate_pos1,
                              const uint32_t* m_ops, uint32_t mn) {
  int32_t read_end = sat_add(sat_sub(this_pos1, 1), sat_ref_len(t_ops, tn));
  int32_t mate_end = sat_add(sat_sub(mate_pos1, 1), sat_ref_len(m_ops, mn));
  uint64_t ls = lead_soft(m_ops, mn), ts = trail_soft(m_ops, mn);
  int32_t lead = ls > 2147483647ull ? 2147483647 : (int32_t)ls;
  int32_t trail = ts > 2147483647ull ? 2147483647 : (int32_t)ts;
  int32_t m_ustart = sat_sub(mate_pos1, lead);
  int32_t m_uend = sat_add(sat_add(sat_sub(mate_pos1, 1), sat_ref_len(m_ops, mn)), trail);
  if (is_reverse) {
    if (this_pos1 > mate_end) {
      uint64_t gap = (uint64_t)(uint32_t)sat_sub(this_pos1, m_ustart);
      return sub0(lead_soft(t_ops, tn), gap);
    }
    if (read_end < mate_pos1) return 0;
    int32_t first_shared = this_pos1 > mate_pos1 ? this_pos1 : mate_pos1;
    return sub0(qbases_up_to(t_ops, tn, this_pos1, first_shared, false), qbases_up_to(m_ops, mn, mate_pos1, first_shared, false));
  }
  if (read_end < mate_pos1) {
    uint64_t gap = (uint64_t)(uint32_t)sat_sub(m_uend, read_end);
    return sub0(trail_soft(t_ops, tn), gap);
  }
  if (mate_end < this_pos1) return 0;
  int32_t last_shared = read_end < mate_end ? read_end : mate_end;
  uint64_t rp = sub0(query_len(t_ops, tn), qbases_up_to(t_ops, tn, this_pos1, last_shared, true));
  uint64_t mp = sub0(query_len(m_ops, mn), qbases_up_to(m_ops, mn, mate_pos1, last_shared, true));
  return sub0(rp, mp);
}

// parse_mc_cigar_ops: returns op count (>0) on success, 0 if malformed, -1 if more than `cap` ops.
BAM_HD int parse_mc(const uint8_t* s, uint32_t n, uint32_t* ops, uint32_t cap) {
  const uint32_t MAXLEN = (1u << 28) - 1;
  uint32_t cnt = 0;
  uint64_t num = 0;
  bool have = false, saw_ref = false;
  // placement rules (overlap.rs:340-366): S valid iff every token before it is H, or every token
  // after it is H; H valid only as the first or the last token.
  bool only_h_so_far = true;        // every token so far is H
  bool need_only_h_after = false;   // a non-leading S was seen: only H may follow
  bool pending_last_h = false;      // a non-first H was seen: nothing may follow
  for (uint32_t i = 0; i < n; i++) {
    uint8_t c = s[i];
    if (c >= '0' && c <= '9') {
      num = num * 10 + (uint64_t)(c - '0');
      if (num > 0xFFFFFFFFull) num = 0xFFFFFFFFull;
      if (num > MAXLEN) return 0;
      have = true;
      continue;
    }
    int code;
    switch (c) { case 'M': code = 0; break; case 'I': code = 1; break; case 'D': code = 2; break; case 'N': code = 3; break;
                 case 'S': code = 4; break; case 'H': code = 5; break; case 'P': code = 6; break; case '=': code = 7; break;
                 case 'X': code = 8; break; default: code = -1; }
    if (!have || num == 0 || code < 0) return 0;
    if (pending_last_h) return 0;
    if (c != 'H' && need_only_h_after) return 0;
    if (c == 'H') { if (cnt != 0) pending_last_h = true; }
    else if (c == 'S') { if (!only_h_so_far) need_only_h_after = true; }
    else if (c == 'M' || c == 'D' || c == 'N' || c == '=' || c == 'X') saw_ref = true;
    if (c != 'H') only_h_so_far = false;
    if (cnt >= cap) return -1;
    ops[cnt++] = ((uint32_t)num << 4) | (uint32_t)code;
    num = 0;
    have = false;
  }
  if (have || cnt == 0) return 0;
  if (!saw_ref) return 0;
  return (int)cnt;
}

// is_fr_pair_raw (overlap.rs:21-69); `ops` = this record's CIGAR ops
BAM_HD bool is_fr_pair(const Rec& v, const uint32_t* ops, uint32_t n_ops) {
  uint16_t f = v.flags();
  if (!(f & F_PAIRED) || (f & F_UNMAPPED) || (f & F_MATE_UNMAPPED)) return false;
  if (v.ref_id() != v.mate_ref_id()) return false;
  bool rev = (f & F_REVERSE) != 0, mrev = (f & F_MATE_REVERSE) != 0;
  if (rev == mrev) return false;
  uint32_t astart = (uint32_t)v.pos() + 1u, mstart = (uint32_t)v.mate_pos() + 1u;
  int32_t p5, n5;
  if (rev) {
    int32_t rl = ref_len_checked0(ops, n_ops);
    int32_t ext = rl - 1 > 0 ? rl - 1 : 0;
    p5 = (int32_t)mstart;
    n5 = (int32_t)(astart + (uint32_t)ext);
  } else {
    p5 = (int32_t)astart;
    n5 = (int32_t)(astart + (uint32_t)v.tlen());
  }
  return p5 < n5;
}

// num_bases_extending_past_mate_raw (overlap.rs:181-207) given the MC value (mc, mc_len; mc == nullptr
// when the tag is absent).  `scratch` holds the parsed mate ops; *overflow is set when it is too small.
BAM_HD uint64_t mate_clip(const Rec& v, const uint32_t* ops, uint32_t n_ops, const uint8_t* mc, uint32_t mc_len, uint32_t* scratch,
                          uint32_t cap, bool* overflow) {
  if (!mc) return 0;
  int n = parse_mc(mc, mc_len, scratch, cap);
  if (n < 0) { *overflow = true; return 0; }
  if (n == 0) return 0;
  int32_t mate_ref_len = sat_ref_len(scratch, (uint32_t)n);
  uint16_t f = v.flags();
  if (!(f & F_PAIRED) || (f & F_UNMAPPED) || (f & F_MATE_UNMAPPED)) return 0;
  if (v.ref_id() != v.mate_ref_id()) return 0;
  bool rev = (f & F_REVERSE) != 0, mrev = (f & F_MATE_REVERSE) != 0;
  if (rev == mrev) return 0;
  int32_t this_pos1 = (int32_t)((uint32_t)v.pos() + 1u), mate_pos1 = (int32_t)((uint32_t)v.mate_pos() + 1u);
  if (rev) { if (!is_fr_pair(v, ops, n_ops)) return 0; }
  else {
    int32_t ext = mate_ref_len - 1 > 0 ? mate_ref_len - 1 : 0;
    int32_t mate_end = sat_add(mate_pos1, ext);
    if (!(this_pos1 < mate_end)) return 0;
  }
  return past_mate_ops(rev, this_pos1, ops, n_ops, mate_pos1, scratch, (uint32_t)n);
}

}  // namespace bam
}  // namespace fgx
