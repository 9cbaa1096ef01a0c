// simgen.h — synthetic grouped reads, restating the record SHAPE of
// `fgumi simulate grouped-reads` (src/lib/commands/simulate/grouped_reads.rs:666-1011) and the
// position-dependent quality model (src/lib/simulate/quality.rs:60-135, 163-198).
//
// The reference draws from rand 0.10 `StdRng` (ChaCha12) + rand_distr Normal, which cannot be
// reproduced here; seeds and streams are this build's own.  What is restated is observable
// record structure: names `mol%08d_read%04d` (`_readA/_readB` for --duplex), flags
// PAIRED|PROPER_PAIR|FIRST/LAST|REVERSE/MATE_REVERSE, CIGAR `<L>M`, MAPQ 60, tag order
// RX:Z, MI:Z, MC:Z, MQ:c (grouped_reads.rs:951-1011), F1R2 / R1F2 geometry incl. the reference
// simulator's choice to store the reverse read as RC(template slice) (:874-887), random padding
// when insert < L (:853-887), substitution errors from a per-mate stream (:866-871), qualities
// ramp 25→37 over 10 bases / plateau to 100 / decay 0.08 per base / N(0,2) noise / clamp [2,41],
// R2 offset −2 (quality.rs).
//
// Everything is integer arithmetic on a counter-based generator so the SAME molecule is produced
// bit-identically by host code (tests, oracle input, CPU baseline) and by the device kernel that
// fills HBM for the full-size benchmark configurations.
#pragma once
#include <stdint.h>
#include "../../include/fgumi_amd.h"

#if defined(__HIPCC__)
#define SIM_HD __host__ __device__ inline
#else
#define SIM_HD inline
#endif

namespace fgx {
namespace sim {

SIM_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
  return x ^ (x >> 31);
}
SIM_HD uint64_t mol_key(uint64_t seed, uint64_t mol) { return splitmix64(seed ^ splitmix64(mol + 0x1234567ULL)); }
SIM_HD uint64_t rnd(uint64_t key, uint32_t stream, uint64_t ctr) {
  return splitmix64(key ^ ((uint64_t)stream << 44) ^ ctr);
}

enum : uint32_t { ST_MOL = 0, ST_TEMPLATE = 1, ST_PAD = 2, ST_READ0 = 16 };

SIM_HD uint32_t ndigits(uint64_t v) {
  uint32_t d = 1;
  while (v >= 10) { v /= 10; d++; }
  return d;
}

// Irwin–Hall(4) standard-normal surrogate scaled to `sd`: returns round-toward-−inf of z*sd*scale
// where scale converts to the caller's fixed point.  S in [0, 4*65535].
SIM_HD int32_t gauss_fixed(uint64_t r, int32_t sd_times_scale) {
  int64_t s = (int64_t)(r & 0xFFFF) + (int64_t)((r >> 16) & 0xFFFF) + (int64_t)((r >> 32) & 0xFFFF) + (int64_t)(r >> 48);
  int64_t c = s - 131070;                       // mean 0, sd = 65536/sqrt(3)
  int64_t v = c * 113512 / 65536;               // * sqrt(3) (113512/65536 = 1.73206)
  int64_t w = v * sd_times_scale;
  return (int32_t)(w >> 16);                    // arithmetic shift = floor
}

SIM_HD uint32_t family_pairs(const fgx_sim_params& p, uint64_t key) {
  if (p.family_size_max <= p.family_size) return p.family_size;
  // count ∝ size^-1.5 on [lo, hi] by inverse transform of the continuous density,
  // in integer arithmetic: size = lo / (1 - u*(1 - sqrt(lo/hi)))^2, evaluated with 32.32 fixed point.
  uint64_t u = rnd(key, ST_MOL, 3) >> 32;                          // 32-bit uniform
  uint64_t lo = p.family_size, hi = p.family_size_max;
  // isqrt of (lo << 32) / hi  -> sqrt(lo/hi) in 16.16
  uint64_t q = (lo << 32) / hi;                                    // 0.32
  uint64_t r = 0, bit = 1ULL << 30;
  uint64_t n = q;
  while (bit > n) bit >>= 2;
  while (bit) { if (n >= r + bit) { n -= r + bit; r = (r >> 1) + bit; } else r >>= 1; bit >>= 2; }
  uint64_t s16 = r;                                                // sqrt(q) in 0.16
  uint64_t one16 = 1ULL << 16;
  uint64_t t = one16 - ((u >> 16) * (one16 - s16) >> 16);          // 1 - u*(1-s) in 0.16, in (s,1]
  if (t == 0) t = 1;
  uint64_t size = (lo << 32) / (t * t);                            // lo / t^2
  if (size < lo) size = lo;
  if (size > hi) size = hi;
  return (uint32_t)size;
}

struct Molecule {
  uint64_t key;
  uint64_t mol_id;
  uint32_t pairs;
  uint32_t a_pairs;      // duplex: pairs on the /A strand (rest are /B)
  uint32_t insert;
  uint8_t is_top;
  uint8_t umi[8];
  int32_t ref_id, local_pos;
};

SIM_HD Molecule make_molecule(const fgx_sim_params& p, uint64_t fam_index) {
  Molecule m;
  m.mol_id = (uint64_t)p.first_family + fam_index;
  m.key = mol_key(p.seed, m.mol_id);
  uint64_t r0 = rnd(m.key, ST_MOL, 0);
  const char B[4] = {'A', 'C', 'G', 'T'};
  for (int i = 0; i < 8; i++) m.umi[i] = (uint8_t)B[(r0 >> (2 * i)) & 3];
  m.is_top = (uint8_t)((r0 >> 20) & 1);
  int32_t ins = (int32_t)p.insert_mean + gauss_fixed(rnd(m.key, ST_MOL, 1), (int32_t)p.insert_sd);
  if (ins < 50) ins = 50;
  if (ins > 800) ins = 800;
  m.insert = (uint32_t)ins;
  m.pairs = family_pairs(p, m.key);
  m.a_pairs = m.pairs;
  if (p.duplex) {
    // split with at least one read per strand when pairs >= 2 (split_reads_with_minimum)
    if (m.pairs >= 2) {
      uint32_t a = 1 + (uint32_t)((rnd(m.key, ST_MOL, 2) >> 33) % (m.pairs - 1));
      // bias toward an even split like the default strand-bias model: average two draws
      uint32_t a2 = 1 + (uint32_t)((rnd(m.key, ST_MOL, 4) >> 33) % (m.pairs - 1));
      m.a_pairs = (a + a2 + 1) / 2;
    }
  }
  m.ref_id = (int32_t)(m.mol_id / 1000000ULL) % 24;
  m.local_pos = 1000 + (int32_t)(m.mol_id % 1000000ULL) * 1000;
  return m;
}

SIM_HD uint8_t template_base(const Molecule& m, uint32_t i) {
  const char B[4] = {'A', 'C', 'G', 'T'};
  uint64_t r = rnd(m.key, ST_TEMPLATE, i >> 5);
  return (uint8_t)B[(r >> (2 * (i & 31))) & 3];
}
SIM_HD uint8_t comp(uint8_t b) { return b == 'A' ? 'T' : b == 'C' ? 'G' : b == 'G' ? 'C' : b == 'T' ? 'A' : b; }

SIM_HD uint32_t name_len(const fgx_sim_params& p) { return p.duplex ? 21u : 20u; }

SIM_HD uint32_t record_size(const fgx_sim_params& p, uint64_t mol_id) {
  uint32_t L = p.read_length;
  uint32_t mi_len = ndigits(mol_id) + (p.duplex ? 2u : 0u);
  uint32_t mc_len = ndigits(L) + 1;
  return 32 + (name_len(p) + 1) + 4 + (L + 1) / 2 + L + (3 + 8 + 1) + (3 + mi_len + 1) + (3 + mc_len + 1) + 4;
}

SIM_HD uint16_t reg2bin(int32_t beg, int32_t end) {
  --end;
  if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

SIM_HD uint8_t quality_at(uint64_t key, uint32_t stream, uint32_t pos, bool is_r2) {
  int32_t base100;
  if (pos < 10) base100 = 2500 + 120 * (int32_t)pos;
  else if (pos < 100) base100 = 3700;
  else { base100 = 3700 - 8 * (int32_t)(pos - 100); if (base100 < 200) base100 = 200; }
  int32_t noisy = base100 + gauss_fixed(rnd(key, stream, pos), 200);
  int32_t q = (noisy + 50) >= 0 ? (noisy + 50) / 100 : -((-(noisy + 50) + 99) / 100);
  if (q < 2) q = 2;
  if (q > 41) q = 41;
  if (is_r2) { q -= 2; if (q < 2) q = 2; if (q > 41) q = 41; }
  return (uint8_t)q;
}

SIM_HD void put16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
SIM_HD void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
SIM_HD uint32_t put_dec(uint8_t* p, uint64_t v, uint32_t width) {  // zero-padded when width>0
  uint32_t d = ndigits(v);
  uint32_t n = d < width ? width : d;
  for (uint32_t i = 0; i < n; i++) { p[n - 1 - i] = (uint8_t)('0' + (v % 10)); v /= 10; }
  return n;
}

// Sequence of one mate in STORED orientation. fwd: template[0..L) ; rev: RC(template[rs..rs+re)) ; both padded
// with random bases to L (grouped_reads.rs:853-887).  p.codec = 1 stores the reverse mate in reference orientation
// (template[rs..rs+re)), the way an aligner writes SEQ, so the two strands of a CODEC pair agree where they overlap.
SIM_HD uint8_t mate_base(const fgx_sim_params& p, const Molecule& m, bool fwd_mate, uint32_t read_stream, uint32_t i) {
  const char B[4] = {'A', 'C', 'G', 'T'};
  uint32_t L = p.read_length, ins = m.insert;
  uint8_t b;
  if (fwd_mate) {
    uint32_t fwd_end = L < ins ? L : ins;
    if (i < fwd_end) b = template_base(m, i);
    else b = (uint8_t)B[(rnd(m.key, ST_PAD, ((uint64_t)read_stream << 16) | i) >> 7) & 3];
  } else {
    uint32_t rs = ins > L ? ins - L : 0;
    uint32_t avail = ins - rs;
    uint32_t re = L < avail ? L : avail;
    if (i < re) b = p.codec ? template_base(m, rs + i) : comp(template_base(m, rs + (re - 1 - i)));
    else b = (uint8_t)B[(rnd(m.key, ST_PAD, ((uint64_t)read_stream << 16) | i) >> 9) & 3];
  }
  if (p.error_rate_ppm) {
    uint64_t r = rnd(m.key, read_stream + 1, i);
    uint64_t thr = ((uint64_t)p.error_rate_ppm << 32) / 1000000ULL;
    if ((r >> 32) < thr) {
      uint32_t k = (uint32_t)(r & 0xFFFF) % 3;
      const char* alts = b == 'A' ? "CGT" : b == 'C' ? "AGT" : b == 'G' ? "ACT" : "ACG";
      b = (uint8_t)alts[k];
    }
  }
  return b;
}
SIM_HD uint8_t base_code(uint8_t b) { return b == 'A' ? 1 : b == 'C' ? 2 : b == 'G' ? 4 : b == 'T' ? 8 : 15; }

// Writes one record (with block_size prefix) at dst; returns bytes written.
SIM_HD uint32_t write_record(const fgx_sim_params& p, const Molecule& m, uint32_t pair_idx, bool strand_a, bool pair_is_top,
                             bool is_first, uint8_t* dst) {
  uint32_t L = p.read_length;
  uint32_t rsz = record_size(p, m.mol_id);
  // pair-in-strand index for names (readA0000.. / readB0000..)
  uint32_t idx_in_strand = (p.duplex && !strand_a) ? pair_idx - m.a_pairs : pair_idx;
  bool r1_is_reverse = !pair_is_top;
  bool is_reverse = is_first ? r1_is_reverse : !r1_is_reverse;
  bool fwd_mate = !is_reverse;
  int32_t rev_pos = m.local_pos + (int32_t)(m.insert > L ? m.insert - L : 0);
  int32_t pos = fwd_mate ? m.local_pos : rev_pos;
  int32_t mate_pos = fwd_mate ? rev_pos : m.local_pos;
  int32_t r1_tlen = pair_is_top ? (int32_t)m.insert : -(int32_t)m.insert;
  int32_t tlen = is_first ? r1_tlen : -r1_tlen;
  uint16_t flag = 0x1 | 0x2 | (is_first ? 0x40 : 0x80) | (is_reverse ? 0x10 : 0x20);
  uint32_t read_stream = ST_READ0 + (pair_idx * 2 + (is_first ? 0u : 1u)) * 4;

  uint8_t* q = dst;
  put32(q, rsz); q += 4;
  put32(q, (uint32_t)m.ref_id); put32(q + 4, (uint32_t)pos);
  q[8] = (uint8_t)(name_len(p) + 1); q[9] = 60;
  put16(q + 10, reg2bin(pos, pos + (int32_t)L));
  put16(q + 12, 1); put16(q + 14, flag); put32(q + 16, L);
  put32(q + 20, (uint32_t)m.ref_id); put32(q + 24, (uint32_t)mate_pos); put32(q + 28, (uint32_t)tlen);
  q += 32;
  q[0] = 'm'; q[1] = 'o'; q[2] = 'l'; q += 3;
  q += put_dec(q, m.mol_id % 100000000ULL, 8);
  q[0] = '_'; q[1] = 'r'; q[2] = 'e'; q[3] = 'a'; q[4] = 'd'; q += 5;
  if (p.duplex) *q++ = strand_a ? 'A' : 'B';
  q += put_dec(q, idx_in_strand % 10000, 4);
  *q++ = 0;
  put32(q, L << 4); q += 4;
  for (uint32_t i = 0; i < L; i += 2) {
    uint8_t hi = base_code(mate_base(p, m, fwd_mate, read_stream, i));
    uint8_t lo = (i + 1 < L) ? base_code(mate_base(p, m, fwd_mate, read_stream, i + 1)) : 0;
    *q++ = (uint8_t)((hi << 4) | lo);
  }
  // r1 quals are drawn first, r2 second with the −2 offset (grouped_reads.rs:893-896)
  for (uint32_t i = 0; i < L; i++) *q++ = quality_at(m.key, read_stream + 2, i, !is_first);
  q[0] = 'R'; q[1] = 'X'; q[2] = 'Z'; q += 3;
  for (int i = 0; i < 8; i++) *q++ = m.umi[i];
  *q++ = 0;
  q[0] = 'M'; q[1] = 'I'; q[2] = 'Z'; q += 3;
  q += put_dec(q, m.mol_id, 0);
  if (p.duplex) { *q++ = '/'; *q++ = strand_a ? 'A' : 'B'; }
  *q++ = 0;
  q[0] = 'M'; q[1] = 'C'; q[2] = 'Z'; q += 3;
  q += put_dec(q, L, 0);
  *q++ = 'M'; *q++ = 0;
  q[0] = 'M'; q[1] = 'Q'; q[2] = 'c'; q[3] = 60; q += 4;
  return (uint32_t)(q - dst);
}

// Writes all records of family `fam_index` (template-coordinate order: pairs in name order, the
// lower-coordinate (forward) mate first), plus its offset/length/group entries.
SIM_HD void write_family(const fgx_sim_params& p, uint64_t fam_index, uint64_t byte_off, uint32_t rec_first, uint8_t* blob,
                         uint64_t* rec_off, uint32_t* rec_len, uint32_t* grp_first) {
  Molecule m = make_molecule(p, fam_index);
  uint32_t rsz = record_size(p, m.mol_id) + 4;
  grp_first[fam_index] = rec_first;
  uint32_t r = rec_first;
  uint64_t off = byte_off;
  for (uint32_t j = 0; j < m.pairs; j++) {
    bool strand_a = j < m.a_pairs;
    bool pair_is_top = (p.duplex && !strand_a) ? !m.is_top : (bool)m.is_top;
    // forward-strand mate has the lower coordinate: R1 when the pair is top-strand, else R2
    bool first_is_fwd = pair_is_top;
    for (int k = 0; k < 2; k++) {
      bool is_first = (k == 0) ? first_is_fwd : !first_is_fwd;
      write_record(p, m, j, strand_a, pair_is_top, is_first, blob + off);
      rec_off[r] = off + 4;
      rec_len[r] = rsz - 4;
      off += rsz;
      r++;
    }
  }
}

}  // namespace sim
}  // namespace fgx
