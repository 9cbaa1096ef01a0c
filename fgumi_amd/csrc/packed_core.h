// packed_core.h — the arithmetic of k_split_cols's PACKED column pass (round 5) as host + device functions: one source for the kernel
// (simplex_split.inc, run_cols_packed) and for tests/devemu, where the same functions run on the host against a column-by-column
// restatement and the oracle's ConsensusBaseBuilder (tests/test_packed_core.py).
//
// A lane of the pass owns EIGHT neighbouring tile positions of one end: position p of its group is quality byte p of a 64-bit word of a
// quality row and code nibble (p ^ 1) of a 32-bit word of a sequence row (BAM packs the even position into the high nibble).  Per row the
// codes of observations below --min-input-base-quality are cleared; the lane keeps the OR of the codes and two words of byte counters
// (8 x the number of codes that are not 0: low nibbles = odd positions, high nibbles = even positions).  A column shows ONE base when its
// OR is 1, 2, 4 or 8; its counter is then the number of its observations.  From unanimous_cap_depth (gate_core.h) observations on such a
// column is (base, cap) whatever the qualities are; a column of no observation is ('N', 2 | 0); every other column — a second base, a
// non-ACGT code, one base seen fewer times — is FLAGGED: its observations travel to k_call_full (or, a single observation: to the table
// of fill_t1).  A code 0 ('=') of good quality is no observation for the reference (ConsensusBaseBuilder::add ignores what is not ACGT,
// base_builder.rs:836-868; k_call_full skips it the same way) and adds nothing here.
#pragma once
#include <cstdint>
#include "consensus_math.h"

namespace fgx {
namespace pk {

constexpr uint32_t H = 0x80808080u;

// v_perm_b32: byte i of the result = byte sel[i] of the eight bytes {a (4 - 7), b (0 - 3)}; selector 0x0C = the constant 0
FGX_HD uint32_t perm(uint32_t a, uint32_t b, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(a, b, sel);
#else
  const unsigned long long src = ((unsigned long long)a << 32) | b;
  uint32_t r = 0;
  for (int i = 0; i < 4; i++) {
    const uint32_t s = (sel >> (8 * i)) & 0xFFu;
    const uint32_t byte = s < 8u ? (uint32_t)(src >> (8 * s)) & 0xFFu : 0u;      // (only the selectors this file uses)
    r |= byte << (8 * i);
  }
  return r;
#endif
}
FGX_HD uint32_t bitrev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bitreverse32(x);
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); return __builtin_bswap32(x);
#endif
}

// positions of the group at or past the end's read length are no columns: masks of the quality bytes that count (bit 7 of each)
FGX_HD void count_masks(uint32_t lenE, uint32_t k, uint32_t* nfl, uint32_t* nfh) {
  *nfl = H; *nfh = H;
  const uint32_t nv = lenE > 8u * k ? lenE - 8u * k : 0u;
  if (nv < 8u) { const unsigned long long f = ~(0x8080808080808080ull << (8u * nv)); *nfl &= (uint32_t)f; *nfh &= (uint32_t)(f >> 32); }
}
// any quality of the group's positions below the floor?  (x - floor) & ~x & 0x80 per byte: not zero <=> some byte is (exact as a TEST for
// floors up to 128: a borrow can only raise a false flag above a true one)
FGX_HD bool any_below(uint32_t qx, uint32_t qy, uint32_t mb4, uint32_t nfl, uint32_t nfh) {
  return ((((qx - mb4) & ~qx & nfl) | ((qy - mb4) & ~qy & nfh))) != 0u;
}
// the code nibbles to keep: 0xF where the position's quality is at or above the floor.  Bit 7 of each byte of gl / gh = "at or above"
// (no borrow between bytes: q | 0x80 >= 128 >= floor; a quality >= 128 is above every floor this pass takes)
FGX_HD uint32_t keep_mask(uint32_t qx, uint32_t qy, uint32_t mb4) {
  const uint32_t gl = ((qx | H) - mb4) | qx, gh = ((qy | H) - mb4) | qy;
  uint32_t keep = 0;
  keep |= (uint32_t)((int32_t)(gl << 24) >> 31) & 0x000000F0u; keep |= (uint32_t)((int32_t)(gl << 16) >> 31) & 0x0000000Fu;
  keep |= (uint32_t)((int32_t)(gl << 8) >> 31) & 0x0000F000u;  keep |= (uint32_t)((int32_t)gl >> 31) & 0x00000F00u;
  keep |= (uint32_t)((int32_t)(gh << 24) >> 31) & 0x00F00000u; keep |= (uint32_t)((int32_t)(gh << 16) >> 31) & 0x000F0000u;
  keep |= (uint32_t)((int32_t)(gh << 8) >> 31) & 0xF0000000u;  keep |= (uint32_t)((int32_t)gh >> 31) & 0x0F000000u;
  return keep;
}

struct Acc { uint32_t f_or, A8, B8; };     // OR of the codes (eight nibbles); 8 x the codes that are not 0 per byte: low nibbles, high nibbles
FGX_HD void acc_reset(Acc& a) { a.f_or = 0; a.A8 = 0; a.B8 = 0; }
// one row of the lane's group: q = its eight qualities (qx: positions 0 - 3), b = its eight codes
FGX_HD void acc_row(Acc& a, uint32_t qx, uint32_t qy, uint32_t b, uint32_t mb4, uint32_t nfl, uint32_t nfh) {
  if (any_below(qx, qy, mb4, nfl, nfh)) b &= keep_mask(qx, qy, mb4);
  a.f_or |= b;
  const uint32_t nz = (((b & 0x77777777u) + 0x77777777u) | b) & 0x88888888u;   // bit 3 of each nibble: the nibble is not 0
  a.A8 += nz & 0x08080808u; a.B8 += (nz >> 4) & 0x08080808u;
}

// What the lane's eight slots are: slot s = column c_lo + s of its end (forward end: position s of the group; reverse end: position
// 7 - s, complemented).  code2 / qual2: a byte per slot; dep4: 16 bits per slot; flag2 / valid2: bit 7 of the slot's byte — flagged
// (k_call_full's) / a column of the end.  Slots [lo_s, hi_s) are the columns.
struct Out { uint32_t code2[2], qual2[2], dep4[4], flag2[2], valid2[2]; int32_t c_lo, lo_s, hi_s; };
FGX_HD void finalize(const Acc& a, bool inl, bool lrev, uint32_t lenE, uint32_t cntE, uint32_t k, uint32_t nsafe, uint32_t cap, uint32_t min_cons_bq, uint32_t min_reads, Out& o) {
  uint32_t f_or = a.f_or, Ce = a.A8 >> 3, Co = a.B8 >> 3;                      // observations per column, a byte each: odd / even positions
  // reverse end: bit reversal of the OR word complements the codes AND puts them in column order; the counters are byte-swapped
  if (lrev) { f_or = bitrev32(f_or); const uint32_t t = __builtin_bswap32(Ce); Ce = __builtin_bswap32(Co); Co = t; }
  o.c_lo = lrev ? (int32_t)lenE - 8 - 8 * (int32_t)k : 8 * (int32_t)k;         // (below zero for the reverse end's last group)
  const int32_t hi_r = (int32_t)cntE - o.c_lo;
  o.lo_s = o.c_lo < 0 ? -o.c_lo : 0; o.hi_s = !inl || hi_r < 0 ? 0 : hi_r > 8 ? 8 : hi_r;
  unsigned long long V = 0;
  if (o.hi_s > o.lo_s) {
    const unsigned long long hiM = o.hi_s >= 8 ? ~0ull : (1ull << (8 * o.hi_s)) - 1ull, loM = (1ull << (8 * (o.lo_s > 7 ? 7 : o.lo_s))) - 1ull;
    V = hiM & ~loM & 0x8080808080808080ull;
  }
  const uint32_t N4 = nsafe * 0x01010101u;
  const bool cap_lt = cap < min_cons_bq;                                       // (the cap itself is below --min-consensus-base-quality: every answered column is masked)
  const uint32_t QA4 = (cap_lt ? (uint32_t)FGX_MIN_PHRED : cap) * 0x01010101u, QZ4 = (min_reads > 0u ? 0u : (uint32_t)FGX_MIN_PHRED) * 0x01010101u;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    // the four slots of this half, a byte each: OR of the codes (O), observations (C)
    const uint32_t Pb = perm(f_or, f_or, h ? 0x03030202u : 0x01010000u);       // bytes {b, b, b', b'}: slots 2i (high nibble), 2i + 1 (low nibble)
    const uint32_t O = ((Pb >> 4) & 0x000F000Fu) | (Pb & 0x0F000F00u);
    const uint32_t C = perm(Ce, Co, h ? 0x07030602u : 0x05010400u);
    const uint32_t Vh = h ? (uint32_t)(V >> 32) : (uint32_t)V;
    // one base: the byte is 1, 2, 4 or 8  <=>  not 0 and x & (x - 1) == 0  ((x | 0x80) - 1: no borrow between bytes)
    const uint32_t Z = O & ((O | H) - 0x01010101u) & 0x0F0F0F0Fu;
    const uint32_t nzO = (O + 0x7F7F7F7Fu) & H, nzZ = (Z + 0x7F7F7F7Fu) & H;
    const uint32_t ge = ((C | H) - N4) & H;                                    // at least nsafe observations (both below 128: no borrow)
    const uint32_t capF = Vh & nzO & ~nzZ & ge;                                // (base, cap)
    o.flag2[h] = Vh & nzO & ~capF;                                             // some observation and no answer here
    o.valid2[h] = Vh;
    const uint32_t capM = (capF << 1) - (capF >> 7);                           // 0xFF in the bytes of the flags
    o.code2[h] = cap_lt ? 0x0F0F0F0Fu : ((O & capM) | (0x0F0F0F0Fu & ~capM));
    o.qual2[h] = (QA4 & capM) | (QZ4 & ~capM);
    const uint32_t D = C & capM;
    o.dep4[2 * h] = perm(0u, D, 0x0C010C00u); o.dep4[2 * h + 1] = perm(0u, D, 0x0C030C02u);
  }
}

// One observation of base b and quality q: ConsensusBaseBuilder::add from zero leaves s[b] = correct[q], s[other] = error_per_alt[q]
// exactly (Kahan's first step adds to 0), and the call (try_unanimous_fast_path, else call_full: base_builder.rs:883-1081) is a function
// of q alone — evaluated here by the very functions k_call_full runs (column_call, consensus_math.h), once per caller.  t1[q] = the
// consensus quality, 0xFF where the call does not come back with the observed base (the column then travels like any other).
inline void fill_t1(uint8_t* t1 /* [96] */, const ConsensusTables& t) {
  for (uint32_t q = 0; q < 96; q++) {
    t1[q] = 0xFF;
    if (q > 93) continue;
    ColumnAcc acc;
    acc.reset();
    acc.add(0, t.correct[q], t.error_per_alt[q]);
    if (!m_isfinite(acc.s[0]) || !m_isfinite(acc.s[1])) continue;
    int bi = -1;
    uint8_t ql = 0;
    column_call(t, acc.s, acc.obs, &bi, &ql);
    if (bi == 0) t1[q] = ql;
  }
}

// Two observations of ONE base, qualities q1 then q2 in file order (round 6): the same evaluation — two steps of ConsensusBaseBuilder::add
// (the Kahan sums in the reference's order), then the call.  t2[q1 * 94 + q2] = the consensus quality, 0xFF where the call does not come
// back with the observed base.  Such columns are the second most frequent kind the packed pass flags (1.1 per depth-8 family of `simulate`
// data: the read-through zone, where a random subset of the rows survives the overlap correction); answered from the table they need no
// k_call_full item.
constexpr uint32_t T2_DIM = 94, T2_BYTES = T2_DIM * T2_DIM;
inline void fill_t2(uint8_t* t2 /* [T2_BYTES] */, const ConsensusTables& t) {
  for (uint32_t q1 = 0; q1 < T2_DIM; q1++)
    for (uint32_t q2 = 0; q2 < T2_DIM; q2++) {
      uint8_t& out = t2[q1 * T2_DIM + q2];
      out = 0xFF;
      ColumnAcc acc;
      acc.reset();
      acc.add(0, t.correct[q1], t.error_per_alt[q1]);
      acc.add(0, t.correct[q2], t.error_per_alt[q2]);
      if (!m_isfinite(acc.s[0]) || !m_isfinite(acc.s[1])) continue;
      int bi = -1;
      uint8_t ql = 0;
      column_call(t, acc.s, acc.obs, &bi, &ql);
      if (bi == 0) out = ql;
    }
}

}  // namespace pk
}  // namespace fgx
