// inflate_core.h — a raw DEFLATE (RFC 1951) decoder for ONE BGZF block, written to run as one GPU lane per block (and, the same
// source, on the host for its unit tests).  It stands in, on the device, for what crates/fgumi-bgzf/src/reader.rs:346-479
// (`decompress_block*`) asks libdeflater to do.  A BGZF block is at most 64 KiB either way, so everything is 32-bit.
//
// Shape: a 64-bit bit buffer refilled with one unaligned 8-byte load, asked for one refill ahead (the input buffer must be READABLE
// 16 bytes past its end);
// literal / length and distance codes through small first-level tables (9 and 6 bits: 1.1 KB, in LDS on the device) with the
// canonical bit-by-bit walk for the rare longer codes (its state after the table's bits and the longer lengths' counts in registers, symbol[] in private memory); output bytes go straight to the destination (matches are copied from there: the window IS the output).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define FGX_HD __host__ __device__
#else
#define FGX_HD
#endif

namespace fgx {

#ifndef FGX_INFL_LIT_BITS
#define FGX_INFL_LIT_BITS 9    /* peeked bits of the literal / length first-level table */
#endif
#ifndef FGX_INFL_DIST_BITS
#define FGX_INFL_DIST_BITS 6   /* ... of the distance table (also holds the 19-symbol code-length code while a header is read) */
#endif
// the first-level tables: the part of a block's state every symbol touches (LDS on the device: this is what bounds the blocks in flight)
struct InflateFast {
  uint16_t lit[1u << FGX_INFL_LIT_BITS];     // peeked bits -> (symbol << 4) | code length; 0 = a longer code
  uint16_t dist[1u << FGX_INFL_DIST_BITS];
};
// the canonical tables behind them, walked bit by bit for the rare longer codes (private memory on the device)
struct InflateSlow {
  uint16_t lit_count[16], dist_count[16];
  uint16_t lit_sym[288];
  uint16_t dist_sym[32];
};
struct InflateTables { InflateFast f; InflateSlow w; };   // (both together: the host entry)

enum InflateStatus : int {
  INFL_OK = 0, INFL_BAD_BLOCK_TYPE = 1, INFL_BAD_STORED = 2, INFL_BAD_CODE_LENGTHS = 3, INFL_BAD_SYMBOL = 4, INFL_BAD_DISTANCE = 5,
  INFL_OUTPUT_OVERFLOW = 6, INFL_INPUT_OVERRUN = 7, INFL_SIZE_MISMATCH = 8
};

struct BitReader {
  const uint8_t* base; uint32_t len;   // the deflate payload
  uint32_t pos;                        // next byte to load
  uint64_t bb; uint32_t bc;            // bit buffer, valid bits
  uint64_t nxt;                        // the eight bytes at `pos`, loaded ahead of their use
};

FGX_HD inline uint64_t infl_load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
// at least 56 valid bits afterwards.  The bytes come from `nxt`, which was asked for at the PREVIOUS refill: on the device a refill is a
// global-memory round trip, and this way it runs under the five or so symbols decoded in between instead of in front of them.
// Reads at most 8 bytes past the payload, whatever the stream claims: once `pos` has passed the end (a corrupt or truncated payload
// that keeps decoding zero bits) the load address stays at base + len — for a valid stream the same bytes as before, since nothing
// beyond the payload is ever consumed — and the decoding loops give up as soon as `pos` proves the overrun (infl_overrun).
FGX_HD inline void infl_refill(BitReader& r) {
  r.bb |= r.nxt << r.bc;
  r.pos += (63u - r.bc) >> 3;
  r.bc |= 56u;
  r.nxt = infl_load64(r.base + (r.pos < r.len ? r.pos : r.len));
}
// after a refill (bc >= 56): more than 8 bytes past the end means bits beyond the payload HAVE been consumed (pos - bc / 8 > len)
FGX_HD inline bool infl_overrun(const BitReader& r) { return r.pos > r.len + 8u; }
FGX_HD inline void infl_seek(BitReader& r, uint32_t pos) { r.pos = pos; r.bb = 0; r.bc = 0; r.nxt = infl_load64(r.base + (pos < r.len ? pos : r.len)); }
FGX_HD inline void infl_open(BitReader& r, const uint8_t* in, uint32_t in_len, void*) { r.base = in; r.len = in_len; r.pos = 0; r.bb = 0; r.bc = 0; r.nxt = 0; infl_seek(r, 0); }
template <class RD>
FGX_HD inline uint32_t infl_bits(RD& r, uint32_t n) {   // n <= 16, bc >= n
  const uint32_t v = (uint32_t)(r.bb & ((1ull << n) - 1ull));
  r.bb >>= n; r.bc -= n;
  return v;
}
FGX_HD inline uint32_t infl_rev(uint32_t code, uint32_t len) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
  return r;
}

// canonical Huffman tables from code lengths (RFC 1951 3.2.2).  Returns false for an over-subscribed set; an incomplete set is
// allowed only for a single code (the one-distance-code case) — what zlib accepts.
template <class FastPtr, class SlowPtr>   // uint16_t* on the host; LDS (address space 3) pointers on the device: ds_read instead of a flat load
FGX_HD inline bool infl_build(const uint8_t* lens, uint32_t n, SlowPtr count, SlowPtr sym, FastPtr fast, uint32_t fast_bits) {
  for (uint32_t l = 0; l < 16; l++) count[l] = 0;
  for (uint32_t s = 0; s < n; s++) count[lens[s]]++;
  for (uint32_t i = 0; i < (1u << fast_bits); i++) fast[i] = 0;
  if (count[0] == n) return true;                    // no codes at all (legal for the distance set of an all-literal block)
  int32_t left = 1;
  for (uint32_t l = 1; l < 16; l++) { left <<= 1; left -= (int32_t)count[l]; if (left < 0) return false; }
  if (left > 0 && !(n - count[0] == 1 && count[1] == 1)) return false;
  uint16_t offs[16];
  offs[1] = 0;
  for (uint32_t l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
  for (uint32_t s = 0; s < n; s++) if (lens[s]) sym[offs[lens[s]]++] = (uint16_t)s;
  // first-level table: every symbol of length <= fast_bits at all the indices whose low bits are its (bit-reversed) code
  uint32_t code = 0, idx = 0;
  for (uint32_t l = 1; l <= fast_bits; l++) {
    for (uint32_t k = 0; k < count[l]; k++, code++, idx++) {
      const uint32_t r = infl_rev(code, l);
      const uint16_t e = (uint16_t)((sym[idx] << 4) | l);
      for (uint32_t hi = 0; hi < (1u << (fast_bits - l)); hi++) fast[r | (hi << l)] = e;
    }
    code <<= 1;
  }
  return true;
}

// What the canonical walk (puff.c's decode) needs for the codes LONGER than the first-level table, held in registers: the walk's state
// after the table's K bits (first code / first symbol index of length K + 1) and the counts of the lengths K + 1 .. 15, two per word.
// (The walk used to start at bit 1 and read count[] from private memory — a global-memory round trip per bit, and with sixteen lanes
// decoding in step nearly every step had a lane on it.)
struct InflWalk { uint32_t first, index; uint32_t cnt[5]; };
template <class SlowPtr>
FGX_HD inline void infl_walk_setup(SlowPtr count, uint32_t K, InflWalk& w) {
  uint32_t first = 0, index = 0;
  for (uint32_t l = 1; l <= K; l++) { index += count[l]; first = (first + count[l]) << 1; }
  w.first = first; w.index = index;
  for (uint32_t j = 0; j < 5; j++) {
    const uint32_t l0 = K + 1 + 2 * j, l1 = l0 + 1;
    w.cnt[j] = (l0 <= 15 ? (uint32_t)count[l0] : 0u) | ((l1 <= 15 ? (uint32_t)count[l1] : 0u) << 16);
  }
}
// one symbol: the first-level table (K peeked bits), else the walk from bit K + 1 on.  Returns the symbol or -1.
template <uint32_t K, class RD, class FastPtr, class SlowPtr>
FGX_HD inline int32_t infl_decode(RD& r, FastPtr fast, const InflWalk& w, SlowPtr sym) {
  static_assert(K >= 5 && K <= 14, "InflWalk holds the counts of at most ten lengths");
  const uint32_t e = fast[(uint32_t)r.bb & ((1u << K) - 1u)];
  if (e & 15u) { const uint32_t l = e & 15u; r.bb >>= l; r.bc -= l; return (int32_t)(e >> 4); }
  // the K bits already seen, first bit highest (a Huffman code is read from its most significant bit), times two: where the walk stands
  int32_t code = (int32_t)((__builtin_bitreverse32((uint32_t)r.bb) >> (32u - K)) << 1);
  r.bb >>= K; r.bc -= K;
  int32_t first = (int32_t)w.first, index = (int32_t)w.index;
#pragma unroll
  for (uint32_t j = 0; j < 15u - K; j++) {
    code |= (int32_t)(r.bb & 1u);
    r.bb >>= 1; r.bc -= 1;
    const int32_t cnt = (int32_t)((w.cnt[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu);
    if (code - cnt < first) return (int32_t)sym[index + (code - first)];
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -1;
}

// base value and extra bits of a length symbol (257 .. 285, here minus 257) and of a distance symbol (RFC 1951 3.2.5), computed: a
// table indexed by a lane's own symbol is a global-memory load on the device, two dependent ones per match.
FGX_HD constexpr uint32_t infl_len_extra(uint32_t s) { return (s < 8u || s == 28u) ? 0u : (s - 4u) >> 2; }
FGX_HD constexpr uint32_t infl_len_base(uint32_t s) { return s < 8u ? 3u + s : s == 28u ? 258u : 3u + ((4u + (s & 3u)) << ((s - 4u) >> 2)); }
FGX_HD constexpr uint32_t infl_dist_extra(uint32_t d) { return d < 4u ? 0u : (d - 2u) >> 1; }
FGX_HD constexpr uint32_t infl_dist_base(uint32_t d) { return d < 4u ? 1u + d : 1u + ((2u + (d & 1u)) << ((d - 2u) >> 1)); }
namespace infl_detail {
constexpr uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr bool formulas_match_rfc1951() {
  for (uint32_t s = 0; s < 29; s++) if (infl_len_base(s) != LEN_BASE[s] || infl_len_extra(s) != LEN_EXTRA[s]) return false;
  for (uint32_t d = 0; d < 30; d++) if (infl_dist_base(d) != DIST_BASE[d] || infl_dist_extra(d) != DIST_EXTRA[d]) return false;
  return true;
}
static_assert(formulas_match_rfc1951(), "length / distance formulas against the RFC's tables");
}  // namespace infl_detail

// A match: `len` bytes (3 .. 258) from `dist` bytes back.  Written so that NO load depends on a store of the same match — on the
// device every dependent load -> store -> load step is a memory round trip of most of a microsecond, and a byte-by-byte copy made
// the matches nine tenths of the decoder's time:
//   dist >= len   the source lies wholly before the match: pieces of up to 32 bytes, their loads first, then their stores (whole
//                 8-byte words while they fit, the tail byte by byte OUT OF THE REGISTERS);
//   dist <  len   the match repeats its own beginning: the `dist` bytes before it are the period; periods of up to 8 bytes are
//                 replicated out of one loaded word, longer ones re-read the period (which again lies before the match).
// Loads may run up to 7 bytes past the bytes they need (inside the output buffer or its slack, never stored).
FGX_HD inline void infl_store_bytes(uint8_t* dst, uint64_t v, uint32_t n) {          // n < 8 low bytes of v
  for (uint32_t k = 0; k < n; k++) dst[k] = (uint8_t)(v >> (8 * k));
}
FGX_HD inline void infl_copy_match(uint8_t* dst, uint32_t dist, uint32_t len) {
  const uint8_t* src = dst - dist;
  if (dist >= len) {
    for (uint32_t i = 0; i < len; i += 32) {
      const uint32_t n = len - i < 32u ? len - i : 32u;
      const uint64_t v0 = infl_load64(src + i), v1 = n > 8 ? infl_load64(src + i + 8) : 0ull, v2 = n > 16 ? infl_load64(src + i + 16) : 0ull,
                     v3 = n > 24 ? infl_load64(src + i + 24) : 0ull;
      if (n >= 8) memcpy(dst + i, &v0, 8); else { infl_store_bytes(dst + i, v0, n); return; }
      if (n >= 16) memcpy(dst + i + 8, &v1, 8); else { infl_store_bytes(dst + i + 8, v1, n - 8); continue; }
      if (n >= 24) memcpy(dst + i + 16, &v2, 8); else { infl_store_bytes(dst + i + 16, v2, n - 16); continue; }
      if (n >= 32) memcpy(dst + i + 24, &v3, 8); else infl_store_bytes(dst + i + 24, v3, n - 24);
    }
    return;
  }
  if (dist <= 8) {
    uint64_t pat = infl_load64(src);                            // the period: its low `dist` bytes
    if ((dist & (dist - 1)) == 0) {
      // a period of 1, 2, 4 or 8 bytes — byte runs, runs of 16- and 32-bit values: the usual long matches of BAM data — divides the
      // 8-byte word: the word is the period repeated, and the match is that word stored over and over (the lanes of a wavefront wait
      // for the one that is copying: a 258-byte run written byte by byte held the other fifteen up for 258 steps)
      if (dist == 1) pat = (pat & 0xFFull) * 0x0101010101010101ull;
      else if (dist == 2) pat = (pat & 0xFFFFull) * 0x0001000100010001ull;
      else if (dist == 4) pat = (pat & 0xFFFFFFFFull) | (pat << 32);
      uint32_t i = 0;
      for (; i + 8 <= len; i += 8) memcpy(dst + i, &pat, 8);
      infl_store_bytes(dst + i, pat, len - i);
      return;
    }
    uint32_t ph = 0;
    for (uint32_t i = 0; i < len; i++) { dst[i] = (uint8_t)(pat >> (8 * ph)); ph = ph + 1 == dist ? 0 : ph + 1; }
    return;
  }
  uint32_t ph = 0;                                              // 8 < dist < len: rare (a long match over a medium period)
  for (uint32_t i = 0; i < len; i++) { dst[i] = src[ph]; ph = ph + 1 == dist ? 0 : ph + 1; }
}

FGX_HD inline void infl_store_pending(uint8_t* dst, uint32_t n, uint64_t v0, uint64_t v1, uint64_t v2, uint64_t v3) {     // n <= 32 bytes
  if (n >= 8) memcpy(dst, &v0, 8); else { infl_store_bytes(dst, v0, n); return; }
  if (n >= 16) memcpy(dst + 8, &v1, 8); else { infl_store_bytes(dst + 8, v1, n - 8); return; }
  if (n >= 24) memcpy(dst + 16, &v2, 8); else { infl_store_bytes(dst + 16, v2, n - 16); return; }
  if (n >= 32) memcpy(dst + 24, &v3, 8); else infl_store_bytes(dst + 24, v3, n - 24);
}

// inflates `in[0 .. in_len)` into `out[0 .. out_len)`; the stream must produce exactly out_len bytes (the block's ISIZE).
// `in` must be readable for 8 bytes past in_len (no load goes further, also for corrupt input).  F / W: this lane's tables (LDS / private memory on the device).
//
// TOK (round 5, the two-phase form): the SAME decoder, but a match is not copied — it becomes a 32-bit ENTRY of the block's entry list,
//     entry = literals in front of it (0 .. 255) << 24 | (distance - 1) << 9 | (length - 2, 0 = no match: a run of 255 literals that goes on)
// while literals (and stored bytes) still go to their final place in `out`.  Nothing the decoder does then depends on a byte it wrote: no
// load -> store -> load round trip per match (2.3 us each on the device: what bound the one-phase kernel), only table lookups and stores.  A
// second pass (inflate_resolve below; k_bgzf_resolve on the device: a wavefront per block with the block in LDS) plays the entries.
// `ent` holds `ent_cap` entries; INFL_ENTRY_CAP covers every valid stream of up to 64 KiB (a match is >= 3 bytes, a literal-run entry 255).
constexpr uint32_t INFL_ENTRY_CAP = 65536 / 3 + 65536 / 255 + 2 * 258 + 64;      // 22 681
// the entries a block of `isize` bytes can need, rounded up to whole 64-byte lines of 32-bit entries: what the device gives a block's list
// (round 6: lists are laid out back to back by ISIZE, engine.h bgzf_inflate_plan; round 5 gave every block the 64 KiB worst case)
FGX_HD constexpr uint32_t infl_entry_cap(uint32_t isize) { return (isize / 3u + isize / 255u + 2u + 15u) & ~15u; }
FGX_HD inline uint32_t infl_entry(uint32_t lit, uint32_t dist, uint32_t len) { return (lit << 24) | ((dist - 1u) << 9) | (len - 2u); }
FGX_HD inline uint32_t infl_entry_lit(uint32_t e) { return e >> 24; }
FGX_HD inline uint32_t infl_entry_len(uint32_t e) { const uint32_t c = e & 511u; return c ? c + 2u : 0u; }
FGX_HD inline uint32_t infl_entry_dist(uint32_t e) { return ((e >> 9) & 0x7FFFu) + 1u; }

// RD: the bit reader (BitReader: 8-byte loads from `in`; the device's tokenizer brings its own, with a window of the payload in LDS);
// SlowPtr: where the canonical tables behind the first-level ones live (private memory / LDS).
template <class FastPtr, bool TOK = false, class RD = BitReader, class SlowPtr = uint16_t*>
FGX_HD inline int inflate_block_x(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, FastPtr f_lit, FastPtr f_dist,
                                  SlowPtr w_lit_count, SlowPtr w_dist_count, SlowPtr w_lit_sym, SlowPtr w_dist_sym, void* rd_arg,
                                  uint32_t* ent = nullptr, uint32_t ent_cap = 0, uint32_t* n_ent = nullptr) {
  constexpr uint32_t LB = FGX_INFL_LIT_BITS, DB = FGX_INFL_DIST_BITS;
  uint32_t ne = 0, lit_run = 0;                                                 // TOK: entries written, literals since the last entry
  if (TOK && n_ent) *n_ent = 0;
  static constexpr uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  RD r;
  infl_open(r, in, in_len, rd_arg);
  uint64_t pv0 = 0, pv1 = 0, pv2 = 0, pv3 = 0;                                  // a match whose bytes are loaded and not stored yet (p_len of them at p_pos)
  uint32_t p_pos = 0, p_len = 0;
  uint32_t pos = 0;
  for (;;) {
    infl_refill(r);
    const uint32_t bfinal = infl_bits(r, 1), btype = infl_bits(r, 2);
    if (btype == 0) {
      // stored: back to the byte boundary, LEN, NLEN, bytes
      const uint32_t drop = r.bc & 7u;
      r.bb >>= drop; r.bc -= drop;
      infl_refill(r);
      const uint32_t len = infl_bits(r, 16), nlen = infl_bits(r, 16);
      if ((len ^ 0xFFFFu) != nlen) return INFL_BAD_STORED;
      // the bytes still in the bit buffer belong to the payload: step the byte position back over them
      const uint32_t src = r.pos - (r.bc >> 3);
      if (src + len > in_len) return INFL_INPUT_OVERRUN;
      if (pos + len > out_len) return INFL_OUTPUT_OVERFLOW;
      for (uint32_t i = 0; i < len; i++) out[pos + i] = in[src + i];
      pos += len;
      if (TOK) {   // stored bytes count as literals
        lit_run += len;
        while (lit_run >= 255u) { if (ne >= ent_cap) return INFL_OUTPUT_OVERFLOW; ent[ne++] = 255u << 24; lit_run -= 255u; }
      }
      infl_seek(r, src + len);
    } else if (btype == 1 || btype == 2) {
      uint8_t lens[320];
      uint32_t hlit, hdist;
      if (btype == 1) {
        hlit = 288; hdist = 32;                        // (the fixed distance code is 32 five-bit codes; 30 and 31 never appear in valid data)
        for (uint32_t s = 0; s < 144; s++) lens[s] = 8;
        for (uint32_t s = 144; s < 256; s++) lens[s] = 9;
        for (uint32_t s = 256; s < 280; s++) lens[s] = 7;
        for (uint32_t s = 280; s < 288; s++) lens[s] = 8;
        for (uint32_t s = 0; s < 32; s++) lens[288 + s] = 5;
      } else {
        hlit = infl_bits(r, 5) + 257; hdist = infl_bits(r, 5) + 1;
        const uint32_t hclen = infl_bits(r, 4) + 4;
        if (hlit > 286 || hdist > 30) return INFL_BAD_CODE_LENGTHS;
        uint8_t cl[19];
        for (uint32_t i = 0; i < 19; i++) cl[i] = 0;
        infl_refill(r);
        for (uint32_t i = 0; i < hclen; i++) { if (r.bc < 3) infl_refill(r); cl[CL_ORDER[i]] = (uint8_t)infl_bits(r, 3); }
        // the code-length code: its tables live in the distance slots for the moment (19 symbols, up to 7 bits)
        if (!infl_build(cl, 19, w_dist_count, w_dist_sym, f_dist, DB)) return INFL_BAD_CODE_LENGTHS;
        InflWalk wc;
        infl_walk_setup(w_dist_count, DB, wc);
        uint32_t n = 0;
        while (n < hlit + hdist) {
          if (r.bc < 32) { infl_refill(r); if (infl_overrun(r)) return INFL_INPUT_OVERRUN; }
          const int32_t s = infl_decode<DB>(r, f_dist, wc, w_dist_sym);
          if (s < 0) return INFL_BAD_CODE_LENGTHS;
          if (s < 16) lens[n++] = (uint8_t)s;
          else {
            uint32_t rep, val = 0;
            if (s == 16) { if (n == 0) return INFL_BAD_CODE_LENGTHS; val = lens[n - 1]; rep = 3 + infl_bits(r, 2); }
            else if (s == 17) rep = 3 + infl_bits(r, 3);
            else rep = 11 + infl_bits(r, 7);
            if (n + rep > hlit + hdist) return INFL_BAD_CODE_LENGTHS;
            for (uint32_t i = 0; i < rep; i++) lens[n++] = (uint8_t)val;
          }
        }
        if (lens[256] == 0) return INFL_BAD_CODE_LENGTHS;        // no end-of-block code
      }
      if (!infl_build(lens, hlit, w_lit_count, w_lit_sym, f_lit, LB)) return INFL_BAD_CODE_LENGTHS;
      if (!infl_build(lens + hlit, hdist, w_dist_count, w_dist_sym, f_dist, DB)) return INFL_BAD_CODE_LENGTHS;
      InflWalk wl, wd;
      infl_walk_setup(w_lit_count, LB, wl);
      infl_walk_setup(w_dist_count, DB, wd);
      for (;;) {
        if (r.bc < 48) { infl_refill(r); if (infl_overrun(r)) return INFL_INPUT_OVERRUN; }   // a length + distance pair takes at most 15 + 5 + 15 + 13 = 48 bits
        int32_t s = infl_decode<LB>(r, f_lit, wl, w_lit_sym);
        if (s < 0) return INFL_BAD_SYMBOL;
        if (s < 256) {
          if (pos >= out_len) return INFL_OUTPUT_OVERFLOW;
          out[pos++] = (uint8_t)s;
          if (TOK) { if (++lit_run == 255u) { if (ne >= ent_cap) return INFL_OUTPUT_OVERFLOW; ent[ne++] = 255u << 24; lit_run = 0; } }
          continue;
        }
        if (s == 256) { if (p_len) { infl_store_pending(out + p_pos, p_len, pv0, pv1, pv2, pv3); p_len = 0; } break; }
        s -= 257;
        if (s >= 29) return INFL_BAD_SYMBOL;
        const uint32_t len = infl_len_base((uint32_t)s) + infl_bits(r, infl_len_extra((uint32_t)s));
        const int32_t d = infl_decode<DB>(r, f_dist, wd, w_dist_sym);
        if (d < 0 || d >= 30) return INFL_BAD_DISTANCE;
        const uint32_t dist = infl_dist_base((uint32_t)d) + infl_bits(r, infl_dist_extra((uint32_t)d));
        if (dist > pos) return INFL_BAD_DISTANCE;
        if (pos + len > out_len) return INFL_OUTPUT_OVERFLOW;
        if (TOK) {
          if (ne >= ent_cap) return INFL_OUTPUT_OVERFLOW;
          ent[ne++] = infl_entry(lit_run, dist, len);
          lit_run = 0;
        } else if (dist >= len && len <= 32) {
          // a short match whose source lies wholly before it (most matches): its bytes are LOADED now and STORED when the next match
          // arrives (or the block ends) — a wavefront's lanes run the loop in step, nearly every step has a lane with a match, and
          // waiting for the match's source inside the step made every step a global-memory round trip.  The stores of the match
          // before go out first, so a source that overlaps that match's destination reads what was just stored (a wavefront's memory
          // operations keep their order).
          if (p_len) infl_store_pending(out + p_pos, p_len, pv0, pv1, pv2, pv3);
          const uint8_t* src = out + pos - dist;
          pv0 = infl_load64(src);
          pv1 = len > 8 ? infl_load64(src + 8) : 0ull;
          pv2 = len > 16 ? infl_load64(src + 16) : 0ull;
          pv3 = len > 24 ? infl_load64(src + 24) : 0ull;
          p_pos = pos; p_len = len;
        } else {
          if (p_len) { infl_store_pending(out + p_pos, p_len, pv0, pv1, pv2, pv3); p_len = 0; }
          infl_copy_match(out + pos, dist, len);
        }
        pos += len;
      }
    } else return INFL_BAD_BLOCK_TYPE;
    if (r.pos - (r.bc >> 3) > in_len) return INFL_INPUT_OVERRUN;   // bits were taken from beyond the payload
    if (bfinal) break;
  }
  if (TOK) {
    if (lit_run) { if (ne >= ent_cap) return INFL_OUTPUT_OVERFLOW; ent[ne++] = lit_run << 24; }
    if (pos == out_len && n_ent) *n_ent = ne;                                    // (a block that failed leaves an EMPTY list: the second pass does nothing)
  }
  return pos == out_len ? INFL_OK : INFL_SIZE_MISMATCH;
}

template <class FastPtr, bool TOK = false>
FGX_HD inline int inflate_block_t(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, FastPtr f_lit, FastPtr f_dist, InflateSlow& W,
                                  uint32_t* ent = nullptr, uint32_t ent_cap = 0, uint32_t* n_ent = nullptr) {
  return inflate_block_x<FastPtr, TOK, BitReader, uint16_t*>(in, in_len, out, out_len, f_lit, f_dist, W.lit_count, W.dist_count, W.lit_sym, W.dist_sym, nullptr, ent, ent_cap,
                                                             n_ent);
}

// The second pass in its plain form (the host tests' reference for k_bgzf_resolve): the entries of one block played in order on `out`,
// whose literal bytes are in place.  Every index was checked by the decoder; the bounds here only keep a corrupt list harmless.
FGX_HD inline bool inflate_resolve(uint8_t* out, uint32_t out_len, const uint32_t* ent, uint32_t ne) {
  uint32_t pos = 0;
  for (uint32_t i = 0; i < ne; i++) {
    const uint32_t e = ent[i], lit = infl_entry_lit(e), len = infl_entry_len(e), dist = infl_entry_dist(e);
    pos += lit;
    if (len) {
      if (dist > pos || pos + len > out_len) return false;
      for (uint32_t k = 0; k < len; k++) out[pos + k] = out[pos - dist + k];
      pos += len;
    }
    if (pos > out_len) return false;
  }
  return pos == out_len || ne == 0;
}

// k_bgzf_resolve's SCHEDULE, lane by lane on the host (tests only; the kernel in bgzf_device.hip follows it line by line): the entries in
// batches of 64, lane j of a batch owning entry j.  A lane's match goes to dpos = (bytes before the batch) + (literals and matches of the
// lanes below) + its own literals, and copies from src = dpos - dist.  Rounds: with F = dpos of the lowest lane whose match is not copied
// yet, every byte below F is final (literals are in place, every lower match is done), so a lane may copy when its whole source —
// min(len, dist) bytes from src — lies below F; the lowest waiting lane always may (its source ends at its own dpos at the latest), so
// every round makes progress.  Lanes of one round write disjoint ranges and read below F only: they are independent, whatever the order.
inline bool inflate_resolve_wave_emulated(uint8_t* out, uint32_t out_len, const uint32_t* ent, uint32_t ne, uint32_t* rounds_out) {
  uint32_t run = 0, rounds = 0;
  for (uint32_t e0 = 0; e0 < ne; e0 += 64) {
    uint32_t dpos[64], len[64], dist[64];
    bool todo[64];
    uint32_t acc = run;
    for (uint32_t j = 0; j < 64; j++) {
      const uint32_t e = e0 + j < ne ? ent[e0 + j] : 0u;
      len[j] = infl_entry_len(e); dist[j] = infl_entry_dist(e);
      dpos[j] = acc + infl_entry_lit(e);
      acc = dpos[j] + len[j];
      todo[j] = len[j] != 0;
      if (todo[j] && (dist[j] > dpos[j] || dpos[j] + len[j] > out_len)) return false;
    }
    for (;;) {
      int first = -1;
      for (int j = 0; j < 64; j++) if (todo[j]) { first = j; break; }
      if (first < 0) break;
      const uint32_t F = dpos[first];
      bool ready[64];
      for (int j = 0; j < 64; j++) {
        const uint32_t span = len[j] < dist[j] ? len[j] : dist[j];
        ready[j] = todo[j] && (j == first || dpos[j] - dist[j] + span <= F);
      }
      for (int j = 63; j >= 0; j--) if (ready[j]) {          // (reverse order: the lanes of a round must not depend on one another)
        for (uint32_t k = 0; k < len[j]; k++) out[dpos[j] + k] = out[dpos[j] - dist[j] + k];
        todo[j] = false;
      }
      rounds++;
    }
    run = acc;
    if (run > out_len) return false;
  }
  if (rounds_out) *rounds_out = rounds;
  return run == out_len || ne == 0;
}

FGX_HD inline int inflate_block(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, InflateFast& F, InflateSlow& W) {
  return inflate_block_t<uint16_t*>(in, in_len, out, out_len, F.lit, F.dist, W);
}

// CRC-32 (IEEE 802.3, reflected, as gzip uses it): the byte-wise table and the pieces of zlib's crc32_combine (multiplication
// modulo the polynomial) that let 64 lanes check one block together.
FGX_HD inline uint32_t crc32_table_entry(uint32_t i) {
  uint32_t c = i;
  for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
  return c;
}
FGX_HD inline uint32_t crc32_multmodp(uint32_t a, uint32_t b) {   // a(x) * b(x) mod p(x), reflected
  uint32_t p = 0;
  for (uint32_t m = 1u << 31; m; m >>= 1) {             // (32 steps at most: a == 0 gives 0)
    if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
    b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return p;
}
// x^(8 n) mod p(x): the operator that moves a CRC past n more bytes
FGX_HD inline uint32_t crc32_shift_op(uint32_t n_bytes) {
  uint32_t p = 1u << 31;                 // x^0
  uint32_t sq = 0x00800000u;             // x^8 (one byte), squared per bit of n
  uint32_t n = n_bytes;
  while (n) {
    if (n & 1u) p = crc32_multmodp(sq, p);
    sq = crc32_multmodp(sq, sq);
    n >>= 1;
  }
  return p;
}

}  // namespace fgx
