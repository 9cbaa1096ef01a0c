// deflate_core.h — a raw DEFLATE (RFC 1951) COMPRESSOR for ONE BGZF block, written to run as one GPU lane per block (and, the same
// source, on the host for its unit tests).  It stands in, on the device, for the compress step of crates/fgumi-bgzf/src/writer.rs
// (libdeflater level 1 per <= 0xff00-byte block) on the consensus output.  Any valid DEFLATE stream is a correct answer — parity is
// "inflates to the same bytes, CRC-32 and ISIZE match" — so the choices here are speed choices:
//   * greedy LZ77 with ONE candidate per position: a 4096-entry table of the last position of each 4-byte hash (16-bit
//     positions: a block is < 64 KiB), matches extended eight bytes at a time, no chains, no lazy evaluation;
//   * one DYNAMIC Huffman code per block built from the block's own token statistics — consensus records are qualities, small
//     16-bit depth / error arrays and 4-bit packed bases: a handful of symbols carry the block, and the fixed code costs 8 bits for
//     every one of them;
//   * two passes over the block: tokens are written (4 bytes each) to a per-lane scratch and counted, the code is built (a
//     length-limited Huffman by sorting + the zlib "overflow repair"), then the tokens are replayed into the bit stream.
// Worst case output: a block that does not compress is returned as "does not fit" and the caller stores it.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define FGX_HD __host__ __device__
#else
#define FGX_HD
#endif

namespace fgx {

constexpr uint32_t DEFL_HASH_BITS = 12, DEFL_HASH_SIZE = 1u << DEFL_HASH_BITS;
constexpr uint32_t DEFL_MAX_TOKENS = 65536;                       // one per input byte at worst (+ end of block)

// per-lane working memory (global memory on the device: ~290 KB per block in flight)
struct DeflateScratch {
  uint16_t head[DEFL_HASH_SIZE];                                  // hash -> last position + 1 (0 = none)
  uint32_t tokens[DEFL_MAX_TOKENS];                               // literal: byte; match: 0x80000000 | (len - 3) << 16 | (dist - 1)
  uint32_t lit_freq[288], dist_freq[32];
  uint16_t lit_code[288], dist_code[32];
  uint8_t lit_len[288], dist_len[32];
};

struct BitWriter { uint8_t* out; uint32_t cap, pos; uint64_t acc; uint32_t n; bool overflow; };
FGX_HD inline void defl_put(BitWriter& w, uint32_t bits, uint32_t count) {        // count <= 32
  w.acc |= (uint64_t)bits << w.n;
  w.n += count;
  if (w.n >= 32) {
    if (w.pos + 4 <= w.cap) { const uint32_t v = (uint32_t)w.acc; memcpy(w.out + w.pos, &v, 4); } else w.overflow = true;
    w.pos += 4; w.acc >>= 32; w.n -= 32;
  }
}
FGX_HD inline void defl_flush(BitWriter& w) {
  while (w.n > 0) {
    if (w.pos < w.cap) w.out[w.pos] = (uint8_t)w.acc; else w.overflow = true;
    w.pos++; w.acc >>= 8; w.n = w.n > 8 ? w.n - 8 : 0;
  }
}
FGX_HD inline uint32_t defl_rev(uint32_t code, uint32_t len) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
  return r;
}
FGX_HD inline uint32_t defl_load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
FGX_HD inline uint64_t defl_load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// length -> (symbol - 257, extra bits, extra value); distance likewise (RFC 1951 3.2.5)
FGX_HD inline void defl_len_code(uint32_t len, uint32_t* sym, uint32_t* ebits, uint32_t* eval) {
  static constexpr uint16_t BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static constexpr uint8_t EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  uint32_t s;
  if (len == 258) s = 28;
  else if (len < 11) s = len - 3;
  else {
    const uint32_t l = len - 3;                                   // 8 .. 254
    const uint32_t hb = 31u - (uint32_t)__builtin_clz(l);         // 3 .. 7
    s = 4u * (hb - 1u) + ((l >> (hb - 2u)) & 3u);
  }
  *sym = s; *ebits = EXTRA[s]; *eval = len - BASE[s];
}
FGX_HD inline void defl_dist_code(uint32_t dist, uint32_t* sym, uint32_t* ebits, uint32_t* eval) {
  const uint32_t d = dist - 1;                                    // 0 .. 32767
  if (d < 4) { *sym = d; *ebits = 0; *eval = 0; return; }
  const uint32_t hb = 31u - (uint32_t)__builtin_clz(d);           // 2 .. 14
  const uint32_t s = 2u * hb + ((d >> (hb - 1u)) & 1u);
  *sym = s; *ebits = hb - 1u; *eval = d & ((1u << (hb - 1u)) - 1u);
}

// Code lengths (<= max_len) for `n` symbol frequencies: symbols sorted by frequency, a Huffman tree over the sorted list (the two
// queue method), depths limited to max_len exactly the way zlib's gen_bitlen does it.  Symbols with frequency 0 get length 0; a single used symbol gets length 1.
FGX_HD inline void defl_code_lengths(const uint32_t* freq, uint32_t n, uint32_t max_len, uint8_t* len_out, uint16_t* order /* n */, uint32_t* node_w /* 2n */,
                                     uint16_t* node_parent /* 2n */) {
  uint32_t m = 0;
  for (uint32_t s = 0; s < n; s++) { len_out[s] = 0; if (freq[s]) order[m++] = (uint16_t)s; }
  if (m == 0) return;
  if (m == 1) { len_out[order[0]] = 1; return; }
  // insertion sort by (frequency, symbol): m <= 288, and the lists are nearly sorted by nature only rarely — still a few 10^4 steps at most
  for (uint32_t i = 1; i < m; i++) {
    const uint16_t s = order[i];
    const uint32_t f = freq[s];
    uint32_t j = i;
    while (j > 0 && (freq[order[j - 1]] > f || (freq[order[j - 1]] == f && order[j - 1] > s))) { order[j] = order[j - 1]; j--; }
    order[j] = s;
  }
  // two-queue Huffman: leaves 0 .. m-1 (sorted), internal nodes m .. 2m-2 in creation order (also sorted)
  for (uint32_t i = 0; i < m; i++) node_w[i] = freq[order[i]];
  uint32_t leaf = 0, inner = m, next = m;
  auto take = [&]() -> uint32_t {
    if (leaf < m && (inner >= next || node_w[leaf] <= node_w[inner])) return leaf++;
    return inner++;
  };
  while (next < 2 * m - 1) {
    const uint32_t a = take(), b = take();
    node_w[next] = node_w[a] + node_w[b];
    node_parent[a] = (uint16_t)next; node_parent[b] = (uint16_t)next;
    next++;
  }
  // depths, zlib's gen_bitlen: top down (internal nodes were created in increasing order: a parent's index is above its children's),
  // a node deeper than max_len is put AT max_len and counted; then, two overflowing nodes at a time, a leaf of the deepest level
  // that still has one becomes an internal node whose children are that leaf and one of the overflowing ones — the code stays complete
  uint32_t bl_count[32];
  for (uint32_t l = 0; l < 32; l++) bl_count[l] = 0;
  node_w[2 * m - 2] = 0;                                          // (node_w is reused for depths from here on)
  int32_t overflow = 0;
  for (int32_t i = (int32_t)(2 * m - 3); i >= 0; i--) {
    uint32_t d = node_w[node_parent[i]] + 1;
    if (d > max_len) { d = max_len; overflow++; }
    node_w[i] = d;
    if ((uint32_t)i < m) bl_count[d]++;
  }
  if (overflow > 0) {
    do {
      uint32_t bits = max_len - 1;
      while (bl_count[bits] == 0) bits--;
      bl_count[bits]--; bl_count[bits + 1] += 2; bl_count[max_len]--;
      overflow -= 2;
    } while (overflow > 0);
    // hand the lengths out again: the longest codes to the rarest symbols
    uint32_t i = 0;
    for (uint32_t l = max_len; l >= 1; l--) for (uint32_t k = 0; k < bl_count[l]; k++) node_w[i++] = l;
  }
  for (uint32_t i = 0; i < m; i++) len_out[order[i]] = (uint8_t)node_w[i];
}
// canonical codes (bit-reversed, ready for the LSB-first stream) from lengths
FGX_HD inline void defl_canonical(const uint8_t* len, uint32_t n, uint16_t* code) {
  uint32_t cnt[16], nxt[16];
  for (uint32_t l = 0; l < 16; l++) cnt[l] = 0;
  for (uint32_t s = 0; s < n; s++) cnt[len[s]]++;
  cnt[0] = 0;
  uint32_t c = 0;
  for (uint32_t l = 1; l < 16; l++) { c = (c + cnt[l - 1]) << 1; nxt[l] = c; }
  for (uint32_t s = 0; s < n; s++) code[s] = len[s] ? (uint16_t)defl_rev(nxt[len[s]]++, len[s]) : (uint16_t)0;
}

// Compresses in[0 .. n) (n <= 65535; `in` readable for 8 bytes past n) into out[0 .. cap) as ONE final dynamic-Huffman block.
// Returns the number of bytes written, or 0 when the stream does not fit `cap` (store the block instead).
FGX_HD inline uint32_t deflate_block(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, DeflateScratch& S) {
  // ---- pass 1: tokens + statistics ------------------------------------------------------------------------------------------------
  for (uint32_t i = 0; i < DEFL_HASH_SIZE; i++) S.head[i] = 0;
  for (uint32_t i = 0; i < 288; i++) S.lit_freq[i] = 0;
  for (uint32_t i = 0; i < 32; i++) S.dist_freq[i] = 0;
  uint32_t nt = 0, i = 0;
  while (i < n) {
    uint32_t best_len = 0, best_dist = 0;
    if (i + 4 <= n) {
      const uint32_t v = defl_load32(in + i);
      const uint32_t h = (v * 2654435761u) >> (32 - DEFL_HASH_BITS);
      const uint32_t c = S.head[h];
      S.head[h] = (uint16_t)(i + 1);
      if (c != 0 && i + 1 - c <= 32768u && defl_load32(in + c - 1) == v) {
        const uint8_t* a = in + i; const uint8_t* b = in + c - 1;
        const uint32_t maxl = n - i < 258u ? n - i : 258u;
        uint32_t l = 4;
        while (l + 8 <= maxl) {
          const uint64_t x = defl_load64(a + l) ^ defl_load64(b + l);
          if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; goto done; }
          l += 8;
        }
        while (l < maxl && a[l] == b[l]) l++;
      done:
        best_len = l < maxl ? l : maxl; best_dist = i + 1 - c;
      }
    }
    if (best_len >= 4) {
      uint32_t ls, le, lv, ds, de, dv;
      defl_len_code(best_len, &ls, &le, &lv);
      defl_dist_code(best_dist, &ds, &de, &dv);
      S.lit_freq[257 + ls]++; S.dist_freq[ds]++;
      S.tokens[nt++] = 0x80000000u | ((best_len - 3) << 16) | (best_dist - 1);
      // (the positions a match covers are not entered into the table, but its last one is: the next match often starts right there)
      const uint32_t e = i + best_len;
      if (e >= 1 && e - 1 + 4 <= n) { const uint32_t v2 = defl_load32(in + e - 1); S.head[(v2 * 2654435761u) >> (32 - DEFL_HASH_BITS)] = (uint16_t)e; }
      i = e;
    } else {
      S.lit_freq[in[i]]++;
      S.tokens[nt++] = in[i];
      i++;
    }
  }
  S.lit_freq[256] = 1;
  // ---- the block's codes -----------------------------------------------------------------------------------------------------------
  {
    // (the code-length routine's work arrays live in the token area's tail: tokens use nt <= n entries, the arrays need < 2000)
    uint32_t* work = S.tokens + DEFL_MAX_TOKENS - 2048;
    if (nt > DEFL_MAX_TOKENS - 2048) return 0;
    uint16_t* order = (uint16_t*)work; uint32_t* node_w = work + 160; uint16_t* node_parent = (uint16_t*)(work + 160 + 600);
    defl_code_lengths(S.lit_freq, 286, 15, S.lit_len, order, node_w, node_parent);
    S.lit_len[286] = S.lit_len[287] = 0;
    defl_code_lengths(S.dist_freq, 30, 15, S.dist_len, order, node_w, node_parent);
    S.dist_len[30] = S.dist_len[31] = 0;
  }
  uint32_t n_dist_used = 0;
  for (uint32_t s = 0; s < 30; s++) n_dist_used += S.dist_len[s] != 0;
  if (n_dist_used == 0) S.dist_len[0] = 1;                        // (at least one distance code must be described)
  defl_canonical(S.lit_len, 286, S.lit_code);
  defl_canonical(S.dist_len, 30, S.dist_code);
  uint32_t hlit = 286, hdist = 30;
  while (hlit > 257 && S.lit_len[hlit - 1] == 0) hlit--;
  while (hdist > 1 && S.dist_len[hdist - 1] == 0) hdist--;
  // ---- header: the code lengths, themselves run-length coded (symbols 16 / 17 / 18) and Huffman coded (a 19-symbol code of <= 7 bits)
  uint8_t lens[320];
  for (uint32_t s = 0; s < hlit; s++) lens[s] = S.lit_len[s];
  for (uint32_t s = 0; s < hdist; s++) lens[hlit + s] = S.dist_len[s];
  const uint32_t nl = hlit + hdist;
  uint16_t rle[320];                                              // symbol | extra << 5
  uint32_t nr = 0, cl_freq[19];
  for (uint32_t s = 0; s < 19; s++) cl_freq[s] = 0;
  for (uint32_t k = 0; k < nl;) {
    const uint32_t v = lens[k];
    uint32_t run = 1;
    while (k + run < nl && lens[k + run] == v) run++;
    uint32_t left = run;
    if (v == 0) {
      while (left >= 11) { const uint32_t r = left < 138u ? left : 138u; rle[nr++] = (uint16_t)(18u | ((r - 11u) << 5)); cl_freq[18]++; left -= r; }
      if (left >= 3) { rle[nr++] = (uint16_t)(17u | ((left - 3u) << 5)); cl_freq[17]++; left = 0; }
      while (left--) { rle[nr++] = 0; cl_freq[0]++; }
    } else {
      rle[nr++] = (uint16_t)v; cl_freq[v]++; left--;
      while (left >= 3) { const uint32_t r = left < 6u ? left : 6u; rle[nr++] = (uint16_t)(16u | ((r - 3u) << 5)); cl_freq[16]++; left -= r; }
      while (left--) { rle[nr++] = (uint16_t)v; cl_freq[v]++; }
    }
    k += run;
  }
  uint8_t cl_len[19]; uint16_t cl_code[19];
  {
    uint16_t order[19]; uint32_t node_w[40]; uint16_t node_parent[40];
    defl_code_lengths(cl_freq, 19, 7, cl_len, order, node_w, node_parent);
  }
  defl_canonical(cl_len, 19, cl_code);
  static constexpr uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  uint32_t hclen = 19;
  while (hclen > 4 && cl_len[CL_ORDER[hclen - 1]] == 0) hclen--;
  BitWriter w{out, cap, 0u, 0ull, 0u, false};
  defl_put(w, 1u | (2u << 1), 3);                                 // BFINAL = 1, BTYPE = 10 (dynamic)
  defl_put(w, hlit - 257, 5); defl_put(w, hdist - 1, 5); defl_put(w, hclen - 4, 4);
  for (uint32_t k = 0; k < hclen; k++) defl_put(w, cl_len[CL_ORDER[k]], 3);
  for (uint32_t k = 0; k < nr; k++) {
    const uint32_t s = rle[k] & 31u, x = rle[k] >> 5;
    defl_put(w, cl_code[s], cl_len[s]);
    if (s == 16) defl_put(w, x, 2); else if (s == 17) defl_put(w, x, 3); else if (s == 18) defl_put(w, x, 7);
  }
  // ---- pass 2: the tokens into the bit stream ------------------------------------------------------------------------------------------
  for (uint32_t k = 0; k < nt; k++) {
    const uint32_t t = S.tokens[k];
    if (t & 0x80000000u) {
      const uint32_t len = ((t >> 16) & 0x7FFFu) + 3, dist = (t & 0xFFFFu) + 1;
      uint32_t ls, le, lv, ds, de, dv;
      defl_len_code(len, &ls, &le, &lv);
      defl_dist_code(dist, &ds, &de, &dv);
      defl_put(w, (uint32_t)S.lit_code[257 + ls] | (lv << S.lit_len[257 + ls]), S.lit_len[257 + ls] + le);      // <= 15 + 5 bits
      defl_put(w, (uint32_t)S.dist_code[ds] | (dv << S.dist_len[ds]), S.dist_len[ds] + de);                      // <= 15 + 13 bits
    } else defl_put(w, S.lit_code[t], S.lit_len[t]);
    if (w.overflow) return 0;
  }
  defl_put(w, S.lit_code[256], S.lit_len[256]);
  defl_flush(w);
  return w.overflow ? 0u : w.pos;
}

}  // namespace fgx
