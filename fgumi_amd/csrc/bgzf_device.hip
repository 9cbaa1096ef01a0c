// bgzf_device.hip — BGZF inflate on the device: the compressed blocks of a BAM file go over PCIe (a third of the bytes), and the
// record stream is produced where the consensus kernels read it.  Stands in, on the device, for the decompress step of
// crates/fgumi-bgzf/src/reader.rs:346-479 (`decompress_block*`: libdeflater inflate + CRC32 / ISIZE check per block).
//
//   k_bgzf_inflate   a LANE per BGZF block (blocks are independent raw DEFLATE streams of at most 64 KiB), eight lanes per
//                    workgroup: a DEFLATE stream is sequential, so the parallelism is across blocks — a 5 GB chunk has 80 000 of them.
//                    Each lane keeps its first-level decode tables (inflate_core.h, 1.1 KB) in LDS: 128 blocks per CU in flight, whatever
//                    the workgroup size.  The lanes of a wavefront diverge (every stream takes its own path) and wait for one another's
//                    loads, so FEWER lanes per wavefront and more wavefronts per SIMD is the faster split of those 128 (measured,
//                    5.1 GB in four chunks: 64 lanes 0.196 s, 32 0.165 s, 16 0.144 s, 8 0.129 s); the kernel is bound by the latency
//                    of one lane's chain of symbols (a 64 KiB block takes ~30 ms), which the blocks in flight hide.
//   k_bgzf_crc       a WAVEFRONT per block: every lane runs the table-driven CRC-32 over its 1/64 of the block, and the 64 values
//                    are folded with the polynomial arithmetic of zlib's crc32_combine (slices are aligned to the END of the
//                    block, so every right-hand operand of the fold is a whole number of full slices).
#include <hipcub/hipcub.hpp>
#include "engine.h"
#include "inflate_core.h"
#include "deflate_core.h"

namespace fgx {

namespace {

#ifndef FGX_INFL_LANES
#define FGX_INFL_LANES 8
#endif
constexpr uint32_t INFL_LANES = FGX_INFL_LANES;

template <uint32_t INFL_LANES>
__global__ __launch_bounds__(INFL_LANES) void k_bgzf_inflate(const uint8_t* __restrict__ raw, const BgzfDevBlock* __restrict__ blk, uint32_t n,
                                                             uint8_t* __restrict__ out, uint32_t* __restrict__ status) {
  __shared__ InflateFast sF[INFL_LANES];
  InflateSlow W;                                                // (private: touched by the rare codes longer than the first-level tables)
  const uint32_t b = blockIdx.x * INFL_LANES + threadIdx.x;
  if (b >= n) return;
  const BgzfDevBlock B = blk[b];
  if (B.isize == 0) return;
  typedef __attribute__((address_space(3))) uint16_t* LdsPtr;     // (typed LDS pointers: table lookups are ds_read, not flat loads)
  const int st = inflate_block_t<LdsPtr>(raw + B.in_off, B.in_len, out + B.out_off, B.isize, (LdsPtr)sF[threadIdx.x].lit, (LdsPtr)sF[threadIdx.x].dist, W);
  if (st != INFL_OK) atomicMax(status, ((b + 1u) << 4) | (uint32_t)st);      // (which block, why: the highest failing block wins)
}

// ---- the two-phase form (round 5) ----------------------------------------------------------------------------------------------------------
// The one-phase kernel above is a chain of ~12 800 symbols per 64 KiB block of BAM records, three quarters of them matches, and every match
// is a global-memory round trip (its source was written moments ago by the same lane): 2.3 us per symbol, 25 - 31 GB/s for the whole chip
// however many blocks are in flight (profiles/r04_experiments.md).  Two passes take the round trip out:
//   k_bgzf_tokenize  a LANE per block, the same decoder with TOK set (inflate_core.h): literals go to their place, a match becomes a
//                    32-bit entry of the block's list.  No load depends on a store: table lookups in LDS, stores that nobody waits for.
//   k_bgzf_resolve   a WAVEFRONT per block with the WHOLE block in LDS (64 KiB: two blocks per CU): the block's bytes (literals in place)
//                    come in with 16-byte loads, the entries are played 64 at a time — a lane per match, a byte loop in LDS, in rounds
//                    ordered by the frontier rule (inflate_resolve_wave_emulated in inflate_core.h is this schedule on the host) — and the
//                    block leaves with 16-byte stores.  A match's round trip is an LDS access.  k_bgzf_crc checks the result as before.
static_assert(INFL_ENTRY_CAP >= 65536 / 3 + 65536 / 255 + 2, "infl_entry_cap (engine.h) is the per-ISIZE form of INFL_ENTRY_CAP");

// The tokenizer's bit reader: a WINDOW of the payload in LDS (INFL_WIN bytes per lane, filled with 16-byte loads), refills out of it.  The
// one-phase kernel's reader loads 8 bytes from global memory per refill — every ~5 symbols — and on this part a load's s_waitcnt also
// waits for every store issued before it (vmcnt counts both): with the tokenizer's stores (a literal, an entry per symbol) that was a
// store's round trip per refill.  With the window the wait comes once per ~70 symbols.  Same contract as BitReader (inflate_core.h): the
// address stays at base + min(pos, len), nothing beyond len + 24 is read (the raw buffer carries 64 bytes of slack), infl_overrun as there.
constexpr uint32_t INFL_WIN = 128;
typedef __attribute__((address_space(3))) uint8_t* LdsBytes;
struct LdsBitReader {
  const uint8_t* base; uint32_t len;
  uint32_t pos; uint64_t bb; uint32_t bc;
  LdsBytes win; uint32_t wbase, wvalid;           // payload bytes [wbase, wbase + INFL_WIN) are in the window (wvalid = 0: nothing yet)
};
__device__ __forceinline__ void infl_fill_window(LdsBitReader& r, uint32_t at) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) u32x4* LdsV;
  r.wbase = at & ~15u;
  r.wvalid = 1;
  // all the loads first (addresses clamped to len + 8: a piece beyond that is never read — the refill's address stays at or below len), ONE
  // wait, then the LDS stores; piece by piece, each behind its own predicate, the compiler made eight round trips of it
  u32x4 v[INFL_WIN / 16];
  const uint32_t lim = r.len + 8u;
#pragma unroll
  for (uint32_t i = 0; i < INFL_WIN / 16; i++) {
    const uint32_t off = r.wbase + 16u * i;
    __builtin_memcpy(&v[i], r.base + (off < lim ? off : lim), 16);
  }
#pragma unroll
  for (uint32_t i = 0; i < INFL_WIN / 16; i++) *(LdsV)(r.win + 16u * i) = v[i];
}
__device__ __forceinline__ void infl_refill(LdsBitReader& r) {
  const uint32_t a = r.pos < r.len ? r.pos : r.len;
  if (!r.wvalid || a < r.wbase || a + 8u > r.wbase + INFL_WIN) infl_fill_window(r, a);
  uint64_t v;
  {
    typedef __attribute__((address_space(3))) uint32_t* LdsW;
    const LdsBytes p = r.win + (a - r.wbase);
    uint32_t lo, hi;
    __builtin_memcpy(&lo, (const void*)(uint8_t*)p, 4); __builtin_memcpy(&hi, (const void*)((uint8_t*)p + 4), 4);
    v = (uint64_t)lo | ((uint64_t)hi << 32);
  }
  r.bb |= v << r.bc;
  r.pos += (63u - r.bc) >> 3;
  r.bc |= 56u;
}
__device__ __forceinline__ bool infl_overrun(const LdsBitReader& r) { return r.pos > r.len + 8u; }
__device__ __forceinline__ void infl_seek(LdsBitReader& r, uint32_t pos) { r.pos = pos; r.bb = 0; r.bc = 0; }
__device__ __forceinline__ void infl_open(LdsBitReader& r, const uint8_t* in, uint32_t in_len, void* win) {
  r.base = in; r.len = in_len; r.pos = 0; r.bb = 0; r.bc = 0; r.win = (LdsBytes)(uint8_t*)win; r.wbase = 0; r.wvalid = 0;
}

template <uint32_t LANES>
__global__ __launch_bounds__(LANES) void k_bgzf_tokenize(const uint8_t* __restrict__ raw, const BgzfDevBlock* __restrict__ blk, uint32_t n,
                                                         uint8_t* __restrict__ out, uint32_t* __restrict__ ent, uint32_t* __restrict__ n_ent,
                                                         uint32_t* __restrict__ status) {
  __shared__ InflateFast sF[LANES];
  __shared__ InflateSlow sW[LANES];                             // (LDS too: a private-memory load in the rare long-code path made EVERY symbol wait for memory)
  __shared__ __align__(16) uint8_t sWin[LANES][INFL_WIN];
  const uint32_t b = blockIdx.x * LANES + threadIdx.x;
  if (b >= n) return;
  const BgzfDevBlock B = blk[b];
  n_ent[b] = 0;
  if (B.isize == 0) return;
  typedef __attribute__((address_space(3))) uint16_t* LdsPtr;
  uint32_t ne = 0;
  InflateSlow& W = sW[threadIdx.x];
  const int st = inflate_block_x<LdsPtr, true, LdsBitReader, LdsPtr>(raw + B.in_off, B.in_len, out + B.out_off, B.isize, (LdsPtr)sF[threadIdx.x].lit, (LdsPtr)sF[threadIdx.x].dist,
                                                                     (LdsPtr)W.lit_count, (LdsPtr)W.dist_count, (LdsPtr)W.lit_sym, (LdsPtr)W.dist_sym, (void*)&sWin[threadIdx.x][0],
                                                                     ent + B.ent_off, infl_entry_cap(B.isize), &ne);
  if (st != INFL_OK) { atomicMax(status, ((b + 1u) << 4) | (uint32_t)st); ne = 0; }
  n_ent[b] = ne;
}

// inclusive prefix sum over the wavefront (values < 2^20: a block is 64 KiB)
__device__ __forceinline__ uint32_t infl_wave_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (uint32_t o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, (int)o); if (lane >= o) v += t; }
  return v;
}

__global__ __launch_bounds__(64) void k_bgzf_resolve(const BgzfDevBlock* __restrict__ blk, uint32_t n, uint8_t* __restrict__ out, const uint32_t* __restrict__ ent,
                                                     const uint32_t* __restrict__ n_ent, uint32_t* __restrict__ status) {
  extern __shared__ __align__(16) uint8_t L[];                  // the block (up to 64 KiB) + 64 bytes of slack for the copies' whole-word loads
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  if (b >= n) return;
  const BgzfDevBlock B = blk[b];
  const uint32_t isize = B.isize;
  if (isize == 0 || isize > 65536u) return;                     // (k_bgzf_crc checks the empty block's CRC; a BGZF block never exceeds 64 KiB)
  const uint32_t ne = n_ent[b];
  uint8_t* const g = out + B.out_off;
  // 1. the block as the decoder left it: literals in place, the matches' bytes still undefined (whole 16-byte pieces: the last one may
  //    take up to 15 bytes of the next block or of the buffer's slack along — they are never stored back)
  for (uint32_t i = 16u * lane; i < isize; i += 1024u) { u32x4 v; __builtin_memcpy(&v, g + i, 16); *(u32x4*)(L + i) = v; }
  __syncthreads();
  // 2. the entries, 64 at a time
  uint32_t run = 0;
  const uint32_t* const E = ent + B.ent_off;
  bool bad = false;
  for (uint32_t e0 = 0; e0 < ne; e0 += 64) {
    const uint32_t e = e0 + lane < ne ? E[e0 + lane] : 0u;
    const uint32_t lit = infl_entry_lit(e), len = infl_entry_len(e), dist = infl_entry_dist(e);
    const uint32_t incl = infl_wave_scan(lit + len, lane);
    const uint32_t dpos = run + incl - len;                     // where this lane's match begins
    run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    bool todo = len != 0;
    if (todo && (dist > dpos || dpos + len > isize)) { bad = true; todo = false; }   // (the decoder checked both: a corrupt list stays harmless)
    const uint32_t src = dpos - dist, span = len < dist ? len : dist;
    unsigned long long w = __ballot(todo);
    while (w) {
      const uint32_t first = (uint32_t)__builtin_ctzll(w);
      const uint32_t F = (uint32_t)__builtin_amdgcn_readlane((int)dpos, (int)first);   // every byte below F is final
      const bool ready = todo && (lane == first || src + span <= F);
      if (ready) {
        if (dist >= len) {
          // the source lies wholly below the match (nearly every match of BAM data): up to 32 bytes per step, the loads first, then the
          // stores — two LDS round trips per step instead of one per byte (loads may take up to 7 bytes more than they need: never stored)
          for (uint32_t k = 0; k < len; k += 32u) {
            const uint32_t nb = len - k < 32u ? len - k : 32u;
            const uint8_t* const sp = L + src + k;
            uint8_t* const dp = L + dpos + k;
            unsigned long long v0, v1 = 0, v2 = 0, v3 = 0;
            __builtin_memcpy(&v0, sp, 8);
            if (nb > 8u) __builtin_memcpy(&v1, sp + 8, 8);
            if (nb > 16u) __builtin_memcpy(&v2, sp + 16, 8);
            if (nb > 24u) __builtin_memcpy(&v3, sp + 24, 8);
            unsigned long long tail = v0;
            uint32_t done = 0;
            if (nb >= 8u) { __builtin_memcpy(dp, &v0, 8); tail = v1; done = 8; }
            if (nb >= 16u) { __builtin_memcpy(dp + 8, &v1, 8); tail = v2; done = 16; }
            if (nb >= 24u) { __builtin_memcpy(dp + 16, &v2, 8); tail = v3; done = 24; }
            if (nb >= 32u) { __builtin_memcpy(dp + 24, &v3, 8); done = 32; }
            uint32_t rest = nb - done;                            // 0 .. 7 bytes out of `tail`
            uint8_t* tp = dp + done;
            if (rest & 4u) { const uint32_t w4 = (uint32_t)tail; __builtin_memcpy(tp, &w4, 4); tp += 4; tail >>= 32; }
            if (rest & 2u) { const uint16_t w2 = (uint16_t)tail; __builtin_memcpy(tp, &w2, 2); tp += 2; tail >>= 16; }
            if (rest & 1u) *tp = (uint8_t)tail;
          }
        } else if (dist == 1u) {
          // a run of one byte (qualities, padding): the byte replicated, eight at a time
          const unsigned long long pat = (unsigned long long)L[src] * 0x0101010101010101ull;
          uint32_t k = 0;
          for (; k + 8u <= len; k += 8u) __builtin_memcpy(L + dpos + k, &pat, 8);
          for (; k < len; k++) L[dpos + k] = (uint8_t)pat;
        } else {
          for (uint32_t k = 0; k < len; k++) L[dpos + k] = L[src + k];      // the match repeats its own beginning: byte by byte, in order
        }
        todo = false;
      }
      __syncthreads();                                          // (one wavefront: orders the LDS writes of this round before the next round's reads)
      w = __ballot(todo);
    }
  }
  if (ne == 0) return;                                          // (the decoder has reported this block: its list is empty)
  if (__any(bad) || run != isize) { if (lane == 0) atomicMax(status, ((b + 1u) << 4) | (uint32_t)INFL_SIZE_MISMATCH); return; }
  __syncthreads();
  // 4. the block to its place: whole 16-byte pieces, the last bytes one by one (the next block's bytes lie right behind)
  const uint32_t whole = isize & ~15u;
  for (uint32_t i = 16u * lane; i < whole; i += 1024u) { const u32x4 v = *(const u32x4*)(L + i); __builtin_memcpy(g + i, &v, 16); }
  if (lane < isize - whole) g[whole + lane] = L[whole + lane];
}

// The same schedule on GLOBAL memory (the default): a wavefront per block, the block where it lies in `out` (its literals are in place), no LDS at
// all — so a CU holds 32 blocks instead of the two that fit its LDS.  A round is then a cache round trip instead of an LDS access, but the
// pass is bound by how many blocks are in flight, not by the latency of one (k_bgzf_resolve: 10.4 ms per 11 738 blocks with 512 in flight).
// Ordering: the lanes of a round read below the frontier and write above it; the stores of a round are made visible to the wavefront's
// later loads by a workgroup-scope release / acquire fence (one wavefront = one workgroup: the same L1).
__global__ __launch_bounds__(64) void k_bgzf_resolve_global(const BgzfDevBlock* __restrict__ blk, uint32_t n, uint8_t* out, const uint32_t* __restrict__ ent,
                                                            const uint32_t* __restrict__ n_ent, uint32_t* __restrict__ status) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  if (b >= n) return;
  const BgzfDevBlock B = blk[b];
  const uint32_t isize = B.isize;
  if (isize == 0 || isize > 65536u) return;
  const uint32_t ne = n_ent[b];
  if (ne == 0) return;                                          // (the decoder has reported this block: its list is empty)
  uint8_t* const L = out + B.out_off;
  uint32_t run = 0;
  const uint32_t* const E = ent + B.ent_off;
  bool bad = false;
  for (uint32_t e0 = 0; e0 < ne; e0 += 64) {
    const uint32_t e = e0 + lane < ne ? E[e0 + lane] : 0u;
    const uint32_t lit = infl_entry_lit(e), len = infl_entry_len(e), dist = infl_entry_dist(e);
    const uint32_t incl = infl_wave_scan(lit + len, lane);
    const uint32_t dpos = run + incl - len;
    run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    bool todo = len != 0;
    if (todo && (dist > dpos || dpos + len > isize)) { bad = true; todo = false; }
    const uint32_t src = dpos - dist, span = len < dist ? len : dist;
    unsigned long long w = __ballot(todo);
    while (w) {
      const uint32_t first = (uint32_t)__builtin_ctzll(w);
      const uint32_t F = (uint32_t)__builtin_amdgcn_readlane((int)dpos, (int)first);
      const bool ready = todo && (lane == first || src + span <= F);
      if (ready) {
        if (dist >= len) {
          for (uint32_t k = 0; k < len; k += 32u) {
            const uint32_t nb = len - k < 32u ? len - k : 32u;
            const uint8_t* const sp = L + src + k;
            uint8_t* const dp = L + dpos + k;
            unsigned long long v0, v1 = 0, v2 = 0, v3 = 0;      // (loads may take up to 7 bytes more than they need: inside the stream or its slack, never stored)
            __builtin_memcpy(&v0, sp, 8);
            if (nb > 8u) __builtin_memcpy(&v1, sp + 8, 8);
            if (nb > 16u) __builtin_memcpy(&v2, sp + 16, 8);
            if (nb > 24u) __builtin_memcpy(&v3, sp + 24, 8);
            infl_store_pending(dp, nb, v0, v1, v2, v3);
          }
        } else infl_copy_match(L + dpos, dist, len);            // the match repeats its own beginning (one lane, in order)
        todo = false;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      w = __ballot(todo);
    }
  }
  if (__any(bad) || run != isize) { if (lane == 0) atomicMax(status, ((b + 1u) << 4) | (uint32_t)INFL_SIZE_MISMATCH); }
}

// CRC-32 of p[0 .. len) by one wavefront (len > 0): every lane its 1/64 slice (slices end at the block's end), then the fold.  The
// result is valid in lane 0.
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t* p, uint32_t len, const uint32_t* tab, uint32_t lane) {
  const uint32_t S = (len + 63u) / 64u;                                    // bytes per lane
  const int64_t hi_s = (int64_t)len - (int64_t)(63u - lane) * S, lo_s = hi_s - (int64_t)S;
  const uint32_t hi = hi_s > 0 ? (uint32_t)hi_s : 0u, lo = lo_s > 0 ? (uint32_t)lo_s : 0u;
  uint32_t crc = 0;
  if (hi > lo) {
    // sixteen bytes per load, the next piece asked for before this one is folded in (a byte per load was a memory round trip per byte)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto ld16 = [&](uint32_t i) { u32x4 v; __builtin_memcpy(&v, p + i, 16); return v; };   // (may read up to 15 bytes past `hi`: inside the stream or its slack)
    crc = 0xFFFFFFFFu;
    uint32_t i = lo;
    u32x4 cur = ld16(i);
    while (i < hi) {
      const uint32_t n = hi - i < 16u ? hi - i : 16u;
      const u32x4 nxt = (i + 16 < hi) ? ld16(i + 16) : cur;
      const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) if (k < n) crc = tab[(crc ^ (w[k >> 2] >> (8 * (k & 3)))) & 0xFFu] ^ (crc >> 8);
      cur = nxt; i += 16;
    }
    crc ^= 0xFFFFFFFFu;
  }
  // fold: at every level the left lane of a pair takes crc(left || right) = left * x^(8 * |right|) + right
  uint32_t op = crc32_shift_op(S);
  for (uint32_t step = 1; step < 64; step <<= 1) {
    const uint32_t right = (uint32_t)__shfl_down((int)crc, (int)step);
    if ((lane & (2 * step - 1)) == 0) crc = crc32_multmodp(op, crc) ^ right;
    op = crc32_multmodp(op, op);
  }
  return crc;
}

__global__ __launch_bounds__(256) void k_bgzf_crc(const uint8_t* __restrict__ out, const BgzfDevBlock* __restrict__ blk, uint32_t n,
                                                  uint32_t* __restrict__ status) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
  __syncthreads();
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= n) return;
  const BgzfDevBlock B = blk[b];
  const uint32_t len = B.isize;
  if (len == 0) { if (lane == 0 && B.crc != 0) atomicMax(status, ((b + 1u) << 4) | 9u); return; }
  const uint32_t crc = wave_crc32(out + B.out_off, len, tab, lane);
  if (lane == 0 && crc != B.crc) atomicMax(status, ((b + 1u) << 4) | 9u);   // 9 = CRC-32 mismatch
}

// ---- deflate: the consensus records, 0xff00 bytes per BGZF block ---------------------------------------------------------------------
constexpr uint32_t BGZF_PAYLOAD = 0xFF00, BGZF_SLOT = 0x10000;

// a LANE per block (deflate_core.h: greedy LZ77 + a dynamic Huffman code per block; hash table and token list in the lane's slice of a
// global scratch); the block lands in its 64 KiB slot as [18-byte header][DEFLATE stream][CRC-32 (k_bgzf_crc_write)][ISIZE]
__global__ __launch_bounds__(64) void k_bgzf_deflate(const uint8_t* __restrict__ in, uint64_t len, uint32_t b0, uint32_t nb, uint8_t* __restrict__ slots,
                                                     uint32_t* __restrict__ sizes, DeflateScratch* __restrict__ scratch) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb) return;
  const uint32_t b = b0 + t;
  const uint64_t off = (uint64_t)b * BGZF_PAYLOAD;
  const uint32_t n = (uint32_t)(len - off < BGZF_PAYLOAD ? len - off : BGZF_PAYLOAD);
  const uint8_t* src = in + off;
  uint8_t* blk = slots + (size_t)b * BGZF_SLOT;
  uint32_t csize = deflate_block(src, n, blk + 18, BGZF_SLOT - 18 - 8, scratch[t]);
  if (csize == 0) {                                             // stored (a payload that does not compress into the slot)
    uint8_t* q = blk + 18;
    q[0] = 1; q[1] = (uint8_t)n; q[2] = (uint8_t)(n >> 8); q[3] = (uint8_t)~n; q[4] = (uint8_t)(~n >> 8);
    for (uint32_t i = 0; i < n; i++) q[5 + i] = src[i];
    csize = 5 + n;
  }
  const uint32_t bsize = 18 + csize + 8 - 1;
  const uint8_t hdr[18] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
  for (int i = 0; i < 18; i++) blk[i] = hdr[i];
  uint8_t* f = blk + 18 + csize + 4;
  f[0] = (uint8_t)n; f[1] = (uint8_t)(n >> 8); f[2] = 0; f[3] = 0;      // ISIZE
  sizes[b] = 18 + csize + 8;
}
// a wavefront per block: the CRC-32 of the block's uncompressed bytes into its footer
__global__ __launch_bounds__(256) void k_bgzf_crc_write(const uint8_t* __restrict__ in, uint64_t len, uint32_t nb, uint8_t* __restrict__ slots,
                                                        const uint32_t* __restrict__ sizes) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
  __syncthreads();
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= nb) return;
  const uint64_t off = (uint64_t)b * BGZF_PAYLOAD;
  const uint32_t n = (uint32_t)(len - off < BGZF_PAYLOAD ? len - off : BGZF_PAYLOAD);
  const uint32_t crc = n ? wave_crc32(in + off, n, tab, lane) : 0u;
  if (lane == 0) { uint8_t* f = slots + (size_t)b * BGZF_SLOT + sizes[b] - 8; f[0] = (uint8_t)crc; f[1] = (uint8_t)(crc >> 8); f[2] = (uint8_t)(crc >> 16); f[3] = (uint8_t)(crc >> 24); }
}
// the CRC-32 of every 0xff00-byte piece of `in` (what the HOST's deflate stage would otherwise compute: zlib's crc32 runs at ~1 GB/s
// per core, half of that stage's time; here the records are still in HBM and a wavefront per block takes a millisecond per gigabyte)
__global__ __launch_bounds__(256) void k_bgzf_crc_blocks(const uint8_t* __restrict__ in, uint64_t len, uint32_t nb, uint32_t* __restrict__ crcs) {
  __shared__ uint32_t tab[256];
  tab[threadIdx.x] = crc32_table_entry(threadIdx.x);
  __syncthreads();
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= nb) return;
  const uint64_t off = (uint64_t)b * BGZF_PAYLOAD;
  const uint32_t n = (uint32_t)(len - off < BGZF_PAYLOAD ? len - off : BGZF_PAYLOAD);
  const uint32_t crc = n ? wave_crc32(in + off, n, tab, lane) : 0u;
  if (lane == 0) crcs[b] = crc;
}
// a wavefront per block: the slot's bytes to their place in the packed stream
__global__ __launch_bounds__(256) void k_bgzf_pack(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes, const uint64_t* __restrict__ offs, uint32_t nb,
                                                   uint8_t* __restrict__ packed) {
  const uint32_t b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= nb) return;
  const uint8_t* s = slots + (size_t)b * BGZF_SLOT;
  uint8_t* d = packed + offs[b];
  const uint32_t n = sizes[b];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t i = 16 * lane;
  for (; i + 16 <= n; i += 16 * 64) { u32x4 v; __builtin_memcpy(&v, s + i, 16); __builtin_memcpy(d + i, &v, 16); }
  if (i < n) for (uint32_t k = i; k < n && k < i + 16; k++) d[k] = s[k];
}

__global__ void k_bgzf_clear_status(uint32_t* status) { *status = 0u; }

}  // namespace

// inflates `n` blocks (descriptors in device memory) from d_raw into d_out and checks every block's CRC-32: the kernels and the
// copy of the status word (into PINNED host memory, `h_status`) are queued on `s`; nothing waits.  After `s` has drained,
// bgzf_inflate_status() turns the word into 0, or 1 with c->err naming the first failing block.
bool bgzf_inflate_two_phase() { static const bool on = [] { const char* e = getenv("FGX_INFL_TWO_PHASE"); return !(e && e[0] == '0'); }(); return on; }

void bgzf_inflate_launch(hipStream_t s, const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status, uint32_t* h_status,
                         void* d_scratch, size_t scratch_bytes) {
  *h_status = 0;
  if (n == 0) return;
  if (d_scratch && scratch_bytes >= (size_t)n * 4u + 64u && bgzf_inflate_two_phase()) {
    // the two-phase form: the blocks' entry lists back to back (BgzfDevBlock::ent_off, bgzf_inflate_plan), then the list lengths [n]
    static const bool in_lds = [] { const char* e = fgx_knob("FGX_INFL_RESOLVE_LDS"); return e && e[0] == '1'; }();      // (measurements: the LDS form of the resolve pass)
    if (in_lds) {   // (the attribute belongs to a device: set per launch, only when this form is the one in use)
      hip_check(hipFuncSetAttribute((const void*)k_bgzf_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64), "hipFuncSetAttribute(k_bgzf_resolve)");
    }
    uint32_t* const ent = (uint32_t*)d_scratch;
    uint32_t* const n_ent = (uint32_t*)((uint8_t*)d_scratch + ((scratch_bytes - 64u - (size_t)n * 4u) & ~(size_t)3));
    hipLaunchKernelGGL(k_bgzf_clear_status, dim3(1), dim3(1), 0, s, d_status);
    // blocks per tokenizer wavefront: the pass is as long as ONE block's chain of symbols whatever the chip has in flight (a 64 KiB block of BAM
    // records: ~13 000 symbols, ~1.1 us each for a wavefront that has its SIMD to itself), so every wavefront should have a SIMD to itself —
    // n / 1024 blocks per wavefront, rounded up to a power of two (measured on 11 738 blocks: 8 lanes 22.0 ms, 16 18.3 ms, 32 17.6 ms)
    static const uint32_t tl_env = [] { const char* e = fgx_knob("FGX_INFL_LANES"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) ? (uint32_t)v : 0u; }();
    uint32_t tl = tl_env;
    if (!tl) { tl = 8; while (tl < 64u && (uint64_t)tl * 1024u < n) tl <<= 1; }
    if (tl == 4) hipLaunchKernelGGL(k_bgzf_tokenize<4>, dim3((n + 3) / 4), dim3(4), 0, s, d_raw, d_blk, n, d_out, ent, n_ent, d_status);
    else if (tl == 8) hipLaunchKernelGGL(k_bgzf_tokenize<8>, dim3((n + 7) / 8), dim3(8), 0, s, d_raw, d_blk, n, d_out, ent, n_ent, d_status);
    else if (tl == 32) hipLaunchKernelGGL(k_bgzf_tokenize<32>, dim3((n + 31) / 32), dim3(32), 0, s, d_raw, d_blk, n, d_out, ent, n_ent, d_status);
    else if (tl == 64) hipLaunchKernelGGL(k_bgzf_tokenize<64>, dim3((n + 63) / 64), dim3(64), 0, s, d_raw, d_blk, n, d_out, ent, n_ent, d_status);
    else hipLaunchKernelGGL(k_bgzf_tokenize<16>, dim3((n + 15) / 16), dim3(16), 0, s, d_raw, d_blk, n, d_out, ent, n_ent, d_status);
    if (in_lds) hipLaunchKernelGGL(k_bgzf_resolve, dim3(n), dim3(64), 65536 + 64, s, d_blk, n, d_out, (const uint32_t*)ent, (const uint32_t*)n_ent, d_status);
    else hipLaunchKernelGGL(k_bgzf_resolve_global, dim3(n), dim3(64), 0, s, d_blk, n, d_out, (const uint32_t*)ent, (const uint32_t*)n_ent, d_status);
    hipLaunchKernelGGL(k_bgzf_crc, dim3((n + 3) / 4), dim3(256), 0, s, (const uint8_t*)d_out, d_blk, n, d_status);
    hip_check(hipMemcpyAsync(h_status, d_status, 4, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipGetLastError(), "bgzf inflate kernels (two-phase)");
    return;
  }
  // the status word is cleared by a KERNEL on the same stream: kernels of one stream run in order, where hipMemsetAsync has been seen to run out
  // of order with the kernels around it on this runtime (boundaries.hip) — a late clear would wipe an inflate or CRC error
  hipLaunchKernelGGL(k_bgzf_clear_status, dim3(1), dim3(1), 0, s, d_status);
  static const uint32_t lanes = [] { const char* e = fgx_knob("FGX_INFL_LANES"); const int v = e ? atoi(e) : 0; return (v == 4 || v == 16 || v == 32 || v == 64) ? (uint32_t)v : INFL_LANES; }();   // (a measuring knob: profiles/r03_experiments.md)
  if (lanes == 4) hipLaunchKernelGGL(k_bgzf_inflate<4>, dim3((n + 3) / 4), dim3(4), 0, s, d_raw, d_blk, n, d_out, d_status);
  else if (lanes == 16) hipLaunchKernelGGL(k_bgzf_inflate<16>, dim3((n + 15) / 16), dim3(16), 0, s, d_raw, d_blk, n, d_out, d_status);
  else if (lanes == 32) hipLaunchKernelGGL(k_bgzf_inflate<32>, dim3((n + 31) / 32), dim3(32), 0, s, d_raw, d_blk, n, d_out, d_status);
  else if (lanes == 64) hipLaunchKernelGGL(k_bgzf_inflate<64>, dim3((n + 63) / 64), dim3(64), 0, s, d_raw, d_blk, n, d_out, d_status);
  else hipLaunchKernelGGL(k_bgzf_inflate<INFL_LANES>, dim3((n + INFL_LANES - 1) / INFL_LANES), dim3(INFL_LANES), 0, s, d_raw, d_blk, n, d_out, d_status);
  hipLaunchKernelGGL(k_bgzf_crc, dim3((n + 3) / 4), dim3(256), 0, s, (const uint8_t*)d_out, d_blk, n, d_status);
  hip_check(hipMemcpyAsync(h_status, d_status, 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipGetLastError(), "bgzf inflate kernels");
}
int bgzf_inflate_status(fgx_caller* c, uint32_t st) {
  if (!st) return 0;
  static const char* why[] = {"", "bad block type", "bad stored block", "bad code lengths", "bad symbol", "bad distance", "more output than ISIZE",
                              "input overrun", "fewer bytes than ISIZE", "CRC-32 mismatch"};
  const uint32_t code = st & 15u;
  c->err = "BGZF block " + std::to_string((st >> 4) - 1) + " of the chunk failed to inflate on the device: " + (code < 10 ? why[code] : "?");
  return 1;
}


// CRC-32 of every BGZF payload (0xff00 bytes, the last one shorter) of d_in[0 .. len) into d_crcs (device), queued on c->stream.
// d_in must be readable for 16 bytes past len.
void bgzf_crc_blocks_device(fgx_caller* c, const uint8_t* d_in, uint64_t len, uint32_t* d_crcs) {
  const uint32_t nb = (uint32_t)((len + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD);
  if (!nb) return;
  hipLaunchKernelGGL(k_bgzf_crc_blocks, dim3((nb + 3) / 4), dim3(256), 0, c->stream, d_in, len, nb, d_crcs);
  hip_check(hipGetLastError(), "k_bgzf_crc_blocks");
}

// compresses d_in[0 .. len) into BGZF blocks packed back to back in `packed` (device); `scratch` / `slots` / `meta` are working buffers
// the caller keeps.  d_in must be readable for 16 bytes past len.  Returns 0 and *packed_len.
int bgzf_deflate_device(fgx_caller* c, const uint8_t* d_in, uint64_t len, DevBuf& slots, DevBuf& scratch, DevBuf& meta, DevBuf& packed, uint64_t* packed_len) {
  *packed_len = 0;
  if (len == 0) return 0;
  hipStream_t s = c->stream;
  const uint64_t nb64 = (len + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD;
  if (nb64 > 0x7FFFFFFFull) { c->err = "bgzf_deflate_device: too many blocks"; return 1; }
  const uint32_t nb = (uint32_t)nb64;
  slots.reserve((size_t)nb * BGZF_SLOT + 64);
  meta.reserve((size_t)nb * 12 + 256);                          // sizes (u32) | offsets (u64)
  uint32_t* d_sizes = meta.as<uint32_t>();
  uint64_t* d_offs = (uint64_t*)(((uintptr_t)(d_sizes + nb) + 15) & ~(uintptr_t)15);
  meta.reserve((size_t)((uint8_t*)(d_offs + nb) - (uint8_t*)meta.p) + 64);
  d_sizes = meta.as<uint32_t>(); d_offs = (uint64_t*)(((uintptr_t)(d_sizes + nb) + 15) & ~(uintptr_t)15);
  // lanes in flight per launch: what the scratch holds (267 KB per lane)
  const uint32_t lanes = nb < 16384u ? nb : 16384u;
  scratch.reserve((size_t)lanes * sizeof(DeflateScratch));
  for (uint32_t b0 = 0; b0 < nb; b0 += lanes) {
    const uint32_t n = nb - b0 < lanes ? nb - b0 : lanes;
    hipLaunchKernelGGL(k_bgzf_deflate, dim3((n + 63) / 64), dim3(64), 0, s, d_in, len, b0, n, slots.as<uint8_t>(), d_sizes, scratch.as<DeflateScratch>());
  }
  hipLaunchKernelGGL(k_bgzf_crc_write, dim3((nb + 3) / 4), dim3(256), 0, s, d_in, len, nb, slots.as<uint8_t>(), (const uint32_t*)d_sizes);
  // offsets: exclusive scan of the block sizes (widened)
  {
    struct Widen { __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; } };
    hipcub::TransformInputIterator<uint64_t, Widen, const uint32_t*> it(d_sizes, Widen());
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, d_offs, (int)nb, s);
    c->d_tiles.reserve(tb + 64);
    hip_check(hipcub::DeviceScan::ExclusiveSum(c->d_tiles.p, tb, it, d_offs, (int)nb, s), "scan block sizes");
  }
  uint64_t last_off = 0; uint32_t last_size = 0;
  hip_check(hipMemcpyAsync(&last_off, d_offs + (nb - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(&last_size, d_sizes + (nb - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  const uint64_t total = last_off + last_size;
  packed.reserve(total + 64);
  hipLaunchKernelGGL(k_bgzf_pack, dim3((nb + 3) / 4), dim3(256), 0, s, (const uint8_t*)slots.p, (const uint32_t*)d_sizes, (const uint64_t*)d_offs, nb, packed.as<uint8_t>());
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "bgzf deflate kernels");
  *packed_len = total;
  return 0;
}

}  // namespace fgx
