// boundaries.hip — FindBoundaries on the device: the record offsets of an uncompressed BAM record stream resident in HBM.
//
// Mirrors   src/lib/unified_pipeline/bam.rs:193-260   BoundaryFinder::find_boundaries: walk the `block_size` chain from the first
//                                                      record; a record that does not end inside the data is left over for the
//                                                      next call.
// The chain is sequential by nature (the start of record i + 1 is known only after block_size of record i is read), and one
// thread walking 80 M records would take tens of seconds.  Here the stream is cut into SEG-byte segments:
//   k_bound_guess   a thread per segment proposes the first record start inside its segment: the first offset at which the bytes
//                   LOOK like a record (checks every valid BAM record passes: block_size covers the fixed part, the name, the
//                   CIGAR, SEQ and QUAL; the name ends in NUL; reference ids and positions >= -1) and are followed by two more
//                   that do.  A guess is only a guess.
//   k_bound_walk    a thread per segment walks the chain from its start through the records that BEGIN in its segment: their
//                   count, and where the first record of a later segment starts.
//   k_bound_check   the end of segment s's walk must be the start segment t = end / SEG proposed, and the segments in between
//                   must have proposed none.  Segment 0's start is given (true), so by induction the whole table is exactly the
//                   sequential walk's.  Where a guess was wrong (or a segment holds no record start at all) the true value is
//                   written and the walk repeated; every round makes at least the first wrong segment right.
//   scan + k_bound_write   record indices from the per-segment counts; offsets and lengths written by a second walk.
// Nothing in the result depends on the heuristic: it only decides how many repair rounds there are (none, in practice).
#ifndef FGX_DEVEMU            // (tests/apiemu compiles this file for the host with a serial scan)
#include <hipcub/hipcub.hpp>
#endif
#include "engine.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace fgx {

namespace {

constexpr uint32_t SEG = 16384;                        // bytes per segment (~50 records of a 150 bp library per thread)
constexpr uint64_t NONE = ~0ull;                       // no record starts in this segment

__device__ inline uint32_t bld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// do the bytes at `p` look like [block_size][record]?  Only properties EVERY valid record has (SAM spec 4.2).
__device__ inline bool looks_like_record(const uint8_t* s, uint64_t len, uint64_t p) {
  if (p + 36 > len) return false;
  const uint32_t bs = bld32(s + p);
  if (bs < 32 || bs > (1u << 28)) return false;
  const int32_t ref_id = (int32_t)bld32(s + p + 4), pos = (int32_t)bld32(s + p + 8);
  const uint32_t w3 = bld32(s + p + 12), w4 = bld32(s + p + 16), l_seq = bld32(s + p + 20);
  const int32_t nref = (int32_t)bld32(s + p + 24), npos = (int32_t)bld32(s + p + 28);
  if (ref_id < -1 || pos < -1 || nref < -1 || npos < -1) return false;
  const uint32_t l_name = w3 & 0xFF, n_cig = w4 & 0xFFFF;
  if (l_name == 0 || l_seq > (1u << 28)) return false;
  const uint64_t need = 32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (need > bs) return false;
  const uint64_t name = p + 4 + 32, nul = name + l_name - 1;
  if (nul >= len) return true;                                  // (the record runs past the data: nothing more to look at)
  if (s[nul] != 0) return false;
  // QNAME is [!-~]+ (SAM spec 1.4: [!-?A-~]{1,254}); CIGAR operations are 0 .. 8 and, when SEQ is present, their query-consuming
  // lengths add up to l_seq (spec 4.2.x).  A frame shifted by a byte or two against a true record passes the integer-range
  // checks above surprisingly often (block_size and reference ids are small numbers padded with zero bytes); it does not pass these.
  for (uint32_t i = 0; i + 1 < l_name; i++) { const uint8_t ch = s[name + i]; if (ch < 0x21 || ch > 0x7E) return false; }
  const uint64_t cig = nul + 1;
  if (n_cig && cig + 4ull * n_cig <= len && n_cig <= 256) {
    uint64_t q_len = 0;
    for (uint32_t i = 0; i < n_cig; i++) {
      const uint32_t op = bld32(s + cig + 4 * i), ty = op & 15u;
      if (ty > 8) return false;
      if (ty == 0 || ty == 1 || ty == 4 || ty == 7 || ty == 8) q_len += op >> 4;
    }
    if (l_seq && q_len != l_seq) return false;
  }
  return true;
}

__global__ void k_bound_guess(const uint8_t* __restrict__ s, uint64_t len, uint64_t start, uint64_t n_seg, uint64_t* __restrict__ first,
                              uint8_t* __restrict__ dirty) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_seg) return;
  dirty[g] = 0;
  const uint64_t lo = g * SEG, hi = lo + SEG < len ? lo + SEG : len;
  if (g == start / SEG) { first[g] = start; return; }          // the true first record
  if (lo < start) { first[g] = NONE; return; }                 // (the BAM header)
  uint64_t found = NONE;
  for (uint64_t p = lo; p < hi; p++) {
    if (!looks_like_record(s, len, p)) continue;
    // ... and the chain from it stays inside the data for two more records (or reaches its end).  A candidate whose block_size
    // points past the end is refused: two bytes before a true start the shifted fields pass every check above with a block_size of
    // tens of megabytes, and near the end of a chunk nothing but this would stop it.  (The true last record, when it is cut off,
    // is refused as well: its predecessor's walk ends there.)
    uint64_t q = p + 4 + (uint64_t)bld32(s + p);
    bool ok = q <= len;
    for (int k = 0; k < 2 && ok && q + 36 <= len; k++) { ok = looks_like_record(s, len, q); if (ok) { q += 4 + (uint64_t)bld32(s + q); ok = q <= len; } }
    if (ok) { found = p; break; }
  }
  first[g] = found;
}

// walk from first[g] through the records that begin below the segment's end.  end[g] = where the next record begins (>= segment end),
// or, when the data runs out first, the offset just past the last COMPLETE record with the top bit set (terminal).
constexpr uint64_t TERMINAL = 1ull << 63;
__global__ void k_bound_walk(const uint8_t* __restrict__ s, uint64_t len, uint64_t n_seg, const uint64_t* __restrict__ first,
                             uint8_t* __restrict__ dirty, int only_dirty, uint64_t* __restrict__ count, uint64_t* __restrict__ end, uint32_t* __restrict__ bad) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_seg || (only_dirty && !dirty[g])) return;
  dirty[g] = 0;                                               // (walked: clean until a check says otherwise)
  uint64_t p = first[g];
  if (p == NONE) { count[g] = 0; end[g] = NONE; bad[g] = 0; return; }
  const uint64_t hi = (g + 1) * (uint64_t)SEG;
  uint64_t n = 0, e;
  uint32_t b = 0;
  for (;;) {
    if (p + 4 > len) { e = p | TERMINAL; break; }              // (also p == len: the data ends with a whole record)
    if (p >= hi) { e = p; break; }
    const uint32_t bs = bld32(s + p);
    if (p + 4 + (uint64_t)bs > len) { e = p | TERMINAL; break; }
    if (bs < 32) b = 1;                                        // not a BAM record (reported once the table is verified)
    n++;
    p += 4 + (uint64_t)bs;
  }
  count[g] = n; end[g] = e; bad[g] = b;
}

// every walked segment tells the segment its walk ended in where that one's first record is, and the segments in between that they
// have none; whatever disagrees is corrected and marked for another walk
__global__ void k_bound_check(uint64_t n_seg, uint64_t* __restrict__ first, const uint64_t* __restrict__ end, uint8_t* __restrict__ dirty,
                              uint32_t* __restrict__ n_changed, unsigned long long* __restrict__ dbg) {   // dbg (may be null): [0] count, then {g, t, old, new}
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_seg || first[g] == NONE) return;
  const uint64_t e = end[g];
  const bool term = (e & TERMINAL) != 0;
  const uint64_t t = term ? n_seg : e / SEG;                   // (terminal: no later segment holds a record start)
  uint32_t changed = 0;
  for (uint64_t u = g + 1; u < t && u < n_seg; u++) if (first[u] != NONE) { first[u] = NONE; dirty[u] = 1; changed++; }
  if (!term && t < n_seg && first[t] != e) {
    if (dbg) { const unsigned long long k = atomicAdd(&dbg[0], 1ull); if (k < 12) { dbg[1 + 4 * k] = g; dbg[2 + 4 * k] = t; dbg[3 + 4 * k] = first[t]; dbg[4 + 4 * k] = e; } }
    first[t] = e; dirty[t] = 1; changed++;
  }
  if (changed) atomicAdd(n_changed, changed);
}

__global__ void k_bound_zero(unsigned long long* __restrict__ ctr) { if (threadIdx.x < 3) ctr[threadIdx.x] = 0; }

__global__ void k_bound_write(const uint8_t* __restrict__ s, uint64_t n_seg, const uint64_t* __restrict__ first, const uint64_t* __restrict__ count,
                              const uint64_t* __restrict__ base, uint64_t cap, uint64_t* __restrict__ rec_off, uint32_t* __restrict__ rec_len) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_seg) return;
  uint64_t p = first[g];
  const uint64_t n = count[g];
  uint64_t k = base[g];
  for (uint64_t i = 0; i < n && k < cap; i++, k++) {
    const uint32_t bs = bld32(s + p);
    rec_off[k] = p + 4; rec_len[k] = bs;
    p += 4 + (uint64_t)bs;
  }
}

// the end of the data the chain covers: the terminal segment's end, or the end of the last walked segment
__global__ void k_bound_tail(uint64_t n_seg, const uint64_t* __restrict__ first, const uint64_t* __restrict__ end, const uint32_t* __restrict__ bad,
                             unsigned long long* __restrict__ out2) {   // out2[0] = consumed (max over segments), out2[1] = any bad record
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_seg || first[g] == NONE) return;
  atomicMax(&out2[0], (unsigned long long)(end[g] & ~TERMINAL));
  if (bad[g]) atomicMax(&out2[1], 1ull);
}

}  // namespace

// returns 0; 1 = malformed stream (c->err); 2 = `cap` too small (*n_rec says how many records there are)
int record_boundaries_device(fgx_caller* c, const uint8_t* d_stream, uint64_t len, uint64_t start, uint64_t* d_rec_off, uint32_t* d_rec_len,
                             uint64_t cap, uint64_t* n_rec, uint64_t* consumed) {
  hipStream_t s = c->stream;
  *n_rec = 0; *consumed = start;
  if (start >= len) return 0;
  const uint64_t n_seg = (len + SEG - 1) / SEG;
  if (n_seg > 0x7FFFFFFFull) { c->err = "fgx_record_boundaries_device: stream longer than 2^31 segments"; return 1; }
  // first | end | base | count (4 x u64), bad (u32), dirty (u8), counters
  DevBuf& A = c->d_scratch_a;
  A.reserve((size_t)n_seg * (4 * 8 + 4 + 1) + 1024);
  uint64_t* d_first = A.as<uint64_t>(); uint64_t* d_end = d_first + n_seg; uint64_t* d_base = d_end + n_seg; uint64_t* d_count = d_base + n_seg;
  uint32_t* d_bad = (uint32_t*)(d_count + n_seg);
  uint8_t* d_dirty = (uint8_t*)(d_bad + n_seg);
  unsigned long long* d_ctr = (unsigned long long*)(((uintptr_t)(d_dirty + n_seg) + 15) & ~(uintptr_t)15);   // [0] changed (u32), [1] consumed, [2] bad
  const dim3 grid((uint32_t)((n_seg + 255) / 256)), block(256);
  // (the per-round flags and counters are cleared by the kernels themselves: hipMemsetAsync between the launches did not stay in
  // order with them on this runtime — the dirty flags of a check were wiped after it had set them, and every repair took
  // thousands of rounds instead of one)
  hipLaunchKernelGGL(k_bound_guess, grid, block, 0, s, d_stream, len, start, n_seg, d_first, d_dirty);
  hipLaunchKernelGGL(k_bound_walk, grid, block, 0, s, d_stream, len, n_seg, d_first, d_dirty, 0, d_count, d_end, d_bad);
  static const bool dbg = [] { const char* e = fgx_knob("FGX_BOUND_DEBUG"); return e && e[0] == '1'; }();
  unsigned long long* d_dbg = d_ctr + 4;                       // 1 + 12 x 4 words (measurement aid: the first corrections of round 0)
  if (dbg) hip_check(hipMemsetAsync(d_dbg, 0, 8 * 64, s), "memset");
  uint32_t rounds = 0;
  for (;;) {
    hipLaunchKernelGGL(k_bound_zero, dim3(1), dim3(64), 0, s, d_ctr);
    hipLaunchKernelGGL(k_bound_check, grid, block, 0, s, n_seg, d_first, d_end, d_dirty, (uint32_t*)d_ctr, (dbg && rounds == 0) ? d_dbg : (unsigned long long*)nullptr);
    uint32_t changed = 0;
    hip_check(hipMemcpyAsync(&changed, d_ctr, 4, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipStreamSynchronize(s), "sync");
    if (!changed) break;
    if (++rounds > n_seg + 1) { c->err = "fgx_record_boundaries_device: the segment table did not settle"; return 1; }
    hipLaunchKernelGGL(k_bound_walk, grid, block, 0, s, d_stream, len, n_seg, d_first, d_dirty, 1, d_count, d_end, d_bad);
  }
  c->last_boundary_rounds = rounds;
  if (dbg) {
    unsigned long long h[64];
    hip_check(hipMemcpy(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost), "D2H");
    fprintf(stderr, "[fgx] boundaries: %llu segments, len %llu, %u repair rounds, %llu corrections in round 0\n", (unsigned long long)n_seg, (unsigned long long)len, rounds, h[0]);
    for (unsigned long long k = 0; k < h[0] && k < 12; k++)
      fprintf(stderr, "   seg %llu -> seg %llu: first %lld (0x%llx) -> %llu (0x%llx)\n", h[1 + 4 * k], h[2 + 4 * k], (long long)h[3 + 4 * k], h[3 + 4 * k], h[4 + 4 * k], h[4 + 4 * k]);
  }
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_count, d_base, (int)n_seg, s);
  c->d_tiles.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(c->d_tiles.p, tb, d_count, d_base, (int)n_seg, s), "scan counts");
  hipLaunchKernelGGL(k_bound_tail, grid, block, 0, s, n_seg, d_first, d_end, d_bad, d_ctr + 1);
  uint64_t last_base = 0, last_count = 0; unsigned long long tail[2] = {0, 0};
  hip_check(hipMemcpyAsync(&last_base, d_base + (n_seg - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(&last_count, d_count + (n_seg - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(tail, d_ctr + 1, 16, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "boundary kernels");
  if (tail[1]) { c->err = "fgx_record_boundaries_device: a record with block_size < 32"; return 1; }
  *n_rec = last_base + last_count;
  *consumed = tail[0] > start ? tail[0] : start;
  if (*n_rec > cap) return d_rec_off ? 2 : 0;                 // (cap = 0 with null arrays: count only)
  if (*n_rec && d_rec_off && d_rec_len) {
    hipLaunchKernelGGL(k_bound_write, grid, block, 0, s, d_stream, n_seg, d_first, d_count, d_base, cap, d_rec_off, d_rec_len);
    hip_check(hipStreamSynchronize(s), "sync");
    hip_check(hipGetLastError(), "k_bound_write");
  }
  return 0;
}

}  // namespace fgx
