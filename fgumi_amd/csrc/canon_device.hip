// canon_device.hip — the canonical form of deferred duplex / CODEC molecules computed ON THE DEVICE (default since round 4, FGX_CANON_DEVICE=0 opts out; on top of
// FGX_DUPLEX_CANON / FGX_CODEC_CANON): one lane per molecule runs the scalar source of canon_core.h — the very functions
// tests/test_canon_core.py and tests/test_canon_codec.py prove through the oracle on the host — over the records that the host entry
// already uploaded, and writes the canonical records into a second device blob.  The records never come back to the host: only the
// per-molecule status, the per-record lengths (4 B each) and the counted delta do, from which the host lists the canonical batch for the
// second device pass (api.cpp: canon_second_pass).
//
// Reference semantics the scalar source restates: overlapping.rs:236-336 (overlap pre-correction), raw-bam/overlap.rs:181-268 (mate clip),
// vanilla_caller.rs:1242-1296 (alignment filter), codec_caller.rs:625-1262 (virtual clip, overlap geometry).
//
// Shape of the work: a molecule is a few KB of records walked byte by byte by one lane, with its lists (canon::Scratch, 19 KB; CODEC
// 33 KB) in a per-lane slab of global memory.  That is latency-bound scalar code, not a bandwidth kernel; it is here so that the
// host's cores (16 on the boxes this was measured on) stop being the resource an indel-rich BAM waits for — tens of thousands of lanes
// in flight against 16 threads.  A grid-stride loop bounds the slabs (at most MAX_LANES of them).
#include "engine.h"
#include "canon_core.h"
#ifndef FGX_DEVEMU            // (tests/devemu compiles this file for the host with a serial scan)
#include <hipcub/hipcub.hpp>
#endif

namespace fgx {

constexpr uint32_t CANON_BLOCK = 64;          // one wavefront per workgroup: divergent scalar code gains nothing from more
constexpr uint32_t CANON_MAX_LANES = 32768;   // slabs: 32768 x 33 KB = 1.1 GB at most

// the 4-byte block_size ahead of every kept canonical record (the slot layout leaves room for it): the blob reads as a BAM record stream
__device__ inline void write_prefixes(uint8_t* out, const uint64_t* out_off, const uint32_t* out_len, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) if (out_len[i]) canon::wr32(out + out_off[i] - 4, out_len[i]);
}

__global__ void __launch_bounds__(CANON_BLOCK)
k_canon_duplex(canon::Params P, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
               const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ def, uint32_t nd, const uint64_t* __restrict__ first,
               uint8_t* out, const uint64_t* __restrict__ out_off, uint32_t* out_len, int* status, canon::Delta* delta, canon::Scratch* slabs) {
  const uint32_t lane = blockIdx.x * CANON_BLOCK + threadIdx.x, stride = gridDim.x * CANON_BLOCK;
  canon::Scratch& S = slabs[lane];
  for (uint32_t k = lane; k < nd; k += stride) {
    const uint32_t r0 = grp_first[def[k]], n = grp_first[def[k] + 1] - r0;
    canon::Delta D;
    const int st = canon::canon_duplex_molecule(P, blob, rec_off + r0, rec_len + r0, n, out, out_off + first[k], out_len + first[k], S, D);
    status[k] = st;
    delta[k] = D;
    if (st == canon::CANON_OK) write_prefixes(out, out_off + first[k], out_len + first[k], n);
  }
}

__global__ void __launch_bounds__(CANON_BLOCK)
k_canon_codec(canon::CodecParams P, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
              const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ def, uint32_t nd, const uint64_t* __restrict__ first,
              uint8_t* out, const uint64_t* __restrict__ out_off, uint32_t* out_len, int* status, canon::CodecScratch* slabs) {
  const uint32_t lane = blockIdx.x * CANON_BLOCK + threadIdx.x, stride = gridDim.x * CANON_BLOCK;
  canon::CodecScratch& S = slabs[lane];
  for (uint32_t k = lane; k < nd; k += stride) {
    const uint32_t r0 = grp_first[def[k]], n = grp_first[def[k] + 1] - r0;
    const int st = canon::canon_codec_molecule(P, blob, rec_off + r0, rec_len + r0, n, out, out_off + first[k], out_len + first[k], S);
    status[k] = st;
    if (st == canon::CANON_OK) write_prefixes(out, out_off + first[k], out_len + first[k], n);
  }
}

// Launches the canonicalisation of the `nd` deferred molecules def[0..nd) of the batch at d_blob / d_rec_off / d_rec_len / d_grp_first.
// d_first[k] = first record slot of molecule k in d_out_off / d_out_len (the caller laid the slots out); results stay on the device.
// `slabs` is grown as needed.  Returns the number of lanes launched.
uint32_t launch_canon_molecules(hipStream_t s, bool codec, const canon::Params& P, const canon::CodecParams& PC, const uint8_t* d_blob,
                                const uint64_t* d_rec_off, const uint32_t* d_rec_len, const uint32_t* d_grp_first, const uint32_t* d_def, uint32_t nd,
                                const uint64_t* d_first, uint8_t* d_out, const uint64_t* d_out_off, uint32_t* d_out_len, int* d_status,
                                canon::Delta* d_delta, DevBuf& slabs) {
  if (nd == 0) return 0;
  uint32_t blocks = (nd + CANON_BLOCK - 1) / CANON_BLOCK;
  if (blocks > CANON_MAX_LANES / CANON_BLOCK) blocks = CANON_MAX_LANES / CANON_BLOCK;
  const uint32_t lanes = blocks * CANON_BLOCK;
  slabs.reserve((size_t)lanes * (codec ? sizeof(canon::CodecScratch) : sizeof(canon::Scratch)));
  if (codec)
    hipLaunchKernelGGL(k_canon_codec, dim3(blocks), dim3(CANON_BLOCK), 0, s, PC, d_blob, d_rec_off, d_rec_len, d_grp_first, d_def, nd, d_first, d_out, d_out_off,
                       d_out_len, d_status, slabs.as<canon::CodecScratch>());
  else
    hipLaunchKernelGGL(k_canon_duplex, dim3(blocks), dim3(CANON_BLOCK), 0, s, P, d_blob, d_rec_off, d_rec_len, d_grp_first, d_def, nd, d_first, d_out, d_out_off,
                       d_out_len, d_status, d_delta, slabs.as<canon::Scratch>());
  hip_check(hipGetLastError(), "k_canon launch");
  return lanes;
}

// =====================================================================================================================================
// The canonical second pass INSIDE the device-resident entry (fgx_process_batch_device with FGX_DUPLEX_CANON / FGX_CODEC_CANON): the
// records never leave HBM and the host never sees a record length, so the slot layout, the list of canonical records and the merge of the
// two passes' outputs are made here, by small lane-per-item kernels around hipcub scans.  api.cpp (canon_resident_pass) orders the calls.
// =====================================================================================================================================
namespace {

constexpr uint32_t NO_CANON = 0xFFFFFFFFu;

// slots and bytes (4-byte block_size + record) each deferred group needs in the canonical blob; element nd of both arrays stays 0
__global__ void k_canon_count(const uint32_t* __restrict__ rec_len, const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ def, uint32_t nd,
                              unsigned long long* cnt, unsigned long long* bytes) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nd) return;
  const uint32_t r0 = grp_first[def[k]], r1 = grp_first[def[k] + 1];
  unsigned long long b = 0;
  for (uint32_t r = r0; r < r1; r++) b += 4ull + rec_len[r];
  cnt[k] = r1 - r0; bytes[k] = b;
}
// out_off of every slot: the record body starts 4 bytes into its room (first = exclusive scan of cnt, base = exclusive scan of bytes)
__global__ void k_canon_fill(const uint32_t* __restrict__ rec_len, const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ def, uint32_t nd,
                             const unsigned long long* __restrict__ first, const unsigned long long* __restrict__ base, uint64_t* out_off) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nd) return;
  const uint32_t r0 = grp_first[def[k]], n = grp_first[def[k] + 1] - r0;
  unsigned long long o = base[k];
  for (uint32_t i = 0; i < n; i++) { out_off[first[k] + i] = o + 4; o += 4ull + rec_len[r0 + i]; }
}
// canonical molecules: kept records and a 0 / 1 flag per deferred group (element nd of both stays 0)
__global__ void k_canon_kept(const int* __restrict__ status, const unsigned long long* __restrict__ first, const uint32_t* __restrict__ out_len, uint32_t nd,
                             unsigned long long* kept, unsigned long long* ok) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nd) return;
  unsigned long long c = 0;
  const bool good = status[k] == canon::CANON_OK;
  if (good) for (unsigned long long i = first[k]; i < first[k + 1]; i++) c += out_len[i] != 0;
  kept[k] = c; ok[k] = good ? 1 : 0;
}
// the canonical batch: rec_off / rec_len of the kept records, grp_first, and the deferred index of each canonical group
__global__ void k_canon_lists(const int* __restrict__ status, const unsigned long long* __restrict__ first, const uint64_t* __restrict__ out_off,
                              const uint32_t* __restrict__ out_len, uint32_t nd, const unsigned long long* __restrict__ rbase, const unsigned long long* __restrict__ gidx,
                              uint64_t* c_off, uint32_t* c_len, uint32_t* c_grp, uint32_t* c_def) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nd) return;
  if (k == nd - 1) c_grp[gidx[nd]] = (uint32_t)rbase[nd];                  // the closing boundary
  if (status[k] != canon::CANON_OK) return;
  const unsigned long long g = gidx[k];
  c_grp[g] = (uint32_t)rbase[k]; c_def[g] = k;
  unsigned long long w = rbase[k];
  for (unsigned long long i = first[k]; i < first[k + 1]; i++) if (out_len[i]) { c_off[w] = out_off[i]; c_len[w] = out_len[i]; w++; }
}
__global__ void k_res_again(const uint32_t* __restrict__ again_list, uint32_t n_again, uint8_t* again) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_again) again[again_list[i]] = 1;
}
// canonical group ci was decided by the second pass: its original group takes the second pass's records
__global__ void k_res_map(const uint32_t* __restrict__ def, const uint32_t* __restrict__ c_def, const uint8_t* __restrict__ again, uint32_t n_cg, uint32_t* g2ci, uint8_t* used) {
  const uint32_t ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= n_cg || again[ci]) return;
  const uint32_t k = c_def[ci];
  g2ci[def[k]] = ci; used[k] = 1;
}
// bytes of every group in the merged stream (a group has first-pass records or second-pass records, never both); element n_grp stays 0
__global__ void k_res_sizes(uint32_t n_grp, const uint64_t* __restrict__ off1, uint64_t len1, const uint32_t* __restrict__ g2ci, const uint64_t* __restrict__ off2, uint64_t len2,
                            uint32_t n_cg, unsigned long long* size) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_grp) return;
  const uint64_t s1 = (g + 1 < n_grp ? off1[3ull * (g + 1)] : len1) - off1[3ull * g];
  const uint32_t ci = g2ci[g];
  const uint64_t s2 = ci != NO_CANON ? (ci + 1 < n_cg ? off2[3ull * (ci + 1)] : len2) - off2[3ull * ci] : 0;
  size[g] = s1 + s2;
}
// the merged stream: a workgroup per group copies its records (lanes take bytes 64 apart: coalesced whatever the alignment of the record)
__global__ void __launch_bounds__(64)
k_res_copy(uint32_t n_grp, const uint64_t* __restrict__ off1, const uint8_t* __restrict__ out1, const uint32_t* __restrict__ g2ci, const uint64_t* __restrict__ off2,
           const uint8_t* __restrict__ out2, const unsigned long long* __restrict__ foff, uint8_t* dst) {
  for (uint32_t g = blockIdx.x; g < n_grp; g += gridDim.x) {
    const uint64_t len = foff[g + 1] - foff[g];
    if (len == 0) continue;
    const uint32_t ci = g2ci[g];
    const uint8_t* src = ci != NO_CANON ? out2 + off2[3ull * ci] : out1 + off1[3ull * g];
    uint8_t* d = dst + foff[g];
    for (uint64_t i = threadIdx.x; i < len; i += 64) d[i] = src[i];
  }
}

inline dim3 grid_for(uint32_t n) { return dim3((n + 255) / 256); }
void scan_u64(hipStream_t s, DevBuf& tmp, const unsigned long long* in, unsigned long long* out, uint32_t n) {
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)n, s);
  tmp.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, in, out, (int)n, s), "canonical pass scan");
}

}  // namespace

// Slot layout for the deferred groups d_def[0..nd) (sorted): d_first[nd+1] (slot index of each group's first record), d_out_off[n_slots].
// `work` holds 4 x (nd+1) u64 of scratch.  Returns the totals.
void canon_layout_device(hipStream_t s, const uint32_t* d_rec_len, const uint32_t* d_grp_first, const uint32_t* d_def, uint32_t nd, unsigned long long* work,
                         unsigned long long* d_first, DevBuf& out_off, DevBuf& scan_tmp, uint64_t* n_slots, uint64_t* bytes) {
  unsigned long long* cnt = work; unsigned long long* byt = work + (nd + 1); unsigned long long* base = work + 2ull * (nd + 1);
  hip_check(hipMemsetAsync(work, 0, 4ull * (nd + 1) * 8, s), "memset layout");
  hipLaunchKernelGGL(k_canon_count, grid_for(nd), dim3(256), 0, s, d_rec_len, d_grp_first, d_def, nd, cnt, byt);
  scan_u64(s, scan_tmp, cnt, d_first, nd + 1);
  scan_u64(s, scan_tmp, byt, base, nd + 1);
  unsigned long long tot[2] = {0, 0};
  hip_check(hipMemcpyAsync(&tot[0], d_first + nd, 8, hipMemcpyDeviceToHost, s), "D2H slots");
  hip_check(hipMemcpyAsync(&tot[1], base + nd, 8, hipMemcpyDeviceToHost, s), "D2H bytes");
  hip_check(hipStreamSynchronize(s), "canonical layout");
  *n_slots = tot[0]; *bytes = tot[1];
  out_off.reserve((size_t)tot[0] * 8 + 8);
  hipLaunchKernelGGL(k_canon_fill, grid_for(nd), dim3(256), 0, s, d_rec_len, d_grp_first, d_def, nd, d_first, base, out_off.as<uint64_t>());
  hip_check(hipGetLastError(), "k_canon_fill launch");
}

// The canonical batch out of the kernel's results: c_off / c_len / c_grp / c_def (grown as needed).  `work` as above.
void canon_compact_device(hipStream_t s, const int* d_status, const unsigned long long* d_first, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t nd,
                          unsigned long long* work, DevBuf& c_off, DevBuf& c_len, DevBuf& c_grp, DevBuf& c_def, DevBuf& scan_tmp, uint32_t* n_cg, uint32_t* n_cr) {
  unsigned long long* kept = work; unsigned long long* ok = work + (nd + 1); unsigned long long* rbase = work + 2ull * (nd + 1); unsigned long long* gidx = work + 3ull * (nd + 1);
  hip_check(hipMemsetAsync(work, 0, 4ull * (nd + 1) * 8, s), "memset compaction");
  hipLaunchKernelGGL(k_canon_kept, grid_for(nd), dim3(256), 0, s, d_status, d_first, d_out_len, nd, kept, ok);
  scan_u64(s, scan_tmp, kept, rbase, nd + 1);
  scan_u64(s, scan_tmp, ok, gidx, nd + 1);
  unsigned long long tot[2] = {0, 0};
  hip_check(hipMemcpyAsync(&tot[0], rbase + nd, 8, hipMemcpyDeviceToHost, s), "D2H kept records");
  hip_check(hipMemcpyAsync(&tot[1], gidx + nd, 8, hipMemcpyDeviceToHost, s), "D2H canonical groups");
  hip_check(hipStreamSynchronize(s), "canonical compaction");
  *n_cr = (uint32_t)tot[0]; *n_cg = (uint32_t)tot[1];
  c_off.reserve((size_t)tot[0] * 8 + 8); c_len.reserve((size_t)tot[0] * 4 + 4); c_grp.reserve((size_t)(tot[1] + 1) * 4); c_def.reserve((size_t)tot[1] * 4 + 4);
  hipLaunchKernelGGL(k_canon_lists, grid_for(nd), dim3(256), 0, s, d_status, d_first, d_out_off, d_out_len, nd, rbase, gidx, c_off.as<uint64_t>(), c_len.as<uint32_t>(),
                     c_grp.as<uint32_t>(), c_def.as<uint32_t>());
  hip_check(hipGetLastError(), "k_canon_lists launch");
}

// Merge of the two passes' record streams in group order.  off1 / out1 / len1: the first pass (3 slots per group, a deferred group holds
// nothing); off2 / out2 / len2: the second pass over the n_cg canonical groups; again_list: the canonical groups the second pass deferred.
// d_used[nd] = 1 for the deferred groups the second pass decided; `aux` holds (n_grp + 1) x 2 u64 + n_grp u32 + n_cg bytes.
void resident_merge_device(hipStream_t s, uint32_t n_grp, const uint64_t* off1, const uint8_t* out1, uint64_t len1, const uint32_t* d_def, uint32_t nd, const uint32_t* c_def,
                           uint32_t n_cg, const uint64_t* off2, const uint8_t* out2, uint64_t len2, const uint32_t* again_list, uint32_t n_again, DevBuf& aux, DevBuf& scan_tmp,
                           uint8_t* d_used, DevBuf& final_out, uint64_t* final_len) {
  aux.reserve(2ull * (n_grp + 1) * 8 + (size_t)n_grp * 4 + n_cg + 64);
  unsigned long long* size = aux.as<unsigned long long>(); unsigned long long* foff = size + (n_grp + 1);
  uint32_t* g2ci = (uint32_t*)(foff + (n_grp + 1)); uint8_t* again = (uint8_t*)(g2ci + n_grp);
  hip_check(hipMemsetAsync(size, 0, 2ull * (n_grp + 1) * 8, s), "memset sizes");
  hip_check(hipMemsetAsync(g2ci, 0xFF, (size_t)n_grp * 4, s), "memset map");
  hip_check(hipMemsetAsync(again, 0, n_cg + 1, s), "memset again");
  hip_check(hipMemsetAsync(d_used, 0, nd, s), "memset used");
  if (n_again) hipLaunchKernelGGL(k_res_again, grid_for(n_again), dim3(256), 0, s, again_list, n_again, again);
  if (n_cg) hipLaunchKernelGGL(k_res_map, grid_for(n_cg), dim3(256), 0, s, d_def, c_def, again, n_cg, g2ci, d_used);
  hipLaunchKernelGGL(k_res_sizes, grid_for(n_grp), dim3(256), 0, s, n_grp, off1, len1, g2ci, off2, len2, n_cg, size);
  scan_u64(s, scan_tmp, size, foff, n_grp + 1);
  unsigned long long tot = 0;
  hip_check(hipMemcpyAsync(&tot, foff + n_grp, 8, hipMemcpyDeviceToHost, s), "D2H merged length");
  hip_check(hipStreamSynchronize(s), "merge sizes");
  *final_len = tot;
  final_out.reserve((size_t)tot + 16);
  const uint32_t blocks = n_grp < (1u << 20) ? n_grp : (1u << 20);
  hipLaunchKernelGGL(k_res_copy, dim3(blocks), dim3(64), 0, s, n_grp, off1, out1, g2ci, off2, out2, foff, final_out.as<uint8_t>());
  hip_check(hipGetLastError(), "k_res_copy launch");
  hip_check(hipStreamSynchronize(s), "k_res_copy");
}

}  // namespace fgx
