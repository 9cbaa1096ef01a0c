// reject_device.hip — the callers' `--rejects` streams on the device.  The simplex caller's (default since round 4, FGX_REJECTS_DEVICE=0 opts out; duplex / CODEC:
// round 6, strand_rejects_device at the end of the file), beside the unchanged consensus
// pipeline: every rejection of the vanilla caller is decided before the per-position arithmetic (reject_core.h has the list and the
// reference lines), so a side kernel evaluates that decision, one lane per MI group, only when rejects are asked for, and the hot kernels
// carry no per-record flags.  tests/test_reject_core.py proves the lane body against the reference restatement on the host.
//
//   k_group_bytes_max   the largest in-scope group's bytes = the size of a lane's working slab
//   k_reject_mask       lane per group (grid-stride over a bounded number of slabs): reject_core's simplex_reject_mask → mask[record],
//                       bytes[group] (block_size + record per rejected record), flag[group] (1 = rejected before the overlap
//                       pre-correction: original bytes; 2 = out of scope), totals by atomics
//   (hipcub exclusive scan of bytes[] → the groups' offsets in the stream: input order, vanilla_caller.rs:1430-1436)
//   k_reject_emit       lane per group with rejects: the group's records again (overlap-corrected copies when the pre-correction ran:
//                       simplex.rs:685-700), the rejected ones written block_size-prefixed at the group's offset
//
// Traffic: k_reject_mask reads every record once (≈ 330 B per raw read; with the overlap option it also writes and re-reads a working
// copy in the lane's slab, which stays in L2 for depth-8 families), writes 1 B per record + 9 B per group; k_reject_emit touches only
// groups that have rejects.  Scalar, divergent, latency-bound code — the point is that `--rejects` stops costing the host-orchestrated
// general path (7 M reads/s) for the whole batch.
#include "engine.h"
#include "reject_core.h"
#ifndef FGX_DEVEMU            // (tests/devemu compiles this file for the host with a serial scan)
#include <hipcub/hipcub.hpp>
#endif
#include <chrono>

namespace fgx {

constexpr uint32_t REJ_BLOCK = 64;
constexpr uint32_t REJ_MAX_LANES = 131072;               // 2 wavefronts per SIMD; 21 KB of lists per lane = 2.8 GB at most
constexpr uint64_t REJ_MAX_WORK = 8ull << 30;            // working slabs: lanes are cut back so that they fit

struct RejectBuffers {
  DevBuf mask, grp, work, slabs, out, misc, scan_tmp;
  void release() { for (DevBuf* b : {&mask, &grp, &work, &slabs, &out, &misc, &scan_tmp}) b->free_(); }
};

__global__ void k_group_bytes_max(const uint32_t* __restrict__ rec_len, const uint32_t* __restrict__ grp_first, uint32_t n_grp, unsigned long long* out_max) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_grp) return;
  const uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
  if (r1 - r0 > canon::MAX_READS) return;                // (out of scope: never touches a slab)
  unsigned long long b = 0;
  for (uint32_t r = r0; r < r1; r++) b += rec_len[r];
  atomicMax(out_max, b);
}

__global__ void __launch_bounds__(REJ_BLOCK)
k_reject_mask(rej::Params P, const uint8_t* __restrict__ blob, uint64_t blob_len, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
              const uint32_t* __restrict__ grp_first, uint32_t n_grp, uint8_t* mask, unsigned long long* grp_bytes, uint8_t* grp_flag,
              unsigned long long* totals, uint8_t* work, uint64_t slab_bytes, rej::Scratch* slabs) {
  const uint32_t lane = blockIdx.x * REJ_BLOCK + threadIdx.x, stride = gridDim.x * REJ_BLOCK;
  rej::Scratch& S = slabs[lane];
  uint8_t* w = work + (uint64_t)lane * slab_bytes;
  for (uint32_t g = lane; g < n_grp; g += stride) {
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    uint8_t whole = 0;
    const int st = rej::simplex_reject_mask(P, blob, blob_len, rec_off + r0, rec_len + r0, n, w, mask + r0, S, &whole);
    if (st != rej::REJ_OK) { grp_bytes[g] = 0; grp_flag[g] = 2; atomicAdd(&totals[1], 1ull); continue; }
    uint32_t cnt = 0;
    grp_bytes[g] = rej::reject_bytes(rec_len + r0, n, mask + r0, &cnt);
    grp_flag[g] = whole;
    if (cnt) atomicAdd(&totals[0], (unsigned long long)cnt);
  }
}

__global__ void __launch_bounds__(REJ_BLOCK)
k_reject_emit(rej::Params P, const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
              const uint32_t* __restrict__ grp_first, uint32_t n_grp, const uint8_t* __restrict__ mask, const unsigned long long* __restrict__ grp_bytes,
              const unsigned long long* __restrict__ grp_off, const uint8_t* __restrict__ grp_flag, uint8_t* out, uint8_t* work, uint64_t slab_bytes,
              rej::Scratch* slabs) {
  const uint32_t lane = blockIdx.x * REJ_BLOCK + threadIdx.x, stride = gridDim.x * REJ_BLOCK;
  rej::Scratch& S = slabs[lane];
  uint8_t* w = work + (uint64_t)lane * slab_bytes;
  for (uint32_t g = lane; g < n_grp; g += stride) {
    if (grp_bytes[g] == 0) continue;
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    rej::emit_rejects(blob, rec_off + r0, rec_len + r0, n, mask + r0, P.overlapping && !(grp_flag[g] & 1), w, S.c.ops, out + grp_off[g]);
  }
}

// ---- the duplex / CODEC callers (round 6): the same side-kernel scheme.  ONE of their decisions needs the per-position arithmetic — did the molecule give
// its consensus — and the device pipeline has just taken it: molecule g's records span [group_off[g * stride], group_off[(g + 1) * stride]) of the
// output (the last one ends at out_len), empty = no consensus.  Everything else is reject_core.h's function of the records.
template <int KIND>      // 1 duplex (codes 1..5, stream class by class), 2 CODEC (mask, input order)
__global__ void __launch_bounds__(REJ_BLOCK)
k_reject_codes_strand(rej::DuplexParams P, uint32_t has_max, const uint8_t* __restrict__ blob, uint64_t blob_len, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
                      const uint32_t* __restrict__ grp_first, uint32_t n_grp, const uint64_t* __restrict__ group_off, uint32_t stride, uint64_t out_len,
                      uint8_t* code, unsigned long long* grp_bytes, uint8_t* grp_flag, unsigned long long* totals, uint8_t* work, uint64_t slab_bytes, uint8_t* slabs) {
  const uint32_t lane = blockIdx.x * REJ_BLOCK + threadIdx.x, step = gridDim.x * REJ_BLOCK;
  uint8_t* w = work + (uint64_t)lane * slab_bytes;
  for (uint32_t g = lane; g < n_grp; g += step) {
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    const uint64_t o0 = group_off[(uint64_t)g * stride], o1 = g + 1 < n_grp ? group_off[(uint64_t)(g + 1) * stride] : out_len;
    const bool kept = o1 > o0;
    uint8_t corrected = 0;
    int st;
    if (KIND == 1) st = rej::duplex_reject_codes(P, blob, blob_len, rec_off + r0, rec_len + r0, n, kept, w, code + r0, ((rej::Scratch*)slabs)[lane], &corrected);
    else st = rej::codec_reject_mask(has_max != 0, blob, blob_len, rec_off + r0, rec_len + r0, n, kept, code + r0, ((canon::CodecScratch*)slabs)[lane]);
    if (st != rej::REJ_OK) { grp_bytes[g] = 0; grp_flag[g] = 2; atomicAdd(&totals[1], 1ull); continue; }
    uint32_t cnt = 0;
    grp_bytes[g] = rej::reject_bytes(rec_len + r0, n, code + r0, &cnt);
    grp_flag[g] = corrected;
    if (cnt) atomicAdd(&totals[0], (unsigned long long)cnt);
  }
}

template <int KIND>
__global__ void __launch_bounds__(REJ_BLOCK)
k_reject_emit_strand(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len, const uint32_t* __restrict__ grp_first, uint32_t n_grp,
                     const uint8_t* __restrict__ code, const unsigned long long* __restrict__ grp_bytes, const unsigned long long* __restrict__ grp_off,
                     const uint8_t* __restrict__ grp_flag, uint8_t* out, uint8_t* work, uint64_t slab_bytes, uint8_t* slabs) {
  const uint32_t lane = blockIdx.x * REJ_BLOCK + threadIdx.x, step = gridDim.x * REJ_BLOCK;
  uint8_t* w = work + (uint64_t)lane * slab_bytes;
  for (uint32_t g = lane; g < n_grp; g += step) {
    if (grp_bytes[g] == 0) continue;
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    if (KIND == 1) rej::emit_rejects_by_class(blob, rec_off + r0, rec_len + r0, n, code + r0, (grp_flag[g] & 1) != 0, w, ((rej::Scratch*)slabs)[lane].c.ops, out + grp_off[g]);
    else rej::emit_rejects(blob, rec_off + r0, rec_len + r0, n, code + r0, false, w, nullptr, out + grp_off[g]);
  }
}

void reject_release(fgx_caller* c) {
  if (!c->rej_state) return;
  RejectBuffers* B = (RejectBuffers*)c->rej_state;
  B->release();
  delete B;
  c->rej_state = nullptr;
}

// The rejects of the batch at d_blob / d_rec_off / d_rec_len / d_grp_first, left in device memory (r->d_out, r->bytes; r->count records).
// r->n_out_of_scope > 0: some group could not be decided here — nothing is to be used, the general path decides the batch.
void simplex_rejects_device(fgx_caller* c, const rej::Params& P, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
                            const uint32_t* d_grp_first, uint32_t n_grp, RejectResult* r) {
  r->d_out = nullptr; r->bytes = 0; r->count = 0; r->n_out_of_scope = 0; r->ms = 0;
  if (n_grp == 0) return;
  const auto t0 = std::chrono::steady_clock::now();
  if (!c->rej_state) c->rej_state = new RejectBuffers();
  RejectBuffers& B = *(RejectBuffers*)c->rej_state;
  hipStream_t s = c->stream;
  // misc: [0] rejected records, [1] groups out of scope, [2] largest in-scope group's bytes
  B.misc.reserve(64);
  unsigned long long* misc = B.misc.as<unsigned long long>();
  hip_check(hipMemsetAsync(misc, 0, 64, s), "memset reject totals");
  hipLaunchKernelGGL(k_group_bytes_max, dim3((n_grp + 255) / 256), dim3(256), 0, s, d_rec_len, d_grp_first, n_grp, misc + 2);
  unsigned long long max_bytes = 0;
  hip_check(hipMemcpyAsync(&max_bytes, misc + 2, 8, hipMemcpyDeviceToHost, s), "D2H largest group");
  hip_check(hipStreamSynchronize(s), "k_group_bytes_max");
  const uint64_t slab_bytes = (max_bytes + 15) & ~15ull;
  uint32_t blocks = (n_grp + REJ_BLOCK - 1) / REJ_BLOCK;
  if (blocks > REJ_MAX_LANES / REJ_BLOCK) blocks = REJ_MAX_LANES / REJ_BLOCK;
  if (slab_bytes && (uint64_t)blocks * REJ_BLOCK * slab_bytes > REJ_MAX_WORK) {
    blocks = (uint32_t)(REJ_MAX_WORK / (slab_bytes * REJ_BLOCK));
    if (blocks == 0) blocks = 1;
  }
  const uint32_t lanes = blocks * REJ_BLOCK;
  B.mask.reserve((size_t)n_rec + 16);
  // grp: bytes[n_grp] u64 | off[n_grp] u64 | flag[n_grp] u8
  B.grp.reserve((size_t)n_grp * 17 + 64);
  unsigned long long* grp_bytes = B.grp.as<unsigned long long>();
  unsigned long long* grp_off = grp_bytes + n_grp;
  uint8_t* grp_flag = (uint8_t*)(grp_off + n_grp);
  B.work.reserve((size_t)lanes * slab_bytes + 16);
  B.slabs.reserve((size_t)lanes * sizeof(rej::Scratch));
  hipLaunchKernelGGL(k_reject_mask, dim3(blocks), dim3(REJ_BLOCK), 0, s, P, d_blob, blob_len, d_rec_off, d_rec_len, d_grp_first, n_grp, B.mask.as<uint8_t>(), grp_bytes, grp_flag, misc,
                     B.work.as<uint8_t>(), slab_bytes, B.slabs.as<rej::Scratch>());
  hip_check(hipGetLastError(), "k_reject_mask launch");
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, grp_bytes, grp_off, (int)n_grp, s);
  B.scan_tmp.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(B.scan_tmp.p, tb, grp_bytes, grp_off, (int)n_grp, s), "scan reject bytes");
  unsigned long long h[2] = {0, 0}, last[2] = {0, 0};
  hip_check(hipMemcpyAsync(h, misc, 16, hipMemcpyDeviceToHost, s), "D2H reject totals");
  hip_check(hipMemcpyAsync(&last[0], grp_off + (n_grp - 1), 8, hipMemcpyDeviceToHost, s), "D2H last offset");
  hip_check(hipMemcpyAsync(&last[1], grp_bytes + (n_grp - 1), 8, hipMemcpyDeviceToHost, s), "D2H last size");
  hip_check(hipStreamSynchronize(s), "k_reject_mask");
  r->n_out_of_scope = (uint32_t)h[1];
  r->count = h[0];
  r->bytes = last[0] + last[1];
  if (r->n_out_of_scope == 0 && r->bytes) {
    B.out.reserve(r->bytes + 16);
    hipLaunchKernelGGL(k_reject_emit, dim3(blocks), dim3(REJ_BLOCK), 0, s, P, d_blob, d_rec_off, d_rec_len, d_grp_first, n_grp, B.mask.as<uint8_t>(), grp_bytes, grp_off, grp_flag,
                       B.out.as<uint8_t>(), B.work.as<uint8_t>(), slab_bytes, B.slabs.as<rej::Scratch>());
    hip_check(hipGetLastError(), "k_reject_emit launch");
    hip_check(hipStreamSynchronize(s), "k_reject_emit");
    r->d_out = B.out.as<uint8_t>();
  }
  r->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// The duplex (kind FGX_CALLER_DUPLEX) / CODEC caller's rejects of the batch the device pipeline has just decided: `group_off` / `stride` / `out_len` are the
// batch's group offsets in its output (fgx_caller::last_group_off).  Same result contract as simplex_rejects_device.
void strand_rejects_device(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
                           const uint32_t* d_grp_first, uint32_t n_grp, const uint64_t* d_group_off, uint32_t stride, uint64_t out_len, RejectResult* r) {
  r->d_out = nullptr; r->bytes = 0; r->count = 0; r->n_out_of_scope = 0; r->ms = 0;
  if (n_grp == 0) return;
  const bool codec = c->opt.caller_kind == FGX_CALLER_CODEC;
  const auto t0 = std::chrono::steady_clock::now();
  if (!c->rej_state) c->rej_state = new RejectBuffers();
  RejectBuffers& B = *(RejectBuffers*)c->rej_state;
  hipStream_t s = c->stream;
  rej::DuplexParams P;
  P.min_bq = c->opt.min_input_base_quality; P.overlapping = c->opt.overlapping_consensus; P.trim = c->opt.trim; P.single_strand_ok = c->opt.duplex_min_reads[2] == 0;
  const uint32_t has_max = codec && c->opt.codec_max_reads_per_strand >= 0;
  B.misc.reserve(64);
  unsigned long long* misc = B.misc.as<unsigned long long>();
  hip_check(hipMemsetAsync(misc, 0, 64, s), "memset reject totals");
  hipLaunchKernelGGL(k_group_bytes_max, dim3((n_grp + 255) / 256), dim3(256), 0, s, d_rec_len, d_grp_first, n_grp, misc + 2);
  unsigned long long max_bytes = 0;
  hip_check(hipMemcpyAsync(&max_bytes, misc + 2, 8, hipMemcpyDeviceToHost, s), "D2H largest group");
  hip_check(hipStreamSynchronize(s), "k_group_bytes_max");
  const uint64_t slab_bytes = codec ? 16 : ((max_bytes + 15) & ~15ull);              // (the CODEC command has no overlap pre-step: no working copies)
  const size_t scratch = codec ? sizeof(canon::CodecScratch) : sizeof(rej::Scratch);
  uint32_t blocks = (n_grp + REJ_BLOCK - 1) / REJ_BLOCK;
  const uint32_t max_lanes = codec ? REJ_MAX_LANES / 2 : REJ_MAX_LANES;                // (33 KB of lists per CODEC lane)
  if (blocks > max_lanes / REJ_BLOCK) blocks = max_lanes / REJ_BLOCK;
  if ((uint64_t)blocks * REJ_BLOCK * slab_bytes > REJ_MAX_WORK) {
    blocks = (uint32_t)(REJ_MAX_WORK / (slab_bytes * REJ_BLOCK));
    if (blocks == 0) blocks = 1;
  }
  const uint32_t lanes = blocks * REJ_BLOCK;
  B.mask.reserve((size_t)n_rec + 16);
  B.grp.reserve((size_t)n_grp * 17 + 64);
  unsigned long long* grp_bytes = B.grp.as<unsigned long long>();
  unsigned long long* grp_off = grp_bytes + n_grp;
  uint8_t* grp_flag = (uint8_t*)(grp_off + n_grp);
  B.work.reserve((size_t)lanes * slab_bytes + 16);
  B.slabs.reserve((size_t)lanes * scratch);
  if (codec) hipLaunchKernelGGL((k_reject_codes_strand<2>), dim3(blocks), dim3(REJ_BLOCK), 0, s, P, has_max, d_blob, blob_len, d_rec_off, d_rec_len, d_grp_first, n_grp, d_group_off, stride, out_len,
                                B.mask.as<uint8_t>(), grp_bytes, grp_flag, misc, B.work.as<uint8_t>(), slab_bytes, B.slabs.as<uint8_t>());
  else hipLaunchKernelGGL((k_reject_codes_strand<1>), dim3(blocks), dim3(REJ_BLOCK), 0, s, P, has_max, d_blob, blob_len, d_rec_off, d_rec_len, d_grp_first, n_grp, d_group_off, stride, out_len,
                          B.mask.as<uint8_t>(), grp_bytes, grp_flag, misc, B.work.as<uint8_t>(), slab_bytes, B.slabs.as<uint8_t>());
  hip_check(hipGetLastError(), "k_reject_codes_strand launch");
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, grp_bytes, grp_off, (int)n_grp, s);
  B.scan_tmp.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(B.scan_tmp.p, tb, grp_bytes, grp_off, (int)n_grp, s), "scan reject bytes");
  unsigned long long h[2] = {0, 0}, last[2] = {0, 0};
  hip_check(hipMemcpyAsync(h, misc, 16, hipMemcpyDeviceToHost, s), "D2H reject totals");
  hip_check(hipMemcpyAsync(&last[0], grp_off + (n_grp - 1), 8, hipMemcpyDeviceToHost, s), "D2H last offset");
  hip_check(hipMemcpyAsync(&last[1], grp_bytes + (n_grp - 1), 8, hipMemcpyDeviceToHost, s), "D2H last size");
  hip_check(hipStreamSynchronize(s), "k_reject_codes_strand");
  r->n_out_of_scope = (uint32_t)h[1];
  r->count = h[0];
  r->bytes = last[0] + last[1];
  if (r->n_out_of_scope == 0 && r->bytes) {
    B.out.reserve(r->bytes + 16);
    if (codec) hipLaunchKernelGGL((k_reject_emit_strand<2>), dim3(blocks), dim3(REJ_BLOCK), 0, s, d_blob, d_rec_off, d_rec_len, d_grp_first, n_grp, B.mask.as<uint8_t>(), grp_bytes, grp_off, grp_flag,
                                  B.out.as<uint8_t>(), B.work.as<uint8_t>(), slab_bytes, B.slabs.as<uint8_t>());
    else hipLaunchKernelGGL((k_reject_emit_strand<1>), dim3(blocks), dim3(REJ_BLOCK), 0, s, d_blob, d_rec_off, d_rec_len, d_grp_first, n_grp, B.mask.as<uint8_t>(), grp_bytes, grp_off, grp_flag,
                            B.out.as<uint8_t>(), B.work.as<uint8_t>(), slab_bytes, B.slabs.as<uint8_t>());
    hip_check(hipGetLastError(), "k_reject_emit_strand launch");
    hip_check(hipStreamSynchronize(s), "k_reject_emit_strand");
    r->d_out = B.out.as<uint8_t>();
  }
  r->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace fgx
