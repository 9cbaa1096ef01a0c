// canon_device.hip — the canonical form of deferred duplex / CODEC molecules computed ON THE DEVICE (opt-in, FGX_CANON_DEVICE=1 on top of
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

}  // namespace fgx
