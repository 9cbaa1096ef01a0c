// bgzf_device.hip — BGZF inflate on the device: the compressed blocks of a BAM file go over PCIe (a third of the bytes), and the
// record stream is produced where the consensus kernels read it.  Stands in, on the device, for the decompress step of
// crates/fgumi-bgzf/src/reader.rs:346-479 (`decompress_block*`: libdeflater inflate + CRC32 / ISIZE check per block).
//
//   k_bgzf_inflate   a LANE per BGZF block (blocks are independent raw DEFLATE streams of at most 64 KiB), sixteen lanes per
//                    workgroup: a DEFLATE stream is sequential, so the parallelism is across blocks — a 5 GB chunk has 80 000 of them.
//                    Each lane keeps its first-level decode tables (inflate_core.h, 1.1 KB) in LDS; 8 workgroups = 128 blocks per CU in flight.
//                    The lanes of a wavefront diverge (every stream takes its own path); the kernel is bound by the latency of a
//                    lane's chain of bit-buffer refills and match copies, which is what many blocks in flight hide.
//   k_bgzf_crc       a WAVEFRONT per block: every lane runs the table-driven CRC-32 over its 1/64 of the block, and the 64 values
//                    are folded with the polynomial arithmetic of zlib's crc32_combine (slices are aligned to the END of the
//                    block, so every right-hand operand of the fold is a whole number of full slices).
#include "engine.h"
#include "inflate_core.h"

namespace fgx {

namespace {

#ifndef FGX_INFL_LANES
#define FGX_INFL_LANES 16
#endif
constexpr uint32_t INFL_LANES = FGX_INFL_LANES;

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
  const uint32_t S = (len + 63u) / 64u;                                    // bytes per lane; slices end at the block's end
  const int64_t hi_s = (int64_t)len - (int64_t)(63u - lane) * S, lo_s = hi_s - (int64_t)S;
  const uint32_t hi = hi_s > 0 ? (uint32_t)hi_s : 0u, lo = lo_s > 0 ? (uint32_t)lo_s : 0u;
  const uint8_t* p = out + B.out_off;
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
  if (lane == 0 && crc != B.crc) atomicMax(status, ((b + 1u) << 4) | 9u);   // 9 = CRC-32 mismatch
}

}  // namespace

// inflates `n` blocks (descriptors in device memory) from d_raw into d_out and checks every block's CRC-32.  Returns 0, or 1 with
// c->err naming the first failing block.
int bgzf_inflate_device(fgx_caller* c, const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status) {
  if (n == 0) return 0;
  hipStream_t s = c->stream;
  hip_check(hipMemsetAsync(d_status, 0, 4, s), "memset");
  hipLaunchKernelGGL(k_bgzf_inflate, dim3((n + INFL_LANES - 1) / INFL_LANES), dim3(INFL_LANES), 0, s, d_raw, d_blk, n, d_out, d_status);
  hipLaunchKernelGGL(k_bgzf_crc, dim3((n + 3) / 4), dim3(256), 0, s, (const uint8_t*)d_out, d_blk, n, d_status);
  uint32_t st = 0;
  hip_check(hipMemcpyAsync(&st, d_status, 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "bgzf inflate kernels");
  if (st) {
    static const char* why[] = {"", "bad block type", "bad stored block", "bad code lengths", "bad symbol", "bad distance", "more output than ISIZE",
                                "input overrun", "fewer bytes than ISIZE", "CRC-32 mismatch"};
    const uint32_t code = st & 15u;
    c->err = "BGZF block " + std::to_string((st >> 4) - 1) + " of the chunk failed to inflate on the device: " + (code < 10 ? why[code] : "?");
    return 1;
  }
  return 0;
}

}  // namespace fgx
