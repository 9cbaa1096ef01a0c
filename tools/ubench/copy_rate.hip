// copy_rate.hip — what one MI355X sustains for the memory patterns of the record writer (k_emit) and of the family kernels' staging:
//   A  streaming copy, 16 aligned bytes per lane, grid-stride (the ceiling)
//   B  one wavefront per "family": two records of REC bytes each, read as unaligned dwords (4 B per lane and instruction) from four
//      arrays and written as unaligned dwords — k_emit's pattern without any of its arithmetic
//   C  B with 16 bytes per lane and instruction
//   D  gather of 330-byte records, 16 B per lane (the staging of a family's reads), no stores
//   hipcc --offload-arch=gfx950 -O3 copy_rate.hip -o copy_rate && ./copy_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ u32x4 ld128(const uint8_t* p) { u32x4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st128(uint8_t* p, u32x4 v) { __builtin_memcpy(p, &v, 16); }

// records of `rec` bytes at odd offsets: family f reads [f * 2 * rec + 1, ...) and writes [f * 2 * rec + 3, ...)
template <int WIDE>
__global__ __launch_bounds__(256) void k_records(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n_fam, uint32_t rec) {
  const uint32_t fam = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (fam >= n_fam) return;
  const uint8_t* s = src + (size_t)fam * 2 * rec + 1;
  uint8_t* d = dst + (size_t)fam * 2 * rec + 3;
  if (WIDE) {
    u32x4 v[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int t = 0; t < 1; t++) { const uint32_t o = min(16u * (lane + 64 * t), rec - 16); v[r][t] = ld128(s + r * rec + o); }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int t = 0; t < 1; t++) { const uint32_t o = min(16u * (lane + 64 * t), rec - 16); if (16u * (lane + 64 * t) < rec) st128(d + r * rec + o, v[r][t]); }
  } else {
    uint32_t v[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int t = 0; t < 4; t++) { const uint32_t o = min(4u * (lane + 64 * t), rec - 4); v[r][t] = ld32(s + r * rec + o); }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int t = 0; t < 4; t++) { const uint32_t o = min(4u * (lane + 64 * t), rec - 4); if (4u * (lane + 64 * t) < rec) st32(d + r * rec + o, v[r][t]); }
  }
}
// staging: one wavefront per family of 16 reads of 330 bytes: 225 of them (seq + qual) as 16-byte pieces, summed (no stores)
__global__ __launch_bounds__(256) void k_gather(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, uint32_t n_fam) {
  const uint32_t fam = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (fam >= n_fam) return;
  const uint8_t* s = src + (size_t)fam * 16 * 330;
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t c = lane + 64 * t;                  // 240 pieces: read r = c / 15, piece c % 15
    if (c < 240) { const uint32_t r = c / 15, k = c - 15 * r; acc += ld128(s + r * 330 + 60 + 16 * k); }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[fam] = acc.x;
}

int main() {
  const size_t BYTES = (size_t)10 << 30;             // 10 GiB each way
  uint8_t *src, *dst;
  CHECK(hipMalloc(&src, BYTES + 64)); CHECK(hipMalloc(&dst, BYTES + 64));
  CHECK(hipMemset(src, 1, BYTES)); CHECK(hipMemset(dst, 0, BYTES));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); launch(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 2;
    printf("%-64s %8.3f ms  %7.2f TB/s (read + written)\n", name, ms, bytes / ms / 1e9);
  };
  timeit("A streaming copy, 16 aligned B per lane", 2.0 * BYTES, [&] { hipLaunchKernelGGL(k_stream, dim3(256 * 32), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, BYTES / 16); });
  for (uint32_t rec : {925u, 1024u}) {
    const uint32_t n_fam = (uint32_t)(BYTES / (2 * rec)) - 1;
    char nm[96];
    snprintf(nm, sizeof nm, "B wave per 2 records of %u B, unaligned dwords", rec);
    timeit(nm, 2.0 * n_fam * 2.0 * rec, [&] { hipLaunchKernelGGL(k_records<0>, dim3((n_fam + 3) / 4), dim3(256), 0, 0, src, dst, n_fam, rec); });
    snprintf(nm, sizeof nm, "C wave per 2 records of %u B, unaligned 16-byte pieces", rec);
    timeit(nm, 2.0 * n_fam * 2.0 * rec, [&] { hipLaunchKernelGGL(k_records<1>, dim3((n_fam + 3) / 4), dim3(256), 0, 0, src, dst, n_fam, rec); });
  }
  {
    const uint32_t n_fam = (uint32_t)(BYTES / (16 * 330)) - 1;
    timeit("D gather: wave per family, 16 x 225 B of 330 B records, 16 B per lane (bytes = whole records)", (double)n_fam * 16 * 330,
           [&] { hipLaunchKernelGGL(k_gather, dim3((n_fam + 3) / 4), dim3(256), 0, 0, src, (uint32_t*)dst, n_fam); });
  }
  return 0;
}
