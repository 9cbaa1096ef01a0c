// dispatch_rate.hip — how fast does one MI355X start workgroups of the consensus kernels' launch shape, and how many wavefronts
// does it keep resident when every wavefront lives `spin` cycles?  A family kernel with one wavefront per family starts 10^6
// short-lived workgroups per launch: if the chip starts them slower than they retire, the SIMDs run under-occupied whatever the
// kernel's instruction count is.
//   hipcc --offload-arch=gfx950 -O3 dispatch_rate.hip -o dispatch_rate && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// every wavefront waits `spin` clock ticks (s_memrealtime: 100 MHz), touches
// its LDS slice once (so the allocation is real) and leaves; with `barrier` the workgroup also meets once, as k_split_cols does.
__global__ void k_live(uint32_t* out, uint32_t spin, uint32_t lds_per_wave, int barrier, int loads, const uint32_t* __restrict__ src) {
  extern __shared__ uint8_t dyn[];
  const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t acc = 0;
  if (loads) {                                   // a chain of `loads` dependent global loads (the prologue round trips of a family kernel)
    uint32_t idx = (blockIdx.x * 4 + wv) * 64 + lane;
    for (int i = 0; i < loads; i++) { idx = src[idx & 0xFFFFFF]; acc += idx; }
  }
  if (lds_per_wave) ((volatile uint32_t*)(dyn + (size_t)wv * lds_per_wave))[lane] = threadIdx.x + acc;
  if (barrier) __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(2);
  if (lds_per_wave) acc += ((volatile uint32_t*)(dyn + (size_t)wv * lds_per_wave))[lane ^ 1];
  if (acc == 0x12345678u) out[blockIdx.x] = acc;  // (never: keeps the work alive)
}

int main() {
  uint32_t *d_out, *d_src;
  CHECK(hipMalloc(&d_out, 4 << 20));
  CHECK(hipMalloc(&d_src, 64 << 20));
  {
    uint32_t* h = (uint32_t*)malloc(64 << 20);
    uint32_t x = 12345;
    for (size_t i = 0; i < (16u << 20); i++) { x = x * 1664525u + 1013904223u; h[i] = x >> 8; }
    CHECK(hipMemcpy(d_src, h, 64 << 20, hipMemcpyHostToDevice));
    free(h);
  }
  CHECK(hipFuncSetAttribute((const void*)k_live, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  struct Shape { const char* name; uint32_t threads, lds_per_wave; int barrier; };
  const Shape shapes[] = {
    {"256 thr, 4 x 4352 B LDS, barrier (k_split_cols)", 256, 4352 + 1480, 1},
    {"256 thr, 4 x 4352 B LDS", 256, 4352 + 1480, 0},
    {"256 thr, no LDS", 256, 0, 0},
    {"128 thr, 2 x 4352 B LDS", 128, 4352 + 1480, 0},
    {"64 thr, 4352 B LDS", 64, 4352 + 1480, 0},
    {"64 thr, no LDS", 64, 0, 0},
    {"512 thr, 8 x 4352 B LDS", 512, 4352 + 1480, 0},
    {"1024 thr, 16 x 4352 B LDS", 1024, 4352 + 1480, 0},
  };
  const uint32_t n_waves = 1u << 20;             // wavefronts per launch (one per family of a 1 M-family batch)
  for (const Shape& S : shapes) {
    for (int loads : {0, 3}) {
      for (uint32_t spin : {0u, 200u, 1000u, 2000u}) {   // s_memrealtime ticks (100 MHz: 10 ns each) -> 0, 2, 10, 20 us of life
        const uint32_t wpb = S.threads / 64, blocks = n_waves / wpb;
        const size_t lds = (size_t)wpb * S.lds_per_wave;
        hipLaunchKernelGGL(k_live, dim3(blocks), dim3(S.threads), lds, 0, d_out, spin, S.lds_per_wave, S.barrier, loads, d_src);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_live, dim3(blocks), dim3(S.threads), lds, 0, d_out, spin, S.lds_per_wave, S.barrier, loads, d_src);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double life_us = spin * 0.01;
        printf("%-50s loads %d life %5.1f us: %7.3f ms per 2^20 waves = %6.1f waves/us; resident >= %.1f waves per SIMD\n", S.name, loads, life_us, ms,
               n_waves / (ms * 1e3), (n_waves / (ms * 1e3)) * life_us / 1024.0);
      }
    }
  }
  return 0;
}
