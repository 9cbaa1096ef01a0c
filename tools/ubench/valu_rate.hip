// valu_rate.hip — issue cost of the instruction classes the consensus kernels are made of, on one MI355X (cycles per
// wave-instruction per SIMD at W waves per SIMD).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096, UNROLL = 16;

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
  uint32_t a[8]; double d[8];
  for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i + seed; d[i] = (double)(threadIdx.x + i) + 0.5; }
  uint32_t s0 = seed, s1 = seed + 1;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const int j = u & 7;
      if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));                 // independent int adds (8 chains)
      if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[0]) : "v"(a[1]));                           // one dependent chain
      if (KIND == 2) asm volatile("v_bfe_u32 %0, %1, %2, 4" : "=v"(a[j]) : "v"(a[(j + 1) & 7]), "v"(a[(j + 2) & 7]));   // VOP3 int
      if (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));                 // independent f64 adds
      if (KIND == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(d[1]));                           // dependent f64 chain
      if (KIND == 5) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");                       // scalar
      if (KIND == 6) asm volatile("v_mov_b64 %0, %1" : "=v"(d[j]) : "v"(d[(j + 1) & 7]));                     // 64-bit move
      if (KIND == 7) asm volatile("v_cmp_le_u32 vcc, %0, %1" :: "v"(a[j]), "v"(a[(j + 1) & 7]) : "vcc");      // compare to mask
      if (KIND == 8) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7])); asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc"); }   // VALU + SALU pairs
      if (KIND == 9) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7])); asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(a[(j + 1) & 7])); }   // f64 + int pairs
      if (KIND == 10) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[j]) : "v"(a[(j + 1) & 7]));
      if (KIND == 11) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s0) : "v"(a[j]));
    }
  }
  uint32_t r = s0;
  for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, int per_iter, uint32_t* d_out) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {                               // waves per SIMD
    const int blocks = 256 * wps;                              // 256 CUs x wps blocks of 4 waves
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)wps * ITERS * UNROLL * per_iter;
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
  }
  return 0;
}

int main() {
  uint32_t* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  run<0>("v_add_u32 independent", 1, d_out);
  run<1>("v_add_u32 dependent", 1, d_out);
  run<2>("v_bfe_u32 (VOP3)", 1, d_out);
  run<10>("v_lshlrev_b32", 1, d_out);
  run<3>("v_add_f64 independent", 1, d_out);
  run<4>("v_add_f64 dependent", 1, d_out);
  run<6>("v_mov_b64", 1, d_out);
  run<7>("v_cmp_le_u32 -> vcc", 1, d_out);
  run<5>("s_add_u32", 1, d_out);
  run<11>("v_readlane_b32", 1, d_out);
  run<8>("v_add_u32 + s_add_u32", 2, d_out);
  run<9>("v_add_f64 + v_add_u32", 2, d_out);
  return 0;
}
