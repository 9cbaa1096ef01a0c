// wavemu.cpp — TEST INFRASTRUCTURE ONLY.  Not part of the product, never loaded by it.
//
// fgumi_amd/csrc/fastpath.hip — the device-resident pipeline: FastPath::run_once and EVERY wavefront kernel it launches (k_col_bound,
// k_split_parse, k_split_cols in its three builds, k_split_finish, k_simplex_seg, k_simplex_wave2, k_family_wave, k_deep_*, k_family,
// k_call_full, k_emit*, …) — compiled for the HOST under the 64-lane lock-step shim of simt.h, and linked with tests/apiemu's host side
// (api.cpp unmodified, the fake HIP runtime, the host-compiled lane-per-item kernels) in place of apiemu's stand-in for the device
// pipeline.  `FGX_LIB=libwavemu.so` then runs the product's real launch chain and real kernel sources on the CPU, lane by lane in
// lock-step: tests/test_wavemu.py compares its output with the oracle (VERDICT r5 item 2 — the tool that was missing for developing
// wavefront kernels without GPU minutes).  What it cannot show: hardware timing, occupancy, the caches' memory model.
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include "simt.h"
#include "../../fgumi_amd/csrc/engine.h"

#define FGX_WAVEMU 1
// ---- the HIP language ---------------------------------------------------------------------------------------------------------------------
#undef __global__
#undef __device__
#undef __host__
#undef __launch_bounds__
#undef __shared__
#undef __forceinline__
#undef hipLaunchKernelGGL
#undef HIP_KERNEL_NAME
#undef HIP_SYMBOL
#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __forceinline__ inline __attribute__((always_inline))
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_SYMBOL(x) (&(x))
#define blockIdx wavemu::blk().bidx
#define blockDim wavemu::blk().bdim
#define gridDim wavemu::blk().gdim
#define threadIdx wavemu::cur().tid
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) wavemu::launch((grid), (block), (size_t)(shmem), [&] { kernel(__VA_ARGS__); })
#define __syncthreads() wavemu::syncthreads(WAVEMU_SITE)
#define __syncthreads_and(p) wavemu::syncthreads_and((p), WAVEMU_SITE)
#define __syncthreads_count(p) wavemu::syncthreads_count((p), WAVEMU_SITE)
#define __ballot(p) wavemu::ballot((p), WAVEMU_SITE)
#define __any(p) (wavemu::ballot((p), WAVEMU_SITE) != 0ull)
#define __all(p) (wavemu::ballot(!(p), WAVEMU_SITE) == 0ull)
template <class T> __attribute__((always_inline)) static inline T wavemu_shfl(T v, int src, const char* site) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); u = wavemu::shfl64(u, (uint32_t)src, site); T r; memcpy(&r, &u, sizeof(T)); return r; }
#define __shfl(v, src) wavemu_shfl((v), (int)(src), WAVEMU_SITE)
#define __shfl_xor(v, m) wavemu_shfl((v), (int)(wavemu::lane_id() ^ (uint32_t)(m)), WAVEMU_SITE)
#define __shfl_down(v, d) wavemu_shfl((v), (int)(wavemu::lane_id() + (uint32_t)(d) < 64u ? wavemu::lane_id() + (uint32_t)(d) : wavemu::lane_id()), WAVEMU_SITE)
#define __shfl_up(v, d) wavemu_shfl((v), (int)(wavemu::lane_id() >= (uint32_t)(d) ? wavemu::lane_id() - (uint32_t)(d) : wavemu::lane_id()), WAVEMU_SITE)
#undef __align__
#define __align__(n) __attribute__((aligned(n)))
#define __clzll(x) __builtin_clzll(x)
#define __clz(x) __builtin_clz(x)
#define __ffsll(x) __builtin_ffsll(x)
#define __ffs(x) __builtin_ffs(x)
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
#define __popcll(x) __builtin_popcountll(x)
#define __popc(x) __builtin_popcount(x)
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define __builtin_amdgcn_readfirstlane(v) ((int)wavemu::readfirstlane((uint32_t)(v), WAVEMU_SITE))
#define __builtin_amdgcn_readlane(v, l) ((int)wavemu::readlane((uint32_t)(v), (uint32_t)(l), WAVEMU_SITE))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) ((int)wavemu::update_dpp((uint32_t)(old), (uint32_t)(src), (uint32_t)(ctrl), (uint32_t)(rm), (uint32_t)(bm), (bc), WAVEMU_SITE))
#define __builtin_amdgcn_ds_bpermute(addr, v) ((int)wavemu_shfl((uint32_t)(v), (int)(((uint32_t)(addr) >> 2) & 63u), WAVEMU_SITE))
#define __builtin_amdgcn_uicmp(a, b, cond) wavemu::uicmp((uint32_t)(a), (uint32_t)(b), (cond), WAVEMU_SITE)
#define __builtin_amdgcn_inverse_ballot_w64(m) ((((unsigned long long)(m)) >> wavemu::lane_id()) & 1ull)
#define __builtin_amdgcn_ubfe(v, off, w) wavemu::ubfe((uint32_t)(v), (uint32_t)(off), (uint32_t)(w))
#define __builtin_amdgcn_alignbyte(hi, lo, sh) wavemu::alignbyte((uint32_t)(hi), (uint32_t)(lo), (uint32_t)(sh))
#define __builtin_amdgcn_perm(a, b, sel) wavemu::perm((uint32_t)(a), (uint32_t)(b), (uint32_t)(sel))
#define __builtin_amdgcn_mbcnt_lo(m, acc) wavemu::mbcnt_lo((uint32_t)(m), (uint32_t)(acc))
#define __builtin_amdgcn_mbcnt_hi(m, acc) wavemu::mbcnt_hi((uint32_t)(m), (uint32_t)(acc))
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() wavemu::wave_barrier(WAVEMU_SITE)
typedef unsigned short wavemu_u16x2 __attribute__((ext_vector_type(2)));
static inline uint32_t wavemu_udot2(wavemu_u16x2 a, wavemu_u16x2 b, uint32_t c, bool) { return (uint32_t)a.x * b.x + (uint32_t)a.y * b.y + c; }
#define __builtin_amdgcn_udot2(a, b, c, clamp) wavemu_udot2((a), (b), (c), (clamp))
// (the packed-arithmetic helpers of packed_core.h take their host twins: __HIP_DEVICE_COMPILE__ is not defined here)
template <class T, class U> static inline T wavemu_atomic_add(T* p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T wavemu_atomic_or(T* p, U v) { const T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T wavemu_atomic_xor(T* p, U v) { const T o = *p; *p = (T)(o ^ (T)v); return o; }
template <class T, class U> static inline T wavemu_atomic_max(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T wavemu_atomic_min(T* p, U v) { const T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T wavemu_atomic_and(T* p, U v) { const T o = *p; *p = (T)(o & (T)v); return o; }
#define atomicAnd wavemu_atomic_and
#define atomicAdd wavemu_atomic_add
#define atomicOr wavemu_atomic_or
#define atomicXor wavemu_atomic_xor
#define atomicMax wavemu_atomic_max
#define atomicMin wavemu_atomic_min
static inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
// ---- the runtime calls of FastPath::run_once that are macros / templates in the HIP headers ------------------------------------------------------
#define hipMemcpyFromSymbol(dst, sym, n, ...) (memcpy((dst), (const void*)(sym), (n)), hipSuccess)
#define hipMemcpyToSymbol(sym, src, n, ...) (memcpy((void*)(sym), (src), (n)), hipSuccess)
namespace hipcub {
template <class V, class Op, class It> struct TransformInputIterator {
  It it; Op op;
  TransformInputIterator(It i, Op o) : it(i), op(o) {}
  V operator[](size_t i) const { return op(it[i]); }
  TransformInputIterator operator+(size_t k) const { return TransformInputIterator(it + k, op); }
};
struct DeviceScan {
  template <class In, class Out> static hipError_t ExclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n, hipStream_t) {
    if (!tmp) { bytes = 16; return hipSuccess; }
    unsigned long long run = 0;
    for (int i = 0; i < n; i++) { const unsigned long long v = in[i]; out[i] = run; run += v; }
    return hipSuccess;
  }
};
}  // namespace hipcub

#include "../../fgumi_amd/csrc/fastpath.hip"
