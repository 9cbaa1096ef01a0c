// devemu.cpp — TEST INFRASTRUCTURE ONLY.  Not part of the product, never loaded by it.
//
// The lane-per-item kernels of fgumi_amd/csrc/reject_device.hip and canon_device.hip, and the host code that launches them (slab sizing,
// grid-stride loops, the scan of the groups' bytes, offsets, totals), compiled for the HOST: a kernel launch becomes a serial loop over
// blocks and threads, device memory is host memory, the hipcub scan a serial sum.  These kernels use no wavefront intrinsics and no
// LDS, so the emulation runs exactly the source the GPU runs, index for index — what it cannot show is the hardware executing it (that is
// what the `-m gpu` tests are for).  There is no GPU where the CPU suite runs; this lets `-m "not gpu"` tests drive the new device
// paths end to end against the oracle and against the per-molecule host entries.
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include "../../fgumi_amd/csrc/engine.h"

// ---- the shim: what the two sources need from the HIP language and runtime ---------------------------------------------------------
#define FGX_DEVEMU 1
#undef __global__
#undef __device__
#undef __host__
#undef __launch_bounds__
#undef hipLaunchKernelGGL
#define __global__
#define __device__
#define __host__
#define __launch_bounds__(...)
namespace devemu {
struct Idx { uint32_t x = 0, y = 0, z = 0; };
static thread_local Idx g_block, g_thread, g_grid, g_bdim;
template <class F> void launch(dim3 grid, dim3 block, F&& body) {
  g_grid.x = grid.x; g_bdim.x = block.x;
  for (uint32_t b = 0; b < grid.x; b++)
    for (uint32_t t = 0; t < block.x; t++) { g_block.x = b; g_thread.x = t; body(); }
}
}  // namespace devemu
#define blockIdx devemu::g_block
#define threadIdx devemu::g_thread
#define gridDim devemu::g_grid
#define blockDim devemu::g_bdim
// (embedded in tests/apiemu, whose fake runtime may run streams as worker threads: what is queued on `stream` finishes before an
// operation of these sources — executed where it is called — starts: in-order streams)
#ifdef DEVEMU_EMBEDDED
#define DEVEMU_ORDER(stream) emu_before_op(stream)
#else
#define DEVEMU_ORDER(stream) ((void)0)
#endif
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) (DEVEMU_ORDER(stream), devemu::launch((grid), (block), [&] { kernel(__VA_ARGS__); }))
template <class T, class U> static inline T emu_atomic_add(T* p, U v) { const T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T emu_atomic_max(T* p, U v) { const T o = *p; if ((T)v > o) *p = (T)v; return o; }
#define atomicAdd emu_atomic_add
#define atomicMax emu_atomic_max
#define hipMemsetAsync(p, v, n, s) (DEVEMU_ORDER(s), memset((p), (v), (n)), hipSuccess)
#define hipMemcpyAsync(d, s, n, k, st) (DEVEMU_ORDER(st), memcpy((d), (s), (n)), hipSuccess)
#define hipStreamSynchronize(s) (DEVEMU_ORDER(s), hipSuccess)
#define hipGetLastError() (hipSuccess)
namespace hipcub {
struct DeviceScan {
  template <class In, class Out> static hipError_t ExclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n, hipStream_t st) {
    if (!tmp) { bytes = 16; return hipSuccess; }
    DEVEMU_ORDER(st);
    unsigned long long run = 0;
    for (int i = 0; i < n; i++) { const unsigned long long v = in[i]; out[i] = run; run += v; }
    return hipSuccess;
  }
  template <class In, class Out> static hipError_t InclusiveSum(void* tmp, size_t& bytes, In in, Out out, int n, hipStream_t st) {
    if (!tmp) { bytes = 16; return hipSuccess; }
    DEVEMU_ORDER(st);
    unsigned long long run = 0;
    for (int i = 0; i < n; i++) { run += in[i]; out[i] = run; }
    return hipSuccess;
  }
};
}  // namespace hipcub

#ifndef DEVEMU_EMBEDDED      // (tests/apiemu includes this file into a build that has api.cpp's own definitions on a fake HIP runtime)
namespace fgx {
void hip_check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(what); }
void DevBuf::reserve(size_t n) { if (n > cap) { free(p); p = malloc(n); if (!p) throw std::bad_alloc(); cap = n; } }
void DevBuf::free_() { free(p); p = nullptr; cap = 0; }
void PinnedBuf::reserve(size_t) {}
void PinnedBuf::free_() {}
}  // namespace fgx
double fgx_caller::run_columns(fgx::ColumnBatch&, fgx::ColParams) { return 0.0; }
#endif

#include "../../fgumi_amd/csrc/reject_device.hip"
#include "../../fgumi_amd/csrc/canon_device.hip"

using namespace fgx;

#include "../../fgumi_amd/csrc/gate_core.h"
#include "../../fgumi_amd/csrc/packed_core.h"

extern "C" {

// simplex_rejects_device over a batch held in host memory.  Returns 0; *n_oos > 0 = nothing written.
int demu_simplex_rejects(const fgx_options* o, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                         const uint32_t* grp_first, uint32_t n_grp, uint8_t* out, uint64_t cap, uint64_t* out_len, uint64_t* count, uint32_t* n_oos) {
  rej::Params P;   // (as api.cpp's reject_params)
  P.min_bq = o->min_input_base_quality; P.overlapping = o->overlapping_consensus; P.trim = o->trim; P.has_max_reads = o->max_reads >= 0;
  P.min_reads = o->min_reads; P.max_reads = o->max_reads < 0 ? 0u : o->max_reads > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)o->max_reads;
  fgx_caller* c = new fgx_caller();
  int rc = 0;
  try {
    RejectResult r;
    simplex_rejects_device(c, P, blob, blob_len, rec_off, rec_len, n_rec, grp_first, n_grp, &r);
    *out_len = r.bytes; *count = r.count; *n_oos = r.n_out_of_scope;
    if (r.n_out_of_scope == 0 && r.bytes) { if (r.bytes > cap) rc = 2; else memcpy(out, r.d_out, r.bytes); }
  } catch (const std::exception&) { rc = 3; }
  reject_release(c);
  delete c;
  return rc;
}

// launch_canon_molecules over the deferred molecules def[0..nd) of a batch held in host memory; the slot layout (first, out_off) is the
// caller's, as in api.cpp's canon_second_pass.  delta5 per molecule as fgx_canon_duplex_host reports it.
int demu_canon(const fgx_options* o, int codec, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first, const uint32_t* def,
               uint32_t nd, const uint64_t* first, uint8_t* out, const uint64_t* out_off, uint32_t* out_len, int* status, uint64_t* delta5) {
  canon::Params P;   // (as api.cpp's canon_params / canon_codec_params)
  P.min_bq = o->min_input_base_quality; P.overlapping = o->overlapping_consensus; P.trim = o->trim; P._pad = 0;
  P.min_total = o->duplex_min_reads[0]; P.min_xy = o->duplex_min_reads[1]; P.min_yx = o->duplex_min_reads[2];
  P.max_reads_per_strand = o->duplex_max_reads_per_strand;
  P.cell_tag[0] = o->cell_tag[0]; P.cell_tag[1] = o->cell_tag[1]; P._pad2[0] = P._pad2[1] = 0;
  canon::CodecParams PC;
  PC.min_reads_per_strand = o->codec_min_reads_per_strand; PC.min_duplex_length = o->codec_min_duplex_length; PC.max_reads_per_strand = o->codec_max_reads_per_strand;
  DevBuf slabs;
  std::vector<canon::Delta> delta(nd ? nd : 1);
  memset(delta.data(), 0, delta.size() * sizeof(canon::Delta));
  int rc = 0;
  try {
    launch_canon_molecules(nullptr, codec != 0, P, PC, blob, rec_off, rec_len, grp_first, def, nd, first, out, out_off, out_len, status, delta.data(), slabs);
  } catch (const std::exception&) { rc = 3; }
  for (uint32_t k = 0; k < nd; k++) { delta5[5 * k] = delta[k].minority; for (int i = 0; i < 4; i++) delta5[5 * k + 1 + i] = delta[k].ov[i]; }
  slabs.free_();
  return rc;
}

// The unanimous-column gates of the simplex kernels (gate_core.h) on the host: for `count` single-base columns — column i observes the
// n[i] qualities quals[i * stride ..] in this order — the answer of the exact gate on Kahan sums (unanimous_call_lds: k_simplex_wave2,
// k_simplex_seg, k_deep_cols), of the approximate gate on f32 sums (unanimous_call_approx: k_split_cols) and of its f32 pre-gate for the
// cap (s2_cap_pregate, with the run's member count m[i] >= n[i]): a quality, or -1 where the gate does not answer.
int demu_gates(uint8_t pre, uint8_t post, uint32_t tie, uint32_t count, const uint8_t* quals, uint32_t stride, const uint32_t* n, const uint32_t* m,
               int32_t* q_exact, int32_t* q_approx, int32_t* q_pre, float* sums_f32) {
  fgx::ConsensusTables t;
  memset(&t, 0, sizeof(t));
  fgx::build_tables(t, pre, post, tie);
  fgx::GateTables G;
  fgx::fill_gate_tables(G, t);
  fgx::CallConst K;
  K.cap = G.cap; K.cap_threshold = G.cap_threshold; K.half_cerr_at_cap = G.half_cerr_at_cap;
  static float pairf[256][2];
  fgx::fill_pairs_f32(pairf, 256, t);
  for (uint32_t i = 0; i < count; i++) {
    double w = 0.0, cw = 0.0, l = 0.0, cl = 0.0;
    float acc[2] = {0.0f, 0.0f};
    for (uint32_t j = 0; j < n[i]; j++) {
      const uint32_t qb = quals[(size_t)i * stride + j], q = qb < 93 ? qb : 93;
      { const double y = t.correct[q] - cw, s = w + y; cw = (s - w) - y; w = s; }                  // (kahan2, chain_observe.inc)
      { const double y = t.error_per_alt[q] - cl, s = l + y; cl = (s - l) - y; l = s; }
      acc[0] = __builtin_fmaf(pairf[qb][0], 1.0f, acc[0]); acc[1] = __builtin_fmaf(pairf[qb][1], 1.0f, acc[1]);   // (S2_OBSERVE: pair * sel + acc, sel = 1)
    }
    uint32_t q = 0;
    q_exact[i] = fgx::unanimous_call_lds(G, t.cerr_min, K, w, l, &q) ? (int32_t)q : -1;
    q_approx[i] = fgx::unanimous_call_approx(G, K, acc[0], acc[1], n[i], &q) ? (int32_t)q : -1;
    q_pre[i] = fgx::s2_cap_pregate(fgx::s2_pregate_consts(G, m[i]), acc[0], acc[1]) ? (int32_t)K.cap : -1;
    if (sums_f32) { sums_f32[2 * (size_t)i] = acc[0]; sums_f32[2 * (size_t)i + 1] = acc[1]; }
  }
  return 0;
}

// unanimous_cap_depth (gate_core.h): the observation count from which a single-base column with every quality >= min_bq is the cap
uint32_t demu_cap_depth(uint8_t pre, uint8_t post, uint32_t tie, uint32_t min_bq, uint32_t n_max, uint32_t* cap) {
  fgx::ConsensusTables t;
  memset(&t, 0, sizeof(t));
  fgx::build_tables(t, pre, post, tie);
  if (cap) *cap = t.cap;
  return fgx::unanimous_cap_depth(t, min_bq, n_max);
}

// The packed column pass of k_split_cols (packed_core.h) on the host, one END of a family: rows of `qs` quality bytes and `ss` sequence
// bytes (two 4-bit codes each, as BAM stores them), m rows, reads of lenE bases, cntE consensus columns, reverse or forward.  A "lane"
// per group of eight positions, exactly as run_cols_packed walks it.  Per column: code / quality / depth as the pass writes them, and
// flagged[c] = 1 where the column is left to k_call_full (or the one-observation table).
int demu_packed_end(const uint8_t* seq, const uint8_t* qual, uint32_t qs, uint32_t ss, uint32_t m, uint32_t lenE, uint32_t cntE, int rev, uint32_t min_bq, uint32_t nsafe,
                    uint32_t cap, uint32_t min_cons_bq, uint32_t min_reads, uint8_t* code, uint8_t* qual_out, uint16_t* depth, uint8_t* flagged) {
  const uint32_t groups = (lenE + 7u) >> 3;
  if (qs < 8u * groups || ss < 4u * groups || min_bq > 128u || m > 31u) return 1;   // (31 rows: the byte counters add 8 per observation)
  const uint32_t mb4 = min_bq * 0x01010101u;
  for (uint32_t k = 0; k < groups; k++) {
    uint32_t nfl, nfh;
    fgx::pk::count_masks(lenE, k, &nfl, &nfh);
    fgx::pk::Acc A;
    fgx::pk::acc_reset(A);
    for (uint32_t j = 0; j < m; j++) {
      uint32_t qx, qy, b;
      memcpy(&qx, qual + (size_t)j * qs + 8u * k, 4); memcpy(&qy, qual + (size_t)j * qs + 8u * k + 4, 4); memcpy(&b, seq + (size_t)j * ss + 4u * k, 4);
      fgx::pk::acc_row(A, qx, qy, b, mb4, nfl, nfh);
    }
    fgx::pk::Out F;
    fgx::pk::finalize(A, true, rev != 0, lenE, cntE, k, nsafe, cap, min_cons_bq, min_reads, F);
    for (int s_ = 0; s_ < 8; s_++) {
      const bool valid = ((F.valid2[s_ >> 2] >> (8 * (s_ & 3) + 7)) & 1u) != 0;
      if (valid != (s_ >= F.lo_s && s_ < F.hi_s)) return 2;                  // the slots [lo_s, hi_s) ARE the columns
      if (!valid) continue;
      const int64_t c = (int64_t)F.c_lo + s_;
      if (c < 0 || c >= (int64_t)cntE) return 3;
      code[c] = (uint8_t)(F.code2[s_ >> 2] >> (8 * (s_ & 3))); qual_out[c] = (uint8_t)(F.qual2[s_ >> 2] >> (8 * (s_ & 3)));
      depth[c] = (uint16_t)(F.dep4[s_ >> 1] >> (16 * (s_ & 1))); flagged[c] = (uint8_t)((F.flag2[s_ >> 2] >> (8 * (s_ & 3) + 7)) & 1u);
    }
  }
  return 0;
}

// S2Lds::t1 (packed_core.h fill_t1): the consensus quality of a column that holds one observation, by its quality
void demu_t1(uint8_t pre, uint8_t post, uint32_t tie, uint8_t* t1) {
  fgx::ConsensusTables t;
  memset(&t, 0, sizeof(t));
  fgx::build_tables(t, pre, post, tie);
  fgx::pk::fill_t1(t1, t);
}

// S2Image::t2 (packed_core.h fill_t2): two observations of one base, by their qualities in file order
void demu_t2(uint8_t pre, uint8_t post, uint32_t tie, uint8_t* t2) {
  fgx::ConsensusTables t;
  memset(&t, 0, sizeof(t));
  fgx::build_tables(t, pre, post, tie);
  fgx::pk::fill_t2(t2, t);
}

}  // extern "C"
