// apiemu.cpp — TEST INFRASTRUCTURE ONLY.  Not part of the product, never loaded by it.
//
// The product's WHOLE host side — api.cpp (fgx_create, fgx_process_batch with its hybrid splice, the canonical second pass, the
// `--rejects` side path, fgx_process_batch_device), the general path's orchestration, the BGZF / pipeline host code — built unmodified
// and linked, instead of against libamdhip64 and the device kernels, against
//   * a fake HIP runtime: device memory is host memory, copies are memcpy, streams and events do nothing;
//   * the lane-per-item kernels compiled for the host — reject_device.hip, canon_device.hip and, REAL sources as well, boundaries.hip
//     (FindBoundaries: segment guesses, walks, mutual check) and grouping.hip (the MI grouper) — through tests/devemu's shim (a launch = a
//     serial loop over the grid, hipcub scans = serial sums);
//   * zlib for the BGZF inflate / CRC kernels (LDS tables, 64-lane folds: no host form), so that fgx_run_bam runs end to end;
//   * kernels.hip itself (k_column_jobs, k_meth_annotate, the device libm self-test, the simulator: plain lane-per-item kernels), so the
//     general path's per-base work is done by the real kernel sources;
//   * a stand-in for the device-resident pipeline (`FastPath::run`): it DEFERS groups by a rule (a read that is not one aligned block,
//     like the real duplex / CODEC kernels; APIEMU_DEFER=mod3 defers every third group as well, =none nothing) and decides the others
//     through the product's general path on a helper caller, handing back records, per-group offsets, counters and the deferred list
//     in the layout fastpath.hip produces.
// `FGX_LIB=libapiemu.so` then lets the `-m "not gpu"` suite run the very bodies of the GPU tests of the opt-in paths (canonical second
// pass on the host and on the "device", `--rejects` side kernels through both entries) against the oracle: every line of host plumbing
// around the kernels executes on the CPU.  What it cannot show is the real kernels producing those inputs and outputs on hardware.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../fgumi_amd/csrc/bamrec.h"
#include "../../fgumi_amd/csrc/engine.h"
#include "../../fgumi_amd/csrc/fastpath.h"

// ---- fake HIP runtime ---------------------------------------------------------------------------------------------------------------
// Two modes.  Default: device memory is host memory, copies are memcpy, streams and events do nothing (everything happens where it is
// called).  APIEMU_ASYNC=1: a stream is a worker thread with a queue — asynchronous copies, memsets, the BGZF inflate stand-in and event
// records are queued and run in order per stream, with random pauses, truly beside the calling thread and beside one another; events
// complete when their record is reached; hipStreamSynchronize / hipEventSynchronize wait, hipEventQuery answers hipErrorNotReady,
// hipFree waits for every stream (it is a device-wide synchronisation), hipMemcpy is immediate (the legacy null stream does not wait for
// non-blocking streams).  A missing ordering in fgx_run_bam's host logic — a copy or a fill racing a reader it should wait for —
// then shows as a byte mismatch on the CPU (tests/test_apiemu.py::test_pipeline_under_asynchronous_streams).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
namespace emu {
static bool async_on() { static const bool on = [] { const char* e = getenv("APIEMU_ASYNC"); return e && e[0] == '1'; }(); return on; }
struct Stream {
  std::thread th; std::mutex m; std::condition_variable cv, idle; std::deque<std::function<void()>> q; bool stop = false, busy = false;
};
struct Event { std::atomic<long> gen{0}, done{0}; };
static std::mutex g_reg_m;
static std::vector<Stream*> g_streams;
static void pause_a_little() {
  static thread_local std::mt19937 rng((uint32_t)std::hash<std::thread::id>()(std::this_thread::get_id()));
  static const uint32_t unit = [] { const char* e = getenv("APIEMU_PAUSE_US"); const int v = e ? atoi(e) : 0; return (uint32_t)(v > 0 ? v : 20); }();
  const uint32_t r = rng() % 8;
  if (r < 3) std::this_thread::sleep_for(std::chrono::microseconds(unit * (1 + rng() % 40)));   // (0.02 - 0.8 ms with the default unit, three times in eight)
  else if (r == 3) std::this_thread::yield();
}
static void worker(Stream* s) {
  for (;;) {
    std::function<void()> f;
    {
      std::unique_lock<std::mutex> l(s->m);
      s->cv.wait(l, [&] { return s->stop || !s->q.empty(); });
      if (s->q.empty()) return;
      f = std::move(s->q.front()); s->q.pop_front(); s->busy = true;
    }
    pause_a_little();
    f();
    { std::lock_guard<std::mutex> l(s->m); s->busy = false; }
    s->idle.notify_all();
  }
}
static Stream* create() {
  Stream* s = new Stream();
  s->th = std::thread(worker, s);
  std::lock_guard<std::mutex> l(g_reg_m);
  g_streams.push_back(s);
  return s;
}
static void enqueue(Stream* s, std::function<void()> f) { { std::lock_guard<std::mutex> l(s->m); s->q.push_back(std::move(f)); } s->cv.notify_one(); }
static void drain(Stream* s) { std::unique_lock<std::mutex> l(s->m); s->idle.wait(l, [&] { return s->q.empty() && !s->busy; }); }
static void drain_all() {
  std::vector<Stream*> v;
  { std::lock_guard<std::mutex> l(g_reg_m); v = g_streams; }
  for (Stream* s : v) drain(s);
}
static void destroy(Stream* s) {
  drain(s);
  { std::lock_guard<std::mutex> l(s->m); s->stop = true; }
  s->cv.notify_all();
  s->th.join();
  { std::lock_guard<std::mutex> l(g_reg_m); for (size_t i = 0; i < g_streams.size(); i++) if (g_streams[i] == s) { g_streams.erase(g_streams.begin() + i); break; } }
  delete s;
}
static Stream* of(hipStream_t s) { return (Stream*)s; }
}  // namespace emu
extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (emu::async_on()) emu::drain_all(); free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
// (APIEMU_D2D_LATE=1, with APIEMU_ASYNC=1: a device-to-device hipMemcpy returns at once and the bytes move on a stream of its own — "for
// transfers from device memory to device memory no host-side synchronization is performed": whoever reads the destination without a
// device-wide synchronisation in between reads what was there before)
static emu::Stream* null_stream() { static emu::Stream* s = emu::create(); return s; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) {
  static const bool late = [] { const char* e = getenv("APIEMU_D2D_LATE"); return e && e[0] == '1'; }();
  if (late && emu::async_on() && k == hipMemcpyDeviceToDevice) { if (n) emu::enqueue(null_stream(), [d, s, n] { memmove(d, s, n); }); return hipSuccess; }
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) {
  if (emu::async_on() && st) { if (n) emu::enqueue(emu::of(st), [d, s, n] { memmove(d, s, n); }); return hipSuccess; }
  if (n) memmove(d, s, n);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t st) {
  auto body = [=] { for (size_t r = 0; r < height; r++) memmove((uint8_t*)d + r * dpitch, (const uint8_t*)s + r * spitch, width); };
  if (emu::async_on() && st) { if (width && height) emu::enqueue(emu::of(st), body); return hipSuccess; }
  body();
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) {
  if (emu::async_on() && st) { if (n) emu::enqueue(emu::of(st), [d, v, n] { memset(d, v, n); }); return hipSuccess; }
  if (n) memset(d, v, n);
  return hipSuccess;
}
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "apiemu"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = emu::async_on() ? (hipStream_t)emu::create() : nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned int, int) { *s = emu::async_on() ? (hipStream_t)emu::create() : nullptr; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { if (emu::async_on() && s) emu::destroy(emu::of(s)); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { if (emu::async_on() && s) emu::drain(emu::of(s)); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new emu::Event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete (emu::Event*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  emu::Event* ev = (emu::Event*)e;
  const long g = ++ev->gen;
  if (emu::async_on() && s) emu::enqueue(emu::of(s), [ev, g] { ev->done = g; });
  else ev->done = g;
  return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned int) {
  emu::Event* ev = (emu::Event*)e;
  const long g = ev->gen.load();
  if (emu::async_on() && s) emu::enqueue(emu::of(s), [ev, g] { while (ev->done.load() < g) std::this_thread::sleep_for(std::chrono::microseconds(20)); });
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { emu::Event* ev = (emu::Event*)e; while (ev->done.load() < ev->gen.load()) std::this_thread::sleep_for(std::chrono::microseconds(20)); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { emu::Event* ev = (emu::Event*)e; return ev->done.load() < ev->gen.load() ? hipErrorNotReady : hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { return hipEventCreate(e); }
hipError_t hipDeviceSynchronize(void) { if (emu::async_on()) emu::drain_all(); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)16 << 30; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
}

namespace fgx {

int simplex_process_general(fgx_caller*, const uint8_t*, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, uint32_t, fgx_output*);
int duplex_process_general(fgx_caller*, const uint8_t*, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, uint32_t, fgx_output*);
int codec_process_general(fgx_caller*, const uint8_t*, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, uint32_t, fgx_output*);

#ifndef APIEMU_REAL_FASTPATH      // (tests/wavemu links the REAL fastpath.hip, compiled for the host under its lock-step shim, in place of the stand-in)
// ---- the device-resident pipeline's stand-in ----------------------------------------------------------------------------------------
static std::map<FastPath*, fgx_caller*> g_helpers;

static bool one_aligned_block(const bam::Rec& v) {
  if (v.flags() & bam::F_UNMAPPED) return false;
  return v.n_cigar() == 1 && (v.cigar_op(0) & 0xF) == 0 && (v.cigar_op(0) >> 4) == v.l_seq();
}

int FastPath::run(fgx_caller* c, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp,
                  FastResult* res) {
  fgx_caller*& h = g_helpers[this];
  if (!h) {
    fgx_options o = c->opt;
    o.read_name_prefix = c->prefix.c_str(); o.read_group_id = c->rg.c_str(); o.device = c->device; o.track_rejects = 0;
    h = fgx_create(&o);
    if (!h) throw std::runtime_error(std::string("apiemu helper caller: ") + fgx_global_error());
  }
  h->opt.track_rejects = 0;
  h->genome = c->genome;                          // (methylation-aware mode, round 4: the simplex caller's batches come through here)
  last_meth_device = (c->opt.methylation_mode != FGX_METHYLATION_DISABLED && c->genome) ? n_grp : 0u;   // (what the real pipeline reports: every family on the streaming kernels)
  const char* mode_s = getenv("APIEMU_DEFER");
  const std::string mode = mode_s ? mode_s : "indel";
  std::vector<uint32_t> def;
  std::vector<uint64_t> k_off;
  std::vector<uint32_t> k_len, k_grp(1, 0), kept;
  for (uint32_t g = 0; g < n_grp; g++) {
    const uint32_t r0 = grp_first[g], r1 = grp_first[g + 1];
    bool d = r1 == r0 || r1 - r0 > 510;             // (the real pipeline: up to 255 retained reads per end since round 4 — simplex_deep.inc; 128 records before)
    if (mode != "none")
      for (uint32_t r = r0; r < r1 && !d; r++) {
        if (rec_len[r] < 32 || rec_off[r] + rec_len[r] > blob_len) { d = true; break; }
        d = !one_aligned_block(bam::Rec{blob + rec_off[r], rec_len[r]});
      }
    if (mode == "mod3" && g % 3 == 1) d = true;
    if (d) { def.push_back(g); continue; }
    for (uint32_t r = r0; r < r1; r++) { k_off.push_back(rec_off[r]); k_len.push_back(rec_len[r]); }
    k_grp.push_back((uint32_t)k_off.size());
    kept.push_back(g);
  }
  fgx_output o;
  memset(&o, 0, sizeof(o));
  h->out_data.clear(); h->grp_out_end.clear();
  if (!kept.empty()) {
    auto fn = c->opt.caller_kind == FGX_CALLER_SIMPLEX ? simplex_process_general : c->opt.caller_kind == FGX_CALLER_DUPLEX ? duplex_process_general : codec_process_general;
    const int rc = fn(h, blob, k_off.data(), k_len.data(), (uint32_t)k_off.size(), k_grp.data(), (uint32_t)kept.size(), &o);
    if (rc != 0) throw std::runtime_error("apiemu device pipeline stand-in: " + h->err);
  }
  d_out.reserve(h->out_data.size() + 16);
  if (!h->out_data.empty()) memcpy(d_out.p, h->out_data.data(), h->out_data.size());
  d_offsets.reserve((size_t)3 * n_grp * 8 + 8);
  uint64_t* off = d_offsets.as<uint64_t>();
  uint64_t pos = 0;
  size_t k = 0;
  for (uint32_t g = 0; g < n_grp; g++) {
    uint64_t end = pos;
    if (k < kept.size() && kept[k] == g) { end = h->grp_out_end[k]; k++; }
    off[3 * g] = pos; off[3 * g + 1] = end; off[3 * g + 2] = end;
    pos = end;
  }
  d_deferred.reserve(def.size() * 4 + 4);
  if (!def.empty()) memcpy(d_deferred.p, def.data(), def.size() * 4);
  memset(res, 0, sizeof(*res));
  res->d_out = d_out.as<uint8_t>(); res->out_len = h->out_data.size(); res->count = o.count;
  for (int i = 0; i < FGX_STATS_LEN; i++) res->stats[i] = o.stats[i];
  res->n_deferred = (uint32_t)def.size(); res->d_deferred = d_deferred.as<uint32_t>();
  res->d_out_off = off; res->d_slot_size = nullptr; res->n_slots = 3 * n_grp;
  return 0;
}

void FastPath::release() {
  auto it = g_helpers.find(this);
  if (it != g_helpers.end()) { fgx_destroy(it->second); g_helpers.erase(it); }
  for (DevBuf* b : {&d_out, &d_offsets, &d_deferred}) b->free_();
}
#endif   // APIEMU_REAL_FASTPATH

}  // namespace fgx

// ---- the lane-per-item kernels, compiled for the host (tests/devemu's shim; this file's runtime definitions stay as they are) -------
static inline void emu_before_op(hipStream_t s) { if (emu::async_on() && s) emu::drain(emu::of(s)); }
#define DEVEMU_EMBEDDED 1
#include "../devemu/devemu.cpp"
// the general path's kernels, FindBoundaries and the MI grouper are lane-per-item kernels (around scans) as well: the real sources, on the host
#include "../../fgumi_amd/csrc/kernels.hip"
#include "../../fgumi_amd/csrc/boundaries.hip"
#include "../../fgumi_amd/csrc/grouping.hip"

// ---- BGZF on the "device": zlib stands in for k_bgzf_inflate / k_bgzf_crc / k_bgzf_crc_blocks (LDS tables, 64-lane CRC folds) -----------
#include <zlib.h>
namespace fgx {
static void inflate_blocks(const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status, uint32_t* h_status);
bool bgzf_inflate_two_phase() { return false; }
void bgzf_inflate_launch(hipStream_t s, const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status, uint32_t* h_status, void*, size_t) {
  if (emu::async_on() && s) {      // (as the real launch: the host's status word is cleared now, everything else is queued on the stream)
    *h_status = 0;
    emu::enqueue(emu::of(s), [=] { inflate_blocks(d_raw, d_blk, n, d_out, d_status, h_status); });
    return;
  }
  inflate_blocks(d_raw, d_blk, n, d_out, d_status, h_status);
}
static void inflate_blocks(const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status, uint32_t* h_status) {
  uint32_t st = 0;
  for (uint32_t i = 0; i < n && !st; i++) {
    const BgzfDevBlock& b = d_blk[i];
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) { st = ((i + 1) << 4) | 1; break; }
    z.next_in = (Bytef*)(d_raw + b.in_off); z.avail_in = b.in_len;
    z.next_out = d_out + b.out_off; z.avail_out = b.isize;
    const int rc = inflate(&z, Z_FINISH);
    const bool ok = (rc == Z_STREAM_END) && z.total_out == b.isize;
    inflateEnd(&z);
    if (!ok) { st = ((i + 1) << 4) | 4; break; }
    if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), d_out + b.out_off, b.isize) != b.crc) st = ((i + 1) << 4) | 9;
  }
  if (n) *d_status = st;
  *h_status = st;
}
int bgzf_inflate_status(fgx_caller* c, uint32_t st) {
  if (!st) return 0;
  c->err = "BGZF block " + std::to_string((st >> 4) - 1) + " of the chunk failed to inflate on the device (apiemu stand-in), code " + std::to_string(st & 15u);
  return 1;
}
void bgzf_crc_blocks_device(fgx_caller*, const uint8_t* d_in, uint64_t len, uint32_t* d_crcs) {
  const uint64_t P = 0xff00;
  for (uint64_t b = 0, o = 0; o < len; b++, o += P) d_crcs[b] = (uint32_t)crc32(crc32(0L, Z_NULL, 0), d_in + o, (uInt)(len - o < P ? len - o : P));
}
}  // namespace fgx
