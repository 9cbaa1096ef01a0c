// api.cpp — the C ABI declared in include/fgumi_amd.h.
#include <chrono>
#include <cstdio>
#include <stdexcept>
#include <cstring>
#include <mutex>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include "engine.h"
#include <gnu/libc-version.h>
#include <cmath>
#include "fastpath.h"
#include "host_common.h"
#include "simgen.h"
#include "canon_core.h"
#include "reject_core.h"
#include "aln_tags_core.h"

namespace fgx {
// canon_device.hip
uint32_t launch_canon_molecules(hipStream_t s, bool codec, const canon::Params& P, const canon::CodecParams& PC, const uint8_t* d_blob,
                                const uint64_t* d_rec_off, const uint32_t* d_rec_len, const uint32_t* d_grp_first, const uint32_t* d_def, uint32_t nd,
                                const uint64_t* d_first, uint8_t* d_out, const uint64_t* d_out_off, uint32_t* d_out_len, int* d_status,
                                canon::Delta* d_delta, DevBuf& slabs);
void canon_layout_device(hipStream_t s, const uint32_t* d_rec_len, const uint32_t* d_grp_first, const uint32_t* d_def, uint32_t nd, unsigned long long* work,
                         unsigned long long* d_first, DevBuf& out_off, DevBuf& scan_tmp, uint64_t* n_slots, uint64_t* bytes);
void canon_compact_device(hipStream_t s, const int* d_status, const unsigned long long* d_first, const uint64_t* d_out_off, const uint32_t* d_out_len, uint32_t nd,
                          unsigned long long* work, DevBuf& c_off, DevBuf& c_len, DevBuf& c_grp, DevBuf& c_def, DevBuf& scan_tmp, uint32_t* n_cg, uint32_t* n_cr);
void resident_merge_device(hipStream_t s, uint32_t n_grp, const uint64_t* off1, const uint8_t* out1, uint64_t len1, const uint32_t* d_def, uint32_t nd, const uint32_t* c_def,
                           uint32_t n_cg, const uint64_t* off2, const uint8_t* out2, uint64_t len2, const uint32_t* again_list, uint32_t n_again, DevBuf& aux, DevBuf& scan_tmp,
                           uint8_t* d_used, DevBuf& final_out, uint64_t* final_len);
}

using namespace fgx;

static thread_local std::string g_global_err;

struct FastState { fgx::FastPath fp; fgx::PinnedBuf pin_out; fgx::FastResult last; bool has_last = false; };   // device pipeline + the pinned landing buffer of its records

namespace fgx {

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
// ---- guard bands (test infrastructure inside the product's allocator): with FGX_GUARD_BAND=<bytes> in the environment every
//      device buffer is allocated at EXACTLY the size asked for (no growth slack) between two bands of sentinel bytes, and
//      fgx_debug_check_guard_bands() verifies the bands of every live buffer — a kernel that stores outside its buffer (scratch columns, item pools,
//      descriptors, output) leaves a mark there instead of in a neighbour's memory.  Off (the default): one static read of the environment per process.
namespace {
constexpr int GUARD_SENTINEL = 0xA5;
struct GuardRegistry {
  std::mutex m;
  std::map<void*, std::pair<size_t, size_t>> live;   // user pointer -> (bytes, band bytes)
  int damaged_freed = 0;                             // buffers whose bands were found damaged when they were freed (most buffers die with their caller, before a test ends)
  long checked_freed = 0;
  std::string first_damage;
};
// the bands of one buffer: "" when intact, else a description (the caller holds the registry's mutex or owns the buffer)
std::string guard_check_one(const uint8_t* user, size_t bytes, size_t G) {
  std::vector<uint8_t> h(2 * G);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), user - G, G, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h.data() + G, user + bytes, G, hipMemcpyDeviceToHost) != hipSuccess)
    return "the bands of a device buffer of " + std::to_string(bytes) + " bytes could not be read";
  for (size_t i = 0; i < 2 * G; i++)
    if (h[i] != (uint8_t)GUARD_SENTINEL) {
      char msg[300];
      snprintf(msg, sizeof(msg), "device buffer of %zu bytes: %s band damaged, first at byte %ld %s the buffer (value 0x%02x)", bytes, i < G ? "FRONT" : "BACK",
               i < G ? (long)(G - i) : (long)(i - G), i < G ? "before" : "past the end of", (unsigned)h[i]);
      return msg;
    }
  return "";
}
GuardRegistry& guard_registry() { static GuardRegistry g; return g; }
size_t guard_band_bytes() {
  static const size_t g = [] { const char* e = getenv("FGX_GUARD_BAND"); const long v = e ? atol(e) : 0; return (size_t)((v > 0 && v <= (1 << 20)) ? ((v + 255) & ~255L) : 0); }();
  return g;
}
}  // namespace
void DevBuf::reserve(size_t n) {
  if (n <= cap) return;
  const size_t G = guard_band_bytes();
  const size_t want = G ? n : n + n / 4 + 256;
  free_();
  void* base = nullptr;
  {
    const hipError_t e = hipMalloc(&base, want + 2 * G);
    if (e != hipSuccess) { p = nullptr; throw std::runtime_error(std::string("hipMalloc of ") + std::to_string(want + 2 * G) + " bytes: " + hipGetErrorString(e)); }
  }
  if (G) {
    hip_check(hipMemset(base, GUARD_SENTINEL, G), "hipMemset (guard band)");
    hip_check(hipMemset((uint8_t*)base + G + want, GUARD_SENTINEL, G), "hipMemset (guard band)");
    hip_check(hipDeviceSynchronize(), "sync (guard band)");
    GuardRegistry& R = guard_registry();
    std::lock_guard<std::mutex> lk(R.m);
    R.live[(uint8_t*)base + G] = std::make_pair(want, G);
  }
  p = (uint8_t*)base + G;
  cap = want;
}
void DevBuf::free_() {
  if (!p) return;
  size_t G = 0;
  if (guard_band_bytes()) {
    GuardRegistry& R = guard_registry();
    std::lock_guard<std::mutex> lk(R.m);
    auto it = R.live.find(p);
    if (it != R.live.end()) {
      G = it->second.second;
      const std::string d = guard_check_one((const uint8_t*)p, it->second.first, G);      // the last look at this buffer's bands
      R.checked_freed++;
      if (!d.empty()) { if (!R.damaged_freed) R.first_damage = d + " (found when the buffer was freed)"; R.damaged_freed++; }
      R.live.erase(it);
    }
  }
  (void)hipFree((uint8_t*)p - G);
  p = nullptr; cap = 0;
}
void PinnedBuf::reserve(size_t n) {
  if (n <= cap) return;
  size_t want = n + n / 4 + 256;
  if (p) hip_check(hipHostFree(p), "hipHostFree");
  p = nullptr;
  cap = 0;
  hip_check(hipHostMalloc(&p, want, hipHostMallocDefault), "hipHostMalloc");
  cap = want;
}
void PinnedBuf::free_() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }

}  // namespace fgx

double fgx_caller::run_columns(ColumnBatch& b, ColParams prm) {
  b.ob.assign(b.n_cols, 0); b.oq.assign(b.n_cols, 0); b.od.assign(b.n_cols, 0); b.oe.assign(b.n_cols, 0);
  b.mflag.clear(); b.mu.clear(); b.mt.clear();
  if (b.jobs.empty() || b.n_cols == 0) return 0.0;
  std::vector<Tile> tiles;
  for (uint32_t j = 0; j < b.jobs.size(); j++)
    for (uint32_t p0 = 0; p0 < b.jobs[j].cons_len; p0 += 64) tiles.push_back(Tile{j, p0});
  hip_check(hipSetDevice(device), "hipSetDevice");
  d_stage.reserve(b.stage.size());
  d_reads.reserve(b.reads.size() * sizeof(ReadDesc));
  d_jobs.reserve(b.jobs.size() * sizeof(ColJob));
  d_tiles.reserve(tiles.size() * sizeof(Tile));
  d_ob.reserve(b.n_cols); d_oq.reserve(b.n_cols); d_od.reserve(2ull * b.n_cols); d_oe.reserve(2ull * b.n_cols);
  hip_check(hipMemcpyAsync(d_stage.p, b.stage.data(), b.stage.size(), hipMemcpyHostToDevice, stream), "H2D stage");
  hip_check(hipMemcpyAsync(d_reads.p, b.reads.data(), b.reads.size() * sizeof(ReadDesc), hipMemcpyHostToDevice, stream), "H2D reads");
  hip_check(hipMemcpyAsync(d_jobs.p, b.jobs.data(), b.jobs.size() * sizeof(ColJob), hipMemcpyHostToDevice, stream), "H2D jobs");
  hip_check(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice, stream), "H2D tiles");
  const bool meth = !b.mjobs.empty() && b.n_mpos > 0 && genome;
  hip_check(hipEventRecord(ev0, stream), "event");
  if (meth) {   // annotate_and_normalize of every call, on the staged reads, before they are called
    std::vector<MethTile> mtiles;
    for (uint32_t j = 0; j < b.mjobs.size(); j++)
      for (uint32_t p0 = 0; p0 < b.mjobs[j].n_pos; p0 += 64) mtiles.push_back(MethTile{j, p0});
    d_mjobs.reserve(b.mjobs.size() * sizeof(MethJob));
    d_mruns.reserve(b.mruns.size() * sizeof(MethRun) + 32);
    d_mtiles.reserve(mtiles.size() * sizeof(MethTile));
    d_mflag.reserve(b.n_mpos); d_mu.reserve(4ull * b.n_mpos); d_mt.reserve(4ull * b.n_mpos);
    hip_check(hipMemcpyAsync(d_mjobs.p, b.mjobs.data(), b.mjobs.size() * sizeof(MethJob), hipMemcpyHostToDevice, stream), "H2D meth jobs");
    if (!b.mruns.empty()) hip_check(hipMemcpyAsync(d_mruns.p, b.mruns.data(), b.mruns.size() * sizeof(MethRun), hipMemcpyHostToDevice, stream), "H2D meth runs");
    hip_check(hipMemcpyAsync(d_mtiles.p, mtiles.data(), mtiles.size() * sizeof(MethTile), hipMemcpyHostToDevice, stream), "H2D meth tiles");
    hip_check(hipStreamSynchronize(stream), "sync");   // (mtiles is a local)
    launch_meth_annotate(stream, d_stage.as<uint8_t>(), d_reads.as<ReadDesc>(), d_mjobs.as<MethJob>(), d_mruns.as<MethRun>(), d_mtiles.as<MethTile>(),
                         (uint32_t)mtiles.size(), genome->d_genome.as<uint8_t>(), d_mflag.as<uint8_t>(), d_mu.as<uint32_t>(), d_mt.as<uint32_t>());
    hip_check(hipGetLastError(), "k_meth_annotate launch");
  }
  launch_column_jobs(stream, d_stage.as<uint8_t>(), d_reads.as<ReadDesc>(), d_jobs.as<ColJob>(), d_tiles.as<Tile>(), (uint32_t)tiles.size(),
                     d_tables.as<DeviceTables>(), prm, d_ob.as<uint8_t>(), d_oq.as<uint8_t>(), d_od.as<uint16_t>(), d_oe.as<uint16_t>());
  hip_check(hipGetLastError(), "k_column_jobs launch");
  hip_check(hipEventRecord(ev1, stream), "event");
  hip_check(hipMemcpyAsync(b.ob.data(), d_ob.p, b.n_cols, hipMemcpyDeviceToHost, stream), "D2H");
  hip_check(hipMemcpyAsync(b.oq.data(), d_oq.p, b.n_cols, hipMemcpyDeviceToHost, stream), "D2H");
  hip_check(hipMemcpyAsync(b.od.data(), d_od.p, 2ull * b.n_cols, hipMemcpyDeviceToHost, stream), "D2H");
  hip_check(hipMemcpyAsync(b.oe.data(), d_oe.p, 2ull * b.n_cols, hipMemcpyDeviceToHost, stream), "D2H");
  if (meth) {
    b.mflag.assign(b.n_mpos, 0); b.mu.assign(b.n_mpos, 0); b.mt.assign(b.n_mpos, 0);
    hip_check(hipMemcpyAsync(b.mflag.data(), d_mflag.p, b.n_mpos, hipMemcpyDeviceToHost, stream), "D2H meth");
    hip_check(hipMemcpyAsync(b.mu.data(), d_mu.p, 4ull * b.n_mpos, hipMemcpyDeviceToHost, stream), "D2H meth");
    hip_check(hipMemcpyAsync(b.mt.data(), d_mt.p, 4ull * b.n_mpos, hipMemcpyDeviceToHost, stream), "D2H meth");
    if (b.want_stage_back) hip_check(hipMemcpyAsync(b.stage.data(), d_stage.p, b.stage.size(), hipMemcpyDeviceToHost, stream), "D2H normalised reads");
  }
  hip_check(hipStreamSynchronize(stream), "sync");
  float ms = 0;
  hip_check(hipEventElapsedTime(&ms, ev0, ev1), "elapsed");
  return ms;
}

extern "C" {

void fgx_options_default(fgx_options* o) {
  memset(o, 0, sizeof(*o));
  o->struct_size = sizeof(fgx_options);
  o->caller_kind = FGX_CALLER_SIMPLEX;
  o->tag[0] = 'M'; o->tag[1] = 'I';
  o->cell_tag[0] = 'C'; o->cell_tag[1] = 'B';
  o->error_rate_pre_umi = 45; o->error_rate_post_umi = 40;
  o->min_input_base_quality = 10; o->min_consensus_base_quality = 2;
  o->produce_per_base_tags = 1; o->trim = 0; o->tie_rule = FGX_TIE_FGBIO_COMPAT;
  o->overlapping_consensus = 1; o->track_rejects = 0;
  o->min_reads = 1; o->max_reads = -1;
  o->read_name_prefix = ""; o->read_group_id = "A";
  o->duplex_min_reads[0] = 1; o->duplex_min_reads[1] = 1; o->duplex_min_reads[2] = 0;
  o->duplex_max_reads_per_strand = -1;
  o->codec_min_reads_per_strand = 1; o->codec_max_reads_per_strand = -1; o->codec_min_duplex_length = 1;
  o->codec_outer_bases_length = 5; o->codec_max_duplex_disagreements = 0xFFFFFFFFu; o->codec_max_duplex_disagreement_rate = 1.0;
  o->device = -1;
}

const char* fgx_global_error(void) { return g_global_err.c_str(); }
const char* fgx_last_error(const fgx_caller* c) { return c ? c->err.c_str() : g_global_err.c_str(); }

// "Identical to the reference" means identical to a Rust build on glibc 2.35 (the FMA ifunc variants x86-64 selects): the
// transcendentals of `call_full` are glibc's (glibc_libm.h).  An integration box with another libm would silently change what the
// reference itself computes, so the port is compared with the process's own libm on a fixed sample before a caller is handed out.
// Returns an empty string when they agree bit for bit.
static std::string libm_self_check() {
  static int verdict = 0;                   // 0 unknown, 1 agree, 2 differ
  static std::string detail;
  static std::once_flag once;               // (fgx_create is called from one thread per GPU: the first calls race)
  std::call_once(once, [&]() {
    uint64_t r = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { r ^= r << 13; r ^= r >> 7; r ^= r << 17; return (double)(r >> 11) * (1.0 / 9007199254740992.0); };
    uint32_t bad = 0;
    char first[160] = {0};
    auto cmp = [&](const char* fn, double x, double mine, double theirs) {
      if (fgx_asuint64(mine) != fgx_asuint64(theirs) && !(mine != mine && theirs != theirs)) {
        if (!bad) snprintf(first, sizeof(first), "%s(%a): port %a, libm %a", fn, x, mine, theirs);
        bad++;
      }
    };
    for (int i = 0; i < 1024; i++) {
      const double u = next();
      const double xe = -745.0 + 760.0 * u;          // exp over the whole finite range of log-probabilities
      const double xs = -40.0 * u;                   // the range call_full lives in
      const double xl = std::ldexp(0.5 + u, (i % 80) - 60);
      cmp("exp", xe, g_exp(xe), std::exp(xe));
      cmp("exp", xs, g_exp(xs), std::exp(xs));
      cmp("log", xl, g_log(xl), std::log(xl));
      cmp("log1p", -u * 0.999, g_log1p(-u * 0.999), std::log1p(-u * 0.999));
      cmp("log1p", xl, g_log1p(xl), std::log1p(xl));
      cmp("expm1", xs, g_expm1(xs), std::expm1(xs));
      cmp("expm1", -u * 0.7, g_expm1(-u * 0.7), std::expm1(-u * 0.7));
    }
    if (bad) {
      detail = std::string("the bit-exact libm port (glibc 2.35, FMA variants) disagrees with this process's libm (glibc ") + gnu_get_libc_version() +
               ") on " + std::to_string(bad) + " of 7168 sample points, first: " + first +
               " — consensus qualities would no longer be those of a reference build on this box; set FGX_ALLOW_LIBM_MISMATCH=1 to run anyway";
      verdict = 2;
    } else verdict = 1;
  });
  return verdict == 2 ? detail : std::string();
}

int fgx_libm_self_check(char* msg, uint64_t msg_cap) {
  const std::string m = libm_self_check();
  if (msg && msg_cap) { snprintf(msg, (size_t)msg_cap, "%s", m.c_str()); }
  return m.empty() ? 0 : 1;
}

fgx_caller* fgx_create(const fgx_options* opts) {
  if (!opts || opts->struct_size != sizeof(fgx_options)) { g_global_err = "fgx_create: bad options struct_size (ABI mismatch)"; return nullptr; }
  fgx_caller* c = nullptr;
  try {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
      g_global_err = std::string("fgx_create: no usable HIP device (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                     "); this engine has no CPU fallback";
      return nullptr;
    }
    {
      const std::string mismatch = libm_self_check();
      const char* allow = getenv("FGX_ALLOW_LIBM_MISMATCH");
      if (!mismatch.empty() && !(allow && allow[0] == '1')) { g_global_err = "fgx_create: " + mismatch; return nullptr; }
    }
    if (opts->methylation_mode > FGX_METHYLATION_TAPS) { g_global_err = "fgx_create: unknown methylation mode"; return nullptr; }
    if (opts->methylation_mode != FGX_METHYLATION_DISABLED && opts->caller_kind == FGX_CALLER_CODEC) {
      g_global_err = "fgx_create: the CODEC caller has no methylation-aware mode (codec_caller.rs:425)"; return nullptr;
    }
    c = new fgx_caller();
    c->opt = *opts;
    c->prefix = opts->read_name_prefix ? opts->read_name_prefix : "";
    c->rg = opts->read_group_id ? opts->read_group_id : "A";
    c->opt.read_name_prefix = nullptr;
    c->opt.read_group_id = nullptr;
    if (opts->device >= 0) c->device = opts->device; else hip_check(hipGetDevice(&c->device), "hipGetDevice");
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    create_compute_stream(&c->stream);
    hip_check(hipEventCreate(&c->ev0), "hipEventCreate");
    hip_check(hipEventCreate(&c->ev1), "hipEventCreate");
    memset(&c->h_tables, 0, sizeof(c->h_tables));
    memset(&c->h_umi_tables, 0, sizeof(c->h_umi_tables));
    build_tables(c->h_tables.t, opts->error_rate_pre_umi, opts->error_rate_post_umi, opts->tie_rule);
    build_single_input_quals(c->h_tables.single_input_quals, opts->error_rate_pre_umi, opts->error_rate_post_umi);
    build_tables(c->h_umi_tables.t, 90, 90, FGX_TIE_FGBIO_COMPAT);
    build_single_input_quals(c->h_umi_tables.single_input_quals, 90, 90);
    c->d_tables.reserve(sizeof(DeviceTables));
    c->d_umi_tables.reserve(sizeof(DeviceTables));
    hip_check(hipMemcpy(c->d_tables.p, &c->h_tables, sizeof(DeviceTables), hipMemcpyHostToDevice), "tables H2D");
    hip_check(hipMemcpy(c->d_umi_tables.p, &c->h_umi_tables, sizeof(DeviceTables), hipMemcpyHostToDevice), "tables H2D");
    return c;
  } catch (const std::exception& ex) {
    g_global_err = std::string("fgx_create: ") + ex.what();
    delete c;
    return nullptr;
  }
}

void fgx_destroy(fgx_caller* c) {
  if (!c) return;
  for (fgx_caller* w : c->workers) fgx_destroy(w);
  c->workers.clear();
  (void)hipSetDevice(c->device);
  for (DevBuf* b : {&c->d_tables, &c->d_umi_tables, &c->d_stage, &c->d_reads, &c->d_jobs, &c->d_tiles, &c->d_ob, &c->d_oq, &c->d_od,
                    &c->d_oe, &c->d_scratch_a, &c->d_scratch_b, &c->d_in_blob, &c->d_in_off, &c->d_in_len, &c->d_in_grp, &c->d_mjobs, &c->d_mruns,
                    &c->d_mtiles, &c->d_mflag, &c->d_mu, &c->d_mt, &c->d_canon_blob, &c->d_canon_off, &c->d_canon_len, &c->d_canon_grp, &c->d_canon_aux, &c->d_canon_slabs,
                    &c->d_res_out1, &c->d_res_off1, &c->d_res_final, &c->d_res_aux, &c->d_res_aux2, &c->d_res_deferred, &c->d_res_scan, &c->d_res_cdef, &c->d_res_outoff})
    b->free_();
  c->genome.reset();
  reject_release(c);
  if (c->fast) { c->fast->fp.release(); c->fast->pin_out.free_(); delete c->fast; }
  if (c->filt) { c->filt->release(); delete c->filt; }
  pipeline_release(c);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// set_reference (vanilla_caller.rs:512-522, duplex_caller.rs:524-536): the genome goes to HBM once, one byte per base, contig after contig.
int fgx_set_reference(fgx_caller* c, uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens) {
  if (!c) return 1;
  c->err.clear();
  try {
    if (n_ref == 0) { c->genome.reset(); for (fgx_caller* w : c->workers) w->genome.reset(); return 0; }
    if (!seqs || !lens) { c->err = "fgx_set_reference: null sequence table"; return 1; }
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    auto g = std::make_shared<GenomeRef>();
    g->device = c->device;
    uint64_t total = 0;
    // a header contig that the FASTA does not hold (seqs[i] == NULL) is an EMPTY contig: every base of it is unknown and the annotation is
    // emitted with zero counts, as the reference does for a contig missing from its reference map (methylation.rs: `reference.get(..)` -> None)
    for (uint32_t i = 0; i < n_ref; i++) { const uint64_t L = seqs[i] ? lens[i] : 0; g->off.push_back(total); g->len.push_back(L); total += L; }
    g->d_genome.reserve(total + 64);
    for (uint32_t i = 0; i < n_ref; i++)
      if (g->len[i]) hip_check(hipMemcpy((uint8_t*)g->d_genome.p + g->off[i], seqs[i], lens[i], hipMemcpyHostToDevice), "genome H2D");
    c->genome = g;
    for (fgx_caller* w : c->workers) w->genome = g;
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// aln_tags_core.h on the host: NM / UQ / MD of ONE record regenerated against the given contigs (contig i of the BAM header = seqs[i]; NULL = the
// FASTA lacks it) — what the lanes of filter.hip's k_aln_plan / k_aln_write run.  The edited record goes to out (cap bytes), *out_len = its
// length.  Returns the aln::Status (0 regenerated, 1 tags removed, >= 2 the reference's fatal errors), or -1 when `cap` is too small.
int fgx_regenerate_alignment_tags_host(const uint8_t* rec, uint32_t len, uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens, uint8_t* out, uint32_t cap,
                                       uint32_t* out_len) {
  if (!rec || !out_len) return -1;
  std::vector<uint8_t> genome;
  std::vector<uint64_t> off(n_ref + 1, 0), ln(n_ref + 1, 0);
  for (uint32_t i = 0; i < n_ref; i++) { off[i] = genome.size(); ln[i] = (seqs && seqs[i]) ? lens[i] : 0; if (ln[i]) genome.insert(genome.end(), seqs[i], seqs[i] + ln[i]); }
  genome.push_back(0);
  aln::Plan P; aln::Geometry G{};
  aln::plan(rec, len, genome.data(), off.data(), ln.data(), n_ref, P, G);
  *out_len = P.new_len;
  if (P.status > aln::ALN_REMOVED) return P.status;
  if (P.new_len > cap || !out) return -1;
  aln::write(rec, len, genome.data(), off.data(), P, G, out);
  return P.status;
}

// ---- methylation-aware mode: the device code's per-position body and the host-side pieces, callable without a device ----------
int fgx_methylation_annotate_host(uint8_t* stage, const uint64_t* read_off, const uint32_t* read_len, uint32_t n_reads, const int64_t* runs, uint32_t n_runs,
                                  const uint8_t* contig, uint64_t contig_len, int top_strand, uint32_t n_pos, uint8_t* is_ref_c, uint32_t* unconverted,
                                  uint32_t* converted) {
  if (!stage || !is_ref_c || !unconverted || !converted) return 1;
  std::vector<MethRead> reads(n_reads);
  for (uint32_t r = 0; r < n_reads; r++) { reads[r].off = read_off[r]; reads[r].len = read_len[r]; reads[r]._pad = 0; }
  std::vector<MethRun> rr(n_runs);
  for (uint32_t r = 0; r < n_runs; r++) rr[r] = MethRun{runs[4 * r], runs[4 * r + 1], runs[4 * r + 2], runs[4 * r + 3]};
  for (uint32_t i = 0; i < n_pos; i++)     // the kernel's lanes, one after the other
    meth_annotate_position(stage, reads.data(), n_reads, rr.data(), n_runs, contig, contig_len, top_strand != 0, i, &is_ref_c[i], &unconverted[i], &converted[i]);
  return 0;
}
uint32_t fgx_methylation_runs_host(const uint32_t* simplified, uint32_t n_s, int64_t alignment_start, int is_reverse, const uint32_t* original, uint32_t n_o,
                                   int64_t* runs, uint32_t cap) {
  SimpCigar s, o;
  for (uint32_t i = 0; i < n_s; i++) s.push_back({(uint8_t)(simplified[i] & 0xF), (uint64_t)(simplified[i] >> 4)});
  for (uint32_t i = 0; i < n_o; i++) o.push_back({(uint8_t)(original[i] & 0xF), (uint64_t)(original[i] >> 4)});
  std::vector<MethRun> rr;
  meth_runs(s, alignment_start, is_reverse != 0, o, rr);
  for (uint32_t r = 0; r < rr.size() && r < cap; r++) { runs[4 * r] = rr[r].q0; runs[4 * r + 1] = rr[r].len; runs[4 * r + 2] = rr[r].ref0; runs[4 * r + 3] = rr[r].step; }
  return (uint32_t)rr.size();
}
int fgx_methylation_mm_ml_host(const uint8_t* bases, uint32_t n, const uint8_t* is_ref_c, const uint32_t* unconverted, const uint32_t* converted, int top_strand,
                               int mode, char* mm, uint32_t mm_cap, uint8_t* ml, uint32_t ml_cap) {
  std::string s;
  std::vector<uint8_t> m;
  if (!meth_build_mm_ml(bases, n, is_ref_c, unconverted, converted, top_strand != 0, mode, s, m)) return -1;
  if (s.size() + 1 > mm_cap || m.size() > ml_cap) return -3;
  memcpy(mm, s.c_str(), s.size() + 1);
  if (!m.empty()) memcpy(ml, m.data(), m.size());
  return (int)m.size();
}

// canon_core.h on the host: the canonical form of ONE duplex molecule (the records at rec_off / rec_len), written to `out` at the same
// offsets; out_len[i] = the canonical record's length, 0 = dropped by the alignment filter.  delta5 = {reads dropped
// (MinorityAlignment), the four CorrectionStats of the overlap pre-step}.  Returns 0, or 1 when the molecule is out of scope.
static canon::Params canon_params(const fgx_options* o) {
  canon::Params P;
  P.min_bq = o->min_input_base_quality; P.overlapping = o->overlapping_consensus; P.trim = o->trim; P._pad = 0;
  P.min_total = o->duplex_min_reads[0]; P.min_xy = o->duplex_min_reads[1]; P.min_yx = o->duplex_min_reads[2];
  P.max_reads_per_strand = o->duplex_max_reads_per_strand;
  P.cell_tag[0] = o->cell_tag[0]; P.cell_tag[1] = o->cell_tag[1]; P._pad2[0] = P._pad2[1] = 0;
  return P;
}
int fgx_canon_duplex_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out, uint32_t* out_len,
                          uint64_t* delta5) {
  if (!o || o->struct_size != sizeof(fgx_options) || !blob || !out || !out_len || !delta5) return 2;
  static thread_local canon::Scratch S;
  canon::Delta D;
  const int rc = canon::canon_duplex_molecule(canon_params(o), blob, rec_off, rec_len, n, out, rec_off, out_len, S, D);
  delta5[0] = D.minority; for (int i = 0; i < 4; i++) delta5[1 + i] = D.ov[i];
  return rc;
}

static canon::CodecParams canon_codec_params(const fgx_options* o) {
  canon::CodecParams P;
  P.min_reads_per_strand = o->codec_min_reads_per_strand; P.min_duplex_length = o->codec_min_duplex_length; P.max_reads_per_strand = o->codec_max_reads_per_strand;
  return P;
}
// canon_core.h on the host: the canonical form of ONE CODEC molecule (same contract; every record is kept).  0, 1 = out of scope, 2 = bad arguments.
int fgx_canon_codec_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* out, uint32_t* out_len) {
  if (!o || o->struct_size != sizeof(fgx_options) || !blob || !out || !out_len) return 2;
  static thread_local std::unique_ptr<canon::CodecScratch> S;
  if (!S) S.reset(new canon::CodecScratch());
  return canon::canon_codec_molecule(canon_codec_params(o), blob, rec_off, rec_len, n, out, rec_off, out_len, *S);
}

// `--rejects` of the simplex caller without the general path (reject_device.hip): the default since round 4, FGX_REJECTS_DEVICE=0 opts out.
static bool rejects_device_enabled() { return opt_in("FGX_REJECTS_DEVICE"); }
static rej::Params reject_params(const fgx_options* o) {
  rej::Params P;
  P.min_bq = o->min_input_base_quality; P.overlapping = o->overlapping_consensus; P.trim = o->trim; P.has_max_reads = o->max_reads >= 0;
  P.min_reads = o->min_reads; P.max_reads = o->max_reads < 0 ? 0u : o->max_reads > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)o->max_reads;
  return P;
}
// reject_core.h on the host: the `--rejects` stream of the simplex caller for a whole batch, computed from the records alone (mask pass
// per group, then the rejected records — overlap-corrected copies, or the original bytes for a group below --min-reads — each with its
// block_size, in input order).  This is what the device side kernels of reject_device.hip run, lane per group.  `out` may be NULL to size:
// *out_len receives the bytes needed.  Returns 0; 1 = some group is out of scope (nothing written); 2 = bad arguments / `cap` too small.
int fgx_simplex_rejects_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first, uint32_t n_grp,
                             uint8_t* out, uint64_t cap, uint64_t* out_len, uint64_t* n_rejects) {
  if (!o || o->struct_size != sizeof(fgx_options) || !out_len || !n_rejects || (n_grp && (!blob || !rec_off || !rec_len || !grp_first))) return 2;
  const rej::Params P = reject_params(o);
  std::unique_ptr<rej::Scratch> S(new rej::Scratch());
  std::vector<uint8_t> work, mask;
  uint64_t pos = 0, cnt = 0;
  for (uint32_t g = 0; g < n_grp; g++) {
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n; i++) bytes += rec_len[r0 + i];
    work.resize(bytes + 16); mask.assign(n + 1, 0);
    uint8_t whole = 0;
    if (rej::simplex_reject_mask(P, blob, ~0ull, rec_off + r0, rec_len + r0, n, work.data(), mask.data(), *S, &whole) != rej::REJ_OK) return 1;
    uint32_t c = 0;
    const uint64_t b = rej::reject_bytes(rec_len + r0, n, mask.data(), &c);
    if (out && b) {
      if (pos + b > cap) return 2;
      rej::emit_rejects(blob, rec_off + r0, rec_len + r0, n, mask.data(), P.overlapping && !whole, work.data(), S->c.ops, out + pos);
    }
    pos += b; cnt += c;
  }
  *out_len = pos; *n_rejects = cnt;
  return 0;
}

static rej::DuplexParams duplex_reject_params(const fgx_options* o) {
  rej::DuplexParams P;
  P.min_bq = o->min_input_base_quality; P.overlapping = o->overlapping_consensus; P.trim = o->trim; P.single_strand_ok = o->duplex_min_reads[2] == 0;
  return P;
}
// reject_core.h on the host: the `--rejects` stream of the DUPLEX (kind 1) or CODEC (kind 2) caller for a batch, given which molecules gave
// their consensus (`kept[g]` != 0: on the device the pipeline's own output slots say so).  Same contract as fgx_simplex_rejects_host.
int fgx_strand_rejects_host(const fgx_options* o, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first, uint32_t n_grp,
                            const uint8_t* kept, uint8_t* out, uint64_t cap, uint64_t* out_len, uint64_t* n_rejects) {
  if (!o || o->struct_size != sizeof(fgx_options) || !out_len || !n_rejects || (n_grp && (!blob || !rec_off || !rec_len || !grp_first || !kept))) return 2;
  if (o->caller_kind != FGX_CALLER_DUPLEX && o->caller_kind != FGX_CALLER_CODEC) return 2;
  const bool codec = o->caller_kind == FGX_CALLER_CODEC;
  const rej::DuplexParams P = duplex_reject_params(o);
  std::unique_ptr<rej::Scratch> S(new rej::Scratch());
  std::unique_ptr<canon::CodecScratch> SC(new canon::CodecScratch());
  std::vector<uint8_t> work, code;
  uint64_t pos = 0, cnt = 0;
  for (uint32_t g = 0; g < n_grp; g++) {
    const uint32_t r0 = grp_first[g], n = grp_first[g + 1] - r0;
    uint64_t bytes = 0;
    for (uint32_t i = 0; i < n; i++) bytes += rec_len[r0 + i];
    work.resize(bytes + 16); code.assign(n + 1, 0);
    uint8_t corrected = 0;
    const int st = codec ? rej::codec_reject_mask(o->codec_max_reads_per_strand >= 0, blob, ~0ull, rec_off + r0, rec_len + r0, n, kept[g] != 0, code.data(), *SC)
                         : rej::duplex_reject_codes(P, blob, ~0ull, rec_off + r0, rec_len + r0, n, kept[g] != 0, work.data(), code.data(), *S, &corrected);
    if (st != rej::REJ_OK) return 1;
    uint32_t c = 0;
    const uint64_t b = rej::reject_bytes(rec_len + r0, n, code.data(), &c);
    if (out && b) {
      if (pos + b > cap) return 2;
      if (codec) rej::emit_rejects(blob, rec_off + r0, rec_len + r0, n, code.data(), false, work.data(), S->c.ops, out + pos);
      else rej::emit_rejects_by_class(blob, rec_off + r0, rec_len + r0, n, code.data(), corrected != 0, work.data(), S->c.ops, out + pos);
    }
    pos += b; cnt += c;
  }
  *out_len = pos; *n_rejects = cnt;
  return 0;
}

// Host-input entry: upload once, run the device-resident pipeline, bring the records back, and send
// only the families the fast path deferred through the general path, splicing both in group order.
typedef int (*general_fn)(fgx_caller*, const uint8_t*, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, uint32_t, fgx_output*);

// The general paths do their per-molecule work (filters, CIGAR majority, record assembly) on the host, one molecule after the
// other.  Molecules are independent, so a large batch is cut into contiguous shards, one helper caller (own stream, own device
// buffers) per host thread, and the shard outputs are concatenated in input order — what `--threads N` does in the reference
// (one caller object per batch, simplex.rs:637-644).  FGX_HOST_THREADS overrides the thread count (1 = inline).
static unsigned host_threads() {
  if (const char* e = getenv("FGX_HOST_THREADS")) { int v = atoi(e); if (v >= 1) return (unsigned)(v > 256 ? 256 : v); }
  unsigned hw = std::thread::hardware_concurrency();
  return hw == 0 ? 1 : hw > 64 ? 64 : hw;
}

static bool ensure_workers(fgx_caller* c, unsigned T) {
  while (c->workers.size() < T) {
    fgx_options o = c->opt;
    o.read_name_prefix = c->prefix.c_str(); o.read_group_id = c->rg.c_str(); o.device = c->device;
    fgx_caller* w = fgx_create(&o);
    if (!w) { c->err = std::string("helper caller: ") + fgx_global_error(); return false; }
    c->workers.push_back(w);
  }
  for (fgx_caller* w : c->workers) w->genome = c->genome;   // (read-only, resident once)
  return true;
}

static int run_general(fgx_caller* c, general_fn fn, const uint8_t* records, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                       const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  constexpr uint32_t MIN_GROUPS = 512;              // below this a shard is not worth a thread
  c->counter_names_used = false;                    // (set by the CODEC general path; stale values must not survive a sharded run)
  unsigned T = host_threads();
  if (T > n_grp / MIN_GROUPS) T = n_grp / MIN_GROUPS;
  if (T <= 1) return fn(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
  if (!ensure_workers(c, T)) return 3;
  // contiguous shards balanced by record count
  std::vector<uint32_t> bound(T + 1, n_grp);
  bound[0] = 0;
  {
    const uint64_t total = grp_first[n_grp] - grp_first[0];
    uint32_t g = 0;
    for (unsigned t = 1; t < T; t++) {
      const uint64_t want = grp_first[0] + total * t / T;
      while (g < n_grp && grp_first[g] < want) g++;
      bound[t] = g;
    }
  }
  std::vector<fgx_output> outs(T);
  std::vector<int> rcs(T, 0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; t++)
    th.emplace_back([&, t]() {
      fgx_caller* w = c->workers[t];
      w->err.clear();
      memset(&outs[t], 0, sizeof(fgx_output));
      if (bound[t + 1] == bound[t]) return;
      try {
        hip_check(hipSetDevice(c->device), "hipSetDevice");
        rcs[t] = fn(w, records, rec_off, rec_len, n_rec, grp_first + bound[t], bound[t + 1] - bound[t], &outs[t]);
      } catch (const std::exception& ex) { w->err = ex.what(); rcs[t] = 3; }
    });
  for (auto& x : th) x.join();
  for (unsigned t = 0; t < T; t++) if (rcs[t] != 0) { c->err = c->workers[t]->err; return rcs[t]; }
  if (c->opt.caller_kind == FGX_CALLER_CODEC)        // counter-named reads depend on everything before them: redo in one piece
    for (unsigned t = 0; t < T; t++) if (c->workers[t]->counter_names_used) return fn(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
  c->out_data.clear(); c->out_rejects.clear();
  c->grp_out_end.assign(n_grp, 0);
  size_t total_data = 0, total_rej = 0;
  for (unsigned t = 0; t < T; t++) { total_data += outs[t].data_len; total_rej += outs[t].rejects_len; }
  c->out_data.reserve(total_data); c->out_rejects.reserve(total_rej);
  memset(out, 0, sizeof(*out));
  for (unsigned t = 0; t < T; t++) {
    const size_t base = c->out_data.size();
    const fgx_caller* w = c->workers[t];
    if (outs[t].data_len) c->out_data.insert(c->out_data.end(), outs[t].data, outs[t].data + outs[t].data_len);
    if (outs[t].rejects_len) c->out_rejects.insert(c->out_rejects.end(), outs[t].rejects, outs[t].rejects + outs[t].rejects_len);
    for (uint32_t k = 0; k < bound[t + 1] - bound[t]; k++) c->grp_out_end[bound[t] + k] = base + (k < w->grp_out_end.size() ? w->grp_out_end[k] : outs[t].data_len);
    out->count += outs[t].count; out->n_rejects += outs[t].n_rejects;
    for (int i = 0; i < FGX_STATS_LEN; i++) out->stats[i] += outs[t].stats[i];
    out->ms_host_prep = std::max(out->ms_host_prep, outs[t].ms_host_prep); out->ms_kernels = std::max(out->ms_kernels, outs[t].ms_kernels);
    out->ms_h2d = std::max(out->ms_h2d, outs[t].ms_h2d); out->ms_emit = std::max(out->ms_emit, outs[t].ms_emit);
  }
  out->data = c->out_data.data(); out->data_len = c->out_data.size();
  out->rejects = c->out_rejects.data(); out->rejects_len = c->out_rejects.size();
  return 0;
}
using clk = std::chrono::steady_clock;
static double ms_between(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

// host batch → the caller's device staging buffers (blocking)
static void hybrid_upload(fgx_caller* c, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                          const uint32_t* grp_first, uint32_t n_grp) {
  hip_check(hipSetDevice(c->device), "hipSetDevice");
  c->d_in_blob.reserve(records_len + 16);
  c->d_in_off.reserve((size_t)n_rec * 8 + 8);
  c->d_in_len.reserve((size_t)n_rec * 4 + 4);
  c->d_in_grp.reserve((size_t)(n_grp + 1) * 4);
  hip_check(hipMemcpyAsync(c->d_in_blob.p, records, records_len, hipMemcpyHostToDevice, c->stream), "H2D blob");
  hip_check(hipMemcpyAsync(c->d_in_off.p, rec_off, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream), "H2D rec_off");
  hip_check(hipMemcpyAsync(c->d_in_len.p, rec_len, (size_t)n_rec * 4, hipMemcpyHostToDevice, c->stream), "H2D rec_len");
  hip_check(hipMemcpyAsync(c->d_in_grp.p, grp_first, (size_t)(n_grp + 1) * 4, hipMemcpyHostToDevice, c->stream), "H2D grp_first");
  hip_check(hipStreamSynchronize(c->stream), "sync");
}

static int hybrid_after_upload(fgx_caller* c, general_fn general, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                               const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp, fgx_output* out,
                               uint8_t* dst, uint64_t dst_cap);

// The methylation-aware mode in the device-resident pipeline (round 4; FGX_METH_DEVICE=0 opts out): simplex caller, a reference handed over,
// no --trim, no --rejects.  Everything else of the mode stays on the general path.
static bool meth_device_enabled(const fgx_caller* c) {
  if (c->opt.caller_kind != FGX_CALLER_SIMPLEX || c->opt.trim || c->opt.track_rejects || !c->genome) return false;
  const char* e = getenv("FGX_METH_DEVICE");
  return !(e && e[0] == '0');
}

static int process_hybrid(fgx_caller* c, general_fn general, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                          const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  // --rejects (record copies in input order) and the methylation-aware mode (reference lookups per source read) are decided by the
  // general path: host orchestration, device kernels for the per-base work
  // Simplex caller (default; FGX_REJECTS_DEVICE=0 opts out): the rejects come from side kernels that evaluate the caller's rejection decisions on the
  // uploaded records, a lane per group (reject_device.hip / reject_core.h), and the records from the device pipeline as without --rejects.
  const bool dev_rejects = c->opt.track_rejects && c->opt.caller_kind == FGX_CALLER_SIMPLEX && c->opt.methylation_mode == FGX_METHYLATION_DISABLED && n_grp != 0 &&
                           rejects_device_enabled();
  // Duplex / CODEC callers (round 6): the same side kernels, AFTER the device pipeline — whether a molecule gave its consensus is read from the pipeline's
  // output slots (reject_core.h duplex_reject_codes / codec_reject_mask).  A batch with a deferred molecule or one out of the kernels' scope: general path.
  const bool strand_rejects = c->opt.track_rejects && (c->opt.caller_kind == FGX_CALLER_DUPLEX || c->opt.caller_kind == FGX_CALLER_CODEC) &&
                              c->opt.methylation_mode == FGX_METHYLATION_DISABLED && n_grp != 0 && rejects_device_enabled();
  c->last_canon_molecules = 0;
  c->last_deferred_groups = n_grp;          // (diagnostics, fgx_debug_last_deferral: the whole batch on the general path counts as every group deferred)
  if ((c->opt.track_rejects && !dev_rejects && !strand_rejects) || (c->opt.methylation_mode != FGX_METHYLATION_DISABLED && !meth_device_enabled(c)) || n_grp == 0) return run_general(c, general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
  if (!c->fast) c->fast = new FastState();
  auto t0 = clk::now();
  hybrid_upload(c, records, records_len, rec_off, rec_len, n_rec, grp_first, n_grp);
  auto t1 = clk::now();
  RejectResult rr = {nullptr, 0, 0, 0, 0.0};
  if (dev_rejects) {
    simplex_rejects_device(c, reject_params(&c->opt), c->d_in_blob.as<uint8_t>(), records_len, c->d_in_off.as<uint64_t>(), c->d_in_len.as<uint32_t>(), n_rec, c->d_in_grp.as<uint32_t>(), n_grp, &rr);
    c->last_reject_oos = rr.n_out_of_scope;
    if (rr.n_out_of_scope) return run_general(c, general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);   // (> 128 records, > 16 CIGAR ops, malformed records)
    c->rejects_host.resize(rr.bytes);
    if (rr.bytes) hip_check(hipMemcpy(c->rejects_host.data(), rr.d_out, rr.bytes, hipMemcpyDeviceToHost), "D2H rejects");
  }
  // (the rejects of EVERY group are in hand, also of the groups the device pipeline defers: the general path need not track them again)
  struct Untrack { fgx_caller* c; uint8_t saved; Untrack(fgx_caller* cc, bool on) : c(cc), saved(cc->opt.track_rejects) { if (on) set(0); }
                   void set(uint8_t v) { c->opt.track_rejects = v; for (fgx_caller* w : c->workers) w->opt.track_rejects = v; }
                   ~Untrack() { set(saved); } } untrack(c, dev_rejects || strand_rejects);
  int rc = hybrid_after_upload(c, general, records, records_len, rec_off, rec_len, n_rec, grp_first, n_grp, out, nullptr, 0);
  if (rc == 0 && strand_rejects) {
    const FastResult& fr = c->fast->last;      // (hybrid_after_upload left the device pass's result there; the uploaded records are still in d_in_*)
    bool served = c->last_deferred_groups == 0;
    if (served) {
      strand_rejects_device(c, c->d_in_blob.as<uint8_t>(), records_len, c->d_in_off.as<uint64_t>(), c->d_in_len.as<uint32_t>(), n_rec, c->d_in_grp.as<uint32_t>(), n_grp,
                            fr.d_out_off, 3, fr.out_len, &rr);
      c->last_reject_oos = rr.n_out_of_scope;
      served = rr.n_out_of_scope == 0;
    }
    if (!served) {                                   // the whole batch again, on the general path, which tracks its rejects itself
      untrack.set(untrack.saved);
      return run_general(c, general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
    }
    c->rejects_host.resize(rr.bytes);
    if (rr.bytes) hip_check(hipMemcpy(c->rejects_host.data(), rr.d_out, rr.bytes, hipMemcpyDeviceToHost), "D2H rejects");
  }
  if (rc == 0 && (dev_rejects || strand_rejects)) { out->rejects = c->rejects_host.data(); out->rejects_len = rr.bytes; out->n_rejects = rr.count; out->ms_kernels += rr.ms; }
  if (rc == 0) out->ms_h2d = ms_between(t0, t1);
  return rc;
}

// Kernels, download and the splice with the general path's output for the deferred families.  `dst` (pinned, dst_cap bytes):
// when the records fit they are downloaded straight there and out->data == dst; otherwise they land in the caller's own buffers.
// Default since round 4 (FGX_DUPLEX_CANON=0 / FGX_CODEC_CANON=0 opt out): duplex molecules the device pipeline deferred because of indel / skip / pad CIGARs are rewritten into
// their canonical form (canon_core.h: overlap correction, mate clip and alignment filter applied; every read `<len>M`) on the host's
// cores and decided by the device pipeline in a SECOND pass; only what the canonical form cannot express (and what the second pass
// still defers) takes the general path.  tests/test_canon_core.py shows through the oracle that the canonical molecule plus the
// counted delta gives the original's result.  On by default since the whole GPU suite ran with it (round 4).
static bool duplex_canon_enabled() { return opt_in("FGX_DUPLEX_CANON"); }
// The same for CODEC molecules (FGX_CODEC_CANON=1; canon_core.h `canon_codec_molecule`, proof tests/test_canon_codec.py): virtual clip
// applied, `<len>M`, reads placed so that the overlap geometry and the consensus length come out the same; every record is kept and
// no counter moves, so there is no delta.  Only molecules the original would EMIT are in scope; rejected ones stay on the general path.
// FGX_CANON_DEVICE=1 (with either flag above): the canonical form is computed by a device kernel (canon_device.hip, the same scalar source,
// a lane per molecule) from the records already uploaded, instead of on the host's cores; the records do not come back.
static bool canon_device_enabled() { return opt_in("FGX_CANON_DEVICE"); }
static bool codec_canon_enabled() { return opt_in("FGX_CODEC_CANON"); }

struct CanonPass {
  std::vector<uint8_t> used;            // per deferred group: 1 = its records come from the second device pass
  std::vector<uint64_t> beg, end;       // ... their byte range in `out`
  std::vector<uint8_t> out;             // the second pass's records (host copy)
  uint64_t count = 0, stats[FGX_STATS_LEN] = {0};
  double ms_kernels = 0, ms_host = 0;
};

static void canon_second_pass(fgx_caller* c, const uint8_t* records, const uint64_t* rec_off, const uint32_t* rec_len, const uint32_t* grp_first,
                              const std::vector<uint32_t>& def, CanonPass& cp) {
  const size_t nd = def.size();
  cp.used.assign(nd, 0); cp.beg.assign(nd, 0); cp.end.assign(nd, 0);
  auto t0 = clk::now();
  // slots of the deferred molecules' records in one compact buffer: 4-byte length prefix + room for the original body
  std::vector<uint64_t> first(nd + 1, 0);
  for (size_t k = 0; k < nd; k++) first[k + 1] = first[k] + (grp_first[def[k] + 1] - grp_first[def[k]]);
  const uint64_t n_slots = first[nd];
  std::vector<uint64_t> out_off(n_slots);
  std::vector<uint32_t> out_len(n_slots, 0);
  uint64_t bytes = 0;
  for (size_t k = 0; k < nd; k++)
    for (uint32_t r = grp_first[def[k]], i = 0; r < grp_first[def[k] + 1]; r++, i++) { out_off[first[k] + i] = bytes + 4; bytes += 4ull + rec_len[r]; }
  const bool on_device = canon_device_enabled();
  std::vector<uint8_t> blob(on_device ? 0 : bytes + 16, 0);
  std::vector<int> status(nd, canon::CANON_OUT_OF_SCOPE);
  std::vector<canon::Delta> delta(nd);
  memset(delta.data(), 0, nd * sizeof(canon::Delta));
  const bool codec = c->opt.caller_kind == FGX_CALLER_CODEC;
  const canon::Params P = canon_params(&c->opt);
  const canon::CodecParams PC = canon_codec_params(&c->opt);
  c->d_canon_blob.reserve(bytes + 16);
  if (on_device) {
    // FGX_CANON_DEVICE=1: a lane per molecule over the records hybrid_upload already put on the device (canon_device.hip); the
    // canonical records are written into d_canon_blob there, and only status / lengths / delta come back
    DevBuf& aux = c->d_canon_aux;      // first[nd+1] u64 | out_off[n_slots] u64 | delta[nd] | def[nd] u32 | status[nd] i32 | out_len[n_slots] u32
    const size_t o_first = 0, o_off = o_first + (nd + 1) * 8, o_delta = o_off + n_slots * 8, o_def = o_delta + nd * sizeof(canon::Delta),
                 o_status = o_def + nd * 4, o_len = o_status + nd * 4, total = o_len + n_slots * 4;
    aux.reserve(total + 16);
    uint8_t* a = aux.as<uint8_t>();
    hipStream_t s = c->stream;
    hip_check(hipMemsetAsync(c->d_canon_blob.p, 0, bytes + 16, s), "memset canonical blob");
    hip_check(hipMemsetAsync(a + o_len, 0, n_slots * 4, s), "memset canonical lengths");
    hip_check(hipMemcpyAsync(a + o_first, first.data(), (nd + 1) * 8, hipMemcpyHostToDevice, s), "H2D canonical slots");
    hip_check(hipMemcpyAsync(a + o_off, out_off.data(), n_slots * 8, hipMemcpyHostToDevice, s), "H2D canonical slot offsets");
    hip_check(hipMemcpyAsync(a + o_def, def.data(), nd * 4, hipMemcpyHostToDevice, s), "H2D deferred groups");
    launch_canon_molecules(s, codec, P, PC, c->d_in_blob.as<uint8_t>(), c->d_in_off.as<uint64_t>(), c->d_in_len.as<uint32_t>(), c->d_in_grp.as<uint32_t>(),
                           (const uint32_t*)(a + o_def), (uint32_t)nd, (const uint64_t*)(a + o_first), c->d_canon_blob.as<uint8_t>(), (const uint64_t*)(a + o_off),
                           (uint32_t*)(a + o_len), (int*)(a + o_status), (canon::Delta*)(a + o_delta), c->d_canon_slabs);
    hip_check(hipMemcpyAsync(status.data(), a + o_status, nd * 4, hipMemcpyDeviceToHost, s), "D2H canonical status");
    hip_check(hipMemcpyAsync(out_len.data(), a + o_len, n_slots * 4, hipMemcpyDeviceToHost, s), "D2H canonical lengths");
    if (!codec) hip_check(hipMemcpyAsync(delta.data(), a + o_delta, nd * sizeof(canon::Delta), hipMemcpyDeviceToHost, s), "D2H canonical delta");
    hip_check(hipStreamSynchronize(s), "canonicalisation kernel");
  } else {
    unsigned T = host_threads();
    if (T > nd / 64 + 1) T = (unsigned)(nd / 64 + 1);
    auto work = [&](unsigned t) {
      std::unique_ptr<canon::Scratch> S(codec ? nullptr : new canon::Scratch());
      std::unique_ptr<canon::CodecScratch> SC(codec ? new canon::CodecScratch() : nullptr);
      for (size_t k = t; k < nd; k += T) {
        const uint32_t r0 = grp_first[def[k]], n = grp_first[def[k] + 1] - r0;
        status[k] = codec ? canon::canon_codec_molecule(PC, records, rec_off + r0, rec_len + r0, n, blob.data(), out_off.data() + first[k], out_len.data() + first[k], *SC)
                          : canon::canon_duplex_molecule(P, records, rec_off + r0, rec_len + r0, n, blob.data(), out_off.data() + first[k], out_len.data() + first[k], *S, delta[k]);
      }
    };
    if (T <= 1) work(0);
    else { std::vector<std::thread> th; for (unsigned t = 0; t < T; t++) th.emplace_back(work, t); for (auto& x : th) x.join(); }
  }
  // the canonical molecules as one batch
  std::vector<uint64_t> c_off;
  std::vector<uint32_t> c_len, c_grp(1, 0), c_def;
  for (size_t k = 0; k < nd; k++) {
    if (status[k] != canon::CANON_OK) continue;
    for (uint64_t i = first[k]; i < first[k + 1]; i++)
      if (out_len[i]) {
        c_off.push_back(out_off[i]); c_len.push_back(out_len[i]);
        if (!on_device) { const uint32_t L = out_len[i]; memcpy(blob.data() + out_off[i] - 4, &L, 4); }   // (the kernel wrote its own prefixes)
      }
    c_grp.push_back((uint32_t)c_off.size());
    c_def.push_back((uint32_t)k);
  }
  cp.ms_host = ms_between(t0, clk::now());
  const uint32_t n_cg = (uint32_t)c_def.size(), n_cr = (uint32_t)c_off.size();
  if (n_cg == 0) return;
  c->d_canon_off.reserve((size_t)n_cr * 8 + 8); c->d_canon_len.reserve((size_t)n_cr * 4 + 4); c->d_canon_grp.reserve((size_t)(n_cg + 1) * 4);
  if (!on_device) hip_check(hipMemcpyAsync(c->d_canon_blob.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream), "H2D canonical blob");
  if (n_cr) {
    hip_check(hipMemcpyAsync(c->d_canon_off.p, c_off.data(), (size_t)n_cr * 8, hipMemcpyHostToDevice, c->stream), "H2D canonical rec_off");
    hip_check(hipMemcpyAsync(c->d_canon_len.p, c_len.data(), (size_t)n_cr * 4, hipMemcpyHostToDevice, c->stream), "H2D canonical rec_len");
  }
  hip_check(hipMemcpyAsync(c->d_canon_grp.p, c_grp.data(), (size_t)(n_cg + 1) * 4, hipMemcpyHostToDevice, c->stream), "H2D canonical grp_first");
  hip_check(hipStreamSynchronize(c->stream), "sync");
  FastResult fr2;
  c->fast->fp.run(c, c->d_canon_blob.as<uint8_t>(), bytes, c->d_canon_off.as<uint64_t>(), c->d_canon_len.as<uint32_t>(), n_cr, c->d_canon_grp.as<uint32_t>(), n_cg, &fr2);
  cp.out.resize(fr2.out_len);
  if (fr2.out_len) hip_check(hipMemcpy(cp.out.data(), fr2.d_out, fr2.out_len, hipMemcpyDeviceToHost), "D2H canonical out");
  std::vector<uint64_t> slot2((size_t)3 * n_cg);
  hip_check(hipMemcpy(slot2.data(), fr2.d_out_off, slot2.size() * 8, hipMemcpyDeviceToHost), "D2H canonical offsets");
  std::vector<uint32_t> def2(fr2.n_deferred);
  if (fr2.n_deferred) hip_check(hipMemcpy(def2.data(), fr2.d_deferred, (size_t)fr2.n_deferred * 4, hipMemcpyDeviceToHost), "D2H canonical deferred");
  std::vector<uint8_t> again(n_cg, 0);
  for (uint32_t g : def2) if (g < n_cg) again[g] = 1;
  for (uint32_t ci = 0; ci < n_cg; ci++) {
    if (again[ci]) continue;                                   // (still deferred: the general path takes the ORIGINAL molecule)
    const size_t k = c_def[ci];
    cp.used[k] = 1;
    cp.beg[k] = slot2[(size_t)3 * ci];
    cp.end[k] = ci + 1 < n_cg ? slot2[(size_t)3 * (ci + 1)] : fr2.out_len;
    cp.stats[0] += delta[k].minority; cp.stats[2] += delta[k].minority; cp.stats[3 + FGX_REJ_MINORITY_ALIGNMENT] += delta[k].minority;
    for (int i = 0; i < 4; i++) cp.stats[24 + i] += delta[k].ov[i];
  }
  for (int i = 0; i < FGX_STATS_LEN; i++) cp.stats[i] += fr2.stats[i];
  cp.count = fr2.count;
  cp.ms_kernels = fr2.ms_kernels;
  for (size_t k = 0; k < nd; k++) c->last_canon_molecules += cp.used[k];
}

static int hybrid_after_upload(fgx_caller* c, general_fn general, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                               const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp, fgx_output* out,
                               uint8_t* dst, uint64_t dst_cap) {
  auto ms = ms_between;
  if (!c->fast) c->fast = new FastState();
  c->fast->has_last = false;   // this run overwrites (and may reallocate) the device buffers a device-resident batch left behind
  c->last_group_off = nullptr;
  hip_check(hipSetDevice(c->device), "hipSetDevice");
  auto t0 = clk::now();
  auto t1 = t0;
  FastResult fr;
  c->fast->fp.run(c, c->d_in_blob.as<uint8_t>(), records_len, c->d_in_off.as<uint64_t>(), c->d_in_len.as<uint32_t>(), n_rec,
                  c->d_in_grp.as<uint32_t>(), n_grp, &fr);
  c->fast->last = fr;          // (has_last stays false: process_hybrid's duplex / CODEC rejects read the slot offsets of THIS pass)
  auto t2 = clk::now();
  // records land in a pinned host buffer owned by the caller object (pageable destinations cost ~10x: first-touch faults + staging)
  const bool direct = dst && fr.n_deferred == 0 && fr.out_len <= dst_cap;
  if (!direct) c->fast->pin_out.reserve(fr.out_len + 16);
  const uint8_t* fast_out = direct ? dst : c->fast->pin_out.as<uint8_t>();
  if (fr.out_len) hip_check(hipMemcpy((void*)fast_out, fr.d_out, fr.out_len, hipMemcpyDeviceToHost), "D2H out");
  auto t3 = clk::now();
  memset(out, 0, sizeof(*out));
  c->last_deferred_groups = fr.n_deferred;
  if (fr.n_deferred == 0) {
    out->data = fast_out; out->data_len = fr.out_len; out->count = fr.count;
    for (int i = 0; i < FGX_STATS_LEN; i++) out->stats[i] = fr.stats[i];
    out->ms_h2d = ms(t0, t1); out->ms_kernels = fr.ms_kernels; out->ms_d2h = ms(t2, t3);
    return 0;
  }
  // deferred families → general path, then splice in group order
  std::vector<uint32_t> def(fr.n_deferred);
  hip_check(hipMemcpy(def.data(), fr.d_deferred, (size_t)fr.n_deferred * 4, hipMemcpyDeviceToHost), "D2H deferred");
  std::sort(def.begin(), def.end());
  std::vector<uint64_t> slot_off((size_t)3 * n_grp);
  hip_check(hipMemcpy(slot_off.data(), fr.d_out_off, slot_off.size() * 8, hipMemcpyDeviceToHost), "D2H offsets");
  CanonPass cp;
  cp.used.assign(def.size(), 0);
  c->last_canon_molecules = 0;
  c->last_deferred_groups = (uint64_t)def.size();
  if ((c->opt.caller_kind == FGX_CALLER_DUPLEX && duplex_canon_enabled()) || (c->opt.caller_kind == FGX_CALLER_CODEC && codec_canon_enabled()))
    canon_second_pass(c, records, rec_off, rec_len, grp_first, def, cp);
  std::vector<uint64_t> d_off;
  std::vector<uint32_t> d_len, d_grp(1, 0);
  for (size_t k = 0; k < def.size(); k++) {
    if (cp.used[k]) continue;
    const uint32_t g = def[k];
    for (uint32_t r = grp_first[g]; r < grp_first[g + 1]; r++) { d_off.push_back(rec_off[r]); d_len.push_back(rec_len[r]); }
    d_grp.push_back((uint32_t)d_off.size());
  }
  fgx_output gen;
  memset(&gen, 0, sizeof(gen));
  c->out_data.clear(); c->grp_out_end.clear();
  c->counter_names_used = false;   // (run_general resets it too; it is not called when the second pass took every deferred group)
  int rc = d_grp.size() > 1 ? run_general(c, general, records, d_off.data(), d_len.data(), (uint32_t)d_off.size(), d_grp.data(), (uint32_t)d_grp.size() - 1, &gen) : 0;
  if (rc != 0) return rc;
  // CODEC molecules without an MI are named by a counter that advances on EVERY emitted record (codec_caller.rs:1568-1577):
  // the deferred subset alone would restart it at 0, so the whole batch goes through the general path in one piece
  if (c->opt.caller_kind == FGX_CALLER_CODEC && c->counter_names_used)
    return run_general(c, general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
  std::vector<uint8_t> merged;
  merged.reserve(fr.out_len + c->out_data.size() + cp.out.size());
  uint64_t fpos = 0;   // fast output is contiguous in group order; deferred groups contributed nothing to it
  uint64_t gprev = 0;
  size_t gk = 0;       // index among the groups the general path took
  for (size_t k = 0; k < def.size(); k++) {
    uint64_t upto = slot_off[(size_t)3 * def[k]];          // fast bytes of all groups before def[k]
    merged.insert(merged.end(), fast_out + fpos, fast_out + upto);
    fpos = upto;
    if (cp.used[k]) merged.insert(merged.end(), cp.out.begin() + cp.beg[k], cp.out.begin() + cp.end[k]);
    else {
      merged.insert(merged.end(), c->out_data.begin() + gprev, c->out_data.begin() + c->grp_out_end[gk]);
      gprev = c->grp_out_end[gk];
      gk++;
    }
  }
  merged.insert(merged.end(), fast_out + fpos, fast_out + fr.out_len);
  c->out_data.swap(merged);
  out->data = c->out_data.data(); out->data_len = c->out_data.size(); out->count = fr.count + gen.count + cp.count;
  for (int i = 0; i < FGX_STATS_LEN; i++) out->stats[i] = fr.stats[i] + gen.stats[i] + cp.stats[i];
  out->ms_h2d = ms(t0, t1); out->ms_kernels = fr.ms_kernels + gen.ms_kernels + cp.ms_kernels; out->ms_d2h = ms(t2, t3);
  out->ms_host_prep = gen.ms_host_prep + cp.ms_host; out->ms_emit = gen.ms_emit;
  return 0;
}

int fgx_process_batch(fgx_caller* c, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off, const uint32_t* rec_len,
                      uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp, fgx_output* out) {
  if (!c || !out) return 1;
  c->err.clear();
  try {
    for (uint32_t r = 0; r < n_rec; r++) {   // overflow-safe: rec_off + rec_len may wrap in u64
      if (rec_len[r] < 32 || rec_len[r] > records_len || rec_off[r] > records_len - rec_len[r]) { c->err = "fgx_process_batch: record outside the blob or shorter than the fixed BAM header"; return 1; }
    }
    for (uint32_t g = 0; g < n_grp; g++)
      if (grp_first[g] > grp_first[g + 1]) { c->err = "fgx_process_batch: group boundaries must be non-decreasing"; return 1; }
    if (n_grp && grp_first[n_grp] > n_rec) { c->err = "fgx_process_batch: group boundaries exceed n_rec"; return 1; }
    switch (c->opt.caller_kind) {
      case FGX_CALLER_SIMPLEX:
        if (c->general_only) return run_general(c, simplex_process_general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
        return process_hybrid(c, simplex_process_general, records, records_len, rec_off, rec_len, n_rec, grp_first, n_grp, out);
      case FGX_CALLER_DUPLEX:
        if (c->general_only) return run_general(c, duplex_process_general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
        if (c->opt.duplex_min_reads[1] > c->opt.duplex_min_reads[0] || c->opt.duplex_min_reads[2] > c->opt.duplex_min_reads[1])
          return duplex_process_general(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);   // raises the reference's error
        return process_hybrid(c, duplex_process_general, records, records_len, rec_off, rec_len, n_rec, grp_first, n_grp, out);
#ifdef FGX_HAVE_CODEC
      case FGX_CALLER_CODEC:
        // the device pipeline emits every molecule that passes the geometry gates: only valid while the duplex-disagreement
        // thresholds cannot reject anything (their defaults); otherwise the general path decides after the strand combine
        if (c->general_only || c->opt.codec_max_duplex_disagreements != 0xFFFFFFFFu || c->opt.codec_max_duplex_disagreement_rate < 1.0)
          return run_general(c, codec_process_general, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
        return process_hybrid(c, codec_process_general, records, records_len, rec_off, rec_len, n_rec, grp_first, n_grp, out);
#endif
      default: c->err = "fgx_process_batch: caller kind not implemented"; return 1;
    }
  } catch (const std::exception& ex) {
    c->err = ex.what();
    return 3;
  }
}

// FGX_CANON_RESIDENT=1 (with FGX_DUPLEX_CANON / FGX_CODEC_CANON): the canonical second pass INSIDE the device-resident entry.  The molecules
// the first pass deferred are canonicalised by the kernel of canon_device.hip where they lie, decided by the device pipeline in a second
// pass, and the two passes' records are merged in group order on the device; what the canonical form cannot express, or the second pass
// defers again, stays in the deferred list the caller re-submits.  The host sees the deferred indices, per-molecule status and counted
// deltas — never a record.  Default since round 4 (FGX_CANON_RESIDENT=0 opts out); tests/test_apiemu.py also runs it on the CPU.
static bool canon_resident_enabled(int kind) {
  if (!opt_in("FGX_CANON_RESIDENT")) return false;
  return (kind == FGX_CALLER_DUPLEX && duplex_canon_enabled()) || (kind == FGX_CALLER_CODEC && codec_canon_enabled());
}

struct ResidentOut { const uint8_t* d_out; uint64_t out_len, count, stats[FGX_STATS_LEN]; uint32_t n_deferred; const uint32_t* d_deferred; uint64_t n_canon; double ms_kernels;
                     const uint64_t* d_group_off; };   // byte offset of every group in the merged stream (n_grp + 1 entries, device)

static bool canon_resident_pass(fgx_caller* c, const uint8_t* d_blob, const uint64_t* d_rec_off, const uint32_t* d_rec_len, const uint32_t* d_grp_first, uint32_t n_grp,
                                const FastResult& fr, ResidentOut* ro) {
  hipStream_t s = c->stream;
  const uint32_t nd = fr.n_deferred;
  const bool codec = c->opt.caller_kind == FGX_CALLER_CODEC;
  std::vector<uint32_t> def(nd);
  hip_check(hipMemcpy(def.data(), fr.d_deferred, (size_t)nd * 4, hipMemcpyDeviceToHost), "D2H deferred");
  std::sort(def.begin(), def.end());
  // the first pass's records and slot offsets are stashed: the second run reuses the pipeline's buffers
  const uint64_t len1 = fr.out_len, count1 = fr.count;
  uint64_t stats1[FGX_STATS_LEN];
  for (int i = 0; i < FGX_STATS_LEN; i++) stats1[i] = fr.stats[i];
  c->d_res_out1.reserve(len1 + 16);
  c->d_res_off1.reserve((size_t)3 * n_grp * 8 + 8);
  if (len1) hip_check(hipMemcpyAsync(c->d_res_out1.p, fr.d_out, len1, hipMemcpyDeviceToDevice, s), "stash first-pass records");
  hip_check(hipMemcpyAsync(c->d_res_off1.p, fr.d_out_off, (size_t)3 * n_grp * 8, hipMemcpyDeviceToDevice, s), "stash first-pass offsets");
  // aux: work 4(nd+1) u64 | first (nd+1) u64 | delta nd x 40 B | def nd u32 | status nd i32 | used nd u8
  const size_t o_work = 0, o_first = o_work + 4ull * (nd + 1) * 8, o_delta = o_first + (nd + 1) * 8ull, o_def = o_delta + (size_t)nd * sizeof(canon::Delta),
               o_status = o_def + (size_t)nd * 4, o_used = o_status + (size_t)nd * 4, total = o_used + nd;
  c->d_res_aux.reserve(total + 64);
  uint8_t* a = c->d_res_aux.as<uint8_t>();
  unsigned long long* work = (unsigned long long*)(a + o_work);
  unsigned long long* d_first = (unsigned long long*)(a + o_first);
  canon::Delta* d_delta = (canon::Delta*)(a + o_delta);
  uint32_t* d_def = (uint32_t*)(a + o_def);
  int* d_status = (int*)(a + o_status);
  uint8_t* d_used = a + o_used;
  hip_check(hipMemcpyAsync(d_def, def.data(), (size_t)nd * 4, hipMemcpyHostToDevice, s), "H2D deferred groups");
  hip_check(hipMemsetAsync(d_delta, 0, (size_t)nd * sizeof(canon::Delta), s), "memset delta");
  uint64_t n_slots = 0, bytes = 0;
  canon_layout_device(s, d_rec_len, d_grp_first, d_def, nd, work, d_first, c->d_res_outoff, c->d_res_scan, &n_slots, &bytes);
  c->d_canon_blob.reserve(bytes + 16);
  c->d_canon_aux.reserve(n_slots * 4 + 16);                     // (here: the canonical lengths alone)
  uint32_t* d_out_len = c->d_canon_aux.as<uint32_t>();
  hip_check(hipMemsetAsync(c->d_canon_blob.p, 0, bytes + 16, s), "memset canonical blob");
  hip_check(hipMemsetAsync(d_out_len, 0, n_slots * 4 + 4, s), "memset canonical lengths");
  launch_canon_molecules(s, codec, canon_params(&c->opt), canon_codec_params(&c->opt), d_blob, d_rec_off, d_rec_len, d_grp_first, d_def, nd, (const uint64_t*)d_first,
                         c->d_canon_blob.as<uint8_t>(), c->d_res_outoff.as<uint64_t>(), d_out_len, d_status, d_delta, c->d_canon_slabs);
  uint32_t n_cg = 0, n_cr = 0;
  canon_compact_device(s, d_status, d_first, c->d_res_outoff.as<uint64_t>(), d_out_len, nd, work, c->d_canon_off, c->d_canon_len, c->d_canon_grp, c->d_res_cdef, c->d_res_scan,
                       &n_cg, &n_cr);
  hip_check(hipStreamSynchronize(s), "canonical lists");
  if (n_cg == 0) return false;                                   // nothing in scope: the first pass's buffers are untouched
  FastResult fr2;
  c->fast->fp.run(c, c->d_canon_blob.as<uint8_t>(), bytes, c->d_canon_off.as<uint64_t>(), c->d_canon_len.as<uint32_t>(), n_cr, c->d_canon_grp.as<uint32_t>(), n_cg, &fr2);
  uint64_t final_len = 0;
  resident_merge_device(s, n_grp, c->d_res_off1.as<uint64_t>(), c->d_res_out1.as<uint8_t>(), len1, d_def, nd, c->d_res_cdef.as<uint32_t>(), n_cg, fr2.d_out_off, fr2.d_out,
                        fr2.out_len, fr2.d_deferred, fr2.n_deferred, c->d_res_aux2, c->d_res_scan, d_used, c->d_res_final, &final_len);
  std::vector<uint8_t> used(nd);
  std::vector<canon::Delta> delta(nd);
  hip_check(hipMemcpy(used.data(), d_used, nd, hipMemcpyDeviceToHost), "D2H used");
  hip_check(hipMemcpy(delta.data(), d_delta, (size_t)nd * sizeof(canon::Delta), hipMemcpyDeviceToHost), "D2H delta");
  for (int i = 0; i < FGX_STATS_LEN; i++) ro->stats[i] = stats1[i] + fr2.stats[i];
  std::vector<uint32_t> left;
  uint64_t n_canon = 0;
  for (uint32_t k = 0; k < nd; k++) {
    if (!used[k]) { left.push_back(def[k]); continue; }
    n_canon++;
    if (!codec) {
      ro->stats[0] += delta[k].minority; ro->stats[2] += delta[k].minority; ro->stats[3 + FGX_REJ_MINORITY_ALIGNMENT] += delta[k].minority;
      for (int i = 0; i < 4; i++) ro->stats[24 + i] += delta[k].ov[i];
    }
  }
  c->d_res_deferred.reserve(left.size() * 4 + 4);
  if (!left.empty()) hip_check(hipMemcpy(c->d_res_deferred.p, left.data(), left.size() * 4, hipMemcpyHostToDevice), "H2D remaining deferred");
  ro->d_out = c->d_res_final.as<uint8_t>(); ro->out_len = final_len; ro->count = count1 + fr2.count;
  ro->n_deferred = (uint32_t)left.size(); ro->d_deferred = c->d_res_deferred.as<uint32_t>(); ro->n_canon = n_canon; ro->ms_kernels = fr2.ms_kernels;
  ro->d_group_off = (const uint64_t*)(c->d_res_aux2.as<unsigned long long>() + (n_grp + 1));     // resident_merge_device: sizes | offsets | ...
  return true;
}

int fgx_process_batch_device(fgx_caller* c, const void* d_records, uint64_t records_len, const void* d_rec_off, const void* d_rec_len,
                             uint32_t n_rec, const void* d_grp_first, uint32_t n_grp, fgx_output* out, uint32_t* n_deferred,
                             const void** d_deferred_groups) {
  if (!c || !out) return 1;
  c->err.clear();
  try {
    if (c->opt.caller_kind == FGX_CALLER_CODEC && (c->opt.codec_max_duplex_disagreements != 0xFFFFFFFFu || c->opt.codec_max_duplex_disagreement_rate < 1.0)) {
      c->err = "fgx_process_batch_device: CODEC duplex-disagreement thresholds need the host path (fgx_process_batch)"; return 1;
    }
    const bool dev_rejects = c->opt.track_rejects && c->opt.caller_kind == FGX_CALLER_SIMPLEX && rejects_device_enabled();
    const bool strand_rejects = c->opt.track_rejects && (c->opt.caller_kind == FGX_CALLER_DUPLEX || c->opt.caller_kind == FGX_CALLER_CODEC) && rejects_device_enabled();
    if (c->opt.track_rejects && !dev_rejects && !strand_rejects) { c->err = "fgx_process_batch_device: --rejects needs the host path (fgx_process_batch)"; return 1; }
    // methylation-aware mode: the simplex caller without --trim runs on the streaming kernels (simplex_deep.inc); FGX_METH_DEVICE=0, duplex or
    // --trim: the host entry (general path)
    if (c->opt.methylation_mode != FGX_METHYLATION_DISABLED && !meth_device_enabled(c)) { c->err = "fgx_process_batch_device: the methylation-aware mode of this caller needs the host entry (fgx_process_batch)"; return 1; }
    if (!c->fast) c->fast = new FastState();
    c->fast->has_last = false;   // set again only when this batch succeeds
    c->last_group_off = nullptr;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    FastResult fr;
    c->fast->fp.run(c, (const uint8_t*)d_records, records_len, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, n_rec,
                    (const uint32_t*)d_grp_first, n_grp, &fr);
    c->fast->last = fr; c->fast->has_last = true;
    memset(out, 0, sizeof(*out));
    out->data = fr.d_out; out->data_len = fr.out_len; out->count = fr.count;
    for (int i = 0; i < FGX_STATS_LEN; i++) out->stats[i] = fr.stats[i];
    out->ms_kernels = fr.ms_kernels; out->ms_k_family = fr.ms_k_family; out->ms_k_emit = fr.ms_k_emit;
    out->ms_emit = (double)fr.full_items;   /* device path: number of columns that needed call_full (diagnostic) */
    if (n_deferred) *n_deferred = fr.n_deferred;
    if (d_deferred_groups) *d_deferred_groups = fr.d_deferred;
    c->last_deferred_groups = fr.n_deferred; c->last_canon_molecules = 0;
    c->last_group_off = fr.d_out_off; c->last_group_stride = 3;
    uint32_t left_deferred = fr.n_deferred;      // (after the canonical second pass, when it runs)
    if (fr.n_deferred > 0 && canon_resident_enabled(c->opt.caller_kind)) {
      ResidentOut ro;
      if (canon_resident_pass(c, (const uint8_t*)d_records, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, (const uint32_t*)d_grp_first, n_grp, fr, &ro)) {
        c->fast->has_last = false;                  // (the slot tables of `last` describe one pass only: fgx_filter_last_output_device has to be given the records)
        out->data = ro.d_out; out->data_len = ro.out_len; out->count = ro.count;
        for (int i = 0; i < FGX_STATS_LEN; i++) out->stats[i] = ro.stats[i];
        out->ms_kernels += ro.ms_kernels;
        if (n_deferred) *n_deferred = ro.n_deferred;
        if (d_deferred_groups) *d_deferred_groups = ro.d_deferred;
        left_deferred = ro.n_deferred;
        c->last_canon_molecules = ro.n_canon;
        c->last_group_off = ro.d_group_off; c->last_group_stride = 1;
      }
    }
    if (strand_rejects) {   // duplex / CODEC (round 6): from the records and the batch's own output slots (which molecules gave their consensus)
      if (left_deferred != 0) { c->err = "fgx_process_batch_device: --rejects: the batch has molecules the device pipeline defers; use the host entry (fgx_process_batch)"; return 1; }
      RejectResult rr;
      strand_rejects_device(c, (const uint8_t*)d_records, records_len, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, n_rec, (const uint32_t*)d_grp_first, n_grp,
                            c->last_group_off, c->last_group_stride, out->data_len, &rr);
      c->last_reject_oos = rr.n_out_of_scope;
      if (rr.n_out_of_scope) { c->err = "fgx_process_batch_device: --rejects: a molecule is out of the side kernels' scope (more than 128 records or 16 CIGAR ops, a per-strand cap, malformed records); use the host entry (fgx_process_batch)"; return 1; }
      out->rejects = rr.d_out; out->rejects_len = rr.bytes; out->n_rejects = rr.count; out->ms_kernels += rr.ms;
    }
    if (dev_rejects) {   // out->rejects is a DEVICE pointer here, like out->data; it covers every group, the deferred ones included
      RejectResult rr;
      simplex_rejects_device(c, reject_params(&c->opt), (const uint8_t*)d_records, records_len, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, n_rec, (const uint32_t*)d_grp_first, n_grp, &rr);
      c->last_reject_oos = rr.n_out_of_scope;
      if (rr.n_out_of_scope) { c->err = "fgx_process_batch_device: --rejects: a group is out of the side kernels' scope (more than 128 records or 16 CIGAR ops, malformed records); use the host entry (fgx_process_batch)"; return 1; }
      out->rejects = rr.d_out; out->rejects_len = rr.bytes; out->n_rejects = rr.count; out->ms_kernels += rr.ms;
    }
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// pipeline.cpp (default; FGX_PIPE_SUBSET=0 opts out): the groups the device-resident entry has JUST deferred are decided by the general path from copies
// of their records alone (one small device-to-host copy per group, instead of the whole batch coming back and going through the host
// entry a second time), and the merged stream — the device's records with the general path's records inserted at the deferred groups'
// places, group order kept — is assembled in the caller object's host buffer.  `dev` is the output of that fgx_process_batch_device call.
// Returns 0 (merged filled: data on the host), or -1 when this cannot be done here (no group offsets of the last device batch; CODEC
// molecules named by the running counter) and the caller must take the whole-batch way.
extern "C++" {
namespace fgx {
int resubmit_deferred(fgx_caller* c, const uint8_t* d_blob, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec, const uint32_t* d_grp_first, uint32_t n_grp,
                      const fgx_output* dev, uint32_t n_def, const uint32_t* d_def, fgx_output* merged) {
  if (!c->fast || !c->last_group_off || n_def == 0) return -1;
  general_fn general = c->opt.caller_kind == FGX_CALLER_SIMPLEX ? simplex_process_general : c->opt.caller_kind == FGX_CALLER_DUPLEX ? duplex_process_general : codec_process_general;
  std::vector<uint32_t> def(n_def);
  hip_check(hipMemcpy(def.data(), d_def, (size_t)n_def * 4, hipMemcpyDeviceToHost), "D2H deferred");
  std::sort(def.begin(), def.end());
  std::vector<uint32_t> grp((size_t)n_grp + 1);
  hip_check(hipMemcpy(grp.data(), d_grp_first, ((size_t)n_grp + 1) * 4, hipMemcpyDeviceToHost), "D2H group boundaries");
  // the deferred groups' record tables and bytes.  Few deferred groups: their table rows alone; many: the whole tables in one copy each
  // (12 B per record).  A group's records lie in stream order, so its bytes are one span; spans that touch or nearly touch (runs of
  // deferred groups) come back as one copy.
  constexpr uint32_t BULK_TABLES = 64;
  constexpr uint64_t JOIN_GAP = 4096;
  std::vector<uint64_t> all_off;
  std::vector<uint32_t> all_len;
  if (n_def > BULK_TABLES) {
    all_off.resize(n_rec); all_len.resize(n_rec);
    if (n_rec) {
      hip_check(hipMemcpy(all_off.data(), d_rec_off, (size_t)n_rec * 8, hipMemcpyDeviceToHost), "D2H rec_off");
      hip_check(hipMemcpy(all_len.data(), d_rec_len, (size_t)n_rec * 4, hipMemcpyDeviceToHost), "D2H rec_len");
    }
  }
  struct Span { uint64_t lo, hi; uint32_t first_rec, n; };          // per deferred group: its byte span, its rows in s_off / s_len
  std::vector<Span> spans;
  std::vector<uint64_t> s_off, g_off;            // s_off: first the offsets in the DEVICE blob, rebased below
  std::vector<uint32_t> s_len, s_grp(1, 0), g_len;
  for (uint32_t g : def) {
    if (g >= n_grp) return -1;
    const uint32_t r0 = grp[g], r1 = grp[g + 1];
    if (r1 > n_rec || r0 > r1) return -1;
    const uint32_t n = r1 - r0;
    Span sp = {~0ull, 0, (uint32_t)s_off.size(), n};
    if (n) {
      const uint64_t* po; const uint32_t* pl;
      if (!all_off.empty()) { po = all_off.data() + r0; pl = all_len.data() + r0; }
      else {
        g_off.resize(n); g_len.resize(n);
        hip_check(hipMemcpy(g_off.data(), d_rec_off + r0, (size_t)n * 8, hipMemcpyDeviceToHost), "D2H rec_off of a deferred group");
        hip_check(hipMemcpy(g_len.data(), d_rec_len + r0, (size_t)n * 4, hipMemcpyDeviceToHost), "D2H rec_len of a deferred group");
        po = g_off.data(); pl = g_len.data();
      }
      for (uint32_t i = 0; i < n; i++) { if (po[i] < sp.lo) sp.lo = po[i]; if (po[i] + pl[i] > sp.hi) sp.hi = po[i] + pl[i]; s_off.push_back(po[i]); s_len.push_back(pl[i]); }
    }
    spans.push_back(sp);
    s_grp.push_back((uint32_t)s_off.size());
  }
  std::vector<uint8_t> blob;
  for (size_t a = 0; a < spans.size();) {
    if (!spans[a].n) { a++; continue; }
    uint64_t lo = spans[a].lo, hi = spans[a].hi;
    size_t b = a + 1;
    while (b < spans.size() && (!spans[b].n || (spans[b].lo >= lo && spans[b].lo <= hi + JOIN_GAP))) { if (spans[b].n && spans[b].hi > hi) hi = spans[b].hi; b++; }
    const size_t at = blob.size();
    blob.resize(at + (size_t)(hi - lo));
    hip_check(hipMemcpy(blob.data() + at, d_blob + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost), "D2H records of deferred groups");
    for (size_t k = a; k < b; k++) for (uint32_t i = 0; i < spans[k].n; i++) s_off[spans[k].first_rec + i] = at + (s_off[spans[k].first_rec + i] - lo);
    a = b;
  }
  blob.resize(blob.size() + 16);
  fgx_output gen;
  memset(&gen, 0, sizeof(gen));
  c->out_data.clear(); c->grp_out_end.clear();
  const int rc = run_general(c, general, blob.data(), s_off.data(), s_len.data(), (uint32_t)s_off.size(), s_grp.data(), n_def, &gen);
  if (rc != 0) return rc;
  if (c->opt.caller_kind == FGX_CALLER_CODEC && c->counter_names_used) return -1;
  // slot offsets of the deferred groups in the device's record stream (they hold nothing there), then the merge
  std::vector<uint64_t> at_dev(n_def);
  if (n_def > BULK_TABLES) {
    std::vector<uint64_t> tab((size_t)n_grp * c->last_group_stride);
    hip_check(hipMemcpy(tab.data(), c->last_group_off, tab.size() * 8, hipMemcpyDeviceToHost), "D2H group offsets");
    for (uint32_t k = 0; k < n_def; k++) at_dev[k] = tab[(size_t)c->last_group_stride * def[k]];
  } else
    for (uint32_t k = 0; k < n_def; k++) hip_check(hipMemcpy(&at_dev[k], c->last_group_off + (size_t)c->last_group_stride * def[k], 8, hipMemcpyDeviceToHost), "D2H slot offset");
  std::vector<uint8_t> m;
  m.resize(dev->data_len + c->out_data.size() + 16);
  uint64_t w = 0, dpos = 0, gprev = 0;
  for (uint32_t k = 0; k < n_def; k++) {
    const uint64_t upto = at_dev[k];
    if (upto < dpos || upto > dev->data_len) return -1;
    if (upto > dpos) hip_check(hipMemcpy(m.data() + w, dev->data + dpos, (size_t)(upto - dpos), hipMemcpyDeviceToHost), "D2H device records");
    w += upto - dpos; dpos = upto;
    const uint64_t gend = c->grp_out_end[k];
    if (gend > gprev) memcpy(m.data() + w, c->out_data.data() + gprev, (size_t)(gend - gprev));
    w += gend - gprev; gprev = gend;
  }
  if (dev->data_len > dpos) hip_check(hipMemcpy(m.data() + w, dev->data + dpos, (size_t)(dev->data_len - dpos), hipMemcpyDeviceToHost), "D2H device records");
  w += dev->data_len - dpos;
  m.resize(w);
  c->out_data.swap(m);
  memset(merged, 0, sizeof(*merged));
  merged->data = c->out_data.data(); merged->data_len = c->out_data.size(); merged->count = dev->count + gen.count;
  for (int i = 0; i < FGX_STATS_LEN; i++) merged->stats[i] = dev->stats[i] + gen.stats[i];
  merged->ms_kernels = dev->ms_kernels + gen.ms_kernels; merged->ms_host_prep = gen.ms_host_prep; merged->ms_emit = gen.ms_emit;
  return 0;
}
}  // namespace fgx
}  // extern "C++"

int fgx_group_records_device(fgx_caller* c, const fgx_group_options* g, const void* d_records, uint64_t records_len, const void* d_rec_off,
                             const void* d_rec_len, uint32_t n_rec, void* d_out_rec_off, void* d_out_rec_len, void* d_grp_first,
                             uint32_t* n_kept, uint32_t* n_grp) {
  if (!c || !g || !n_kept || !n_grp) return 1;
  c->err.clear();
  try {
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    return group_records_device(c, g, (const uint8_t*)d_records, records_len, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, n_rec,
                                (uint64_t*)d_out_rec_off, (uint32_t*)d_out_rec_len, (uint32_t*)d_grp_first, n_kept, n_grp);
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

int fgx_record_boundaries_device(fgx_caller* c, const void* d_stream, uint64_t stream_len, uint64_t start, void* d_rec_off, void* d_rec_len,
                                 uint64_t cap, uint64_t* n_rec, uint64_t* consumed) {
  if (!c || !n_rec || !consumed) return 1;
  c->err.clear();
  try {
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    return record_boundaries_device(c, (const uint8_t*)d_stream, stream_len, start, (uint64_t*)d_rec_off, (uint32_t*)d_rec_len, cap, n_rec, consumed);
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// host buffers: upload, group on the device, download (the staging buffers of fgx_process_batch are reused, so a following
// fgx_process_batch on the same records could skip its upload in a later revision)
int fgx_group_records(fgx_caller* c, const fgx_group_options* g, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                      const uint32_t* rec_len, uint32_t n_rec, uint64_t* out_rec_off, uint32_t* out_rec_len, uint32_t* grp_first,
                      uint32_t* n_kept, uint32_t* n_grp) {
  if (!c || !g || !n_kept || !n_grp) return 1;
  c->err.clear();
  try {
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    for (uint32_t r = 0; r < n_rec; r++)
      if (rec_len[r] > records_len || rec_off[r] > records_len - rec_len[r]) { c->err = "fgx_group_records: record outside the blob"; return 1; }
    c->d_in_blob.reserve(records_len + 16);
    c->d_in_off.reserve((size_t)n_rec * 8 + 8);
    c->d_in_len.reserve((size_t)n_rec * 4 + 4);
    c->d_in_grp.reserve((size_t)(n_rec + 1) * 4);
    c->d_stage.reserve((size_t)n_rec * 8 + 8);
    c->d_reads.reserve((size_t)n_rec * 4 + 4);
    hip_check(hipMemcpyAsync(c->d_in_blob.p, records, records_len, hipMemcpyHostToDevice, c->stream), "H2D blob");
    hip_check(hipMemcpyAsync(c->d_in_off.p, rec_off, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream), "H2D rec_off");
    hip_check(hipMemcpyAsync(c->d_in_len.p, rec_len, (size_t)n_rec * 4, hipMemcpyHostToDevice, c->stream), "H2D rec_len");
    int rc = group_records_device(c, g, c->d_in_blob.as<uint8_t>(), records_len, c->d_in_off.as<uint64_t>(), c->d_in_len.as<uint32_t>(), n_rec,
                                  c->d_stage.as<uint64_t>(), c->d_reads.as<uint32_t>(), c->d_in_grp.as<uint32_t>(), n_kept, n_grp);
    if (rc != 0) return rc;
    if (*n_kept) {
      hip_check(hipMemcpy(out_rec_off, c->d_stage.p, (size_t)*n_kept * 8, hipMemcpyDeviceToHost), "D2H rec_off");
      hip_check(hipMemcpy(out_rec_len, c->d_reads.p, (size_t)*n_kept * 4, hipMemcpyDeviceToHost), "D2H rec_len");
    }
    hip_check(hipMemcpy(grp_first, c->d_in_grp.p, (size_t)(*n_grp + 1) * 4, hipMemcpyDeviceToHost), "D2H grp_first");
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// ---- `fgumi filter` (filter.hip) -----------------------------------------------------------------------------------------
void fgx_filter_options_default(fgx_filter_options* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->struct_size = sizeof(fgx_filter_options);
  for (int i = 0; i < 3; i++) { o->min_reads[i] = 1; o->max_read_error_rate[i] = 0.025; o->max_base_error_rate[i] = 0.1; }
  o->max_no_call_fraction = 0.2;
  o->filter_by_template = 1;
}

// Filter::validate_parameters (src/lib/commands/filter.rs:1021-1107) + FilterConfig::new ordering asserts (filter.rs:284-318)
static bool filter_options_valid(const fgx_filter_options* f, std::string& err) {
  if (f->struct_size != sizeof(fgx_filter_options)) { err = "fgx_filter_options.struct_size mismatch"; return false; }
  for (int i = 0; i < 3; i++) {
    if (!(f->max_read_error_rate[i] >= 0.0 && f->max_read_error_rate[i] <= 1.0)) { err = "--max-read-error-rate must be between 0.0 and 1.0"; return false; }
    if (!(f->max_base_error_rate[i] >= 0.0 && f->max_base_error_rate[i] <= 1.0)) { err = "--max-base-error-rate must be between 0.0 and 1.0"; return false; }
  }
  if (!(f->max_no_call_fraction >= 0.0)) { err = "--max-no-call-fraction must be >= 0.0"; return false; }
  if (f->max_no_call_fraction >= 1.0 && f->max_no_call_fraction != (double)(long long)f->max_no_call_fraction && f->max_no_call_fraction < 9e18) {
    err = "--max-no-call-fraction >= 1.0 must be an integer (count of bases)"; return false;
  }
  if (f->min_reads[1] > f->min_reads[0] || f->min_reads[2] > f->min_reads[1]) { err = "min-reads values must be specified high to low (duplex >= AB >= BA)"; return false; }
  if (f->max_read_error_rate[1] > f->max_read_error_rate[2]) { err = "max-read-error-rate for AB must be <= BA (more stringent)"; return false; }
  if (f->max_base_error_rate[1] > f->max_base_error_rate[2]) { err = "max-base-error-rate for AB must be <= BA (more stringent)"; return false; }
  // the methylation filters (filter.rs:1110-1154)
  if (f->has_min_methylation_depth) {
    if (f->min_methylation_depth[0] < f->min_methylation_depth[1]) { err = "min-methylation-depth values must be specified high to low (duplex >= AB)"; return false; }
    if (f->min_methylation_depth[1] < f->min_methylation_depth[2]) { err = "min-methylation-depth values must be specified high to low (AB >= BA)"; return false; }
  }
  if (f->require_strand_methylation_agreement && !f->regenerate_alignment_tags) { err = "--require-strand-methylation-agreement requires --ref to identify CpG sites"; return false; }
  if (f->has_min_conversion_fraction) {
    if (!(f->min_conversion_fraction >= 0.0 && f->min_conversion_fraction <= 1.0)) { err = "--min-conversion-fraction must be between 0.0 and 1.0"; return false; }
    if (!f->regenerate_alignment_tags) { err = "--min-conversion-fraction requires --ref to identify non-CpG cytosines"; return false; }
    if (f->methylation_mode != FGX_METHYLATION_EM_SEQ && f->methylation_mode != FGX_METHYLATION_TAPS) { err = "--min-conversion-fraction requires --methylation-mode to be set"; return false; }
  }
  if (f->methylation_mode > FGX_METHYLATION_TAPS) { err = "fgx_filter_options.methylation_mode: not a FGX_METHYLATION_* value"; return false; }
  return true;
}

int fgx_filter_records_device(fgx_caller* c, const fgx_filter_options* f, void* d_records, uint64_t records_len, const void* d_rec_off,
                              const void* d_rec_len, uint32_t n_rec, fgx_filter_output* out) {
  if (!c || !f || !out) return 1;
  c->err.clear();
  try {
    if (!filter_options_valid(f, c->err)) return 1;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    if (!c->filt) c->filt = new FilterBuffers();
    return filter_records_device(c, *c->filt, f, (uint8_t*)d_records, records_len, (const uint64_t*)d_rec_off, (const uint32_t*)d_rec_len, n_rec, out);
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// The consensus records the handle's last fgx_process_batch_device left in HBM, filtered where they are: the slot table of
// the device pipeline becomes the record list, no copy and no host round trip in between.
int fgx_filter_last_output_device(fgx_caller* c, const fgx_filter_options* f, fgx_filter_output* out) {
  if (!c || !f || !out) return 1;
  c->err.clear();
  try {
    if (!filter_options_valid(f, c->err)) return 1;
    if (!c->fast || !c->fast->has_last) { c->err = "fgx_filter_last_output_device: no device-resident batch on this handle"; return 1; }
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    if (!c->filt) c->filt = new FilterBuffers();
    const FastResult L = c->fast->last;
    c->fast->has_last = false;   // single use: the records are masked and reversed IN PLACE, a second pass would re-apply both
    return filter_slots_device(c, *c->filt, f, (uint8_t*)L.d_out, L.out_len, L.d_out_off, L.d_slot_size, L.n_slots, out);
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

int fgx_filter_records(fgx_caller* c, const fgx_filter_options* f, const uint8_t* records, uint64_t records_len, const uint64_t* rec_off,
                       const uint32_t* rec_len, uint32_t n_rec, fgx_filter_output* out) {
  if (!c || !f || !out) return 1;
  c->err.clear();
  try {
    if (!filter_options_valid(f, c->err)) return 1;
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    for (uint32_t r = 0; r < n_rec; r++)
      if (rec_len[r] > records_len || rec_off[r] > records_len - rec_len[r]) { c->err = "fgx_filter_records: record outside the blob"; return 1; }
    if (!c->filt) c->filt = new FilterBuffers();
    FilterBuffers& B = *c->filt;
    B.in_blob.reserve(records_len + 64);
    B.in_off.reserve((size_t)n_rec * 8 + 8);
    B.in_len.reserve((size_t)n_rec * 4 + 4);
    if (records_len) hip_check(hipMemcpyAsync(B.in_blob.p, records, records_len, hipMemcpyHostToDevice, c->stream), "H2D records");
    if (n_rec) {
      hip_check(hipMemcpyAsync(B.in_off.p, rec_off, (size_t)n_rec * 8, hipMemcpyHostToDevice, c->stream), "H2D rec_off");
      hip_check(hipMemcpyAsync(B.in_len.p, rec_len, (size_t)n_rec * 4, hipMemcpyHostToDevice, c->stream), "H2D rec_len");
    }
    int rc = filter_records_device(c, B, f, B.in_blob.as<uint8_t>(), records_len, B.in_off.as<uint64_t>(), B.in_len.as<uint32_t>(), n_rec, out);
    if (rc != 0) return rc;
    B.pin_keep.reserve(out->data_len + 64);
    B.pin_rej.reserve(out->rejects_len + 64);
    if (out->data_len) hip_check(hipMemcpyAsync(B.pin_keep.p, out->data, out->data_len, hipMemcpyDeviceToHost, c->stream), "D2H kept records");
    if (out->rejects_len) hip_check(hipMemcpyAsync(B.pin_rej.p, out->rejects, out->rejects_len, hipMemcpyDeviceToHost, c->stream), "D2H rejected records");
    hip_check(hipStreamSynchronize(c->stream), "sync");
    out->data = B.pin_keep.as<uint8_t>();
    out->rejects = B.pin_rej.as<uint8_t>();
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

// diagnostics of the last fgx_process_batch call that deferred groups: out2 = {groups the first device pass deferred, of those the
// molecules the canonical second pass decided (FGX_DUPLEX_CANON=1)}
// Guard bands (FGX_GUARD_BAND, see DevBuf::reserve): checks the sentinel bytes around every live device buffer of the process.  Returns the number of
// buffers with a damaged band (0: clean, also when the mode is off; -1: the device could not be read) and describes the first one in `msg`.
int fgx_debug_check_guard_bands(char* msg, int msg_len) {
  if (msg && msg_len > 0) msg[0] = 0;
  const size_t G = fgx::guard_band_bytes();
  if (!G) return 0;
  fgx::GuardRegistry& R = fgx::guard_registry();
  std::lock_guard<std::mutex> lk(R.m);
  int bad = R.damaged_freed;                      // (buffers freed since the process began: checked when they were freed)
  std::string first = R.first_damage;
  for (const auto& kv : R.live) {
    const std::string d = fgx::guard_check_one((const uint8_t*)kv.first, kv.second.first, G);
    if (!d.empty()) { if (!bad) first = d + "; " + std::to_string(R.live.size()) + " buffers live"; bad++; }
  }
  if (bad && msg && msg_len > 0) snprintf(msg, (size_t)msg_len, "%s", first.c_str());
  return bad;
}
// buffers whose bands have been verified so far (when they were freed) + the live ones (0 when the mode is off): the test asserts that the mode was really on
int fgx_debug_guarded_buffers(void) {
  if (!fgx::guard_band_bytes()) return 0;
  fgx::GuardRegistry& R = fgx::guard_registry();
  std::lock_guard<std::mutex> lk(R.m);
  return (int)(R.live.size() + (size_t)R.checked_freed);
}
// the mechanism's own check: a buffer of 1000 bytes, one byte stored right behind it — returns what fgx_debug_check_guard_bands then reports (1 when the mode is on)
int fgx_debug_guard_self_test(void) {
  if (!fgx::guard_band_bytes()) return 0;
  fgx::DevBuf b;
  b.reserve(1000);
  (void)hipMemset((uint8_t*)b.p + 1000, 0, 1);
  int r;
  {
    fgx::GuardRegistry& R = fgx::guard_registry();
    std::lock_guard<std::mutex> lk(R.m);
    r = fgx::guard_check_one((const uint8_t*)b.p, 1000, fgx::guard_band_bytes()).empty() ? 0 : 1;
    R.live.erase(b.p);                           // (its damage is deliberate: not for the books)
  }
  (void)hipFree((uint8_t*)b.p - fgx::guard_band_bytes());
  b.p = nullptr; b.cap = 0;
  return r;
}
void fgx_debug_last_deferral(const fgx_caller* c, uint64_t* out2) { if (c && out2) { out2[0] = c->last_deferred_groups; out2[1] = c->last_canon_molecules; } }
// Multi-GPU, for a host that is not Python: contiguous shards of a weighted family stream with roughly equal total weight
// (weight = record bytes of the family; long-tail family sizes make equal-count shards unbalanced).  Shard k = families
// [cuts[k], cuts[k + 1]); it ends after the first family at which the running weight reaches k / world of the total — the same cuts as
// fgumi_amd/distributed.py balanced_shards (bench.py --scaling strong).  `cuts` holds world + 1 entries.  Returns 0, or 1 on bad arguments.
int fgx_balanced_shards(const uint64_t* weights, uint32_t n, uint32_t world, uint32_t* cuts) {
  if (!cuts || world == 0 || (n && !weights)) return 1;
  cuts[0] = 0;
  if (n == 0) { for (uint32_t k = 1; k <= world; k++) cuts[k] = 0; return 0; }
  // (float64 running sums, as numpy's cumsum over float64 weights: the cut positions are compared, not the sums)
  std::vector<double> acc(n);
  double run = 0.0;
  for (uint32_t i = 0; i < n; i++) { run += (double)weights[i]; acc[i] = run; }
  const double total = acc[n - 1];
  for (uint32_t k = 1; k < world; k++) {
    const double want = total * (double)k / (double)world;
    const uint32_t idx = (uint32_t)(std::lower_bound(acc.begin(), acc.end(), want) - acc.begin());   // first family whose running weight reaches `want`
    uint32_t c = idx + 1 > n ? n : idx + 1;
    if (c < cuts[k - 1]) c = cuts[k - 1];
    cuts[k] = c;
  }
  cuts[world] = n;
  for (uint32_t k = 1; k <= world; k++) if (cuts[k] < cuts[k - 1]) cuts[k] = cuts[k - 1];
  return 0;
}
// chunks of the record / column split pipeline in the last device batch (FGX_SPLIT_CHUNKS, or 8 / 4 / 1 by batch size); 0 = it did not run
uint32_t fgx_debug_last_split_chunks(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_split_chunks : 0u; }
// how the last device batch produced its records: 0 column scratch + k_emit, 1 written directly by the split pipeline, 2 directly + merge with the
// records of the families that left it
// families of the last device batch that went to the workgroup-per-family kernel because they have more than 64 records / that the split
// pipeline handed down the k_simplex_wave2 chain
uint32_t fgx_debug_last_big_families(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_big_families : 0u; }
uint32_t fgx_debug_last_meth_device(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_meth_device : 0u; }
uint32_t fgx_debug_last_deep_families(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_deep_families : 0u; }
// the split pipeline's first stage in the last device batch: out4[0] families finished by k_split_cols's packed build, [1] by its classic builds
// (k_split_finish counts both), [2] the build launched first (0 classic alone / no split pipeline, 1 packed alone, 2 packed + partner launch),
// [3] families the first stage handed to the next launch
void fgx_debug_last_split_builds(const fgx_caller* c, uint64_t* out4) {
  if (!out4) return;
  out4[0] = out4[1] = out4[2] = out4[3] = 0;
  if (c && c->fast) { const FastPath& f = c->fast->fp; out4[0] = f.last_packed_families; out4[1] = f.last_classic_families; out4[2] = f.last_split_build; out4[3] = f.last_first_stage_retries; }
}
// the launch chain of the last device batch: out2[0] kernel launches of FastPath::run_once (the scans of the library not counted), out2[1] host synchronisations in it
void fgx_debug_last_chain(const fgx_caller* c, uint32_t* out2) { if (out2) { out2[0] = (c && c->fast) ? c->fast->fp.last_launches : 0u; out2[1] = (c && c->fast) ? c->fast->fp.last_host_syncs : 0u; } }
uint32_t fgx_debug_last_routed(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_routed : 0u; }
int fgx_debug_last_direct(const fgx_caller* c) { return (c && c->fast) ? c->fast->fp.last_direct : 0; }
// 1: route everything through the general host path (parity tests of that path); 0: hybrid (default)
void fgx_set_general_only(fgx_caller* c, int on) { if (c) c->general_only = on != 0; }
// dynamic LDS bytes of the large-family launch of the family kernel (default 48 KiB)
void fgx_set_fast_lds_bytes(fgx_caller* c, uint32_t bytes) { if (c) { if (!c->fast) c->fast = new FastState(); c->fast->fp.lds_tile_bytes_large = bytes; } }

int fgx_call_columns(fgx_caller* c, const uint8_t* bases, const uint8_t* quals, uint32_t n_cols, uint32_t depth, uint8_t* out_base,
                     uint8_t* out_qual, uint32_t* out_depth, uint32_t* out_errors) {
  if (!c) return 1;
  c->err.clear();
  try {
    // Each column becomes a job of `depth` one-base source reads … laid out transposed so that one job
    // holds up to 64 columns: read i of the job = observation i of each column.
    ColumnBatch& B = c->batch;
    B.clear();
    std::vector<uint8_t> rb, rq;
    for (uint32_t j0 = 0; j0 < n_cols; j0 += 4096) {
      uint32_t w = std::min<uint32_t>(4096, n_cols - j0);
      uint32_t rd0 = (uint32_t)B.reads.size();
      for (uint32_t i = 0; i < depth; i++) {
        rb.resize(w); rq.resize(w);
        for (uint32_t k = 0; k < w; k++) { rb[k] = bases[(size_t)(j0 + k) * depth + i]; rq[k] = quals[(size_t)(j0 + k) * depth + i]; }
        B.add_read(rb.data(), rq.data(), w);
      }
      if (depth == 0) { B.add_job(rd0, 0, w); continue; }
      // n_reads == 1 would take the single-read LUT path; pad with an ignored all-'N' read to stay in the builder path
      if (depth == 1) { rb.assign(w, 'N'); rq.assign(w, 0); B.add_read(rb.data(), rq.data(), w); B.add_job(rd0, 2, w); }
      else B.add_job(rd0, depth, w);
    }
    ColParams prm{0, 0};   // no thresholds: raw call() result
    c->run_columns(B, prm);
    // raw call(): thresholds disabled, but a no-call comes back as ('N', 2) and errors = depth
    for (uint32_t j = 0; j < n_cols; j++) { out_base[j] = B.ob[j]; out_qual[j] = B.oq[j]; out_depth[j] = B.od[j]; out_errors[j] = B.oe[j]; }
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

int fgx_device_libm(fgx_caller* c, int op, const double* x, double* y, uint64_t n) {
  if (!c) return 1;
  try {
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    c->d_scratch_a.reserve(n * 8); c->d_scratch_b.reserve(n * 8);
    hip_check(hipMemcpyAsync(c->d_scratch_a.p, x, n * 8, hipMemcpyHostToDevice, c->stream), "H2D");
    launch_libm_test(c->stream, op, c->d_scratch_a.as<double>(), c->d_scratch_b.as<double>(), n);
    hip_check(hipGetLastError(), "k_libm launch");
    hip_check(hipMemcpyAsync(y, c->d_scratch_b.p, n * 8, hipMemcpyDeviceToHost, c->stream), "D2H");
    hip_check(hipStreamSynchronize(c->stream), "sync");
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

int fgx_get_table(const fgx_caller* c, int which, double* out94, uint32_t* cap) {
  if (!c) return 1;
  const ConsensusTables& t = c->h_tables.t;
  const double* src = which == 0 ? t.correct : which == 1 ? t.error_per_alt : which == 2 ? t.thresholds : t.cerr_min;
  memcpy(out94, src, 94 * sizeof(double));
  if (cap) *cap = t.cap;
  return 0;
}

// Tables without a device (host-only; used by the CPU-side tests of the table builders).
int fgx_build_tables_host(uint8_t pre, uint8_t post, int which, double* out94, uint32_t* cap, uint8_t* single94) {
  ConsensusTables t;
  build_tables(t, pre, post, 0);
  const double* src = which == 0 ? t.correct : which == 1 ? t.error_per_alt : which == 2 ? t.thresholds : t.cerr_min;
  memcpy(out94, src, 94 * sizeof(double));
  if (cap) *cap = t.cap;
  if (single94) build_single_input_quals(single94, pre, post);
  return 0;
}
// Host evaluation of the shared device/host math, for CPU-side parity tests of glibc_libm.h + consensus_math.h.
double fgx_host_libm(int op, double x) { return op == 0 ? g_exp(x) : op == 1 ? g_log(x) : op == 2 ? g_log1p(x) : g_expm1(x); }
void fgx_host_libm_array(int op, const double* x, double* y, uint64_t n) { for (uint64_t i = 0; i < n; i++) y[i] = fgx_host_libm(op, x[i]); }

// ---- synthetic reads ----------------------------------------------------------------------------
static void sim_layout(const fgx_sim_params* p, std::vector<uint64_t>& byte_off, std::vector<uint32_t>& rec_first, uint64_t* blob_len,
                       uint64_t* n_rec) {
  byte_off.resize(p->n_families);
  rec_first.resize(p->n_families);
  uint64_t off = 0, r = 0;
  for (uint32_t f = 0; f < p->n_families; f++) {
    sim::Molecule m = sim::make_molecule(*p, f);
    byte_off[f] = off;
    rec_first[f] = (uint32_t)r;
    off += 2ull * m.pairs * (sim::record_size(*p, m.mol_id) + 4);
    r += 2ull * m.pairs;
  }
  *blob_len = off;
  *n_rec = r;
}

int fgx_sim_sizes(const fgx_sim_params* p, uint64_t* blob_len, uint64_t* n_rec) {
  std::vector<uint64_t> bo;
  std::vector<uint32_t> rf;
  sim_layout(p, bo, rf, blob_len, n_rec);
  return *n_rec > 0xFFFFFFFFull ? 1 : 0;
}

int fgx_record_boundaries(const uint8_t* stream, uint64_t stream_len, uint64_t start, uint64_t* rec_off, uint32_t* rec_len, uint64_t cap,
                          uint64_t* n_rec) {
  if (!stream || !n_rec) return 1;
  uint64_t p = start, n = 0;
  while (p < stream_len) {
    if (stream_len - p < 4) { *n_rec = n; return 1; }
    const uint32_t ln = (uint32_t)stream[p] | ((uint32_t)stream[p + 1] << 8) | ((uint32_t)stream[p + 2] << 16) | ((uint32_t)stream[p + 3] << 24);
    if (ln > stream_len - p - 4) { *n_rec = n; return 1; }
    if (n < cap) { rec_off[n] = p + 4; rec_len[n] = ln; }
    n++;
    p += 4ull + ln;
  }
  *n_rec = n;
  return 0;
}

int fgx_sim_family_bytes(const fgx_sim_params* p, uint64_t* bytes_per_family) {
  if (!p || !bytes_per_family) return 1;
  std::vector<uint64_t> bo;
  std::vector<uint32_t> rf;
  uint64_t bl, nr;
  sim_layout(p, bo, rf, &bl, &nr);
  for (uint32_t f = 0; f < p->n_families; f++) bytes_per_family[f] = (f + 1 < p->n_families ? bo[f + 1] : bl) - bo[f];
  return 0;
}

int fgx_sim_generate_host(const fgx_sim_params* p, uint8_t* blob, uint64_t* rec_off, uint32_t* rec_len, uint32_t* grp_first) {
  std::vector<uint64_t> bo;
  std::vector<uint32_t> rf;
  uint64_t bl, nr;
  sim_layout(p, bo, rf, &bl, &nr);
  for (uint32_t f = 0; f < p->n_families; f++) sim::write_family(*p, f, bo[f], rf[f], blob, rec_off, rec_len, grp_first);
  grp_first[p->n_families] = (uint32_t)nr;
  return 0;
}

int fgx_sim_generate_device(fgx_caller* c, const fgx_sim_params* p, void* d_blob, void* d_rec_off, void* d_rec_len, void* d_grp_first) {
  if (!c) return 1;
  try {
    std::vector<uint64_t> bo;
    std::vector<uint32_t> rf;
    uint64_t bl, nr;
    sim_layout(p, bo, rf, &bl, &nr);
    hip_check(hipSetDevice(c->device), "hipSetDevice");
    DevBuf dbo, drf;
    dbo.reserve(bo.size() * 8 + 8); drf.reserve(rf.size() * 4 + 4);
    hip_check(hipMemcpy(dbo.p, bo.data(), bo.size() * 8, hipMemcpyHostToDevice), "H2D");
    hip_check(hipMemcpy(drf.p, rf.data(), rf.size() * 4, hipMemcpyHostToDevice), "H2D");
    launch_sim_generate(c->stream, *p, dbo.as<uint64_t>(), drf.as<uint32_t>(), (uint8_t*)d_blob, (uint64_t*)d_rec_off, (uint32_t*)d_rec_len,
                        (uint32_t*)d_grp_first);
    hip_check(hipGetLastError(), "k_sim_generate launch");
    uint32_t total = (uint32_t)nr;
    hip_check(hipMemcpyAsync((uint32_t*)d_grp_first + p->n_families, &total, 4, hipMemcpyHostToDevice, c->stream), "H2D");
    hip_check(hipStreamSynchronize(c->stream), "sync");
    dbo.free_(); drf.free_();
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

}  // extern "C"
