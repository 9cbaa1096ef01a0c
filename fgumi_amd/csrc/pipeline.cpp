// pipeline.cpp — a BAM file in, a consensus BAM file out: the container work on BOTH sides of the device path as one streaming
// pipeline.  What it stands in for, for this path: the reader / decompress / find-boundaries / group /
// process / compress / write steps of the reference's unified pipeline (src/lib/unified_pipeline/bam.rs; BGZF framing
// crates/fgumi-bgzf/src/{reader,writer}.rs; FindBoundaries bam.rs:193-260; MiGrouper src/lib/mi_group.rs:227-310), as five stages
// over a ring of chunks:
//
//   read      the file, RAW_CHUNK compressed bytes at a time, cut at the last whole BGZF block (the BSIZE chain)
//   inflate   every block of the chunk in parallel on the worker pool (zlib raw inflate, one z_stream per worker, CRC32 / ISIZE
//             checked) straight into a PINNED buffer that is reused chunk after chunk (no page faults, full-rate DMA)
//   device    upload behind what the previous chunk left over, record boundaries (boundaries.hip), MI grouping (grouping.hip), the
//             consensus batch of every group but the last (it may continue in the next chunk; its bytes move to the front of the other
//             device buffer), records back into a pinned buffer
//   deflate   the consensus records, 0xff00 bytes per block, in parallel on the same pool (level 1: the reference's default)
//   write     header blocks, the chunks' blocks in order, the EOF marker
//
// Each stage is a thread; chunk s enters a stage when the stage before has finished it, and the reader reuses a chunk's buffers when
// the writer is done with them.  The stages of different chunks overlap: the file is read and inflated while the device works on the
// previous chunk and the pool compresses the one before.
#include "pipeline_stages.h"

namespace {
using Pipeline = PipelineT<3>;

thread_local std::string t_perr;

// what fgx_run_bam keeps between runs: the pinned chunk buffers and the device buffers (allocating 3 x ~1 GB of pinned memory takes
// longer than a whole chunk's work)
struct PipeState {
  Pipeline P;
  fgx::DevBuf D[2], d_off, d_len, d_koff, d_klen, d_grp, d_raw, d_blk, d_ent, d_slots, d_dscratch, d_dmeta, d_packed, d_crcs;   // (d_ent: the entry lists of the two-phase inflate)
  uint64_t pad[2] = {0, 0};          // bytes of D[i] in front of a chunk's inflated stream: room for what the chunk before it leaves over
  hipStream_t s_in = nullptr;        // uploads and inflates the NEXT chunk while the device stage works on this one
  hipEvent_t ev_up0 = nullptr, ev_up1 = nullptr, ev_in = nullptr;   // upload begins / upload done / stream inflated and checked
  uint32_t* h_status = nullptr;      // (pinned) the inflate kernels' status word
};

}  // namespace

namespace fgx {
void pipeline_release(fgx_caller* c) {
  if (!c || !c->pipe_state) return;
  PipeState* S = (PipeState*)c->pipe_state;
  for (auto* b : {&S->D[0], &S->D[1], &S->d_off, &S->d_len, &S->d_koff, &S->d_klen, &S->d_grp, &S->d_raw, &S->d_blk, &S->d_ent, &S->d_slots, &S->d_dscratch, &S->d_dmeta, &S->d_packed, &S->d_crcs}) b->free_();
  if (S->s_in) { (void)hipStreamSynchronize(S->s_in); (void)hipStreamDestroy(S->s_in); }
  for (hipEvent_t e : {S->ev_up0, S->ev_up1, S->ev_in}) if (e) (void)hipEventDestroy(e);
  if (S->h_status) (void)hipHostFree(S->h_status);
  delete S;
  c->pipe_state = nullptr;
}
}  // namespace fgx

extern "C" {

// reader / inflate / deflate / writer around an identity middle stage: re-blocks a BGZF file (no device needed)
int fgx_bgzf_recompress_file(const char* in_path, const char* out_path, uint32_t threads, int level, uint64_t chunk_raw_bytes, uint64_t* inflated_bytes) {
  if (!in_path || !out_path) { t_perr = "fgx_bgzf_recompress_file: null argument"; return 1; }
  auto P = std::make_unique<Pipeline>();
  const int rc = P->run(in_path, out_path, nullptr, 0, threads, level, chunk_raw_bytes ? chunk_raw_bytes : (64ull << 20), false, false, [&](Chunk& c, uint64_t) {
    c.out.reserve(c.inf_len + 64, false);
    memcpy(c.out.p, c.inf.p, c.inf_len);
    c.out_len = c.inf_len;
  });
  if (inflated_bytes) *inflated_bytes = P->inflated_bytes;
  if (rc != 0) t_perr = P->err;
  return rc;
}
const char* fgx_pipeline_last_error(void) { return t_perr.c_str(); }

// The device's BGZF inflate (+ CRC-32 check) alone, for measurements and tests: the whole blocks of raw[0 .. raw_len) are uploaded once and
// inflated `reps` times; *ms = average device time of one pass (HIP events), *inflated_len = bytes produced.  When `out` is given
// (inflated_cap bytes) the stream of the last pass is copied there.  Returns 0, or non-zero with fgx_last_error(c).
int fgx_bgzf_inflate_device_bench(fgx_caller* c, const uint8_t* raw, uint64_t raw_len, uint32_t reps, double* ms, uint64_t* inflated_len, uint8_t* out,
                                  uint64_t inflated_cap) {
  if (!c || !raw || !ms || !inflated_len) return 1;
  c->err.clear();
  try {
    fgx::hip_check(hipSetDevice(c->device), "hipSetDevice");
    std::vector<Block> blocks;
    uint64_t infl = 0;
    std::string e;
    const size_t used = block_table(raw, raw_len, blocks, &infl, &e);
    if (used == (size_t)-1) { c->err = e; return 1; }
    std::vector<fgx::BgzfDevBlock> dev(blocks.size());
    for (size_t i = 0; i < blocks.size(); i++) {
      const Block& b = blocks[i];
      const uint32_t xlen = raw[b.in_off + 10] | (raw[b.in_off + 11] << 8);
      fgx::BgzfDevBlock d;
      d.in_off = b.in_off + 12 + xlen; d.out_off = b.out_off; d.in_len = b.in_size - 12 - xlen - 8; d.isize = b.isize;
      memcpy(&d.crc, raw + b.in_off + b.in_size - 8, 4);
      d.ent_off = 0;
      dev[i] = d;
    }
    // everything this entry allocates is released on every way out (a hip_check that throws included)
    struct Scope {
      fgx::DevBuf d_raw, d_blk, d_out, d_ent;
      uint32_t* h_status = nullptr;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      ~Scope() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (h_status) (void)hipHostFree(h_status);
        d_raw.free_(); d_blk.free_(); d_out.free_(); d_ent.free_();
      }
    } R;
    fgx::DevBuf &d_raw = R.d_raw, &d_blk = R.d_blk, &d_out = R.d_out;
    d_raw.reserve(used + 64); d_blk.reserve(dev.size() * sizeof(fgx::BgzfDevBlock) + 64); d_out.reserve(infl + 256);
    size_t ent_bytes = fgx::bgzf_inflate_two_phase() ? fgx::bgzf_inflate_plan(dev.data(), (uint32_t)dev.size()) : 0;
    if (ent_bytes) { try { R.d_ent.reserve(ent_bytes); } catch (const std::exception&) { R.d_ent.free_(); ent_bytes = 0; (void)hipGetLastError(); } }   // (no room for the lists: the one-phase kernel)
    fgx::hip_check(hipHostMalloc((void**)&R.h_status, 64, hipHostMallocDefault), "hipHostMalloc");
    uint32_t* const h_status = R.h_status;
    hipStream_t s = c->stream;
    fgx::hip_check(hipMemcpyAsync(d_raw.p, raw, used, hipMemcpyHostToDevice, s), "H2D");
    fgx::hip_check(hipMemsetAsync((uint8_t*)d_raw.p + used, 0, 64, s), "memset");
    const size_t blk_bytes = dev.size() * sizeof(fgx::BgzfDevBlock);
    if (blk_bytes) fgx::hip_check(hipMemcpyAsync(d_blk.p, dev.data(), blk_bytes, hipMemcpyHostToDevice, s), "H2D");
    uint32_t* d_status = (uint32_t*)((uint8_t*)d_blk.p + ((blk_bytes + 15) & ~(size_t)15));
    fgx::hip_check(hipEventCreate(&R.e0), "event"); fgx::hip_check(hipEventCreate(&R.e1), "event");
    const hipEvent_t e0 = R.e0, e1 = R.e1;
    int rc = 0;
    fgx::bgzf_inflate_launch(s, d_raw.as<uint8_t>(), d_blk.as<fgx::BgzfDevBlock>(), (uint32_t)dev.size(), d_out.as<uint8_t>(), d_status, h_status, ent_bytes ? R.d_ent.p : nullptr, ent_bytes);   // warm-up
    fgx::hip_check(hipStreamSynchronize(s), "sync");
    if (fgx::bgzf_inflate_status(c, *h_status) != 0) rc = 1;
    fgx::hip_check(hipEventRecord(e0, s), "event");
    for (uint32_t r = 0; r < (reps ? reps : 1u) && rc == 0; r++)
      fgx::bgzf_inflate_launch(s, d_raw.as<uint8_t>(), d_blk.as<fgx::BgzfDevBlock>(), (uint32_t)dev.size(), d_out.as<uint8_t>(), d_status, h_status, ent_bytes ? R.d_ent.p : nullptr, ent_bytes);
    fgx::hip_check(hipEventRecord(e1, s), "event");
    fgx::hip_check(hipStreamSynchronize(s), "sync");
    if (rc == 0 && fgx::bgzf_inflate_status(c, *h_status) != 0) rc = 1;
    float t = 0;
    fgx::hip_check(hipEventElapsedTime(&t, e0, e1), "elapsed");
    *ms = (double)t / (double)(reps ? reps : 1u);
    *inflated_len = infl;
    if (rc == 0 && out && inflated_cap >= infl && infl) fgx::hip_check(hipMemcpy(out, d_out.p, infl, hipMemcpyDeviceToHost), "D2H");
    return rc;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

int fgx_run_bam(fgx_caller* c, const char* in_path, const char* out_path, const uint8_t* out_header, uint64_t out_header_len,
                const fgx_group_options* g, uint32_t threads, int level, uint64_t chunk_raw_bytes, uint32_t flags, fgx_bam_run_stats* st) {
  return fgx_run_bam_rejects(c, in_path, out_path, nullptr, out_header, out_header_len, g, threads, level, chunk_raw_bytes, flags, st, nullptr);
}

// fgx_run_bam with the reference's `--rejects <file>` (src/lib/commands/simplex.rs:7-12, 260-285, 613-720): a second BAM that advertises the
// INPUT header and holds the rejected input records — the records of MI groups below --min-reads as they stand, the caller's rejects
// (overlap-corrected copies) — in batch-input order.  The caller must have been created with track_rejects.
int fgx_run_bam_rejects(fgx_caller* c, const char* in_path, const char* out_path, const char* rejects_path, const uint8_t* out_header, uint64_t out_header_len,
                        const fgx_group_options* g, uint32_t threads, int level, uint64_t chunk_raw_bytes, uint32_t flags, fgx_bam_run_stats* st,
                        uint64_t* rejected_records) {
  if (!c || !in_path || !out_path || !g || !st) return 1;
  if (rejects_path && !c->opt.track_rejects) { c->err = "fgx_run_bam_rejects: the caller was not created with track_rejects"; return 1; }
  if (rejected_records) *rejected_records = 0;
  c->err.clear();
  memset(st, 0, sizeof(*st));
  const auto t_begin = Clock::now();
  try {
    fgx::hip_check(hipSetDevice(c->device), "hipSetDevice");
    hipStream_t s = c->stream;
    if (!c->pipe_state) c->pipe_state = new PipeState();
    PipeState* S = (PipeState*)c->pipe_state;
    fgx::DevBuf* D = S->D;
    fgx::DevBuf &d_off = S->d_off, &d_len = S->d_len, &d_koff = S->d_koff, &d_klen = S->d_klen, &d_grp = S->d_grp;
    if (!S->s_in) {
      fgx::hip_check(hipStreamCreateWithFlags(&S->s_in, hipStreamNonBlocking), "hipStreamCreate");
      for (hipEvent_t* e : {&S->ev_up0, &S->ev_up1, &S->ev_in}) fgx::hip_check(hipEventCreate(e), "hipEventCreate");
      fgx::hip_check(hipHostMalloc((void**)&S->h_status, 64, hipHostMallocDefault), "hipHostMalloc");
    }
    // Layout of D[i]: [ front pad | the chunk's inflated stream | slack ].  What a chunk leaves over (its last MI group and the
    // partial record behind it) is copied to the END of the other buffer's pad, so the next chunk's stream can be uploaded and
    // inflated to a fixed place BEFORE that length is known — on s_in, under this chunk's boundaries / grouping / consensus / download.
    const uint64_t FRONT_PAD = [] { const char* e = getenv("FGX_FRONT_PAD"); const long long v = e ? atoll(e) : 0; return v >= 256 ? ((uint64_t)v + 255) & ~255ull : 8ull << 20; }();   // (the variable: for the test of the widening path)
    uint64_t left_len = 0;                 // bytes the previous chunk left in front of D[cur]'s stream
    int cur = 0;
    bool header_done = false;
    double sec_h2d = 0, sec_bound = 0, sec_group = 0, sec_cons = 0, sec_d2h = 0, sec_infl = 0;
    const bool device_inflate = !(flags & FGX_RUN_HOST_INFLATE);
    const bool device_deflate = (flags & FGX_RUN_DEVICE_DEFLATE) != 0 && level == 1;
    double sec_defl = 0;
    fgx::DevBuf &d_raw = S->d_raw, &d_blk = S->d_blk;
    std::vector<uint8_t> h_blob; std::vector<uint64_t> h_off; std::vector<uint32_t> h_len, h_grp;   // (only for chunks with deferred families)
    Pipeline* P = &S->P;
    P->reset();
    const bool want_rej = rejects_path != nullptr;
    if (want_rej) P->rej_path = rejects_path;
    uint64_t n_rejected = 0;
    bool ahead = false;                    // the chunk after the one in the device stage is already on its way into D[cur ^ 1]
    uint64_t ahead_seq = 0, ahead_inf_len = 0;
    // room for a stream of inf_len bytes behind the pad of D[buf]; `preserve` bytes at the end of the pad survive a regrowth
    auto ensure_room = [&](int buf, uint64_t inf_len, uint64_t preserve) {
      if (!S->pad[buf] || !D[buf].cap) S->pad[buf] = FRONT_PAD;
      if (S->pad[buf] + inf_len + 64 <= D[buf].cap) return;
      fgx::DevBuf bigger;
      bigger.reserve(S->pad[buf] + inf_len + inf_len / 4 + 64);
      if (preserve) fgx::hip_check(hipMemcpy((uint8_t*)bigger.p + S->pad[buf] - preserve, (const uint8_t*)D[buf].p + S->pad[buf] - preserve, preserve, hipMemcpyDeviceToDevice), "D2D leftover");
      D[buf].free_();
      D[buf] = bigger;
    };
    // a wider pad for D[buf] (a leftover larger than the pad: one enormous MI group); `stream_len` bytes of stream are kept
    auto widen_pad = [&](int buf, uint64_t keep, uint64_t stream_len, uint64_t next_len) {
      const uint64_t new_pad = (keep + keep / 4 + 255) & ~255ull;
      fgx::DevBuf bigger;
      bigger.reserve(new_pad + next_len + next_len / 4 + 64);
      if (stream_len) fgx::hip_check(hipMemcpy((uint8_t*)bigger.p + new_pad, (const uint8_t*)D[buf].p + S->pad[buf], stream_len, hipMemcpyDeviceToDevice), "D2D stream");
      D[buf].free_();
      D[buf] = bigger;
      S->pad[buf] = new_pad;
    };
    // chunk `ch` into D[buf]: the compressed bytes and block descriptors over PCIe, DEFLATE + CRC-32 on the device (or, with
    // FGX_RUN_HOST_INFLATE, the inflated bytes over PCIe) — queued on s_in, ev_in marks the end
    auto launch_fill = [&](Chunk& ch, int buf, uint64_t preserve) {
      ensure_room(buf, ch.inf_len, preserve);
      uint8_t* dst = (uint8_t*)D[buf].p + S->pad[buf];
      hipStream_t si = S->s_in;
      fgx::hip_check(hipEventRecord(S->ev_up0, si), "hipEventRecord");
      if (device_inflate) {
        const size_t blk_bytes = ch.dev_blocks.size() * sizeof(fgx::BgzfDevBlock);
        d_raw.reserve(ch.raw_len + 64);
        d_blk.reserve(blk_bytes + 64 + 16);
        fgx::hip_check(hipMemcpyAsync(d_raw.p, ch.inf.p, ch.raw_len + 64, hipMemcpyHostToDevice, si), "H2D compressed chunk");
        // the entry lists of the two-phase inflate: sized block by block from ISIZE; without room for them (or beyond 32-bit offsets) the one-phase kernel
        size_t ent_bytes = fgx::bgzf_inflate_two_phase() ? fgx::bgzf_inflate_plan(ch.dev_blocks.data(), (uint32_t)ch.dev_blocks.size()) : 0;
        if (ent_bytes) { try { S->d_ent.reserve(ent_bytes); } catch (const std::exception&) { S->d_ent.free_(); ent_bytes = 0; (void)hipGetLastError(); } }
        if (blk_bytes) fgx::hip_check(hipMemcpyAsync(d_blk.p, ch.dev_blocks.data(), blk_bytes, hipMemcpyHostToDevice, si), "H2D block table");
        fgx::hip_check(hipEventRecord(S->ev_up1, si), "hipEventRecord");
        fgx::bgzf_inflate_launch(si, d_raw.as<uint8_t>(), d_blk.as<fgx::BgzfDevBlock>(), (uint32_t)ch.dev_blocks.size(), dst,
                                 (uint32_t*)((uint8_t*)d_blk.p + ((blk_bytes + 15) & ~(size_t)15)), S->h_status, ent_bytes ? S->d_ent.p : nullptr, ent_bytes);
      } else {
        *S->h_status = 0;
        if (ch.inf_len) fgx::hip_check(hipMemcpyAsync(dst, ch.inf.p, ch.inf_len, hipMemcpyHostToDevice, si), "H2D chunk");
        fgx::hip_check(hipEventRecord(S->ev_up1, si), "hipEventRecord");
      }
      fgx::hip_check(hipEventRecord(S->ev_in, si), "hipEventRecord");
    };
    const int rc = P->run(in_path, out_path, out_header, out_header_len, threads, level, chunk_raw_bytes ? chunk_raw_bytes : (512ull << 20), true, device_inflate,
                          [&](Chunk& ch, uint64_t seq) {
      fgx::hip_check(hipSetDevice(c->device), "hipSetDevice");
      ch.out_len = 0; ch.packed_len = 0; ch.precompressed = false; ch.have_crcs = false;
      // ---- this chunk's stream: started while the chunk before was worked on, or now ----
      if (!(ahead && ahead_seq == seq)) launch_fill(ch, cur, left_len);
      ahead = false;
      fgx::hip_check(hipEventSynchronize(S->ev_in), "hipEventSynchronize");
      {
        float ms_up = 0, ms_in = 0;
        fgx::hip_check(hipEventElapsedTime(&ms_up, S->ev_up0, S->ev_up1), "hipEventElapsedTime");
        fgx::hip_check(hipEventElapsedTime(&ms_in, S->ev_up1, S->ev_in), "hipEventElapsedTime");
        sec_h2d += ms_up * 1e-3;
        if (device_inflate) sec_infl += ms_in * 1e-3;
      }
      if (fgx::bgzf_inflate_status(c, *S->h_status) != 0) throw std::runtime_error(c->err);
      uint64_t h = 0;
      if (!header_done) {
        h = device_inflate ? ch.header_size : bam_header_size(ch.inf.p, ch.inf_len);
        if (h == 0) {
          if (ch.last && ch.inf_len == 0) return;              // an empty file
          throw std::runtime_error("the first chunk does not hold the whole BAM header (not a BAM file, or chunk_raw_bytes too small)");
        }
        header_done = true;                                    // (the header is uploaded / inflated with the rest and skipped by offset)
      }
      // the stream the kernels see starts at a 256-byte boundary at or before the leftover; `start` skips what lies in between
      const uint64_t lead = S->pad[cur] - left_len, base_off = lead & ~255ull;
      uint8_t* const base = (uint8_t*)D[cur].p + base_off;
      const uint64_t start = (lead - base_off) + h;
      const uint64_t total = (lead - base_off) + left_len + ch.inf_len;
      ch.rej_len = 0;
      if (want_rej && h) {                                     // the rejects file advertises the input's own header: its bytes as the stream holds them
        P->rej_header.resize(h);
        fgx::hip_check(hipMemcpy(P->rej_header.data(), base + (lead - base_off), h, hipMemcpyDeviceToHost), "D2H header");
      }
      // ---- the next chunk, if the host stages have it ready: upload + inflate under everything below ----
      auto try_ahead = [&] {
        if (ahead || ch.last || !P->staged(seq + 1)) return;
        Chunk& nx = P->chunks[(seq + 1) % Pipeline::N_CHUNKS];
        launch_fill(nx, cur ^ 1, 0);
        ahead = true; ahead_seq = seq + 1; ahead_inf_len = nx.inf_len;
      };
      try_ahead();
      auto t0 = Clock::now();
      // ---- record boundaries ----
      t0 = Clock::now();
      uint64_t n_rec = 0, consumed = 0;
      const uint64_t cap_guess = total / 64 + 16;              // (a record is at least 36 bytes; typical libraries: 200 - 400)
      d_off.reserve(cap_guess * 8); d_len.reserve(cap_guess * 4);
      int brc = fgx::record_boundaries_device(c, base, total, start, d_off.as<uint64_t>(), d_len.as<uint32_t>(), cap_guess, &n_rec, &consumed);
      if (brc == 2) {
        d_off.reserve(n_rec * 8 + 64); d_len.reserve(n_rec * 4 + 64);
        brc = fgx::record_boundaries_device(c, base, total, start, d_off.as<uint64_t>(), d_len.as<uint32_t>(), n_rec, &n_rec, &consumed);
      }
      if (brc != 0) throw std::runtime_error(c->err);
      st->boundary_repair_rounds += c->last_boundary_rounds;
      sec_bound += since(t0);
      if (ch.last && consumed != total) throw std::runtime_error("the BAM stream ends inside a record");
      if (n_rec > 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 records in one chunk: lower chunk_raw_bytes");
      // ---- MI groups ----
      t0 = Clock::now();
      uint32_t n_kept = 0, n_grp = 0;
      d_koff.reserve(n_rec * 8 + 64); d_klen.reserve(n_rec * 4 + 64); d_grp.reserve((n_rec + 2) * 4);
      if (n_rec) {
        const int grc = fgx::group_records_device(c, g, base, total, d_off.as<uint64_t>(), d_len.as<uint32_t>(), (uint32_t)n_rec,
                                                  d_koff.as<uint64_t>(), d_klen.as<uint32_t>(), d_grp.as<uint32_t>(), &n_kept, &n_grp);
        if (grc != 0) throw std::runtime_error(c->err);
      }
      sec_group += since(t0);
      // the last group may continue in the next chunk: it stays behind (with whatever follows it), unless this is the end of the file
      uint32_t batch_grp = n_grp, batch_rec = n_kept;
      uint64_t batch_end = consumed;                           // bytes of D[cur] the batch covers
      if (!ch.last) {
        if (n_grp >= 1) {
          uint32_t first_of_last = 0;
          fgx::hip_check(hipMemcpy(&first_of_last, d_grp.as<uint32_t>() + (n_grp - 1), 4, hipMemcpyDeviceToHost), "D2H");
          uint64_t o = 0;
          fgx::hip_check(hipMemcpy(&o, d_koff.as<uint64_t>() + first_of_last, 8, hipMemcpyDeviceToHost), "D2H");
          batch_grp = n_grp - 1; batch_rec = first_of_last; batch_end = o - 4;
        } else { batch_grp = 0; batch_rec = 0; batch_end = consumed; }   // (nothing kept: only the partial record behind `consumed` is pending — not the header, not the alignment bytes in front)
      }
      // ---- consensus ----
      t0 = Clock::now();
      if (batch_grp) {
        fgx_output out;
        memset(&out, 0, sizeof(out));
        uint32_t n_def = 0;
        const void* d_def = nullptr;
        int prc = fgx_process_batch_device(c, base, batch_end, d_koff.p, d_klen.p, batch_rec, d_grp.p, batch_grp, &out, &n_def, &d_def);
        // --rejects: the device entry serves the callers through its side kernels (duplex / CODEC since round 6); what it refuses (a group outside their
        // scope: more than 128 records, malformed records; a duplex / CODEC batch with deferred molecules; FGX_REJECTS_DEVICE=0) goes through the host entry in one piece
        // (and the methylation-aware mode, whose annotation runs on the general path: every batch of such a caller)
        const bool host_whole = prc == 1 && (c->opt.track_rejects || c->opt.methylation_mode != FGX_METHYLATION_DISABLED);
        if (prc != 0 && !host_whole) throw std::runtime_error(c->err);
        const void* const rej_dev = host_whole ? nullptr : out.rejects;
        const uint64_t rej_dev_len = host_whole ? 0 : out.rejects_len, rej_dev_n = host_whole ? 0 : out.n_rejects;
        if (host_whole) n_def = batch_grp;
        if (!host_whole && n_def == 0 && device_deflate) {
          // the records are cut into BGZF blocks and compressed where they lie; an eighth of the bytes comes back
          sec_cons += since(t0);
          t0 = Clock::now();
          uint64_t plen = 0;
          if (fgx::bgzf_deflate_device(c, (const uint8_t*)out.data, out.data_len, S->d_slots, S->d_dscratch, S->d_dmeta, S->d_packed, &plen) != 0) throw std::runtime_error(c->err);
          sec_defl += since(t0);
          t0 = Clock::now();
          ch.packed.reserve(plen + 64, true);
          if (plen) fgx::hip_check(hipMemcpy(ch.packed.p, S->d_packed.p, plen, hipMemcpyDeviceToHost), "D2H blocks");
          ch.packed_len = plen; ch.out_len = out.data_len; ch.precompressed = true;
          sec_d2h += since(t0);
        } else if (!host_whole && n_def == 0) {
          sec_cons += since(t0);
          t0 = Clock::now();
          // the records come back into pinned memory, and with them the CRC-32 of every BGZF payload they will be cut into (a
          // wavefront per 0xff00 bytes while they are still in HBM: the host's deflate stage is left with the compressor alone)
          ch.out.reserve(out.data_len + 64, true);
          const size_t nb = (size_t)((out.data_len + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD);
          ch.crcs.resize(nb);
          if (nb) {
            S->d_crcs.reserve(nb * 4 + 64);
            fgx::bgzf_crc_blocks_device(c, (const uint8_t*)out.data, out.data_len, S->d_crcs.as<uint32_t>());
            fgx::hip_check(hipMemcpyAsync(ch.out.p, out.data, out.data_len, hipMemcpyDeviceToHost, s), "D2H records");
            fgx::hip_check(hipMemcpyAsync(ch.crcs.data(), S->d_crcs.p, nb * 4, hipMemcpyDeviceToHost, s), "D2H block CRCs");
            fgx::hip_check(hipStreamSynchronize(s), "sync");
          }
          ch.out_len = out.data_len; ch.have_crcs = true;
          sec_d2h += since(t0);
        } else if (!host_whole && subset_enabled() && [&] {
                     // Default (FGX_PIPE_SUBSET=0 opts out): only the deferred groups come back (their records, a span per group), the general path
                     // decides them, and the merged stream is assembled on the host — the batch is not uploaded and run a second time.
                     fgx_output merged;
                     const int mrc = fgx::resubmit_deferred(c, base, d_koff.as<uint64_t>(), d_klen.as<uint32_t>(), batch_rec, d_grp.as<uint32_t>(), batch_grp, &out, n_def,
                                                            (const uint32_t*)d_def, &merged);
                     if (mrc > 0) throw std::runtime_error(c->err);
                     if (mrc < 0) return false;                          // (not possible here: the whole-batch way below)
                     st->deferred_groups += n_def;
                     if (pipe_debug()) fprintf(stderr, "fgx_run_bam: %u of %u groups deferred: decided alone (general path on their records), merged on the host\n", n_def, batch_grp);
                     ch.out.reserve(merged.data_len + 64, true);
                     if (merged.data_len) memcpy(ch.out.p, merged.data, merged.data_len);
                     ch.out_len = merged.data_len;
                     out = merged;
                     sec_cons += since(t0);
                     return true;
                   }()) {
        } else {
          // families the device pipelines do not decide: the whole batch through the host entry (it splices both paths in group order)
          st->deferred_groups += host_whole ? 0 : n_def;
          st->host_entry_batches++;
          if (pipe_debug()) fprintf(stderr, host_whole ? "fgx_run_bam: --rejects / the methylation-aware mode of this batch need the host entry: the whole batch (%u of %u groups)\n"
                                                       : "fgx_run_bam: %u of %u groups deferred: the whole batch through the host entry\n", n_def, batch_grp);
          h_blob.resize(batch_end + 16); h_off.resize(batch_rec); h_len.resize(batch_rec); h_grp.resize((size_t)batch_grp + 1);
          fgx::hip_check(hipMemcpy(h_blob.data(), base, batch_end, hipMemcpyDeviceToHost), "D2H");
          fgx::hip_check(hipMemcpy(h_off.data(), d_koff.p, (size_t)batch_rec * 8, hipMemcpyDeviceToHost), "D2H");
          fgx::hip_check(hipMemcpy(h_len.data(), d_klen.p, (size_t)batch_rec * 4, hipMemcpyDeviceToHost), "D2H");
          fgx::hip_check(hipMemcpy(h_grp.data(), d_grp.p, ((size_t)batch_grp + 1) * 4, hipMemcpyDeviceToHost), "D2H");
          memset(&out, 0, sizeof(out));
          prc = fgx_process_batch(c, h_blob.data(), batch_end, h_off.data(), h_len.data(), batch_rec, h_grp.data(), batch_grp, &out);
          if (prc != 0) throw std::runtime_error(c->err);
          ch.out.reserve(out.data_len + 64, true);
          if (out.data_len) memcpy(ch.out.p, out.data, out.data_len);
          ch.out_len = out.data_len;
          sec_cons += since(t0);
        }
        if (want_rej) {   // the batch's rejected input records: the side kernels' stream (HBM) or the host entry's
          const uint64_t rl = host_whole ? out.rejects_len : rej_dev_len;
          ch.rej.reserve(rl + 64, true);
          if (rl) {
            if (host_whole) memcpy(ch.rej.p, out.rejects, rl);
            else fgx::hip_check(hipMemcpy(ch.rej.p, rej_dev, rl, hipMemcpyDeviceToHost), "D2H rejects");
          }
          ch.rej_len = rl;
          n_rejected += host_whole ? out.n_rejects : rej_dev_n;
        }
        for (int i = 0; i < FGX_STATS_LEN; i++) st->stats[i] += out.stats[i];
        st->consensus_records += out.count;
        st->groups += batch_grp;
        st->kept_records += batch_rec;
      }
      // ---- what stays behind moves in front of the other buffer's stream ----
      const uint64_t keep = total - batch_end;
      const int nb = cur ^ 1;
      if (!ch.last) {
        try_ahead();
        if (ahead) {
          if (keep > S->pad[nb]) {                               // (an enormous last group: wait for the stream, move it behind a wider pad)
            fgx::hip_check(hipEventSynchronize(S->ev_in), "hipEventSynchronize");
            widen_pad(nb, keep, ahead_inf_len, ahead_inf_len);   // (`ahead` stays: the stream is in place, its events have fired)
          }
        } else {
          if (!S->pad[nb] || !D[nb].cap) S->pad[nb] = FRONT_PAD;
          if (keep > S->pad[nb]) widen_pad(nb, keep, 0, ch.inf_len);
          else ensure_room(nb, ch.inf_len, 0);
        }
        if (keep) fgx::hip_check(hipMemcpyAsync((uint8_t*)D[nb].p + S->pad[nb] - keep, base + batch_end, keep, hipMemcpyDeviceToDevice, s), "D2D leftover");
      }
      fgx::hip_check(hipStreamSynchronize(s), "sync");
      left_len = keep; cur ^= 1;
      st->chunks = seq + 1;
    });
    (void)hipStreamSynchronize(S->s_in);   // (a failed run may leave the next chunk's upload in flight)
    st->in_bytes = P->in_bytes; st->inflated_bytes = P->inflated_bytes; st->out_bytes = P->out_bytes; st->out_file_bytes = P->out_file_bytes;
    st->seconds_read = P->busy[0]; st->seconds_inflate = P->busy[1]; st->seconds_device = P->busy[2]; st->seconds_deflate = P->busy[3]; st->seconds_write = P->busy[4];
    st->seconds_device_inflate = sec_infl; st->device_inflate = device_inflate ? 1u : 0u;
    st->seconds_device_deflate = sec_defl; st->device_deflate = device_deflate ? 1u : 0u;
    st->seconds_h2d = sec_h2d; st->seconds_boundaries = sec_bound; st->seconds_grouping = sec_group; st->seconds_consensus = sec_cons; st->seconds_d2h = sec_d2h;
    st->seconds_total = since(t_begin);
    if (rejected_records) *rejected_records = n_rejected;
    if (rc != 0) { c->err = P->err; return 1; }
    return 0;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

}  // extern "C"
