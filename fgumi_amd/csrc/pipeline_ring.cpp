// pipeline_ring.cpp — fgx_run_bam with SEVERAL chunks on their way into the device (FGX_PIPE_RING=1, opt-in; round 4).
//
// The same five host stages as pipeline.cpp; the device stage keeps a ring of five stream buffers, every buffer with its own stream, and
// while it works on one chunk up to four later ones are uploaded and inflated side by side (one chunk's BGZF blocks do not fill the chip:
// the inflate kernel is a latency chain per block).  Measured: 60 -> 72 M raw reads/s on a 1 M-family file (profiles/r04_experiments.md).
// NOT the default: with the runtime's default of four hardware queues per process the whole GPU suite is green on it, but with eight
// (GPU_MAX_HW_QUEUES=8) or with high-priority compute streams — i.e. as soon as the device stage's kernels really run beside the fills —
// the multi-chunk device-inflate tests find groups cut in two (bytes of the buffer's previous chunk in front of a stream); pipeline.cpp
// passes the same tests under the same settings.  The cause was not found before the round's GPU time ran out.
#include "pipeline_stages.h"

namespace {
using Pipeline = PipelineT<8>;   // (one chunk per stage + the chunks the device stage has on their way in, PipeState::NB - 1)

// what fgx_run_bam keeps between runs: the pinned chunk buffers and the device buffers (allocating 3 x ~1 GB of pinned memory takes
// longer than a whole chunk's work)
struct PipeState {
  // A ring of NB stream buffers: chunk `seq` lives in D[seq % NB].  While the device stage works on one chunk, up to NB - 1 later chunks
  // are on their way in, each on its OWN stream — the inflate kernel is a latency chain per BGZF block (bgzf_device.hip) that a single
  // chunk's blocks do not fill the chip with, so the chunks' kernels run side by side, and the uploads run under them.
  static constexpr int NB = 5;
  Pipeline P;
  fgx::DevBuf D[NB], d_raw[NB], d_blk[NB], d_off, d_len, d_koff, d_klen, d_grp, d_slots, d_dscratch, d_dmeta, d_packed, d_crcs;
  uint64_t pad[NB] = {0, 0, 0, 0, 0};   // bytes of D[i] in front of a chunk's inflated stream: room for what the chunk before it leaves over
  uint64_t fill_len[NB] = {0, 0, 0, 0, 0};   // inflated bytes of the stream D[i] holds (or is being filled with)
  hipStream_t s_in[NB] = {};          // uploads and inflates a LATER chunk while the device stage works on this one
  hipEvent_t ev_up0[NB] = {}, ev_up1[NB] = {}, ev_in[NB] = {};   // upload begins / upload done / stream inflated and checked
  uint32_t* h_status = nullptr;       // (pinned) the inflate kernels' status words, 16 words apart; behind them 64 zero bytes (the source of the status clears)
  HostBuf h_blk[NB];                  // FGX_PIPE_PINNED_TABLE=1: a chunk's block table goes through pinned memory (a truly asynchronous copy, ordered on its stream)
  uint32_t last_max_ahead = 0;        // (diagnostics) the most later chunks that were on their way at once in the last run
};

}  // namespace

namespace fgx {
void pipeline_ring_release(fgx_caller* c) {
  if (!c || !c->pipe_state_ring) return;
  PipeState* S = (PipeState*)c->pipe_state_ring;
  for (int i = 0; i < PipeState::NB; i++) {
    if (S->s_in[i]) { (void)hipStreamSynchronize(S->s_in[i]); (void)hipStreamDestroy(S->s_in[i]); }
    for (hipEvent_t e : {S->ev_up0[i], S->ev_up1[i], S->ev_in[i]}) if (e) (void)hipEventDestroy(e);
    for (auto* b : {&S->D[i], &S->d_raw[i], &S->d_blk[i]}) b->free_();
  }
  for (auto* b : {&S->d_off, &S->d_len, &S->d_koff, &S->d_klen, &S->d_grp, &S->d_slots, &S->d_dscratch, &S->d_dmeta, &S->d_packed, &S->d_crcs}) b->free_();
  if (S->h_status) (void)hipHostFree(S->h_status);
  delete S;
  c->pipe_state_ring = nullptr;
}
}  // namespace fgx

extern "C" {
// (diagnostics, tests) the most later chunks fgx_run_bam had on their way beside the one in its device stage, last run of this caller
uint32_t fgx_debug_last_chunks_ahead(const fgx_caller* c) { return (c && c->pipe_state_ring) ? ((const PipeState*)c->pipe_state_ring)->last_max_ahead : 0u; }
}  // extern "C"

namespace fgx {

// fgx_run_bam with the reference's `--rejects <file>` (src/lib/commands/simplex.rs:7-12, 260-285, 613-720): a second BAM that advertises the
// INPUT header and holds the rejected input records — the records of MI groups below --min-reads as they stand, the caller's rejects
// (overlap-corrected copies) — in batch-input order.  The caller must have been created with track_rejects.
int run_bam_rejects_ring(fgx_caller* c, const char* in_path, const char* out_path, const char* rejects_path, const uint8_t* out_header, uint64_t out_header_len,
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
    if (!c->pipe_state_ring) c->pipe_state_ring = new PipeState();
    PipeState* S = (PipeState*)c->pipe_state_ring;
    fgx::DevBuf* D = S->D;
    fgx::DevBuf &d_off = S->d_off, &d_len = S->d_len, &d_koff = S->d_koff, &d_klen = S->d_klen, &d_grp = S->d_grp;
    constexpr int NB = PipeState::NB;
    if (!S->h_status) {
      for (int i = 0; i < NB; i++) {
        fgx::hip_check(hipStreamCreateWithFlags(&S->s_in[i], hipStreamNonBlocking), "hipStreamCreate");
        for (hipEvent_t* e : {&S->ev_up0[i], &S->ev_up1[i], &S->ev_in[i]}) fgx::hip_check(hipEventCreate(e), "hipEventCreate");
      }
      fgx::hip_check(hipHostMalloc((void**)&S->h_status, 64 * NB + 64, hipHostMallocDefault), "hipHostMalloc");
      memset(S->h_status, 0, 64 * NB + 64);
    }
    // chunks on their way in beside the one in the device stage (FGX_PIPE_AHEAD = 1 .. NB - 1: a measuring knob)
    const uint64_t max_ahead = [] { const char* e = getenv("FGX_PIPE_AHEAD"); const int v = e ? atoi(e) : 0; return (uint64_t)((v >= 1 && v < NB) ? v : NB - 1); }();
    // two knobs that bring this form back to pipeline.cpp's behaviour one step at a time (for the hunt of DESIGN.md section 9 item 1):
    // FGX_PIPE_POLL_AHEAD=0 — no fill is started while the wait for the current chunk's fill lasts (fills strictly one after the other when
    // FGX_PIPE_AHEAD=1); FGX_PIPE_ONE_STREAM=1 — every fill on the same stream (the ring of buffers stays)
    const bool poll_ahead = [] { const char* e = getenv("FGX_PIPE_POLL_AHEAD"); return !(e && e[0] == '0'); }();
    const bool one_stream = [] { const char* e = getenv("FGX_PIPE_ONE_STREAM"); return e && e[0] == '1'; }();
    // FGX_PIPE_WAIT_EVENT=1: the compute stream waits for the fill's event ON THE DEVICE as well (hipStreamWaitEvent) — the first candidate
    // fix: a dependency between the two hardware queues that the host's wait alone does not create
    // FGX_PIPE_PINNED_TABLE=1: the block table is uploaded from pinned memory (the pageable vector it lives in makes the copy a staged one)
    const bool pinned_table = [] { const char* e = getenv("FGX_PIPE_PINNED_TABLE"); return e && e[0] == '1'; }();
    const bool wait_event = [] { const char* e = getenv("FGX_PIPE_WAIT_EVENT"); return e && e[0] == '1'; }();
    // Layout of D[i]: [ front pad | the chunk's inflated stream | slack ].  What a chunk leaves over (its last MI group and the
    // partial record behind it) is copied to the END of the other buffer's pad, so the next chunk's stream can be uploaded and
    // inflated to a fixed place BEFORE that length is known — on s_in[.], under this chunk's boundaries / grouping / consensus / download.
    const uint64_t FRONT_PAD = [] { const char* e = getenv("FGX_FRONT_PAD"); const long long v = e ? atoll(e) : 0; return v >= 256 ? ((uint64_t)v + 255) & ~255ull : 8ull << 20; }();   // (the variable: for the test of the widening path)
    uint64_t left_len = 0;                 // bytes the previous chunk left in front of D[cur]'s stream
    int cur = 0;
    bool header_done = false;
    double sec_h2d = 0, sec_bound = 0, sec_group = 0, sec_cons = 0, sec_d2h = 0, sec_infl = 0;
    const bool device_inflate = !(flags & FGX_RUN_HOST_INFLATE);
    const bool device_deflate = (flags & FGX_RUN_DEVICE_DEFLATE) != 0 && level == 1;
    double sec_defl = 0;
    std::vector<uint8_t> h_blob; std::vector<uint64_t> h_off; std::vector<uint32_t> h_len, h_grp;   // (only for chunks with deferred families)
    Pipeline* P = &S->P;
    P->reset();
    const bool want_rej = rejects_path != nullptr;
    if (want_rej) P->rej_path = rejects_path;
    uint64_t n_rejected = 0;
    S->last_max_ahead = 0;
    uint64_t n_filled = 0;                 // chunks 0 .. n_filled - 1 are in their buffers or on their way (chunk q into D[q % NB])
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
    // FGX_RUN_HOST_INFLATE, the inflated bytes over PCIe) — queued on the buffer's own stream, ev_in[buf] marks the end
    auto launch_fill = [&](Chunk& ch, int buf, uint64_t preserve) {
      ensure_room(buf, ch.inf_len, preserve);
      S->fill_len[buf] = ch.inf_len;
      uint8_t* dst = (uint8_t*)D[buf].p + S->pad[buf];
      hipStream_t si = S->s_in[one_stream ? 0 : buf];
      uint32_t* const h_status = S->h_status + 16 * buf;
      fgx::DevBuf &d_raw = S->d_raw[buf], &d_blk = S->d_blk[buf];
      fgx::hip_check(hipEventRecord(S->ev_up0[buf], si), "hipEventRecord");
      if (device_inflate) {
        const size_t blk_bytes = ch.dev_blocks.size() * sizeof(fgx::BgzfDevBlock);
        d_raw.reserve(ch.raw_len + 64);
        d_blk.reserve(blk_bytes + 64 + 16);
        fgx::hip_check(hipMemcpyAsync(d_raw.p, ch.inf.p, ch.raw_len + 64, hipMemcpyHostToDevice, si), "H2D compressed chunk");
        const void* blk_src = ch.dev_blocks.data();
        if (pinned_table && blk_bytes) { S->h_blk[buf].reserve(blk_bytes, true); memcpy(S->h_blk[buf].p, ch.dev_blocks.data(), blk_bytes); blk_src = S->h_blk[buf].p; }
        if (blk_bytes) fgx::hip_check(hipMemcpyAsync(d_blk.p, blk_src, blk_bytes, hipMemcpyHostToDevice, si), "H2D block table");
        // (the device's status word is cleared by a COPY of zeros, ordered like the uploads around it, not by hipMemsetAsync: bgzf_inflate_launch's note)
        uint32_t* const d_status = (uint32_t*)((uint8_t*)d_blk.p + ((blk_bytes + 15) & ~(size_t)15));
        fgx::hip_check(hipMemcpyAsync(d_status, S->h_status + 16 * NB, 16, hipMemcpyHostToDevice, si), "H2D status clear");
        fgx::hip_check(hipEventRecord(S->ev_up1[buf], si), "hipEventRecord");
        fgx::bgzf_inflate_launch(si, d_raw.as<uint8_t>(), d_blk.as<fgx::BgzfDevBlock>(), (uint32_t)ch.dev_blocks.size(), dst, d_status, h_status, false);
      } else {
        *h_status = 0;
        if (ch.inf_len) fgx::hip_check(hipMemcpyAsync(dst, ch.inf.p, ch.inf_len, hipMemcpyHostToDevice, si), "H2D chunk");
        fgx::hip_check(hipEventRecord(S->ev_up1[buf], si), "hipEventRecord");
      }
      fgx::hip_check(hipEventRecord(S->ev_in[buf], si), "hipEventRecord");
    };
    const int rc = P->run(in_path, out_path, out_header, out_header_len, threads, level, chunk_raw_bytes ? chunk_raw_bytes : (128ull << 20), true, device_inflate,
                          [&](Chunk& ch, uint64_t seq) {
      fgx::hip_check(hipSetDevice(c->device), "hipSetDevice");
      ch.out_len = 0; ch.packed_len = 0; ch.precompressed = false; ch.have_crcs = false;
      // ---- this chunk's stream: started while an earlier chunk was worked on, or now ----
      cur = (int)(seq % NB);
      if (n_filled <= seq) { launch_fill(ch, cur, left_len); n_filled = seq + 1; }
      // ---- the next chunks, as soon as the host stages have them ready: upload + inflate under everything below ----
      auto try_ahead = [&] {
        while (!ch.last && n_filled <= seq + max_ahead && P->staged(n_filled)) {
          Chunk& nx = P->chunks[n_filled % Pipeline::N_CHUNKS];
          launch_fill(nx, (int)(n_filled % NB), 0);
          n_filled++;
          if (n_filled - 1 - seq > S->last_max_ahead) S->last_max_ahead = (uint32_t)(n_filled - 1 - seq);
          if (nx.last) break;
        }
      };
      // (the wait for this chunk's stream polls: a chunk the host stages deliver meanwhile starts at once, not when this wait ends)
      for (;;) {
        const hipError_t q = hipEventQuery(S->ev_in[cur]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) fgx::hip_check(q, "hipEventQuery");
        if (poll_ahead) try_ahead();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      // (hipEventQuery says the fill has finished; hipEventSynchronize is what the kernels of the device stage — another stream, another
      // hardware queue, other XCDs — may rely on for SEEING what it wrote: with eight hardware queues the multi-chunk tests read stale
      // bytes of the buffer's previous chunk after the query alone, the round-3 form with the synchronize never did)
      fgx::hip_check(hipEventSynchronize(S->ev_in[cur]), "hipEventSynchronize");
      if (wait_event) fgx::hip_check(hipStreamWaitEvent(s, S->ev_in[cur], 0), "hipStreamWaitEvent");
      {
        // (with several chunks on their way these are the chunk's OWN upload and inflate times: they overlap one another's)
        float ms_up = 0, ms_in = 0;
        fgx::hip_check(hipEventElapsedTime(&ms_up, S->ev_up0[cur], S->ev_up1[cur]), "hipEventElapsedTime");
        fgx::hip_check(hipEventElapsedTime(&ms_in, S->ev_up1[cur], S->ev_in[cur]), "hipEventElapsedTime");
        sec_h2d += ms_up * 1e-3;
        if (device_inflate) sec_infl += ms_in * 1e-3;
      }
      if (fgx::bgzf_inflate_status(c, S->h_status[16 * cur]) != 0) throw std::runtime_error(c->err);
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
        // --rejects: the device entry serves the simplex caller through its side kernels; what it refuses (a group outside their scope: more
        // than 128 records, malformed records; the duplex / CODEC callers; FGX_REJECTS_DEVICE=0) goes through the host entry in one piece
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
      // ---- what stays behind moves in front of the next buffer's stream ----
      const uint64_t keep = total - batch_end;
      const int nb = (int)((seq + 1) % NB);
      if (!ch.last) {
        try_ahead();
        if (n_filled > seq + 1) {                                // (the next chunk is in D[nb] or on its way)
          if (keep > S->pad[nb]) {                               // (an enormous last group: wait for the stream, move it behind a wider pad)
            fgx::hip_check(hipEventSynchronize(S->ev_in[nb]), "hipEventSynchronize");
            widen_pad(nb, keep, S->fill_len[nb], S->fill_len[nb]);   // (the stream stays in place, its events have fired)
          }
        } else {
          if (!S->pad[nb] || !D[nb].cap) S->pad[nb] = FRONT_PAD;
          if (keep > S->pad[nb]) widen_pad(nb, keep, 0, ch.inf_len);
          else ensure_room(nb, ch.inf_len, 0);
        }
        if (keep) fgx::hip_check(hipMemcpyAsync((uint8_t*)D[nb].p + S->pad[nb] - keep, base + batch_end, keep, hipMemcpyDeviceToDevice, s), "D2D leftover");
      }
      fgx::hip_check(hipStreamSynchronize(s), "sync");
      left_len = keep;
      st->chunks = seq + 1;
    });
    for (int i = 0; i < NB; i++) (void)hipStreamSynchronize(S->s_in[i]);   // (a failed run may leave later chunks' uploads in flight)
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


}  // namespace fgx
