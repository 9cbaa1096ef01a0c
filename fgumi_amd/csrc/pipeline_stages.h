// pipeline_stages.h — the host stages of fgx_run_bam (reader, inflate / staging, the middle stage's frame, deflate, writer: a thread each
// over a ring of chunks) for pipeline.cpp (one chunk ahead).  Everything has internal linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "engine.h"
#include "deflate_core.h"
#include "../../include/fgumi_amd.h"


namespace {
bool subset_enabled() { return fgx::opt_in("FGX_PIPE_SUBSET"); }
bool pipe_debug() { static const bool on = [] { const char* e = getenv("FGX_PIPE_DEBUG"); return e && e[0] == '1'; }(); return on; }

using Clock = std::chrono::steady_clock;
double since(Clock::time_point t) { return std::chrono::duration<double>(Clock::now() - t).count(); }

constexpr uint32_t BGZF_PAYLOAD = 0xFF00;
constexpr size_t BGZF_SLOT = 0x10000;

// CPUs this process may actually use: the hardware threads, capped by the cgroup's CPU quota (a container that shows 256 logical
// CPUs with `cpu.max = 1600000 100000` runs 16 cores' worth of threads; 256 workers there only queue behind the throttle)
unsigned usable_cpus() {
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    unsigned long long period = 0;
    if (fscanf(f, "%63s %llu", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const unsigned long long quota = strtoull(q, nullptr, 10);
      const unsigned c = (unsigned)((quota + period - 1) / period);
      if (c >= 1 && c < n) n = c;
    }
    fclose(f);
  }
  return n;
}

// ---- worker pool: parallel_for from several stage threads at once ---------------------------------------------------------------
class Pool {
 public:
  explicit Pool(unsigned n) {
    if (n == 0) n = 1;
    for (unsigned i = 0; i < n; i++) ts_.emplace_back([this, i] { run(i); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : ts_) t.join();
  }
  unsigned size() const { return (unsigned)ts_.size(); }
  static constexpr unsigned MAX_HELPERS = 8;   // stage threads that may be inside parallel_for at the same time
  // fn(index, worker id); worker ids are 0 .. size() + MAX_HELPERS - 1: the pool's threads, then one id PER CALLING THREAD that is
  // helping right now (two stage threads inside parallel_for at once used to share the id size(), and with it any per-worker scratch)
  void parallel_for(size_t n, size_t grain, const std::function<void(size_t, unsigned)>& fn) {
    if (n == 0) return;
    auto job = std::make_shared<Job>();
    job->n = n; job->grain = grain ? grain : 1; job->fn = &fn;
    int helper = -1;
    {
      std::lock_guard<std::mutex> l(m_);
      jobs_.push_back(job);
      for (unsigned k = 0; k < MAX_HELPERS; k++) if (!(helpers_busy_ & (1u << k))) { helpers_busy_ |= 1u << k; helper = (int)k; break; }
    }
    cv_.notify_all();
    if (helper >= 0) work(*job, size() + (unsigned)helper);   // (every helper id taken: this caller only waits)
    std::unique_lock<std::mutex> l(m_);
    job->cv.wait(l, [&] { return job->done.load() >= job->n; });
    if (helper >= 0) helpers_busy_ &= ~(1u << helper);
    for (size_t i = 0; i < jobs_.size(); i++) if (jobs_[i] == job) { jobs_.erase(jobs_.begin() + (long)i); break; }
  }

 private:
  struct Job {
    size_t n = 0, grain = 1;
    std::atomic<size_t> next{0}, done{0};
    const std::function<void(size_t, unsigned)>* fn = nullptr;
    std::condition_variable cv;
  };
  void work(Job& j, unsigned wid) {
    for (;;) {
      const size_t i0 = j.next.fetch_add(j.grain);
      if (i0 >= j.n) return;
      const size_t i1 = i0 + j.grain < j.n ? i0 + j.grain : j.n;
      for (size_t i = i0; i < i1; i++) (*j.fn)(i, wid);
      if (j.done.fetch_add(i1 - i0) + (i1 - i0) >= j.n) { std::lock_guard<std::mutex> l(m_); j.cv.notify_all(); }
    }
  }
  void run(unsigned wid) {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] {
          if (stop_) return true;
          for (auto& j : jobs_) if (j->next.load() < j->n) return true;
          return false;
        });
        if (stop_) return;
        for (auto& j : jobs_) if (j->next.load() < j->n) { job = j; break; }
      }
      if (job) work(*job, wid);
    }
  }
  std::vector<std::thread> ts_;
  std::vector<std::shared_ptr<Job>> jobs_;
  std::mutex m_;
  unsigned helpers_busy_ = 0;   // bit k: helper id size() + k is in use (under m_)
  std::condition_variable cv_;
  bool stop_ = false;
};

// ---- host buffers: pinned when a HIP device is there, plain otherwise ------------------------------------------------------------
struct HostBuf {
  uint8_t* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  void reserve(size_t n, bool want_pinned) {
    if (n <= cap) return;
    release();
    const size_t want = n + n / 8 + 4096;
    if (want_pinned && hipHostMalloc((void**)&p, want, hipHostMallocDefault) == hipSuccess && p) { pinned = true; cap = want; return; }
    (void)hipGetLastError();
    p = (uint8_t*)malloc(want);
    if (!p) throw std::runtime_error("out of host memory");
    memset(p, 0, want);                                     // (touch the pages now, not inside a timed stage)
    pinned = false; cap = want;
  }
  void release() {
    if (!p) return;
    if (pinned) (void)hipHostFree(p); else free(p);
    p = nullptr; cap = 0;
  }
  ~HostBuf() { release(); }
};

struct Block { uint64_t in_off; uint32_t in_size, isize; uint64_t out_off; };

struct Chunk {
  const uint8_t* raw = nullptr;             // compressed bytes: whole BGZF blocks of the (memory-mapped) input file
  size_t raw_len = 0;
  std::vector<Block> blocks;
  HostBuf inf;                              // inflated bytes
  uint64_t inf_len = 0;
  HostBuf out;                              // the records to write (uncompressed)
  uint64_t out_len = 0;
  HostBuf comp;                             // BGZF blocks, one 64 KiB slot each (reused: a vector would zero 64 KiB per block every chunk)
  std::vector<uint32_t> comp_size;
  HostBuf packed;                           // the chunk's BGZF blocks back to back: what the writer writes
  uint64_t packed_len = 0;
  bool last = false;                        // the file's last chunk
  // device inflate: `inf` holds the chunk's COMPRESSED bytes (staged in pinned memory), `dev_blocks` one descriptor per block
  std::vector<fgx::BgzfDevBlock> dev_blocks;
  std::vector<uint32_t> crcs;               // CRC-32 of every 0xff00-byte piece of `out`, computed on the device while the records were still there
  bool have_crcs = false;
  bool precompressed = false;               // `packed` already holds the chunk's BGZF blocks (device deflate): the deflate stage passes it on
  uint64_t header_size = 0;                 // first chunk: bytes of the BAM header at the start of the inflated stream (0 = not found)
  // --rejects: the chunk's rejected input records (block_size prefixes included, batch-input order), then their BGZF blocks
  HostBuf rej, rej_comp, rej_packed;
  std::vector<uint32_t> rej_sizes;
  uint64_t rej_len = 0, rej_packed_len = 0;
};

// parses the BSIZE chain of raw[0 .. len): whole blocks into `blocks`; returns the bytes they cover
size_t block_table(const uint8_t* raw, size_t len, std::vector<Block>& blocks, uint64_t* inflated, std::string* err) {
  blocks.clear();
  size_t p = 0;
  uint64_t total = 0;
  while (len - p >= 18) {
    if (raw[p] != 0x1F || raw[p + 1] != 0x8B || raw[p + 2] != 8 || !(raw[p + 3] & 4)) { *err = "not a BGZF block at chunk offset " + std::to_string(p); return (size_t)-1; }
    const uint32_t xlen = raw[p + 10] | (raw[p + 11] << 8);
    if (len - p < 12 + (size_t)xlen) break;
    size_t q = p + 12;
    const size_t end = p + 12 + xlen;
    uint32_t bsize = 0;
    while (q + 4 <= end) {
      const uint32_t slen = raw[q + 2] | (raw[q + 3] << 8);
      if (q + 4 + slen > end) break;
      if (raw[q] == 'B' && raw[q + 1] == 'C' && slen == 2) bsize = (uint32_t)(raw[q + 4] | (raw[q + 5] << 8)) + 1;
      q += 4 + slen;
    }
    if (bsize < 12 + xlen + 8) { *err = "BGZF block without a BC subfield at chunk offset " + std::to_string(p); return (size_t)-1; }
    if (len - p < bsize) break;                               // the block continues in the next read
    uint32_t isize;
    memcpy(&isize, raw + p + bsize - 4, 4);
    if (isize > 0x10000) { *err = "BGZF block claims more than 64 KiB"; return (size_t)-1; }
    blocks.push_back(Block{p, bsize, isize, total});
    total += isize;
    p += bsize;
  }
  *inflated = total;
  return p;
}

uint64_t bam_header_size(const uint8_t* p, uint64_t n);

// the five stages over a ring of chunks; `middle` turns chunk.inf into chunk.out (the device stage, or a copy)
template <int N_CHUNKS_T>        // chunks in the ring: one per stage + what the device stage has on its way in
struct PipelineT {
  static constexpr int N_CHUNKS = N_CHUNKS_T, N_STAGES = 5;
  Chunk chunks[N_CHUNKS];
  std::mutex m;
  std::condition_variable cv;
  uint64_t progress[N_STAGES] = {0, 0, 0, 0, 0};   // chunks each stage has finished
  uint64_t n_chunks_total = ~0ull;                 // known once the reader has seen the end of the file
  bool failed = false;
  std::string err;
  double busy[N_STAGES] = {0, 0, 0, 0, 0};
  uint64_t in_bytes = 0, inflated_bytes = 0, out_bytes = 0, out_file_bytes = 0;
  // --rejects (simplex.rs:7-12, 260-285): a second BGZF file that advertises the INPUT header and holds the rejected input records in
  // batch-input order.  `rej_header` is filled by the middle stage of the first chunk (that is where the header is first seen whole).
  std::string rej_path;
  std::vector<uint8_t> rej_header;
  uint64_t rej_bytes = 0, rej_file_bytes = 0;

  void reset() {                                     // before a run (the chunks keep their buffers)
    for (int k = 0; k < N_STAGES; k++) { progress[k] = 0; busy[k] = 0; }
    n_chunks_total = ~0ull; failed = false; err.clear();
    in_bytes = inflated_bytes = out_bytes = out_file_bytes = 0;
    rej_path.clear(); rej_header.clear(); rej_bytes = rej_file_bytes = 0;
  }
  void fail(const std::string& e) {
    std::lock_guard<std::mutex> l(m);
    if (!failed) { failed = true; err = e; }
    cv.notify_all();
  }
  // wait until chunk `s` may enter stage `k`; false = nothing more to do (or failure)
  bool enter(int k, uint64_t s) {
    std::unique_lock<std::mutex> l(m);
    cv.wait(l, [&] {
      if (failed) return true;
      if (s >= n_chunks_total) return true;
      if (k == 0) return s < progress[N_STAGES - 1] + N_CHUNKS;          // the chunk's buffers are free again
      return progress[k - 1] > s;
    });
    return !failed && s < n_chunks_total;
  }
  void leave(int k) {
    std::lock_guard<std::mutex> l(m);
    progress[k]++;
    cv.notify_all();
  }
  // has chunk `s` left the stage / inflate stage already?  (the device stage looks one chunk ahead without waiting for it)
  bool staged(uint64_t s) {
    std::lock_guard<std::mutex> l(m);
    return !failed && s < n_chunks_total && progress[1] > s;
  }

  int run(const char* in_path, const char* out_path, const uint8_t* out_header, uint64_t out_header_len, unsigned threads, int level,
          uint64_t raw_chunk, bool pinned, bool device_inflate, const std::function<void(Chunk&, uint64_t)>& middle) {
    // the input is memory-mapped: the inflate workers read the compressed blocks where the page cache holds them (reading the file
    // into a buffer first was a single-threaded copy of every byte: the slowest stage)
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + in_path; return 1; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); err = std::string("cannot stat ") + in_path; return 1; }
    const size_t file_len = (size_t)sb.st_size;
    const uint8_t* file = nullptr;
    if (file_len) {
      void* m = mmap(nullptr, file_len, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); err = std::string("cannot map ") + in_path; return 1; }
      (void)madvise(m, file_len, MADV_SEQUENTIAL);
      file = (const uint8_t*)m;
    }
    FILE* fout = fopen(out_path, "wb");
    if (!fout) { if (file) munmap((void*)file, file_len); close(fd); err = std::string("cannot create ") + out_path; return 1; }
    FILE* frej = nullptr;
    if (!rej_path.empty()) {
      frej = fopen(rej_path.c_str(), "wb");
      if (!frej) { fclose(fout); if (file) munmap((void*)file, file_len); close(fd); err = std::string("cannot create ") + rej_path; return 1; }
    }
    if (raw_chunk < (1u << 16)) raw_chunk = 1u << 16;             // (a BGZF block is at most 64 KiB: every chunk holds at least one)
    Pool pool(threads ? threads : usable_cpus());
    const unsigned n_workers = pool.size() + Pool::MAX_HELPERS;   // (+ the stage threads that help, each under its own id)

    std::thread t_read([&] {
      try {
        size_t pos = 0;
        bool eof = false;
        for (uint64_t s = 0; !eof; s++) {
          if (!enter(0, s)) return;
          const auto t0 = Clock::now();
          Chunk& c = chunks[s % N_CHUNKS];
          const size_t have = file_len - pos < raw_chunk ? file_len - pos : (size_t)raw_chunk;
          eof = pos + have == file_len;
          std::string e;
          uint64_t infl = 0;
          const size_t used = block_table(file + pos, have, c.blocks, &infl, &e);
          if (used == (size_t)-1) { fail(e); return; }
          if (eof && used != have) { fail("the file ends inside a BGZF block"); return; }
          if (!eof && used == 0) { fail("no whole BGZF block inside a chunk"); return; }
          c.raw = file + pos; c.raw_len = used; c.inf_len = infl; c.last = eof;
          pos += used; in_bytes += used;
          busy[0] += since(t0);
          if (eof) { std::lock_guard<std::mutex> l(m); n_chunks_total = s + 1; }
          leave(0);
        }
      } catch (const std::exception& ex) { fail(ex.what()); }
    });

    std::thread t_inflate([&] {
      try {
        std::vector<z_stream> zs(n_workers);
        std::vector<char> zs_init(n_workers, 0);
        std::vector<uint8_t> scratch((size_t)n_workers * BGZF_SLOT);
        for (uint64_t s = 0;; s++) {
          if (!enter(1, s)) break;
          const auto t0 = Clock::now();
          Chunk& c = chunks[s % N_CHUNKS];
          if (device_inflate) {
            // the blocks are inflated on the device: here the compressed bytes only move into pinned memory (parallel copy), and the
            // descriptors are written.  The first chunk's leading blocks are inflated here as well, just to measure the BAM header.
            c.inf.reserve(c.raw_len + 64, pinned);
            const size_t piece = 1u << 20, np = (c.raw_len + piece - 1) / piece;
            pool.parallel_for(np, 1, [&](size_t i, unsigned) { const size_t o = i * piece, n = c.raw_len - o < piece ? c.raw_len - o : piece; memcpy(c.inf.p + o, c.raw + o, n); });
            memset(c.inf.p + c.raw_len, 0, 64);
            c.dev_blocks.resize(c.blocks.size());
            for (size_t i = 0; i < c.blocks.size(); i++) {
              const Block& b = c.blocks[i];
              const uint32_t xlen = c.raw[b.in_off + 10] | (c.raw[b.in_off + 11] << 8);
              fgx::BgzfDevBlock d;
              d.in_off = b.in_off + 12 + xlen; d.out_off = b.out_off; d.in_len = b.in_size - 12 - xlen - 8; d.isize = b.isize;
              memcpy(&d.crc, c.raw + b.in_off + b.in_size - 8, 4);
              d.ent_off = 0;                                     // (bgzf_inflate_plan fills it when the chunk is launched)
              c.dev_blocks[i] = d;
            }
            c.header_size = 0;
            if (s == 0) {
              std::vector<uint8_t> head;
              for (size_t i = 0; i < c.blocks.size() && c.header_size == 0; i++) {
                const Block& b = c.blocks[i];
                const size_t at = head.size();
                head.resize(at + b.isize);
                if (b.isize) {
                  const uint32_t xlen = c.raw[b.in_off + 10] | (c.raw[b.in_off + 11] << 8);
                  z_stream z;
                  memset(&z, 0, sizeof(z));
                  if (inflateInit2(&z, -15) != Z_OK) { fail("zlib"); break; }
                  z.next_in = (Bytef*)(c.raw + b.in_off + 12 + xlen); z.avail_in = b.in_size - 12 - xlen - 8;
                  z.next_out = head.data() + at; z.avail_out = b.isize;
                  const int rc = inflate(&z, Z_FINISH);
                  inflateEnd(&z);
                  if (rc != Z_STREAM_END) { fail("a BGZF block of the header failed to inflate"); break; }
                }
                c.header_size = bam_header_size(head.data(), head.size());
                if (head.size() > (64u << 20)) break;
              }
            }
            inflated_bytes += c.inf_len;
            busy[1] += since(t0);
            leave(1);
            continue;
          }
          c.inf.reserve(c.inf_len + 64, pinned);
          std::atomic<int> bad(0);
          pool.parallel_for(c.blocks.size(), 8, [&](size_t i, unsigned w) {
            const Block& b = c.blocks[i];
            if (b.isize == 0) return;
            const uint8_t* raw = c.raw;
            const uint32_t xlen = raw[b.in_off + 10] | (raw[b.in_off + 11] << 8);
            z_stream& z = zs[w];
            if (!zs_init[w]) { memset(&z, 0, sizeof(z)); if (inflateInit2(&z, -15) != Z_OK) { bad = 1; return; } zs_init[w] = 1; }
            else if (inflateReset(&z) != Z_OK) { bad = 1; return; }
            // inflate into the worker's own 64 KiB block, then ONE copy into the pinned chunk buffer: LZ77 matches are copies out of
            // what was just written, and reading pinned (device-visible) memory back is far slower than reading a block that sits in L2
            uint8_t* tmp = scratch.data() + (size_t)w * BGZF_SLOT;
            z.next_in = (Bytef*)(raw + b.in_off + 12 + xlen); z.avail_in = b.in_size - 12 - xlen - 8;
            z.next_out = tmp; z.avail_out = b.isize;
            const int rc = inflate(&z, Z_FINISH);
            uint32_t crc;
            memcpy(&crc, raw + b.in_off + b.in_size - 8, 4);
            if (rc != Z_STREAM_END || z.total_out != b.isize || (uint32_t)crc32(0L, tmp, b.isize) != crc) { bad = 1; return; }
            memcpy(c.inf.p + b.out_off, tmp, b.isize);
          });
          if (bad) { fail("a BGZF block failed to inflate or its CRC32 / ISIZE does not match"); break; }
          inflated_bytes += c.inf_len;
          busy[1] += since(t0);
          leave(1);
        }
        for (unsigned w = 0; w < n_workers; w++) if (zs_init[w]) inflateEnd(&zs[w]);
      } catch (const std::exception& ex) { fail(ex.what()); }
    });

    std::thread t_middle([&] {
      try {
        for (uint64_t s = 0;; s++) {
          if (!enter(2, s)) return;
          const auto t0 = Clock::now();
          middle(chunks[s % N_CHUNKS], s);
          out_bytes += chunks[s % N_CHUNKS].out_len;
          busy[2] += since(t0);
          leave(2);
        }
      } catch (const std::exception& ex) { fail(ex.what()); }
    });

    // cuts `src` into BGZF blocks (parallel, one 64 KiB slot each), then packs them back to back into `packed` (parallel copies):
    // the writer hands the file system one large buffer per chunk instead of tens of thousands of 8 KB pieces
    std::vector<uint8_t> dscratch((size_t)n_workers * BGZF_SLOT + 64);
    std::vector<std::unique_ptr<fgx::DeflateScratch>> dstate(n_workers);
    auto deflate_stream = [&](const uint8_t* src, uint64_t len, HostBuf& comp, std::vector<uint32_t>& sizes, HostBuf& packed, uint64_t* packed_len,
                              bool use_scratch, const uint32_t* crcs) -> bool {
      const size_t nb = (size_t)((len + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD);
      comp.reserve(nb * BGZF_SLOT + 64, false);
      sizes.assign(nb, 0);
      std::atomic<int> bad(0);
      pool.parallel_for(nb, 4, [&](size_t i, unsigned w) {
        const uint8_t* in = src + i * (uint64_t)BGZF_PAYLOAD;
        const uint32_t n = (uint32_t)((len - i * (uint64_t)BGZF_PAYLOAD) < BGZF_PAYLOAD ? (len - i * (uint64_t)BGZF_PAYLOAD) : BGZF_PAYLOAD);
        if (use_scratch) {                         // (the records come out of pinned memory: one read of it, not zlib's several)
          uint8_t* tmp = dscratch.data() + (size_t)w * BGZF_SLOT;
          memcpy(tmp, in, n);
          in = tmp;
        }
        uint8_t* blk = comp.p + i * BGZF_SLOT;
        uint32_t csize = 0;
        // level 1 (the reference's default for consensus output): this repository's own block compressor (deflate_core.h — the one the
        // device runs a lane per block; on the host it is about twice as fast as zlib level 1 and a little smaller on consensus records)
        if (level == 1) {
          if (!dstate[w]) dstate[w].reset(new fgx::DeflateScratch());
          if (in != dscratch.data() + (size_t)w * BGZF_SLOT) { memcpy(dscratch.data() + (size_t)w * BGZF_SLOT, in, n); in = dscratch.data() + (size_t)w * BGZF_SLOT; }   // (8 readable bytes behind the block)
          csize = fgx::deflate_block(in, n, blk + 18, (uint32_t)(BGZF_SLOT - 18 - 8), *dstate[w]);
        }
        for (int lv = level == 1 ? 0 : level; csize == 0; lv = 0) {   // zlib for the other levels; a payload that does not fit is stored (always fits)
          z_stream z;
          memset(&z, 0, sizeof(z));
          if (deflateInit2(&z, lv, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; return; }
          z.next_in = (Bytef*)in; z.avail_in = n;
          z.next_out = blk + 18; z.avail_out = (uInt)(BGZF_SLOT - 18 - 8);
          const int rc = deflate(&z, Z_FINISH);
          const uint32_t got = (uint32_t)z.total_out;
          deflateEnd(&z);
          if (rc == Z_STREAM_END) { csize = got; break; }
          if (lv == 0) { bad = 1; return; }
        }
        const uint32_t bsize = 18 + csize + 8 - 1;
        const uint8_t hdr[18] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
        memcpy(blk, hdr, 18);
        const uint32_t crc = crcs ? crcs[i] : (uint32_t)crc32(0L, in, n);   // (zlib's crc32: ~1 GB/s per core — as much time as the compressor takes)
        memcpy(blk + 18 + csize, &crc, 4);
        memcpy(blk + 18 + csize + 4, &n, 4);
        sizes[i] = bsize + 1;
      });
      if (bad) return false;
      std::vector<uint64_t> offs(nb + 1, 0);
      for (size_t i = 0; i < nb; i++) offs[i + 1] = offs[i] + sizes[i];
      packed.reserve(offs[nb] + 64, false);
      pool.parallel_for(nb, 16, [&](size_t i, unsigned) { memcpy(packed.p + offs[i], comp.p + i * BGZF_SLOT, sizes[i]); });
      *packed_len = offs[nb];
      return true;
    };

    std::thread t_deflate([&] {
      try {
        for (uint64_t s = 0;; s++) {
          if (!enter(3, s)) return;
          const auto t0 = Clock::now();
          Chunk& c = chunks[s % N_CHUNKS];
          if (!c.precompressed && !deflate_stream(c.out.p, c.out_len, c.comp, c.comp_size, c.packed, &c.packed_len, c.out.pinned, c.have_crcs ? c.crcs.data() : nullptr)) { fail("deflate failed"); return; }
          c.rej_packed_len = 0;
          if (frej && c.rej_len && !deflate_stream(c.rej.p, c.rej_len, c.rej_comp, c.rej_sizes, c.rej_packed, &c.rej_packed_len, c.rej.pinned, nullptr)) { fail("deflate failed"); return; }
          busy[3] += since(t0);
          leave(3);
        }
      } catch (const std::exception& ex) { fail(ex.what()); }
    });

    std::thread t_write([&] {
      try {
        auto put = [&](const uint8_t* p, size_t n) { if (n && fwrite(p, 1, n, fout) != n) throw std::runtime_error("write failed"); out_file_bytes += n; };
        {
          HostBuf hc, hp; std::vector<uint32_t> hs;
          uint64_t hl = 0;
          if (out_header_len) {
            if (!deflate_stream(out_header, out_header_len, hc, hs, hp, &hl, false, nullptr)) { fail("deflate failed"); return; }
            put(hp.p, hl);
          }
        }
        auto put_rej = [&](const uint8_t* p, size_t n) { if (n && fwrite(p, 1, n, frej) != n) throw std::runtime_error("write of the rejects file failed"); rej_file_bytes += n; };
        bool rej_header_written = false;
        auto rej_head = [&] {                          // the input's own header, as its own BGZF block(s) (known once the first chunk has passed the middle stage)
          if (!frej || rej_header_written) return;
          rej_header_written = true;
          HostBuf hc, hp; std::vector<uint32_t> hs;
          uint64_t hl = 0;
          if (!rej_header.empty()) {
            if (!deflate_stream(rej_header.data(), rej_header.size(), hc, hs, hp, &hl, false, nullptr)) throw std::runtime_error("deflate failed");
            put_rej(hp.p, hl);
          }
        };
        for (uint64_t s = 0;; s++) {
          if (!enter(4, s)) break;
          const auto t0 = Clock::now();
          Chunk& c = chunks[s % N_CHUNKS];
          put(c.packed.p, c.packed_len);
          if (frej) { rej_head(); put_rej(c.rej_packed.p, c.rej_packed_len); rej_bytes += c.rej_len; }
          busy[4] += since(t0);
          leave(4);
        }
        static const uint8_t EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (!failed) put(EOF_BLOCK, 28);
        if (!failed && frej) { rej_head(); put_rej(EOF_BLOCK, 28); }
      } catch (const std::exception& ex) { fail(ex.what()); }
    });

    t_read.join(); t_inflate.join(); t_middle.join(); t_deflate.join(); t_write.join();
    if (file) munmap((void*)file, file_len);
    close(fd);
    if (fclose(fout) != 0 && !failed) { failed = true; err = "closing the output file failed"; }
    if (frej && fclose(frej) != 0 && !failed) { failed = true; err = "closing the rejects file failed"; }
    return failed ? 1 : 0;
  }
};

// size of the BAM header at the start of an uncompressed stream, or 0 when the data does not hold all of it
uint64_t bam_header_size(const uint8_t* p, uint64_t n) {
  if (n < 12 || memcmp(p, "BAM\1", 4) != 0) return 0;
  uint32_t l_text;
  memcpy(&l_text, p + 4, 4);
  uint64_t o = 8ull + l_text;
  if (o + 4 > n) return 0;
  uint32_t n_ref;
  memcpy(&n_ref, p + o, 4);
  o += 4;
  for (uint32_t i = 0; i < n_ref; i++) {
    if (o + 4 > n) return 0;
    uint32_t l_name;
    memcpy(&l_name, p + o, 4);
    o += 8ull + l_name;
    if (o > n) return 0;
  }
  return o;
}

}  // namespace
