// pipeline_ring.cpp — fgx_run_bam with SEVERAL chunks on their way into the device (FGX_PIPE_RING=1, opt-in; round 4).
//
// The same five host stages as pipeline.cpp; the device stage keeps a ring of five stream buffers, every buffer with its own stream, and
// while it works on one chunk up to four later ones are uploaded and inflated side by side (one chunk's BGZF blocks do not fill the chip:
// the inflate kernel is a latency chain per block).  Measured: 60 -> 72 M raw reads/s on a 1 M-family file (profiles/r04_experiments.md).
// NOT the default: with the runtime's default of four hardware queues per process the whole GPU suite is green on it, but with eight
// (GPU_MAX_HW_QUEUES=8) or with high-priority compute streams — i.e. as soon as the device stage's kernels really run beside the fills —
// the multi-chunk device-inflate tests find groups cut in two (bytes of the buffer's previous chunk in front of a stream); pipeline.cpp
// passes the same tests under the same settings.  The cause was not found before the round's GPU time ran out.
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
struct Pipeline {
  static constexpr int N_CHUNKS = 8, N_STAGES = 5;   // (N_CHUNKS: one chunk per stage + the chunks the device stage has on their way in, PipeState::NB - 1)
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
              d._pad = 0;
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

thread_local std::string t_perr;

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
  uint32_t* h_status = nullptr;       // (pinned) the inflate kernels' status words, 16 words apart
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
      fgx::hip_check(hipHostMalloc((void**)&S->h_status, 64 * NB, hipHostMallocDefault), "hipHostMalloc");
    }
    // chunks on their way in beside the one in the device stage (FGX_PIPE_AHEAD = 1 .. NB - 1: a measuring knob)
    const uint64_t max_ahead = [] { const char* e = getenv("FGX_PIPE_AHEAD"); const int v = e ? atoi(e) : 0; return (uint64_t)((v >= 1 && v < NB) ? v : NB - 1); }();
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
      hipStream_t si = S->s_in[buf];
      uint32_t* const h_status = S->h_status + 16 * buf;
      fgx::DevBuf &d_raw = S->d_raw[buf], &d_blk = S->d_blk[buf];
      fgx::hip_check(hipEventRecord(S->ev_up0[buf], si), "hipEventRecord");
      if (device_inflate) {
        const size_t blk_bytes = ch.dev_blocks.size() * sizeof(fgx::BgzfDevBlock);
        d_raw.reserve(ch.raw_len + 64);
        d_blk.reserve(blk_bytes + 64 + 16);
        fgx::hip_check(hipMemcpyAsync(d_raw.p, ch.inf.p, ch.raw_len + 64, hipMemcpyHostToDevice, si), "H2D compressed chunk");
        if (blk_bytes) fgx::hip_check(hipMemcpyAsync(d_blk.p, ch.dev_blocks.data(), blk_bytes, hipMemcpyHostToDevice, si), "H2D block table");
        fgx::hip_check(hipEventRecord(S->ev_up1[buf], si), "hipEventRecord");
        fgx::bgzf_inflate_launch(si, d_raw.as<uint8_t>(), d_blk.as<fgx::BgzfDevBlock>(), (uint32_t)ch.dev_blocks.size(), dst,
                                 (uint32_t*)((uint8_t*)d_blk.p + ((blk_bytes + 15) & ~(size_t)15)), h_status);
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
        try_ahead();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      // (hipEventQuery says the fill has finished; hipEventSynchronize is what the kernels of the device stage — another stream, another
      // hardware queue, other XCDs — may rely on for SEEING what it wrote: with eight hardware queues the multi-chunk tests read stale
      // bytes of the buffer's previous chunk after the query alone, the round-3 form with the synchronize never did)
      fgx::hip_check(hipEventSynchronize(S->ev_in[cur]), "hipEventSynchronize");
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
