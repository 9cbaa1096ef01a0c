// bgzf_host.cpp — BGZF container work on the host cores: what sits between a BAM file and the
// engine's uncompressed record streams.  Replaces, for this path, crates/fgumi-bgzf/src/{reader,writer}.rs (block framing,
// inflate / deflate, CRC32) with a block-parallel implementation over zlib: BGZF blocks are independent gzip members of at
// most 64 KiB, so T threads take blocks from a shared counter; output positions come from a prefix sum of the ISIZE trailers
// (inflate) or of the compressed sizes (deflate).  Level 1 is the reference's default for consensus output.
#include <zlib.h>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/fgumi_amd.h"
#include "inflate_core.h"
#include "deflate_core.h"

namespace {
constexpr uint32_t BGZF_PAYLOAD = 0xFF00;     // uncompressed bytes per block (htslib / noodles / fgumi-bgzf)
struct Block { uint64_t in_off; uint32_t in_size; uint32_t isize; uint64_t out_off; };

thread_local std::string t_err;

unsigned pick_threads(uint32_t threads, size_t n_blocks) {
  unsigned T = threads ? threads : std::thread::hardware_concurrency();
  if (T == 0) T = 1;
  if (T > n_blocks) T = (unsigned)(n_blocks ? n_blocks : 1);
  return T;
}
template <class F> void run_pool(unsigned T, size_t n, F fn) {
  std::atomic<size_t> next(0);
  auto worker = [&]() { for (;;) { size_t i = next.fetch_add(16); if (i >= n) return; size_t e = i + 16 < n ? i + 16 : n; for (; i < e; i++) fn(i); } };
  if (T <= 1) { worker(); return; }
  std::vector<std::thread> ts;
  for (unsigned t = 0; t < T; t++) ts.emplace_back(worker);
  for (auto& t : ts) t.join();
}
}  // namespace

extern "C" {

const char* fgx_bgzf_last_error(void) { return t_err.c_str(); }
void fgx_bgzf_free(uint8_t* p) { free(p); }

int fgx_bgzf_inflate(const uint8_t* raw, uint64_t raw_len, uint32_t threads, uint8_t** out, uint64_t* out_len) {
  if (!raw || !out || !out_len) { t_err = "fgx_bgzf_inflate: null argument"; return 1; }
  std::vector<Block> blocks;
  uint64_t p = 0, total = 0;
  while (p < raw_len) {                        // BSIZE chain
    if (raw_len - p < 26 || raw[p] != 0x1F || raw[p + 1] != 0x8B || raw[p + 2] != 8 || !(raw[p + 3] & 4)) { t_err = "not a BGZF block at offset " + std::to_string(p); return 1; }
    const uint32_t xlen = raw[p + 10] | (raw[p + 11] << 8);
    uint64_t q = p + 12, end = p + 12 + xlen;
    uint32_t bsize = 0;
    while (q + 4 <= end && end <= raw_len) {
      const uint32_t slen = raw[q + 2] | (raw[q + 3] << 8);
      if (q + 4 + slen > end) break;           // a subfield that overruns XLEN: no BC taken from it (the block is refused below)
      if (raw[q] == 'B' && raw[q + 1] == 'C' && slen == 2) bsize = (uint32_t)(raw[q + 4] | (raw[q + 5] << 8)) + 1;
      q += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || p + bsize > raw_len) { t_err = "BGZF block at offset " + std::to_string(p) + " has no BC subfield or is truncated"; return 1; }
    uint32_t isize;
    memcpy(&isize, raw + p + bsize - 4, 4);
    if (isize > 0x10000) { t_err = "BGZF block at offset " + std::to_string(p) + " claims more than 64 KiB"; return 1; }
    blocks.push_back(Block{p, bsize, isize, total});
    total += isize;
    p += bsize;
  }
  uint8_t* dst = (uint8_t*)malloc(total ? total : 1);
  if (!dst) { t_err = "out of memory"; return 1; }
  std::atomic<int> bad(0);
  run_pool(pick_threads(threads, blocks.size()), blocks.size(), [&](size_t i) {
    const Block& b = blocks[i];
    if (b.isize == 0) return;
    const uint32_t xlen = raw[b.in_off + 10] | (raw[b.in_off + 11] << 8);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; return; }
    zs.next_in = (Bytef*)(raw + b.in_off + 12 + xlen); zs.avail_in = b.in_size - 12 - xlen - 8;
    zs.next_out = dst + b.out_off; zs.avail_out = b.isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    uint32_t crc;
    memcpy(&crc, raw + b.in_off + b.in_size - 8, 4);
    if (rc != Z_STREAM_END || zs.total_out != b.isize || (uint32_t)crc32(0L, dst + b.out_off, b.isize) != crc) bad = 1;
  });
  if (bad) { free(dst); t_err = "BGZF block failed to inflate or its CRC32 / ISIZE does not match"; return 1; }
  *out = dst; *out_len = total;
  return 0;
}

// the device's DEFLATE decoder (inflate_core.h) run on the host: the same source, for the CPU tests.  `in` must be readable for
// 8 bytes past in_len.  Returns its status (0 = ok) and the CRC-32 of the output computed with the device's fold (slices of 1/64).
int fgx_inflate_block_host(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, uint32_t* crc_out) {
  static thread_local fgx::InflateTables T;
  const int st = fgx::inflate_block(in, in_len, out, out_len, T.f, T.w);
  if (crc_out) {
    uint32_t tab[256];
    for (uint32_t i = 0; i < 256; i++) tab[i] = fgx::crc32_table_entry(i);
    uint32_t lane_crc[64];
    const uint32_t S = (out_len + 63u) / 64u;
    for (uint32_t lane = 0; lane < 64; lane++) {
      const int64_t hi_s = (int64_t)out_len - (int64_t)(63u - lane) * S, lo_s = hi_s - (int64_t)S;
      const uint32_t hi = hi_s > 0 ? (uint32_t)hi_s : 0u, lo = lo_s > 0 ? (uint32_t)lo_s : 0u;
      uint32_t crc = 0;
      if (hi > lo) { crc = 0xFFFFFFFFu; for (uint32_t i = lo; i < hi; i++) crc = tab[(crc ^ out[i]) & 0xFFu] ^ (crc >> 8); crc ^= 0xFFFFFFFFu; }
      lane_crc[lane] = crc;
    }
    uint32_t op = out_len ? fgx::crc32_shift_op(S) : 0u;
    for (uint32_t step = 1; step < 64; step <<= 1) {
      for (uint32_t lane = 0; lane < 64; lane += 2 * step) lane_crc[lane] = fgx::crc32_multmodp(op, lane_crc[lane]) ^ lane_crc[lane + step];
      op = fgx::crc32_multmodp(op, op);
    }
    *crc_out = out_len ? lane_crc[0] : 0u;
  }
  return st;
}

// The two-phase form (round 5) on the host, for the CPU tests: inflate_block_t<.., TOK> makes the entry list, then either the plain
// second pass (mode 0: inflate_resolve) or k_bgzf_resolve's schedule emulated lane by lane (mode 1: inflate_resolve_wave_emulated, the
// batches of 64 entries and the frontier rule, with the "parallel" copies of a round done in REVERSE lane order so that a copy that
// reads bytes a lower lane of the same round writes would show).  Same contract as fgx_inflate_block_host; *n_entries = the list's length.
int fgx_inflate_block_two_phase_host(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t out_len, int mode, uint32_t* n_entries, uint32_t* rounds) {
  static thread_local fgx::InflateTables T;
  static thread_local std::vector<uint32_t> ent(fgx::INFL_ENTRY_CAP);
  uint32_t ne = 0;
  // (the list gets what the device gives it: the per-ISIZE bound of infl_entry_cap, engine.h — the CPU tests therefore exercise that bound)
  const uint32_t cap = fgx::infl_entry_cap(out_len) < (uint32_t)ent.size() ? fgx::infl_entry_cap(out_len) : (uint32_t)ent.size();
  const int st = fgx::inflate_block_t<uint16_t*, true>(in, in_len, out, out_len, T.f.lit, T.f.dist, T.w, ent.data(), cap, &ne);
  if (n_entries) *n_entries = ne;
  if (rounds) *rounds = 0;
  if (st != fgx::INFL_OK) return st;
  if (mode == 0) return fgx::inflate_resolve(out, out_len, ent.data(), ne) ? 0 : fgx::INFL_SIZE_MISMATCH;
  return fgx::inflate_resolve_wave_emulated(out, out_len, ent.data(), ne, rounds) ? 0 : fgx::INFL_SIZE_MISMATCH;
}

// the device's DEFLATE compressor (deflate_core.h) run on the host: the same source, for the CPU tests.  `in` must be readable for
// 8 bytes past n.  Returns the compressed size, 0 when the stream does not fit `cap` (a block to be stored).
uint32_t fgx_deflate_block_host(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap) {
  static thread_local fgx::DeflateScratch* S = nullptr;
  if (!S) S = new fgx::DeflateScratch();
  return fgx::deflate_block(in, n, out, cap, *S);
}

int fgx_bgzf_deflate(const uint8_t* in, uint64_t len, int level, uint32_t threads, int with_eof, uint8_t** out, uint64_t* out_len) {
  if ((!in && len) || !out || !out_len) { t_err = "fgx_bgzf_deflate: null argument"; return 1; }
  const size_t nb = (size_t)((len + BGZF_PAYLOAD - 1) / BGZF_PAYLOAD);
  constexpr size_t SLOT = 0x10000;             // a BGZF block never exceeds 64 KiB
  uint8_t* tmp = (uint8_t*)malloc(nb ? nb * SLOT : 1);
  std::vector<uint32_t> sizes(nb, 0);
  if (!tmp) { t_err = "out of memory"; return 1; }
  std::atomic<int> bad(0);
  const unsigned T = pick_threads(threads, nb);
  run_pool(T, nb, [&](size_t i) {
    const uint8_t* src = in + i * (uint64_t)BGZF_PAYLOAD;
    const uint32_t n = (uint32_t)((len - i * (uint64_t)BGZF_PAYLOAD) < BGZF_PAYLOAD ? (len - i * (uint64_t)BGZF_PAYLOAD) : BGZF_PAYLOAD);
    uint8_t* blk = tmp + i * SLOT;
    uint32_t csize = 0;
    for (int lv = level; ; lv = 0) {          // an incompressible payload falls back to stored blocks (always fits)
      z_stream zs;
      memset(&zs, 0, sizeof(zs));
      if (deflateInit2(&zs, lv, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { bad = 1; return; }
      zs.next_in = (Bytef*)src; zs.avail_in = n;
      zs.next_out = blk + 18; zs.avail_out = (uInt)(SLOT - 18 - 8);
      const int rc = deflate(&zs, Z_FINISH);
      csize = (uint32_t)zs.total_out;
      deflateEnd(&zs);
      if (rc == Z_STREAM_END) break;
      if (lv == 0) { bad = 1; return; }
    }
    const uint32_t bsize = 18 + csize + 8 - 1;
    const uint8_t hdr[18] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 'B', 'C', 2, 0, (uint8_t)bsize, (uint8_t)(bsize >> 8)};
    memcpy(blk, hdr, 18);
    const uint32_t crc = (uint32_t)crc32(0L, src, n);
    memcpy(blk + 18 + csize, &crc, 4);
    memcpy(blk + 18 + csize + 4, &n, 4);
    sizes[i] = bsize + 1;
  });
  if (bad) { free(tmp); t_err = "deflate failed"; return 1; }
  std::vector<uint64_t> offs(nb + 1, 0);
  for (size_t i = 0; i < nb; i++) offs[i + 1] = offs[i] + sizes[i];
  static const uint8_t EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t total = offs[nb] + (with_eof ? 28 : 0);
  uint8_t* dst = (uint8_t*)malloc(total ? total : 1);
  if (!dst) { free(tmp); t_err = "out of memory"; return 1; }
  run_pool(T, nb, [&](size_t i) { memcpy(dst + offs[i], tmp + i * SLOT, sizes[i]); });
  if (with_eof) memcpy(dst + offs[nb], EOF_BLOCK, 28);
  free(tmp);
  *out = dst; *out_len = total;
  return 0;
}

}  // extern "C"
