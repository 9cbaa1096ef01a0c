// grouping.hip — MI grouping of a BAM record stream on the device: which records a consensus command keeps and where
// the runs of equal group key begin.  Output = the (rec_off, rec_len, grp_first) arrays fgx_process_batch[_device] takes.
//
// Mirrors   src/lib/mi_group.rs:227-310          MiGrouper::get_mi_tag / add_records (consecutive records with an equal key;
//                                                 key = <MI value, transformed> + '\t' + <cell tag value> when a cell tag is
//                                                 configured; records without the MI tag are skipped)
//           src/lib/commands/common.rs:384-397   consensus_pregroup_keep_flags (secondary / supplementary always dropped,
//                                                 unmapped dropped unless --allow-unmapped)
//           crates/fgumi-umi/src/lib.rs:370-375  extract_mi_base (duplex: cut the value at its last '/', if not leading)
// Keys are compared as bytes (the reference compares `from_utf8_lossy` strings: identical for valid UTF-8 tag values).
#ifndef FGX_DEVEMU            // (tests/apiemu compiles this file for the host with a serial scan)
#include <hipcub/hipcub.hpp>
#endif
#include "bamrec.h"
#include "engine.h"

namespace fgx {

namespace {

// find_z_tag (bamrec.h) with 4- and 8-byte loads: the tag walk of one record is a chain of dependent HBM loads per thread, so
// fewer, wider loads is what shortens it.  `safe` = bytes that may be read from `aux` on (the rest of the blob); semantics
// identical to bam::find_z_tag (first occurrence wins, a malformed entry ends the walk).
__device__ inline uint32_t ldg32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ inline unsigned long long ldg64u(const uint8_t* p) { unsigned long long v; __builtin_memcpy(&v, p, 8); return v; }
__device__ inline int64_t find_nul_wide(const uint8_t* p, uint32_t n, uint64_t safe) {
  uint32_t i = 0;
  for (; i + 8 <= n && (uint64_t)i + 8 <= safe; i += 8) {
    const unsigned long long v = ldg64u(p + i);
    const unsigned long long t = (v - 0x0101010101010101ULL) & ~v & 0x8080808080808080ULL;
    if (t) return (int64_t)i + (__builtin_ctzll(t) >> 3);
  }
  for (; i < n; i++) if (p[i] == 0) return i;
  return -1;
}
__device__ inline int64_t find_z_tag_wide(const uint8_t* aux, uint32_t n, uint64_t safe, uint8_t t0, uint8_t t1, uint32_t* vlen) {
  const uint32_t want = (uint32_t)t0 | ((uint32_t)t1 << 8);
  uint32_t p = 0;
  while (p + 3 <= n) {
    uint32_t hd;
    if ((uint64_t)p + 4 <= safe) hd = ldg32u(aux + p);
    else hd = (uint32_t)aux[p] | ((uint32_t)aux[p + 1] << 8) | ((uint32_t)aux[p + 2] << 16) | (p + 3 < n ? (uint32_t)aux[p + 3] << 24 : 0u);
    const uint8_t vt = (uint8_t)(hd >> 16);
    if ((hd & 0xFFFF) == want) {
      if (vt != 'Z') return -1;
      const int64_t e = find_nul_wide(aux + p + 3, n - (p + 3), safe - (p + 3));
      if (e < 0) return -1;
      *vlen = (uint32_t)e;
      return (int64_t)p + 3;
    }
    const int fixed = bam::tag_fixed_size(vt);
    uint32_t size;
    if (fixed > 0) size = (uint32_t)fixed;
    else if (vt == 'Z' || vt == 'H') {
      const int64_t e = find_nul_wide(aux + p + 3, n - (p + 3), safe - (p + 3));
      if (e < 0) return -1;
      size = (uint32_t)e + 1;
    } else if (vt == 'B') {
      if (n - (p + 3) < 5) return -1;
      const int es = bam::tag_fixed_size((uint8_t)(hd >> 24));
      if (es == 0) return -1;
      const uint64_t cnt = (uint64_t)p + 8 <= safe ? ldg32u(aux + p + 4) : bam::rd32(aux + p + 4);
      const uint64_t sz = 5 + cnt * (uint64_t)es;
      if (sz > 0xFFFFFFFFull) return -1;
      size = (uint32_t)sz;
    } else return -1;
    const uint64_t np = (uint64_t)p + 3 + size;
    if (np > n) break;
    p = (uint32_t)np;
  }
  return -1;
}

struct KeyLoc { uint64_t mi_off; uint32_t mi_len; uint32_t cb_len; uint64_t cb_off; };   // positions of the key's two parts in the blob

// one thread per record: keep decision and key location
__global__ void k_group_keys(const uint8_t* __restrict__ blob, uint64_t blob_len, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len,
                             uint32_t n, fgx_group_options o, uint32_t* __restrict__ keep, KeyLoc* __restrict__ loc) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t off = rec_off[r];
  const uint32_t len = rec_len[r];
  uint32_t k = 0;
  KeyLoc L{0, 0, 0, 0};
  if (len >= 32 && off + len <= blob_len) {
    bam::Rec v{blob + off, len};
    const uint16_t f = v.flags();
    const bool flags_ok = !(f & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) && (o.allow_unmapped || !(f & bam::F_UNMAPPED));
    const uint64_t aux = (uint64_t)v.aux_off();
    if (flags_ok && aux <= len) {
      const uint32_t an = len - (uint32_t)aux;
      uint32_t vl = 0;
      const uint64_t safe = blob_len - (off + aux);        // bytes of the blob from the aux block on
      const int64_t m = find_z_tag_wide(v.b + aux, an, safe, (uint8_t)o.tag[0], (uint8_t)o.tag[1], &vl);
      if (m >= 0) {
        k = 1;
        if (o.strip_strand_suffix) {          // extract_mi_base: everything before the last '/', unless that '/' leads the value
          const uint8_t* s = v.b + aux + m;
          for (uint32_t i = vl; i-- > 1;) if (s[i] == '/') { vl = i; break; }
        }
        L.mi_off = off + aux + (uint64_t)m; L.mi_len = vl;
        if (o.cell_tag[0]) {
          uint32_t cl = 0;
          const int64_t cpos = find_z_tag_wide(v.b + aux, an, safe, (uint8_t)o.cell_tag[0], (uint8_t)o.cell_tag[1], &cl);
          if (cpos >= 0) { L.cb_off = off + aux + (uint64_t)cpos; L.cb_len = cl; }      // an absent cell tag and an empty one give the same key
        }
      }
    }
  }
  keep[r] = k;
  loc[r] = L;
}

__global__ void k_group_compact(const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len, uint32_t n, const uint32_t* __restrict__ keep,
                                const uint32_t* __restrict__ kpos, const KeyLoc* __restrict__ loc, uint64_t* __restrict__ out_off,
                                uint32_t* __restrict__ out_len, KeyLoc* __restrict__ kloc) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || !keep[r]) return;
  const uint32_t k = kpos[r];
  out_off[k] = rec_off[r]; out_len[k] = rec_len[r]; kloc[k] = loc[r];
}

// one thread per kept record: does it open a new group?
__global__ void k_group_bounds(const uint8_t* __restrict__ blob, const KeyLoc* __restrict__ kloc, uint32_t n_kept, uint32_t* __restrict__ bound) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_kept) return;
  uint32_t b = 1;
  if (k > 0) {
    const KeyLoc A = kloc[k - 1], B = kloc[k];
    if (A.mi_len == B.mi_len && A.cb_len == B.cb_len) {
      auto equal = [&](uint64_t x, uint64_t y, uint32_t n) {      // n bytes at blob + x and blob + y (both runs lie inside records)
        uint32_t i = 0;
        for (; i + 8 <= n; i += 8) if (ldg64u(blob + x + i) != ldg64u(blob + y + i)) return false;
        for (; i < n; i++) if (blob[x + i] != blob[y + i]) return false;
        return true;
      };
      if (equal(A.mi_off, B.mi_off, B.mi_len) && equal(A.cb_off, B.cb_off, B.cb_len)) b = 0;
    }
  }
  bound[k] = b;
}

__global__ void k_group_firsts(const uint32_t* __restrict__ bound, const uint32_t* __restrict__ gincl, uint32_t n_kept, uint32_t* __restrict__ grp_first) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_kept) return;
  if (bound[k]) grp_first[gincl[k] - 1] = k;
  if (k == n_kept - 1) grp_first[gincl[k]] = n_kept;
}

}  // namespace

int group_records_device(fgx_caller* c, const fgx_group_options* o, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off,
                         const uint32_t* d_rec_len, uint32_t n, uint64_t* d_out_off, uint32_t* d_out_len, uint32_t* d_grp_first, uint32_t* n_kept,
                         uint32_t* n_grp) {
  hipStream_t s = c->stream;
  *n_kept = 0; *n_grp = 0;
  if (n == 0) { uint32_t z = 0; hip_check(hipMemcpyAsync(d_grp_first, &z, 4, hipMemcpyHostToDevice, s), "H2D"); hip_check(hipStreamSynchronize(s), "sync"); return 0; }
  DevBuf& keep = c->d_scratch_a;      // keep | kpos | bound | gincl (4 x n u32)
  DevBuf& locs = c->d_scratch_b;      // loc | kloc (2 x n KeyLoc)
  keep.reserve((size_t)n * 16 + 64);
  locs.reserve((size_t)n * 2 * sizeof(KeyLoc) + 64);
  uint32_t* d_keep = keep.as<uint32_t>(); uint32_t* d_kpos = d_keep + n; uint32_t* d_bound = d_kpos + n; uint32_t* d_gincl = d_bound + n;
  KeyLoc* d_loc = locs.as<KeyLoc>(); KeyLoc* d_kloc = d_loc + n;
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(k_group_keys, grid, block, 0, s, d_blob, blob_len, d_rec_off, d_rec_len, n, *o, d_keep, d_loc);
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_keep, d_kpos, (int)n, s);
  size_t tb2 = 0;
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb2, d_bound, d_gincl, (int)n, s);
  c->d_tiles.reserve(std::max(tb, tb2) + 64);      // scan workspace (free between the column-job launches of the general path)
  hip_check(hipcub::DeviceScan::ExclusiveSum(c->d_tiles.p, tb, d_keep, d_kpos, (int)n, s), "scan keep");
  uint32_t last[2];
  hip_check(hipMemcpyAsync(&last[0], d_kpos + (n - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(&last[1], d_keep + (n - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  const uint32_t nk = last[0] + last[1];
  *n_kept = nk;
  if (nk == 0) { uint32_t z = 0; hip_check(hipMemcpyAsync(d_grp_first, &z, 4, hipMemcpyHostToDevice, s), "H2D"); hip_check(hipStreamSynchronize(s), "sync"); return 0; }
  hipLaunchKernelGGL(k_group_compact, grid, block, 0, s, d_rec_off, d_rec_len, n, d_keep, d_kpos, d_loc, d_out_off, d_out_len, d_kloc);
  const dim3 gk((nk + 255) / 256);
  hipLaunchKernelGGL(k_group_bounds, gk, block, 0, s, d_blob, d_kloc, nk, d_bound);
  hip_check(hipcub::DeviceScan::InclusiveSum(c->d_tiles.p, tb2, d_bound, d_gincl, (int)nk, s), "scan bounds");
  hipLaunchKernelGGL(k_group_firsts, gk, block, 0, s, d_bound, d_gincl, nk, d_grp_first);
  uint32_t ng = 0;
  hip_check(hipMemcpyAsync(&ng, d_gincl + (nk - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "grouping kernels");
  *n_grp = ng;
  return 0;
}

}  // namespace fgx
