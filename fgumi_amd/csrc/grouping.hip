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
#include <hipcub/hipcub.hpp>
#include "bamrec.h"
#include "engine.h"

namespace fgx {

namespace {

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
      const int64_t m = bam::find_z_tag(v.b + aux, an, (uint8_t)o.tag[0], (uint8_t)o.tag[1], &vl);
      if (m >= 0) {
        k = 1;
        if (o.strip_strand_suffix) {          // extract_mi_base: everything before the last '/', unless that '/' leads the value
          const uint8_t* s = v.b + aux + m;
          for (uint32_t i = vl; i-- > 1;) if (s[i] == '/') { vl = i; break; }
        }
        L.mi_off = off + aux + (uint64_t)m; L.mi_len = vl;
        if (o.cell_tag[0]) {
          uint32_t cl = 0;
          const int64_t cpos = bam::find_z_tag(v.b + aux, an, (uint8_t)o.cell_tag[0], (uint8_t)o.cell_tag[1], &cl);
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
      bool same = true;
      for (uint32_t i = 0; i < B.mi_len && same; i++) same = blob[A.mi_off + i] == blob[B.mi_off + i];
      for (uint32_t i = 0; i < B.cb_len && same; i++) same = blob[A.cb_off + i] == blob[B.cb_off + i];
      if (same) b = 0;
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
