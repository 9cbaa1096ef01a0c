// kernels.hip — HIP kernels for gfx950 (MI355X).  Compile with -ffp-contract=off: the f64
// Kahan loop and the margin arithmetic must not be contracted into FMAs (bit-exact contract).
#include "engine.h"
#include "bamrec.h"
#include "simgen.h"

namespace fgx {

// -----------------------------------------------------------------------------------------
// K1 (general path): one wavefront = 64 consecutive consensus positions of one job; each lane
// owns one column and walks the job's source reads serially, in retained input order
// (summation order is observable).  Loads are coalesced along the read (lane p reads byte p).
// Bound: HBM/L2 streaming of 2 B per observation vs 16 dependent f64 add/sub per observation.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_column_jobs(const uint8_t* __restrict__ stage, const ReadDesc* __restrict__ reads,
                                                     const ColJob* __restrict__ jobs, const Tile* __restrict__ tiles, uint32_t n_tiles,
                                                     const DeviceTables* __restrict__ T, ColParams prm, uint8_t* __restrict__ ob,
                                                     uint8_t* __restrict__ oq, uint16_t* __restrict__ od, uint16_t* __restrict__ oe) {
  uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  if (wave >= n_tiles) return;
  Tile t = tiles[wave];
  ColJob j = jobs[t.job];
  uint32_t p = t.p0 + lane;
  if (p >= j.cons_len) return;
  uint32_t o = j.out_off + p;

  if (j.n_reads == 1) {  // single-read family: LUT path (vanilla_caller.rs:1677-1708)
    ReadDesc rd = reads[j.rd0];
    uint8_t raw_base = stage[rd.off + p];
    uint32_t qi = stage[rd.off + rd.len + p];
    uint8_t adj = qi < 94 ? T->single_input_quals[qi] : 0;
    if (adj < prm.min_consensus_base_quality) { ob[o] = 'N'; oq[o] = FGX_MIN_PHRED; }
    else { ob[o] = raw_base; oq[o] = adj; }
    od[o] = raw_base != 'N' ? 1 : 0;
    oe[o] = 0;
    return;
  }

  ColumnAcc acc;
  acc.reset();
  for (uint32_t r = 0; r < j.n_reads; r++) {
    ReadDesc rd = reads[j.rd0 + r];
    if (p < rd.len) {
      uint8_t base = stage[rd.off + p];
      if (base != 'N') {
        int idx = bam::ascii_to_lane(base);
        if (idx != 255) {
          uint32_t q = stage[rd.off + rd.len + p];
          q = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
          acc.add(idx, T->t.correct[q], T->t.error_per_alt[q]);
        }
      }
    }
  }
  int bi;
  uint8_t q;
  column_call(T->t, acc.s, acc.obs, &bi, &q);
  uint32_t depth = acc.contributions();
  uint32_t match = acc.obs_of(bi);
  uint32_t err = depth - match;
  od[o] = (uint16_t)(depth < 32767u ? depth : 32767u);
  oe[o] = (uint16_t)(err < 32767u ? err : 32767u);
  const char B[4] = {'A', 'C', 'G', 'T'};
  uint8_t base = bi >= 0 ? (uint8_t)B[bi] : (uint8_t)'N';
  if (depth < prm.min_reads) { ob[o] = 'N'; oq[o] = 0; }
  else if (q < prm.min_consensus_base_quality) { ob[o] = 'N'; oq[o] = FGX_MIN_PHRED; }
  else { ob[o] = base; oq[o] = q; }
}

void launch_column_jobs(hipStream_t s, const uint8_t* d_stage, const ReadDesc* d_reads, const ColJob* d_jobs, const Tile* d_tiles,
                        uint32_t n_tiles, const DeviceTables* d_tables, ColParams prm, uint8_t* d_ob, uint8_t* d_oq, uint16_t* d_od,
                        uint16_t* d_oe) {
  if (n_tiles == 0) return;
  uint32_t blocks = (n_tiles + 3) / 4;
  hipLaunchKernelGGL(k_column_jobs, dim3(blocks), dim3(256), 0, s, d_stage, d_reads, d_jobs, d_tiles, n_tiles, d_tables, prm, d_ob, d_oq,
                     d_od, d_oe);
}

// -----------------------------------------------------------------------------------------
// Methylation-aware mode (general path): one wavefront = 64 consecutive positions of one annotation job (one
// annotate_and_normalize call, vanilla_caller.rs:781-860); a lane looks its position's reference base up in the genome (resident
// in HBM) through the anchor's aligned runs, counts unconverted / converted bases down the call's source reads (coalesced along
// the read, like the column kernel) and rewrites converted bases in the staged bytes.  Runs BEFORE k_column_jobs on the stream.
// Bound: HBM / L2 streaming of 1 B per source base; the genome reads are 64 consecutive bytes per wavefront.
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_meth_annotate(uint8_t* __restrict__ stage, const ReadDesc* __restrict__ reads, const MethJob* __restrict__ jobs,
                                                       const MethRun* __restrict__ runs, const MethTile* __restrict__ tiles, uint32_t n_tiles,
                                                       const uint8_t* __restrict__ genome, uint8_t* __restrict__ flag, uint32_t* __restrict__ unconverted,
                                                       uint32_t* __restrict__ converted) {
  uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  if (wave >= n_tiles) return;
  MethTile t = tiles[wave];
  MethJob j = jobs[t.job];
  uint32_t p = t.p0 + lane;
  if (p >= j.n_pos) return;
  uint32_t o = j.out_off + p;
  meth_annotate_position(stage, reads + j.rd0, j.n_reads, runs + j.run0, j.n_runs, genome + j.contig_off, j.contig_len, j.top != 0, p, &flag[o], &unconverted[o],
                         &converted[o]);
}
void launch_meth_annotate(hipStream_t s, uint8_t* d_stage, const ReadDesc* d_reads, const MethJob* d_jobs, const MethRun* d_runs, const MethTile* d_tiles,
                          uint32_t n_tiles, const uint8_t* d_genome, uint8_t* d_flag, uint32_t* d_unconverted, uint32_t* d_converted) {
  if (n_tiles == 0) return;
  uint32_t blocks = (n_tiles + 3) / 4;
  hipLaunchKernelGGL(k_meth_annotate, dim3(blocks), dim3(256), 0, s, d_stage, d_reads, d_jobs, d_runs, d_tiles, n_tiles, d_genome, d_flag, d_unconverted, d_converted);
}

// -----------------------------------------------------------------------------------------
// Device self-test of the glibc-compatible libm.
// -----------------------------------------------------------------------------------------
__global__ void k_libm(int op, const double* __restrict__ x, double* __restrict__ y, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], r;
  switch (op) {
    case 0: r = g_exp(v); break;
    case 1: r = g_log(v); break;
    case 2: r = g_log1p(v); break;
    default: r = g_expm1(v); break;
  }
  y[i] = r;
}
void launch_libm_test(hipStream_t s, int op, const double* d_x, double* d_y, uint64_t n) {
  if (!n) return;
  hipLaunchKernelGGL(k_libm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op, d_x, d_y, n);
}

// -----------------------------------------------------------------------------------------
// Synthetic grouped reads straight into HBM (one thread per family; setup, not timed).
// -----------------------------------------------------------------------------------------
__global__ void k_sim_generate(fgx_sim_params p, const uint64_t* __restrict__ fam_byte_off, const uint32_t* __restrict__ fam_rec_first,
                               uint8_t* blob, uint64_t* rec_off, uint32_t* rec_len, uint32_t* grp_first) {
  uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= p.n_families) return;
  sim::write_family(p, f, fam_byte_off[f], fam_rec_first[f], blob, rec_off, rec_len, grp_first);
}
void launch_sim_generate(hipStream_t s, fgx_sim_params p, const uint64_t* d_fam_byte_off, const uint32_t* d_fam_rec_first,
                         uint8_t* d_blob, uint64_t* d_rec_off, uint32_t* d_rec_len, uint32_t* d_grp_first) {
  if (!p.n_families) return;
  hipLaunchKernelGGL(k_sim_generate, dim3((p.n_families + 63) / 64), dim3(64), 0, s, p, d_fam_byte_off, d_fam_rec_first, d_blob,
                     d_rec_off, d_rec_len, d_grp_first);
}

}  // namespace fgx
