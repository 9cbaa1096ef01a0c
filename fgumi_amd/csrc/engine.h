// engine.h — internal declarations shared by the host side and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "../../include/fgumi_amd.h"
#include "consensus_math.h"
#include "methylation_core.h"
#include <memory>

// Measurement-only switches — launch shapes, pacing, wavefronts per workgroup, verbose / debug prints — exist only in PROFILING builds
// (`python -m fgumi_amd.build --variant knobs -DFGX_KNOBS=1`, then FGX_LIB=fgumi_amd/variant_knobs.so): fgx_knob() is a constant nullptr in the
// product library, which therefore reads none of them (23 of the 37 environment variables of round 5).  What users and
// tests switch — the opt-outs of DESIGN §3 (FGX_SPLIT, FGX_DIRECT, FGX_S2_PACKED, FGX_SPLIT_CHUNKS, FGX_DEEP, FGX_METH_DEVICE, FGX_POOL_*,
// the canonical passes, FGX_INFL_TWO_PHASE, ...) — stays with getenv.
#ifndef FGX_KNOBS
#define FGX_KNOBS 0
#endif
inline const char* fgx_knob(const char* name) { return FGX_KNOBS ? getenv(name) : nullptr; }

namespace fgx {

// ---- device-visible job descriptions (general path: SourceReads staged by the host) ------------
struct ReadDesc {      // one SourceRead: bases at stage[off, off+len), quals at stage[off+len, off+2*len)
  uint64_t off;
  uint32_t len;
  uint32_t _pad;
};
struct ColJob {        // one single-strand consensus call (vanilla_caller.rs:1652-1755)
  uint32_t rd0;        // first ReadDesc
  uint32_t n_reads;
  uint32_t cons_len;   // consensus length = min_reads-th longest source read
  uint32_t out_off;    // first output column of this job in the SoA result arrays
};
struct Tile {          // 64 consecutive columns of one job = one wavefront
  uint32_t job;
  uint32_t p0;
};
struct ColParams {
  uint32_t min_reads;              // depth < min_reads → ('N', 0)
  uint32_t min_consensus_base_quality;  // qual < this → ('N', 2)
};
struct DeviceTables {
  ConsensusTables t;
  uint8_t single_input_quals[96];  // vanilla_caller.rs:469-501 (94 used)
};

struct DevBuf {  // grow-only device allocation
  void* p = nullptr;
  size_t cap = 0;
  void reserve(size_t n);
  void free_();
  template <class T> T* as() { return (T*)p; }
};
struct PinnedBuf {  // grow-only pinned host allocation
  void* p = nullptr;
  size_t cap = 0;
  void reserve(size_t n);
  void free_();
  template <class T> T* as() { return (T*)p; }
};

void hip_check(hipError_t e, const char* what);

// The device paths that were opt-in in round 3 (canonical second pass for indel duplex / CODEC molecules, its device kernel, the pass
// inside the device-resident entry, the --rejects side kernels, subset resubmission in the streaming pipeline) are the DEFAULT since
// round 4; the environment only opts OUT: a path is off when its own switch is "0", or, without one, when FGX_OPT_IN_ALL=0 turns every
// one of them off (the round-3 behaviour, kept for A/B measurements and the opt-out tests).
inline bool opt_in(const char* name) {
  const char* e = getenv(name);
  if (e && e[0]) return e[0] != '0';
  const char* a = getenv("FGX_OPT_IN_ALL");
  return !(a && a[0] == '0');
}

// kernels.hip
void launch_column_jobs(hipStream_t s, const uint8_t* d_stage, const ReadDesc* d_reads, const ColJob* d_jobs, const Tile* d_tiles,
                        uint32_t n_tiles, const DeviceTables* d_tables, ColParams prm, uint8_t* d_ob, uint8_t* d_oq, uint16_t* d_od,
                        uint16_t* d_oe);
void launch_meth_annotate(hipStream_t s, uint8_t* d_stage, const ReadDesc* d_reads, const MethJob* d_jobs, const MethRun* d_runs, const MethTile* d_tiles,
                          uint32_t n_tiles, const uint8_t* d_genome, uint8_t* d_flag, uint32_t* d_unconverted, uint32_t* d_converted);
void launch_libm_test(hipStream_t s, int op, const double* d_x, double* d_y, uint64_t n);
void launch_sim_generate(hipStream_t s, fgx_sim_params p, const uint64_t* d_fam_byte_off, const uint32_t* d_fam_rec_first,
                         uint8_t* d_blob, uint64_t* d_rec_off, uint32_t* d_rec_len, uint32_t* d_grp_first);

// One batch of single-strand consensus jobs, staged on the host and run on the device.
struct ColumnBatch {
  std::vector<uint8_t> stage;
  std::vector<ReadDesc> reads;
  std::vector<ColJob> jobs;
  uint32_t n_cols = 0;
  // results (host copies)
  std::vector<uint8_t> ob, oq;
  std::vector<uint16_t> od, oe;
  // methylation-aware mode: annotation jobs run before the column jobs (they normalise the staged bases in place)
  std::vector<MethJob> mjobs;
  std::vector<MethRun> mruns;
  uint32_t n_mpos = 0;
  bool want_stage_back = false;          // duplex: the error recount reads the NORMALISED source reads on the host
  std::vector<uint8_t> mflag;            // results: is_ref_c, unconverted, converted per annotated position
  std::vector<uint32_t> mu, mt;
  void clear() { stage.clear(); reads.clear(); jobs.clear(); n_cols = 0; mjobs.clear(); mruns.clear(); n_mpos = 0; want_stage_back = false; }
  // One annotate_and_normalize call over the contiguous ReadDescs [rd0, rd0 + n_reads); returns the job id.
  uint32_t add_meth_job(uint32_t rd0, uint32_t n_reads, uint32_t n_pos, const std::vector<MethRun>& runs, bool top, uint64_t contig_off, uint64_t contig_len) {
    MethJob j; j.rd0 = rd0; j.n_reads = n_reads; j.n_pos = n_pos; j.out_off = n_mpos; j.run0 = (uint32_t)mruns.size(); j.n_runs = (uint32_t)runs.size();
    j.top = top ? 1u : 0u; j._pad = 0; j.contig_off = contig_off; j.contig_len = contig_len;
    mruns.insert(mruns.end(), runs.begin(), runs.end());
    n_mpos += n_pos;
    mjobs.push_back(j);
    return (uint32_t)mjobs.size() - 1;
  }
  // Appends a source read; returns its ReadDesc index.
  uint32_t add_read(const uint8_t* bases, const uint8_t* quals, uint32_t len) {
    ReadDesc d; d.off = stage.size(); d.len = len; d._pad = 0;
    stage.insert(stage.end(), bases, bases + len);
    stage.insert(stage.end(), quals, quals + len);
    reads.push_back(d);
    return (uint32_t)reads.size() - 1;
  }
  uint32_t add_job(uint32_t rd0, uint32_t n_reads, uint32_t cons_len) {
    ColJob j; j.rd0 = rd0; j.n_reads = n_reads; j.cons_len = cons_len; j.out_off = n_cols;
    n_cols += cons_len;
    jobs.push_back(j);
    return (uint32_t)jobs.size() - 1;
  }
};

// filter.hip — device buffers of the consensus-read filter (grow-only, owned by the caller object)
struct FilterBuffers {
  DevBuf pass, masked, newt, incl, first, ord_src, keep_size, rej_size, keep_off, rej_off, misc, scan_tmp, out_keep, out_rej, in_blob, in_off, in_len, slot_flag, slot_pos;
  DevBuf aln_len, aln_contigs;      // --ref: lengths of the records with regenerated NM / UQ / MD; contig offsets | lengths of the caller's genome
  PinnedBuf pin_keep, pin_rej;
  uint32_t lds_slice = 0;           // LDS bytes per wavefront for the staged record; 0 = sized from the mean record length
  void release() {
    for (DevBuf* b : {&pass, &masked, &newt, &incl, &first, &ord_src, &keep_size, &rej_size, &keep_off, &rej_off, &misc, &scan_tmp, &out_keep, &out_rej,
                      &in_blob, &in_off, &in_len, &slot_flag, &slot_pos, &aln_len, &aln_contigs})
      b->free_();
    pin_keep.free_(); pin_rej.free_();
  }
};

}  // namespace fgx

namespace fgx {
// The reference genome of the methylation-aware mode, resident in HBM (fgx_set_reference): contig i of the BAM header at
// [off[i], off[i] + len[i]) of `d_genome`.  Shared (read-only) with the helper callers of the multi-threaded general path.
struct GenomeRef {
  int device = 0;
  DevBuf d_genome;
  std::vector<uint64_t> off, len;
  ~GenomeRef() { (void)hipSetDevice(device); d_genome.free_(); }
};
}  // namespace fgx

// The caller object behind the C ABI.
struct FastState;
struct fgx_caller {
  fgx_options opt;
  std::string prefix, rg;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  fgx::DeviceTables h_tables;       // main tables (pre, post from options)
  fgx::DeviceTables h_umi_tables;   // consensus_umis tables (90, 90), simple_umi.rs:12-18
  fgx::DevBuf d_tables, d_umi_tables;
  fgx::DevBuf d_stage, d_reads, d_jobs, d_tiles, d_ob, d_oq, d_od, d_oe, d_scratch_a, d_scratch_b;
  fgx::DevBuf d_mjobs, d_mruns, d_mtiles, d_mflag, d_mu, d_mt;    // methylation annotation jobs and their results
  std::shared_ptr<fgx::GenomeRef> genome;                          // fgx_set_reference (null: no reference)
  fgx::ColumnBatch batch;
  // outputs of the last call
  std::vector<uint8_t> out_data, out_rejects;
  std::vector<uint64_t> grp_out_end;      // general path: cumulative out_data size after each group
  bool general_only = false;
  bool counter_names_used = false;          // CODEC general path: a read was named by the running counter (no MI): order-dependent
  std::vector<fgx_caller*> workers;         // helper callers (own stream and buffers) of the multi-threaded general path
  struct FastState* fast = nullptr;        // device-resident pipeline state (fastpath.hip)
  fgx::DevBuf d_in_blob, d_in_off, d_in_len, d_in_grp;   // host-input staging for fgx_process_batch
  fgx::DevBuf d_res_out1, d_res_off1, d_res_final, d_res_aux, d_res_aux2, d_res_deferred, d_res_scan, d_res_cdef, d_res_outoff;   // canonical second pass inside the device-resident entry
  fgx::DevBuf d_canon_aux, d_canon_slabs;   // device canonicalisation (canon_device.hip): slot tables / status / lengths, per-lane lists
  fgx::DevBuf d_canon_blob, d_canon_off, d_canon_len, d_canon_grp;   // canonical duplex molecules of the second device pass (canon_core.h)
  fgx::FilterBuffers* filt = nullptr;      // fgx_filter_records[_device] state (filter.hip)
  std::vector<uint8_t> rejects_host;       // host entry: the device-made rejects of the last batch (FGX_REJECTS_DEVICE=1)
  const uint64_t* last_group_off = nullptr; uint32_t last_group_stride = 0;   // device-resident entry: byte offset of group g in its record stream = last_group_off[g * stride] (device memory; nullptr: none)
  uint32_t last_reject_oos = 0;            // ... groups its side kernels could not decide (the batch then took the general path)
  void* rej_state = nullptr;               // buffers of the device `--rejects` side kernels (reject_device.hip: reject_release)
  void* pipe_state = nullptr;              // buffers of fgx_run_bam, kept from run to run (pipeline.cpp: fgx_pipeline_release)
  uint64_t last_deferred_groups = 0, last_canon_molecules = 0;   // host entry: groups the first device pass deferred / molecules the canonical second pass decided
  uint32_t last_boundary_rounds = 0;       // repair rounds of the last fgx_record_boundaries_device call (boundaries.hip; 0 = every guess was right)

  // Runs the staged column jobs of `b` on the device and fills b.ob/oq/od/oe. Returns kernel ms.
  double run_columns(fgx::ColumnBatch& b, fgx::ColParams prm);
};

namespace fgx {
// simplex_host.cpp — general path (any CIGAR, any family shape): host orchestration, device columns.
int simplex_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                            const uint32_t* grp_first, uint32_t n_grp, fgx_output* out);
// duplex_host.cpp / codec_host.cpp — general paths of the duplex and CODEC callers
int duplex_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                           const uint32_t* grp_first, uint32_t n_grp, fgx_output* out);
// grouping.hip — MI grouping on the device
// reject_device.hip: the simplex caller's `--rejects` stream from side kernels (reject_core.h, a lane per MI group)
namespace rej { struct Params; }
struct RejectResult { const uint8_t* d_out; uint64_t bytes, count; uint32_t n_out_of_scope; double ms; };
void simplex_rejects_device(fgx_caller* c, const rej::Params& P, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
                            const uint32_t* d_grp_first, uint32_t n_grp, RejectResult* r);
void strand_rejects_device(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
                           const uint32_t* d_grp_first, uint32_t n_grp, const uint64_t* d_group_off, uint32_t stride, uint64_t out_len, RejectResult* r);
void reject_release(fgx_caller* c);
int group_records_device(fgx_caller* c, const fgx_group_options* o, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off,
                         const uint32_t* d_rec_len, uint32_t n, uint64_t* d_out_off, uint32_t* d_out_len, uint32_t* d_grp_first, uint32_t* n_kept,
                         uint32_t* n_grp);
// boundaries.hip — FindBoundaries on the device
int record_boundaries_device(fgx_caller* c, const uint8_t* d_stream, uint64_t len, uint64_t start, uint64_t* d_rec_off, uint32_t* d_rec_len,
                             uint64_t cap, uint64_t* n_rec, uint64_t* consumed);
// bgzf_device.hip — BGZF inflate on the device: one descriptor per block (offsets into the compressed bytes / the inflated stream)
struct BgzfDevBlock { uint64_t in_off, out_off; uint32_t in_len, isize, crc, ent_off; };   // in_off / in_len: the raw DEFLATE payload; ent_off: bgzf_inflate_plan
// entries the tokenizer can write for a block of `isize` bytes (= inflate_core.h infl_entry_cap; restated here because the host-compiled test
// builds that include this header do not take inflate_core.h's compiler builtins — bgzf_device.hip asserts that the two agree)
constexpr uint32_t bgzf_entry_cap(uint32_t isize) { return (isize / 3u + isize / 255u + 2u + 15u) & ~15u; }
// Lays the blocks' entry lists out back to back, each sized by ITS ISIZE (round 6; round 5 gave every block the 64 KiB worst case — 90 KB
// of scratch per block whatever its size, tens of GB for a file of small blocks): fills ent_off and returns the bytes of scratch the
// two-phase form needs (entries, then one 32-bit list length per block), or 0 when the lists do not fit 32-bit offsets (one-phase form).
inline size_t bgzf_inflate_plan(BgzfDevBlock* blk, uint32_t n) {
  uint64_t total = 0;
  for (uint32_t i = 0; i < n; i++) { blk[i].ent_off = (uint32_t)total; total += bgzf_entry_cap(blk[i].isize); if (total > 0xFFFF0000ull) return 0; }
  return (size_t)total * 4u + (size_t)n * 4u + 64u;
}
// `d_scratch` (bgzf_inflate_plan's bytes of device memory, or null): the entry lists of the two-phase form (k_bgzf_tokenize + k_bgzf_resolve,
// the default with a scratch; FGX_INFL_TWO_PHASE=0 or no scratch: the one-phase kernel of rounds 2 - 4).  `scratch_bytes`: what the plan returned.
void bgzf_inflate_launch(hipStream_t s, const uint8_t* d_raw, const BgzfDevBlock* d_blk, uint32_t n, uint8_t* d_out, uint32_t* d_status, uint32_t* h_status_pinned,
                         void* d_scratch = nullptr, size_t scratch_bytes = 0);
bool bgzf_inflate_two_phase();
int bgzf_inflate_status(fgx_caller* c, uint32_t status_word);
void bgzf_crc_blocks_device(fgx_caller* c, const uint8_t* d_in, uint64_t len, uint32_t* d_crcs);
int bgzf_deflate_device(fgx_caller* c, const uint8_t* d_in, uint64_t len, DevBuf& slots, DevBuf& scratch, DevBuf& meta, DevBuf& packed, uint64_t* packed_len);
// api.cpp — fgx_run_bam's way of deciding ONLY the groups the device entry just deferred (general path on copies of their records) and
// merging them into the device's record stream on the host; -1 = not possible here, take the whole-batch way
int resubmit_deferred(fgx_caller* c, const uint8_t* d_blob, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec, const uint32_t* d_grp_first, uint32_t n_grp,
                      const fgx_output* dev, uint32_t n_def, const uint32_t* d_def, fgx_output* merged);
// FGX_STREAM_PRIORITY=1 (opt-in, for measurements): the streams the consensus kernels run on get the device's highest priority.  It takes them
// out of the hardware queues they share with fgx_run_bam's fill streams (a short consensus kernel queued behind a 30 ms inflate kernel cost the
// device stage of the bench process a third of its time: file -> file 50 -> 68 M raw reads/s) — but tests/test_gpu_pipeline.py's multi-chunk
// cases then produced groups cut in two (stale bytes in front of a chunk's stream), every time; until that is understood the streams stay at
// normal priority, where the whole GPU suite is green (profiles/r04_experiments.md).
inline void create_compute_stream(hipStream_t* s) {
  static const bool high = [] { const char* e = fgx_knob("FGX_STREAM_PRIORITY"); return e && e[0] == '1'; }();
  int least = 0, greatest = 0;
  if (high && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
    hip_check(hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest), "hipStreamCreateWithPriority");
    return;
  }
  hip_check(hipStreamCreateWithFlags(s, hipStreamNonBlocking), "hipStreamCreate");
}

// pipeline.cpp — frees what fgx_run_bam keeps in c->pipe_state
void pipeline_release(fgx_caller* c);
// filter.hip — `fgumi filter` on the device
int filter_records_device(fgx_caller* c, FilterBuffers& B, const fgx_filter_options* o, uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off,
                          const uint32_t* d_rec_len, uint32_t n, fgx_filter_output* out);
int filter_slots_device(fgx_caller* c, FilterBuffers& B, const fgx_filter_options* o, uint8_t* d_out, uint64_t out_len, const uint64_t* d_slot_off,
                        const uint64_t* d_slot_size, uint32_t n_slots, fgx_filter_output* out);
int codec_process_general(fgx_caller* c, const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec,
                          const uint32_t* grp_first, uint32_t n_grp, fgx_output* out);
}
