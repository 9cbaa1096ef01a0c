// hostemu.cpp — TEST INFRASTRUCTURE ONLY.  Not part of the product, never loaded by it.
//
// The general path's host orchestration (fgumi_amd/csrc/simplex_host.cpp, duplex_host.cpp, codec_host.cpp: source-read preparation, annotation jobs,
// the duplex strand combine, record assembly) linked against a stand-in for `fgx_caller::run_columns` that walks the staged jobs on
// the host, lane by lane: the annotation job through the product's own host + device source (methylation_core.h), the column job
// through consensus_math.h's ColumnAcc / column_call — the functions the kernels call.  There is no GPU where the CPU suite runs; this
// lets `-m "not gpu"` tests drive the orchestration end to end against the oracle.  The kernels themselves, their launches and the
// copies around them are what the `-m gpu` tests (tests/test_gpu_methylation.py, through libfgumi_amd.so) cover.
#include <cstring>
#include <stdexcept>
#include "../../fgumi_amd/csrc/bamrec.h"
#include "../../fgumi_amd/csrc/engine.h"

using namespace fgx;

namespace fgx {
void hip_check(hipError_t e, const char* what) { if (e != hipSuccess) throw std::runtime_error(what); }
void DevBuf::reserve(size_t) {}
void DevBuf::free_() {}
void PinnedBuf::reserve(size_t) {}
void PinnedBuf::free_() {}
}  // namespace fgx

#include "column_emu.h"
static void column_position(const ColumnBatch& b, const DeviceTables& T, ColParams prm, const ColJob& j, uint32_t p, uint8_t* ob, uint8_t* oq, uint16_t* od, uint16_t* oe) {
  emu::column_position(b.stage.data(), b.reads.data(), T, prm, j, p, ob, oq, od, oe);
}

static std::vector<uint8_t> g_genome_host;   // the bytes a GenomeRef would hold in HBM (one emulated caller at a time)

double fgx_caller::run_columns(ColumnBatch& b, ColParams prm) {
  b.ob.assign(b.n_cols, 0); b.oq.assign(b.n_cols, 0); b.od.assign(b.n_cols, 0); b.oe.assign(b.n_cols, 0);
  b.mflag.clear(); b.mu.clear(); b.mt.clear();
  if (b.jobs.empty() || b.n_cols == 0) return 0.0;
  const bool meth = !b.mjobs.empty() && b.n_mpos > 0 && genome;
  if (meth) {
    b.mflag.assign(b.n_mpos, 0); b.mu.assign(b.n_mpos, 0); b.mt.assign(b.n_mpos, 0);
    std::vector<uint8_t> dev_stage = b.stage;          // the device copy: the host copy changes only when the path asks for it back
    for (const MethJob& j : b.mjobs)
      for (uint32_t p = 0; p < j.n_pos; p++)
        meth_annotate_position(dev_stage.data(), b.reads.data() + j.rd0, j.n_reads, b.mruns.data() + j.run0, j.n_runs, g_genome_host.data() + j.contig_off, j.contig_len,
                               j.top != 0, p, &b.mflag[j.out_off + p], &b.mu[j.out_off + p], &b.mt[j.out_off + p]);
    std::vector<uint8_t> host_stage;
    if (!b.want_stage_back) host_stage = b.stage;
    b.stage.swap(dev_stage);
    for (const ColJob& j : b.jobs) for (uint32_t p = 0; p < j.cons_len; p++) column_position(b, h_tables, prm, j, p, b.ob.data(), b.oq.data(), b.od.data(), b.oe.data());
    if (!b.want_stage_back) b.stage.swap(host_stage);
    return 0.0;
  }
  for (const ColJob& j : b.jobs) for (uint32_t p = 0; p < j.cons_len; p++) column_position(b, h_tables, prm, j, p, b.ob.data(), b.oq.data(), b.od.data(), b.oe.data());
  return 0.0;
}

extern "C" {

fgx_caller* hemu_create(const fgx_options* opts) {
  if (!opts || opts->struct_size != sizeof(fgx_options)) return nullptr;
  fgx_caller* c = new fgx_caller();
  c->opt = *opts;
  c->prefix = opts->read_name_prefix ? opts->read_name_prefix : "";
  c->rg = opts->read_group_id ? opts->read_group_id : "A";
  c->opt.read_name_prefix = nullptr; c->opt.read_group_id = nullptr;
  memset(&c->h_tables, 0, sizeof(c->h_tables));
  memset(&c->h_umi_tables, 0, sizeof(c->h_umi_tables));
  build_tables(c->h_tables.t, opts->error_rate_pre_umi, opts->error_rate_post_umi, opts->tie_rule);
  build_single_input_quals(c->h_tables.single_input_quals, opts->error_rate_pre_umi, opts->error_rate_post_umi);
  build_tables(c->h_umi_tables.t, 90, 90, FGX_TIE_FGBIO_COMPAT);
  build_single_input_quals(c->h_umi_tables.single_input_quals, 90, 90);
  return c;
}
void hemu_destroy(fgx_caller* c) { delete c; }
const char* hemu_last_error(const fgx_caller* c) { return c->err.c_str(); }
int hemu_set_reference(fgx_caller* c, uint32_t n_ref, const uint8_t* const* seqs, const uint64_t* lens) {
  if (n_ref == 0) { c->genome.reset(); g_genome_host.clear(); return 0; }
  auto g = std::make_shared<GenomeRef>();
  uint64_t total = 0;
  for (uint32_t i = 0; i < n_ref; i++) { g->off.push_back(total); g->len.push_back(lens[i]); total += lens[i]; }
  g_genome_host.assign(total + 64, 0);
  for (uint32_t i = 0; i < n_ref; i++) if (lens[i]) memcpy(g_genome_host.data() + g->off[i], seqs[i], lens[i]);
  c->genome = g;
  return 0;
}
int hemu_process_batch(fgx_caller* c, const uint8_t* records, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec, const uint32_t* grp_first, uint32_t n_grp,
                       fgx_output* out) {
  c->err.clear();
  try {
    if (c->opt.caller_kind == FGX_CALLER_SIMPLEX) return simplex_process_general(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
    if (c->opt.caller_kind == FGX_CALLER_DUPLEX) return duplex_process_general(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
    if (c->opt.caller_kind == FGX_CALLER_CODEC) return codec_process_general(c, records, rec_off, rec_len, n_rec, grp_first, n_grp, out);
    c->err = "hostemu: caller kind not covered";
    return 1;
  } catch (const std::exception& ex) { c->err = ex.what(); return 3; }
}

}  // extern "C"
