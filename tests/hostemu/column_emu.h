// column_emu.h — TEST INFRASTRUCTURE ONLY: what k_column_jobs does for one position (fgumi_amd/csrc/kernels.hip), on the host, through the
// functions the kernel calls (consensus_math.h's ColumnAcc / column_call).  Shared by tests/hostemu and tests/apiemu.
#pragma once
#include "../../fgumi_amd/csrc/bamrec.h"
#include "../../fgumi_amd/csrc/engine.h"

namespace emu {
using namespace fgx;

inline void column_position(const uint8_t* stage, const ReadDesc* reads, const DeviceTables& T, ColParams prm, const ColJob& j, uint32_t p, uint8_t* ob, uint8_t* oq,
                            uint16_t* od, uint16_t* oe) {

  const uint32_t o = j.out_off + p;
  if (j.n_reads == 1) {
    const ReadDesc& rd = reads[j.rd0];
    uint8_t raw = stage[rd.off + p];
    uint32_t qi = stage[rd.off + rd.len + p];
    uint8_t adj = qi < 94 ? T.single_input_quals[qi] : 0;
    if (adj < prm.min_consensus_base_quality) { ob[o] = 'N'; oq[o] = FGX_MIN_PHRED; } else { ob[o] = raw; oq[o] = adj; }
    od[o] = raw != 'N' ? 1 : 0;
    oe[o] = 0;
    return;
  }
  ColumnAcc acc;
  acc.reset();
  for (uint32_t r = 0; r < j.n_reads; r++) {
    const ReadDesc& rd = reads[j.rd0 + r];
    if (p >= rd.len) continue;
    uint8_t base = stage[rd.off + p];
    if (base == 'N') continue;
    int idx = bam::ascii_to_lane(base);
    if (idx == 255) continue;
    uint32_t q = stage[rd.off + rd.len + p];
    q = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
    acc.add(idx, T.t.correct[q], T.t.error_per_alt[q]);
  }
  int bi;
  uint8_t q;
  column_call(T.t, acc.s, acc.obs, &bi, &q);
  uint32_t depth = acc.contributions(), err = depth - acc.obs_of(bi);
  od[o] = (uint16_t)(depth < 32767u ? depth : 32767u);
  oe[o] = (uint16_t)(err < 32767u ? err : 32767u);
  uint8_t base = bi >= 0 ? (uint8_t)"ACGT"[bi] : (uint8_t)'N';
  if (depth < prm.min_reads) { ob[o] = 'N'; oq[o] = 0; }
  else if (q < prm.min_consensus_base_quality) { ob[o] = 'N'; oq[o] = FGX_MIN_PHRED; }
  else { ob[o] = base; oq[o] = q; }
}

}  // namespace emu
