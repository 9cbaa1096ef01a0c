// host_reads.h — SourceRead construction and the alignment filter for the general (host-orchestrated)
// paths of the duplex and CODEC callers.  Reference: vanilla_caller.rs:1080-1190 (create_source_read),
// 1217-1296 (drop_unmapped_if_any_mapped + filter_source_reads_by_alignment), 706-779 (consensus_call).
#pragma once
#include "engine.h"
#include "host_common.h"

namespace fgx {

struct SrcRead {
  uint32_t orig_idx;   // index into the list the read was created from
  uint32_t rd;         // ReadDesc index in the ColumnBatch (bases at stage[off..), quals after them)
  uint32_t len;
  uint16_t flags;
  int32_t name_hash;
  SimpCigar cigar;
  int32_t ref_id = -1;       // methylation-aware mode: reference id, 0-based start, simplified CIGAR before reversal / truncation
  int64_t aln_start = -1;
  SimpCigar orig_cigar;
};

struct SrcParams {
  uint8_t min_bq;
  bool trim;
  bool want_rank;   // max_reads set → Murmur3 name rank, else 0
  bool meth = false;   // methylation-aware mode: keep the alignment of each source read
};

// create_source_read: orient, (trim), mask, mate-clip, strip trailing N; stages the read.  0 dropped / 1 ok / -1 fatal.
inline int make_source_read(ColumnBatch& B, const SrcParams& sp, const uint8_t* p, uint32_t n, uint32_t idx, uint64_t mate_clip, SrcRead& out,
                            std::vector<uint8_t>& tb, std::vector<uint8_t>& tq, std::string& err) {
  bam::Rec v{p, n};
  uint16_t flg = v.flags();
  bool neg = flg & bam::F_REVERSE;
  uint32_t read_len = v.l_seq();
  if (read_len == 0) return 0;
  if ((uint64_t)v.qual_off() + read_len > n) { err = "input read has invalid base qualities (length does not match sequence length): " + std::string((const char*)v.name(), v.name_len()); return -1; }
  const uint8_t* q = p + v.qual_off();
  bool all_ff = true;
  for (uint32_t i = 0; i < read_len; i++) if (q[i] != 0xFF) { all_ff = false; break; }
  if (all_ff) { err = "input read is missing base qualities (BAM QUAL is '*'): " + std::string((const char*)v.name(), v.name_len()); return -1; }
  tb.resize(read_len); tq.resize(read_len);
  if (neg) for (uint32_t i = 0; i < read_len; i++) { uint32_t s = read_len - 1 - i; tb[i] = bam::code_to_ascii(bam::code_complement(v.base_code(s))); tq[i] = q[s]; }
  else for (uint32_t i = 0; i < read_len; i++) { tb[i] = bam::code_to_ascii(v.base_code(i)); tq[i] = q[i]; }
  uint32_t trim_to = sp.trim ? quality_trim_point(tq.data(), read_len, sp.min_bq) : read_len;
  for (uint32_t i = 0; i < trim_to; i++) if (tq[i] < sp.min_bq) { tb[i] = 'N'; tq[i] = FGX_MIN_PHRED; }
  uint64_t clip_position = read_len > mate_clip ? read_len - mate_clip : 0;
  uint32_t final_len = (uint32_t)std::min<uint64_t>(clip_position, trim_to);
  while (final_len > 0 && tb[final_len - 1] == 'N') final_len--;
  if (final_len == 0) return 0;
  out.orig_idx = idx; out.len = final_len; out.flags = flg;
  out.rd = B.add_read(tb.data(), tq.data(), final_len);
  SimpCigar sc = simplify_cigar(v);
  if (sp.meth) { out.ref_id = v.ref_id(); out.aln_start = (int64_t)v.pos(); out.orig_cigar = sc; }
  if (neg) std::reverse(sc.begin(), sc.end());
  out.cigar = truncate_cigar(sc, final_len);
  out.name_hash = sp.want_rank ? read_name_rank(v.name(), v.name_len()) : 0;
  return 1;
}

// filter_by_alignment: unmapped reads dropped when any read is mapped, then the most common
// prefix-compatible CIGAR group; survivors keep input order.  Rejected orig_idx are appended.
inline void filter_by_alignment(std::vector<SrcRead>& srs, HostStats& st, std::vector<uint32_t>& rejected_orig) {
  bool any_un = false, all_un = true;
  for (auto& s : srs) { bool u = s.flags & bam::F_UNMAPPED; any_un |= u; all_un &= u; }
  if (any_un && !all_un) {
    std::vector<SrcRead> kept;
    size_t dropped = 0;
    for (auto& s : srs) { if (s.flags & bam::F_UNMAPPED) { dropped++; rejected_orig.push_back(s.orig_idx); } else kept.push_back(std::move(s)); }
    st.reject(FGX_REJ_UNMAPPED, dropped);
    srs.swap(kept);
  }
  if (srs.size() < 2) return;
  std::vector<uint32_t> order(srs.size());
  for (uint32_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return srs[a].len > srs[b].len; });
  std::vector<const SimpCigar*> cigs;
  for (uint32_t i : order) cigs.push_back(&srs[i].cigar);
  std::vector<uint32_t> keep_sorted = most_common_alignment_group(cigs);
  std::vector<bool> keep(srs.size(), false);
  for (uint32_t k : keep_sorted) keep[order[k]] = true;
  size_t n_keep = 0;
  for (bool b : keep) n_keep += b;
  size_t rejected = srs.size() - n_keep;
  if (rejected) {
    st.reject(FGX_REJ_MINORITY_ALIGNMENT, rejected);
    std::vector<SrcRead> kept;
    for (size_t i = 0; i < srs.size(); i++) { if (keep[i]) kept.push_back(std::move(srs[i])); else rejected_orig.push_back(srs[i].orig_idx); }
    srs.swap(kept);
  }
}

// consensus_call (vanilla_caller.rs:706-779) with the single-strand settings the duplex / CODEC callers use
// (min_reads = 1): optional per-strand cap by name rank, then one column job over the (capped) reads with
// consensus length = the longest read.  Returns the job id or -1 for None.
// With a genome (methylation-aware mode) the call is annotated first, over ALL its reads (the cap shapes the consensus only,
// vanilla_caller.rs:715-724): *meth_job = the annotation job, or -1 (annotate_and_normalize returned None).
inline int64_t stage_consensus_call(ColumnBatch& B, const std::vector<SrcRead>& srs, int64_t max_reads, const GenomeRef* genome = nullptr, int64_t* meth_job = nullptr) {
  if (meth_job) *meth_job = -1;
  if (srs.empty()) return -1;
  if (genome && meth_job) {
    size_t a = 0;   // max_by_key: the LAST longest read
    for (size_t i = 1; i < srs.size(); i++) if (srs[i].len >= srs[a].len) a = i;
    const SrcRead& an = srs[a];
    if (an.ref_id >= 0 && an.aln_start >= 0 && (size_t)an.ref_id < genome->len.size()) {
      std::vector<MethRun> runs;
      meth_runs(an.cigar, an.aln_start, (an.flags & bam::F_REVERSE) != 0, an.orig_cigar, runs);
      uint32_t rd0 = (uint32_t)B.reads.size();
      for (auto& s : srs) B.reads.push_back(B.reads[s.rd]);   // descriptors only
      *meth_job = (int64_t)B.add_meth_job(rd0, (uint32_t)srs.size(), an.len, runs, meth_is_top_strand(an.flags), genome->off[(size_t)an.ref_id], genome->len[(size_t)an.ref_id]);
    }
  }
  std::vector<const SrcRead*> use;
  if (max_reads >= 0 && srs.size() > (size_t)max_reads) {
    std::vector<uint32_t> idx(srs.size());
    for (uint32_t i = 0; i < idx.size(); i++) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return srs[a].name_hash < srs[b].name_hash; });
    idx.resize((size_t)max_reads);
    for (uint32_t i : idx) use.push_back(&srs[i]);
  } else for (auto& s : srs) use.push_back(&s);
  if (use.empty()) return -1;
  uint32_t cons_len = 0;
  for (auto* s : use) cons_len = std::max(cons_len, s->len);
  uint32_t rd0 = (uint32_t)B.reads.size();
  for (auto* s : use) B.reads.push_back(B.reads[s->rd]);   // descriptors only; staged bytes are shared
  return (int64_t)B.add_job(rd0, (uint32_t)use.size(), cons_len);
}

}  // namespace fgx
