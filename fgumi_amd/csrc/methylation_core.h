// methylation_core.h — the methylation-aware mode (EM-Seq / TAPs) of the simplex and duplex callers.
//
// Reference: crates/fgumi-consensus/src/methylation.rs:116-178 (query_to_ref_positions), 193-242 (annotate_simplex_methylation),
// 264-343 (build_mm_ml_tags), 392-398 (is_top_strand), 404-427 (combine_methylation_annotations);
// vanilla_caller.rs:781-860 (annotate_and_normalize).
//
// Split of the work: the host knows a call's anchor read (the longest source read) and turns its simplified CIGAR into ALIGNED
// RUNS (query start, length, reference position of the first base, step +1 / -1) — a handful of integers per call; the per-base
// work runs on the device with the genome resident in HBM: a lane per annotated position looks its reference base up through the
// runs, counts unconverted / converted bases down the call's staged source reads and rewrites converted bases to the unconverted
// one IN the staged bytes, so the column kernel that runs next on the stream calls the normalised reads.  The per-position body is
// host + device source: the CPU tests run it lane by lane (fgx_methylation_annotate_host).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#if defined(__HIPCC__)
#define FGX_METH_HD __host__ __device__
#else
#define FGX_METH_HD
#endif

namespace fgx {

struct MethRun {           // positions [q0, q0 + len) of the anchor are aligned to ref0, ref0 + step, ...
  int64_t q0, len, ref0, step;
};
struct MethRead {          // a staged source read: bases at stage[off, off + len)
  uint64_t off;
  uint32_t len, _pad;
};
struct MethJob {           // one annotate_and_normalize call
  uint32_t rd0, n_reads;   // the call's source reads: contiguous read descriptors (ALL of them: the per-strand cap applies to the consensus only)
  uint32_t n_pos;          // annotated positions = the anchor's length
  uint32_t out_off;        // first slot of this job in the flag / count arrays
  uint32_t run0, n_runs;   // the anchor's aligned runs
  uint32_t top;            // is_top_strand(anchor.flags): C / T against reference C; otherwise G / A against reference G
  uint32_t _pad;
  uint64_t contig_off, contig_len;   // the anchor's contig inside the genome buffer
};
struct MethTile { uint32_t job, p0; };   // 64 consecutive positions of one job = one wavefront

FGX_METH_HD inline uint8_t meth_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// fetch_ref_bases_at_positions ∘ query_to_ref_positions for ONE query position: 0 = None (insertion, outside the contig)
FGX_METH_HD inline uint8_t meth_ref_base(const MethRun* runs, uint32_t n_runs, const uint8_t* contig, uint64_t contig_len, uint32_t i) {
  for (uint32_t r = 0; r < n_runs; r++) {
    const int64_t q0 = runs[r].q0, len = runs[r].len;
    if ((int64_t)i >= q0 && (int64_t)i < q0 + len) {
      const int64_t p = runs[r].ref0 + runs[r].step * ((int64_t)i - q0);
      return (p >= 0 && (uint64_t)p < contig_len) ? contig[p] : (uint8_t)0;
    }
  }
  return 0;
}

// annotate_simplex_methylation + the normalisation loop of annotate_and_normalize for ONE position of one call.
template <class ReadT>
FGX_METH_HD inline void meth_annotate_position(uint8_t* stage, const ReadT* reads, uint32_t n_reads, const MethRun* runs, uint32_t n_runs, const uint8_t* contig,
                                          uint64_t contig_len, bool top, uint32_t i, uint8_t* is_ref_c, uint32_t* unconverted, uint32_t* converted) {
  const uint8_t rb = meth_upper(meth_ref_base(runs, n_runs, contig, contig_len, i));
  const uint8_t target = top ? (uint8_t)'C' : (uint8_t)'G';      // the reference base of a cytosine on this strand = the unconverted read base
  const uint8_t conv = top ? (uint8_t)'T' : (uint8_t)'A';
  uint32_t u = 0, t = 0;
  const bool is_c = rb == target;
  if (is_c) {
    for (uint32_t r = 0; r < n_reads; r++) {
      if (i >= reads[r].len) continue;
      uint8_t* b = stage + reads[r].off + i;
      const uint8_t v = meth_upper(*b);
      if (v == target) { if (u != 0xFFFFFFFFu) u++; }
      else if (v == conv) { if (t != 0xFFFFFFFFu) t++; *b = target; }
    }
  }
  *is_ref_c = is_c ? 1 : 0;
  *unconverted = u;
  *converted = t;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------

// The aligned runs of an anchor read (query_to_ref_positions): `simplified` = its simplified CIGAR after reversal and truncation,
// `original` = before (ops as (BAM op code, length) with S, =, X, H already folded into M).
template <class Cigar>
inline void meth_runs(const Cigar& simplified, int64_t alignment_start, bool is_reverse, const Cigar& original, std::vector<MethRun>& out) {
  int64_t q = 0, ref_pos, step;
  if (is_reverse) {
    int64_t span = 0;
    for (auto& op : original) if (op.first == 0 || op.first == 2 || op.first == 3 || op.first == 7 || op.first == 8) span += (int64_t)op.second;
    ref_pos = alignment_start + span - 1;
    step = -1;
  } else { ref_pos = alignment_start; step = 1; }
  for (auto& op : simplified) {
    const int64_t len = (int64_t)op.second;
    if (op.first == 0 || op.first == 7 || op.first == 8) { if (len > 0) out.push_back(MethRun{q, len, ref_pos, step}); q += len; ref_pos += step * len; }
    else if (op.first == 1 || op.first == 4) q += len;                   // insertion / soft clip: no reference base
    else if (op.first == 2 || op.first == 3) ref_pos += step * len;      // deletion / skip
  }
}

inline bool meth_is_top_strand(uint16_t flags) { return ((flags & 0x10) != 0) == ((flags & 0x80) != 0); }   // methylation.rs:392-398

// build_mm_ml_tags (methylation.rs:264-329): false = no tag
inline bool meth_build_mm_ml(const uint8_t* bases, uint32_t n, const uint8_t* is_ref_c, const uint32_t* unconverted, const uint32_t* converted, bool top, int mode,
                             std::string& mm, std::vector<uint8_t>& ml) {
  const uint8_t track = top ? 'C' : 'G';
  ml.clear();
  mm = top ? "C+m" : "G-m";
  uint64_t skip = 0;
  bool any = false;
  for (uint32_t i = 0; i < n; i++) {
    if (meth_upper(bases[i]) != track) continue;
    const uint64_t total = is_ref_c[i] ? (uint64_t)unconverted[i] + converted[i] : 0;
    if (total > 0) {
      if (mode != 1 && mode != 2) return false;
      const uint64_t num = mode == 1 ? unconverted[i] : converted[i];     // EM-Seq: unconverted / total; TAPs: converted / total
      const uint64_t p = num * 255 / total;
      ml.push_back((uint8_t)(p > 255 ? 255 : p));
      mm += ",";
      mm += std::to_string(skip);
      skip = 0;
      any = true;
    } else skip++;
  }
  if (!any) return false;
  mm += ";";
  return true;
}

}  // namespace fgx
