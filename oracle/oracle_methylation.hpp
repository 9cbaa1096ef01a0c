// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Restatement of the methylation-aware mode (EM-Seq / TAPs):
//   crates/fgumi-consensus/src/methylation.rs:25-86 (evidence / annotation), 88-107 (is_cpg_context), 116-178
//       (query_to_ref_positions), 193-242 (annotate_simplex_methylation), 264-343 (build_mm_ml_tags / build_mm_tag_no_ml),
//       346-372 (fetch_ref_bases_at_positions), 392-398 (is_top_strand), 404-427 (combine_methylation_annotations)
//   crates/fgumi-consensus/src/lib.rs:45-68 (MethylationMode)
// Pinned by tests/test_oracle_methylation_pins.py against the reference's own unit tests (methylation.rs:461-926,
// vanilla_caller.rs:5856-6250, duplex_caller.rs:6741-7170).
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "oracle_bam.hpp"

namespace orc {

enum MethylationMode : int { MethDisabled = 0, MethEmSeq = 1, MethTaps = 2 };   // lib.rs:50-60

struct MethylationEvidence {   // methylation.rs:25-38
  bool is_ref_c = false;
  uint32_t unconverted = 0, converted = 0;
};

inline uint32_t sat_add_u32(uint32_t a, uint32_t b) { uint64_t s = (uint64_t)a + b; return s > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)s; }
inline int16_t clamp_i16(uint32_t v) { return v > 32767u ? (int16_t)32767 : (int16_t)v; }   // i16::try_from(..).unwrap_or(i16::MAX)

struct MethylationAnnotation {   // methylation.rs:41-86
  std::vector<MethylationEvidence> evidence;
  std::vector<int16_t> unconverted_counts() const { std::vector<int16_t> v; for (auto& e : evidence) v.push_back(clamp_i16(e.unconverted)); return v; }
  std::vector<int16_t> converted_counts() const { std::vector<int16_t> v; for (auto& e : evidence) v.push_back(clamp_i16(e.converted)); return v; }
  MethylationAnnotation truncate(size_t len) const { MethylationAnnotation m; m.evidence.assign(evidence.begin(), evidence.begin() + std::min(len, evidence.size())); return m; }
};

// The reference genome the caller was given: `set_reference(reference, ref_names)` (vanilla_caller.rs:512-522) with every header
// contig present in the FASTA (common.rs:131-140): contig i of the BAM header = seqs[i].
struct Reference {
  std::vector<Bytes> seqs;
};

inline uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

inline bool is_cpg_context(const uint8_t* ref, size_t n, size_t pos, bool top) {   // methylation.rs:88-107
  if (pos >= n) return false;
  if (top) return pos + 1 < n && upper(ref[pos]) == 'C' && upper(ref[pos + 1]) == 'G';
  return pos > 0 && upper(ref[pos]) == 'G' && upper(ref[pos - 1]) == 'C';
}

// query_to_ref_positions :116-178.  Ops are simplified CIGAR ops (BAM op code after S,=,X,H → M; length).  INT64_MIN = None.
constexpr int64_t NO_REF_POS = INT64_MIN;
inline std::vector<int64_t> query_to_ref_positions(const SimpCigar& simplified, int64_t alignment_start, bool is_reverse, const SimpCigar& original) {
  std::vector<int64_t> pos;
  auto is_m = [](uint8_t k) { return k == 0 || k == 7 || k == 8; };
  if (is_reverse) {
    int64_t ref_span = 0;
    for (auto& op : original) if (op.first == 0 || op.first == 2 || op.first == 3 || op.first == 7 || op.first == 8) ref_span += (int64_t)op.second;
    int64_t ref_pos = alignment_start + ref_span - 1;
    for (auto& op : simplified) {
      if (is_m(op.first)) for (size_t k = 0; k < op.second; k++) { pos.push_back(ref_pos); ref_pos -= 1; }
      else if (op.first == 1 || op.first == 4) for (size_t k = 0; k < op.second; k++) pos.push_back(NO_REF_POS);
      else if (op.first == 2 || op.first == 3) ref_pos -= (int64_t)op.second;
    }
  } else {
    int64_t ref_pos = alignment_start;
    for (auto& op : simplified) {
      if (is_m(op.first)) for (size_t k = 0; k < op.second; k++) { pos.push_back(ref_pos); ref_pos += 1; }
      else if (op.first == 1 || op.first == 4) for (size_t k = 0; k < op.second; k++) pos.push_back(NO_REF_POS);
      else if (op.first == 2 || op.first == 3) ref_pos += (int64_t)op.second;
    }
  }
  return pos;
}

// fetch_ref_bases_at_positions :346-372 with the in-memory reference (`sequence_for`): 0 = None
inline Bytes fetch_ref_bases_at_positions(const std::vector<int64_t>& positions, const Bytes& seq) {
  Bytes out;
  for (int64_t p : positions) out.push_back(p != NO_REF_POS && p >= 0 && (uint64_t)p < seq.size() ? seq[(size_t)p] : (uint8_t)0);
  return out;
}

inline bool is_top_strand(uint16_t flg) { return ((flg & flags::REVERSE) != 0) == ((flg & flags::LAST_SEGMENT) != 0); }   // :392-398

// annotate_simplex_methylation :193-242.  `read_bases[r]` = the source reads' bases; ref_bases[i] == 0 = None.
inline MethylationAnnotation annotate_simplex_methylation(size_t len, const std::vector<const Bytes*>& read_bases, const Bytes& ref_bases, bool top) {
  MethylationAnnotation a;
  a.evidence.resize(len);
  const uint8_t ref_target = top ? 'C' : 'G', unconv = top ? 'C' : 'G', conv = top ? 'T' : 'A';
  for (size_t i = 0; i < len; i++) {
    uint8_t rb = i < ref_bases.size() ? ref_bases[i] : 0;
    if (rb == 0 || upper(rb) != ref_target) continue;
    MethylationEvidence& ev = a.evidence[i];
    ev.is_ref_c = true;
    for (const Bytes* b : read_bases) {
      if (i >= b->size()) continue;
      uint8_t base = upper((*b)[i]);
      if (base == unconv) ev.unconverted = sat_add_u32(ev.unconverted, 1);
      else if (base == conv) ev.converted = sat_add_u32(ev.converted, 1);
    }
  }
  return a;
}

// build_mm_ml_tags :264-329.  Returns false for None.
inline bool build_mm_ml_tags(const Bytes& bases, const MethylationAnnotation& annot, bool top, int mode, std::string& mm, Bytes& ml) {
  if (bases.size() != annot.evidence.size()) throw OracleError{"consensus_bases and annotation.evidence must have the same length"};
  const uint8_t track = top ? 'C' : 'G';
  std::vector<size_t> skips;
  ml.clear();
  size_t skip = 0;
  for (size_t i = 0; i < bases.size(); i++) {
    if (upper(bases[i]) != track) continue;
    const MethylationEvidence& ev = annot.evidence[i];
    if (ev.is_ref_c) {
      uint64_t total = (uint64_t)ev.unconverted + ev.converted;
      if (total > 0) {
        if (mode == MethDisabled) return false;
        uint64_t num = mode == MethEmSeq ? ev.unconverted : ev.converted;
        ml.push_back((uint8_t)std::min<uint64_t>(num * 255 / total, 255));
        skips.push_back(skip);
        skip = 0;
      } else skip++;
    } else skip++;
  }
  if (skips.empty()) return false;
  mm = top ? "C+m" : "G-m";
  for (size_t s : skips) { mm += ","; mm += std::to_string(s); }
  mm += ";";
  return true;
}

// combine_methylation_annotations :404-427
inline MethylationAnnotation combine_methylation_annotations(const MethylationAnnotation& ab, const MethylationAnnotation& ba, size_t len) {
  MethylationAnnotation out;
  for (size_t i = 0; i < len; i++) {
    const MethylationEvidence* a = i < ab.evidence.size() ? &ab.evidence[i] : nullptr;
    const MethylationEvidence* b = i < ba.evidence.size() ? &ba.evidence[i] : nullptr;
    MethylationEvidence e;
    e.is_ref_c = (a && a->is_ref_c) || (b && b->is_ref_c);
    e.unconverted = sat_add_u32(a ? a->unconverted : 0, b ? b->unconverted : 0);
    e.converted = sat_add_u32(a ? a->converted : 0, b ? b->converted : 0);
    out.evidence.push_back(e);
  }
  return out;
}

inline void append_u8_array_tag(Bytes& r, const char tag[2], const uint8_t* vals, size_t n) {   // raw-bam tags: B:C
  r.push_back((uint8_t)tag[0]); r.push_back((uint8_t)tag[1]); r.push_back('B'); r.push_back('C');
  uint32_t c = (uint32_t)n;
  for (int i = 0; i < 4; i++) r.push_back((c >> (8 * i)) & 0xFF);
  r.insert(r.end(), vals, vals + n);
}

}  // namespace orc
