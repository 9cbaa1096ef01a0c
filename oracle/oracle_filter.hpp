// oracle_filter.hpp — CPU restatement of `fgumi filter` on unmapped consensus records.  TEST INFRASTRUCTURE ONLY: nothing in
// the product path includes or links this file (see oracle/Makefile header).
//
// Follows, function by function:
//   crates/fgumi-consensus/src/filter.rs   filter_read :523-551, filter_duplex_read :558-637, compute_read_stats :646-671,
//                                           mean_base_quality_full_length :688-705, find_string_or_uint8_array :736-751,
//                                           mask_bases :765-811, mask_duplex_bases :824-923, template_passes :371-395,
//                                           retained_primary_masked_bases :419-442, is_duplex_consensus :493-496
//   crates/fgumi-raw-bam/src/tags.rs       find_tag_position :13-34, find_float_tag :123-138, find_int_tag/extract_int_value
//                                           :143-201, find_array_tag / parse_array_tag_at :478-513, array_tag_element_u16
//                                           :590-610, reverse_array_tag_in_place :892-925, reverse_string_tag_in_place :947-952,
//                                           reverse_complement_string_tag_in_place :960-972
//   crates/fgumi-raw-bam/src/sequence.rs   mask_base :65-72, is_base_n :77-79
//   crates/fgumi-tag/src/tag.rs            PER_BASE_TAGS_TO_REVERSE :283-298, PER_BASE_TAGS_TO_REVCOMP :302
//   src/lib/tag_reversal.rs                reverse_per_base_tags_raw :27-67
//   src/lib/commands/filter.rs             process_record_raw :762-940 (reference == None, methylation filters off),
//                                           check_no_call_and_quality :951-972, check_filters_raw :978-991,
//                                           check_duplex_filters_raw :998-1014, single-read process_fn :581-625,
//                                           template process_fn :653-731
//   src/lib/grouper.rs                     TemplateGrouper::add_records :220-243 (consecutive records with equal QNAME)
//   src/lib/template.rs                    Template::from_records :170-352 (R1, R2, supplementaries, secondaries ordering)
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../include/fgumi_amd.h"
#include "oracle_bam.hpp"
#include "oracle_methylation.hpp"   // struct Reference: the genome by header index
#include "oracle_phred.hpp"

namespace orc_filter {
using namespace orc;

struct Thr { uint64_t min_reads; double max_read_error_rate, max_base_error_rate; };

// tags.rs:143-201
inline bool find_int_tag(Slice aux, const char tag[2], int64_t& v) {
  uint8_t vt = 0;
  long p = find_tag_position(aux.p, aux.n, tag, vt);
  if (p < 0) return false;
  size_t s = (size_t)p + 3;
  auto fits = [&](size_t w) { return s + w <= aux.n; };
  switch (vt) {
    case 'c': if (!fits(1)) return false; v = (int8_t)aux.p[s]; return true;
    case 'C': if (!fits(1)) return false; v = aux.p[s]; return true;
    case 's': if (!fits(2)) return false; v = (int16_t)rd16(aux.p + s); return true;
    case 'S': if (!fits(2)) return false; v = rd16(aux.p + s); return true;
    case 'i': if (!fits(4)) return false; v = rdi32(aux.p + s); return true;
    case 'I': if (!fits(4)) return false; v = rd32(aux.p + s); return true;
    default: return false;
  }
}
// tags.rs:123-138
inline bool find_float_tag(Slice aux, const char tag[2], float& v) {
  uint8_t vt = 0;
  long p = find_tag_position(aux.p, aux.n, tag, vt);
  if (p < 0 || vt != 'f' || (size_t)p + 7 > aux.n) return false;
  uint32_t bits = rd32(aux.p + p + 3);
  memcpy(&v, &bits, 4);
  return true;
}
inline bool has_tag(Slice aux, const char tag[2]) { uint8_t vt; return find_tag_position(aux.p, aux.n, tag, vt) >= 0; }

struct ArrayRef { const uint8_t* data = nullptr; uint8_t elem_type = 0; size_t count = 0, elem_size = 0; bool some = false; };
// tags.rs:478-513
inline ArrayRef parse_array_at(Slice aux, size_t data_start) {
  ArrayRef r;
  if (data_start + 5 > aux.n) return r;
  r.elem_type = aux.p[data_start];
  r.count = rd32(aux.p + data_start + 1);
  r.elem_size = (size_t)tag_fixed_size(r.elem_type);
  if (r.elem_size == 0) return r;
  size_t start = data_start + 5, total = r.count * r.elem_size;
  if (start + total > aux.n) return r;
  r.data = aux.p + start;
  r.some = true;
  return r;
}
inline ArrayRef find_array_tag(Slice aux, const char tag[2]) {
  uint8_t vt = 0;
  long p = find_tag_position(aux.p, aux.n, tag, vt);
  if (p < 0 || vt != 'B') return ArrayRef{};
  return parse_array_at(aux, (size_t)p + 3);
}
// tags.rs:590-610
inline uint16_t elem_u16(const ArrayRef& a, size_t i) {
  if (!a.some || i >= a.count) return 0;
  size_t off = i * a.elem_size;
  switch (a.elem_type) {
    case 'C': return a.data[off];
    case 'S': return rd16(a.data + off);
    case 's': { int16_t v = (int16_t)rd16(a.data + off); return (uint16_t)(v < 0 ? 0 : v); }
    case 'c': { int8_t v = (int8_t)a.data[off]; return (uint16_t)(v < 0 ? 0 : v); }
    default: return 0;
  }
}
// filter.rs:736-751
inline bool string_or_u8_array(Slice aux, const char tag[2], Bytes& out) {
  Slice s = find_string_tag(aux, tag);
  if (s.some) { out.assign(s.p, s.p + s.n); return true; }
  ArrayRef a = find_array_tag(aux, tag);
  if (!a.some || !(a.elem_type == 'C' || a.elem_type == 'c')) return false;
  out.resize(a.count);
  for (size_t i = 0; i < a.count; i++) out[i] = (uint8_t)elem_u16(a, i);
  return true;
}

inline bool is_duplex_consensus(Slice aux) { return has_tag(aux, "aD") && has_tag(aux, "bD"); }   // filter.rs:493-496

enum Result { PASS, INSUFFICIENT_READS, EXCESSIVE_ERROR_RATE };

// filter.rs:523-551
inline Result filter_read(Slice aux, const Thr& t) {
  int64_t depth = 0; float err = 0.f;
  bool hd = find_int_tag(aux, "cD", depth), he = find_float_tag(aux, "cE", err);
  if (!hd || !he)
    throw OracleError{"read does not appear to have consensus calling tags (cD/cE) present; FilterConsensusReads requires reads produced by consensus calling"};
  int64_t min_reads = t.min_reads > (uint64_t)INT64_MAX ? INT64_MAX : (int64_t)t.min_reads;
  if (depth < min_reads) return INSUFFICIENT_READS;
  if ((double)err > t.max_read_error_rate) return EXCESSIVE_ERROR_RATE;
  return PASS;
}

// filter.rs:558-637
inline Result filter_duplex_read(Slice aux, const Thr& cc, const Thr& ab, const Thr& ba) {
  Result r = filter_read(aux, cc);
  if (r != PASS) return r;
  int64_t a_d = 0, b_d = 0; float a_e = 0.f, b_e = 0.f;
  bool ha = find_int_tag(aux, "aD", a_d) || find_int_tag(aux, "aM", a_d);
  bool hb = find_int_tag(aux, "bD", b_d) || find_int_tag(aux, "bM", b_d);
  bool hae = find_float_tag(aux, "aE", a_e), hbe = find_float_tag(aux, "bE", b_e);
  int64_t worst_depth, best_depth;
  if (ha && hb) { if (a_d < b_d) { worst_depth = a_d; best_depth = b_d; } else { worst_depth = b_d; best_depth = a_d; } }
  else if (ha) { worst_depth = 0; best_depth = a_d; }
  else if (hb) { worst_depth = 0; best_depth = b_d; }
  else return PASS;
  float best_error, worst_error;
  if (hae && hbe) { if (a_e < b_e) { best_error = a_e; worst_error = b_e; } else { best_error = b_e; worst_error = a_e; } }
  else if (hae) best_error = worst_error = a_e;
  else if (hbe) best_error = worst_error = b_e;
  else best_error = worst_error = 0.f;
  if ((uint64_t)best_depth < ab.min_reads) return INSUFFICIENT_READS;
  if ((double)best_error > ab.max_read_error_rate) return EXCESSIVE_ERROR_RATE;
  if ((uint64_t)worst_depth < ba.min_reads) return INSUFFICIENT_READS;
  if ((double)worst_error > ba.max_read_error_rate) return EXCESSIVE_ERROR_RATE;
  return PASS;
}

inline bool is_n(const uint8_t* rec, size_t seq_off, size_t i) {
  uint8_t b = rec[seq_off + i / 2];
  return ((i % 2 == 0) ? (b >> 4) : (b & 0xF)) == 0xF;
}
inline void mask_base(uint8_t* rec, size_t seq_off, size_t i) {
  size_t k = seq_off + i / 2;
  rec[k] = (i % 2 == 0) ? (uint8_t)((rec[k] & 0x0F) | 0xF0) : (uint8_t)((rec[k] & 0xF0) | 0x0F);
}

// filter.rs:765-811
inline uint64_t mask_bases(uint8_t* rec, size_t rec_len, const Thr& t, bool has_minq, uint8_t minq) {
  RecView v(rec, rec_len);
  size_t seq_off = v.seq_offset(), qual_off = v.qual_offset(), len = v.l_seq();
  Slice aux = v.aux();
  if (v.aux_offset() > rec_len) aux = Slice{rec, 0, true};
  ArrayRef cd = find_array_tag(aux, "cd"), ce = find_array_tag(aux, "ce");
  bool per_base = cd.some && ce.some;
  uint64_t masked = 0;
  for (size_t i = 0; i < len; i++) {
    uint16_t depth = elem_u16(cd, i), errors = elem_u16(ce, i);
    uint8_t q = rec[qual_off + i];
    bool m = (has_minq && q < minq) || (per_base && (uint64_t)depth < t.min_reads) ||
             (per_base && depth > 0 && ((double)errors / (double)depth) > t.max_base_error_rate);
    if (m) {
      if (!is_n(rec, seq_off, i)) masked++;
      mask_base(rec, seq_off, i);
      rec[qual_off + i] = 2;
    }
  }
  return masked;
}

// filter.rs:824-923
inline uint64_t mask_duplex_bases(uint8_t* rec, size_t rec_len, const Thr& cc, const Thr& ab, const Thr& ba, bool has_minq, uint8_t minq, bool ss_agree) {
  RecView v(rec, rec_len);
  size_t seq_off = v.seq_offset(), qual_off = v.qual_offset(), len = v.l_seq();
  Slice aux = v.aux();
  ArrayRef ad = find_array_tag(aux, "ad"), ae = find_array_tag(aux, "ae"), bd = find_array_tag(aux, "bd"), be = find_array_tag(aux, "be");
  Bytes ac, bc;
  bool hac = ss_agree && string_or_u8_array(aux, "ac", ac), hbc = ss_agree && string_or_u8_array(aux, "bc", bc);
  uint64_t masked = 0;
  for (size_t i = 0; i < len; i++) {
    if (is_n(rec, seq_off, i)) continue;
    uint16_t abd = elem_u16(ad, i), bad = elem_u16(bd, i), abe = elem_u16(ae, i), bae = elem_u16(be, i);
    uint16_t best_depth = abd > bad ? abd : bad, worst_depth = abd < bad ? abd : bad;
    double ab_rate = abd > 0 ? (double)abe / (double)abd : 0.0, ba_rate = bad > 0 ? (double)bae / (double)bad : 0.0;
    double best_rate = ab_rate < ba_rate ? ab_rate : ba_rate, worst_rate = ab_rate > ba_rate ? ab_rate : ba_rate;   // f64::min / max, no NaNs here
    uint32_t total_depth = (uint32_t)abd + bad;
    double total_rate = total_depth > 0 ? (double)((uint32_t)abe + bae) / (double)total_depth : 0.0;
    uint8_t q = rec[qual_off + i];
    bool m = (has_minq && q < minq) || (uint64_t)total_depth < cc.min_reads || total_rate > cc.max_base_error_rate ||
             (uint64_t)best_depth < ab.min_reads || best_rate > ab.max_base_error_rate || (uint64_t)worst_depth < ba.min_reads ||
             worst_rate > ba.max_base_error_rate;
    bool dis = false;
    if (ss_agree && abd > 0 && bad > 0) {
      uint8_t a = (hac && i < ac.size()) ? ac[i] : (uint8_t)'N', b = (hbc && i < bc.size()) ? bc[i] : (uint8_t)'N';
      dis = a != b;
    }
    if (m || dis) {
      masked++;
      mask_base(rec, seq_off, i);
      rec[qual_off + i] = 2;
    }
  }
  return masked;
}

// tag_reversal.rs:27-67
inline void reverse_per_base_tags(uint8_t* rec, size_t rec_len) {
  RecView v(rec, rec_len);
  if (!(v.flags() & flags::REVERSE)) return;
  size_t aux_off = v.aux_offset();
  if (aux_off >= rec_len) return;
  uint8_t* aux = rec + aux_off;
  size_t n = rec_len - aux_off;
  static const char* REV[14] = {"cd", "ce", "ad", "ae", "bd", "be", "aq", "bq", "cu", "ct", "au", "at", "bu", "bt"};
  auto string_range = [&](const char* tag, size_t& s, size_t& e) {
    uint8_t vt = 0;
    long p = find_tag_position(aux, n, tag, vt);
    if (p < 0 || vt != 'Z') return false;
    s = (size_t)p + 3;
    const void* z = memchr(aux + s, 0, n - s);
    if (!z) return false;
    e = (size_t)((const uint8_t*)z - aux);
    return e > s;
  };
  for (const char* tag : REV) {
    uint8_t vt = 0;
    long p = find_tag_position(aux, n, tag, vt);
    if (p < 0) continue;
    if (vt == 'B') {
      ArrayRef a = parse_array_at(Slice{aux, n, true}, (size_t)p + 3);
      if (!a.some || a.count == 0) continue;
      uint8_t* e = aux + p + 3 + 5;
      for (size_t i = 0, j = a.count - 1; i < j; i++, j--)
        for (size_t k = 0; k < a.elem_size; k++) std::swap(e[i * a.elem_size + k], e[j * a.elem_size + k]);
    } else if (vt == 'Z') {
      size_t s, e;
      if (string_range(tag, s, e)) std::reverse(aux + s, aux + e);
    }
  }
  for (const char* tag : {"ac", "bc"}) {
    size_t s, e;
    if (string_range(tag, s, e)) {
      std::reverse(aux + s, aux + e);
      for (size_t i = s; i < e; i++) aux[i] = complement_base(aux[i]);
    }
  }
}

// filter.rs (command) :951-972
inline bool check_no_call_and_quality(const uint8_t* rec, size_t rec_len, double mean_qual, bool has_min_mean, double min_mean, double max_frac) {
  if (has_min_mean && mean_qual < min_mean) return false;
  RecView v(rec, rec_len);
  size_t len = v.l_seq(), seq_off = v.seq_offset(), n = 0;
  for (size_t i = 0; i < len; i++) n += is_n(rec, seq_off, i);
  if (max_frac >= 1.0) return (double)n <= max_frac;
  double frac = len > 0 ? (double)n / (double)len : 0.0;
  return frac <= max_frac;
}

// ---- raw tag editing (crates/fgumi-raw-bam/src/tags.rs) --------------------------------------------------------------------------
// find_tag_bounds :104-109 (first entry with the key; None when that entry's size cannot be told); returns false for None
inline bool find_tag_bounds(const uint8_t* aux, size_t n, const char tag[2], size_t& start, size_t& end) {
  uint8_t vt = 0;
  long p = find_tag_position(aux, n, tag, vt);
  if (p < 0) return false;
  long size = tag_value_size(vt, aux + p + 3, n - ((size_t)p + 3));
  if (size < 0) return false;
  start = (size_t)p; end = (size_t)p + 3 + (size_t)size;
  return true;
}
inline size_t aux_offset_or_len(const Bytes& rec) {   // aux_data_offset_from_record(..).unwrap_or(record.len())
  if (rec.size() < 32) return rec.size();
  RecView v(rec.data(), rec.size());
  size_t off = 32 + (size_t)v.l_read_name() + 4 * (size_t)v.n_cigar_op() + ((size_t)v.l_seq() + 1) / 2 + (size_t)v.l_seq();
  return off <= rec.size() ? off : rec.size();
}
inline void remove_tag(Bytes& rec, const char tag[2]) {   // :808-821
  size_t a = aux_offset_or_len(rec);
  if (a >= rec.size()) return;
  size_t st, en;
  if (find_tag_bounds(rec.data() + a, rec.size() - a, tag, st, en)) rec.erase(rec.begin() + (long)(a + st), rec.begin() + (long)(a + en));
}
inline void update_string_tag(Bytes& rec, const char tag[2], const uint8_t* v, size_t n) {   // :832-859
  size_t a = aux_offset_or_len(rec), st, en;
  if (a < rec.size() && find_tag_bounds(rec.data() + a, rec.size() - a, tag, st, en)) {
    size_t abs_start = a + st, abs_end = a + en;
    size_t old_value_len = en - st - 4;          // tag(2) + type(1) + NUL(1)
    if (old_value_len == n) { if (n) memcpy(rec.data() + abs_start + 3, v, n); }
    else {
      Bytes repl;
      repl.push_back((uint8_t)tag[0]); repl.push_back((uint8_t)tag[1]); repl.push_back('Z');
      repl.insert(repl.end(), v, v + n); repl.push_back(0);
      rec.erase(rec.begin() + (long)abs_start, rec.begin() + (long)abs_end);
      rec.insert(rec.begin() + (long)abs_start, repl.begin(), repl.end());
    }
    return;
  }
  append_string_tag(rec, tag, v, n);
}
inline void update_int_tag(Bytes& rec, const char tag[2], int32_t value) {   // :867-888
  size_t a = aux_offset_or_len(rec), st, en;
  if (a < rec.size() && find_tag_bounds(rec.data() + a, rec.size() - a, tag, st, en)) {
    size_t abs_start = a + st, abs_end = a + en;
    uint8_t vt = rec[abs_start + 2];
    if ((vt == 'i' || vt == 'I') && abs_end - abs_start == 7) { uint32_t u = (uint32_t)value; for (int i = 0; i < 4; i++) rec[abs_start + 3 + i] = (uint8_t)(u >> (8 * i)); return; }
    rec.erase(rec.begin() + (long)abs_start, rec.begin() + (long)abs_end);
    append_int_tag(rec, tag, value);
    return;
  }
  append_int_tag(rec, tag, value);
}

// regenerate_alignment_tags_raw (crates/fgumi-sam/src/alignment_tags.rs:259-433).  `ref` = the reference by header index (contig i of the BAM
// header = seqs[i], as fgx_set_reference / orc_set_reference hand it over; a contig the FASTA lacks is empty).  Returns true when the tags
// were recomputed, false when they were removed.
inline bool regenerate_alignment_tags_raw(Bytes& rec, const Reference& ref) {
  if (rec.size() < 32) throw OracleError{"BAM record too short (minimum 32 bytes)"};   // MIN_BAM_RECORD_LEN = 32 (fgumi-raw-bam/src/fields.rs:31)
  RecView v0(rec.data(), rec.size());
  if (v0.flags() & flags::UNMAPPED) { remove_tag(rec, "NM"); remove_tag(rec, "UQ"); remove_tag(rec, "MD"); return false; }
  int32_t ref_id = v0.ref_id();
  if (ref_id < 0) { remove_tag(rec, "NM"); remove_tag(rec, "UQ"); remove_tag(rec, "MD"); return false; }
  if ((size_t)ref_id >= ref.seqs.size()) throw OracleError{"Reference sequence ID not found in header"};
  int32_t pos0 = v0.pos();
  if (pos0 < 0) throw OracleError{"Invalid alignment start position"};
  size_t ref_span = (size_t)reference_length_from_raw_bam(v0);
  if (ref_span == 0) { update_int_tag(rec, "NM", 0); update_int_tag(rec, "UQ", 0); update_string_tag(rec, "MD", (const uint8_t*)"0", 1); return true; }
  const Bytes& contig = ref.seqs[(size_t)ref_id];
  size_t start_idx = (size_t)pos0, end_idx = (size_t)pos0 + ref_span;          // reference.rs:291-310 (1-based inclusive -> [start, end))
  if (end_idx > contig.size() || start_idx >= end_idx) throw OracleError{"Invalid parameter 'region': reference span outside the contig"};
  const uint8_t* all = contig.data() + start_idx;
  size_t seq_off = v0.seq_offset(), qual_off = v0.qual_offset(), l_seq = v0.l_seq();
  if (seq_off + (l_seq + 1) / 2 > rec.size() || qual_off + l_seq > rec.size()) throw OracleError{"Truncated BAM record: seq/qual extends past record end"};
  int32_t nm = 0; uint32_t uq = 0;
  std::string md;
  size_t ref_offset = 0, seq_pos = 0, match_count = 0;
  size_t n_ops = v0.n_cigar_op(), cig = 32 + (size_t)v0.l_read_name();
  for (size_t k = 0; k < n_ops; k++) {
    uint32_t op = rd32(rec.data() + cig + 4 * k), t = op & 0xF; size_t len = op >> 4;
    if (t == 0 || t == 7 || t == 8) {
      if (ref_offset + len > ref_span) throw OracleError{"CIGAR references beyond fetched reference span"};
      if (seq_pos + len > l_seq) throw OracleError{"CIGAR consumes more bases than sequence length"};
      for (size_t i = 0; i < len; i++) {
        uint8_t rb = all[ref_offset + i];
        uint8_t code = (rec[seq_off + seq_pos / 2] >> ((seq_pos & 1) ? 0 : 4)) & 15, sb = BAM_BASE_TO_ASCII[code], q = rec[qual_off + seq_pos];
        auto lower = [](uint8_t c) { return (uint8_t)((c >= 'A' && c <= 'Z') ? c + 32 : c); };
        if (sb == 'N' || lower(sb) != lower(rb)) { nm += 1; uq += q; md += std::to_string(match_count); match_count = 0; md.push_back((char)rb); }
        else match_count++;
        seq_pos++;
      }
      ref_offset += len;
    } else if (t == 1) {
      if (seq_pos + len > l_seq) throw OracleError{"CIGAR insertion consumes more bases than sequence length"};
      nm += (int32_t)len; seq_pos += len;
    } else if (t == 2) {
      if (ref_offset + len > ref_span) throw OracleError{"CIGAR deletion references beyond fetched reference span"};
      nm += (int32_t)len; md += std::to_string(match_count); match_count = 0; md.push_back('^');
      for (size_t i = 0; i < len; i++) md.push_back((char)all[ref_offset + i]);
      ref_offset += len;
    } else if (t == 4) {
      if (seq_pos + len > l_seq) throw OracleError{"CIGAR soft clip consumes more bases than sequence length"};
      seq_pos += len;
    } else if (t == 3) ref_offset += len;
  }
  md += std::to_string(match_count);
  update_int_tag(rec, "NM", nm);
  update_int_tag(rec, "UQ", (int32_t)std::min<uint32_t>(uq, (uint32_t)INT32_MAX));
  update_string_tag(rec, "MD", (const uint8_t*)md.data(), md.size());
  return true;
}

// ---- methylation (EM-Seq / TAPs) filters (crates/fgumi-consensus/src/filter.rs:925-1340) ------------------------------------------
// MethylationTags::from_record :447-484: the six per-base count arrays copied out of the record (the record is edited afterwards)
struct MethTag { std::vector<uint16_t> v; bool some = false; uint16_t at(size_t i) const { return (some && i < v.size()) ? v[i] : (uint16_t)0; } };
struct MethTags { MethTag cu, ct, au, at, bu, bt; };
inline MethTags methylation_tags_from_record(const uint8_t* rec, size_t rec_len) {
  size_t a = rec_len;
  if (rec_len >= 32) {
    RecView v(rec, rec_len);
    size_t off = 32 + (size_t)v.l_read_name() + 4 * (size_t)v.n_cigar_op() + ((size_t)v.l_seq() + 1) / 2 + (size_t)v.l_seq();
    if (off <= rec_len) a = off;
  }
  Slice aux{rec + a, rec_len - a, true};
  auto get = [&](const char* tag) {
    MethTag t;
    ArrayRef r = find_array_tag(aux, tag);
    if (r.some) { t.some = true; t.v.resize(r.count); for (size_t i = 0; i < r.count; i++) t.v[i] = elem_u16(r, i); }
    return t;
  };
  MethTags m;
  m.cu = get("cu"); m.ct = get("ct"); m.au = get("au"); m.at = get("at"); m.bu = get("bu"); m.bt = get("bt");
  return m;
}
// mask_methylation_depth_simplex_raw_with_tags :973-1004
inline uint64_t mask_methylation_depth_simplex(uint8_t* rec, size_t rec_len, uint64_t min_depth, const MethTags& t) {
  if (rec_len < 32) throw OracleError{"BAM record too short"};
  RecView v(rec, rec_len);
  size_t seq_off = v.seq_offset(), qual_off = v.qual_offset(), len = v.l_seq();
  if (!t.cu.some && !t.ct.some) return 0;
  uint64_t masked = 0;
  for (size_t i = 0; i < len; i++) {
    if (is_n(rec, seq_off, i)) continue;
    uint64_t total = (uint64_t)t.cu.at(i) + t.ct.at(i);
    if (total < min_depth) { masked++; mask_base(rec, seq_off, i); rec[qual_off + i] = 2; }
  }
  return masked;
}
// mask_methylation_depth_duplex_raw_with_tags :1026-1066 (thr = [duplex, AB, BA])
inline uint64_t mask_methylation_depth_duplex(uint8_t* rec, size_t rec_len, const uint32_t thr[3], const MethTags& t) {
  if (rec_len < 32) throw OracleError{"BAM record too short"};
  RecView v(rec, rec_len);
  size_t seq_off = v.seq_offset(), qual_off = v.qual_offset(), len = v.l_seq();
  if (!t.cu.some && !t.ct.some) return 0;
  uint64_t masked = 0;
  for (size_t i = 0; i < len; i++) {
    if (is_n(rec, seq_off, i)) continue;
    uint64_t cc = (uint64_t)t.cu.at(i) + t.ct.at(i), ab = (uint64_t)t.au.at(i) + t.at.at(i), ba = (uint64_t)t.bu.at(i) + t.bt.at(i);
    if (cc < thr[0] || ab < thr[1] || ba < thr[2]) { masked++; mask_base(rec, seq_off, i); rec[qual_off + i] = 2; }
  }
  return masked;
}
// resolve_ref_bases_for_record :1072-1133: per query position the upper-cased reference base, -1 for None; false = the whole map is None.
// (`ref_names.get(tid)` of a tid beyond the header and the u64 arithmetic of a negative pos are not restated: with a reference present
// regenerate_alignment_tags_raw refuses both records right after — callers of the filters never see their output.)
inline bool resolve_ref_bases(const uint8_t* rec, size_t rec_len, const Reference& ref, std::vector<int16_t>& out) {
  RecView v(rec, rec_len);
  out.clear();
  if (v.flags() & flags::UNMAPPED) return false;
  int32_t tid = v.ref_id();
  if (tid < 0 || (size_t)tid >= ref.seqs.size()) return false;
  const Bytes& contig = ref.seqs[(size_t)tid];
  uint64_t ref_pos = (uint64_t)(int64_t)v.pos();
  size_t len = v.l_seq(), n_ops = v.n_cigar_op(), cig = 32 + (size_t)v.l_read_name();
  for (size_t k = 0; k < n_ops && out.size() < len; k++) {
    uint32_t op = rd32(rec + cig + 4 * k), t = op & 0xF; size_t n = op >> 4;
    if (t == 0 || t == 7 || t == 8) {
      for (size_t i = 0; i < n && out.size() < len; i++) out.push_back(ref_pos + i < contig.size() ? (int16_t)orc::upper(contig[(size_t)(ref_pos + i)]) : (int16_t)-1);
      ref_pos += n;
    } else if (t == 1 || t == 4) {
      for (size_t i = 0; i < n && out.size() < len; i++) out.push_back(-1);
    } else if (t == 2 || t == 3) ref_pos += n;
  }
  out.resize(len, -1);
  return true;
}
// mask_strand_methylation_agreement_raw_with_ref_bases_and_tags :1166-1236 (`map` null = None)
inline uint64_t mask_strand_methylation_agreement(uint8_t* rec, size_t rec_len, const std::vector<int16_t>* map, const MethTags& t) {
  if (rec_len < 32) throw OracleError{"BAM record too short"};
  if (!map) return 0;
  RecView v(rec, rec_len);
  size_t seq_off = v.seq_offset(), qual_off = v.qual_offset(), len = v.l_seq();
  if (!t.au.some && !t.bu.some) return 0;
  std::vector<uint8_t> should(len, 0);
  for (size_t i = 0; i + 1 < len; i++) {
    if (!(i + 1 < map->size() && (*map)[i] == 'C' && (*map)[i + 1] == 'G')) continue;
    uint32_t top_u = t.au.at(i), top_c = t.at.at(i), bot_u = t.bu.at(i + 1), bot_c = t.bt.at(i + 1);
    if (top_u + top_c == 0 || bot_u + bot_c == 0) continue;
    if ((top_u > top_c) != (bot_u > bot_c)) should[i] = should[i + 1] = 1;
  }
  uint64_t masked = 0;
  for (size_t i = 0; i < len; i++)
    if (should[i] && !is_n(rec, seq_off, i)) { masked++; mask_base(rec, seq_off, i); rec[qual_off + i] = 2; }
  return masked;
}
// check_conversion_fraction_raw_with_ref_bases_and_tags :1279-1340 (mode: FGX_METHYLATION_*)
inline bool check_conversion_fraction(const uint8_t* rec, size_t rec_len, double min_fraction, const std::vector<int16_t>* map, const MethTags& t, int mode) {
  if (mode == FGX_METHYLATION_DISABLED) return true;
  if (!map) return true;
  size_t len = RecView(rec, rec_len).l_seq();
  if (!t.cu.some && !t.ct.some) return true;
  uint64_t num = 0, evi = 0;
  for (size_t i = 0; i < len; i++) {
    if (!(i < map->size() && (*map)[i] == 'C')) continue;
    if (i + 1 < len && i + 1 < map->size() && (*map)[i + 1] == 'G') continue;
    uint64_t cu = t.cu.at(i), ct = t.ct.at(i), e = cu + ct;
    if (e > 0) { num += mode == FGX_METHYLATION_TAPS ? cu : ct; evi += e; }
  }
  if (evi == 0) return true;
  return (double)num / (double)evi >= min_fraction;
}

// filter.rs (command) :762-940.  `ref` (may be null) = --ref: mapped reads are accepted and NM / UQ / MD are
// regenerated after the masking (the record may change its length: `rec` is a vector of its own).
inline void process_record_raw(Bytes& recv, const fgx_filter_options* o, const Reference* ref, uint64_t& masked, bool& pass) {
  uint8_t* rec = recv.data();
  size_t rec_len = recv.size();
  if (rec_len < 32) throw OracleError{"BAM record too short"};
  RecView v(rec, rec_len);
  if (!ref && !(v.flags() & flags::UNMAPPED)) throw OracleError{"--ref is required when filtering mapped reads to keep NM/UQ/MD tags consistent"};
  if (o->reverse_per_base_tags) reverse_per_base_tags(rec, rec_len);
  double pre_mask_mean = 0.0;
  if (o->has_min_mean_base_quality) {   // filter.rs:688-705
    size_t len = v.l_seq(), qo = v.qual_offset();
    uint64_t sum = 0;
    for (size_t i = 0; i < len; i++) sum += rec[qo + i];
    pre_mask_mean = len == 0 ? 0.0 : (double)sum / (double)len;
  }
  Thr cc{o->min_reads[0], o->max_read_error_rate[0], o->max_base_error_rate[0]}, ab{o->min_reads[1], o->max_read_error_rate[1], o->max_base_error_rate[1]},
      ba{o->min_reads[2], o->max_read_error_rate[2], o->max_base_error_rate[2]};
  bool duplex = is_duplex_consensus(v.aux());
  masked = duplex ? mask_duplex_bases(rec, rec_len, cc, ab, ba, o->has_min_base_quality, o->min_base_quality, o->require_single_strand_agreement)
                  : mask_bases(rec, rec_len, cc, o->has_min_base_quality, o->min_base_quality);
  // the methylation filters :833-886
  const bool strand = o->require_strand_methylation_agreement && duplex, conv = o->has_min_conversion_fraction != 0;
  MethTags mt;
  if (o->has_min_methylation_depth || strand || conv) mt = methylation_tags_from_record(rec, rec_len);
  if (o->has_min_methylation_depth)
    masked += duplex ? mask_methylation_depth_duplex(rec, rec_len, o->min_methylation_depth, mt) : mask_methylation_depth_simplex(rec, rec_len, o->min_methylation_depth[0], mt);
  std::vector<int16_t> map;
  bool has_map = false;
  if ((strand || conv) && ref) has_map = resolve_ref_bases(rec, rec_len, *ref, map);
  if (strand) masked += mask_strand_methylation_agreement(rec, rec_len, has_map ? &map : nullptr, mt);
  if (ref) regenerate_alignment_tags_raw(recv, *ref);       // :888-890 (the vector may have been reallocated / resized)
  RecView v2(recv.data(), recv.size());
  Result r = duplex ? filter_duplex_read(v2.aux(), cc, ab, ba) : filter_read(v2.aux(), cc);
  pass = r == PASS && check_no_call_and_quality(recv.data(), recv.size(), pre_mask_mean, o->has_min_mean_base_quality, o->min_mean_base_quality, o->max_no_call_fraction);
  if (pass && conv && !check_conversion_fraction(recv.data(), recv.size(), o->min_conversion_fraction, has_map ? &map : nullptr, mt, o->methylation_mode)) pass = false;   // :924-937
}
// (the in-place form of the round-3 tests: no reference)
inline void process_record_raw(uint8_t* rec, size_t rec_len, const fgx_filter_options* o, uint64_t& masked, bool& pass) {
  Bytes v(rec, rec + rec_len);
  process_record_raw(v, o, nullptr, masked, pass);
  memcpy(rec, v.data(), rec_len);
}

struct BatchResult {
  Bytes data, rejects;
  uint64_t records_count = 0, passed_count = 0, bases_masked = 0, rejected_count = 0;
};

inline void put_record(Bytes& out, const uint8_t* rec, uint32_t len) {
  uint8_t h[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
  out.insert(out.end(), h, h + 4);
  out.insert(out.end(), rec, rec + len);
}
inline bool is_primary(const uint8_t* rec) { uint16_t f = rd16(rec + 14); return !(f & flags::SECONDARY) && !(f & flags::SUPPLEMENTARY); }

// The whole stream: TemplateGrouper + the Process closure of either mode.  `blob` is mutated (masking is in place).
inline void filter_stream(const fgx_filter_options* o, uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n_rec, BatchResult& res,
                          const Reference* ref = nullptr) {
  if (!o->filter_by_template) {   // filter.rs:581-625
    for (uint32_t r = 0; r < n_rec; r++) {
      Bytes rec(blob + rec_off[r], blob + rec_off[r] + rec_len[r]);
      uint64_t masked = 0; bool pass = false;
      process_record_raw(rec, o, ref, masked, pass);
      res.records_count++;
      if (pass && is_primary(rec.data())) res.bases_masked += masked;
      if (pass) { res.passed_count++; put_record(res.data, rec.data(), (uint32_t)rec.size()); }
      else if (o->track_rejects) { res.rejected_count++; put_record(res.rejects, rec.data(), (uint32_t)rec.size()); }
    }
    return;
  }
  auto name_of = [&](uint32_t r) { RecView v(blob + rec_off[r], rec_len[r]); return v.read_name(); };
  uint32_t r = 0;
  while (r < n_rec) {
    if (rec_len[r] < 32) throw OracleError{"BAM record too short"};
    uint32_t e = r + 1;
    Slice nm = name_of(r);
    while (e < n_rec) {
      if (rec_len[e] < 32) throw OracleError{"BAM record too short"};
      Slice ne = name_of(e);
      if (ne.n != nm.n || memcmp(ne.p, nm.p, nm.n) != 0) break;
      e++;
    }
    // Template::from_records ordering (template.rs:243-352)
    std::vector<uint32_t> order;
    long r1 = -1, r2 = -1;
    std::vector<uint32_t> r1s, r2s, r1x, r2x;
    for (uint32_t i = r; i < e; i++) {
      uint16_t f = rd16(blob + rec_off[i] + 14);
      bool sec = f & flags::SECONDARY, sup = f & flags::SUPPLEMENTARY, is_r1 = !(f & flags::PAIRED) || (f & flags::FIRST_SEGMENT);
      if (is_r1) {
        if (sec) r1x.push_back(i); else if (sup) r1s.push_back(i);
        else if (r1 >= 0) throw OracleError{"Multiple non-secondary, non-supplemental R1 records"};
        else r1 = i;
      } else if (sec) r2x.push_back(i); else if (sup) r2s.push_back(i);
      else if (r2 >= 0) throw OracleError{"Multiple non-secondary, non-supplemental R2 records"};
      else r2 = i;
    }
    if (r1 >= 0) order.push_back((uint32_t)r1);
    if (r2 >= 0) order.push_back((uint32_t)r2);
    for (auto* lst : {&r1s, &r2s, &r1x, &r2x}) for (size_t k = lst->size(); k-- > 0;) order.push_back((*lst)[k]);
    // filter.rs:660-721
    std::vector<uint64_t> masked(order.size());
    std::vector<uint8_t> pass(order.size());
    std::vector<Bytes> recs(order.size());
    bool has_primary = false, all_pass = true;
    for (size_t k = 0; k < order.size(); k++) {
      uint32_t i = order[k];
      bool p = false;
      res.records_count++;
      recs[k].assign(blob + rec_off[i], blob + rec_off[i] + rec_len[i]);
      process_record_raw(recs[k], o, ref, masked[k], p);
      pass[k] = p;
    }
    for (size_t k = 0; k < order.size(); k++)   // template_passes: first failing primary breaks, result identical
      if (is_primary(blob + rec_off[order[k]])) { has_primary = true; if (!pass[k]) all_pass = false; }
    bool tpass = has_primary && all_pass;
    for (size_t k = 0; k < order.size(); k++) {
      const uint8_t* rec = recs[k].data();
      bool prim = is_primary(rec);
      if (tpass && prim) res.bases_masked += masked[k];
      bool keep = prim ? tpass : (tpass && pass[k]);
      if (keep) { res.passed_count++; put_record(res.data, rec, (uint32_t)recs[k].size()); }
      else if (o->track_rejects) { res.rejected_count++; put_record(res.rejects, rec, (uint32_t)recs[k].size()); }
    }
    r = e;
  }
}

}  // namespace orc_filter
