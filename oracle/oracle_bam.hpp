// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Restatement of the raw-BAM byte helpers the consensus path needs:
//   crates/fgumi-raw-bam/src/fields.rs (accessors, tag_value_size :309-330, seq/qual offsets :508-522)
//   crates/fgumi-raw-bam/src/tags.rs:13-48 (find_tag_position / find_string_tag), :650-812 (append_*)
//   crates/fgumi-raw-bam/src/sequence.rs:9-48,183-209 (4-bit codec)
//   crates/fgumi-raw-bam/src/builder.rs:122-301 (UnmappedSamBuilder)
//   crates/fgumi-raw-bam/src/cigar.rs:82-232, 363-384 ; overlap.rs:21-648 ; hash.rs:14-89
//   crates/fgumi-dna/src/dna.rs:24-105 (complement table)
//   crates/fgumi-sam/src/clipper.rs:1183-1241 (simplify_cigar / is_cigar_prefix)
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace orc {

namespace flags {
constexpr uint16_t PAIRED = 0x1, PROPER_PAIR = 0x2, UNMAPPED = 0x4, MATE_UNMAPPED = 0x8, REVERSE = 0x10,
                   MATE_REVERSE = 0x20, FIRST_SEGMENT = 0x40, LAST_SEGMENT = 0x80, SECONDARY = 0x100,
                   QC_FAIL = 0x200, DUPLICATE = 0x400, SUPPLEMENTARY = 0x800;
}

using Bytes = std::vector<uint8_t>;

struct Slice {
  const uint8_t* p = nullptr;
  size_t n = 0;
  bool some = false;
  std::string str() const { return std::string((const char*)p, n); }
};

inline uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rdi32(const uint8_t* p) { return (int32_t)rd32(p); }

static const uint8_t BAM_BASE_TO_ASCII[16] = {'=', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};

inline uint8_t seq_code(uint8_t base) {  // SEQ_CODES sequence.rs:17-33
  static uint8_t table[256];
  static bool init = false;
  if (!init) {
    for (int i = 0; i < 256; i++) table[i] = 0x0F;
    const char* B = "=ACMGRSVTWYHKDBN";
    for (int i = 0; i < 16; i++) {
      table[(uint8_t)B[i]] = (uint8_t)i;
      uint8_t lower = (B[i] >= 'A' && B[i] <= 'Z') ? (uint8_t)(B[i] + 32) : (uint8_t)B[i];
      table[lower] = (uint8_t)i;
    }
    init = true;
  }
  return table[base];
}

inline uint8_t complement_base(uint8_t b) {  // dna.rs:24-83
  switch (b) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; case 'U': return 'A';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'S': return 'S'; case 'W': return 'W'; case 'K': return 'M';
    case 'M': return 'K'; case 'B': return 'V'; case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
    case 'N': return 'N';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; case 'u': return 'a';
    case 'r': return 'y'; case 'y': return 'r'; case 's': return 's'; case 'w': return 'w'; case 'k': return 'm';
    case 'm': return 'k'; case 'b': return 'v'; case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
    case 'n': return 'n';
    default: return b;
  }
}

struct RecView {
  const uint8_t* b;
  size_t len;
  RecView(const uint8_t* p, size_t n) : b(p), len(n) {}
  int32_t ref_id() const { return rdi32(b + 0); }
  int32_t pos() const { return rdi32(b + 4); }
  uint8_t l_read_name() const { return b[8]; }
  uint16_t n_cigar_op() const { return rd16(b + 12); }
  uint16_t flags() const { return rd16(b + 14); }
  uint32_t l_seq() const { return rd32(b + 16); }
  int32_t mate_ref_id() const { return rdi32(b + 20); }
  int32_t mate_pos() const { return rdi32(b + 24); }
  int32_t template_length() const { return rdi32(b + 28); }
  Slice read_name() const {
    size_t l = l_read_name();
    return Slice{b + 32, l > 0 ? l - 1 : 0, true};
  }
  size_t seq_offset() const { return 32 + (size_t)l_read_name() + (size_t)n_cigar_op() * 4; }
  size_t qual_offset() const { return seq_offset() + ((size_t)l_seq() + 1) / 2; }
  size_t aux_offset() const { return qual_offset() + l_seq(); }
  Slice aux() const {
    size_t off = aux_offset();
    if (off <= len) return Slice{b + off, len - off, true};
    return Slice{b, 0, true};
  }
  uint8_t base_code(size_t i) const {
    uint8_t byte = b[seq_offset() + i / 2];
    return (i % 2 == 0) ? (byte >> 4) : (byte & 0xF);
  }
  Bytes sequence_vec() const {
    size_t l = l_seq();
    Bytes out(l);
    for (size_t i = 0; i < l; i++) out[i] = BAM_BASE_TO_ASCII[base_code(i)];
    return out;
  }
  Bytes quality_vec() const {
    size_t l = l_seq(), off = qual_offset();
    return Bytes(b + off, b + off + l);
  }
  std::vector<uint32_t> cigar_ops() const {  // get_cigar_ops cigar.rs:82-103
    size_t n = n_cigar_op();
    std::vector<uint32_t> ops;
    if (n == 0) return ops;
    size_t start = 32 + (size_t)l_read_name();
    if (start + n * 4 > len) return ops;
    for (size_t i = 0; i < n; i++) ops.push_back(rd32(b + start + i * 4));
    return ops;
  }
};

inline void set_base(uint8_t* bam, size_t seq_off, size_t position, uint8_t base) {  // sequence.rs:44-52
  uint8_t enc = seq_code(base);
  size_t bi = seq_off + position / 2;
  if (position % 2 == 0) bam[bi] = (uint8_t)((enc << 4) | (bam[bi] & 0x0F));
  else bam[bi] = (uint8_t)((bam[bi] & 0xF0) | enc);
}

// ---- aux tags ----------------------------------------------------------------------
inline int tag_fixed_size(uint8_t t) {
  switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; }
}
// fields.rs:309-330 ; returns -1 for None
inline long tag_value_size(uint8_t val_type, const uint8_t* data, size_t n) {
  int fixed = tag_fixed_size(val_type);
  if (fixed > 0) return fixed;
  if (val_type == 'Z' || val_type == 'H') {
    const void* z = memchr(data, 0, n);
    if (!z) return -1;
    return (long)((const uint8_t*)z - data) + 1;
  }
  if (val_type == 'B') {
    if (n < 5) return -1;
    int es = tag_fixed_size(data[0]);
    size_t count = rd32(data + 1);
    if (es == 0) return -1;
    return (long)(5 + count * (size_t)es);
  }
  return -1;
}
// tags.rs:13-34 ; returns position or -1
inline long find_tag_position(const uint8_t* aux, size_t n, const char tag[2], uint8_t& val_type) {
  size_t p = 0;
  while (p + 3 <= n) {
    val_type = aux[p + 2];
    if (aux[p] == (uint8_t)tag[0] && aux[p + 1] == (uint8_t)tag[1]) return (long)p;
    long size = tag_value_size(val_type, aux + p + 3, n - (p + 3));
    if (size < 0) break;
    p += 3 + (size_t)size;
  }
  return -1;
}
// tags.rs:39-48
inline Slice find_string_tag(Slice aux, const char tag[2]) {
  uint8_t vt = 0;
  long p = find_tag_position(aux.p, aux.n, tag, vt);
  if (p < 0 || vt != 'Z') return Slice{};
  size_t start = (size_t)p + 3;
  const void* z = memchr(aux.p + start, 0, aux.n - start);
  if (!z) return Slice{};
  return Slice{aux.p + start, (size_t)((const uint8_t*)z - (aux.p + start)), true};
}

// ---- tag encoders (tags.rs:650-760) -----------------------------------------------
inline void append_string_tag(Bytes& r, const char tag[2], const uint8_t* v, size_t n) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('Z');
  r.insert(r.end(), v, v + n); r.push_back(0);
}
inline void append_int_tag(Bytes& r, const char tag[2], int32_t value) {  // signed-first: c, C, S, s, i
  r.push_back(tag[0]); r.push_back(tag[1]);
  if (value >= -128 && value <= 127) { r.push_back('c'); r.push_back((uint8_t)(int8_t)value); }
  else if (value >= 0 && value <= 255) { r.push_back('C'); r.push_back((uint8_t)value); }
  else if (value >= 0 && value <= 65535) { r.push_back('S'); uint16_t v = (uint16_t)value; r.push_back(v & 0xFF); r.push_back(v >> 8); }
  else if (value >= -32768 && value <= 32767) { r.push_back('s'); uint16_t v = (uint16_t)(int16_t)value; r.push_back(v & 0xFF); r.push_back(v >> 8); }
  else { r.push_back('i'); uint32_t v = (uint32_t)value; for (int i = 0; i < 4; i++) r.push_back((v >> (8 * i)) & 0xFF); }
}
inline void append_float_tag(Bytes& r, const char tag[2], float value) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('f');
  uint32_t v; memcpy(&v, &value, 4);
  for (int i = 0; i < 4; i++) r.push_back((v >> (8 * i)) & 0xFF);
}
inline void append_i16_array_tag(Bytes& r, const char tag[2], const int16_t* vals, size_t n) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('B'); r.push_back('s');
  uint32_t c = (uint32_t)n;
  for (int i = 0; i < 4; i++) r.push_back((c >> (8 * i)) & 0xFF);
  for (size_t i = 0; i < n; i++) { uint16_t v = (uint16_t)vals[i]; r.push_back(v & 0xFF); r.push_back(v >> 8); }
}
inline void append_phred33_string_tag(Bytes& r, const char tag[2], const uint8_t* q, size_t n) {
  r.push_back(tag[0]); r.push_back(tag[1]); r.push_back('Z');
  for (size_t i = 0; i < n; i++) { unsigned v = (unsigned)q[i] + 33; r.push_back((uint8_t)(v > 255 ? 255 : v)); }
  r.push_back(0);
}

// pack_sequence_into sequence.rs:183-209
inline void pack_sequence_into(Bytes& dst, const uint8_t* bases, size_t n) {
  for (size_t i = 0; i + 1 < n; i += 2) dst.push_back((uint8_t)((seq_code(bases[i]) << 4) | seq_code(bases[i + 1])));
  if (n % 2 == 1) dst.push_back((uint8_t)(seq_code(bases[n - 1]) << 4));
}

// UnmappedSamBuilder::build_record builder.rs:122-190 ; returns false when the name is too long
inline bool build_unmapped_record(Bytes& buf, const uint8_t* name, size_t name_len, uint16_t flag,
                                  const uint8_t* bases, const uint8_t* quals, size_t n) {
  buf.clear();
  if (name_len >= 255) return false;
  auto p32 = [&](uint32_t v) { for (int i = 0; i < 4; i++) buf.push_back((v >> (8 * i)) & 0xFF); };
  auto p16 = [&](uint16_t v) { buf.push_back(v & 0xFF); buf.push_back(v >> 8); };
  p32((uint32_t)-1); p32((uint32_t)-1);
  buf.push_back((uint8_t)(name_len + 1)); buf.push_back(0);
  p16(4680); p16(0); p16(flag); p32((uint32_t)n);
  p32((uint32_t)-1); p32((uint32_t)-1); p32(0);
  buf.insert(buf.end(), name, name + name_len); buf.push_back(0);
  pack_sequence_into(buf, bases, n);
  buf.insert(buf.end(), quals, quals + n);
  return true;
}
inline void write_with_block_size(const Bytes& rec, Bytes& out) {
  uint32_t bs = (uint32_t)rec.size();
  for (int i = 0; i < 4; i++) out.push_back((bs >> (8 * i)) & 0xFF);
  out.insert(out.end(), rec.begin(), rec.end());
}

// ---- CIGAR helpers ---------------------------------------------------------------------
inline bool consumes_ref(uint32_t t) { return t == 0 || t == 2 || t == 3 || t == 7 || t == 8; }
inline bool consumes_query(uint32_t t) { return t == 0 || t == 1 || t == 4 || t == 7 || t == 8; }

inline int32_t sat_add_i32(int32_t a, int32_t b) {
  int64_t s = (int64_t)a + b;
  if (s > INT32_MAX) return INT32_MAX;
  if (s < INT32_MIN) return INT32_MIN;
  return (int32_t)s;
}
inline int32_t sat_sub_i32(int32_t a, int32_t b) {
  int64_t s = (int64_t)a - b;
  if (s > INT32_MAX) return INT32_MAX;
  if (s < INT32_MIN) return INT32_MIN;
  return (int32_t)s;
}
inline int32_t len_as_i32(uint32_t oplen) { return oplen > (uint32_t)INT32_MAX ? INT32_MAX : (int32_t)oplen; }

// reference_length_from_raw_bam (checked variant → 0 on None) cigar.rs:160-213
inline int32_t reference_length_from_raw_bam(const RecView& v) {
  if (v.len < 32) return 0;
  size_t n = v.n_cigar_op();
  size_t start = 32 + (size_t)v.l_read_name();
  if (start + n * 4 > v.len) return 0;
  int64_t ref_len = 0;
  for (size_t i = 0; i < n; i++) {
    uint32_t op = rd32(v.b + start + i * 4);
    if (consumes_ref(op & 0xF)) {
      ref_len += (int32_t)(op >> 4);
      if (ref_len > INT32_MAX) return 0;  // checked_add overflow → None → 0
    }
  }
  return (int32_t)ref_len;
}
inline size_t query_length_from_cigar(const std::vector<uint32_t>& ops) {
  size_t len = 0;
  for (uint32_t op : ops) if (consumes_query(op & 0xF)) len += (op >> 4);
  return len;
}

// Simplified CIGAR: (kind, len) with kind = BAM op code after S,=,X,H → M
using SimpCigar = std::vector<std::pair<uint8_t, size_t>>;
inline SimpCigar simplify_cigar_from_raw(const std::vector<uint32_t>& ops) {  // noodles_compat.rs:10-55
  SimpCigar out;
  for (uint32_t raw : ops) {
    size_t len = raw >> 4;
    uint32_t t = raw & 0xF;
    if (t > 8) continue;
    uint8_t k = (t == 4 || t == 7 || t == 8 || t == 5) ? 0 : (uint8_t)t;
    if (!out.empty() && out.back().first == k) { out.back().second += len; continue; }
    out.push_back({k, len});
  }
  return out;
}
inline bool is_cigar_prefix(const SimpCigar& a, const SimpCigar& b) {  // clipper.rs:1218-1241
  if (a.size() > b.size()) return false;
  size_t last = a.empty() ? 0 : a.size() - 1;
  for (size_t i = 0; i < a.size(); i++) {
    if (a[i].first != b[i].first) return false;
    if (i == last) { if (a[i].second > b[i].second) return false; }
    else if (a[i].second != b[i].second) return false;
  }
  return true;
}

// ---- Murmur3 name rank hash.rs:14-89 ------------------------------------------------------
inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
inline int32_t fgbio_read_name_rank(const uint8_t* name, size_t len) {
  uint32_t h1 = 42;
  auto mix_k1 = [](uint32_t k1) { k1 *= 0xcc9e2d51u; k1 = rotl32(k1, 15); k1 *= 0x1b873593u; return k1; };
  size_t i = 1;
  while (i < len) {
    uint32_t k1 = (uint32_t)name[i - 1] | ((uint32_t)name[i] << 16);
    h1 ^= mix_k1(k1);
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
    i += 2;
  }
  if (len & 1) h1 ^= mix_k1((uint32_t)name[len - 1]);
  h1 ^= (uint32_t)(2 * len);
  h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
  return (int32_t)h1;
}

// ---- mate-overlap clip (overlap.rs) -------------------------------------------------------
inline int32_t saturating_reference_length(const std::vector<uint32_t>& ops) {  // :283-291
  int32_t r = 0;
  for (uint32_t op : ops) if (consumes_ref(op & 0xF)) r = sat_add_i32(r, len_as_i32(op >> 4));
  return r;
}
inline int32_t alignment_end_1based(int32_t pos1, const std::vector<uint32_t>& ops) {  // :279-281
  return sat_add_i32(sat_sub_i32(pos1, 1), saturating_reference_length(ops));
}
inline size_t trailing_soft_clip(const std::vector<uint32_t>& ops) {
  size_t t = 0;
  for (size_t i = ops.size(); i-- > 0;) {
    uint32_t ty = ops[i] & 0xF;
    if (ty == 4) t += ops[i] >> 4; else if (ty == 5) {} else break;
  }
  return t;
}
inline size_t leading_soft_clip(const std::vector<uint32_t>& ops) {
  size_t t = 0;
  for (uint32_t op : ops) {
    uint32_t ty = op & 0xF;
    if (ty == 4) t += op >> 4; else if (ty == 5) {} else break;
  }
  return t;
}
inline void mate_soft_unclipped(int32_t mate_pos1, const std::vector<uint32_t>& ops, int32_t& ustart, int32_t& uend) {  // :293-307
  size_t ls = leading_soft_clip(ops), ts = trailing_soft_clip(ops);
  int32_t lead = ls > (size_t)INT32_MAX ? INT32_MAX : (int32_t)ls;
  int32_t trail = ts > (size_t)INT32_MAX ? INT32_MAX : (int32_t)ts;
  int32_t ref_len = saturating_reference_length(ops);
  ustart = sat_sub_i32(mate_pos1, lead);
  uend = sat_add_i32(sat_add_i32(sat_sub_i32(mate_pos1, 1), ref_len), trail);
}
// parse_mc_cigar_ops :311-376 ; false → None
inline bool parse_mc_cigar_ops(const uint8_t* s, size_t n, std::vector<uint32_t>& ops) {
  const uint32_t MAX_LEN = (1u << 28) - 1;
  auto code = [](uint8_t c) -> int {
    switch (c) { case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
                 case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return -1; }
  };
  std::vector<std::pair<uint32_t, uint8_t>> tokens;
  uint64_t num = 0; bool have = false;
  for (size_t i = 0; i < n; i++) {
    uint8_t c = s[i];
    if (c >= '0' && c <= '9') {
      num = num * 10 + (c - '0');
      if (num > 0xFFFFFFFFull) num = 0xFFFFFFFFull;  // saturating u32
      if (num > MAX_LEN) return false;
      have = true;
      continue;
    }
    if (!have || num == 0 || code(c) < 0) return false;
    tokens.push_back({(uint32_t)num, c});
    num = 0; have = false;
  }
  if (have || tokens.empty()) return false;
  size_t last = tokens.size() - 1;
  bool saw_ref = false;
  ops.clear();
  for (size_t i = 0; i < tokens.size(); i++) {
    uint8_t op = tokens[i].second;
    switch (op) {
      case 'M': case 'D': case 'N': case '=': case 'X': saw_ref = true; break;
      case 'I': case 'P': break;
      case 'S': {
        bool leading = true, trailing = true;
        for (size_t j = 0; j < i; j++) if (tokens[j].second != 'H') leading = false;
        for (size_t j = i + 1; j < tokens.size(); j++) if (tokens[j].second != 'H') trailing = false;
        if (!leading && !trailing) return false;
        break;
      }
      case 'H': if (i == 0 || i == last) break; return false;
      default: return false;
    }
    ops.push_back((tokens[i].first << 4) | (uint32_t)code(op));
  }
  return saw_ref;
}
// query_bases_up_to_ref_pos :384-431
inline size_t query_bases_up_to_ref_pos(const std::vector<uint32_t>& ops, int32_t start1, int32_t target_pos, bool inclusive) {
  int64_t target = target_pos;
  int64_t incl = inclusive ? 1 : 0;
  int32_t ref_pos = start1;
  size_t q = 0;
  for (uint32_t op : ops) {
    if ((int64_t)ref_pos > target) break;
    uint32_t ty = op & 0xF;
    size_t len = op >> 4;
    if (ty == 0 || ty == 7 || ty == 8) {
      int64_t span = target - (int64_t)ref_pos + incl;
      size_t take = std::min(len, (size_t)std::max<int64_t>(span, 0));
      q += take;
      ref_pos = sat_add_i32(ref_pos, len_as_i32(op >> 4));
      if (take < len) break;
    } else if (ty == 1 || ty == 4) q += len;
    else if (ty == 2 || ty == 3) ref_pos = sat_add_i32(ref_pos, len_as_i32(op >> 4));
  }
  return q;
}
inline size_t sat_sub_sz(size_t a, size_t b) { return a > b ? a - b : 0; }
// bases_extending_past_mate_ops :207-268
inline size_t bases_extending_past_mate_ops(bool is_reverse, int32_t this_pos1, const std::vector<uint32_t>& this_ops,
                                            int32_t mate_pos1, const std::vector<uint32_t>& mate_ops) {
  int32_t read_end = alignment_end_1based(this_pos1, this_ops);
  int32_t mate_end = alignment_end_1based(mate_pos1, mate_ops);
  if (is_reverse) {
    if (this_pos1 > mate_end) {
      int32_t us, ue; mate_soft_unclipped(mate_pos1, mate_ops, us, ue);
      size_t gap = (size_t)(uint32_t)sat_sub_i32(this_pos1, us);
      return sat_sub_sz(leading_soft_clip(this_ops), gap);
    }
    if (read_end < mate_pos1) return 0;
    int32_t first_shared = std::max(this_pos1, mate_pos1);
    size_t rb = query_bases_up_to_ref_pos(this_ops, this_pos1, first_shared, false);
    size_t mb = query_bases_up_to_ref_pos(mate_ops, mate_pos1, first_shared, false);
    return sat_sub_sz(rb, mb);
  }
  if (read_end < mate_pos1) {
    int32_t us, ue; mate_soft_unclipped(mate_pos1, mate_ops, us, ue);
    size_t gap = (size_t)(uint32_t)sat_sub_i32(ue, read_end);
    return sat_sub_sz(trailing_soft_clip(this_ops), gap);
  }
  if (mate_end < this_pos1) return 0;
  int32_t last_shared = std::min(read_end, mate_end);
  size_t rp = sat_sub_sz(query_length_from_cigar(this_ops), query_bases_up_to_ref_pos(this_ops, this_pos1, last_shared, true));
  size_t mp = sat_sub_sz(query_length_from_cigar(mate_ops), query_bases_up_to_ref_pos(mate_ops, mate_pos1, last_shared, true));
  return sat_sub_sz(rp, mp);
}
// is_fr_pair_raw :21-69
inline bool is_fr_pair_raw(const RecView& v) {
  uint16_t f = v.flags();
  if (!(f & flags::PAIRED)) return false;
  if ((f & flags::UNMAPPED) || (f & flags::MATE_UNMAPPED)) return false;
  if (v.ref_id() != v.mate_ref_id()) return false;
  bool rev = f & flags::REVERSE, mrev = f & flags::MATE_REVERSE;
  if (rev == mrev) return false;
  // Rust i32 arithmetic here is plain (+), wrapping in release; values are small in practice.
  int32_t astart = (int32_t)((uint32_t)v.pos() + 1u);
  int32_t mstart = (int32_t)((uint32_t)v.mate_pos() + 1u);
  int32_t isize = v.template_length();
  int32_t pos5, neg5;
  if (rev) {
    int32_t ref_len = reference_length_from_raw_bam(v);
    int32_t end = (int32_t)((uint32_t)astart + (uint32_t)std::max(ref_len - 1, 0));
    pos5 = mstart; neg5 = end;
  } else {
    pos5 = astart; neg5 = (int32_t)((uint32_t)astart + (uint32_t)isize);
  }
  return pos5 < neg5;
}
// is_primary_fr_pair_raw :83-108
inline bool is_primary_fr_pair_raw(const RecView& a, const RecView& b) {
  uint16_t fa = a.flags(), fb = b.flags();
  if ((fa & flags::UNMAPPED) || (fb & flags::UNMAPPED)) return false;
  if ((fa & flags::MATE_UNMAPPED) || (fb & flags::MATE_UNMAPPED)) return false;
  if (a.ref_id() != b.ref_id()) return false;
  bool ar = fa & flags::REVERSE, br = fb & flags::REVERSE;
  if (ar == br) return false;
  return is_fr_pair_raw(ar ? a : b);
}
// is_fr_pair_with_mate_cigar_raw :129-161
inline bool is_fr_pair_with_mate_cigar_raw(const RecView& v, int32_t mate_ref_len) {
  uint16_t f = v.flags();
  if (!(f & flags::PAIRED)) return false;
  if ((f & flags::UNMAPPED) || (f & flags::MATE_UNMAPPED)) return false;
  if (v.ref_id() != v.mate_ref_id()) return false;
  bool rev = f & flags::REVERSE, mrev = f & flags::MATE_REVERSE;
  if (rev == mrev) return false;
  if (rev) return is_fr_pair_raw(v);
  int32_t this_start = (int32_t)((uint32_t)v.pos() + 1u);
  int32_t mate_start = (int32_t)((uint32_t)v.mate_pos() + 1u);
  int32_t mate_end = sat_add_i32(mate_start, std::max(mate_ref_len - 1, 0));
  return this_start < mate_end;
}
// num_bases_extending_past_mate_raw :181-207
inline size_t num_bases_extending_past_mate_raw(const RecView& v) {
  Slice mc = find_string_tag(v.aux(), "MC");
  if (!mc.some) return 0;
  // (UTF-8 validity: any non-ASCII byte fails the CIGAR tokenizer below just the same.)
  for (size_t i = 0; i < mc.n; i++) if (mc.p[i] >= 0x80) { /* from_utf8 may fail or tokenizer rejects */ return 0; }
  std::vector<uint32_t> mate_ops;
  if (!parse_mc_cigar_ops(mc.p, mc.n, mate_ops)) return 0;
  if (!is_fr_pair_with_mate_cigar_raw(v, saturating_reference_length(mate_ops))) return 0;
  bool rev = v.flags() & flags::REVERSE;
  int32_t this_pos = (int32_t)((uint32_t)v.pos() + 1u);
  int32_t mate_pos1 = (int32_t)((uint32_t)v.mate_pos() + 1u);
  return bases_extending_past_mate_ops(rev, this_pos, v.cigar_ops(), mate_pos1, mate_ops);
}
// num_bases_extending_past_mate_vs_mate_raw :223-230
inline size_t num_bases_extending_past_mate_vs_mate_raw(const RecView& rec, const RecView& mate) {
  if (!is_primary_fr_pair_raw(rec, mate)) return 0;
  int32_t mate_pos1 = (int32_t)((uint32_t)mate.pos() + 1u);
  bool rev = rec.flags() & flags::REVERSE;
  int32_t this_pos = (int32_t)((uint32_t)rec.pos() + 1u);
  return bases_extending_past_mate_ops(rev, this_pos, rec.cigar_ops(), mate_pos1, mate.cigar_ops());
}

}  // namespace orc
