// aln_tags_core.h — NM / UQ / MD of a BAM record against the reference, and the record with those tags brought up to date: what `fgumi filter
// --ref` does to every record after the masking (src/lib/commands/filter.rs:115-118, 888-890 -> regenerate_alignment_tags_raw,
// crates/fgumi-sam/src/alignment_tags.rs:259-433, with the raw tag editing rules of crates/fgumi-raw-bam/src/tags.rs:808-888).
//
// Scalar source for host and device: a GPU lane per record runs it (filter.hip: k_aln_plan, k_aln_write), the CPU tests run the same
// functions (fgx_regenerate_alignment_tags_host).  The reference edits the record's byte vector three times in a row (NM, UQ, MD: remove /
// overwrite / splice / append); here the RESULT of those edits is laid out analytically from one scan of the original tag block, because a
// lane has no vector to splice in — `plan` finds where the three tags are and how long the new record is, `write` produces it:
//
//   NM, UQ (update_int_tag :867-888)     found as a 4-byte integer ('i' / 'I'): the value is overwritten where it is; found as anything else: the
//                                         entry goes away and the value is appended with the smallest signed-first type (c C S s i); absent: appended
//   MD     (update_string_tag :832-859)  found with a value as long as the new one: overwritten where it is (type byte untouched); found with another
//                                         length: a new Z entry takes its place; absent: appended
//   appended entries keep the order NM, UQ, MD; an unmapped record (or a mapped one with a negative reference id) LOSES the three tags
//   (remove_tag :808-821); "found" = the FIRST entry with the key, reached through well-formed entries, with a size that can be told
//   (find_tag_bounds :104-109).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define FGX_ALN_HD __host__ __device__
#else
#define FGX_ALN_HD
#endif

namespace fgx {
namespace aln {

enum Status : int {
  ALN_OK = 0,            // tags regenerated
  ALN_REMOVED = 1,       // unmapped / no reference id: tags removed
  ALN_TOO_SHORT = 2,     // "BAM record too short"
  ALN_REF_ID = 3,        // "Reference sequence ID not found in header"
  ALN_BAD_START = 4,     // "Invalid alignment start position"
  ALN_REGION = 5,        // the alignment leaves the contig (reference.rs:291-310)
  ALN_TRUNCATED = 6,     // "Truncated BAM record: seq/qual extends past record end"
  ALN_CIGAR_SEQ = 7      // "CIGAR consumes more bases than sequence length"
};

struct TagLoc { uint32_t start, end; uint8_t type, found; };   // entry [start, end) inside the tag block; found: the entry exists AND its size can be told

struct Plan {
  int status;
  uint32_t aux_off;        // where the tag block starts (= record length when the record has none)
  TagLoc nm, uq, md;
  int32_t nm_val, uq_val;  // regenerated values
  uint32_t md_len;         // length of the regenerated MD text
  uint32_t new_len;        // bytes of the edited record
};

FGX_ALN_HD inline uint32_t rd32u(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
FGX_ALN_HD inline uint32_t fixed_size(uint8_t t) { return (t == 'A' || t == 'c' || t == 'C') ? 1u : (t == 's' || t == 'S') ? 2u : (t == 'i' || t == 'I' || t == 'f') ? 4u : 0u; }
// tag_value_size (fields.rs:309-330): bytes of the value of type `t` at `d` (n bytes left), or -1 when it cannot be told
FGX_ALN_HD inline int64_t value_size(uint8_t t, const uint8_t* d, uint32_t n) {
  const uint32_t fx = fixed_size(t);
  if (fx) return fx;
  if (t == 'Z' || t == 'H') { for (uint32_t i = 0; i < n; i++) if (d[i] == 0) return (int64_t)i + 1; return -1; }
  if (t == 'B') {
    if (n < 5) return -1;
    const uint32_t es = fixed_size(d[0]);
    if (!es) return -1;
    return 5ll + (int64_t)rd32u(d + 1) * (int64_t)es;
  }
  return -1;
}
// bytes of `ab:<smallest signed-first integer type>:v` (append_int_tag, tags.rs:671-691)
FGX_ALN_HD inline uint32_t int_tag_bytes(int32_t v) { return (v >= -128 && v <= 127) ? 4u : (v >= 0 && v <= 255) ? 4u : (v >= 0 && v <= 65535) ? 5u : (v >= -32768 && v <= 32767) ? 5u : 7u; }
FGX_ALN_HD inline uint32_t put_int_tag(uint8_t* o, char a, char b, int32_t v) {
  o[0] = (uint8_t)a; o[1] = (uint8_t)b;
  if (v >= -128 && v <= 127) { o[2] = 'c'; o[3] = (uint8_t)(int8_t)v; return 4; }
  if (v >= 0 && v <= 255) { o[2] = 'C'; o[3] = (uint8_t)v; return 4; }
  if (v >= 0 && v <= 65535) { o[2] = 'S'; o[3] = (uint8_t)v; o[4] = (uint8_t)(v >> 8); return 5; }
  if (v >= -32768 && v <= 32767) { o[2] = 's'; o[3] = (uint8_t)v; o[4] = (uint8_t)((uint32_t)v >> 8); return 5; }
  o[2] = 'i'; const uint32_t u = (uint32_t)v; o[3] = (uint8_t)u; o[4] = (uint8_t)(u >> 8); o[5] = (uint8_t)(u >> 16); o[6] = (uint8_t)(u >> 24); return 7;
}
FGX_ALN_HD inline uint32_t dec_digits(uint32_t v) { uint32_t n = 1; while (v >= 10) { v /= 10; n++; } return n; }
FGX_ALN_HD inline uint32_t put_dec(uint8_t* o, uint32_t v) {
  const uint32_t n = dec_digits(v);
  for (uint32_t i = n; i-- > 0;) { o[i] = (uint8_t)('0' + v % 10); v /= 10; }
  return n;
}
FGX_ALN_HD inline uint8_t lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// The CIGAR walk over the reference (alignment_tags.rs:335-422): NM, UQ and the MD text (written to `md` when it is not null; its length either
// way).  `ref` = the contig's bases from the alignment start on, `span` of them.  Returns ALN_OK or ALN_CIGAR_SEQ.
FGX_ALN_HD inline int walk(const uint8_t* rec, uint32_t cig_off, uint32_t n_ops, uint32_t seq_off, uint32_t qual_off, uint32_t l_seq, const uint8_t* ref, uint32_t span,
                           int32_t* nm_out, uint32_t* uq_out, uint32_t* md_len, uint8_t* md) {
  const char B2A[17] = "=ACMGRSVTWYHKDBN";
  int32_t nm = 0;
  uint32_t uq = 0, ml = 0, ref_o = 0, sp = 0, run = 0;
  auto flush_run = [&]() { if (md) ml += put_dec(md + ml, run); else ml += dec_digits(run); run = 0; };
  for (uint32_t k = 0; k < n_ops; k++) {
    const uint32_t op = rd32u(rec + cig_off + 4 * k), t = op & 15u, len = op >> 4;
    if (t == 0 || t == 7 || t == 8) {
      if ((uint64_t)ref_o + len > span) return ALN_REGION;            // (cannot happen: the span IS the sum of these; kept as the reference keeps it)
      if ((uint64_t)sp + len > l_seq) return ALN_CIGAR_SEQ;
      for (uint32_t i = 0; i < len; i++, sp++) {
        const uint8_t rb = ref[ref_o + i];
        const uint32_t code = (rec[seq_off + (sp >> 1)] >> ((sp & 1u) ? 0 : 4)) & 15u;
        const uint8_t sb = (uint8_t)B2A[code];
        if (sb == 'N' || lower(sb) != lower(rb)) {                    // a masked base counts as a mismatch
          nm++; uq += rec[qual_off + sp];
          flush_run();
          if (md) md[ml] = rb;
          ml++;
        } else run++;
      }
      ref_o += len;
    } else if (t == 1) {
      if ((uint64_t)sp + len > l_seq) return ALN_CIGAR_SEQ;
      nm += (int32_t)len; sp += len;
    } else if (t == 2) {
      if ((uint64_t)ref_o + len > span) return ALN_REGION;
      nm += (int32_t)len;
      flush_run();
      if (md) md[ml] = '^';
      ml++;
      for (uint32_t i = 0; i < len; i++) { if (md) md[ml] = ref[ref_o + i]; ml++; }
      ref_o += len;
    } else if (t == 4) {
      if ((uint64_t)sp + len > l_seq) return ALN_CIGAR_SEQ;
      sp += len;
    } else if (t == 3) ref_o += len;
  }
  flush_run();
  *nm_out = nm; *uq_out = uq; *md_len = ml;
  return ALN_OK;
}

struct Geometry { uint32_t cig_off, n_ops, seq_off, qual_off, l_seq, span; int32_t ref_id, pos; bool zero_span; };

// The first occurrences of NM / UQ / MD in the tag block [aux, aux + n): one scan that stops at the first entry whose size cannot be told
FGX_ALN_HD inline void locate(const uint8_t* aux, uint32_t n, TagLoc& nm, TagLoc& uq, TagLoc& md) {
  nm = TagLoc{0, 0, 0, 0}; uq = nm; md = nm;
  bool s_nm = false, s_uq = false, s_md = false;                     // key seen (the first occurrence decides, also when its size cannot be told)
  uint32_t p = 0;
  while (p + 3 <= n) {
    const uint8_t a = aux[p], b = aux[p + 1], t = aux[p + 2];
    const int64_t sz = value_size(t, aux + p + 3, n - (p + 3));
    const bool ok = sz >= 0 && (uint64_t)p + 3 + (uint64_t)sz <= (uint64_t)n;   // (a B array that claims more elements than the block holds: not an entry to edit)
    const uint32_t end = ok ? p + 3 + (uint32_t)sz : 0;
    if (a == 'N' && b == 'M' && !s_nm) { s_nm = true; if (ok) nm = TagLoc{p, end, t, 1}; }
    else if (a == 'U' && b == 'Q' && !s_uq) { s_uq = true; if (ok) uq = TagLoc{p, end, t, 1}; }
    else if (a == 'M' && b == 'D' && !s_md) { s_md = true; if (ok) md = TagLoc{p, end, t, 1}; }
    if (!ok) break;
    p = end;
  }
}

// `genome` + contig_off[i] .. + contig_len[i]: contig i of the BAM header (one byte per base, as the FASTA holds them); n_ref contigs.
FGX_ALN_HD inline void plan(const uint8_t* rec, uint32_t len, const uint8_t* genome, const uint64_t* contig_off, const uint64_t* contig_len, uint32_t n_ref, Plan& P,
                            Geometry& G) {
  P = Plan{};
  P.aux_off = len; P.new_len = len;
  if (len < 32) { P.status = ALN_TOO_SHORT; return; }     // MIN_BAM_RECORD_LEN (crates/fgumi-raw-bam/src/fields.rs:31; alignment_tags.rs:264)
  const uint32_t l_name = rec[8], n_ops = (uint32_t)rec[12] | ((uint32_t)rec[13] << 8), flag = (uint32_t)rec[14] | ((uint32_t)rec[15] << 8), l_seq = rd32u(rec + 16);
  G.cig_off = 32 + l_name; G.n_ops = n_ops; G.l_seq = l_seq;
  const uint64_t seq_off = 32ull + l_name + 4ull * n_ops, qual_off = seq_off + ((uint64_t)l_seq + 1) / 2, aux_off = qual_off + l_seq;
  P.aux_off = aux_off <= len ? (uint32_t)aux_off : len;
  const uint8_t* const aux = rec + P.aux_off;
  const uint32_t aux_n = len - P.aux_off;
  locate(aux, aux_n, P.nm, P.uq, P.md);
  G.ref_id = (int32_t)rd32u(rec); G.pos = (int32_t)rd32u(rec + 4);
  if ((flag & 4u) || G.ref_id < 0) {                                  // unmapped, or mapped without a reference id: the three tags go (alignment_tags.rs:276-300)
    P.status = ALN_REMOVED;
    uint32_t gone = 0;
    if (P.nm.found) gone += P.nm.end - P.nm.start;
    if (P.uq.found) gone += P.uq.end - P.uq.start;
    if (P.md.found) gone += P.md.end - P.md.start;
    P.new_len = len - gone;
    return;
  }
  if ((uint32_t)G.ref_id >= n_ref) { P.status = ALN_REF_ID; return; }
  if (G.pos < 0) { P.status = ALN_BAD_START; return; }
  // reference span: the sum of the reference-consuming operations; 0 when the CIGAR leaves the buffer or the sum leaves 31 bits (cigar.rs:160-213)
  uint64_t span = 0;
  if ((uint64_t)G.cig_off + 4ull * n_ops <= len) {
    for (uint32_t k = 0; k < n_ops; k++) {
      const uint32_t op = rd32u(rec + G.cig_off + 4 * k), t = op & 15u;
      if (t == 0 || t == 2 || t == 3 || t == 7 || t == 8) { span += op >> 4; if (span > 0x7FFFFFFFull) { span = 0; break; } }
    }
  }
  G.span = (uint32_t)span; G.zero_span = span == 0;
  G.seq_off = (uint32_t)(seq_off <= len ? seq_off : len); G.qual_off = (uint32_t)(qual_off <= len ? qual_off : len);
  if (G.zero_span) { P.nm_val = 0; P.uq_val = 0; P.md_len = 1; }
  else {
    const uint64_t clen = contig_len[G.ref_id];
    if ((uint64_t)G.pos + span > clen) { P.status = ALN_REGION; return; }
    if (seq_off + ((uint64_t)l_seq + 1) / 2 > len || qual_off + l_seq > len) { P.status = ALN_TRUNCATED; return; }
    uint32_t uq = 0;
    const int st = walk(rec, G.cig_off, n_ops, G.seq_off, G.qual_off, l_seq, genome + contig_off[G.ref_id] + (uint64_t)G.pos, G.span, &P.nm_val, &uq, &P.md_len, nullptr);
    if (st != ALN_OK) { P.status = st; return; }
    P.uq_val = (int32_t)(uq > 0x7FFFFFFFu ? 0x7FFFFFFFu : uq);
  }
  // the length of the edited record
  int64_t nl = len;
  auto int_delta = [&](const TagLoc& L, int32_t v) -> int64_t {
    if (L.found && (L.type == 'i' || L.type == 'I')) return 0;                                   // overwritten where it is
    return (int64_t)int_tag_bytes(v) - (L.found ? (int64_t)(L.end - L.start) : 0);              // removed (if there) and appended
  };
  nl += int_delta(P.nm, P.nm_val) + int_delta(P.uq, P.uq_val);
  if (P.md.found) { const uint32_t old_val = (P.md.end - P.md.start) - 4u; if (old_val != P.md_len) nl += (int64_t)(3 + P.md_len + 1) - (int64_t)(P.md.end - P.md.start); }
  else nl += 3 + P.md_len + 1;
  P.new_len = (uint32_t)nl;
  P.status = ALN_OK;
}

// the edited record into out[0 .. P.new_len) (P.status ALN_OK or ALN_REMOVED)
FGX_ALN_HD inline void write(const uint8_t* rec, uint32_t len, const uint8_t* genome, const uint64_t* contig_off, const Plan& P, const Geometry& G, uint8_t* out) {
  uint32_t o = 0;
  auto copy = [&](uint32_t a, uint32_t b) { for (uint32_t i = a; i < b; i++) out[o++] = rec[i]; };
  const bool removing = P.status == ALN_REMOVED;
  // the three entries in the order they lie in the tag block
  const TagLoc* L[3] = {&P.nm, &P.uq, &P.md};
  int kind[3] = {0, 1, 2};
  for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) {
    const uint32_t si = L[i]->found ? L[i]->start : 0xFFFFFFFFu, sj = L[j]->found ? L[j]->start : 0xFFFFFFFFu;
    if (sj < si) { const TagLoc* t = L[i]; L[i] = L[j]; L[j] = t; const int k = kind[i]; kind[i] = kind[j]; kind[j] = k; }
  }
  auto md_text = [&](uint8_t* dst) {
    if (G.zero_span) { dst[0] = '0'; return; }
    int32_t nm; uint32_t uq, ml;
    (void)walk(rec, G.cig_off, G.n_ops, G.seq_off, G.qual_off, G.l_seq, genome + contig_off[G.ref_id] + (uint64_t)G.pos, G.span, &nm, &uq, &ml, dst);
  };
  uint32_t cur = 0;                                   // next byte of the original record to copy
  bool app_nm = !removing && !(P.nm.found && (P.nm.type == 'i' || P.nm.type == 'I'));
  bool app_uq = !removing && !(P.uq.found && (P.uq.type == 'i' || P.uq.type == 'I'));
  bool app_md = !removing && !P.md.found;
  for (int i = 0; i < 3; i++) {
    if (!L[i]->found) continue;
    const uint32_t st = P.aux_off + L[i]->start, en = P.aux_off + L[i]->end;
    copy(cur, st);
    if (removing) { cur = en; continue; }
    if (kind[i] == 2) {                              // MD
      const uint32_t old_val = (L[i]->end - L[i]->start) - 4u;
      if (old_val == P.md_len) { out[o] = rec[st]; out[o + 1] = rec[st + 1]; out[o + 2] = rec[st + 2]; md_text(out + o + 3); o += 3 + P.md_len; copy(st + 3 + P.md_len, en); }
      else { out[o] = 'M'; out[o + 1] = 'D'; out[o + 2] = 'Z'; md_text(out + o + 3); out[o + 3 + P.md_len] = 0; o += 3 + P.md_len + 1; }
    } else {
      const int32_t v = kind[i] == 0 ? P.nm_val : P.uq_val;
      if (L[i]->type == 'i' || L[i]->type == 'I') { out[o] = rec[st]; out[o + 1] = rec[st + 1]; out[o + 2] = rec[st + 2]; const uint32_t u = (uint32_t)v; out[o + 3] = (uint8_t)u; out[o + 4] = (uint8_t)(u >> 8); out[o + 5] = (uint8_t)(u >> 16); out[o + 6] = (uint8_t)(u >> 24); o += 7; }
      // (any other type: the entry goes; the value is appended below)
    }
    cur = en;
  }
  copy(cur, len);
  if (app_nm) o += put_int_tag(out + o, 'N', 'M', P.nm_val);
  if (app_uq) o += put_int_tag(out + o, 'U', 'Q', P.uq_val);
  if (app_md) { out[o] = 'M'; out[o + 1] = 'D'; out[o + 2] = 'Z'; md_text(out + o + 3); out[o + 3 + P.md_len] = 0; o += 3 + P.md_len + 1; }
}

}  // namespace aln
}  // namespace fgx
