// reject_core.h — which input records of an MI group does the simplex caller REJECT (`--rejects`), decided from the records alone.
//
// Every rejection the vanilla caller makes is taken BEFORE the per-position arithmetic (vanilla_caller.rs:1329-1646): secondary /
// supplementary records (:1345-1362), a group or a subgroup below --min-reads (:1364-1378, :1467-1476), reads of zero length after
// quality trimming / masking / the mate clip / the trailing no-call strip (create_source_read :1080-1190), unmapped reads among mapped
// ones and the minority alignments (filter_source_reads_by_alignment :1217-1296), reads the --max-reads downsampling drops (:902-932),
// a subgroup that falls below --min-reads after any of these, and the surviving reads of an orphan R1 / R2 consensus (:1402-1418).
// None of it depends on a consensus base.  So the reject set is a function of the group's records and the options, and the device
// pipeline that produces the consensus records (whose counters already carry the per-reason totals) does not have to be touched to
// produce it: a side kernel evaluates this function, one lane per group (reject_device.hip), only when the caller asks for rejects.
//
// The rejected records are written as the reference writes them: copies of the group's records AFTER the R1 / R2 overlap
// pre-correction (`apply_overlapping_consensus`, overlapping.rs:627-684, runs on the group before the caller sees it — simplex.rs:685-
// 700), each with its 4-byte block_size, in input order (the caller sorts a group's rejects by their index in the group, :1430-1436);
// a group below --min-reads is rejected whole before the pre-correction (simplex.rs:673-683), so its records are the original bytes.
//
// Out of scope (REJ_OUT_OF_SCOPE: the whole batch takes the general path, which also reports the reference's errors): more than
// MAX_READS records, more than MAX_OPS CIGAR ops or MAX_GROUPS alignment groups, malformed records, reads without usable qualities.
//
// Host + device source, scalar, no allocation (the helpers are canon_core.h's); tests/test_reject_core.py checks it against the
// oracle's rejects, byte for byte.
#pragma once
#include "canon_core.h"

namespace fgx {
namespace rej {

enum : int { REJ_OK = 0, REJ_OUT_OF_SCOPE = 1 };

struct Params {
  uint8_t min_bq;                 // min_input_base_quality
  uint8_t overlapping;            // overlapping_consensus option of the command
  uint8_t trim;                   // --trim
  uint8_t has_max_reads;
  uint32_t min_reads;
  uint32_t max_reads;
};

struct Scratch {
  canon::Scratch c;               // per-read info, the alignment filter's list
  int32_t rank[canon::MAX_READS]; // downsampling ranks of one subgroup
  uint32_t sub[canon::MAX_READS]; // one subgroup's record indices, in input order
  uint8_t cls[canon::MAX_READS];  // 0 = not in a subgroup, 1 fragment, 2 R1, 3 R2, 4 secondary / supplementary
  uint64_t woff[canon::MAX_READS];// offsets of the working copies
};

// fgbio_read_name_rank: Murmur3_32 over UTF-16 code units, seed 42 (raw-bam/hash.rs:14-89)
CANON_HD int32_t name_rank(const uint8_t* name, uint32_t len) {
  uint32_t h = 42;
  for (uint32_t i = 1; i < len; i += 2) {
    uint32_t k = (uint32_t)name[i - 1] | ((uint32_t)name[i] << 16);
    k *= 0xcc9e2d51u; k = (k << 15) | (k >> 17); k *= 0x1b873593u;
    h ^= k; h = ((h << 13) | (h >> 19)) * 5 + 0xe6546b64u;
  }
  if (len & 1) { uint32_t k = name[len - 1]; k *= 0xcc9e2d51u; k = (k << 15) | (k >> 17); k *= 0x1b873593u; h ^= k; }
  h ^= 2 * len;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return (int32_t)h;
}

CANON_HD bool eligible(const bam::Rec& v) { return (v.flags() & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) == 0; }

// Basic shape of every record of the group; false = out of scope.
CANON_HD bool records_in_scope(const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n) {
  if (n > canon::MAX_READS) return false;
  for (uint32_t i = 0; i < n; i++) {
    const uint32_t len = rec_len[i];
    if (len < 32 || len > blob_len || rec_off[i] > blob_len - len) return false;      // (a record outside the blob: the general path raises the error)
    bam::Rec v{blob + rec_off[i], len};
    if (v.l_read_name() == 0 || (uint64_t)v.aux_off() > len) return false;
    const uint32_t nc = v.n_cigar();
    if (nc > canon::MAX_OPS) return false;
    for (uint32_t k = 0; k < nc; k++) if ((v.cigar_op(k) & 0xF) > 8) return false;
  }
  return true;
}

// The group's working copies (room for the sum of rec_len bytes in `work`, record i at the running sum of the lengths before it) with
// the R1 / R2 overlap pre-correction applied: pairs the LAST primary R1 and the LAST primary R2 of each name (overlapping.rs:627-684).
CANON_HD void corrected_copies(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* work, uint32_t* ops_scratch) {
  uint64_t w = 0;
  for (uint32_t i = 0; i < n; i++) { const uint8_t* s = blob + rec_off[i]; for (uint32_t k = 0; k < rec_len[i]; k++) work[w + k] = s[k]; w += rec_len[i]; }
  uint64_t st[4] = {0, 0, 0, 0};
  uint64_t wi = 0;
  for (uint32_t i = 0; i < n; wi += rec_len[i], i++) {
    bam::Rec vi{work + wi, rec_len[i]};
    if (!eligible(vi) || !(vi.flags() & (bam::F_FIRST | bam::F_LAST))) continue;
    bool seen = false;
    uint64_t wj = 0;
    for (uint32_t j = 0; j < i && !seen; wj += rec_len[j], j++) {
      bam::Rec vj{work + wj, rec_len[j]};
      seen = eligible(vj) && (vj.flags() & (bam::F_FIRST | bam::F_LAST)) && canon::names_equal(vi, vj);
    }
    if (seen) continue;
    int64_t r1 = -1, r2 = -1;
    uint64_t o1 = 0, o2 = 0;
    wj = wi;
    for (uint32_t j = i; j < n; wj += rec_len[j], j++) {
      bam::Rec vj{work + wj, rec_len[j]};
      if (!eligible(vj) || !canon::names_equal(vi, vj)) continue;
      if (vj.flags() & bam::F_FIRST) { r1 = j; o1 = wj; }
      else if (vj.flags() & bam::F_LAST) { r2 = j; o2 = wj; }
    }
    if (r1 >= 0 && r2 >= 0) canon::overlap_pair(work + o1, rec_len[r1], work + o2, rec_len[r2], ops_scratch, st);
  }
}

// find_quality_trim_point (vanilla_caller.rs:992-1016) over the ORIENTED qualities of a read (stored qualities q[0..n), reversed when rev)
CANON_HD uint32_t trim_point(const uint8_t* q, uint32_t n, bool rev, uint8_t trim_qual) {
  if (trim_qual < 1 || n == 0) return 0;
  int32_t score = 0, max_score = 0;
  uint32_t point = n;
  for (uint32_t i = n; i-- > 0;) {
    score += (int32_t)trim_qual - (int32_t)q[rev ? n - 1 - i : i];
    if (score < 0) break;
    if (score > max_score) { max_score = score; point = i; }
  }
  return point;
}

// create_source_read (vanilla_caller.rs:1080-1190) as far as the filters need it: the read's final length (quality trimming, masking, the
// mate clip, the trailing no-call strip) and its simplified CIGAR (reversed for reverse reads, truncated to the final length) in R.
// 1 = a source read, 0 = dropped (zero length), -1 = out of scope (qualities the reference refuses, an MC tag of too many ops).
CANON_HD int source_info(uint8_t min_bq, bool trim, const bam::Rec& v, canon::ReadInfo& R, canon::Scratch& C) {
  const uint32_t l = v.l_seq(), nc = v.n_cigar();
  R.len = l; R.keep = 1; R.clip = 0; R.final_len = 0; R.n_simp = 0;
  if (l == 0) return 0;                                                          // Ok(None): zero length
  if ((uint64_t)v.qual_off() + l > v.len) return -1;
  const uint8_t* q = v.b + v.qual_off();
  bool all_ff = true;
  for (uint32_t x = 0; x < l; x++) if (q[x] != 0xFF) { all_ff = false; break; }
  if (all_ff) return -1;                                                         // "input read is missing base qualities"
  for (uint32_t x = 0; x < nc; x++) C.ops[x] = v.cigar_op(x);
  const uint32_t an = v.len > v.aux_off() ? v.len - v.aux_off() : 0;
  uint32_t mcl = 0;
  const int64_t mco = bam::find_z_tag(v.b + v.aux_off(), an, 'M', 'C', &mcl);
  bool overflow = false;
  const uint64_t clip = bam::mate_clip(v, C.ops, nc, mco >= 0 ? v.b + v.aux_off() + mco : nullptr, mcl, C.mc_ops, canon::MAX_OPS + 1, &overflow);
  if (overflow) return -1;
  const bool rev = (v.flags() & bam::F_REVERSE) != 0;
  const uint32_t trim_to = trim ? trim_point(q, l, rev, min_bq) : l;
  const uint32_t clip_pos = (uint64_t)l > clip ? l - (uint32_t)clip : 0;
  uint32_t fl = clip_pos < trim_to ? clip_pos : trim_to;
  while (fl > 0) {                                   // oriented position fl-1 = stored position (rev ? l - fl : fl - 1); masked inside [0, trim_to)
    const uint32_t s = rev ? l - fl : fl - 1;
    if (v.base_code(s) == 15 || q[s] < min_bq) fl--; else break;
  }
  R.final_len = fl;
  if (fl == 0) return 0;                                                         // ZeroLengthAfterTrimming
  canon::SimpOp tmp[canon::MAX_OPS];
  uint32_t nt = 0;
  for (uint32_t x = 0; x < nc; x++) {
    const uint32_t t = C.ops[x] & 0xF;
    const uint8_t kk = (t == 4 || t == 5 || t == 7 || t == 8) ? (uint8_t)0 : (uint8_t)t;
    if (nt && tmp[nt - 1].k == kk) tmp[nt - 1].len += C.ops[x] >> 4;
    else { tmp[nt].k = kk; tmp[nt].len = C.ops[x] >> 4; nt++; }
  }
  uint32_t remaining = fl;
  for (uint32_t x = 0; x < nt && remaining > 0; x++) {
    const canon::SimpOp& op = tmp[rev ? nt - 1 - x : x];
    if (op.k == 0 || op.k == 1) { const uint32_t take = op.len < remaining ? op.len : remaining; R.simp[R.n_simp].k = op.k; R.simp[R.n_simp].len = take; R.n_simp++; remaining -= take; }
    else { R.simp[R.n_simp] = op; R.n_simp++; }
  }
  return 1;
}

// process_subgroup (vanilla_caller.rs:1454-1646) over S.sub[0..m): sets mask of the reads it rejects; returns true when the subgroup
// gives a consensus, with the survivors left in S.sub[0..*n_surv).  status: REJ_OUT_OF_SCOPE on anything it cannot decide.
CANON_HD bool subgroup(const Params& P, const uint8_t* base, const uint64_t* off, const uint32_t* rec_len, uint32_t m, uint8_t* mask, Scratch& S, uint32_t* n_surv,
                       int* status) {
  *n_surv = 0;
  if (m == 0) return false;
  if (m < P.min_reads) { for (uint32_t k = 0; k < m; k++) mask[S.sub[k]] = 1; return false; }
  // source reads: final length and simplified CIGAR (create_source_read :1080-1190)
  uint32_t ns = 0;
  for (uint32_t k = 0; k < m; k++) {
    const uint32_t i = S.sub[k];
    const int sr = source_info(P.min_bq, P.trim != 0, bam::Rec{base + off[i], rec_len[i]}, S.c.r[i], S.c);
    if (sr < 0) { *status = REJ_OUT_OF_SCOPE; return false; }
    if (sr == 0) { mask[i] = 1; continue; }                                      // zero length / ZeroLengthAfterTrimming
    S.sub[ns++] = i;                                   // (compaction in place: ns <= k)
  }
  if (ns < P.min_reads) { for (uint32_t k = 0; k < ns; k++) mask[S.sub[k]] = 1; return false; }
  // drop_unmapped_if_any_mapped (:1217-1232)
  bool any_un = false, all_un = true;
  for (uint32_t k = 0; k < ns; k++) { const bool u = (bam::Rec{base + off[S.sub[k]], rec_len[S.sub[k]]}.flags() & bam::F_UNMAPPED) != 0; any_un |= u; all_un &= u; }
  if (any_un && !all_un) {
    uint32_t w = 0;
    for (uint32_t k = 0; k < ns; k++) {
      const uint32_t i = S.sub[k];
      if (bam::Rec{base + off[i], rec_len[i]}.flags() & bam::F_UNMAPPED) mask[i] = 1; else S.sub[w++] = i;
    }
    ns = w;
  }
  // the alignment filter (:1242-1296): S.c.list is its work list (it sorts it); rejected reads lose `keep`
  if (ns >= 2) {
    for (uint32_t k = 0; k < ns; k++) S.c.list[k] = S.sub[k];
    const int rj = canon::alignment_filter(S.c, ns);
    if (rj < 0) { *status = REJ_OUT_OF_SCOPE; return false; }
    uint32_t w = 0;
    for (uint32_t k = 0; k < ns; k++) { const uint32_t i = S.sub[k]; if (S.c.r[i].keep) S.sub[w++] = i; else mask[i] = 1; }
    ns = w;
  }
  if (ns < P.min_reads) { for (uint32_t k = 0; k < ns; k++) mask[S.sub[k]] = 1; return false; }
  // downsample_filtered_source_reads (:902-932; select_lowest_ranking caller.rs:665-674): the max_reads lowest ranks stay, ties by order
  if (P.has_max_reads && ns > P.max_reads) {
    for (uint32_t k = 0; k < ns; k++) { bam::Rec v{base + off[S.sub[k]], rec_len[S.sub[k]]}; S.rank[k] = name_rank(v.name(), v.name_len()); }
    uint32_t w = 0;
    for (uint32_t k = 0; k < ns; k++) {                // position of k in the stable order by rank = ranks below it + equal ranks before it
      uint32_t before = 0;
      for (uint32_t j = 0; j < ns; j++) if (S.rank[j] < S.rank[k] || (S.rank[j] == S.rank[k] && j < k)) before++;
      if (before < P.max_reads) S.c.list[w++] = S.sub[k]; else mask[S.sub[k]] = 1;
    }
    for (uint32_t k = 0; k < w; k++) S.sub[k] = S.c.list[k];
    ns = w;
  }
  if (ns < P.min_reads) { for (uint32_t k = 0; k < ns; k++) mask[S.sub[k]] = 1; return false; }
  *n_surv = ns;
  return true;
}

// The reject mask of one MI group: mask[i] = 1 when record i is written to the rejects.  `work` (sum of rec_len bytes) receives the
// overlap-corrected working copies when the option is on and the group reaches the pre-correction.  *whole = 1 when the group was
// rejected before the pre-correction (its rejects are the ORIGINAL bytes).  Returns REJ_OK or REJ_OUT_OF_SCOPE.
CANON_HD int simplex_reject_mask(const Params& P, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, uint8_t* work,
                                 uint8_t* mask, Scratch& S, uint8_t* whole) {
  *whole = 0;
  if (!records_in_scope(blob, blob_len, rec_off, rec_len, n)) return REJ_OUT_OF_SCOPE;
  for (uint32_t i = 0; i < n; i++) mask[i] = 0;
  if (n == 0) return REJ_OK;
  if (n < P.min_reads) { for (uint32_t i = 0; i < n; i++) mask[i] = 1; *whole = 1; return REJ_OK; }      // simplex.rs:673-683
  // record i of the group as the caller sees it: the corrected working copy, or the input itself
  const uint8_t* base = blob;
  const uint64_t* off = rec_off;
  if (P.overlapping) {
    corrected_copies(blob, rec_off, rec_len, n, work, S.c.ops);
    uint64_t w = 0;
    for (uint32_t i = 0; i < n; i++) { S.woff[i] = w; w += rec_len[i]; }
    base = work; off = S.woff;
  }
  // process_group (:1329-1422)
  uint32_t n_reads = 0;
  for (uint32_t i = 0; i < n; i++) {
    bam::Rec v{base + off[i], rec_len[i]};
    const uint16_t f = v.flags();
    if (!eligible(v)) { S.cls[i] = 4; mask[i] = 1; continue; }
    n_reads++;
    S.cls[i] = !(f & bam::F_PAIRED) ? 1 : (f & bam::F_FIRST) ? 2 : (f & bam::F_LAST) ? 3 : 0;
  }
  if (n_reads == 0) return REJ_OK;
  if (n_reads < P.min_reads) { for (uint32_t i = 0; i < n; i++) if (S.cls[i] != 4) mask[i] = 1; return REJ_OK; }
  int status = REJ_OK;
  bool ok[4] = {false, false, false, false};
  uint32_t surv_n[4] = {0, 0, 0, 0};
  // (the orphan rule below needs R1's survivors after R2 was processed: a survivor is a read of the subgroup whose mask stayed 0)
  for (uint32_t c = 1; c <= 3; c++) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; i++) if (S.cls[i] == c) S.sub[m++] = i;
    ok[c] = subgroup(P, base, off, rec_len, m, mask, S, &surv_n[c], &status);
    if (status != REJ_OK) return status;
  }
  if (ok[2] != ok[3]) {                                  // an orphan R1 / R2 consensus is dropped and its source reads are rejects (:1402-1418)
    const uint8_t c = ok[2] ? 2 : 3;
    for (uint32_t i = 0; i < n; i++) if (S.cls[i] == c && !mask[i]) mask[i] = 1;
  }
  return REJ_OK;
}

// Bytes the group's rejects take in the stream (4-byte block_size + record each) and their number.
CANON_HD uint64_t reject_bytes(const uint32_t* rec_len, uint32_t n, const uint8_t* mask, uint32_t* count) {
  uint64_t b = 0;
  uint32_t c = 0;
  for (uint32_t i = 0; i < n; i++) if (mask[i]) { b += 4ull + rec_len[i]; c++; }
  *count = c;
  return b;
}

// Writes the group's rejected records at `out` (reject_bytes of room).  corrected = the group reached the pre-correction and the option is
// on: the records are recomputed into `work` (corrected_copies) and copied from there; otherwise the input bytes are copied.
CANON_HD void emit_rejects(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, const uint8_t* mask, bool corrected, uint8_t* work,
                           uint32_t* ops_scratch, uint8_t* out) {
  if (corrected) corrected_copies(blob, rec_off, rec_len, n, work, ops_scratch);
  uint64_t w = 0, o = 0;
  for (uint32_t i = 0; i < n; w += rec_len[i], i++) {
    if (!mask[i]) continue;
    const uint8_t* s = corrected ? work + w : blob + rec_off[i];
    canon::wr32(out + o, rec_len[i]);
    for (uint32_t k = 0; k < rec_len[i]; k++) out[o + 4 + k] = s[k];
    o += 4ull + rec_len[i];
  }
}


// =====================================================================================================================================
// The duplex caller's rejects (duplex_caller.rs:1944-2120, 2545-2610; src/lib/commands/duplex.rs:742-830).  What it writes for a
// molecule, in this order:
//   1  the fragment (unpaired) records, in input order — always (consensus_reads :2545-2570);
//   then, for a molecule that gave NO consensus pair (too few reads, a strand-orientation collision, a missing end, the per-base
//   read-count gate after the strand combine):
//   2  every paired /A record, 3  every paired /B record, each in input order;
//   or, for a molecule that gave its pair, the reads the single-strand calls dropped:
//   2 / 3  the reads of zero length after trimming (/A records in input order, then /B: the caller sorts them by their ordinal in
//          AB ++ BA), 4 / 5  the reads the alignment filter dropped (unmapped among mapped ones, minority alignments), same order.
// Whether the molecule gave its pair is the ONE thing that depends on the per-position arithmetic (the depth gate looks at the
// single-strand consensus): the device pipeline has just decided it, so `kept` comes from there (the molecule's output slots are not
// empty), and everything else is a function of the records, as for the simplex caller.  The records are the overlap-corrected copies
// when the command's conditional pre-step ran (duplex.rs:786-795): *corrected.
// code[i] = 0 (not a reject) or the class 1..5 above; the stream is class by class, input order inside a class.
struct DuplexParams {
  uint8_t min_bq, overlapping, trim;
  uint8_t single_strand_ok;       // min-reads YX == 0: the pre-step runs for every molecule
};

CANON_HD int duplex_reject_codes(const DuplexParams& P, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, bool kept,
                                 uint8_t* work, uint8_t* code, Scratch& S, uint8_t* corrected) {
  *corrected = 0;
  if (!records_in_scope(blob, blob_len, rec_off, rec_len, n)) return REJ_OUT_OF_SCOPE;
  for (uint32_t i = 0; i < n; i++) code[i] = 0;
  if (n == 0) return REJ_OK;
  // cls: bit 0 paired, bit 1 R2, bit 2 strand /B
  bool ha = false, hb = false;
  uint32_t n_paired = 0;
  for (uint32_t i = 0; i < n; i++) {
    bam::Rec v{blob + rec_off[i], rec_len[i]};
    const uint16_t f = v.flags();
    const int st = canon::mi_strand(v);
    if (st == 0) ha = true; else if (st == 1) hb = true;
    if (!(f & bam::F_PAIRED)) { S.cls[i] = 0; code[i] = 1; continue; }
    if (st < 0) return REJ_OUT_OF_SCOPE;                                   // a paired read without MI / without /A, /B: the reference raises an error
    if (((f & bam::F_FIRST) != 0) == ((f & bam::F_LAST) != 0)) return REJ_OUT_OF_SCOPE;
    S.cls[i] = (uint8_t)(1 | ((f & bam::F_LAST) ? 2 : 0) | (st ? 4 : 0));
    n_paired++;
  }
  if (n_paired == 0) return REJ_OK;
  if (!kept) { for (uint32_t i = 0; i < n; i++) if (S.cls[i] & 1) code[i] = (S.cls[i] & 4) ? 3 : 2; }
  const bool pre = P.overlapping && (P.single_strand_ok || (n >= 2 && ha && hb));
  *corrected = pre ? 1 : 0;
  if (!kept) return REJ_OK;
  const uint8_t* base = blob;
  const uint64_t* off = rec_off;
  if (pre) {
    corrected_copies(blob, rec_off, rec_len, n, work, S.c.ops);
    uint64_t w = 0;
    for (uint32_t i = 0; i < n; i++) { S.woff[i] = w; w += rec_len[i]; }
    base = work; off = S.woff;
  }
  for (int set = 0; set < 2; set++) {            // set 0: AB-R1 ++ BA-R2 (the R1 consensus), set 1: AB-R2 ++ BA-R1
    uint32_t m = 0;
    for (int pass = 0; pass < 2; pass++)
      for (uint32_t i = 0; i < n; i++) {
        const uint8_t c = S.cls[i];
        if (!(c & 1) || ((c >> 2) & 1) != pass) continue;
        const int r2 = (c >> 1) & 1;
        if ((pass == 0) ? (r2 == set) : (r2 != set)) S.sub[m++] = i;
      }
    uint32_t ns = 0;
    for (uint32_t k = 0; k < m; k++) {
      const uint32_t i = S.sub[k];
      const int sr = source_info(P.min_bq, P.trim != 0, bam::Rec{base + off[i], rec_len[i]}, S.c.r[i], S.c);
      if (sr < 0) return REJ_OUT_OF_SCOPE;
      if (sr == 0) { code[i] = (S.cls[i] & 4) ? 3 : 2; continue; }
      S.sub[ns++] = i;
    }
    bool any_un = false, all_un = true;
    for (uint32_t k = 0; k < ns; k++) { const bool u = (bam::Rec{base + off[S.sub[k]], rec_len[S.sub[k]]}.flags() & bam::F_UNMAPPED) != 0; any_un |= u; all_un &= u; }
    if (any_un && !all_un) {
      uint32_t w = 0;
      for (uint32_t k = 0; k < ns; k++) {
        const uint32_t i = S.sub[k];
        if (bam::Rec{base + off[i], rec_len[i]}.flags() & bam::F_UNMAPPED) code[i] = (S.cls[i] & 4) ? 5 : 4; else S.sub[w++] = i;
      }
      ns = w;
    }
    if (ns >= 2) {
      for (uint32_t k = 0; k < ns; k++) S.c.list[k] = S.sub[k];
      if (canon::alignment_filter(S.c, ns) < 0) return REJ_OUT_OF_SCOPE;
      for (uint32_t k = 0; k < ns; k++) { const uint32_t i = S.sub[k]; if (!S.c.r[i].keep) code[i] = (S.cls[i] & 4) ? 5 : 4; }
    }
  }
  return REJ_OK;
}

// The group's rejects at `out` (reject_bytes of room: `code` is a mask to it), class by class.
CANON_HD void emit_rejects_by_class(const uint8_t* blob, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, const uint8_t* code, bool corrected, uint8_t* work,
                                    uint32_t* ops_scratch, uint8_t* out) {
  if (corrected) corrected_copies(blob, rec_off, rec_len, n, work, ops_scratch);
  uint64_t o = 0;
  for (uint8_t cl = 1; cl <= 5; cl++) {
    uint64_t w = 0;
    for (uint32_t i = 0; i < n; w += rec_len[i], i++) {
      if (code[i] != cl) continue;
      const uint8_t* s = corrected ? work + w : blob + rec_off[i];
      canon::wr32(out + o, rec_len[i]);
      for (uint32_t k = 0; k < rec_len[i]; k++) out[o + 4 + k] = s[k];
      o += 4ull + rec_len[i];
    }
  }
}

// =====================================================================================================================================
// The CODEC caller's rejects (codec_caller.rs:1767-1834 over :625-1262): a mask over the group's records, written in input order, original
// bytes (the CODEC command has no overlap pre-step).  Fragments; the records of a template that is not exactly one primary FR pair; the
// reads each strand's alignment filter drops; and, when the molecule gives no consensus (too few templates, the overlap geometry, the
// strand combine), every read that survived until then.  `kept` again comes from the device pipeline.  Secondary / supplementary
// records are ignored by the caller: never rejects.  Out of scope: a per-strand cap (the reads it drops are counted but NOT rejects,
// while a later whole-molecule rejection lists only the reads in hand), more than MAX_READS records.
CANON_HD int codec_reject_mask(bool has_max_reads, const uint8_t* blob, uint64_t blob_len, const uint64_t* rec_off, const uint32_t* rec_len, uint32_t n, bool kept,
                               uint8_t* mask, canon::CodecScratch& S) {
  if (!records_in_scope(blob, blob_len, rec_off, rec_len, n)) return REJ_OUT_OF_SCOPE;
  for (uint32_t i = 0; i < n; i++) mask[i] = 0;
  if (n == 0) return REJ_OK;
  if (has_max_reads) return REJ_OUT_OF_SCOPE;
  // phase 1 (:625-660): paired primaries only; S.r[i].mate: 0xFFFFFFFF = a paired primary not yet in a template, 0xFFFFFFFE = not one
  uint32_t n_paired = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint16_t f = bam::Rec{blob + rec_off[i], rec_len[i]}.flags();
    S.r[i].rec = i; S.r[i].mate = 0xFFFFFFFEu;
    if (!(f & bam::F_PAIRED)) { mask[i] = 1; continue; }
    if (f & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) continue;
    S.r[i].mate = 0xFFFFFFFFu; n_paired++;
  }
  if (n_paired == 0) return REJ_OK;
  // phase 2: templates in first-appearance order; exactly one primary FR pair each
  uint32_t nt = 0;
  uint32_t opsa[canon::MAX_OPS], opsb[canon::MAX_OPS];
  for (uint32_t i = 0; i < n; i++) {
    if (S.r[i].mate != 0xFFFFFFFFu) continue;
    bam::Rec vi{blob + rec_off[i], rec_len[i]};
    uint32_t members = 1, j_found = 0xFFFFFFFFu;
    for (uint32_t j = i + 1; j < n; j++) {
      if (S.r[j].mate != 0xFFFFFFFFu || !canon::names_equal(vi, bam::Rec{blob + rec_off[j], rec_len[j]})) continue;
      members++;
      if (j_found == 0xFFFFFFFFu) j_found = j;
      S.r[j].mate = 0xFFFFFFFDu;                          // taken by this template
    }
    S.r[i].mate = 0xFFFFFFFDu;
    bool fr = false;
    if (members == 2) {
      bam::Rec a{blob + rec_off[i], rec_len[i]}, b{blob + rec_off[j_found], rec_len[j_found]};
      for (uint32_t k = 0; k < a.n_cigar(); k++) opsa[k] = a.cigar_op(k);
      for (uint32_t k = 0; k < b.n_cigar(); k++) opsb[k] = b.cigar_op(k);
      fr = canon::primary_fr_pair(a, opsa, b, opsb);
    }
    if (!fr) {                                            // NotPrimaryFrPair: every record of the name
      mask[i] = 1;
      for (uint32_t j = i + 1; j < n; j++) if (S.r[j].mate == 0xFFFFFFFDu && canon::names_equal(vi, bam::Rec{blob + rec_off[j], rec_len[j]})) { mask[j] = 1; S.r[j].mate = 0xFFFFFFFEu; }
      S.r[i].mate = 0xFFFFFFFEu;
      continue;
    }
    S.r[i].mate = j_found; S.r[j_found].mate = i;
    if (nt >= canon::MAX_READS / 2) return REJ_OUT_OF_SCOPE;
    const bool i_first = (vi.flags() & bam::F_FIRST) != 0;
    S.r1[nt] = i_first ? i : j_found; S.r2[nt] = i_first ? j_found : i;
    nt++;
  }
  if (nt == 0) return REJ_OK;
  if (kept) {
    // phase 3 (:1130-1174): each strand's alignment filter over the simplified CLIPPED CIGARs (reversed for reverse reads), ordered by the clipped length
    if (nt >= 2) {
      for (uint32_t t = 0; t < nt; t++) {
        const uint32_t i1 = S.r1[t], i2 = S.r2[t];
        bam::Rec a{blob + rec_off[i1], rec_len[i1]}, b{blob + rec_off[i2], rec_len[i2]};
        for (uint32_t k = 0; k < a.n_cigar(); k++) opsa[k] = a.cigar_op(k);
        for (uint32_t k = 0; k < b.n_cigar(); k++) opsb[k] = b.cigar_op(k);
        for (int side = 0; side < 2; side++) {
          const bam::Rec& v = side ? b : a;
          const bam::Rec& mt = side ? a : b;
          const uint32_t* vo = side ? opsb : opsa;
          const uint32_t* mo = side ? opsa : opsb;
          canon::CodecInfo& I = S.r[side ? i2 : i1];
          const bool rev = (v.flags() & bam::F_REVERSE) != 0;
          const uint64_t clip = bam::past_mate_ops(rev, (int32_t)((uint32_t)v.pos() + 1u), vo, v.n_cigar(), (int32_t)((uint32_t)mt.pos() + 1u), mo, mt.n_cigar());
          const uint32_t l = v.l_seq();
          I.keep = (uint64_t)l > clip ? l - (uint32_t)clip : 0;
          I.reverse = rev;
          uint64_t ref_consumed = 0;
          const int no = canon::clip_cigar(vo, v.n_cigar(), clip, rev, I.ops, canon::MAX_OPS + 2, &ref_consumed);
          if (no < 0) return REJ_OUT_OF_SCOPE;
          I.n_ops = (uint32_t)no;
        }
      }
      for (int strand = 0; strand < 2; strand++) {
        for (uint32_t t = 0; t < nt; t++) {
          const canon::CodecInfo& I = S.r[strand ? S.r2[t] : S.r1[t]];
          canon::ReadInfo& R = S.filt.r[t];
          R.final_len = I.keep; R.keep = 1; R.n_simp = 0;
          canon::SimpOp tmp[canon::MAX_OPS + 2];
          uint32_t m = 0;
          for (uint32_t k = 0; k < I.n_ops; k++) {
            const uint32_t ty = I.ops[k] & 0xF;
            const uint8_t kk = (ty == 4 || ty == 5 || ty == 7 || ty == 8) ? (uint8_t)0 : (uint8_t)ty;
            if (m && tmp[m - 1].k == kk) tmp[m - 1].len += I.ops[k] >> 4;
            else { tmp[m].k = kk; tmp[m].len = I.ops[k] >> 4; m++; }
          }
          if (m > canon::MAX_OPS) return REJ_OUT_OF_SCOPE;
          for (uint32_t k = 0; k < m; k++) R.simp[k] = tmp[I.reverse ? m - 1 - k : k];
          R.n_simp = (uint8_t)m;
          S.filt.list[t] = t;
        }
        if (canon::alignment_filter(S.filt, nt) < 0) return REJ_OUT_OF_SCOPE;
        for (uint32_t t = 0; t < nt; t++) if (!S.filt.r[t].keep) mask[strand ? S.r2[t] : S.r1[t]] = 1;
      }
    }
    return REJ_OK;
  }
  // no consensus: the filter's rejects and the reads it left — every read of an FR template
  for (uint32_t t = 0; t < nt; t++) { mask[S.r1[t]] = 1; mask[S.r2[t]] = 1; }
  return REJ_OK;
}

}  // namespace rej
}  // namespace fgx
