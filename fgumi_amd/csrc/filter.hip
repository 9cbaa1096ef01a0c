// filter.hip — `fgumi filter` on a stream of unmapped consensus records, on the device.
//
// Mirrors   src/lib/commands/filter.rs:762-940      Filter::process_record_raw (no --ref, methylation filters off): tag reversal,
//                                                    pre-mask mean quality, per-base masking, read-level thresholds, no-call check
//           src/lib/commands/filter.rs:581-625      single-read Process closure; :653-731 template Process closure
//           crates/fgumi-consensus/src/filter.rs    filter_read :523-551, filter_duplex_read :558-637, mask_bases :765-811,
//                                                    mask_duplex_bases :824-923, mean_base_quality_full_length :688-705,
//                                                    template_passes :371-395, retained_primary_masked_bases :419-442
//           crates/fgumi-raw-bam/src/tags.rs        find_tag_position :13-34 (first occurrence wins; the walk stops at a malformed
//                                                    entry), extract_int_value :178-201, parse_array_tag_at :492-513,
//                                                    array_tag_element_u16 :590-610, reverse_*_tag_in_place :892-972
//           src/lib/tag_reversal.rs:27-67           reverse_per_base_tags_raw
//           src/lib/grouper.rs:220-243              TemplateGrouper (consecutive records with an equal QNAME)
//           src/lib/template.rs:243-352             Template::from_records ordering (R1, R2, supplementaries, secondaries)
//
// Kernels: k_filter_records (one wavefront per record: the record is staged into LDS, its aux block is walked once by the whole
// wave — each lane owns one tag of interest — then the lanes sweep the positions two per lane, one packed sequence byte each, and
// write the masked bytes back in place); k_template_flags + scan (template boundaries); k_template_decide (one thread per
// template: keep / reject per record and its place in the output order); two 64-bit scans; k_copy_records (one wavefront per
// record: unaligned copy with aligned dword loads and stores).
#include <hipcub/hipcub.hpp>
#include "bamrec.h"
#include "engine.h"
#include "aln_tags_core.h"

namespace fgx {

namespace {

constexpr int N_TAGS = 24;
enum { T_cD, T_cE, T_cd, T_ce, T_aD, T_bD, T_aM, T_bM, T_aE, T_bE, T_ad, T_ae, T_bd, T_be, T_ac, T_bc, T_aq, T_bq, T_cu, T_ct, T_au, T_at, T_bu, T_bt };
#define TG(a, b) (uint16_t)((uint8_t)(a) | ((uint16_t)(uint8_t)(b) << 8))
__device__ const uint16_t FILTER_TAGS[N_TAGS] = {TG('c', 'D'), TG('c', 'E'), TG('c', 'd'), TG('c', 'e'), TG('a', 'D'), TG('b', 'D'), TG('a', 'M'), TG('b', 'M'),
                                                 TG('a', 'E'), TG('b', 'E'), TG('a', 'd'), TG('a', 'e'), TG('b', 'd'), TG('b', 'e'), TG('a', 'c'), TG('b', 'c'),
                                                 TG('a', 'q'), TG('b', 'q'), TG('c', 'u'), TG('c', 't'), TG('a', 'u'), TG('a', 't'), TG('b', 'u'), TG('b', 't')};

enum : uint32_t { ERR_SHORT = 1, ERR_MAPPED = 2, ERR_NO_TAGS = 3, ERR_MULTI_R1 = 4, ERR_MULTI_R2 = 5, ERR_ALN = 16 /* + aln::Status (2 .. 7) */ };
// per-position value sources: K_NONE reads 0 everywhere
enum : uint32_t { K_NONE = 0, K_U8 = 1, K_U16 = 2, K_I16 = 3, K_I8 = 4, K_ZERO = 5, K_BYTES = 6 };

struct FilterParams {
  uint8_t* blob; uint64_t blob_len; const uint64_t* rec_off; const uint32_t* rec_len; uint32_t n_rec;
  fgx_filter_options o;
  uint8_t* pass; uint32_t* masked; unsigned long long* error;
  uint32_t lds_slice;
  // the reference-dependent methylation filters (k_filter_records<1>): the genome of fgx_set_reference, contig i of the header = [off, off + len)
  const uint8_t* genome; const uint64_t* contig_off; const uint64_t* contig_len; uint32_t n_ref;
};

__device__ inline uint32_t rl(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ inline uint64_t wave_sum(uint64_t v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ inline void report(unsigned long long* err, uint32_t rec, uint32_t code) { atomicMin(err, ((unsigned long long)rec << 8) | code); }

__device__ inline uint8_t comp_ascii(uint8_t b) {      // fgumi_dna::COMPLEMENT (crates/fgumi-dna/src/dna.rs:24-83)
  switch (b) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C'; case 'U': return 'A';
    case 'R': return 'Y'; case 'Y': return 'R'; case 'K': return 'M'; case 'M': return 'K'; case 'B': return 'V';
    case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; case 'u': return 'a';
    case 'r': return 'y'; case 'y': return 'r'; case 'k': return 'm'; case 'm': return 'k'; case 'b': return 'v';
    case 'v': return 'b'; case 'd': return 'h'; case 'h': return 'd';
    default: return b;
  }
}

// array_tag_element_u16 / string byte at position i; `kind` and the location are wave-uniform
__device__ inline uint32_t elem(const uint8_t* A, uint32_t kind, uint32_t off, uint32_t count, uint32_t i) {
  if (i >= count) return 0;
  switch (kind) {
    case K_U8: case K_BYTES: return A[off + i];
    case K_U16: return (uint32_t)A[off + 2 * i] | ((uint32_t)A[off + 2 * i + 1] << 8);
    case K_I16: { const int16_t v = (int16_t)((uint16_t)A[off + 2 * i] | ((uint16_t)A[off + 2 * i + 1] << 8)); return v < 0 ? 0u : (uint32_t)v; }
    case K_I8: { const int8_t v = (int8_t)A[off + i]; return v < 0 ? 0u : (uint32_t)v; }
    default: return 0;
  }
}

struct Src { uint32_t kind, off, count; };

// TAG_FIXED_SIZES without a branch tree: bit (type - 64) of three masks (A c C → 1, s S → 2, i I f → 4)
__device__ inline int fixed_size_fast(uint32_t ty) {
#define BIT(ch) (1ull << ((ch) - 64))
  constexpr uint64_t M1 = BIT('A') | BIT('C') | BIT('c'), M2 = BIT('S') | BIT('s'), M4 = BIT('I') | BIT('i') | BIT('f');
#undef BIT
  const uint32_t i = (ty - 64) & 63;
  const int v = (int)(((M1 >> i) & 1) | (((M2 >> i) & 1) << 1) | (((M4 >> i) & 1) << 2));
  return (ty - 64) < 64 ? v : 0;
}

#ifndef FGX_PHASE_TIMING
#define FGX_PHASE_TIMING 0   /* 1: per-phase s_memtime deltas of k_filter_records into g_fphase (profiling builds only) */
#endif
#if FGX_PHASE_TIMING
__device__ unsigned long long g_fphase[64 * 8];
#define FPH(i) { unsigned long long _n = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&g_fphase[(blockIdx.x & 63) * 8 + (i)], _n - _t); _t = _n; }
#define FPH_ARG , unsigned long long _t
#define FPH_PASS , _t
#else
#define FPH(i)
#define FPH_ARG
#define FPH_PASS
#endif

// Everything after staging; R points at the record in LDS (or in HBM for a record larger than the slice) — called once per
// address space so that the LDS path compiles to ds_read / ds_write instead of flat accesses.
template <int METH>
__device__ __forceinline__ void filter_body(const FilterParams& P, const uint32_t r, const uint32_t lane, uint8_t* g, uint8_t* R, const bool staged,
                                            const uint32_t len FPH_ARG) {
  const uint32_t l_name = R[8], n_cig = bam::rd16(R + 12), flags = bam::rd16(R + 14), l_seq = bam::rd32(R + 16);
  const uint64_t seq_off64 = 32ull + l_name + 4ull * n_cig, aux_off64 = seq_off64 + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (aux_off64 > len) { if (lane == 0) { report(P.error, r, ERR_SHORT); P.pass[r] = 0; P.masked[r] = 0; } return; }
  if (!(flags & bam::F_UNMAPPED) && !P.o.regenerate_alignment_tags) { if (lane == 0) { report(P.error, r, ERR_MAPPED); P.pass[r] = 0; P.masked[r] = 0; } return; }   // (filter.rs:782-792: only without --ref)
  const uint32_t seq_off = (uint32_t)seq_off64, qual_off = seq_off + (l_seq + 1) / 2, aux_off = (uint32_t)aux_off64, an = len - aux_off;
  const uint8_t* A = R + aux_off;

  FPH(0)
  // ---- one walk over the aux block; lane k (< N_TAGS) remembers the first entry of its tag ----
  const uint16_t my_tag = FILTER_TAGS[lane < N_TAGS ? lane : 0];
  int32_t my_pos = -1;
  uint32_t my_ty = 0;
  int64_t my_size = -1;
  {
    uint32_t p = 0;
    while (p + 3 <= an) {
      // a 64-byte window of the aux block, one byte per lane; the entries that start inside it are taken apart with scalar
      // lane reads, so a run of short tags costs one LDS round trip instead of one (or three) per tag
      const uint32_t wbase = p;
      const uint32_t wi = wbase + lane;
      const uint32_t w = wi < an ? (uint32_t)A[wi] : 0x100u;              // 0x100: past the end, never NUL
      const uint64_t nul = __ballot(w == 0);
      bool stop = false;
      for (;;) {
        const uint32_t o = p - wbase;                                      // p and o are wave-uniform
        const uint32_t t = (rl(w, o) & 0xFF) | ((rl(w, o + 1) & 0xFF) << 8);
        const uint32_t ty = rl(w, o + 2) & 0xFF;
        const bool m = lane < N_TAGS && t == my_tag && my_pos < 0;
        if (m) { my_pos = (int32_t)p; my_ty = ty; }
        int64_t size = -1;
        const int fx = fixed_size_fast(ty);
        if (fx > 0) size = fx;
        else if (ty == 'Z' || ty == 'H') {
          const uint64_t in_window = nul >> (o + 3);
          if (in_window) size = (int64_t)(__ffsll((unsigned long long)in_window) - 1) + 1;
          else {
            const uint32_t s0 = p + 3;
            for (uint32_t base = wbase + 64; base < an; base += 64) {
              const uint32_t i = base + lane;
              const uint8_t ch = i < an ? A[i] : (uint8_t)1;
              const uint64_t bm = __ballot(ch == 0);
              if (bm) { size = (int64_t)(base - s0) + (__ffsll((unsigned long long)bm) - 1) + 1; break; }
            }
          }
        } else if (ty == 'B') {
          if (an - (p + 3) >= 5) {
            const int es = fixed_size_fast(rl(w, o + 3) & 0xFF);
            const uint32_t cnt = (rl(w, o + 4) & 0xFF) | ((rl(w, o + 5) & 0xFF) << 8) | ((rl(w, o + 6) & 0xFF) << 16) | ((rl(w, o + 7) & 0xFF) << 24);
            if (es > 0) size = 5 + (int64_t)cnt * es;
          }
        }
        if (m) my_size = size;
        if (size < 0) { stop = true; break; }
        const uint64_t np = (uint64_t)p + 3 + (uint64_t)size;
        if (np + 3 > an) { stop = true; break; }
        p = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)np);
        if (p - wbase + 8 > 64) break;                                     // next entry header not wholly inside: new window
      }
      if (stop) break;
    }
  }
  FPH(1)
  // lane-local decode of the remembered entry
  uint32_t has_int = 0, iv_lo = 0, iv_hi = 0, has_f = 0, fbits = 0;
  uint32_t arr_ok = 0, arr_et = 0, arr_es = 0, arr_cnt = 0;
  if (my_pos >= 0) {
    const uint32_t s = (uint32_t)my_pos + 3;
    int64_t v = 0;
    switch (my_ty) {
      case 'c': if (s + 1 <= an) { has_int = 1; v = (int8_t)A[s]; } break;
      case 'C': if (s + 1 <= an) { has_int = 1; v = A[s]; } break;
      case 's': if (s + 2 <= an) { has_int = 1; v = (int16_t)bam::rd16(A + s); } break;
      case 'S': if (s + 2 <= an) { has_int = 1; v = bam::rd16(A + s); } break;
      case 'i': if (s + 4 <= an) { has_int = 1; v = (int32_t)bam::rd32(A + s); } break;
      case 'I': if (s + 4 <= an) { has_int = 1; v = bam::rd32(A + s); } break;
      case 'f': if ((uint32_t)my_pos + 7 <= an) { has_f = 1; fbits = bam::rd32(A + s); } break;
      case 'B':
        if ((uint64_t)s + 5 <= an) {
          arr_et = A[s]; arr_cnt = bam::rd32(A + s + 1); arr_es = (uint32_t)bam::tag_fixed_size((uint8_t)arr_et);
          if (arr_es && (uint64_t)s + 5 + (uint64_t)arr_cnt * arr_es <= an) arr_ok = 1;
        }
        break;
      default: break;
    }
    iv_lo = (uint32_t)(uint64_t)v; iv_hi = (uint32_t)((uint64_t)v >> 32);
  }
  const uint32_t z_ok = (my_pos >= 0 && my_ty == 'Z' && my_size >= 1) ? 1u : 0u;
  const uint32_t z_len = z_ok ? (uint32_t)(my_size - 1) : 0u;
  // per-position source of this lane's tag (B arrays; ac / bc also accept a Z string and only C / c arrays: filter.rs:736-751)
  uint32_t src_kind = K_NONE, src_off = 0, src_cnt = 0;
  {
    const bool bases_tag = lane == T_ac || lane == T_bc;
    if (bases_tag && z_ok) { src_kind = K_BYTES; src_off = (uint32_t)my_pos + 3; src_cnt = z_len; }
    else if (arr_ok && !(bases_tag && my_pos >= 0 && my_ty == 'Z')) {
      const uint32_t k = arr_et == 'C' ? K_U8 : arr_et == 'S' ? K_U16 : arr_et == 's' ? K_I16 : arr_et == 'c' ? K_I8 : K_ZERO;
      if (!bases_tag || k == K_U8 || k == K_I8) { src_kind = k; src_off = (uint32_t)my_pos + 8; src_cnt = arr_cnt; }
    }
  }
  const uint32_t my_present = my_pos >= 0 ? 1u : 0u;

  // ---- reverse_per_base_tags_raw: in the LDS copy (read below) and in HBM (the record that is written out) ----
  if (P.o.reverse_per_base_tags && (flags & bam::F_REVERSE) && an > 0) {
    auto put = [&](uint32_t idx, uint8_t v) { R[aux_off + idx] = v; if (staged) g[aux_off + idx] = v; };
    const int REV[14] = {T_cd, T_ce, T_ad, T_ae, T_bd, T_be, T_aq, T_bq, T_cu, T_ct, T_au, T_at, T_bu, T_bt};
#pragma unroll
    for (int q = 0; q < 14; q++) {
      const int k = REV[q];
      if (!rl(my_present, k)) continue;
      const uint32_t ty = rl(my_ty, k), pos = rl((uint32_t)my_pos, k);
      if (ty == 'B') {
        const uint32_t ok = rl(arr_ok, k), cnt = rl(arr_cnt, k), es = rl(arr_es, k);
        if (!ok || cnt == 0) continue;
        const uint32_t e0 = pos + 8;
        for (uint32_t j = lane; j < cnt / 2; j += 64) {
          const uint32_t a = e0 + j * es, b = e0 + (cnt - 1 - j) * es;
          for (uint32_t t = 0; t < es; t++) { const uint8_t x = A[a + t], y = A[b + t]; put(a + t, y); put(b + t, x); }
        }
      } else if (ty == 'Z') {
        if (!rl(z_ok, k)) continue;
        const uint32_t n = rl(z_len, k), s = pos + 3;
        for (uint32_t j = lane; j < n / 2; j += 64) { const uint8_t x = A[s + j], y = A[s + n - 1 - j]; put(s + j, y); put(s + n - 1 - j, x); }
      }
    }
    const int RC[2] = {T_ac, T_bc};
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int k = RC[q];
      if (!rl(z_ok, k)) continue;
      const uint32_t n = rl(z_len, k), s = rl((uint32_t)my_pos, k) + 3;
      for (uint32_t j = lane; j < (n + 1) / 2; j += 64) {
        const uint8_t x = A[s + j], y = A[s + n - 1 - j];
        put(s + j, comp_ascii(y));
        if (n - 1 - j != j) put(s + n - 1 - j, comp_ascii(x));
      }
    }
    __threadfence();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  FPH(2)
  // ---- thresholds and the tags the sweep needs, wave-uniform ----
  const bool duplex = rl(my_present, T_aD) && rl(my_present, T_bD);       // is_duplex_consensus: both tags, any type
  auto src_of = [&](int k) { return Src{rl(src_kind, k), rl(src_off, k), rl(src_cnt, k)}; };
  const bool has_minq = P.o.has_min_base_quality != 0;
  const uint32_t minq = P.o.min_base_quality;
  const uint64_t cc_min = P.o.min_reads[0], ab_min = P.o.min_reads[1], ba_min = P.o.min_reads[2];
  const double cc_be = P.o.max_base_error_rate[0], ab_be = P.o.max_base_error_rate[1], ba_be = P.o.max_base_error_rate[2];
  uint32_t n_masked = 0, n_nocall = 0;
  uint64_t qsum = 0;
  const uint32_t n_pairs = (l_seq + 1) / 2;
  if (!duplex) {
    const Src cd = src_of(T_cd), ce = src_of(T_ce);
    const bool per_base = rl(arr_ok, T_cd) && rl(arr_ok, T_ce);            // both B arrays present (filter.rs:786)
    for (uint32_t j = lane; j < n_pairs; j += 64) {
      const uint8_t b = R[seq_off + j];
      uint8_t nb = b;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t i = 2 * j + h;
        if (i >= l_seq) break;
        const uint32_t q = R[qual_off + i];
        qsum += q;
        const bool was_n = (h == 0 ? (b >> 4) : (b & 0xF)) == 0xF;
        bool m = has_minq && q < minq;
        if (per_base) {
          const uint32_t depth = elem(A, cd.kind, cd.off, cd.count, i), errors = elem(A, ce.kind, ce.off, ce.count, i);
          m = m || (uint64_t)depth < cc_min || (depth > 0 && ((double)errors / (double)depth) > cc_be);
        }
        if (m) { n_masked += !was_n; nb |= h == 0 ? 0xF0 : 0x0F; g[qual_off + i] = 2; }
        n_nocall += (m || was_n);
      }
      if (nb != b) g[seq_off + j] = nb;
    }
  } else {
    const Src ad = src_of(T_ad), ae = src_of(T_ae), bd = src_of(T_bd), be = src_of(T_be), ac = src_of(T_ac), bc = src_of(T_bc);
    const bool ss = P.o.require_single_strand_agreement != 0;
    for (uint32_t j = lane; j < n_pairs; j += 64) {
      const uint8_t b = R[seq_off + j];
      uint8_t nb = b;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t i = 2 * j + h;
        if (i >= l_seq) break;
        const uint32_t q = R[qual_off + i];
        qsum += q;
        const bool was_n = (h == 0 ? (b >> 4) : (b & 0xF)) == 0xF;
        if (was_n) { n_nocall++; continue; }                                   // filter.rs:865
        const uint32_t abd = elem(A, ad.kind, ad.off, ad.count, i), bad = elem(A, bd.kind, bd.off, bd.count, i);
        const uint32_t abe = elem(A, ae.kind, ae.off, ae.count, i), bae = elem(A, be.kind, be.off, be.count, i);
        const uint32_t best_d = abd > bad ? abd : bad, worst_d = abd < bad ? abd : bad, total_d = abd + bad;
        const double ab_r = abd > 0 ? (double)abe / (double)abd : 0.0, ba_r = bad > 0 ? (double)bae / (double)bad : 0.0;
        const double best_r = ab_r < ba_r ? ab_r : ba_r, worst_r = ab_r > ba_r ? ab_r : ba_r;
        const double total_r = total_d > 0 ? (double)(abe + bae) / (double)total_d : 0.0;
        bool m = (has_minq && q < minq) || (uint64_t)total_d < cc_min || total_r > cc_be || (uint64_t)best_d < ab_min || best_r > ab_be ||
                 (uint64_t)worst_d < ba_min || worst_r > ba_be;
        if (ss && abd > 0 && bad > 0) {
          const uint32_t x = i < ac.count ? elem(A, ac.kind, ac.off, ac.count, i) : (uint32_t)'N';
          const uint32_t y = i < bc.count ? elem(A, bc.kind, bc.off, bc.count, i) : (uint32_t)'N';
          m = m || x != y;
        }
        if (m) { n_masked++; nb |= h == 0 ? 0xF0 : 0x0F; g[qual_off + i] = 2; n_nocall++; }
      }
      if (nb != b) g[seq_off + j] = nb;
    }
  }
  FPH(3)
  // ---- the methylation (EM-Seq / TAPs) filters (src/lib/commands/filter.rs:833-886, 924-937; crates/fgumi-consensus/src/filter.rs:925-1340).
  // Only in the <1> instantiation (the host picks it when one of the three options is set).  They run after the masking above like the
  // reference's: a lane re-reads ITS OWN pair bytes from HBM (`g`: the masks above went there, not into the LDS copy), the count arrays
  // come from the (already reversed) LDS copy.  A base that is N by now is skipped, so every newly masked base is counted once.
  bool conv_ok = true;
  if constexpr (METH != 0) {
    const Src cu = src_of(T_cu), ct = src_of(T_ct), au = src_of(T_au), at = src_of(T_at), bu = src_of(T_bu), bt = src_of(T_bt);
    auto ev = [&](const Src& sc, uint32_t i) { return elem(A, sc.kind, sc.off, sc.count, i); };
    const bool has_cuct = rl(arr_ok, T_cu) || rl(arr_ok, T_ct);             // MethylationTags: a B array that parses (filter.rs:462-484)
    // mask_methylation_depth_{simplex,duplex}_raw_with_tags :973-1066
    if (P.o.has_min_methylation_depth && has_cuct) {
      const uint64_t t_cc = P.o.min_methylation_depth[0], t_ab = P.o.min_methylation_depth[1], t_ba = P.o.min_methylation_depth[2];
      for (uint32_t j = lane; j < n_pairs; j += 64) {
        const uint8_t b = g[seq_off + j];
        uint8_t nb = b;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const uint32_t i = 2 * j + h;
          if (i >= l_seq) break;
          if ((h == 0 ? (b >> 4) : (b & 0xF)) == 0xF) continue;
          bool m = (uint64_t)(ev(cu, i) + ev(ct, i)) < t_cc;
          if (duplex) m = m || (uint64_t)(ev(au, i) + ev(at, i)) < t_ab || (uint64_t)(ev(bu, i) + ev(bt, i)) < t_ba;
          if (m) { n_masked++; n_nocall++; nb |= h == 0 ? 0xF0 : 0x0F; g[qual_off + i] = 2; }
        }
        if (nb != b) g[seq_off + j] = nb;
      }
    }
    // resolve_ref_bases_for_record :1072-1133 as a cursor over the CIGAR that every lane advances for its own (increasing) positions: 0 = None.
    // (A ref_id beyond the contigs or a negative pos: regenerate_alignment_tags refuses the record right after, nothing of it is kept.)
    const int32_t ref_id = (int32_t)bam::rd32(R), pos = (int32_t)bam::rd32(R + 4);
    const bool has_map = P.o.regenerate_alignment_tags && !(flags & bam::F_UNMAPPED) && ref_id >= 0 && (uint32_t)ref_id < P.n_ref;
    const bool strand = P.o.require_strand_methylation_agreement && duplex && (rl(arr_ok, T_au) || rl(arr_ok, T_bu));   // :1166-1190
    const bool conv = P.o.has_min_conversion_fraction && P.o.methylation_mode != FGX_METHYLATION_DISABLED && has_cuct;  // :1279-1301
    if (has_map && (strand || conv)) {
      const uint64_t coff = P.contig_off[ref_id], clen = P.contig_len[ref_id];
      const uint8_t* CG = R + 32 + l_name;
      uint32_t ck = 0, cq = 0;
      uint64_t crp = (uint64_t)(int64_t)pos;
      auto ref_at = [&](uint32_t i) -> uint32_t {
        while (ck < n_cig) {
          const uint32_t op = bam::rd32(CG + 4 * ck), t = op & 15, n = op >> 4;
          const bool mm = t == 0 || t == 7 || t == 8;
          if (mm || t == 1 || t == 4) {
            if (i - cq < n) {
              if (!mm) return 0;
              const uint64_t p = crp + (i - cq);
              if (p >= clen) return 0;
              const uint32_t c = P.genome[coff + p];
              return (c >= 'a' && c <= 'z') ? c - 32 : c;
            }
            cq += n;
            if (mm) crp += n;
          } else if (t == 2 || t == 3) crp += n;
          ck++;
        }
        return 0;
      };
      // the pair (k, k+1) is a reference CpG whose strands call the methylation differently (:1197-1224)
      auto discordant = [&](uint32_t k, uint32_t rk, uint32_t rk1) {
        if (rk != 'C' || rk1 != 'G') return false;
        const uint32_t tu = ev(au, k), tc = ev(at, k), qu = ev(bu, k + 1), qc = ev(bt, k + 1);
        if (tu + tc == 0 || qu + qc == 0) return false;
        return (tu > tc) != (qu > qc);
      };
      const bool taps = P.o.methylation_mode == FGX_METHYLATION_TAPS;
      uint64_t num = 0, evi = 0;
      for (uint32_t j = lane; j < n_pairs; j += 64) {
        const uint32_t i0 = 2 * j;
        const uint32_t rm = i0 > 0 ? ref_at(i0 - 1) : 0u, r0 = ref_at(i0), r1 = i0 + 1 < l_seq ? ref_at(i0 + 1) : 0u, r2 = i0 + 2 < l_seq ? ref_at(i0 + 2) : 0u;
        if (strand) {
          const bool dm = i0 > 0 && discordant(i0 - 1, rm, r0), d0 = i0 + 1 < l_seq && discordant(i0, r0, r1), d1 = i0 + 2 < l_seq && discordant(i0 + 1, r1, r2);
          const bool mk[2] = {dm || d0, i0 + 1 < l_seq && (d0 || d1)};
          if (mk[0] || mk[1]) {
            const uint8_t b = g[seq_off + j];
            uint8_t nb = b;
#pragma unroll
            for (int h = 0; h < 2; h++) {
              if (!mk[h] || (h == 0 ? (b >> 4) : (b & 0xF)) == 0xF) continue;
              n_masked++; n_nocall++; nb |= h == 0 ? 0xF0 : 0x0F; g[qual_off + i0 + h] = 2;
            }
            if (nb != b) g[seq_off + j] = nb;
          }
        }
        if (conv) {
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const uint32_t i = i0 + h;
            if (i >= l_seq) break;
            const uint32_t ri = h == 0 ? r0 : r1, rn = h == 0 ? r1 : r2;           // (rn is None past the read: `i + 1 < len` of :1313)
            if (ri != 'C' || rn == 'G') continue;
            const uint32_t u = ev(cu, i), t = ev(ct, i);
            if (u + t > 0) { num += taps ? u : t; evi += u + t; }
          }
        }
      }
      if (conv) {
        const uint64_t num_t = wave_sum(num), evi_t = wave_sum(evi);
        if (evi_t > 0 && !((double)num_t / (double)evi_t >= P.o.min_conversion_fraction)) conv_ok = false;
      }
    }
  }
  const uint64_t masked_total = wave_sum(n_masked), nocall_total = wave_sum(n_nocall), qsum_total = wave_sum(qsum);

  // ---- read-level filters (filter_read / filter_duplex_read, check_no_call_and_quality) ----
  auto int_of = [&](int k, bool& has) { has = rl(has_int, k) != 0; return (int64_t)(((uint64_t)rl(iv_hi, k) << 32) | rl(iv_lo, k)); };
  auto f_of = [&](int k, bool& has) { has = rl(has_f, k) != 0; return __uint_as_float(rl(fbits, k)); };
  bool hd, he;
  const int64_t depth = int_of(T_cD, hd);
  const float err = f_of(T_cE, he);
  if (!hd || !he) { if (lane == 0) { report(P.error, r, ERR_NO_TAGS); P.pass[r] = 0; P.masked[r] = 0; } return; }
  bool ok = !(depth < (int64_t)cc_min) && !((double)err > P.o.max_read_error_rate[0]);
  if (ok && duplex) {
    bool ha, hb, h2, hae, hbe;
    int64_t a_d = int_of(T_aD, ha);
    if (!ha) { a_d = int_of(T_aM, h2); ha = h2; }
    int64_t b_d = int_of(T_bD, hb);
    if (!hb) { b_d = int_of(T_bM, h2); hb = h2; }
    const float a_e = f_of(T_aE, hae), b_e = f_of(T_bE, hbe);
    if (ha || hb) {
      int64_t worst_d, best_d;
      if (ha && hb) { if (a_d < b_d) { worst_d = a_d; best_d = b_d; } else { worst_d = b_d; best_d = a_d; } }
      else if (ha) { worst_d = 0; best_d = a_d; }
      else { worst_d = 0; best_d = b_d; }
      float best_e, worst_e;
      if (hae && hbe) { if (a_e < b_e) { best_e = a_e; worst_e = b_e; } else { best_e = b_e; worst_e = a_e; } }
      else if (hae) best_e = worst_e = a_e;
      else if (hbe) best_e = worst_e = b_e;
      else best_e = worst_e = 0.f;
      ok = !((uint64_t)best_d < ab_min) && !((double)best_e > P.o.max_read_error_rate[1]) && !((uint64_t)worst_d < ba_min) &&
           !((double)worst_e > P.o.max_read_error_rate[2]);
    }
  }
  if (ok && P.o.has_min_mean_base_quality) {
    const double mean = l_seq == 0 ? 0.0 : (double)qsum_total / (double)l_seq;      // over the full read, before masking
    if (mean < P.o.min_mean_base_quality) ok = false;
  }
  if (ok) {
    if (P.o.max_no_call_fraction >= 1.0) ok = (double)nocall_total <= P.o.max_no_call_fraction;
    else ok = (l_seq > 0 ? (double)nocall_total / (double)l_seq : 0.0) <= P.o.max_no_call_fraction;
  }
  if (!conv_ok) ok = false;
  if (lane == 0) { P.pass[r] = ok ? 1 : 0; P.masked[r] = (uint32_t)masked_total; }
  FPH(4)
}

#ifndef FGX_FILTER_OCC
#define FGX_FILTER_OCC 8
#endif
template <int METH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FGX_FILTER_OCC, FGX_FILTER_OCC))) void k_filter_records(const FilterParams P) {
  extern __shared__ __align__(16) uint8_t lds_raw[];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t r = blockIdx.x * 4 + wave;
  if (r >= P.n_rec) return;
#if FGX_PHASE_TIMING
  unsigned long long _t = __builtin_amdgcn_s_memtime();
#endif
  const uint64_t off = P.rec_off[r];
  const uint32_t len = P.rec_len[r];
  if (len < 32 || off + len > P.blob_len) { if (lane == 0) { report(P.error, r, ERR_SHORT); P.pass[r] = 0; P.masked[r] = 0; } return; }
  uint8_t* g = P.blob + off;
  // stage the record into this wave's LDS slice, same alignment mod 4 as in HBM so that the copy is whole dwords
  const uint32_t shift = (uint32_t)((uintptr_t)g & 3);
  const uint32_t n_dw = (shift + len + 3) / 4;
  // whole-dword staging reads up to 3 bytes either side of the record: only when those bytes are still inside the blob
  if ((uint64_t)n_dw * 4 <= P.lds_slice && off - shift + (uint64_t)n_dw * 4 <= P.blob_len) {
    uint8_t* slice = lds_raw + (size_t)wave * P.lds_slice;
    const uint32_t* src = (const uint32_t*)(g - shift);
    uint32_t* dst = (uint32_t*)slice;
    for (uint32_t base = 0; base < n_dw; base += 256) {       // four loads in flight per lane before the first LDS write
      uint32_t v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint32_t i = base + k * 64 + lane; v[k] = src[i < n_dw ? i : n_dw - 1]; }
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint32_t i = base + k * 64 + lane; if (i < n_dw) dst[i] = v[k]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    filter_body<METH>(P, r, lane, g, slice + shift, true, len FPH_PASS);
  } else {
    filter_body<METH>(P, r, lane, g, g, false, len FPH_PASS);      // beyond the slice: read from HBM
  }
}

// ---- templates ------------------------------------------------------------------------------------------------------
__global__ void k_template_flags(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_off, const uint32_t* __restrict__ rec_len, uint32_t n,
                                 uint32_t* __restrict__ newt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t nt = 1;
  if (i > 0 && rec_len[i] >= 32 && rec_len[i - 1] >= 32) {
    const uint8_t* a = blob + rec_off[i - 1];
    const uint8_t* b = blob + rec_off[i];
    const uint32_t la = a[8] > 1 ? a[8] - 1u : 0u, lb = b[8] > 1 ? b[8] - 1u : 0u;
    if (la == lb && 32ull + lb <= rec_len[i] && 32ull + la <= rec_len[i - 1]) {
      bool same = true;
      for (uint32_t k = 0; k < lb && same; k++) same = a[32 + k] == b[32 + k];
      if (same) nt = 0;
    }
  }
  newt[i] = nt;
}

__global__ void k_template_firsts(const uint32_t* __restrict__ newt, const uint32_t* __restrict__ incl, uint32_t n, uint32_t* __restrict__ first) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (newt[i]) first[incl[i] - 1] = i;
  if (i == n - 1) first[incl[i]] = n;
}

struct DecideParams {
  const uint8_t* blob; const uint64_t* rec_off; const uint32_t* rec_len; uint32_t n_rec;
  const uint32_t* tmpl_first; uint32_t n_tmpl;       // tmpl_first == nullptr: every record is its own unit (single-read mode)
  const uint8_t* pass; const uint32_t* masked;
  uint32_t track_rejects;
  uint32_t* ord_src; uint64_t* keep_size; uint64_t* rej_size;
  unsigned long long* counters;                      // [0] passed, [1] rejected, [2] bases masked
  unsigned long long* error;
};

__device__ inline int category(uint32_t f) {        // template.rs:252-282
  const bool sec = f & bam::F_SECONDARY, sup = f & bam::F_SUPPLEMENTARY, r1 = !(f & bam::F_PAIRED) || (f & bam::F_FIRST);
  if (r1) return sec ? 4 : sup ? 2 : 0;
  return sec ? 5 : sup ? 3 : 1;
}

__global__ void k_template_decide(const DecideParams P) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t n_pass = 0, n_rej = 0, n_mask = 0;
  if (t < P.n_tmpl) {
    if (!P.tmpl_first) {                             // filter.rs:581-625
      const uint32_t f = bam::rd16(P.blob + P.rec_off[t] + 14);
      const bool prim = !(f & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)), keep = P.pass[t] != 0;
      const uint64_t sz = 4ull + P.rec_len[t];
      P.ord_src[t] = t;
      P.keep_size[t] = keep ? sz : 0;
      P.rej_size[t] = (!keep && P.track_rejects) ? sz : 0;
      n_pass = keep; n_rej = (!keep && P.track_rejects); n_mask = (keep && prim) ? P.masked[t] : 0;
    } else {                                         // filter.rs:660-721 on the Template::from_records order
      const uint32_t a = P.tmpl_first[t], b = P.tmpl_first[t + 1];
      uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
      bool all_pass = true;
      for (uint32_t i = a; i < b; i++) {
        const uint32_t f = bam::rd16(P.blob + P.rec_off[i] + 14);
        const int c = category(f);
        cnt[c]++;
        if (c < 2 && !P.pass[i]) all_pass = false;
      }
      if (cnt[0] > 1) report(P.error, a, ERR_MULTI_R1);
      if (cnt[1] > 1) report(P.error, a, ERR_MULTI_R2);
      const bool tpass = (cnt[0] + cnt[1]) > 0 && all_pass;
      uint32_t base[6], seen[6] = {0, 0, 0, 0, 0, 0};
      base[0] = 0;
      for (int c = 1; c < 6; c++) base[c] = base[c - 1] + cnt[c - 1];
      for (uint32_t i = a; i < b; i++) {
        const uint32_t f = bam::rd16(P.blob + P.rec_off[i] + 14);
        const int c = category(f);
        const uint32_t rank = base[c] + (c < 2 ? seen[c] : cnt[c] - 1 - seen[c]);     // supplementaries / secondaries come out reversed
        seen[c]++;
        const bool prim = c < 2;
        const bool keep = prim ? tpass : (tpass && P.pass[i] != 0);
        const uint64_t sz = 4ull + P.rec_len[i];
        const uint32_t o = a + rank;
        P.ord_src[o] = i;
        P.keep_size[o] = keep ? sz : 0;
        P.rej_size[o] = (!keep && P.track_rejects) ? sz : 0;
        n_pass += keep; n_rej += (!keep && P.track_rejects);
        if (tpass && prim) n_mask += P.masked[i];
      }
    }
  }
  n_pass = wave_sum(n_pass); n_rej = wave_sum(n_rej); n_mask = wave_sum(n_mask);
  if ((threadIdx.x & 63) == 0) {
    if (n_pass) atomicAdd(&P.counters[0], (unsigned long long)n_pass);
    if (n_rej) atomicAdd(&P.counters[1], (unsigned long long)n_rej);
    if (n_mask) atomicAdd(&P.counters[2], (unsigned long long)n_mask);
  }
}

// ---- output: one wavefront per record, [block_size][record] at its scanned offset -------------------------------------------
struct CopyParams {
  const uint8_t* blob; uint64_t blob_len; const uint64_t* rec_off; const uint32_t* rec_len; uint32_t n_rec;
  const uint32_t* ord_src; const uint64_t* keep_size; const uint64_t* rej_size; const uint64_t* keep_off; const uint64_t* rej_off;
  uint8_t* out_keep; uint8_t* out_rej;
};

__global__ __launch_bounds__(256) void k_copy_records(const CopyParams P) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= P.n_rec) return;
  uint8_t* D;
  if (P.keep_size[o]) D = P.out_keep + P.keep_off[o];
  else if (P.rej_size[o]) D = P.out_rej + P.rej_off[o];
  else return;
  const uint32_t src = P.ord_src[o];
  const uint32_t len = P.rec_len[src];
  const uint8_t* S = P.blob + P.rec_off[src];
  const uint32_t total = len + 4;
  const uint8_t* blob_end = P.blob + P.blob_len;
  auto out_byte = [&](uint32_t k) -> uint8_t { return k < 4 ? (uint8_t)(len >> (8 * k)) : S[k - 4]; };
  const uint32_t head = (uint32_t)((4 - ((uintptr_t)D & 3)) & 3);                       // bytes before the first aligned destination dword
  const uint32_t n_dw = (total - (head < total ? head : total)) / 4;
  const uint32_t tail0 = head + 4 * n_dw;
  if (lane < head && lane < total) D[lane] = out_byte(lane);
  if (lane >= 32 && lane - 32 < total - tail0 && tail0 < total) D[tail0 + (lane - 32)] = out_byte(tail0 + (lane - 32));
  uint32_t* D4 = (uint32_t*)(D + head);
  for (uint32_t w = lane; w < n_dw; w += 64) {
    const uint32_t k = head + 4 * w;                                                   // output byte index of this dword
    uint32_t v;
    const uint8_t* s = S + ((int64_t)k - 4);
    const uint32_t sh = (uint32_t)((uintptr_t)s & 3);
    if (k < 4 || (s - sh) + 8 > blob_end)        // the prefix, or an aligned pair of source dwords that would cross the end of the blob
      v = (uint32_t)out_byte(k) | ((uint32_t)out_byte(k + 1) << 8) | ((uint32_t)out_byte(k + 2) << 16) | ((uint32_t)out_byte(k + 3) << 24);
    else {
      const uint32_t* sa = (const uint32_t*)(s - sh);
      const uint32_t lo = sa[0];
      v = sh ? (lo >> (8 * sh)) | (sa[1] << (32 - 8 * sh)) : lo;
    }
    D4[w] = v;
  }
}

// ---- --ref: NM / UQ / MD regenerated after the masking (filter.rs:888-890; aln_tags_core.h) ---------------------------------------------------
// A lane per record runs the scalar source the CPU tests prove against the oracle: k_aln_plan measures the edited record (the sizes the
// template kernel and the scans work with), k_aln_write produces it at its scanned place — every record, kept or rejected, as the reference
// edits a record before it decides about it.  The stores scatter and the walks diverge: this is the simple form (a consensus BAM has an
// eighth of the raw reads), not the fast one.
struct AlnParams {
  const uint8_t* blob; uint64_t blob_len; const uint64_t* rec_off; const uint32_t* rec_len; uint32_t n_rec;
  const uint8_t* genome; const uint64_t* contig_off; const uint64_t* contig_len; uint32_t n_ref;
  uint32_t* new_len; unsigned long long* error;
  // k_aln_write
  const uint32_t* ord_src; const uint64_t* keep_size; const uint64_t* rej_size; const uint64_t* keep_off; const uint64_t* rej_off; uint8_t* out_keep; uint8_t* out_rej;
};
__global__ void k_aln_plan(const AlnParams P) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= P.n_rec) return;
  const uint32_t len = P.rec_len[r];
  const uint64_t off = P.rec_off[r];
  if (len < 32 || off + len > P.blob_len) { P.new_len[r] = len; return; }     // (k_filter_records has reported it)
  aln::Plan pl; aln::Geometry G{};
  aln::plan(P.blob + off, len, P.genome, P.contig_off, P.contig_len, P.n_ref, pl, G);
  if (pl.status > aln::ALN_REMOVED) { report(P.error, r, ERR_ALN + (uint32_t)pl.status); P.new_len[r] = len; return; }
  P.new_len[r] = pl.new_len;
}
__global__ void k_aln_write(const AlnParams P) {
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= P.n_rec) return;
  uint8_t* D;
  if (P.keep_size[o]) D = P.out_keep + P.keep_off[o];
  else if (P.rej_size[o]) D = P.out_rej + P.rej_off[o];
  else return;
  const uint32_t src = P.ord_src[o];
  const uint32_t len = P.rec_len[src];
  const uint8_t* rec = P.blob + P.rec_off[src];
  aln::Plan pl; aln::Geometry G{};
  aln::plan(rec, len, P.genome, P.contig_off, P.contig_len, P.n_ref, pl, G);
  const uint32_t nl = pl.new_len;
  D[0] = (uint8_t)nl; D[1] = (uint8_t)(nl >> 8); D[2] = (uint8_t)(nl >> 16); D[3] = (uint8_t)(nl >> 24);
  aln::write(rec, len, P.genome, P.contig_off, pl, G, D + 4);
}

}  // namespace

// Device buffers in (masked in place), device buffers out; fills the counters and the device pointers of `out`.
int filter_records_device(fgx_caller* c, FilterBuffers& B, const fgx_filter_options* o, uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off,
                          const uint32_t* d_rec_len, uint32_t n, fgx_filter_output* out) {
  hipStream_t s = c->stream;
  memset(out, 0, sizeof(*out));
  out->records_count = n;
  if (n == 0) return 0;
  B.pass.reserve(n + 64);
  B.masked.reserve((size_t)n * 4 + 64);
  B.newt.reserve((size_t)n * 4 + 64);
  B.incl.reserve((size_t)n * 4 + 64);
  B.first.reserve((size_t)(n + 1) * 4 + 64);
  B.ord_src.reserve((size_t)n * 4 + 64);
  B.keep_size.reserve((size_t)n * 8 + 64);
  B.rej_size.reserve((size_t)n * 8 + 64);
  B.keep_off.reserve((size_t)n * 8 + 64);
  B.rej_off.reserve((size_t)n * 8 + 64);
  B.misc.reserve(64);
  unsigned long long* d_err = B.misc.as<unsigned long long>();
  unsigned long long* d_cnt = d_err + 1;
  unsigned long long init[4] = {~0ull, 0, 0, 0};
  hip_check(hipMemcpyAsync(d_err, init, sizeof(init), hipMemcpyHostToDevice, s), "H2D filter counters");

  FilterParams P{};
  P.blob = d_blob; P.blob_len = blob_len; P.rec_off = d_rec_off; P.rec_len = d_rec_len; P.n_rec = n; P.o = *o;
  // LDS slice per wavefront: 1.5 x the mean record (records beyond it are read from HBM — correct, slower), small enough to keep
  // the CU full of waves for short records
  uint32_t slice = B.lds_slice;
  if (slice == 0) {
    const uint64_t mean = blob_len / n;
    slice = (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(1536, ((mean * 3 / 2 + 64 + 255) / 256) * 256));
  }
  P.pass = B.pass.as<uint8_t>(); P.masked = B.masked.as<uint32_t>(); P.error = d_err; P.lds_slice = slice;
  const dim3 block(256), grid_w((n + 3) / 4), grid_t((n + 255) / 256);
  // --ref: the contig table next to the genome of fgx_set_reference (the reference-dependent methylation filters and the NM / UQ / MD kernels)
  const bool regen = o->regenerate_alignment_tags != 0;
  const GenomeRef* gr = regen ? c->genome.get() : nullptr;
  const uint32_t n_ref = gr ? (uint32_t)gr->off.size() : 0u;
  if (regen) {
    B.aln_contigs.reserve((size_t)(n_ref + 1) * 16 + 64);
    if (n_ref) {
      hip_check(hipMemcpyAsync(B.aln_contigs.p, gr->off.data(), (size_t)n_ref * 8, hipMemcpyHostToDevice, s), "H2D contig offsets");
      hip_check(hipMemcpyAsync(B.aln_contigs.as<uint64_t>() + n_ref, gr->len.data(), (size_t)n_ref * 8, hipMemcpyHostToDevice, s), "H2D contig lengths");
    }
    P.genome = gr ? (const uint8_t*)gr->d_genome.p : nullptr; P.contig_off = B.aln_contigs.as<uint64_t>(); P.contig_len = B.aln_contigs.as<uint64_t>() + n_ref; P.n_ref = n_ref;
  }
  if (o->has_min_methylation_depth || o->require_strand_methylation_agreement || o->has_min_conversion_fraction)
    hipLaunchKernelGGL(k_filter_records<1>, grid_w, block, (size_t)slice * 4, s, P);
  else
    hipLaunchKernelGGL(k_filter_records<0>, grid_w, block, (size_t)slice * 4, s, P);

  uint32_t n_tmpl = n;
  const uint32_t* d_first = nullptr;
  if (o->filter_by_template) {
    uint32_t* d_newt = B.newt.as<uint32_t>();
    uint32_t* d_incl = B.incl.as<uint32_t>();
    hipLaunchKernelGGL(k_template_flags, grid_t, block, 0, s, d_blob, d_rec_off, d_rec_len, n, d_newt);
    size_t tb = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, tb, d_newt, d_incl, (int)n, s);
    B.scan_tmp.reserve(tb + 64);
    hip_check(hipcub::DeviceScan::InclusiveSum(B.scan_tmp.p, tb, d_newt, d_incl, (int)n, s), "scan templates");
    hipLaunchKernelGGL(k_template_firsts, grid_t, block, 0, s, d_newt, d_incl, n, B.first.as<uint32_t>());
    hip_check(hipMemcpyAsync(&n_tmpl, d_incl + (n - 1), 4, hipMemcpyDeviceToHost, s), "D2H template count");
    hip_check(hipStreamSynchronize(s), "sync");
    d_first = B.first.as<uint32_t>();
  }
  // --ref: the edited records' lengths (every record: the reference edits before it decides)
  AlnParams A{};
  if (regen) {
    B.aln_len.reserve((size_t)n * 4 + 64);
    A.blob = d_blob; A.blob_len = blob_len; A.rec_off = d_rec_off; A.rec_len = d_rec_len; A.n_rec = n;
    A.genome = gr ? (const uint8_t*)gr->d_genome.p : nullptr; A.contig_off = B.aln_contigs.as<uint64_t>(); A.contig_len = B.aln_contigs.as<uint64_t>() + n_ref; A.n_ref = n_ref;
    A.new_len = B.aln_len.as<uint32_t>(); A.error = d_err;
    hipLaunchKernelGGL(k_aln_plan, grid_t, block, 0, s, A);
  }
  DecideParams Q{};
  Q.blob = d_blob; Q.rec_off = d_rec_off; Q.rec_len = regen ? (const uint32_t*)A.new_len : d_rec_len; Q.n_rec = n; Q.tmpl_first = d_first; Q.n_tmpl = n_tmpl;
  Q.pass = P.pass; Q.masked = P.masked; Q.track_rejects = o->track_rejects;
  Q.ord_src = B.ord_src.as<uint32_t>(); Q.keep_size = B.keep_size.as<uint64_t>(); Q.rej_size = B.rej_size.as<uint64_t>();
  Q.counters = d_cnt; Q.error = d_err;
  hipLaunchKernelGGL(k_template_decide, dim3((n_tmpl + 255) / 256), block, 0, s, Q);

  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, Q.keep_size, B.keep_off.as<uint64_t>(), (int)n, s);
  B.scan_tmp.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(B.scan_tmp.p, tb, Q.keep_size, B.keep_off.as<uint64_t>(), (int)n, s), "scan kept sizes");
  if (o->track_rejects) hip_check(hipcub::DeviceScan::ExclusiveSum(B.scan_tmp.p, tb, Q.rej_size, B.rej_off.as<uint64_t>(), (int)n, s), "scan rejected sizes");
  uint64_t last[4] = {0, 0, 0, 0};
  unsigned long long res[4];
  hip_check(hipMemcpyAsync(&last[0], B.keep_off.as<uint64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(&last[1], Q.keep_size + (n - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  if (o->track_rejects) {
    hip_check(hipMemcpyAsync(&last[2], B.rej_off.as<uint64_t>() + (n - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipMemcpyAsync(&last[3], Q.rej_size + (n - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
  }
  hip_check(hipMemcpyAsync(res, d_err, sizeof(res), hipMemcpyDeviceToHost, s), "D2H filter counters");
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "filter kernels");
  if (res[0] != ~0ull) {
    const uint32_t code = (uint32_t)(res[0] & 0xFF);
    const unsigned long long rec = res[0] >> 8;
    const char* msg = code == ERR_SHORT    ? "BAM record too short"
                      : code == ERR_MAPPED ? "--ref is required when filtering mapped reads to keep NM/UQ/MD tags consistent"
                      : code == ERR_NO_TAGS
                          ? "read does not appear to have consensus calling tags (cD/cE) present; FilterConsensusReads requires reads produced by consensus calling"
                      : code == ERR_MULTI_R1 ? "Multiple non-secondary, non-supplemental R1 records for a read name"
                      : code == ERR_MULTI_R2 ? "Multiple non-secondary, non-supplemental R2 records for a read name"
                      // --ref (regenerate_alignment_tags_raw, crates/fgumi-sam/src/alignment_tags.rs:259-433)
                      : code == ERR_ALN + aln::ALN_TOO_SHORT ? "BAM record too short"
                      : code == ERR_ALN + aln::ALN_REF_ID ? "Reference sequence ID not found in header"
                      : code == ERR_ALN + aln::ALN_BAD_START ? "Invalid alignment start position"
                      : code == ERR_ALN + aln::ALN_REGION ? "the alignment leaves its reference sequence (region out of bounds, or the FASTA lacks the contig)"
                      : code == ERR_ALN + aln::ALN_TRUNCATED ? "Truncated BAM record: seq/qual extends past record end"
                                                             : "CIGAR consumes more bases than sequence length";
    c->err = std::string(msg) + " (record " + std::to_string(rec) + ")";
    return 2;
  }
  const uint64_t keep_total = last[0] + last[1], rej_total = last[2] + last[3];
  B.out_keep.reserve(keep_total + 64);
  B.out_rej.reserve(rej_total + 64);
  CopyParams C{};
  C.blob = d_blob; C.blob_len = blob_len; C.rec_off = d_rec_off; C.rec_len = d_rec_len; C.n_rec = n; C.ord_src = Q.ord_src; C.keep_size = Q.keep_size; C.rej_size = Q.rej_size;
  C.keep_off = B.keep_off.as<uint64_t>(); C.rej_off = B.rej_off.as<uint64_t>(); C.out_keep = B.out_keep.as<uint8_t>(); C.out_rej = B.out_rej.as<uint8_t>();
  if (regen) {
    A.ord_src = Q.ord_src; A.keep_size = Q.keep_size; A.rej_size = Q.rej_size; A.keep_off = C.keep_off; A.rej_off = C.rej_off; A.out_keep = C.out_keep; A.out_rej = C.out_rej;
    hipLaunchKernelGGL(k_aln_write, grid_t, block, 0, s, A);
  } else hipLaunchKernelGGL(k_copy_records, grid_w, block, 0, s, C);
  hip_check(hipStreamSynchronize(s), "sync");
  hip_check(hipGetLastError(), "k_copy_records");
  out->data = B.out_keep.as<uint8_t>(); out->data_len = keep_total;
  out->rejects = B.out_rej.as<uint8_t>(); out->rejects_len = rej_total;
  out->passed_count = res[1]; out->rejected_count = res[2]; out->bases_masked = res[3];
  return 0;
}

// Slot table of a device-resident consensus batch (3 slots per group, empty slots have size 0) → record list → filter.
namespace {
__global__ void k_slot_flags(const uint64_t* __restrict__ size, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = size[i] > 4 ? 1u : 0u;
}
__global__ void k_slot_records(const uint64_t* __restrict__ off, const uint64_t* __restrict__ size, const uint32_t* __restrict__ flag,
                               const uint32_t* __restrict__ pos, uint32_t n, uint64_t* __restrict__ rec_off, uint32_t* __restrict__ rec_len) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) { rec_off[pos[i]] = off[i] + 4; rec_len[pos[i]] = (uint32_t)(size[i] - 4); }
}
}  // namespace

int filter_slots_device(fgx_caller* c, FilterBuffers& B, const fgx_filter_options* o, uint8_t* d_out, uint64_t out_len, const uint64_t* d_slot_off,
                        const uint64_t* d_slot_size, uint32_t n_slots, fgx_filter_output* out) {
  hipStream_t s = c->stream;
  if (n_slots == 0) { memset(out, 0, sizeof(*out)); return 0; }
  B.slot_flag.reserve((size_t)n_slots * 4 + 64);
  B.slot_pos.reserve((size_t)n_slots * 4 + 64);
  B.in_off.reserve((size_t)n_slots * 8 + 8);
  B.in_len.reserve((size_t)n_slots * 4 + 4);
  uint32_t* d_flag = B.slot_flag.as<uint32_t>();
  uint32_t* d_pos = B.slot_pos.as<uint32_t>();
  const dim3 block(256), grid((n_slots + 255) / 256);
  hipLaunchKernelGGL(k_slot_flags, grid, block, 0, s, d_slot_size, n_slots, d_flag);
  size_t tb = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_flag, d_pos, (int)n_slots, s);
  B.scan_tmp.reserve(tb + 64);
  hip_check(hipcub::DeviceScan::ExclusiveSum(B.scan_tmp.p, tb, d_flag, d_pos, (int)n_slots, s), "scan slots");
  hipLaunchKernelGGL(k_slot_records, grid, block, 0, s, d_slot_off, d_slot_size, d_flag, d_pos, n_slots, B.in_off.as<uint64_t>(), B.in_len.as<uint32_t>());
  uint32_t last[2] = {0, 0};
  hip_check(hipMemcpyAsync(&last[0], d_pos + (n_slots - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipMemcpyAsync(&last[1], d_flag + (n_slots - 1), 4, hipMemcpyDeviceToHost, s), "D2H");
  hip_check(hipStreamSynchronize(s), "sync");
  return filter_records_device(c, B, o, d_out, out_len, B.in_off.as<uint64_t>(), B.in_len.as<uint32_t>(), last[0] + last[1], out);
}

}  // namespace fgx

#if FGX_PHASE_TIMING
extern "C" int fgx_debug_filter_phase_cycles(unsigned long long* out8, int reset) {
  unsigned long long h[64 * 8];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fgx::g_fphase), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 8; i++) { out8[i] = 0; for (int b = 0; b < 64; b++) out8[i] += h[b * 8 + i]; }
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(fgx::g_fphase), h, sizeof(h)) != hipSuccess) return 1; }
  return 0;
}
#endif
