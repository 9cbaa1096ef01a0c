// fastpath.hip — the device-resident pipelines of the three callers: raw BAM records in HBM → consensus BAM records in HBM,
// no host work per family.  Sections, in file order:
//
//   shared device helpers     ReadInfo / Shared (LDS state of the workgroup kernel), name_rank (fgbio Murmur3 name rank), general-CIGAR
//                             helpers (map_ref_to_query, simplified CIGARs, alignment_filter = select_most_common_alignment_group)
//   k_family                  ONE WORKGROUP (512 threads) PER FAMILY: families with more than 64 records, a byte span beyond the largest
//                             wave slice, or any read with an indel / skip.  Records copied HBM→LDS once; a thread per record parses
//                             (general CIGARs up to 6 ops), unpacks, pairs mates, corrects overlaps through both CIGARs, computes the
//                             source-read geometry, runs the gates (alignment filter, --max-reads downsampling, min-reads at every
//                             shrink point, orphan rule); a thread per column calls the consensus; per-character UMI consensus.
//   k_family_wave<MODE>       ONE WAVEFRONT PER FAMILY / MOLECULE, the common case.  MODE 0 simplex, 1 duplex (strand partition, four
//                             single-strand column sets, descriptors of the two duplex records), 2 CODEC (FR pairs, virtual hard clip
//                             against the mate, overlap geometry, two strand column sets).  Phases: stage raw records into LDS → parse
//                             (lane = record) → overlap pre-correction → geometry → gates → columns (lane = column, reads walked in file
//                             order: summation order is observable) → UMI consensus → descriptors + stats.  Columns the unanimous fast
//                             path cannot decide go to 1024 chained lists for k_call_full.
//   k_call_full               one lane per deferred column (or UMI character): ln_sum_exp chain on the device libm, tie rule, Phred.
//   k_emit / k_emit_duplex[_fast] / k_emit_codec[_fast]
//                             one wavefront per consensus record: header, name `<prefix>:<MI>`, 4-bit sequence, qualities and the
//                             caller's tag set at the record's scanned offset (the _fast writers take the records of common shape,
//                             the generic per-field writers the rest); duplex / CODEC combine the strands here.
//   k_col_bound, k_reduce_stats, FastPath::run
//                             column-slot bounds, counter reduction, and the host orchestration: staged wave launches over growing
//                             LDS slices with retry lists (4 / 2 / 1 wavefronts per workgroup), workgroup kernel for what is left,
//                             scans, emit.
//
// Reference semantics restated (bit-exact contract), paths under crates/fgumi-consensus/src/ unless noted:
//   src/lib/commands/simplex.rs:669-701 (group skip, overlap pre-step, consensus_reads)
//   overlapping.rs:236-336, 565-684 ; vanilla_caller.rs:48-120, 862-932, 1080-1296, 1304-1422, 1454-1646, 1652-1755, 1767-1881 ;
//   simple_umi.rs:46-117 ; duplex_caller.rs:560-1405, 1837-2624 ; codec_caller.rs:504-1911 ;
//   crates/fgumi-raw-bam/src/{overlap.rs:21-357, hash.rs:14-89, builder.rs:122-301, cigar.rs:404-500}
//
// Families the device does not decide are DEFERRED untouched to the general host path (simplex_host.cpp, duplex_host.cpp,
// codec_host.cpp): reads with more than 6 CIGAR ops or SEQ / CIGAR length mismatch, unmapped reads, malformed records (so that
// the general path raises the reference's fatal error), more than FAST_MAX_READS reads or more LDS than the launch provides,
// duplex / CODEC molecules with indels or a biting per-strand read cap.  See DESIGN.md 4.
#if !defined(FGX_WAVEMU)      // (tests/wavemu compiles this file for the HOST under a 64-lane lock-step shim, with stand-ins for the runtime and the scans)
#include <hipcub/hipcub.hpp>
#endif
#include "bamrec.h"
#include "engine.h"
#include <cstdlib>
#include <type_traits>
#include "fastpath.h"
#include "gate_core.h"
#include "packed_core.h"

// The device-only spellings behind macros, so that tests/wavemu (a wave-level host emulator of the wavefront kernels) can compile this
// file for the host: dynamic LDS, and the inline assembly — scheduling pins for the device compiler ("+v" / "s" constraints mean nothing on a CPU) and
// four scalar-unit idioms with plain C++ twins.
#if defined(FGX_WAVEMU)
#define FGX_DYN_LDS(name) uint8_t* const name = wavemu::dyn_lds()
#define FGX_PIN(...) ((void)0)
#define FGX_UNDEF4(a, b, c, d) do { a = 0; b = 0; c = 0; d = 0; } while (0)
#define FGX_MEM_PIN() ((void)0)
#define FGX_CONST_AS
#else
#define FGX_DYN_LDS(name) extern __shared__ __align__(16) uint8_t name[]
#define FGX_PIN(...) asm volatile("" : __VA_ARGS__)
#define FGX_UNDEF4(a, b, c, d) asm volatile("" : "=v"(a), "=v"(b), "=v"(c), "=v"(d))     /* "whatever the registers hold" costs no instruction */
#define FGX_MEM_PIN() asm volatile("" ::: "memory")
#define FGX_CONST_AS __attribute__((address_space(4)))
#endif

namespace fgx {

using bam::rd16;
using bam::rd32;

namespace {

#ifndef FGX_BLOCK_NT
#define FGX_BLOCK_NT 512
#endif
constexpr int NT = FGX_BLOCK_NT;        // threads per family workgroup
constexpr int FAST_MAX_READS = 128;     // per-read LDS tables
constexpr int MAX_MC_OPS = 8;
constexpr uint32_t MAX_CIG_OPS = 6;     // clips + aligned ops of a read the WAVEFRONT kernels take
// Round 5: the workgroup-per-family kernel (k_family: indel / clipped simplex families of up to 128 records) takes reads of up to 16 CIGAR ops
// and mates whose MC tag holds up to 17 — real aligner output carries such CIGARs in a fraction of a percent of its reads, and round 4 sent
// every family that holds one through the general path (host orchestration, ~5 M reads/s); the wavefront kernels hand them on
constexpr uint32_t WG_CIG_OPS = 16;
constexpr int WG_MC_OPS = 17;
constexpr uint32_t COL_BOUND_PAD = 48;   // columns on top of a family's column bound: the padding of the packed build's scratch layout (simplex_split.inc, round 6)
constexpr int STAT_SLOTS = 1024;        // spread the per-batch counters over many addresses (atomic contention)

struct ReadInfo {          // LDS, one per record of the family
  uint64_t goff;           // record body offset in the blob
  int32_t pos, ref_id;
  uint32_t name_hash;
  uint32_t row;            // row offset (bytes) of this read's bases / quals in the LDS tiles
  uint16_t l_seq, flags, seq_off, name_len;
  uint16_t final_len, trim_to, clip, mi_off;
  uint16_t rx_off, cb_off;
  int16_t mate;            // index of the R2 paired with this R1 (only set on the R1), else -1
  uint8_t mi_len, rx_len, cb_len, end;   // end: 0 fragment, 1 R1, 2 R2, 255 = not a candidate
  uint8_t has_mi, has_rx, has_cb, all_ff;
  uint8_t excluded, zero_len, minority, _p1;   // minority: dropped by the alignment filter (select_most_common_alignment_group)
};

struct Shared {
  ReadInfo ri[FAST_MAX_READS];
  uint16_t members[FAST_MAX_READS];     // kept reads, grouped by end, file order inside an end
  uint32_t stats[FGX_STATS_LEN];
  uint32_t defer;
  uint32_t n_ends;
  uint32_t end_type[3], end_first[3], end_cnt[3], end_len[3], end_coloff[3];
  uint32_t end_maxd[3], end_mind[3], end_sumd[3], end_sume[3];
  uint32_t end_rx_cnt[3], end_rx_len[3], end_rx_first[3];
  char end_rx[3][FAST_RX_CAP];
  uint32_t rx_bad;
  uint64_t col_base;
  uint32_t tile_bytes;
  unsigned long long raw_lo, raw_hi;    // byte span of the family's records in the blob
  uint32_t g_wcnt[2][3];                // gates: kept reads per wave and end
  uint32_t g_best[3], g_rxcnt[3], g_rxpos[3], g_rxbad[3];
  uint32_t scig[FAST_MAX_READS][WG_CIG_OPS];   // simplified CIGAR of each read (S, H, =, X folded into M, neighbours merged): len << 4 | kind
  uint8_t n_scig[FAST_MAX_READS];
  uint8_t b4_order[FAST_MAX_READS];             // alignment filter: members of one end, longest first
  uint16_t b4_mask[FAST_MAX_READS];             // alignment filter: groups a read belongs to
  uint32_t any_complex;                         // some read has more than one aligned block or clips
  int32_t name_rank[FAST_MAX_READS];            // --max-reads: fgbio name rank of each kept read
};

__device__ __forceinline__ void defer(Shared& S) { S.defer = 1; }

__device__ __forceinline__ uint32_t int_tag_width(uint32_t v) { return v <= 255 ? 1 : 2; }   // depth <= 32767: c / C / S

// Oriented, mask-aware view of read `r` at consensus position p.  Returns the 4-bit code (15 = N, also
// for masked bases) and the quality the reference's SourceRead would hold there.
__device__ __forceinline__ void oriented(const Shared& S, const uint8_t* lb, const uint8_t* lq, const ReadInfo& R, uint32_t p,
                                         uint32_t min_bq, uint8_t* code, uint8_t* qual) {
  bool rev = (R.flags & bam::F_REVERSE) != 0;
  uint32_t idx = rev ? (uint32_t)R.l_seq - 1 - p : p;
  uint8_t c = lb[R.row + idx];
  uint8_t q = lq[R.row + idx];
  if (rev) c = bam::code_complement(c);
  if (p < R.trim_to && q < min_bq) { c = 15; q = FGX_MIN_PHRED; }
  *code = c;
  *qual = q;
}

// fgbio_read_name_rank (raw-bam/hash.rs:14-89): Murmur3_32, seed 42, over the UTF-16 code units of the read name, as i32
__device__ inline int32_t name_rank(const uint8_t* name, uint32_t len) {
  auto rotl = [](uint32_t v, int r) { return (v << r) | (v >> (32 - r)); };
  auto mixk = [&](uint32_t k) { k *= 0xcc9e2d51u; k = rotl(k, 15); return k * 0x1b873593u; };
  uint32_t h = 42;
  for (uint32_t i = 1; i < len; i += 2) { h ^= mixk((uint32_t)name[i - 1] | ((uint32_t)name[i] << 16)); h = rotl(h, 13) * 5 + 0xe6546b64u; }
  if (len & 1) h ^= mixk(name[len - 1]);
  h ^= 2 * len;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return (int32_t)h;
}

// ---- general CIGARs in the workgroup kernel ------------------------------------------------------------------------------------
// query offset of reference position x (1-based) in a read that starts at ref1, or -1 when x is not inside an aligned block or
// falls past the stored bases (ReadMateAndRefPosIterator, overlapping.rs:565-620)
__device__ __forceinline__ int32_t map_ref_to_query(const uint32_t* ops, uint32_t n, int64_t ref1, int64_t x, uint32_t l_seq) {
  int64_t ref = ref1, q = 0;
  for (uint32_t i = 0; i < WG_CIG_OPS; i++) {
    if (i >= n) break;
    const uint32_t t = ops[i] & 15;
    const int64_t len = ops[i] >> 4;
    if (t == 0 || t == 7 || t == 8) {
      if (x < ref + len) { if (x < ref) return -1; const int64_t qi = q + (x - ref); return qi < (int64_t)l_seq ? (int32_t)qi : -1; }
      ref += len; q += len;
    } else if (t == 1 || t == 4) q += len;
    else if (t == 2 || t == 3) ref += len;
  }
  return -1;
}
// SourceRead::simplified_cigar of read r: reversed for a reverse-strand read, truncated to its final length
// (create_source_read :1172-1176, truncate_simplified_cigar :1028-1062)
__device__ inline uint32_t oriented_truncated_cigar(const Shared& S, uint32_t r, uint32_t* out) {
  const ReadInfo& R = S.ri[r];
  const uint32_t n = S.n_scig[r];
  const bool rev = (R.flags & bam::F_REVERSE) != 0;
  uint32_t remaining = R.final_len, m = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (remaining == 0) break;
    const uint32_t v = S.scig[r][rev ? n - 1 - i : i], k = v & 15, len = v >> 4;
    if (k == 0 || k == 1) { const uint32_t take = len < remaining ? len : remaining; out[m++] = (take << 4) | k; remaining -= take; }
    else out[m++] = v;
  }
  return m;
}
__device__ inline bool cigar_is_prefix(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb) {   // clipper.rs:1212-1241
  if (na > nb) return false;
  for (uint32_t i = 0; i < na; i++) {
    if ((a[i] & 15) != (b[i] & 15)) return false;
    if (i + 1 == na) { if ((a[i] >> 4) > (b[i] >> 4)) return false; }
    else if ((a[i] >> 4) != (b[i] >> 4)) return false;
  }
  return true;
}
__device__ inline int cigar_cmp(const uint32_t* a, uint32_t na, const uint32_t* b, uint32_t nb) {   // vanilla_caller.rs:96-113: length, then op kind, then op count
  const uint32_t n = na < nb ? na : nb;
  for (uint32_t i = 0; i < n; i++) {
    if ((a[i] >> 4) != (b[i] >> 4)) return (a[i] >> 4) < (b[i] >> 4) ? -1 : 1;
    if ((a[i] & 15) != (b[i] & 15)) return (a[i] & 15) < (b[i] & 15) ? -1 : 1;
  }
  return na == nb ? 0 : (na < nb ? -1 : 1);
}
// filter_source_reads_by_alignment (vanilla_caller.rs:1242-1296) for the kept reads of end e, by one thread: stable sort by
// length (longest first), greedy multi-membership prefix groups, the largest group wins (ties: the smaller CIGAR, then the
// later group); everyone outside it is marked `minority`.
__device__ inline void alignment_filter(Shared& S, uint32_t e, uint32_t n) {
  uint32_t m = 0;
  bool single_block = true;
  for (uint32_t i = 0; i < n; i++)
    if (S.ri[i].end == e && !S.ri[i].zero_len) { S.b4_order[m++] = (uint8_t)i; if (S.n_scig[i] != 1) single_block = false; }
  if (m < 2 || single_block) return;        // (M, len) CIGARs are all prefixes of the longest one: a single group, nothing dropped
  for (uint32_t k = 1; k < m; k++) {        // stable insertion sort, longest first
    const uint8_t v = S.b4_order[k];
    const uint32_t lv = S.ri[v].final_len;
    uint32_t j = k;
    while (j > 0 && S.ri[S.b4_order[j - 1]].final_len < lv) { S.b4_order[j] = S.b4_order[j - 1]; j--; }
    S.b4_order[j] = v;
  }
  constexpr uint32_t MAX_GROUPS = 16;
  uint32_t ng = 0;
  uint8_t founder[MAX_GROUPS];
  uint32_t count[MAX_GROUPS];
  uint32_t ca[WG_CIG_OPS], cb[WG_CIG_OPS];
  for (uint32_t k = 0; k < m; k++) {
    const uint32_t r = S.b4_order[k];
    const uint32_t na = oriented_truncated_cigar(S, r, ca);
    uint32_t mask = 0;
    for (uint32_t g = 0; g < ng; g++) {
      const uint32_t nb = oriented_truncated_cigar(S, founder[g], cb);
      if (cigar_is_prefix(ca, na, cb, nb)) { count[g]++; mask |= 1u << g; }      // no break: a read joins every group it is a prefix of
    }
    if (!mask) {
      if (ng == MAX_GROUPS) { defer(S); return; }
      founder[ng] = (uint8_t)r; count[ng] = 1; mask = 1u << ng; ng++;
    }
    S.b4_mask[r] = (uint16_t)mask;
  }
  uint32_t best = 0;                         // Iterator::max_by keeps the LAST maximal element
  for (uint32_t i = 1; i < ng; i++) {
    int c;
    if (count[best] != count[i]) c = count[best] < count[i] ? -1 : 1;
    else {
      const uint32_t na = oriented_truncated_cigar(S, founder[i], ca), nb = oriented_truncated_cigar(S, founder[best], cb);
      c = cigar_cmp(ca, na, cb, nb);
    }
    if (c <= 0) best = i;
  }
  for (uint32_t k = 0; k < m; k++) { const uint32_t r = S.b4_order[k]; if (!((S.b4_mask[r] >> best) & 1)) S.ri[r].minority = 1; }
}


#include "chain_observe.inc"
// mask of the lanes l with p0 + l < limit (both scalar), on the scalar unit (k_simplex_wave2's lanes_below, which is defined after this kernel)
__device__ __forceinline__ unsigned long long fw_lanes_below(uint32_t limit, uint32_t p0) {
#if defined(FGX_WAVEMU)
  const uint32_t span = (limit > p0 ? limit : p0) - p0;
  return span >= 64u ? ~0ull : ((1ull << span) - 1ull);
#else
  unsigned long long m;
  uint32_t span;
  asm("s_max_u32 %1, %2, %3\n\ts_sub_u32 %1, %1, %3\n\ts_bfm_b64 %0, %1, 0\n\ts_cmp_ge_u32 %1, 64\n\ts_cselect_b64 %0, -1, %0"
      : "=&s"(m), "=&s"(span) : "s"(limit), "s"(p0) : "scc");
  return m;
#endif
}

// ---- wavefront helpers, unaligned LDS words, the aux walk (shared by every family kernel) ----------------------------
constexpr int WAVES_PER_BLOCK = 4;
#ifndef FGX_W2_WPB_DEFAULT
#define FGX_W2_WPB_DEFAULT 3
#endif

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }   // v is the same in every lane: say so
__device__ __forceinline__ uint32_t rlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
// Wavefront reductions of a 32-bit value through the data-parallel primitives of the vector ALU: quad swap, quad-pair swap, half-row mirror,
// row mirror (every lane of a row of 16 then holds the row's result), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 — six
// instructions, each with its operation folded in, the total in lane 63, returned as a scalar.  (A butterfly of __shfl_xor is six LDS
// permutes + six operations + their addresses.)  `idn` = the operation's identity: what a lane outside a step's row mask contributes.
template <int CTRL, int ROWS> __device__ __forceinline__ uint32_t wave_dpp(uint32_t idn, uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)idn, (int)v, CTRL, ROWS, 0xF, false); }
#define FGX_WAVE_REDUCE(v, idn, OP) do { \
    v = OP(v, wave_dpp<0xB1, 0xF>(idn, v)); v = OP(v, wave_dpp<0x4E, 0xF>(idn, v)); v = OP(v, wave_dpp<0x141, 0xF>(idn, v)); v = OP(v, wave_dpp<0x140, 0xF>(idn, v)); \
    v = OP(v, wave_dpp<0x142, 0xA>(idn, v)); v = OP(v, wave_dpp<0x143, 0xC>(idn, v)); } while (0)
__device__ __forceinline__ uint32_t wr_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wr_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t wr_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t wave_max(uint32_t v) { FGX_WAVE_REDUCE(v, 0u, wr_max); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_min(uint32_t v) { FGX_WAVE_REDUCE(v, 0xFFFFFFFFu, wr_min); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { FGX_WAVE_REDUCE(v, 0u, wr_add); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ unsigned long long wave_max64(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }
__device__ __forceinline__ unsigned long long wave_min64(unsigned long long v) { for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; } return v; }
__device__ __forceinline__ uint8_t comp_code(uint8_t c) { return (uint8_t)((0xF7B3D591E6A2C480ULL >> (4 * (c & 15))) & 15); }

// unaligned LDS reads built from aligned dwords (v_alignbyte_b32)
__device__ __forceinline__ uint32_t ldsw(const uint8_t* W, uint32_t o) { return *(const uint32_t*)(W + o); }
#ifndef FGX_LDS_UNALIGNED
#define FGX_LDS_UNALIGNED 1   /* gfx950 LDS takes unaligned ds_read_b32 / b64: one instruction instead of two or three aligned reads + alignbyte */
#endif
__device__ __forceinline__ uint32_t ld32u(const uint8_t* W, uint32_t o) {
#if FGX_LDS_UNALIGNED
  uint32_t v;
  __builtin_memcpy(&v, W + o, 4);
  return v;
#else
  uint32_t a = o & ~3u;
  return __builtin_amdgcn_alignbyte(ldsw(W, a + 4), ldsw(W, a), o & 3);
#endif
}
__device__ __forceinline__ unsigned long long ld64u(const uint8_t* W, uint32_t o) {
#if FGX_LDS_UNALIGNED
  unsigned long long v;
  __builtin_memcpy(&v, W + o, 8);
  return v;
#else
  uint32_t a = o & ~3u, sh = o & 3;
  uint32_t w0 = ldsw(W, a), w1 = ldsw(W, a + 4), w2 = ldsw(W, a + 8);
  return (unsigned long long)__builtin_amdgcn_alignbyte(w1, w0, sh) | ((unsigned long long)__builtin_amdgcn_alignbyte(w2, w1, sh) << 32);
#endif
}
// offset of the first NUL in W[o, o+n), 8 bytes per step; -1 if none
__device__ __forceinline__ int find_nul64(const uint8_t* W, uint32_t o, uint32_t n) {
  for (uint32_t i = 0; i < n; i += 8) {
    unsigned long long v = ld64u(W, o + i);
    unsigned long long t = (v - 0x0101010101010101ULL) & ~v & 0x8080808080808080ULL;
    if (t) { uint32_t k = i + ((uint32_t)__builtin_ctzll(t) >> 3); return k < n ? (int)k : -1; }
  }
  return -1;
}

// Aux walk of one record per lane, in LDS (tags.rs:13-34): first occurrence of MC / <tag> / RX / <cell tag>.
// The walk keeps its state in integer lanes (bit 0 MC, 1 <tag>, 2 RX, 3 <cell tag>) and picks with selects: boolean state
// would live in scalar lane masks, and every `if` on it costs scalar mask instructions for all 64 records.
struct AuxTags {
  uint32_t got, oddw;                          // keys found with a Z value ; <tag> / RX / <cell tag> values longer than 255 bytes
  uint32_t pk_mc, pk_mi, pk_rx, pk_cb;         // value offset in LDS | value length << 16
};
// cls: aux value type -> 1 / 2 / 4 (fixed size), 8 (Z), 16 (H), 32 (B), 0 (unknown).  The table lies in device memory, built at compile time
// (round 6): a wavefront copies it with one load and one LDS store.  Until round 5 every workgroup COMPUTED its 256 entries (a switch per
// entry: ~190 vector instructions per wavefront — a quarter of k_split_parse's instructions per family, whose wavefronts live for one round
// of 64 records).
struct TagClassTable {
  uint8_t v[256];
  constexpr TagClassTable() : v{} {
    v['A'] = 1; v['c'] = 1; v['C'] = 1; v['s'] = 2; v['S'] = 2; v['i'] = 4; v['I'] = 4; v['f'] = 4; v['Z'] = 8; v['H'] = 16; v['B'] = 32;
  }
};
__device__ const TagClassTable g_tag_classes = TagClassTable();
static_assert(TagClassTable().v['i'] == 4 && TagClassTable().v['Z'] == 8 && TagClassTable().v['x'] == 0, "tag classes");
__device__ __forceinline__ void fill_tag_classes(uint8_t* cls /* 256 bytes of LDS, 4-byte aligned */) {
  if (threadIdx.x < 64u) ((uint32_t*)cls)[threadIdx.x] = ((const uint32_t*)g_tag_classes.v)[threadIdx.x];
}
template <class ParamsT>
__device__ __forceinline__ void aux_walk(const uint8_t* W, const uint8_t* cls_of, uint32_t a0, uint32_t an, const ParamsT& P, AuxTags& A) {
  uint32_t q = 0, seen = 0, got = 0, oddw = 0;
  uint32_t pk_mc = 0, pk_mi = 0, pk_rx = 0, pk_cb = 0;
  const uint32_t key_mi = (uint32_t)(uint8_t)P.tag0 | ((uint32_t)(uint8_t)P.tag1 << 8);
  const uint32_t key_cb = P.cell0 ? ((uint32_t)(uint8_t)P.cell0 | ((uint32_t)(uint8_t)P.cell1 << 8)) : 0xFFFFFFFFu;   // (no key is > 0xFFFF)
  while (q + 3 <= an) {
    const uint32_t hd = ld32u(W, a0 + q);
    const uint32_t key = hd & 0xFFFF, cls = cls_of[(hd >> 16) & 0xFF];
    const uint32_t rem = an - (q + 3);
    uint32_t size = cls & 7, zend = 0;
    uint32_t stop = cls == 0 ? 1u : 0u;                    // a key match at an entry of unknown size still counts as seen
    if (cls & 24) {
      const int z = find_nul64(W, a0 + q + 3, rem);
      if (z < 0) break;                                    // unterminated: nothing here or after it is reachable
      zend = (uint32_t)z; size = zend + 1;
    }
    if (cls & 32) {
      if (rem < 5) break;
      const uint32_t es = cls_of[hd >> 24] & 7;
      const unsigned long long sz = 5ull + (unsigned long long)ld32u(W, a0 + q + 4) * (unsigned long long)es;
      if (sz > 0xFFFFFFFFull) break;
      size = (uint32_t)sz;
      stop = es == 0 ? 1u : 0u;
    }
    const uint32_t kb = (key == ('M' | ('C' << 8)) ? 1u : 0u) | (key == key_mi ? 2u : 0u) | (key == ('R' | ('X' << 8)) ? 4u : 0u) | (key == key_cb ? 8u : 0u);
    const uint32_t fresh = kb & ~seen;                     // first occurrence of its key
    seen |= kb;
    const uint32_t zb = cls == 8 ? fresh : 0u;             // ... with a Z value
    const uint32_t rec = zend > 255 ? (zb & 1u) : zb;      // values longer than 255 bytes: only MC may be (it is then not `<n>M`)
    oddw |= zb & ~rec;
    got |= rec;
    const uint32_t pk = (a0 + q + 3) | ((zend < 0xFFFFu ? zend : 0xFFFFu) << 16);
    pk_mc = (rec & 1u) ? pk : pk_mc; pk_mi = (rec & 2u) ? pk : pk_mi; pk_rx = (rec & 4u) ? pk : pk_rx; pk_cb = (rec & 8u) ? pk : pk_cb;
    const unsigned long long nq = (unsigned long long)q + 3 + size;
    if (stop || nq > an) break;
    q = (uint32_t)nq;
  }
  A.got = got; A.oddw = oddw; A.pk_mc = pk_mc; A.pk_mi = pk_mi; A.pk_rx = pk_rx; A.pk_cb = pk_cb;
}

#ifndef FGX_PHASE_TIMING
#define FGX_PHASE_TIMING 0   /* 1: per-phase s_memtime deltas of k_family_wave into g_phase (profiling builds only) */
#endif
#if FGX_PHASE_TIMING
__device__ unsigned long long g_phase[64 * 16];
#define PH(i) { unsigned long long _n = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&g_phase[(blockIdx.x & 63) * 16 + (i)], _n - _t); _t = _n; }
#else
#define PH(i)
#endif
#if FGX_PHASE_TIMING
#define PHB(i) { __syncthreads(); unsigned long long _n = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_phase[(blockIdx.x & 63) * 16 + (i)], _n - _tb); _tb = _n; }
#else
#define PHB(i)
#endif
__global__ __launch_bounds__(NT) void k_family(FastParams P) {
  FGX_DYN_LDS(dyn);
  __shared__ Shared S;
  __shared__ __align__(16) double sPairB[94][2];   // {correct[q], error_per_alt[q]}: one LDS read per observation instead of two global ones
  __shared__ __align__(16) uint8_t sTagCls[256];   // aux value type classes (aux_walk)
  const uint32_t tid = threadIdx.x;
  if (tid < 94) { sPairB[tid][0] = P.T->t.correct[tid]; sPairB[tid][1] = P.T->t.error_per_alt[tid]; }
  fill_tag_classes(sTagCls);
  const uint32_t g = P.group_list ? P.group_list[blockIdx.x] : P.g0 + blockIdx.x;
  const uint32_t r0 = P.grp_first[g], r1 = P.grp_first[g + 1];
  const uint32_t n = r1 - r0;
  const uint32_t slot0 = 3 * g;

  if (tid < FGX_STATS_LEN) S.stats[tid] = 0;
  if (tid == 0) { S.defer = 0; S.n_ends = 0; S.rx_bad = 0; S.raw_lo = ~0ull; S.raw_hi = 0ull; S.any_complex = 0; }
  if (tid < 3) { P.ends[slot0 + tid].valid = 0; P.rec_sizes[slot0 + tid] = 0; }
  __syncthreads();

  // ---- group-level short circuit (simplex.rs:673-683) ---------------------------------------------
  if (n < P.min_reads) {
    if (tid == 0) {
      unsigned long long* st = P.stats + (size_t)(blockIdx.x & (STAT_SLOTS - 1)) * 32;
      atomicAdd(&st[0], (unsigned long long)n);
      atomicAdd(&st[2], (unsigned long long)n);
      atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_READS], (unsigned long long)n);
    }
    return;
  }
  if (n > FAST_MAX_READS) {
    if (tid == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    return;
  }

#if FGX_PHASE_TIMING
  unsigned long long _tb = __builtin_amdgcn_s_memtime();
#endif
  // ---- 0. the family's raw records into LDS with coalesced 16-byte loads: everything below (tag walk, name compares,
  // unpacking of bases and qualities) reads LDS instead of walking HBM byte by byte ------------------------------------
  {
    unsigned long long off = tid < n ? P.rec_off[r0 + tid] : ~0ull, end = tid < n ? P.rec_off[r0 + tid] + P.rec_len[r0 + tid] : 0ull;
    for (int o = 32; o > 0; o >>= 1) { unsigned long long a = __shfl_xor(off, o), b = __shfl_xor(end, o); off = a < off ? a : off; end = b > end ? b : end; }
    if ((tid & 63) == 0) { atomicMin(&S.raw_lo, off); atomicMax(&S.raw_hi, end); }
  }
  __syncthreads();
  const unsigned long long base16 = S.raw_lo & ~15ull;
  const uint32_t raw_bytes = (uint32_t)(((S.raw_hi - base16) + 15 + 16) & ~15ull);      // + slack for multi-byte reads past the last record
  if ((S.raw_hi - base16) + 32 > (unsigned long long)P.lds_tile_bytes || S.raw_hi > P.blob_len) {   // not even the raw span fits (or a record ends outside the blob)
    if (tid == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    return;
  }
  for (uint32_t i = tid * 16; i < raw_bytes - 16; i += NT * 16) *(uint4*)(dyn + i) = *(const uint4*)(P.blob + base16 + i);
  __syncthreads();
  const uint8_t* const blobL = dyn - base16;          // blobL + <offset in the blob> addresses the LDS copy

  PHB(9)
  // ---- 1. parse one record per lane -----------------------------------------------------------------
  if (tid < n) {
    ReadInfo R;
    memset(&R, 0, sizeof(R));
    R.mate = -1;
    R.end = 255;
    uint64_t off = P.rec_off[r0 + tid];
    uint32_t len = P.rec_len[r0 + tid];
    const uint8_t* p = blobL + off;
    R.goff = off;
    bool bad = len < 32;
    if (!bad) {
      uint32_t l_name = p[8], n_cig = rd16(p + 12), l_seq = rd32(p + 16);
      uint16_t flag = rd16(p + 14);
      uint64_t seq_off = 32ull + l_name + 4ull * n_cig;
      uint64_t qual_off = seq_off + ((uint64_t)l_seq + 1) / 2;
      uint64_t aux_off = qual_off + l_seq;
      if (aux_off > len || l_seq > 65535 || l_name == 0) bad = true;
      else {
        R.flags = flag; R.l_seq = (uint16_t)l_seq; R.seq_off = (uint16_t)seq_off; R.name_len = (uint16_t)(l_name - 1);
        R.pos = (int32_t)rd32(p + 4); R.ref_id = (int32_t)rd32(p);
        R.excluded = (flag & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) ? 1 : 0;
        // CIGAR: up to WG_CIG_OPS ops of any kind (clips, indels, skips); consensus is called in query space, the CIGAR matters
        // for the mate clip, the overlap correction and the alignment filter
        uint32_t ops[WG_CIG_OPS];
        uint32_t n_ops = 0;
        for (uint32_t i = 0; i < WG_CIG_OPS; i++) ops[i] = 0;
        if (!R.excluded) {
          if ((flag & bam::F_UNMAPPED) || n_cig == 0 || n_cig > WG_CIG_OPS || l_seq == 0 || R.pos < 0) bad = true;
          else {
            uint64_t qsum = 0;
            uint32_t ns = 0;
            for (uint32_t i = 0; i < n_cig; i++) {
              const uint32_t o = rd32(p + 32 + l_name + 4 * i), t = o & 15;
              ops[i] = o;
              if (t > 8 || (o >> 4) > 0x3FFFFFu) bad = true;
              if (t == 0 || t == 1 || t == 4 || t == 7 || t == 8) qsum += o >> 4;
              const uint32_t k = (t == 4 || t == 5 || t == 7 || t == 8) ? 0u : t;      // simplify_cigar (clipper.rs:1183-1210)
              if (ns > 0 && (S.scig[tid][ns - 1] & 15) == k) S.scig[tid][ns - 1] += (o >> 4) << 4;
              else S.scig[tid][ns++] = ((o >> 4) << 4) | k;
            }
            n_ops = n_cig;
            S.n_scig[tid] = (uint8_t)ns;
            if (qsum != l_seq) bad = true;          // SEQ and CIGAR disagree: the general path decides
            if (n_cig != 1 || (ops[0] & 15) == 1) S.any_complex = 1;
          }
        }
        // aux walk: first occurrence of MC / <tag> / RX / <cell tag>; malformed aux stops the walk (tags.rs:13-34)
        const uint32_t lo_r = (uint32_t)(off - base16);            // LDS offset of this record
        if ((unsigned long long)lo_r + len > 65535ull) bad = true;  // (the walk packs LDS offsets into 16 bits: tiles beyond 64 KB defer)
        int mc_off = -1; uint32_t mc_len = 0;
        {
          AuxTags ax;
          aux_walk(dyn, sTagCls, lo_r + (uint32_t)aux_off, (uint32_t)(len - aux_off), P, ax);
          if (ax.oddw) bad = true;                                  // <tag> / RX / <cell tag> value longer than 255 bytes
          if (ax.got & 1u) { mc_off = (int)((ax.pk_mc & 0xFFFF) - lo_r); mc_len = ax.pk_mc >> 16; }
          const uint32_t mi_o = (ax.pk_mi & 0xFFFF) - lo_r, rx_o = (ax.pk_rx & 0xFFFF) - lo_r, cb_o = (ax.pk_cb & 0xFFFF) - lo_r;
          if (ax.got & 2u) { R.has_mi = 1; R.mi_off = (uint16_t)mi_o; R.mi_len = (uint8_t)(ax.pk_mi >> 16); }
          if (ax.got & 4u) { R.has_rx = 1; R.rx_off = (uint16_t)rx_o; R.rx_len = (uint8_t)(ax.pk_rx >> 16); }
          if (ax.got & 8u) { R.has_cb = 1; R.cb_off = (uint16_t)cb_o; R.cb_len = (uint8_t)(ax.pk_cb >> 16); }
        }
        if (!bad && !R.excluded) {
          // mate-overlap clip (raw-bam/overlap.rs:181-207)
          uint32_t mops[WG_MC_OPS];
          bool overflow = false;
          bam::Rec v{p, len};
          uint64_t clip = bam::mate_clip(v, ops, n_ops, mc_off >= 0 ? p + mc_off : nullptr, mc_len, mops, WG_MC_OPS, &overflow);
          if (overflow) bad = true;
          R.clip = (uint16_t)(clip > 65535 ? 65535 : clip);
          // name hash for mate pairing (a filter: candidates are compared word by word below): whole 8-byte steps, then one step
          // over the LAST 8 bytes (it may overlap)
          {
            const uint32_t nl = R.name_len;
            uint32_t h = nl, i = 0;
            for (; i + 8 <= nl; i += 8) { h = __builtin_rotateleft32(h, 5) ^ ld32u(dyn, lo_r + 32 + i); h = __builtin_rotateleft32(h, 11) + ld32u(dyn, lo_r + 36 + i); }
            if (i < nl) {
              uint32_t w0, w1;
              if (nl >= 8) { w0 = ld32u(dyn, lo_r + 24 + nl); w1 = ld32u(dyn, lo_r + 28 + nl); }
              else { const unsigned long long v = ld64u(dyn, lo_r + 32) & ((1ULL << (8 * nl)) - 1ULL); w0 = (uint32_t)v; w1 = (uint32_t)(v >> 32); }
              h = __builtin_rotateleft32(h, 5) ^ w0; h = __builtin_rotateleft32(h, 11) + w1;
            }
            R.name_hash = h ^ (h >> 15) ^ (h << 7);
          }
        }
      }
    }
    if (bad) defer(S);
    S.ri[tid] = R;
  }
  __syncthreads();
  // UMI = MI tag of the first record (vanilla_caller.rs:1897-1908); missing → fatal in the reference
  if (tid == 0 && !S.defer) {
    if (!S.ri[0].has_mi) defer(S);
    else if ((uint32_t)P.prefix_len + 1 + S.ri[0].mi_len >= 255) defer(S);   // read name too long → fatal
    // LDS tile rows
    uint32_t row = 0;
    for (uint32_t i = 0; i < n; i++) { S.ri[i].row = row; if (!S.ri[i].excluded) row += ((uint32_t)S.ri[i].l_seq + 3) & ~3u; }
    S.tile_bytes = row;
    if ((unsigned long long)raw_bytes + 2ull * row > P.lds_tile_bytes && !S.defer) S.defer = P.retry ? 2 : 1;
  }
  __syncthreads();
  if (S.defer) {
    if (tid == 0) {
      if (S.defer == 2) { uint32_t k = atomicAdd(P.n_retry, 1u); P.retry[k] = g; }
      else { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    }
    return;
  }
  uint8_t* lb = dyn + raw_bytes;
  uint8_t* lq = lb + S.tile_bytes;

  PHB(10)
  // ---- 2. stage bases (unpacked 4-bit codes) and quals into LDS -------------------------------------
  for (uint32_t r = tid >> 6; r < n; r += NT / 64) {          // one wavefront per record: NT / 64 records in flight
    const ReadInfo& R = S.ri[r];
    if (R.excluded) continue;
    const uint8_t* p = blobL + R.goff;
    const uint8_t* sq = p + R.seq_off;
    const uint8_t* ql = sq + ((uint32_t)R.l_seq + 1) / 2;
    for (uint32_t i = tid & 63; i < R.l_seq; i += 64) {
      uint8_t b = sq[i >> 1];
      lb[R.row + i] = (i & 1) ? (b & 0xF) : (b >> 4);
      lq[R.row + i] = ql[i];
    }
  }
  // absent qualities (every byte 0xFF) are a fatal input error in the reference (:1119-1124) → general path
  __syncthreads();
  if (tid < n && !S.ri[tid].excluded) {
    // serial re-check only when the first qual is 0xFF (cheap early-out; the all-0xFF case is a fatal error)
    const ReadInfo& R = S.ri[tid];
    if (lq[R.row] == 0xFF) {
      bool all = true;
      for (uint32_t i = 0; i < R.l_seq; i++) if (lq[R.row + i] != 0xFF) { all = false; break; }
      if (all) defer(S);
    }
  }
  __syncthreads();
  if (S.defer) {
    if (tid == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    return;
  }

  PHB(11)
  // ---- 3. overlapping-bases pre-correction (overlapping.rs:236-336, 627-684) ------------------------
  if (P.overlap) {
    // pair map semantics: for each name the LAST primary record with FIRST set and the LAST with (not FIRST and) LAST set.
    // One wavefront per R1 candidate, the lanes test 64 records at a time (n <= 128: two steps).
    {
      const uint32_t wave_p = tid >> 6, lane_p = tid & 63;
      for (uint32_t a = wave_p; a < n; a += NT / 64) {
        const ReadInfo& R = S.ri[a];
        if (R.excluded || !(R.flags & bam::F_FIRST)) continue;
        const uint32_t rh = R.name_hash, rl_ = R.name_len;
        uint64_t later_r1[2], r2s[2];
        for (int c = 0; c < 2; c++) {
          const uint32_t u = lane_p + 64u * c;
          bool is_later_r1 = false, is_r2 = false;
          if (u < n) {
            const ReadInfo& O = S.ri[u];
            if (!O.excluded && O.name_hash == rh && O.name_len == rl_) {
              const bool first = (O.flags & bam::F_FIRST) != 0;
              const bool cand = first ? (u > a) : ((O.flags & bam::F_LAST) != 0);
              if (cand) {
                const uint32_t on = (uint32_t)(O.goff - base16) + 32, nn = (uint32_t)(R.goff - base16) + 32;
                unsigned long long diff = 0;                  // whole 8-byte steps + the last 8 bytes (names of one length)
                uint32_t i = 0;
                for (; i + 8 <= rl_; i += 8) diff |= ld64u(dyn, on + i) ^ ld64u(dyn, nn + i);
                if (i < rl_) {
                  if (rl_ >= 8) diff |= ld64u(dyn, on + rl_ - 8) ^ ld64u(dyn, nn + rl_ - 8);
                  else diff |= (ld64u(dyn, on) ^ ld64u(dyn, nn)) & ((1ULL << (8 * rl_)) - 1ULL);
                }
                const bool same = diff == 0;
                is_later_r1 = same && first; is_r2 = same && !first;
              }
            }
          }
          later_r1[c] = __ballot(is_later_r1); r2s[c] = __ballot(is_r2);
        }
        if (lane_p == 0 && !(later_r1[0] | later_r1[1])) {
          int m = -1;
          if (r2s[1]) m = 64 + (63 - __clzll((long long)r2s[1]));
          else if (r2s[0]) m = 63 - __clzll((long long)r2s[0]);
          S.ri[a].mate = (int16_t)m;
        }
      }
    }
    __syncthreads();
    uint32_t ov_bases = 0, ov_agree = 0, ov_dis = 0, ov_corr = 0;
    const uint32_t wave = tid >> 6, lane = tid & 63;
    for (uint32_t a = wave; a < n; a += NT / 64) {     // pairs are disjoint, so their order does not matter
      const ReadInfo& A = S.ri[a];
      if (A.mate < 0) continue;
      const ReadInfo& B = S.ri[A.mate];
      if (A.ref_id != B.ref_id) continue;
      // both CIGARs out of the LDS copy of the records; positions shared by two aligned blocks are corrected (overlapping.rs:236-336)
      uint32_t na = 1, nb = 1;
      uint32_t oa[WG_CIG_OPS], ob[WG_CIG_OPS];
      int32_t rla = A.l_seq, rlb = B.l_seq;
      oa[0] = (uint32_t)A.l_seq << 4; ob[0] = (uint32_t)B.l_seq << 4;     // family of single-block reads: <l_seq>M, nothing to look up
      if (S.any_complex) {
        const uint8_t* pa = blobL + A.goff;
        const uint8_t* pb = blobL + B.goff;
        na = rd16(pa + 12); nb = rd16(pb + 12);
        for (uint32_t i = 0; i < WG_CIG_OPS; i++) {
          oa[i] = i < na ? rd32(pa + 32 + A.name_len + 1 + 4 * i) : 0u;
          ob[i] = i < nb ? rd32(pb + 32 + B.name_len + 1 + 4 * i) : 0u;
        }
        rla = bam::ref_len_checked0(oa, na); rlb = bam::ref_len_checked0(ob, nb);
      }
      if (rla == 0 || rlb == 0) continue;
      const int64_t s1 = (int64_t)A.pos + 1, e1 = (int64_t)A.pos + rla, s2 = (int64_t)B.pos + 1, e2 = (int64_t)B.pos + rlb;
      const int64_t lo = s1 > s2 ? s1 : s2, hi = e1 < e2 ? e1 : e2;
      const bool plain = !S.any_complex;                     // both reads are one aligned block: query offset = x - start
      for (int64_t x = lo + lane; x <= hi; x += 64) {
        int32_t i1, i2;
        if (plain) { i1 = (int32_t)(x - s1); i2 = (int32_t)(x - s2); }
        else { i1 = map_ref_to_query(oa, na, s1, x, A.l_seq); i2 = map_ref_to_query(ob, nb, s2, x, B.l_seq); }
        if (i1 < 0 || i2 < 0) continue;
        const uint32_t ia = A.row + (uint32_t)i1, ib = B.row + (uint32_t)i2;
        uint8_t c1 = lb[ia], c2 = lb[ib];
        if (c1 == 15 || c2 == 15) continue;
        ov_bases++;
        uint8_t qa = lq[ia], qb = lq[ib];
        if (c1 == c2) {
          ov_agree++;
          uint32_t s = (uint32_t)qa + qb;
          uint8_t nq = (uint8_t)(s < 93 ? s : 93);
          lq[ia] = nq; lq[ib] = nq;
          if (nq != qa || nq != qb) ov_corr++;
        } else {
          ov_dis++;
          uint8_t cb, cq;
          if (qa == qb) { cb = 15; cq = FGX_MIN_PHRED; }
          else if (qa > qb) { cb = c1; cq = (uint8_t)(qa - qb); if (cq < FGX_MIN_PHRED) cq = FGX_MIN_PHRED; }
          else { cb = c2; cq = (uint8_t)(qb - qa); if (cq < FGX_MIN_PHRED) cq = FGX_MIN_PHRED; }
          lb[ia] = cb; lb[ib] = cb; lq[ia] = cq; lq[ib] = cq;
          ov_corr += 2;
        }
      }
    }
    if (ov_bases) atomicAdd(&S.stats[24], ov_bases);
    if (ov_agree) atomicAdd(&S.stats[25], ov_agree);
    if (ov_dis) atomicAdd(&S.stats[26], ov_dis);
    if (ov_corr) atomicAdd(&S.stats[27], ov_corr);
    __syncthreads();
  }

  PHB(12)
  // ---- 4. per-read source-read geometry (vanilla_caller.rs:1129-1160) ---------------------------------
  if (tid < n && !S.ri[tid].excluded) {
    ReadInfo& R = S.ri[tid];
    bool rev = (R.flags & bam::F_REVERSE) != 0;
    uint32_t L = R.l_seq;
    uint32_t trim_to = L;
    if (P.trim) {   // find_quality_trim_point on the oriented qualities (:992-1016)
      uint32_t tq = P.min_input_bq;
      if (tq < 1 || L == 0) trim_to = 0;
      else {
        int32_t score = 0, max_score = 0;
        uint32_t point = L;
        for (uint32_t i = L; i-- > 0;) {
          uint32_t idx = rev ? L - 1 - i : i;
          score += (int32_t)tq - (int32_t)lq[R.row + idx];
          if (score < 0) break;
          if (score > max_score) { max_score = score; point = i; }
        }
        trim_to = point;
      }
    }
    R.trim_to = (uint16_t)trim_to;
    uint32_t clip_pos = L > R.clip ? L - R.clip : 0;
    uint32_t fl = clip_pos < trim_to ? clip_pos : trim_to;
    while (fl > 0) {
      uint8_t c, q;
      oriented(S, lb, lq, R, fl - 1, P.min_input_bq, &c, &q);
      if (c != 15) break;
      fl--;
    }
    R.final_len = (uint16_t)fl;
    R.zero_len = fl == 0;
  }
  __syncthreads();

  PHB(13)
  // ---- 5. family gates (process_group :1329-1422, process_subgroup :1454-1646) -------------------------
  // One thread per record (n <= FAST_MAX_READS = 128: two wavefronts); counts are barrier-counts, the member lists are
  // built with ballots, and thread 0 only takes the scalar decisions.
  {
    if (tid < 3) { S.g_best[tid] = 0; S.g_rxcnt[tid] = 0; S.g_rxpos[tid] = 0xFFFFFFFFu; S.g_rxbad[tid] = 0; S.g_wcnt[0][tid] = 0; S.g_wcnt[1][tid] = 0; }
    const bool mine = tid < n;
    const bool exc = mine && S.ri[tid].excluded;
    const uint32_t n_sec = (uint32_t)__syncthreads_count(exc);
    const uint32_t n_reads = n - n_sec;
    bool go = n_reads > 0 && n_reads >= P.min_reads;
    uint32_t my_end = 255;
    bool my_zero = false;
    if (go && mine && !exc) {
      ReadInfo& R = S.ri[tid];
      if (!(R.flags & bam::F_PAIRED)) my_end = 0;
      else if (R.flags & bam::F_FIRST) my_end = 1;
      else if (R.flags & bam::F_LAST) my_end = 2;
      R.end = (uint8_t)my_end;
      my_zero = R.zero_len != 0;
    }
    uint32_t cnt[3], zero[3], rem[3], first[3] = {0, 0, 0};
    bool ok[3] = {false, false, false};
    for (int e = 0; e < 3; e++) {
      cnt[e] = (uint32_t)__syncthreads_count(my_end == (uint32_t)e);
      zero[e] = (uint32_t)__syncthreads_count(my_end == (uint32_t)e && my_zero);
      rem[e] = cnt[e] - zero[e];
    }
    // alignment filter (filter_source_reads_by_alignment :1242-1296): only when some read of the family has clips or indels —
    // reads with one aligned block all fall into one prefix-compatible group
    uint32_t minor[3] = {0, 0, 0};
    bool my_minor = false;
    if (go && S.any_complex) {
      if (tid == 0)
        for (uint32_t e = 0; e < 3; e++)
          if (cnt[e] >= P.min_reads && rem[e] >= P.min_reads && rem[e] >= 2) alignment_filter(S, e, n);
      __syncthreads();
      my_minor = my_end < 3 && !my_zero && S.ri[tid].minority != 0;
      for (int e = 0; e < 3; e++) minor[e] = (uint32_t)__syncthreads_count(my_minor && my_end == (uint32_t)e);
    }
    bool want_defer = false;
    uint32_t n_members = 0;
    uint32_t kept[3] = {0, 0, 0};               // reads of the end after the zero-length drop and the alignment filter
    uint32_t down[3] = {0, 0, 0};               // ... of which --max-reads drops
    bool bites = false;
    for (int e = 0; e < 3; e++) {
      if (!go || cnt[e] == 0 || cnt[e] < P.min_reads || rem[e] < P.min_reads) continue;
      kept[e] = rem[e] - minor[e];
      if (kept[e] >= P.min_reads && P.max_reads >= 0 && (int64_t)kept[e] > P.max_reads) bites = true;
    }
    bool my_down = false;
    if (bites) {   // downsample_filtered_source_reads (:902-932): the max_reads lowest fgbio name ranks stay, ties in file order
      const bool cand = my_end < 3 && !my_zero && !my_minor && kept[my_end] >= P.min_reads && (int64_t)kept[my_end] > P.max_reads;
      if (cand) S.name_rank[tid] = name_rank(blobL + S.ri[tid].goff + 32, S.ri[tid].name_len);
      __syncthreads();
      if (cand) {
        const int32_t rk = S.name_rank[tid];
        uint32_t before = 0;
        for (uint32_t j = 0; j < n; j++) {
          const ReadInfo& O = S.ri[j];
          if (j == tid || O.end != my_end || O.zero_len || O.minority) continue;
          const int32_t rj = S.name_rank[j];
          before += (rj < rk || (rj == rk && j < tid)) ? 1u : 0u;
        }
        my_down = (int64_t)before >= P.max_reads;
      }
      for (int e = 0; e < 3; e++) down[e] = (uint32_t)__syncthreads_count(my_down && my_end == (uint32_t)e);
    }
    uint32_t fin[3] = {0, 0, 0};                // reads of the end that go into the consensus
    for (int e = 0; e < 3; e++) {
      if (!go || cnt[e] == 0 || cnt[e] < P.min_reads || rem[e] < P.min_reads || kept[e] < P.min_reads) continue;
      fin[e] = kept[e] - down[e];
      if (fin[e] < P.min_reads) continue;
      ok[e] = true; first[e] = n_members; n_members += fin[e];
    }
    // member lists: file order inside an end
    const bool keep = my_end < 3 && ok[my_end] && !my_zero && !my_minor && !my_down;
    const uint32_t wv = tid >> 6, ln = tid & 63;
    uint32_t my_rank = 0;
    for (int e = 0; e < 3; e++) {
      const uint64_t bm = __ballot(keep && my_end == (uint32_t)e);
      if (wv < 2 && ln == 0) S.g_wcnt[wv][e] = (uint32_t)__popcll(bm);
      if (keep && my_end == (uint32_t)e) my_rank = (uint32_t)__popcll(bm & ((1ull << ln) - 1));
    }
    __syncthreads();
    uint32_t my_pos = 0;                      // position inside the end's member list
    if (keep) {
      my_pos = my_rank + (wv == 1 ? S.g_wcnt[0][my_end] : 0);
      S.members[first[my_end] + my_pos] = (uint16_t)tid;
      const ReadInfo& R = S.ri[tid];
      if (P.min_reads <= 1) atomicMax(&S.g_best[my_end], (uint32_t)R.final_len);      // consensus length = the longest kept read
      if (R.has_rx) { atomicAdd(&S.g_rxcnt[my_end], 1u); atomicMin(&S.g_rxpos[my_end], my_pos); }
    }
    __syncthreads();
    if (keep) {
      const ReadInfo& R = S.ri[tid];
      if (P.min_reads > 1) {                  // min_reads-th longest kept read (:1661-1669): every kept read ranks itself
        const uint32_t la = R.final_len;
        uint32_t ge = 0;
        for (uint32_t bq = 0; bq < fin[my_end]; bq++) if (S.ri[S.members[first[my_end] + bq]].final_len >= la) ge++;
        if (ge >= P.min_reads) atomicMax(&S.g_best[my_end], la);
      }
      if (R.has_rx) {                         // UMIs of unequal length (vanilla_caller.rs:1842-1856 → consensus_umis panics)
        const uint32_t len0 = S.ri[S.members[first[my_end] + S.g_rxpos[my_end]]].rx_len;
        if (R.rx_len != len0) S.g_rxbad[my_end] = 1;
      }
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t* st = S.stats;
      st[0] += n;
      if (n_sec) { st[2] += n_sec; st[3 + FGX_REJ_SECONDARY_OR_SUPPLEMENTARY] += n_sec; }
      if (n_reads > 0 && n_reads < P.min_reads) { st[2] += n_reads; st[3 + FGX_REJ_INSUFFICIENT_READS] += n_reads; }
      if (want_defer) defer(S);
      if (go && !want_defer) {
        for (int e = 0; e < 3; e++) {
          if (cnt[e] == 0) continue;
          if (cnt[e] < P.min_reads) { st[2] += cnt[e]; st[3 + FGX_REJ_INSUFFICIENT_READS] += cnt[e]; continue; }
          if (zero[e]) { st[2] += zero[e]; st[3 + FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING] += zero[e]; }
          if (rem[e] < P.min_reads) { if (rem[e]) { st[2] += rem[e]; st[3 + FGX_REJ_INSUFFICIENT_READS] += rem[e]; } continue; }
          if (minor[e]) { st[2] += minor[e]; st[3 + FGX_REJ_MINORITY_ALIGNMENT] += minor[e]; }
          if (kept[e] < P.min_reads) { if (kept[e]) { st[2] += kept[e]; st[3 + FGX_REJ_INSUFFICIENT_READS] += kept[e]; } continue; }
          if (down[e]) { st[2] += down[e]; st[3 + FGX_REJ_DOWNSAMPLED] += down[e]; }
          if (fin[e] < P.min_reads && fin[e]) { st[2] += fin[e]; st[3 + FGX_REJ_INSUFFICIENT_READS] += fin[e]; }
        }
      }
      if (!S.defer) {
        uint32_t ne = 0;
        auto push_end = [&](uint32_t e) {
          S.end_type[ne] = e; S.end_first[ne] = first[e]; S.end_cnt[ne] = fin[e]; S.end_len[ne] = S.g_best[e];
          S.end_maxd[ne] = 0; S.end_mind[ne] = 0xFFFFFFFFu; S.end_sumd[ne] = 0; S.end_sume[ne] = 0;
          // UMIs carried by the kept reads of this end
          const uint32_t rc = S.g_rxcnt[e];
          const uint32_t fr = rc ? S.members[first[e] + S.g_rxpos[e]] : 0;
          const uint32_t len0 = rc ? S.ri[fr].rx_len : 0;
          S.end_rx_cnt[ne] = rc; S.end_rx_len[ne] = len0; S.end_rx_first[ne] = fr;
          if (rc > 1 && S.g_rxbad[e]) defer(S);          // consensus_umis panics on unequal lengths → general path reports it
          if (rc >= 1 && len0 > FAST_RX_CAP) defer(S);
          ne++;
        };
        if (ok[0]) { st[1] += 1; push_end(0); }
        if (ok[1] && ok[2]) { st[1] += 2; push_end(1); push_end(2); }
        else if (ok[1]) { st[2] += fin[1]; st[3 + FGX_REJ_ORPHAN_CONSENSUS] += fin[1]; }
        else if (ok[2]) { st[2] += fin[2]; st[3 + FGX_REJ_ORPHAN_CONSENSUS] += fin[2]; }
        uint32_t total = 0;
        for (uint32_t k = 0; k < ne; k++) { S.end_coloff[k] = total; total += S.end_len[k]; }
        S.n_ends = ne;
        S.col_base = P.col_base[g];   // deterministic scratch slot: exclusive scan of the per-family column bound
      }
    }
  }
  __syncthreads();
  if (S.defer) {
    if (tid == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    return;
  }

  PHB(14)
  // ---- 6. consensus columns (create_consensus_from_source_reads :1652-1755) -------------------------
  const DeviceTables* T = P.T;
  const uint32_t ne = S.n_ends;
  uint32_t total_cols = ne ? S.end_coloff[ne - 1] + S.end_len[ne - 1] : 0;
  for (uint32_t c = tid; c < total_cols; c += NT) {
    uint32_t k = 0;
    while (k + 1 < ne && c >= S.end_coloff[k + 1]) k++;
    uint32_t p = c - S.end_coloff[k];
    uint32_t m0 = S.end_first[k], mc = S.end_cnt[k];
    uint8_t ob, oq;
    uint32_t depth, err;
    if (mc == 1) {   // single-read consensus: LUT keyed by the unclamped quality (:1677-1708)
      uint8_t code, q;
      oriented(S, lb, lq, S.ri[S.members[m0]], p, P.min_input_bq, &code, &q);
      uint8_t adj = q < 94 ? T->single_input_quals[q] : 0;
      if (adj < P.min_cons_bq) { ob = 15; oq = FGX_MIN_PHRED; } else { ob = code; oq = adj; }
      depth = code != 15 ? 1 : 0;
      err = 0;
    } else {
      ChainAcc acc;                             // two Kahan chains while the column shows one base (consensus_math.h)
      acc.reset();
      for (uint32_t a = 0; a < mc; a++) {
        const ReadInfo& R = S.ri[S.members[m0 + a]];
        if (p < R.final_len) {
          uint8_t code, q;
          oriented(S, lb, lq, R, p, P.min_input_bq, &code, &q);
          // N (incl. masked) and IUPAC codes contribute nothing: only the one-hot codes A C G T are observations
          const uint32_t qq = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
          const double2 pr = *(const double2*)&sPairB[qq][0];
          acc.add(__popc((uint32_t)code) == 1, code, pr.x, pr.y);
        }
      }
      double ll[4];
      uint32_t obs[4];
      acc.finish(ll, obs);
      int bi;
      uint8_t q;
      column_call(T->t, ll, obs, &bi, &q);
      depth = obs[0] + obs[1] + obs[2] + obs[3];
      err = depth - (bi == 0 ? obs[0] : bi == 1 ? obs[1] : bi == 2 ? obs[2] : bi == 3 ? obs[3] : 0u);
      const uint8_t LANE_CODE[4] = {1, 2, 4, 8};
      uint8_t code = bi >= 0 ? LANE_CODE[bi] : 15;
      if (depth < P.min_reads) { ob = 15; oq = 0; }
      else if (q < P.min_cons_bq) { ob = 15; oq = FGX_MIN_PHRED; }
      else { ob = code; oq = q; }
    }
    uint32_t d16 = depth < 32767u ? depth : 32767u, e16 = err < 32767u ? err : 32767u;
    uint64_t o = S.col_base + c;
    P.col_code[o] = ob; P.col_qual[o] = oq; P.col_depth[o] = (uint16_t)d16; P.col_err[o] = (uint16_t)e16;
    atomicMax(&S.end_maxd[k], d16); atomicMin(&S.end_mind[k], d16); atomicAdd(&S.end_sumd[k], d16); atomicAdd(&S.end_sume[k], e16);
  }

  PHB(15)
  // ---- 7. consensus UMI per end (simple_umi.rs:46-117): Q20 observations at (Q90, Q90) -----------------
  {
    const DeviceTables* TU = P.TU;
    for (uint32_t w = tid; w < ne * FAST_RX_CAP; w += NT) {
      uint32_t k = w / FAST_RX_CAP, i = w % FAST_RX_CAP;
      uint32_t cnt = S.end_rx_cnt[k];
      if (cnt == 0 || i >= S.end_rx_len[k]) continue;
      if (cnt == 1) { const ReadInfo& R = S.ri[S.end_rx_first[k]]; S.end_rx[k][i] = (char)blobL[R.goff + R.rx_off + i]; continue; }
      ColumnAcc acc;
      acc.reset();
      uint32_t non_dna = 0, seen = 0;
      uint8_t fc = 0;
      bool mixed = false;
      for (uint32_t a = 0; a < S.end_cnt[k]; a++) {
        const ReadInfo& R = S.ri[S.members[S.end_first[k] + a]];
        if (!R.has_rx) continue;
        uint8_t ch = blobL[R.goff + R.rx_off + i];
        if (seen == 0) fc = ch;
        seen++;
        uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
        bool dna = up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'N';
        if (dna) { int lane = bam::ascii_to_lane(ch); if (lane != 255) acc.add(lane, TU->t.correct[20], TU->t.error_per_alt[20]); }
        else { non_dna++; if (ch != fc) mixed = true; }
      }
      char out;
      if (non_dna == 0) { int bi; uint8_t q; column_call(TU->t, acc.s, acc.obs, &bi, &q); out = bi >= 0 ? "ACGT"[bi] : 'N'; }
      else if (non_dna == seen && !mixed) out = (char)fc;
      else { out = '?'; S.rx_bad = 1; }      // the reference panics here
      S.end_rx[k][i] = out;
    }
  }
  __syncthreads();
  if (S.rx_bad) {   // undo: the general path raises the reference's error
    if (tid == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; }
    return;
  }

  // ---- 8. descriptors + stats ---------------------------------------------------------------------------
  if (tid < ne) {
    uint32_t k = tid;
    EndDesc D;
    memset(&D, 0, sizeof(D));
    uint32_t Lc = S.end_len[k];
    D.col_off = S.col_base + S.end_coloff[k];
    D.cons_len = Lc;
    D.first_off = S.ri[0].goff;
    D.kept_off = S.ri[S.members[S.end_first[k]]].goff;
    D.type = (uint8_t)S.end_type[k];
    const uint32_t maxd = Lc ? S.end_maxd[k] : 0, mind = Lc ? S.end_mind[k] : 0;   // widths of the cD / cM integer tags
    D.mi_off = S.ri[0].mi_off; D.mi_len = S.ri[0].mi_len;
    const ReadInfo& F = S.ri[S.members[S.end_first[k]]];
    D.has_cb = (P.cell0 && F.has_cb) ? 1 : 0; D.cb_off = F.cb_off; D.cb_len = F.cb_len;
    D.has_rx = S.end_rx_cnt[k] > 0; D.rx_len = (uint8_t)S.end_rx_len[k];
    for (uint32_t i = 0; i < D.rx_len; i++) D.rx[i] = S.end_rx[k][i];
    D.valid = 1;
    uint32_t name_len = P.prefix_len + 1 + D.mi_len;
    uint32_t size = 32 + name_len + 1 + (Lc + 1) / 2 + Lc + (3 + P.rg_len + 1) + (3 + int_tag_width(maxd)) + (3 + int_tag_width(mind)) + 7 +
                    (P.per_base_tags ? 2 * (8 + 2 * Lc) : 0) + (3 + D.mi_len + 1) + (D.has_cb ? 3 + D.cb_len + 1 : 0) +
                    (D.has_rx ? 3 + D.rx_len + 1 : 0);
    D.rec_size = size;
    // emission order inside a family: fragment, R1, R2 → slot = end type
    P.ends[slot0 + D.type] = D;
    P.rec_sizes[slot0 + D.type] = (uint64_t)size + 4;
  }
  if (tid < FGX_STATS_LEN && S.stats[tid]) atomicAdd(&P.stats[(size_t)(blockIdx.x & (STAT_SLOTS - 1)) * 32 + tid], (unsigned long long)S.stats[tid]);
}

// -----------------------------------------------------------------------------------------------------
// k_family_wave — the common case: ONE WAVEFRONT PER FAMILY (≤ 64 records whose raw bytes fit the wave's
// LDS slice).  The family's records are contiguous in the BAM stream, so the wave copies that byte span
// into LDS with coalesced 16-byte loads (one pass over HBM at full line efficiency) and then does ALL
// parsing, aux-tag walking, overlap correction and column reads out of LDS.  Lane r owns record r; per-read
// state lives in lane r's registers and is broadcast with v_readlane, the family gates are ballots and
// popcounts — no serial lane-0 section and no block barrier.  Semantics identical to k_family below.
// -----------------------------------------------------------------------------------------------------
// column_call_fast for the wave kernel: same decisions as consensus_math.h's unanimous_fast_path, arranged for the
// GPU — the four likelihoods stay in registers (selects instead of dynamic indexing, which would go through scratch
// memory) and the bracket q with thresholds[q] <= gap < thresholds[q+1] comes from the qguess hint plus ONE parallel
// probe of five neighbouring thresholds instead of a 7-step dependent binary search through LDS.
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long u) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double uniform_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double sel4(const double* v, uint32_t i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; }
__device__ __forceinline__ bool column_call_fast_lds(const ConsensusTables& T, const CallConst& K, const double* ll, const uint32_t* obs,
                                                     int* base_idx, uint8_t* qual) {
  if (obs[0] + obs[1] + obs[2] + obs[3] == 0) { *base_idx = -1; *qual = FGX_MIN_PHRED; return true; }
  const uint32_t n_obs = (obs[0] > 0) + (obs[1] > 0) + (obs[2] > 0) + (obs[3] > 0);
  if (n_obs != 1) return false;
  const uint32_t observed = obs[3] > 0 ? 3u : obs[2] > 0 ? 2u : obs[1] > 0 ? 1u : 0u;
  const double w = sel4(ll, observed), l = sel4(ll, (observed + 1) & 3);
  const double gap = w - l;
  if (!(m_isfinite(gap) && gap > FGX_DBL_EPSILON)) return false;
  if (gap >= K.cap_threshold) {
    const double delta = unanimous_margin(w, l, 1.0);
    if (gap - K.cap_threshold >= FGX_LN_2 && delta < K.half_cerr_at_cap) { *base_idx = (int)observed; *qual = (uint8_t)K.cap; return true; }
    return false;
  }
  uint32_t lo;
  double thq, thq1;
  bool probed = false;
  if (gap < 32.0) {
    const uint32_t b = T.qguess[(uint32_t)(gap * 8.0)];            // thresholds[0..b) <= gap for sure; b >= 1 (thresholds[0] = 0)
    if (b >= 1) {
      double t[5];
#pragma unroll
      for (int j = 0; j < 5; j++) { uint32_t idx = b - 1 + j; t[j] = T.thresholds[idx < 93 ? idx : 93]; }
      const uint32_t c = (t[1] <= gap) + (t[2] <= gap) + (t[3] <= gap) + (t[4] <= gap);
      if (c < 4) { lo = b + c; thq = c == 0 ? t[0] : c == 1 ? t[1] : c == 2 ? t[2] : t[3]; thq1 = c == 0 ? t[1] : c == 1 ? t[2] : c == 2 ? t[3] : t[4]; probed = true; }
    }
  }
  if (!probed) {
    uint32_t a = 0, hi = 94;
    while (a < hi) { uint32_t mid = a + (hi - a) / 2; if (T.thresholds[mid] <= gap) a = mid + 1; else hi = mid; }
    lo = a; thq = T.thresholds[lo - 1]; thq1 = T.thresholds[lo];
  }
  const uint32_t q = lo - 1;
  const double margin = unanimous_margin(w, l, T.cerr_min[q]);
  if (gap - thq > margin && thq1 - gap > margin) { *base_idx = (int)observed; *qual = (uint8_t)q; return true; }
  return false;
}

#ifndef FGX_WAVE_OCC
#define FGX_WAVE_OCC 5   /* measured on MI355X, 1M depth-8 families: occ 4 19.2 ms, 5 17.2 ms, 6 18.1 ms */
#endif
// MODE 0: simplex (vanilla caller).  MODE 1: duplex — phases 1-4 are shared, the strand partition, the four single-strand
// column sets and the descriptors of the two duplex records replace phases 5-8 (see the MODE == 1 branch).
// What k_family_wave keeps in LDS beside the wave slices, as one block with a host-built image (FastPath::d_fwimg), copied like
// k_simplex_wave2's (W2Lds): 16-byte loads that leave with the family's first loads, LDS writes after the records are staged.
struct FwLds {
  ConsensusTables t;                 // thresholds / cerr_min / scalars (and the plain tables for reference)
  alignas(16) double pair[94][2];    // {correct[q], error_per_alt[q]} interleaved: one ds_read_b128 per observation
  uint8_t tagcls[256];               // aux value type classes (aux_walk)
};
static_assert(sizeof(FwLds) % 16 == 0 && sizeof(FwLds) <= 6 * 64 * 16, "FwLds is copied in 16-byte pieces, six per thread at most");
inline void build_fw_image(FwLds& L, const ConsensusTables& t) {
  memset(&L, 0, sizeof(L));
  L.t = t;
  for (uint32_t i = 0; i < 94; i++) { L.pair[i][0] = t.correct[i]; L.pair[i][1] = t.error_per_alt[i]; }
  for (uint32_t i = 0; i < 256; i++) { const int fx = bam::tag_fixed_size((uint8_t)i); L.tagcls[i] = (uint8_t)(fx ? fx : i == 'Z' ? 8 : i == 'H' ? 16 : i == 'B' ? 32 : 0); }
}

template <int MODE>
#ifndef FGX_WAVE_OCC_DUPLEX
#define FGX_WAVE_OCC_DUPLEX 4   /* the duplex branch holds four column sets' worth of state beside the record lanes: at 5 waves per SIMD (96 VGPRs) its
                                   member loop spilled */
#endif
__global__ __launch_bounds__(256, MODE == 1 ? FGX_WAVE_OCC_DUPLEX : FGX_WAVE_OCC) void k_family_wave(FastParams P, uint32_t n_grp_total) {
  FGX_DYN_LDS(dyn);
  __shared__ __align__(16) FwLds sL;
  ConsensusTables& sT = sL.t;
  double (&sPair)[94][2] = sL.pair;
  uint8_t* const sTagCls = sL.tagcls;
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t gi_raw = blockIdx.x * (blockDim.x >> 6) + wv;   // n_grp_total counts the groups of this launch (all of them, or a retry list)
  bool bail = gi_raw >= n_grp_total;                          // a wavefront without a group still takes part in the table copy
  const uint32_t gi = bail ? 0u : gi_raw;
  const uint32_t g = P.group_list ? P.group_list[gi] : P.g0 + gi;   // (asked for before the image: waiting for it does not wait for the image)
  // the table image: global -> registers now, registers -> LDS after the records have been staged (one barrier for both)
  constexpr uint32_t IMG_V = (uint32_t)(sizeof(FwLds) / 16);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* const img_src = (const u32x4*)P.fw_image;
  const uint32_t ii0 = threadIdx.x, ii1 = ii0 + blockDim.x, ii2 = ii1 + blockDim.x, ii3 = ii2 + blockDim.x, ii4 = ii3 + blockDim.x, ii5 = ii4 + blockDim.x;
  const u32x4 im0 = img_src[ii0 < IMG_V ? ii0 : 0u], im1 = img_src[ii1 < IMG_V ? ii1 : 0u], im2 = img_src[ii2 < IMG_V ? ii2 : 0u],
              im3 = img_src[ii3 < IMG_V ? ii3 : 0u], im4 = img_src[ii4 < IMG_V ? ii4 : 0u], im5 = img_src[ii5 < IMG_V ? ii5 : 0u];
  uint8_t* W = dyn + (size_t)wv * P.lds_wave_bytes;
  const uint32_t my_list = blockIdx.x & (N_LISTS - 1);
  bool list_overflow = false;
  // wave-aggregated append of the columns whose call needs call_full (processed densely by k_call_full)
  auto push_full = [&](bool need, uint64_t dest, const double* ll, const uint32_t* obs) {
    bool pending = need;
    uint32_t l = my_list;
    for (uint32_t tries = 0;; tries++) {           // a full list chains to the next one; only a globally full pool overflows
      unsigned long long m = __ballot(pending);
      if (!m) break;
      if (tries >= N_LISTS) { list_overflow |= pending; break; }
      uint32_t cnt = (uint32_t)__popcll(m), leader = (uint32_t)__builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&P.full_count[l], cnt);
      base = __shfl(base, leader);
      if (pending) {
        uint32_t idx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (idx < P.full_cap) {
          FullItem* it = &P.full_items[(size_t)l * P.full_cap + idx];
          it->dest = dest; it->ll[0] = ll[0]; it->ll[1] = ll[1]; it->ll[2] = ll[2]; it->ll[3] = ll[3];
          it->obs = obs[0] | (obs[1] << 8) | (obs[2] << 16) | (obs[3] << 24); it->chains = 0;
          pending = false;
        }
      }
      l = (l + 1) & (N_LISTS - 1);
    }
  };
  unsigned long long* st = P.stats + (size_t)(blockIdx.x & (STAT_SLOTS - 1)) * 32;
  const uint32_t r0 = P.grp_first[g], n = P.grp_first[g + 1] - r0;
  const uint32_t slot0 = 3 * g;
  if (!bail && lane < 3) { if (MODE == 0) P.ends[slot0 + lane].valid = 0; else if (MODE == 1) P.dends[slot0 + lane].valid = 0; else P.cends[slot0 + lane].valid = 0; P.rec_sizes[slot0 + lane] = 0; }

  if (MODE == 0 && !bail && n < P.min_reads) {   // simplex.rs:673-683
    if (lane == 0) { atomicAdd(&st[0], (unsigned long long)n); atomicAdd(&st[2], (unsigned long long)n); atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_READS], (unsigned long long)n); }
    bail = true;
  }
  auto to_defer = [&]() { if (lane == 0) { uint32_t k = atomicAdd(P.n_deferred, 1u); P.deferred[k] = g; } };
  // does not fit this launch's LDS slice: the next launch (bigger slices, fewer waves per CU) picks it up; after the last one
  // simplex families go to the workgroup-per-family kernel, duplex / CODEC molecules to the general path
  auto to_retry = [&]() { if (!P.retry) { to_defer(); return; } if (lane == 0) { uint32_t k = atomicAdd(P.n_retry, 1u); P.retry[k] = g; } };
  if (!bail && n > 64) {
    if (MODE == 0 && P.big) { if (lane == 0) { uint32_t k = atomicAdd(P.n_big, 1u); P.big[k] = g; } }   // k_family's, directly
    else if (MODE == 0) to_retry();
    else to_defer();
    bail = true;
  }
  // an empty group has no byte span (the minimum below would be ~0 and the staging loop would read the 16 bytes BEFORE the blob —
  // a fault when the blob starts an allocation): the general path emits its nothing
  if (!bail && n == 0) { to_defer(); bail = true; }

#if FGX_PHASE_TIMING
  unsigned long long _t = __builtin_amdgcn_s_memtime();
#endif
  // ---- 1. stage the family's raw records into LDS ------------------------------------------------------
  const bool act = !bail && lane < n;
  unsigned long long off = act ? P.rec_off[r0 + lane] : ~0ull;
  uint32_t len = act ? P.rec_len[r0 + lane] : 0;
  unsigned long long lo_off = wave_min64(off);
  unsigned long long hi_end = wave_max64(act ? off + len : 0ull);
  if (!bail && (__any(act && len < 32) || hi_end > P.blob_len)) { to_defer(); bail = true; }   // a record outside the blob: the general path raises the error
  unsigned long long base16 = lo_off & ~15ull;
  unsigned long long span = hi_end - base16;
  if (!bail && span + 16 > (unsigned long long)P.lds_wave_bytes) { to_retry(); bail = true; }   // +16: slack for the dword-composed reads
  const uint32_t span16 = bail ? 0u : ((uint32_t)span + 15) & ~15u;
  {
    const uint8_t* src = P.blob + base16;
    for (uint32_t i = lane * 16; i < span16; i += 64 * 16) *(uint4*)(W + i) = *(const uint4*)(src + i);
  }
  {
    u32x4* const img_dst = (u32x4*)&sL;
    if (ii0 < IMG_V) img_dst[ii0] = im0;
    if (ii1 < IMG_V) img_dst[ii1] = im1;
    if (ii2 < IMG_V) img_dst[ii2] = im2;
    if (ii3 < IMG_V) img_dst[ii3] = im3;
    if (ii4 < IMG_V) img_dst[ii4] = im4;
    if (ii5 < IMG_V) img_dst[ii5] = im5;
  }
  __syncthreads();                                            // the tables (workgroup) and this wavefront's records are in LDS
  if (bail) return;

  PH(1)
  // ---- 2. parse: lane r owns record r ---------------------------------------------------------------------
  const uint32_t lo = act ? (uint32_t)(off - base16) : 0;       // LDS offset of this lane's record
  uint32_t flags = 0, l_seq = 0, seq_lo = 0, qual_lo = 0, name_len = 0, clip = 0, hash = 0;
  int32_t pos = 0, ref_id = 0;
  uint32_t mi_lo = 0, mi_len = 0, rx_lo = 0, rx_len = 0, cb_lo = 0, cb_len = 0;   // LDS offsets of tag values
  bool has_mi = false, has_rx = false, has_cb = false, excluded = false, bad = false;
  bool long_cigar = false;          // simplex: more CIGAR ops than MAX_CIG_OPS — nothing below may index `ops` by them
  bool cplx = false;                // simplex: a CIGAR with I / D / N / P ops — the family goes to the workgroup-per-family kernel
  uint32_t lead_s = 0, m_len = 0;   // query offset of the first aligned base, length of the aligned block
  uint32_t strand = 0;   // duplex: 1 = MI ends in /A, 2 = /B
  if (act) {
    const uint32_t h2 = ld32u(W, lo + 8), h3 = ld32u(W, lo + 12);
    const uint32_t l_name = h2 & 0xFF, n_cig = h3 & 0xFFFF;
    l_seq = ld32u(W, lo + 16);
    flags = h3 >> 16;
    unsigned long long seq_off = 32ull + l_name + 4ull * n_cig;
    unsigned long long qual_off = seq_off + ((unsigned long long)l_seq + 1) / 2;
    unsigned long long aux_off = qual_off + l_seq;
    if (aux_off > len || l_seq > 65535 || l_name == 0) bad = true;
    else {
      name_len = l_name - 1;
      ref_id = (int32_t)ld32u(W, lo); pos = (int32_t)ld32u(W, lo + 4);
      seq_lo = lo + (uint32_t)seq_off; qual_lo = lo + (uint32_t)qual_off;
      excluded = (flags & (bam::F_SECONDARY | bam::F_SUPPLEMENTARY)) != 0;
      if (MODE == 1 && excluded) bad = true;   // the duplex caller has no secondary/supplementary filter: general path
      if (MODE == 2 && (excluded || !(flags & bam::F_PAIRED))) bad = true;   // CODEC: fragments / non-primary records take the general path
      // CIGAR: one aligned block of M/=/X ops, optionally between soft / hard clips (what an aligner gives a read without
      // indels).  Such reads all simplify to (M, length) for the alignment filter; clips only shift query offsets.
      uint32_t ops[MAX_CIG_OPS];
      ops[0] = 0;
      if (!excluded) {
        if ((flags & bam::F_UNMAPPED) || n_cig == 0 || l_seq == 0 || pos < 0) bad = true;
        else if (n_cig == 1) {
          ops[0] = ld32u(W, lo + 32 + l_name);
          const uint32_t ty = ops[0] & 15;
          if (!(ty == 0 || ty == 7 || ty == 8) || (ops[0] >> 4) != l_seq) bad = true;
          m_len = l_seq;
        } else if (MODE != 2 && n_cig <= MAX_CIG_OPS) {
          uint32_t phase = 0, trail_s = 0;          // 0: leading clips, 1: aligned block, 2: trailing clips
          bool okc = true;
#pragma unroll
          for (uint32_t i = 0; i < MAX_CIG_OPS; i++) {
            if (i >= n_cig) break;
            const uint32_t o = ld32u(W, lo + 32 + l_name + 4 * i), t = o & 15, ln = o >> 4;
            ops[i] = o;
            if (t == 0 || t == 7 || t == 8) { if (phase == 2) okc = false; phase = 1; m_len += ln; }
            else if (t == 4) { if (phase == 0) lead_s += ln; else { phase = 2; trail_s += ln; } }
            else if (t == 5) { if (phase == 1) phase = 2; }
            else { okc = false; if (t == 1 || t == 2 || t == 3 || t == 6) cplx = true; }   // I, D, N, P: the workgroup kernel
          }
          if (cplx && MODE == 0) { /* decided below: the whole family moves to the workgroup-per-family kernel */ }
          else if (!okc || m_len == 0 || m_len > 65535 || (unsigned long long)lead_s + m_len + trail_s != l_seq) bad = true;
        } else if (MODE == 0 && n_cig <= WG_CIG_OPS) { cplx = true; long_cigar = true; }   // (more ops than this kernel reads: the workgroup kernel's)
        else bad = true;
      }
      // aux walk in LDS (tags.rs:13-34): first occurrence of MC / <tag> / RX / <cell tag>
      const uint32_t a0 = lo + (uint32_t)aux_off, an = len - (uint32_t)aux_off;
      AuxTags ax;
      aux_walk(W, sTagCls, a0, an, P, ax);
      if (ax.oddw) bad = true;
      const bool has_mc = (ax.got & 1u) != 0;
      const uint32_t mc_lo = ax.pk_mc & 0xFFFF, mc_len = ax.pk_mc >> 16;
      has_mi = (ax.got & 2u) != 0; mi_lo = ax.pk_mi & 0xFFFF; mi_len = ax.pk_mi >> 16;
      has_rx = (ax.got & 4u) != 0; rx_lo = ax.pk_rx & 0xFFFF; rx_len = ax.pk_rx >> 16;
      has_cb = (ax.got & 8u) != 0; cb_lo = ax.pk_cb & 0xFFFF; cb_len = ax.pk_cb >> 16;
      if (MODE == 1) {   // strand from the MI suffix; every paired read needs `<id>/A` or `<id>/B` (duplex_caller.rs:2557-2566: fatal otherwise)
        if (has_mi && mi_len >= 2 && W[mi_lo + mi_len - 2] == '/') { const uint8_t sc = W[mi_lo + mi_len - 1]; strand = sc == 'A' ? 1u : sc == 'B' ? 2u : 0u; }
        if ((flags & bam::F_PAIRED) && (strand == 0 || P.prefix_len + 1 + (mi_len - 2) >= 255)) bad = true;
      }
      if (MODE == 2 && !bad) {   // is_fr_pair_raw of THIS record (overlap.rs:21-69); only the reverse record's answer is used
        const int32_t mpos = (int32_t)ld32u(W, lo + 24), mref = (int32_t)ld32u(W, lo + 20);
        if (pos >= (1 << 30) || mpos < -1 || mpos >= (1 << 30)) bad = true;
        const bool rv = (flags & bam::F_REVERSE) != 0, mrv = (flags & bam::F_MATE_REVERSE) != 0;
        const bool okp = !(flags & bam::F_MATE_UNMAPPED) && ref_id == mref && rv != mrv;
        strand = (okp && ((long long)mpos + 1 < (long long)pos + 1 + ((long long)l_seq - 1))) ? 1u : 0u;   // `strand` doubles as frself here
      }
      if (MODE != 2 && !bad && !excluded && !long_cigar) {
        // mate-overlap clip (raw-bam/overlap.rs:181-357).  Closed form when the MC tag is one M op and all
        // coordinates are in the ordinary range (no saturating arithmetic can trigger); general code otherwise.
        bool fastclip = false;
        uint32_t ML = 0;
        if (n_cig == 1 && has_mc && mc_len >= 2 && mc_len <= 8) {
          unsigned long long v = ld64u(W, mc_lo);
          uint32_t k = 0, val = 0;
          while (k < mc_len - 1) { uint32_t ch = (uint32_t)(v >> (8 * k)) & 0xFF; if (ch < '0' || ch > '9') break; val = val * 10 + (ch - '0'); k++; }
          if (k == mc_len - 1 && ((v >> (8 * k)) & 0xFF) == 'M' && val > 0) { fastclip = true; ML = val; }
        }
        const int32_t mpos = (int32_t)ld32u(W, lo + 24);
        if (fastclip && pos < (1 << 30) && mpos >= 0 && mpos < (1 << 30)) {
          uint64_t cl = 0;
          const int32_t mref = (int32_t)ld32u(W, lo + 20);
          const bool rv = (flags & bam::F_REVERSE) != 0, mrv = (flags & bam::F_MATE_REVERSE) != 0;
          const long long L = l_seq, tp = (long long)pos + 1, mp = (long long)mpos + 1;
          bool fr = (flags & bam::F_PAIRED) && !(flags & bam::F_MATE_UNMAPPED) && ref_id == mref && rv != mrv;
          if (fr) fr = rv ? (mp < tp + (L - 1)) : (tp < mp + ((long long)ML - 1));
          if (fr) {
            const long long read_end = tp - 1 + L, mate_end = mp - 1 + (long long)ML;
            if (rv) {
              if (!(tp > mate_end) && !(read_end < mp)) {
                long long fs = tp > mp ? tp : mp;
                long long rb = fs - tp; if (rb > L) rb = L;
                long long mb = fs - mp; if (mb > (long long)ML) mb = ML;
                cl = rb > mb ? (uint64_t)(rb - mb) : 0;
              }
            } else {
              if (!(read_end < mp) && !(mate_end < tp)) {
                long long ls = read_end < mate_end ? read_end : mate_end;
                long long ra = ls - tp + 1; if (ra > L) ra = L;
                long long ma = ls - mp + 1; if (ma > (long long)ML) ma = ML;
                long long rp = L - ra, mq = (long long)ML - ma;
                cl = rp > mq ? (uint64_t)(rp - mq) : 0;
              }
            }
          }
          clip = (uint32_t)(cl > 65535 ? 65535 : cl);
        } else {
          uint32_t mops[MAX_MC_OPS];
          bool overflow = false;
          bam::Rec v{W + lo, len};
          unsigned long long cl = bam::mate_clip(v, ops, n_cig, has_mc ? W + mc_lo : nullptr, mc_len, mops, MAX_MC_OPS, &overflow);
          if (overflow) { if (MODE == 0) cplx = true; else bad = true; }   // (a mate of more ops than this kernel parses: the workgroup kernel's, which takes WG_MC_OPS)
          clip = (uint32_t)(cl > 65535 ? 65535 : cl);
        }
      }
      if (!bad && !excluded) {
        // name hash (8 bytes per step) for mate pairing
        unsigned long long h = 0x9E3779B97F4A7C15ULL ^ name_len;
        for (uint32_t i = 0; i < name_len; i += 8) {
          unsigned long long w = ld64u(W, lo + 32 + i);
          if (i + 8 > name_len) w &= (1ULL << (8 * (name_len - i))) - 1;
          h = (h ^ w) * 0xBF58476D1CE4E5B9ULL;
        }
        hash = (uint32_t)(h >> 32);
      }
    }
  }
  // first record must carry the MI tag and give a legal read name (vanilla_caller.rs:1897-1908, 1795-1797)
  const uint32_t mi0_lo = rlane(mi_lo, 0), mi0_len = rlane(mi_len, 0);
  if (MODE != 1 && lane == 0 && (!has_mi || P.prefix_len + 1 + mi_len >= 255)) bad = true;   // (CODEC without an MI names reads by a running counter: general path)
  // absent qualities (all 0xFF) are a fatal input error (:1119-1124)
  if (MODE != 2 && act && !bad && !excluded && W[qual_lo] == 0xFF) {
    bool all = true;
    for (uint32_t i = 0; i < l_seq; i++) if (W[qual_lo + i] != 0xFF) { all = false; break; }
    if (all) bad = true;
  }
  if (__any(bad)) { to_defer(); return; }
  if (MODE == 0 && __any(cplx)) { to_retry(); return; }
  const bool cand = act && !excluded;
  const bool rev = (flags & bam::F_REVERSE) != 0;
  const unsigned long long rxmask = __ballot(cand && has_rx), cbmask = __ballot(cand && has_cb);

  PH(2)
  // ---- 3. overlapping-bases pre-correction in LDS (overlapping.rs:236-336, 627-684) ----------------------
  uint32_t ov_bases = 0, ov_agree = 0, ov_dis = 0, ov_corr = 0;
  bool do_overlap = P.overlap != 0;
  if (MODE == 2) do_overlap = false;
  if (MODE == 1 && do_overlap && P.dmin_yx != 0)   // duplex.rs:786-795: only molecules with both strands when single-strand output is off
    do_overlap = n >= 2 && __any(act && strand == 1) && __any(act && strand == 2);
  if (do_overlap) {
    const bool is_r1 = cand && (flags & bam::F_FIRST);
    const bool is_r2 = cand && !(flags & bam::F_FIRST) && (flags & bam::F_LAST);
    // pair map: per name, the LAST R1-type and the LAST R2-type primary record
    int mate = -1;
    bool later_r1 = false;
    const unsigned long long r1mask = __ballot(is_r1), r2mask = __ballot(is_r2);
    // candidates of each R1 lane: the other R1/R2 lanes with the same (name hash, name length) — one compare per lane
    // pair here, the byte-wise name check below runs for all R1 lanes at once (usually a single round: the mate)
    uint32_t eq_lo = 0, eq_hi = 0;
    for (unsigned long long um = r1mask | r2mask; um; um &= um - 1) {
      const uint32_t u = (uint32_t)__builtin_ctzll(um);
      const uint32_t hash_u = rlane(hash, u), nlen_u = rlane(name_len, u);      // (both cross-lane reads by every lane: no short circuit between them)
      const bool eq = hash_u == hash && nlen_u == name_len && u != lane;
      if (u < 32) eq_lo |= (eq ? 1u : 0u) << u; else eq_hi |= (eq ? 1u : 0u) << (u - 32);
    }
    unsigned long long todo = is_r1 ? ((unsigned long long)eq_lo | ((unsigned long long)eq_hi << 32)) : 0ull;
    while (__any(todo != 0)) {
      const bool mine = todo != 0;
      const uint32_t u = mine ? (uint32_t)__builtin_ctzll(todo) : lane;
      todo &= todo - 1;
      const uint32_t lou = (uint32_t)__shfl((int)lo, (int)u);
      if (mine) {
        bool same = true;
        for (uint32_t i = 0; i < name_len; i += 8) {
          unsigned long long wa = ld64u(W, lou + 32 + i), wb = ld64u(W, lo + 32 + i);
          if (i + 8 > name_len) { unsigned long long mk = (1ULL << (8 * (name_len - i))) - 1; wa &= mk; wb &= mk; }
          if (wa != wb) { same = false; break; }
        }
        if (same) {
          if ((r1mask >> u) & 1) { if (u > lane) later_r1 = true; }
          else mate = (int)u;     // ascending u: the LAST R2-type record of the name wins
        }
      }
    }
    if (later_r1) mate = -1;
    // Every overlapping position of every pair is one work item; items are laid out pair after pair (exclusive scan
    // of the per-pair overlap lengths) and taken 64 at a time, so short overlaps of several pairs share an iteration.
    const int msrc = mate >= 0 ? mate : (int)lane;
    const int32_t pos_b = (int32_t)__shfl((int)pos, msrc), ref_b = (int32_t)__shfl((int)ref_id, msrc);
    const uint32_t aln_a = lead_s | (m_len << 16), aln_b = (uint32_t)__shfl((int)aln_a, msrc);   // query offset of the aligned block | its length
    const uint32_t dsc_a = seq_lo | (qual_lo << 16), dsc_b = (uint32_t)__shfl((int)dsc_a, msrc);
    uint32_t cntp = 0, off1 = 0, off2 = 0;
    if (mate >= 0 && ref_id == ref_b) {
      const long long s1 = (long long)pos + 1, e1 = (long long)pos + (aln_a >> 16), s2 = (long long)pos_b + 1, e2 = (long long)pos_b + (aln_b >> 16);
      const long long lox = s1 > s2 ? s1 : s2, hix = e1 < e2 ? e1 : e2;
      if (hix >= lox) { cntp = (uint32_t)(hix - lox + 1); off1 = (uint32_t)(lox - s1) + (aln_a & 0xFFFF); off2 = (uint32_t)(lox - s2) + (aln_b & 0xFFFF); }
    }
    uint32_t incl = cntp;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += t; }
    const uint32_t first_item = incl - cntp, n_items = rlane(incl, 63);
    for (uint32_t base = 0; base < n_items; base += 64) {
      const uint32_t w = base + lane;
      bool have = false;
      uint32_t i1 = 0, i2 = 0, sa = 0, sb = 0, qa0 = 0, qb0 = 0;
      for (unsigned long long pm = __ballot(cntp != 0 && first_item < base + 64 && first_item + cntp > base); pm; pm &= pm - 1) {
        const uint32_t a = (uint32_t)__builtin_ctzll(pm);
        const uint32_t st_a = rlane(first_item, a), cn_a = rlane(cntp, a), da = rlane(dsc_a, a), db = rlane(dsc_b, a);
        const uint32_t o1a = rlane(off1, a), o2a = rlane(off2, a);
        if (w >= st_a && w - st_a < cn_a) {
          have = true;
          i1 = o1a + (w - st_a); i2 = o2a + (w - st_a);
          sa = da & 0xFFFF; qa0 = da >> 16; sb = db & 0xFFFF; qb0 = db >> 16;
        }
      }
      if (have) {
        uint32_t o1 = sa + (i1 >> 1), o2 = sb + (i2 >> 1);
        uint8_t c1 = (i1 & 1) ? (W[o1] & 15) : (W[o1] >> 4), c2 = (i2 & 1) ? (W[o2] & 15) : (W[o2] >> 4);
        if (c1 != 15 && c2 != 15) {
          ov_bases++;
          uint8_t qa = W[qa0 + i1], qb = W[qb0 + i2];
          if (c1 == c2) {
            ov_agree++;
            uint32_t sm = (uint32_t)qa + qb;
            uint8_t nq = (uint8_t)(sm < 93 ? sm : 93);
            W[qa0 + i1] = nq; W[qb0 + i2] = nq;
            if (nq != qa || nq != qb) ov_corr++;
          } else {
            ov_dis++;
            uint8_t cb, cq;
            if (qa == qb) { cb = 15; cq = FGX_MIN_PHRED; }
            else if (qa > qb) { cb = c1; cq = (uint8_t)(qa - qb); if (cq < FGX_MIN_PHRED) cq = FGX_MIN_PHRED; }
            else { cb = c2; cq = (uint8_t)(qb - qa); if (cq < FGX_MIN_PHRED) cq = FGX_MIN_PHRED; }
            // nibble updates: neighbouring lanes own the other nibble of the same byte → 32-bit LDS atomics
            auto setn = [&](uint32_t o, uint32_t idx, uint8_t code) {
              uint32_t* wd = (uint32_t*)(W + (o & ~3u));
              uint32_t sh = 8 * (o & 3) + ((idx & 1) ? 0 : 4);
              atomicAnd(wd, ~(0xFu << sh));
              atomicOr(wd, (uint32_t)code << sh);
            };
            setn(o1, i1, cb); setn(o2, i2, cb);
            W[qa0 + i1] = cq; W[qb0 + i2] = cq;
            ov_corr += 2;
          }
        }
      }
    }
  }
  wave_sync();

  // Oriented, mask-aware view of one read at consensus position p.  d0 = seq_lo | qual_lo << 16, `rv` uniform.
  // Returns the 4-bit code in read orientation (15 = N / masked) and the SourceRead quality.
  auto view = [&](uint32_t s_lo, uint32_t q_lo, uint32_t L, bool rv, uint32_t trim_to, uint32_t p, uint32_t* code, uint32_t* qual) {
    uint32_t idx = rv ? L - 1 - p : p;
    uint32_t bb = W[s_lo + (idx >> 1)];
    uint32_t c = (bb >> ((~idx & 1) << 2)) & 15;
    uint32_t q = W[q_lo + idx];
    if (rv) c = comp_code((uint8_t)c);
    if (p < trim_to && q < P.min_input_bq) { c = 15; q = FGX_MIN_PHRED; }
    *code = c; *qual = q;
  };

  PH(3)
  // ---- 4. source-read geometry per lane (vanilla_caller.rs:1129-1160) -----------------------------------------
  uint32_t trim_to = l_seq, final_len = 0;
  if (MODE != 2 && cand && P.trim) {
    uint32_t tq = P.min_input_bq;
    if (tq < 1 || l_seq == 0) trim_to = 0;
    else {
      int32_t score = 0, max_score = 0;
      uint32_t point = l_seq;
      for (uint32_t i = l_seq; i-- > 0;) {
        uint32_t idx = rev ? l_seq - 1 - i : i;
        score += (int32_t)tq - (int32_t)W[qual_lo + idx];
        if (score < 0) break;
        if (score > max_score) { max_score = score; point = i; }
      }
      trim_to = point;
    }
  }
  if (MODE != 2) {
    uint32_t clip_pos = l_seq > clip ? l_seq - clip : 0;
    uint32_t fl = clip_pos < trim_to ? clip_pos : trim_to;
    bool tail_n = false;
    if (cand && fl > 0) { uint32_t c, q; view(seq_lo, qual_lo, l_seq, rev, trim_to, fl - 1, &c, &q); tail_n = (c == 15); }
    final_len = cand ? fl : 0;
    // reads ending in N / masked bases: strip the run 64 positions at a time (lanes scan backwards together)
    for (unsigned long long tm = __ballot(tail_n); tm; tm &= tm - 1) {
      const uint32_t r = (uint32_t)__builtin_ctzll(tm);
      uint32_t flr = rlane(final_len, r);
      const uint32_t s_r = rlane(seq_lo, r), q_r = rlane(qual_lo, r), L_r = rlane(l_seq, r), t_r = rlane(trim_to, r);
      const bool rv_r = (rlane(flags, r) & bam::F_REVERSE) != 0;
      while (flr > 0) {
        bool real = false;
        if (lane < flr) { uint32_t c, q; view(s_r, q_r, L_r, rv_r, t_r, flr - 1 - lane, &c, &q); real = (c != 15); }
        unsigned long long bm = __ballot(real);
        if (bm) { flr -= (uint32_t)__builtin_ctzll(bm); break; }
        flr -= flr < 64 ? flr : 64;
      }
      if (lane == r) final_len = flr;
    }
  }

  PH(4)
  if constexpr (MODE == 0) {
  // ---- 5. family gates with ballots (process_group :1329-1422, process_subgroup :1454-1646) -----------------
  uint32_t s_total = n, s_cons = 0, s_filtered = 0, s_sec = 0, s_insuf = 0, s_zero = 0, s_orphan = 0, s_down = 0;
  const uint32_t n_sec = (uint32_t)__popcll(__ballot(act && excluded));
  const uint32_t n_reads = n - n_sec;
  if (n_sec) { s_filtered += n_sec; s_sec = n_sec; }
  bool go = n_reads > 0;
  if (go && n_reads < P.min_reads) { s_filtered += n_reads; s_insuf += n_reads; go = false; }
  uint32_t my_end = 255;
  if (cand) { if (!(flags & bam::F_PAIRED)) my_end = 0; else if (flags & bam::F_FIRST) my_end = 1; else if (flags & bam::F_LAST) my_end = 2; }
  bool ok[3] = {false, false, false};
  unsigned long long mem[3] = {0, 0, 0};
  uint32_t surv[3] = {0, 0, 0}, clen[3] = {0, 0, 0};
  bool need_defer = false;
  if (go) {
#pragma unroll
    for (uint32_t e = 0; e < 3; e++) {
      unsigned long long in_e = __ballot(my_end == e);
      uint32_t cnt = (uint32_t)__popcll(in_e);
      if (cnt == 0) continue;
      if (cnt < P.min_reads) { s_filtered += cnt; s_insuf += cnt; continue; }
      unsigned long long kept = __ballot(my_end == e && final_len > 0);
      uint32_t rem = (uint32_t)__popcll(kept), zero = cnt - rem;
      if (zero) { s_filtered += zero; s_zero += zero; }
      if (rem < P.min_reads) { if (rem) { s_filtered += rem; s_insuf += rem; } continue; }
      if (P.max_reads >= 0 && (long long)rem > P.max_reads) {
        // downsample_filtered_source_reads (:902-932): the max_reads lowest fgbio name ranks stay, ties in file order
        const int32_t rk = name_rank(W + lo + 32, name_len);
        const bool cand = (kept >> lane) & 1;
        uint32_t before = 0;
        for (unsigned long long m = kept; m; m &= m - 1) {
          const uint32_t j = (uint32_t)__builtin_ctzll(m);
          const int32_t rj = (int32_t)rlane((uint32_t)rk, j);
          before += (rj < rk || (rj == rk && j < lane)) ? 1u : 0u;
        }
        const unsigned long long stay = __ballot(cand && (long long)before < P.max_reads);
        const uint32_t dropped = rem - (uint32_t)__popcll(stay);
        s_filtered += dropped; s_down += dropped;
        kept = stay; rem = (uint32_t)__popcll(stay);
        if (rem < P.min_reads) { if (rem) { s_filtered += rem; s_insuf += rem; } continue; }
      }
      // consensus length = min_reads-th longest kept read
      bool mine = (kept >> lane) & 1;
      if (P.min_reads == 1) clen[e] = wave_max(mine ? final_len : 0);
      else {
        uint32_t ge = 0;
        for (unsigned long long m = kept; m; m &= m - 1) { uint32_t fj = rlane(final_len, (uint32_t)__builtin_ctzll(m)); ge += (fj >= final_len) ? 1 : 0; }
        clen[e] = wave_max((mine && ge >= P.min_reads) ? final_len : 0);
      }
      mem[e] = kept; surv[e] = rem; ok[e] = true;
    }
  }
  if (need_defer) { to_defer(); return; }
  // emitted ends in output order (Fragment, R1, R2), written with static indices only (a dynamically indexed local
  // array would live in scratch memory)
  uint32_t e_type[3], e_len[3], e_coloff[3];
  unsigned long long e_mem[3];
  const bool has_frag = ok[0], has_pair = ok[1] && ok[2];
  const uint32_t ne = (has_frag ? 1u : 0u) + (has_pair ? 2u : 0u);
  e_type[0] = has_frag ? 0u : 1u; e_len[0] = has_frag ? clen[0] : clen[1]; e_mem[0] = has_frag ? mem[0] : mem[1];
  e_type[1] = has_frag ? 1u : 2u; e_len[1] = has_frag ? clen[1] : clen[2]; e_mem[1] = has_frag ? mem[1] : mem[2];
  e_type[2] = 2u; e_len[2] = clen[2]; e_mem[2] = mem[2];
  if (has_frag) s_cons += 1;
  if (has_pair) s_cons += 2;
  else if (ok[1]) { s_filtered += surv[1]; s_orphan += surv[1]; }
  else if (ok[2]) { s_filtered += surv[2]; s_orphan += surv[2]; }
  uint32_t total_cols = 0;
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) if (k < ne) { e_coloff[k] = total_cols; total_cols += e_len[k]; }

  // UMIs carried by the kept reads of each end; unequal lengths → consensus_umis panics → general path
  uint32_t e_rx_cnt[3] = {0, 0, 0}, e_rx_len[3] = {0, 0, 0}, e_rx_first[3] = {0, 0, 0};
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    if (k >= ne) break;
    unsigned long long with = e_mem[k] & rxmask;
    uint32_t cnt = (uint32_t)__popcll(with);
    e_rx_cnt[k] = cnt;
    if (cnt) {
      uint32_t f = (uint32_t)__builtin_ctzll(with);
      e_rx_first[k] = f; e_rx_len[k] = rlane(rx_len, f);
      if (__any(((with >> lane) & 1) && rx_len != e_rx_len[k])) need_defer = true;
      if (e_rx_len[k] > FAST_RX_CAP) need_defer = true;
    }
  }
  if (need_defer) { to_defer(); return; }

  PH(5)
  // ---- 6. consensus columns: one lane per column, member reads walked in file order -------------------------
  // per-read descriptors packed for broadcast: d0 = seq_lo | qual_lo << 16 ; d1 = l_seq | final_len << 16 ; d2 = trim_to | rev << 16
  const uint32_t d0 = seq_lo | (qual_lo << 16), d1 = l_seq | (final_len << 16), d2 = trim_to | ((rev ? 1u : 0u) << 16);
  const DeviceTables* T = P.T;
  const uint64_t col_base = P.col_base[g];
  const uint32_t min_bq = P.min_input_bq;
  CallConst KC;   // wave-uniform, pinned into SGPRs (readfirstlane): no VGPRs held, nothing to spill
  KC.cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)sT.cap);
  KC.cap_threshold = uniform_f64(sT.cap_threshold); KC.half_cerr_at_cap = uniform_f64(sT.half_cerr_at_cap);
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    if (k >= ne) break;
    const unsigned long long members = e_mem[k];
    const uint32_t mc = (uint32_t)__popcll(members);
    for (uint32_t p = lane; p < e_len[k]; p += 64) {
      uint8_t ob, oq;
      uint32_t depth, err;
      if (mc == 1) {   // single-read consensus: LUT keyed by the unclamped quality (:1677-1708)
        uint32_t r = (uint32_t)__builtin_ctzll(members);
        uint32_t code, q;
        view(rlane(seq_lo, r), rlane(qual_lo, r), rlane(l_seq, r), (rlane(flags, r) & bam::F_REVERSE) != 0, rlane(trim_to, r), p, &code, &q);
        uint8_t adj = q < 94 ? T->single_input_quals[q] : 0;
        if (adj < P.min_cons_bq) { ob = 15; oq = FGX_MIN_PHRED; } else { ob = (uint8_t)code; oq = adj; }
        depth = code != 15 ? 1 : 0;
        err = 0;
      } else {
        ChainAcc acc;                           // two Kahan chains while the column shows one base (consensus_math.h)
        acc.reset();
        for (unsigned long long m = members; m; m &= m - 1) {
          const uint32_t r = (uint32_t)__builtin_ctzll(m);
          const uint32_t x0 = rlane(d0, r), x1 = rlane(d1, r), x2 = rlane(d2, r);
          bool valid = p < (x1 >> 16);
          const bool rv = (x2 >> 16) != 0;
          const uint32_t idx = valid ? (rv ? (x1 & 0xFFFF) - 1 - p : p) : 0;
          const uint32_t bb = W[(x0 & 0xFFFF) + (idx >> 1)];
          const uint32_t c = (bb >> ((~idx & 1) << 2)) & 15;
          const uint32_t q = W[(x0 >> 16) + idx];
          // the code in read orientation (complement = bit reversal of the 4-bit code); only the one-hot codes A C G T count
          const uint32_t co = rv ? __builtin_bitreverse32(c) >> 28 : c;
          const bool masked = p < (x2 & 0xFFFF) && q < min_bq;
          valid = valid && __popc(co) == 1 && !masked;
          const uint32_t qq = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
          const double2 pr = *(const double2*)&sPair[qq][0];
          acc.add(valid, co, pr.x, pr.y);
        }
        double ll[4];
        uint32_t obs[4];
        acc.finish(ll, obs);
        int bi;
        uint8_t q;
        bool resolved = column_call_fast_lds(sT, KC, ll, obs, &bi, &q);
        depth = obs[0] + obs[1] + obs[2] + obs[3];
        uint64_t o = col_base + e_coloff[k] + p;
        uint32_t d16 = depth < 32767u ? depth : 32767u;
        P.col_depth[o] = (uint16_t)d16;
        push_full(!resolved, o, ll, obs);
        if (!resolved) continue;           // code / qual / errors arrive from k_call_full
        err = depth - (bi == 0 ? obs[0] : bi == 1 ? obs[1] : bi == 2 ? obs[2] : bi == 3 ? obs[3] : 0u);
        uint8_t code = bi >= 0 ? (uint8_t)(1u << bi) : 15;
        if (depth < P.min_reads) { ob = 15; oq = 0; }
        else if (q < P.min_cons_bq) { ob = 15; oq = FGX_MIN_PHRED; }
        else { ob = code; oq = q; }
        uint32_t e16 = err < 32767u ? err : 32767u;
        P.col_code[o] = ob; P.col_qual[o] = oq; P.col_err[o] = (uint16_t)e16;
        continue;
      }
      uint32_t d16 = depth < 32767u ? depth : 32767u, e16 = err < 32767u ? err : 32767u;
      uint64_t o = col_base + e_coloff[k] + p;
      P.col_code[o] = ob; P.col_qual[o] = oq; P.col_depth[o] = (uint16_t)d16; P.col_err[o] = (uint16_t)e16;
    }
  }

  PH(6)
  // ---- 7. consensus UMI per end (simple_umi.rs:46-117) ---------------------------------------------------------
  const DeviceTables* TU = P.TU;
  char my_rx[3] = {0, 0, 0};      // lane i holds character i of each end's consensus UMI
  bool rx_bad = false;
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    if (k >= ne) break;
    if (e_rx_cnt[k] == 0) continue;                     // uniform
    const bool mychar = lane < e_rx_len[k];
    if (e_rx_cnt[k] == 1) { if (mychar) my_rx[k] = (char)W[rlane(rx_lo, e_rx_first[k]) + lane]; continue; }
    // Byte-identical UMIs (the usual case) need no arithmetic: n >= 2 agreeing observations of base b leave b the
    // strict maximum of the four likelihoods whatever n is, so the call is the upper-cased base; a column of N/n has
    // no observations and calls 'N'; a non-DNA character shared by every string is kept (simple_umi.rs:69-103).
    {
      const uint32_t f_lo = rlane(rx_lo, e_rx_first[k]), ulen = e_rx_len[k];
      bool differs = false;
      if ((e_mem[k] & rxmask) >> lane & 1) {
        for (uint32_t i = 0; i < ulen; i += 8) {
          unsigned long long a = ld64u(W, rx_lo + i), b = ld64u(W, f_lo + i);
          if (i + 8 > ulen) { unsigned long long mk = (1ULL << (8 * (ulen - i))) - 1; a &= mk; b &= mk; }
          differs |= a != b;
        }
      }
      if (!__any(differs)) {
        if (mychar) {
          const uint8_t ch = W[f_lo + lane];
          const int bl = bam::ascii_to_lane(ch);
          my_rx[k] = bl != 255 ? "ACGT"[bl] : (ch == 'N' || ch == 'n') ? 'N' : (char)ch;
        }
        continue;
      }
    }
    // UMIs that differ (sequencing errors inside a family): per-character likelihood columns
    ColumnAcc acc;
    acc.reset();
    uint32_t non_dna = 0, seen = 0;
    uint8_t fc = 0;
    bool mixed = false;
    const double uc = TU->t.correct[20], ue = TU->t.error_per_alt[20];
    for (unsigned long long m = e_mem[k] & rxmask; m; m &= m - 1) {
      uint32_t r = (uint32_t)__builtin_ctzll(m);
      uint8_t ch = mychar ? W[rlane(rx_lo, r) + lane] : (uint8_t)'A';
      if (seen == 0) fc = ch;
      seen++;
      uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
      bool dna = up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'N';
      if (dna) { int bl = bam::ascii_to_lane(ch); if (bl != 255) acc.add(bl, uc, ue); }
      else { non_dna++; if (ch != fc) mixed = true; }
    }
    bool need_full = false;
    if (!mychar) { /* lanes past the UMI length only take part in the ballots below */ }
    else if (non_dna == 0) {
      int bi; uint8_t q;
      CallConst KU;
      KU.cap = TU->t.cap; KU.cap_threshold = TU->t.cap_threshold; KU.half_cerr_at_cap = TU->t.half_cerr_at_cap;
      bool resolved = column_call_fast_lds(TU->t, KU, acc.s, acc.obs, &bi, &q);
      my_rx[k] = bi >= 0 ? "ACGT"[bi] : 'N';
      need_full = !resolved;
    }
    else if (non_dna == seen && !mixed) my_rx[k] = (char)fc;
    else rx_bad = true;
    if (!mychar) { need_full = false; rx_bad = false; }
    push_full(need_full, (1ull << 63) | ((uint64_t)(slot0 + e_type[k]) << 8) | lane, acc.s, acc.obs);
  }
  if (__any(rx_bad) || __any(list_overflow)) { to_defer(); return; }

  PH(7)
  // ---- 8. descriptors + stats ---------------------------------------------------------------------------------------
#pragma unroll
  for (uint32_t k = 0; k < 3; k++) {
    if (k >= ne) break;
    uint32_t Lc = e_len[k];
    uint32_t slot = slot0 + e_type[k];
    EndDesc* D = &P.ends[slot];
    if (lane < e_rx_len[k] && e_rx_cnt[k]) D->rx[lane] = my_rx[k];
    uint32_t fk = (uint32_t)__builtin_ctzll(e_mem[k]);     // first retained read (cell barcode source)
    uint32_t fk_has_cb = (uint32_t)((cbmask >> fk) & 1), fk_cb_lo = rlane(cb_lo, fk), fk_cb_len = rlane(cb_len, fk), fk_lo = rlane(lo, fk);
    const unsigned long long off0 = (unsigned long long)rlane((uint32_t)off, 0) | ((unsigned long long)rlane((uint32_t)(off >> 32), 0) << 32);
    const unsigned long long off_fk = (unsigned long long)rlane((uint32_t)off, fk) | ((unsigned long long)rlane((uint32_t)(off >> 32), fk) << 32);
    if (lane == 0) {
      D->col_off = col_base + e_coloff[k];
      D->cons_len = Lc;
      D->first_off = off0; D->kept_off = off_fk;
      D->type = (uint8_t)e_type[k];
      D->mi_off = (uint16_t)(mi0_lo - lo); D->mi_len = (uint8_t)mi0_len;          // lane 0: lo = record 0
      bool hcb = P.cell0 && fk_has_cb;
      D->has_cb = hcb ? 1 : 0; D->cb_off = (uint16_t)(fk_cb_lo - fk_lo); D->cb_len = (uint8_t)fk_cb_len;
      D->has_rx = e_rx_cnt[k] > 0; D->rx_len = (uint8_t)e_rx_len[k];
      uint32_t nm = P.prefix_len + 1 + mi0_len;
      // cD / cM <= member count <= 64 → one-byte int tags (k_emit derives the values from the arrays)
      uint32_t size = 32 + nm + 1 + (Lc + 1) / 2 + Lc + (3 + P.rg_len + 1) + 4 + 4 + 7 +
                      (P.per_base_tags ? 2 * (8 + 2 * Lc) : 0) + (3 + mi0_len + 1) + (hcb ? 3 + fk_cb_len + 1 : 0) +
                      (e_rx_cnt[k] ? 3 + e_rx_len[k] + 1 : 0);
      D->rec_size = size;
      D->valid = 1;
      P.rec_sizes[slot] = (uint64_t)size + 4;
    }
  }
  ov_bases = wave_sum(ov_bases); ov_agree = wave_sum(ov_agree); ov_dis = wave_sum(ov_dis); ov_corr = wave_sum(ov_corr);
  if (lane == 0) {
    atomicAdd(&st[0], (unsigned long long)s_total);
    if (s_cons) atomicAdd(&st[1], (unsigned long long)s_cons);
    if (s_filtered) atomicAdd(&st[2], (unsigned long long)s_filtered);
    if (s_sec) atomicAdd(&st[3 + FGX_REJ_SECONDARY_OR_SUPPLEMENTARY], (unsigned long long)s_sec);
    if (s_insuf) atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_READS], (unsigned long long)s_insuf);
    if (s_zero) atomicAdd(&st[3 + FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING], (unsigned long long)s_zero);
    if (s_orphan) atomicAdd(&st[3 + FGX_REJ_ORPHAN_CONSENSUS], (unsigned long long)s_orphan);
    if (s_down) atomicAdd(&st[3 + FGX_REJ_DOWNSAMPLED], (unsigned long long)s_down);
    if (ov_bases) atomicAdd(&st[24], (unsigned long long)ov_bases);
    if (ov_agree) atomicAdd(&st[25], (unsigned long long)ov_agree);
    if (ov_dis) atomicAdd(&st[26], (unsigned long long)ov_dis);
    if (ov_corr) atomicAdd(&st[27], (unsigned long long)ov_corr);
  }
  PH(8)
  } else if constexpr (MODE == 1) {
  // =====================================================================================================================
  // MODE 1 — duplex.  Mirrors DuplexConsensusCaller::consensus_reads / process_group (duplex_caller.rs:2545-2624,
  // 1944-2540) for the molecules the device can decide exactly; anything else is deferred to duplex_host.cpp.
  // =====================================================================================================================
  // ---- 5D. fragments out, strand partition, min-reads and collision gates ------------------------------------------------
  const bool paired = act && (flags & bam::F_PAIRED);
  const unsigned long long pmask = __ballot(paired);
  const uint32_t n_frag = n - (uint32_t)__popcll(pmask);
  uint32_t s_cons = 0, s_filtered = n_frag, rj_insuf = 0, rj_coll = 0, rj_zero = 0;
  auto flush_stats = [&]() {
    ov_bases = wave_sum(ov_bases); ov_agree = wave_sum(ov_agree); ov_dis = wave_sum(ov_dis); ov_corr = wave_sum(ov_corr);
    if (lane == 0) {
      atomicAdd(&st[0], (unsigned long long)n);
      if (s_cons) atomicAdd(&st[1], (unsigned long long)s_cons);
      if (s_filtered) atomicAdd(&st[2], (unsigned long long)s_filtered);
      if (n_frag) atomicAdd(&st[3 + FGX_REJ_FRAGMENT_READ], (unsigned long long)n_frag);
      if (rj_insuf) atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_READS], (unsigned long long)rj_insuf);
      if (rj_coll) atomicAdd(&st[3 + FGX_REJ_POTENTIAL_COLLISION], (unsigned long long)rj_coll);
      if (rj_zero) atomicAdd(&st[3 + FGX_REJ_ZERO_LENGTH_AFTER_TRIMMING], (unsigned long long)rj_zero);
      if (ov_bases) atomicAdd(&st[24], (unsigned long long)ov_bases);
      if (ov_agree) atomicAdd(&st[25], (unsigned long long)ov_agree);
      if (ov_dis) atomicAdd(&st[26], (unsigned long long)ov_dis);
      if (ov_corr) atomicAdd(&st[27], (unsigned long long)ov_corr);
    }
  };
  if (!pmask) { flush_stats(); return; }                                  // nothing but fragments: no MI to work from
  const bool is_r1 = paired && (flags & bam::F_FIRST), is_r2 = paired && (flags & bam::F_LAST);
  if (__any(paired && is_r1 == is_r2)) { to_defer(); return; }            // FIRST and LAST both set or both clear: general path
  const unsigned long long amask = __ballot(paired && strand == 1), bmask = __ballot(paired && strand == 2);
  const unsigned long long r1m = __ballot(is_r1), r2m = __ballot(is_r2);
  const uint32_t n_ab = (uint32_t)__popcll(amask | bmask);
  auto min_ok = [&](uint32_t x, uint32_t y) {                             // has_minimum_number_of_reads :824-862
    const uint32_t xy = x > y ? x : y, yx = x > y ? y : x;
    return P.dmin_total <= xy + yx && P.dmin_xy <= xy && P.dmin_yx <= yx;
  };
  if (!min_ok((uint32_t)__popcll(amask & r1m), (uint32_t)__popcll(bmask & r1m))) { rj_insuf = n_ab; s_filtered += n_ab; flush_stats(); return; }
  if (amask && bmask) {                                                   // /A R1 with /B R2 (and /A R2 with /B R1) must share a strand
    const unsigned long long revm = __ballot(paired && rev);
    const unsigned long long g1 = (amask & r1m) | (bmask & r2m), g2 = (amask & r2m) | (bmask & r1m);
    const bool same1 = (g1 & revm) == 0 || (g1 & revm) == g1, same2 = (g2 & revm) == 0 || (g2 & revm) == g2;
    if (!same1 || !same2) { rj_coll = n_ab; s_filtered += n_ab; flush_stats(); return; }
  }
  const unsigned long long nzm = __ballot(paired && final_len > 0);
  const uint32_t n_zero = (uint32_t)__popcll((amask | bmask) & ~nzm);     // ZeroLengthAfterTrimming (single-strand caller's stats)
  // the four single-strand read sets: 0 = AB-R1, 1 = AB-R2, 2 = BA-R1, 3 = BA-R2
  unsigned long long em[4];
  em[0] = uniform_u64(amask & r1m & nzm); em[1] = uniform_u64(amask & r2m & nzm); em[2] = uniform_u64(bmask & r1m & nzm); em[3] = uniform_u64(bmask & r2m & nzm);
  uint32_t elen[4], eoff[4];
  {
    bool cap_bites = false;
    uint32_t tot = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (P.dmax_reads >= 0 && (long long)__popcll(em[k]) > P.dmax_reads) cap_bites = true;
      elen[k] = em[k] ? wave_max(((em[k] >> lane) & 1) ? final_len : 0) : 0;
      eoff[k] = tot; tot += elen[k];
    }
    if (cap_bites) { to_defer(); return; }                                 // --max-reads-per-strand that bites: name-rank downsampling on the host
  }
  const uint32_t plen1 = (em[0] && em[3]) ? (elen[0] < elen[3] ? elen[0] : elen[3]) : 0;    // R1 duplex: AB-R1 with BA-R2
  const uint32_t plen2 = (em[1] && em[2]) ? (elen[1] < elen[2] ? elen[1] : elen[2]) : 0;    // R2 duplex: AB-R2 with BA-R1

  PH(5)
  // ---- 6D. the four single-strand column sets (ss caller: min_reads 1, min consensus base quality 2) ----------------------
  const uint32_t d0 = seq_lo | (qual_lo << 16), d1 = l_seq | (final_len << 16), d2 = trim_to | ((rev ? 1u : 0u) << 16);
  const DeviceTables* T = P.T;
  const uint64_t col_base = P.col_base[g];
  const uint32_t min_bq = P.min_input_bq;
  CallConst KC;
  KC.cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)sT.cap);
  KC.cap_threshold = uniform_f64(sT.cap_threshold); KC.half_cerr_at_cap = uniform_f64(sT.half_cerr_at_cap);
  uint32_t dm_tr[4] = {0, 0, 0, 0}, dm_fu[4] = {0, 0, 0, 0};     // max depth over the paired (truncated) span / the whole strand
  bool odd_base = false;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const unsigned long long members = em[k];
    if (!members) continue;
    const uint32_t mc = (uint32_t)__popcll(members);
    const uint32_t plen = (k == 0 || k == 3) ? plen1 : plen2;
    // A REGULAR set — every member on one strand (reverse: one read length), untrimmed: the usual set — takes k_simplex_wave2's column
    // step (chain_observe.inc: ~20 vector instructions per observation instead of ~50; the lane's index into a member is the same for
    // every member, a member's final length is a scalar lane mask, the codes stay as stored and the chains' bases are complemented once
    // at the end).  Anything else: the general loop below.  (Final lengths DO differ inside a set: a read that ends in a base below
    // --min-input-base-quality loses it — 44 % of the benchmark's molecules have such a set.)
    bool regular = false;
    uint32_t Lrev = 0;
    bool set_rev = false;
    if (mc > 1) {
      const bool mine = (members >> lane) & 1;
      const unsigned long long rm = members & __ballot(rev);
      set_rev = rm != 0;
      if (rm) Lrev = rlane(l_seq, (uint32_t)__builtin_ctzll(rm));
      const bool oddm = mine && (trim_to != l_seq || (rm != 0 && l_seq != Lrev));
      regular = !__any(oddm) && (rm == 0 || rm == members);
    }
#ifdef FGX_DUPLEX_REQUIRE_REGULAR   /* measurement builds: a molecule with a set that is not regular is deferred — the bench line's deferred count says how many */
    if (mc > 1 && !regular) { to_defer(); return; }
#endif
    if (regular) {
      const uint8_t* const pairs = (const uint8_t*)&sPair[0][0];
      for (uint32_t p0 = 0; p0 < elen[k]; p0 += 64) {
        const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p0) + lane;
        const bool incol = p < elen[k];
        const uint32_t ix = incol ? (set_rev ? Lrev - 1 - p : p) : 0u;
        const uint32_t iq = ix, is = ix >> 1, ish = (~ix & 1) << 2;
        double s1 = 0.0, c1 = 0.0, sR = 0.0, cR = 0.0, s2, c2, s3, c3;
        FGX_UNDEF4(s2, c2, s3, c3);   // (chains 2 / 3: written when opened)
        uint32_t allow = incol ? 0x116u : 0u, b1 = 0, b2 = 0, b3 = 0, n1 = 0, n2 = 0, n3 = 0, nR = 0;
        for (unsigned long long m = members; m;) {
          const uint32_t r = (uint32_t)__builtin_ctzll(m);
          m &= m - 1;
          const uint32_t x0 = rlane(d0, r), x1 = rlane(d1, r);
          const bool in = __builtin_amdgcn_inverse_ballot_w64(fw_lanes_below(x1 >> 16, (uint32_t)__builtin_amdgcn_readfirstlane((int)p0)));   // inside this member's final length
          const uint32_t q = W[iq + (x0 >> 16)];
          const uint32_t c = __builtin_amdgcn_ubfe((uint32_t)W[is + (x0 & 0xFFFF)], ish, 4u);
          W2_OBSERVE(0, in, q, c, *(const double2*)(pairs + (((q) < 93u ? (q) : 93u) << 4)));
        }
        if (allow != 0) b1 = (uint32_t)__builtin_ctz(allow);
        if (set_rev) { b1 = __builtin_bitreverse32(b1) >> 28; b2 = __builtin_bitreverse32(b2) >> 28; b3 = __builtin_bitreverse32(b3) >> 28; }
        double ll[4];
        uint32_t obs[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
          const uint32_t code = 1u << i;
          ll[i] = code == b1 ? s1 : code == b2 ? s2 : code == b3 ? s3 : sR;
          obs[i] = code == b1 ? n1 : code == b2 ? n2 : code == b3 ? n3 : nR;
        }
        const uint64_t o = col_base + eoff[k] + p;
        int bi = -1;
        uint8_t q = 0;
        const bool resolved = incol ? column_call_fast_lds(sT, KC, ll, obs, &bi, &q) : true;
        const uint32_t depth = obs[0] + obs[1] + obs[2] + obs[3];
        push_full(incol && !resolved, o, ll, obs);
        if (incol) {
          if (resolved) {
            const uint32_t err = depth - (bi == 0 ? obs[0] : bi == 1 ? obs[1] : bi == 2 ? obs[2] : bi == 3 ? obs[3] : 0u);
            uint8_t ob, oq;
            if (depth < 1) { ob = 15; oq = 0; }
            else if (q < FGX_MIN_PHRED) { ob = 15; oq = FGX_MIN_PHRED; }
            else { ob = bi >= 0 ? (uint8_t)(1u << bi) : (uint8_t)15; oq = q; }
            P.col_code[o] = ob; P.col_qual[o] = oq; P.col_err[o] = (uint16_t)err;
          }
          P.col_obs[o] = obs[0] | (obs[1] << 8) | (obs[2] << 16) | (obs[3] << 24);
          dm_fu[k] = depth > dm_fu[k] ? depth : dm_fu[k];
          if (p < plen) dm_tr[k] = depth > dm_tr[k] ? depth : dm_tr[k];
        }
      }
      continue;
    }
    // (a single-read set: the read's fields once, by every lane, ahead of the column loop — its trip count differs from lane to lane)
    uint32_t one_seq = 0, one_qual = 0, one_lseq = 0, one_flags = 0, one_trim = 0;
    if (mc == 1) {
      const uint32_t r = (uint32_t)__builtin_ctzll(members);
      one_seq = rlane(seq_lo, r); one_qual = rlane(qual_lo, r); one_lseq = rlane(l_seq, r); one_flags = rlane(flags, r); one_trim = rlane(trim_to, r);
    }
    for (uint32_t p = lane; p < elen[k]; p += 64) {
      const uint64_t o = col_base + eoff[k] + p;
      uint32_t depth, obs[4];
      if (mc == 1) {   // single-read consensus: LUT keyed by the unclamped quality (vanilla_caller.rs:1677-1708)
        uint32_t code, q;
        view(one_seq, one_qual, one_lseq, (one_flags & bam::F_REVERSE) != 0, one_trim, p, &code, &q);
        const uint8_t adj = q < 94 ? T->single_input_quals[q] : 0;
        const bool low = adj < FGX_MIN_PHRED;
        P.col_code[o] = low ? (uint8_t)15 : (uint8_t)code; P.col_qual[o] = low ? (uint8_t)FGX_MIN_PHRED : adj; P.col_err[o] = 0;
        obs[0] = code == 1; obs[1] = code == 2; obs[2] = code == 4; obs[3] = code == 8;
        depth = code != 15 ? 1u : 0u;
        if (depth && !(obs[0] | obs[1] | obs[2] | obs[3])) odd_base = true;     // IUPAC code in a lone read: general path
      } else {
        ChainAcc acc;                           // two Kahan chains while the column shows one base (consensus_math.h)
        acc.reset();
        for (unsigned long long m = members; m; m &= m - 1) {
          const uint32_t r = (uint32_t)__builtin_ctzll(m);
          const uint32_t x0 = rlane(d0, r), x1 = rlane(d1, r), x2 = rlane(d2, r);
          bool valid = p < (x1 >> 16);
          const bool rv = (x2 >> 16) != 0;
          const uint32_t idx = valid ? (rv ? (x1 & 0xFFFF) - 1 - p : p) : 0;
          const uint32_t bb = W[(x0 & 0xFFFF) + (idx >> 1)];
          const uint32_t c = (bb >> ((~idx & 1) << 2)) & 15;
          const uint32_t q = W[(x0 >> 16) + idx];
          const uint32_t co = rv ? __builtin_bitreverse32(c) >> 28 : c;     // read orientation: complement = bit reversal of the code
          const bool masked = p < (x2 & 0xFFFF) && q < min_bq;
          valid = valid && __popc(co) == 1 && !masked;                      // only the one-hot codes A C G T are observations
          const uint32_t qq = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
          const double2 pr = *(const double2*)&sPair[qq][0];
          acc.add(valid, co, pr.x, pr.y);
        }
        double ll[4];
        acc.finish(ll, obs);
        int bi;
        uint8_t q;
        const bool resolved = column_call_fast_lds(sT, KC, ll, obs, &bi, &q);
        depth = obs[0] + obs[1] + obs[2] + obs[3];
        push_full(!resolved, o, ll, obs);
        if (resolved) {
          const uint32_t err = depth - (bi == 0 ? obs[0] : bi == 1 ? obs[1] : bi == 2 ? obs[2] : bi == 3 ? obs[3] : 0u);
          uint8_t ob, oq;
          if (depth < 1) { ob = 15; oq = 0; }
          else if (q < FGX_MIN_PHRED) { ob = 15; oq = FGX_MIN_PHRED; }
          else { ob = bi >= 0 ? (uint8_t)(1u << bi) : (uint8_t)15; oq = q; }
          P.col_code[o] = ob; P.col_qual[o] = oq; P.col_err[o] = (uint16_t)err;
        }
      }
      P.col_obs[o] = obs[0] | (obs[1] << 8) | (obs[2] << 16) | (obs[3] << 24);
      dm_fu[k] = depth > dm_fu[k] ? depth : dm_fu[k];
      if (p < plen) dm_tr[k] = depth > dm_tr[k] ? depth : dm_tr[k];
    }
  }
  if (__any(odd_base)) { to_defer(); return; }
#pragma unroll
  for (int k = 0; k < 4; k++) { dm_fu[k] = wave_max(dm_fu[k]); dm_tr[k] = wave_max(dm_tr[k]); }

  PH(6)
  // ---- 7D. which records come out (duplex_consensus :931-1108, process_group :2400-2540) ------------------------------------
  // A duplex record takes strand `sa` (and `sb` when both strands are covered); a strand alone is passed through.
  struct Out { bool ok; bool has_ba; uint32_t sa, sb, len; };
  auto duplex_of = [&](int ka, int kb, uint32_t plen) {          // ka: AB-side set or -1, kb: BA-side set or -1
    Out r; r.ok = false; r.has_ba = false; r.sa = 0; r.sb = 0; r.len = 0;
    const bool have_a = ka >= 0, have_b = kb >= 0;
    const uint32_t da = have_a ? (have_b ? dm_tr[ka < 0 ? 0 : ka] : dm_fu[ka < 0 ? 0 : ka]) : 0;
    const uint32_t db = have_b ? (have_a ? dm_tr[kb < 0 ? 0 : kb] : dm_fu[kb < 0 ? 0 : kb]) : 0;
    const bool cov_a = have_a && da > 0, cov_b = have_b && db > 0;
    if (!cov_a && !cov_b) return r;
    if (cov_a && cov_b) { r.ok = min_ok(da, db); r.has_ba = true; r.sa = (uint32_t)ka; r.sb = (uint32_t)kb; r.len = plen; return r; }
    const int ks = cov_a ? ka : kb;                                 // one covered strand: copied whole, depth over its full length
    const uint32_t ds = dm_fu[ks < 0 ? 0 : ks];
    r.ok = min_ok(ds, 0); r.sa = (uint32_t)ks; r.sb = (uint32_t)ks; r.len = elen[ks < 0 ? 0 : ks];
    return r;
  };
  const bool h1 = em[0] != 0, h2 = em[1] != 0, h3 = em[2] != 0, h4 = em[3] != 0;
  Out o1, o2;
  o1.ok = o2.ok = false; o1.has_ba = o2.has_ba = false; o1.sa = o1.sb = o2.sa = o2.sb = 0; o1.len = o2.len = 0;
  if (h1 && h2 && h3 && h4) { o1 = duplex_of(0, 3, plen1); o2 = duplex_of(1, 2, plen2); }
  else if (h1 && h2 && !h3 && !h4) { if (P.dmin_yx == 0) { o1 = duplex_of(0, -1, 0); o2 = duplex_of(1, -1, 0); } }
  else if (!h1 && !h2 && h3 && h4) { if (P.dmin_yx == 0) { o1 = duplex_of(-1, 2, 0); o2 = duplex_of(-1, 3, 0); } }
  const bool emitted = o1.ok && o2.ok;
  // statistics (duplex_caller.rs:2587-2610 with the re-attribution of :1894-1926)
  if (emitted) { s_cons = 2; rj_zero = n_zero; s_filtered += n_zero; }
  else { rj_insuf = n_ab - n_zero; rj_zero = n_zero; s_filtered += n_ab; }

  // ---- 8D. consensus UMI and descriptor of each record ----------------------------------------------------------------------
  if (emitted) {
    const DeviceTables* TU = P.TU;
    const unsigned long long rxm = rxmask;
    // '-' structure of this lane's RX (flipping a paired UMI swaps the halves around the single dash)
    uint32_t n_dash = 0, dash_pos = 0;
    if (act && has_rx) for (uint32_t i = 0; i < rx_len; i++) if (W[rx_lo + i] == '-') { if (!n_dash) dash_pos = i; n_dash++; }
    const uint32_t fp = (uint32_t)__builtin_ctzll(pmask);                     // first paired record: MI → name and MI tag
    const uint32_t fp_mi_lo = rlane(mi_lo, fp), fp_mi_len = rlane(mi_len, fp) - 2, fp_lo = rlane(lo, fp);
    const uint32_t fc = amask ? (uint32_t)__builtin_ctzll(amask) : (uint32_t)__builtin_ctzll(bmask);   // cell barcode source
    const bool fc_has_cb = P.cell0 && ((cbmask >> fc) & 1);
    const uint32_t fc_cb_lo = rlane(cb_lo, fc), fc_cb_len = rlane(cb_len, fc), fc_lo = rlane(lo, fc);
    bool rx_fail = false;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      const Out& O = t == 0 ? o1 : o2;
      // source reads of this record's UMIs, A-side first (file order inside a side); a read is flipped when its FIRST flag
      // differs from the record's (duplex_read_into :1336-1368)
      unsigned long long sa_m = 0, sb_m = 0;
      if (h1 && h2 && h3 && h4) { sa_m = t == 0 ? em[0] : em[1]; sb_m = t == 0 ? em[3] : em[2]; }
      else if (h1) { sa_m = t == 0 ? em[0] : em[1]; }
      else { sb_m = t == 0 ? em[2] : em[3]; }
      sa_m &= rxm; sb_m &= rxm;
      const bool rec_first = t == 0;
      const uint32_t cnt = (uint32_t)__popcll(sa_m | sb_m);
      uint32_t ulen = 0;
      char my_ch = 0;
      if (cnt) {
        const uint32_t f = sa_m ? (uint32_t)__builtin_ctzll(sa_m) : (uint32_t)__builtin_ctzll(sb_m);
        ulen = rlane(rx_len, f);
        const bool in_set = ((sa_m | sb_m) >> lane) & 1;
        const bool flip_me = in_set && (((flags & bam::F_FIRST) != 0) != rec_first);
        if (__any(in_set && rx_len != ulen) || ulen > FAST_RX_CAP || __any(flip_me && n_dash > 1)) rx_fail = true;
        else {
          // character j of read r's UMI as it enters the consensus
          auto chr = [&](uint32_t r, uint32_t j) -> uint8_t {
            const uint32_t rlo = rlane(rx_lo, r), nd = rlane(n_dash, r), dp = rlane(dash_pos, r);
            const bool fl = (((rlane(flags, r) & bam::F_FIRST) != 0) != rec_first) && nd == 1;
            uint32_t src = j;
            if (fl) { const uint32_t ql = ulen - dp - 1; src = j < ql ? dp + 1 + j : j == ql ? dp : j - ql - 1; }
            return j < ulen ? W[rlo + src] : (uint8_t)'A';
          };
          const bool mychar = lane < ulen;
          const uint8_t c0 = chr(f, lane);
          bool differs = false;
          for (unsigned long long m = sa_m | sb_m; m; m &= m - 1) differs |= chr((uint32_t)__builtin_ctzll(m), lane) != c0;
          if (cnt == 1 || !__any(mychar && differs)) {
            const int bl = bam::ascii_to_lane(c0);
            my_ch = cnt == 1 ? (char)c0 : bl != 255 ? "ACGT"[bl] : (c0 == 'N' || c0 == 'n') ? 'N' : (char)c0;
          } else {
            ColumnAcc acc;
            acc.reset();
            uint32_t non_dna = 0, seen = 0;
            uint8_t fch = 0;
            bool mixed = false;
            const double uc = TU->t.correct[20], ue = TU->t.error_per_alt[20];
#pragma unroll
            for (int side = 0; side < 2; side++)
              for (unsigned long long m = side == 0 ? sa_m : sb_m; m; m &= m - 1) {
                const uint8_t ch = chr((uint32_t)__builtin_ctzll(m), lane);
                if (seen == 0) fch = ch;
                seen++;
                const uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
                const bool dna = up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'N';
                if (dna) { const int bl = bam::ascii_to_lane(ch); if (bl != 255) acc.add(bl, uc, ue); }
                else { non_dna++; if (ch != fch) mixed = true; }
              }
            bool need_full = false, bad_col = false;
            if (non_dna == 0) {
              CallConst KU;
              KU.cap = TU->t.cap; KU.cap_threshold = TU->t.cap_threshold; KU.half_cerr_at_cap = TU->t.half_cerr_at_cap;
              int bi; uint8_t q;
              const bool resolved = column_call_fast_lds(TU->t, KU, acc.s, acc.obs, &bi, &q);
              my_ch = bi >= 0 ? "ACGT"[bi] : 'N';
              need_full = !resolved;
            } else if (non_dna == seen && !mixed) my_ch = (char)fch;
            else bad_col = true;
            if (!mychar) { need_full = false; bad_col = false; }
            if (__any(bad_col)) rx_fail = true;
            push_full(need_full, (1ull << 63) | ((uint64_t)(slot0 + 1 + t) << 8) | lane, acc.s, acc.obs);
          }
        }
      }
      const uint32_t slot = slot0 + 1 + t;
      DuplexDesc* D = &P.dends[slot];
      if (lane < ulen && cnt) D->rx[lane] = my_ch;
      if (lane == 0) {
        const uint32_t L = O.len;
        D->a_off = col_base + eoff[O.sa]; D->b_off = col_base + eoff[O.sb];
        D->len = L; D->first_rec = r0 + fp; D->cb_rec = r0 + fc;
        D->mi_off = (uint16_t)(fp_mi_lo - fp_lo); D->mi_len = (uint8_t)fp_mi_len;
        D->has_cb = fc_has_cb ? 1 : 0; D->cb_off = (uint16_t)(fc_cb_lo - fc_lo); D->cb_len = (uint8_t)fc_cb_len;
        D->has_rx = cnt > 0; D->rx_len = (uint8_t)ulen; D->type = (uint8_t)(1 + t); D->has_ba = O.has_ba ? 1 : 0;
        const uint32_t nm = P.prefix_len + 1 + fp_mi_len;
        const uint32_t per_strand = P.per_base_tags ? 2 * (3 + L + 1) + 2 * (8 + 2 * L) : 0;     // ac/aq strings + ad/ae arrays
        // every int tag holds a depth <= 128 here (<= 64 records per wavefront): 4 bytes each
        const uint32_t size = 32 + nm + 1 + (L + 1) / 2 + L + (3 + fp_mi_len + 1) + (fc_has_cb ? 3 + fc_cb_len + 1 : 0) + (3 + P.rg_len + 1) +
                              3 * (4 + 7 + 4) + per_strand + (O.has_ba ? per_strand : 0) + (cnt ? 3 + ulen + 1 : 0);
        D->rec_size = size;
        D->valid = 1;
        P.rec_sizes[slot] = (uint64_t)size + 4;
      }
    }
    if (rx_fail || __any(list_overflow)) {            // undo: the general path owns this molecule
      if (lane < 2) { P.dends[slot0 + 1 + lane].valid = 0; P.rec_sizes[slot0 + 1 + lane] = 0; }
      to_defer();
      return;
    }
  } else if (__any(list_overflow)) { to_defer(); return; }
  flush_stats();
  PH(8)
  } else {
  // =====================================================================================================================
  // MODE 2 — CODEC.  Mirrors CodecConsensusCaller::consensus_reads_raw (codec_caller.rs:625-1004) for molecules whose
  // records are all paired primary reads with a single M op, mates adjacent; everything else goes to codec_host.cpp.
  // The duplex-disagreement thresholds are at their defaults on this path (the host checks), so no molecule is rejected
  // after the strand combine and every record size is known here.
  // =====================================================================================================================
  // ---- 5C. templates: the partner of each record is the one other record with its read name ---------------------------------
  uint32_t s_filtered = 0, rj_notfr = 0, rj_insuf = 0, rj_overlap = 0, rj_indel = 0, rj_clipfail = 0, s_cons = 0;
  auto flush_stats = [&]() {
    if (lane == 0) {
      atomicAdd(&st[0], (unsigned long long)n);
      if (s_cons) atomicAdd(&st[1], (unsigned long long)s_cons);
      if (s_filtered) atomicAdd(&st[2], (unsigned long long)s_filtered);
      if (rj_notfr) atomicAdd(&st[3 + FGX_REJ_NOT_PRIMARY_FR_PAIR], (unsigned long long)rj_notfr);
      if (rj_insuf) atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_READS], (unsigned long long)rj_insuf);
      if (rj_overlap) atomicAdd(&st[3 + FGX_REJ_INSUFFICIENT_OVERLAP], (unsigned long long)rj_overlap);
      if (rj_indel) atomicAdd(&st[3 + FGX_REJ_INDEL_ERROR_BETWEEN_STRANDS], (unsigned long long)rj_indel);
      if (rj_clipfail) atomicAdd(&st[3 + FGX_REJ_CLIP_OVERLAP_FAILED], (unsigned long long)rj_clipfail);
    }
  };
  {
    uint32_t eq_lo = 0, eq_hi = 0;
    for (unsigned long long um = __ballot(act); um; um &= um - 1) {
      const uint32_t u = (uint32_t)__builtin_ctzll(um);
      const uint32_t hash_u = rlane(hash, u), nlen_u = rlane(name_len, u);      // (both cross-lane reads by every lane: no short circuit between them)
      const bool eq = hash_u == hash && nlen_u == name_len && u != lane;
      if (u < 32) eq_lo |= (eq ? 1u : 0u) << u; else eq_hi |= (eq ? 1u : 0u) << (u - 32);
    }
    unsigned long long todo = act ? ((unsigned long long)eq_lo | ((unsigned long long)eq_hi << 32)) : 0ull;
    uint32_t n_same = 0;
    bool neighbour = false;
    while (__any(todo != 0)) {
      const bool mine = todo != 0;
      const uint32_t u = mine ? (uint32_t)__builtin_ctzll(todo) : lane;
      todo &= todo - 1;
      const uint32_t lou = (uint32_t)__shfl((int)lo, (int)u);
      if (mine) {
        bool same = true;
        for (uint32_t i = 0; i < name_len; i += 8) {
          unsigned long long wa = ld64u(W, lou + 32 + i), wb = ld64u(W, lo + 32 + i);
          if (i + 8 > name_len) { unsigned long long mk = (1ULL << (8 * (name_len - i))) - 1; wa &= mk; wb &= mk; }
          if (wa != wb) { same = false; break; }
        }
        if (same) { n_same++; neighbour |= u == (lane ^ 1); }
      }
    }
    // exactly two records per name, mates adjacent (template order by first appearance is then the order of the R1s and of
    // the R2s — the order the likelihoods are summed in); singletons, triples, interleaved pairs: general path
    if (__any(act && (n_same != 1 || !neighbour))) { to_defer(); return; }
  }
  const bool first = (flags & bam::F_FIRST) != 0;
  const uint32_t m_flags = (uint32_t)__shfl((int)flags, (int)(lane ^ 1)), m_lseq = (uint32_t)__shfl((int)l_seq, (int)(lane ^ 1));
  const int32_t m_pos = (int32_t)__shfl((int)pos, (int)(lane ^ 1)), m_ref = (int32_t)__shfl((int)ref_id, (int)(lane ^ 1));
  const uint32_t m_frself = (uint32_t)__shfl((int)strand, (int)(lane ^ 1));
  if (__any(act && first == ((m_flags & bam::F_FIRST) != 0))) { to_defer(); return; }          // both mates FIRST or neither
  // is_primary_fr_pair_raw (overlap.rs:83-108): mapped, mates mapped, one reference, opposite strands, FR by the reverse record
  const bool m_rev = (m_flags & bam::F_REVERSE) != 0;
  const bool fr = act && !(flags & bam::F_MATE_UNMAPPED) && !(m_flags & bam::F_MATE_UNMAPPED) && ref_id == m_ref && rev != m_rev &&
                  (rev ? strand : m_frself) != 0;
  const uint32_t n_notfr = (uint32_t)__popcll(__ballot(act && !fr));
  rj_notfr = n_notfr; s_filtered += n_notfr;
  // overlap clip against the mate in hand (overlap.rs:223-357 for two single-M alignments)
  if (fr) {
    const long long L = l_seq, ML = m_lseq, tp = (long long)pos + 1, mp = (long long)m_pos + 1;
    const long long read_end = tp - 1 + L, mate_end = mp - 1 + ML;
    long long cl = 0;
    if (rev) {
      if (!(tp > mate_end) && !(read_end < mp)) {
        const long long fs = tp > mp ? tp : mp;
        long long rb = fs - tp; if (rb > L) rb = L;
        long long mb = fs - mp; if (mb > ML) mb = ML;
        cl = rb > mb ? rb - mb : 0;
      }
    } else {
      if (!(read_end < mp) && !(mate_end < tp)) {
        const long long ls = read_end < mate_end ? read_end : mate_end;
        long long ra = ls - tp + 1; if (ra > L) ra = L;
        long long ma = ls - mp + 1; if (ma > ML) ma = ML;
        const long long rp = L - ra, mq = ML - ma;
        cl = rp > mq ? rp - mq : 0;
      }
    }
    clip = (uint32_t)cl;
  }
  // ClippedRecordInfo (:1006-1040): hard clip at the 3' end; a reverse read loses its start and moves right
  const uint32_t clen = fr ? (l_seq > clip ? l_seq - clip : 0) : 0;                 // clipped_seq_len = reference span of the clipped M
  const unsigned long long adj = (unsigned long long)((long long)pos + 1) + (rev ? (clip < l_seq ? clip : l_seq) : 0);
  const unsigned long long r1set = uniform_u64(__ballot(fr && first)), r2set = uniform_u64(__ballot(fr && !first));
  const uint32_t n_strand = (uint32_t)__popcll(r1set | r2set);
  if (!r1set) { flush_stats(); return; }
  auto reject_all = [&](uint32_t& counter) { counter += n_strand; s_filtered += n_strand; flush_stats(); };
  if ((uint32_t)__popcll(r1set) < P.cmin_reads) { reject_all(rj_insuf); return; }
  // most common alignment: with one M op per read the simplified clipped CIGAR is (M, l_seq) — equal read lengths ⇒ one group
  {
    const uint32_t l1ref = rlane(l_seq, (uint32_t)__builtin_ctzll(r1set)), l2ref = rlane(l_seq, (uint32_t)__builtin_ctzll(r2set));
    if (__any(fr && l_seq != (first ? l1ref : l2ref))) { to_defer(); return; }
  }
  if (P.cmax_reads >= 0) {
    if (P.cmax_reads == 0) { reject_all(rj_insuf); return; }
    if ((long long)__popcll(r1set) > P.cmax_reads) { to_defer(); return; }       // the cap bites: name-rank downsampling on the host
  }
  // longest R1 / R2 by clipped reference length (first maximum)
  const uint32_t mx1 = wave_max((fr && first) ? clen + 1 : 0), mx2 = wave_max((fr && !first) ? clen + 1 : 0);
  const uint32_t L1 = (uint32_t)__builtin_ctzll(__ballot(fr && first && clen + 1 == mx1)), L2 = (uint32_t)__builtin_ctzll(__ballot(fr && !first && clen + 1 == mx2));
  const bool r1_neg = (rlane(flags, L1) & bam::F_REVERSE) != 0, r2_neg = (rlane(flags, L2) & bam::F_REVERSE) != 0;
  const uint32_t adj_lo = (uint32_t)adj, adj_hi = (uint32_t)(adj >> 32);
  auto adj_of = [&](uint32_t r) { return ((unsigned long long)rlane(adj_hi, r) << 32) | rlane(adj_lo, r); };
  const uint32_t Lp = r1_neg ? L2 : L1, Ln = r1_neg ? L1 : L2;
  const unsigned long long adj_p = adj_of(Lp), adj_n = adj_of(Ln), adj_1 = adj_of(L1), adj_2 = adj_of(L2);
  const uint32_t rl_p = rlane(clen, Lp), rl_n = rlane(clen, Ln), rl_1 = rlane(clen, L1), rl_2 = rlane(clen, L2);
  const unsigned long long pos_end = adj_p + (rl_p ? rl_p - 1 : 0), neg_end = adj_n + (rl_n ? rl_n - 1 : 0);
  const unsigned long long ov_s = adj_n > adj_p ? adj_n : adj_p, ov_e = pos_end < neg_end ? pos_end : neg_end;
  const long long duplex_len = (long long)ov_e - (long long)ov_s + 1;
  if (duplex_len < (long long)P.cmin_duplex_len) { reject_all(rj_overlap); return; }
  // read_pos_at_ref_pos_raw (cigar.rs:461-500) on [H?] M [H?]: 1-based query position, 0 = htsjdk's out-of-range sentinel
  auto read_pos = [](unsigned long long a, uint32_t rl, unsigned long long p) -> long long { return (p >= a && rl && p <= a + rl - 1) ? (long long)(p - a + 1) : 0; };
  if (read_pos(adj_1, rl_1, ov_s) - read_pos(adj_2, rl_2, ov_s) != read_pos(adj_1, rl_1, ov_e) - read_pos(adj_2, rl_2, ov_e)) { reject_all(rj_indel); return; }
  const long long prp = read_pos(adj_p, rl_p, ov_e), nrp = read_pos(adj_n, rl_n, ov_e);
  if (prp == 0 || nrp == 0) { reject_all(rj_indel); return; }
  if (prp + (long long)rl_n < nrp) { to_defer(); return; }                       // the general path raises the error
  const uint32_t cons_len = (uint32_t)(prp + (long long)rl_n - nrp);
  const uint32_t len1 = wave_max((fr && first) ? clen : 0), len2 = wave_max((fr && !first) ? clen : 0);
  if (cons_len < len1 || cons_len < len2) { reject_all(rj_clipfail); return; }

  PH(5)
  // ---- 6C. the two single-strand column sets (ss caller: min_reads 1, no quality masking, min consensus base quality 0) ----------
  final_len = clen;
  const uint32_t d0 = seq_lo | (qual_lo << 16), d1 = l_seq | (final_len << 16), d2 = (rev ? 1u : 0u) << 16;
  const DeviceTables* T = P.T;
  const uint64_t col_base = P.col_base[g];
  CallConst KC;
  KC.cap = (uint32_t)__builtin_amdgcn_readfirstlane((int)sT.cap);
  KC.cap_threshold = uniform_f64(sT.cap_threshold); KC.half_cerr_at_cap = uniform_f64(sT.half_cerr_at_cap);
  bool odd_base = false;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const unsigned long long members = k == 0 ? r1set : r2set;
    const uint32_t mc = (uint32_t)__popcll(members), elen = k == 0 ? len1 : len2, eoff = k == 0 ? 0 : len1;
    // (the column loop runs the SAME number of passes in every lane — a lane past the end of the set takes part in the cross-lane reads of a pass and
    // stores nothing —, and a single-read set reads its read's fields once ahead of it: cross-lane operations stay in wave-uniform control flow, which
    // tests/wavemu checks)
    uint32_t one_seq = 0, one_qual = 0, one_lseq = 0, one_flags = 0;
    if (mc == 1) {
      const uint32_t r = (uint32_t)__builtin_ctzll(members);
      one_seq = rlane(seq_lo, r); one_qual = rlane(qual_lo, r); one_lseq = rlane(l_seq, r); one_flags = rlane(flags, r);
    }
    for (uint32_t p0 = 0; p0 < elen; p0 += 64) {
      const uint32_t p = p0 + lane;
      const bool incol = p < elen;
      const uint64_t o = col_base + eoff + p;
      if (mc == 1) {   // single-read consensus: LUT keyed by the unclamped quality (vanilla_caller.rs:1677-1708)
        if (incol) {
          uint32_t code, q;
          view(one_seq, one_qual, one_lseq, (one_flags & bam::F_REVERSE) != 0, 0, p, &code, &q);
          const uint8_t adjq = q < 94 ? T->single_input_quals[q] : 0;
          P.col_code[o] = (uint8_t)code; P.col_qual[o] = adjq; P.col_depth[o] = code != 15 ? 1 : 0; P.col_err[o] = 0;
          if (code != 15 && code != 1 && code != 2 && code != 4 && code != 8) odd_base = true;   // IUPAC code in a lone read: general path
        }
      } else {
        ChainAcc acc;                           // two Kahan chains while the column shows one base (consensus_math.h)
        acc.reset();
        for (unsigned long long m = members; m; m &= m - 1) {
          const uint32_t r = (uint32_t)__builtin_ctzll(m);
          const uint32_t x0 = rlane(d0, r), x1 = rlane(d1, r), x2 = rlane(d2, r);
          bool valid = incol && p < (x1 >> 16);
          const bool rv = (x2 >> 16) != 0;
          const uint32_t idx = valid ? (rv ? (x1 & 0xFFFF) - 1 - p : p) : 0;
          const uint32_t bb = W[(x0 & 0xFFFF) + (idx >> 1)];
          const uint32_t c = (bb >> ((~idx & 1) << 2)) & 15;
          const uint32_t q = W[(x0 >> 16) + idx];
          const uint32_t co = rv ? __builtin_bitreverse32(c) >> 28 : c;     // read orientation: complement = bit reversal of the code
          valid = valid && __popc(co) == 1;                                 // only the one-hot codes A C G T are observations
          const uint32_t qq = q < FGX_MAX_PHRED ? q : FGX_MAX_PHRED;
          const double2 pr = *(const double2*)&sPair[qq][0];
          acc.add(valid, co, pr.x, pr.y);
        }
        double ll[4];
        uint32_t obs[4];
        acc.finish(ll, obs);
        int bi = -1;
        uint8_t q = 0;
        const bool resolved = !incol || column_call_fast_lds(sT, KC, ll, obs, &bi, &q);
        const uint32_t depth = obs[0] + obs[1] + obs[2] + obs[3];
        if (incol) P.col_depth[o] = (uint16_t)depth;
        push_full(!resolved, o, ll, obs);
        if (resolved && incol) {
          const uint32_t err = depth - (bi == 0 ? obs[0] : bi == 1 ? obs[1] : bi == 2 ? obs[2] : bi == 3 ? obs[3] : 0u);
          P.col_code[o] = depth < 1 ? (uint8_t)15 : bi >= 0 ? (uint8_t)(1u << bi) : (uint8_t)15;
          P.col_qual[o] = depth < 1 ? (uint8_t)0 : q;
          P.col_err[o] = (uint16_t)err;
        }
      }
    }
  }
  if (__any(odd_base)) { to_defer(); return; }

  PH(6)
  // ---- 7C. consensus UMI over EVERY record of the group that carries RX (codec_caller.rs:1725-1744) -----------------------------
  const DeviceTables* TU = P.TU;
  const unsigned long long with = rxmask;
  const uint32_t rx_cnt = (uint32_t)__popcll(with);
  uint32_t ulen = 0;
  char my_ch = 0;
  bool rx_fail = false;
  const uint32_t slot = slot0 + 1;
  if (rx_cnt) {
    const uint32_t f = (uint32_t)__builtin_ctzll(with);
    ulen = rlane(rx_len, f);
    if (__any(((with >> lane) & 1) && rx_len != ulen) || ulen > FAST_RX_CAP) rx_fail = true;
    else {
      const bool mychar = lane < ulen;
      const uint32_t f_lo = rlane(rx_lo, f);
      const uint8_t c0 = mychar ? W[f_lo + lane] : (uint8_t)'A';
      bool differs = false;
      if ((with >> lane) & 1) {
        for (uint32_t i = 0; i < ulen; i += 8) {
          unsigned long long a = ld64u(W, rx_lo + i), b = ld64u(W, f_lo + i);
          if (i + 8 > ulen) { unsigned long long mk = (1ULL << (8 * (ulen - i))) - 1; a &= mk; b &= mk; }
          differs |= a != b;
        }
      }
      if (rx_cnt == 1 || !__any(differs)) {
        const int bl = bam::ascii_to_lane(c0);
        my_ch = rx_cnt == 1 ? (char)c0 : bl != 255 ? "ACGT"[bl] : (c0 == 'N' || c0 == 'n') ? 'N' : (char)c0;
      } else {
        ColumnAcc acc;
        acc.reset();
        uint32_t non_dna = 0, seen = 0;
        uint8_t fch = 0;
        bool mixed = false;
        const double uc = TU->t.correct[20], ue = TU->t.error_per_alt[20];
        for (unsigned long long m = with; m; m &= m - 1) {
          const uint32_t r = (uint32_t)__builtin_ctzll(m);
          const uint8_t ch = mychar ? W[rlane(rx_lo, r) + lane] : (uint8_t)'A';
          if (seen == 0) fch = ch;
          seen++;
          const uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
          const bool dna = up == 'A' || up == 'C' || up == 'G' || up == 'T' || up == 'N';
          if (dna) { const int bl = bam::ascii_to_lane(ch); if (bl != 255) acc.add(bl, uc, ue); }
          else { non_dna++; if (ch != fch) mixed = true; }
        }
        bool need_full = false, bad_col = false;
        if (non_dna == 0) {
          CallConst KU;
          KU.cap = TU->t.cap; KU.cap_threshold = TU->t.cap_threshold; KU.half_cerr_at_cap = TU->t.half_cerr_at_cap;
          int bi; uint8_t q;
          const bool resolved = column_call_fast_lds(TU->t, KU, acc.s, acc.obs, &bi, &q);
          my_ch = bi >= 0 ? "ACGT"[bi] : 'N';
          need_full = !resolved;
        } else if (non_dna == seen && !mixed) my_ch = (char)fch;
        else bad_col = true;
        if (!mychar) { need_full = false; bad_col = false; }
        if (__any(bad_col)) rx_fail = true;
        push_full(need_full, (1ull << 63) | ((uint64_t)slot << 8) | lane, acc.s, acc.obs);
      }
    }
  }
  if (rx_fail || __any(list_overflow)) { to_defer(); return; }

  // ---- 8C. descriptor ------------------------------------------------------------------------------------------------------------
  {
    CodecDesc* D = &P.cends[slot];
    bool rx_empty = true;                                    // an all-empty consensus UMI is not written (:1740-1743)
    if (rx_cnt && ulen) rx_empty = false;
    if (lane < ulen && rx_cnt) D->rx[lane] = my_ch;
    // cell barcode: first record, R1s then R2s, that carries a non-empty value (:1708-1722)
    const unsigned long long cbm = __ballot(act && has_cb && cb_len > 0);
    const unsigned long long c1 = cbm & r1set, c2 = cbm & r2set;
    const bool any_cb = P.cell0 && (c1 | c2);
    const uint32_t fc = c1 ? (uint32_t)__builtin_ctzll(c1) : c2 ? (uint32_t)__builtin_ctzll(c2) : 0;
    const uint32_t fc_cb_lo = rlane(cb_lo, fc), fc_cb_len = rlane(cb_len, fc), fc_lo = rlane(lo, fc);
    if (lane == 0) {
      const uint32_t C = cons_len;
      D->s1_off = col_base; D->s2_off = col_base + len1; D->l1 = len1; D->l2 = len2; D->cons_len = C;
      D->first_rec = r0; D->cb_rec = r0 + fc;
      D->mi_off = (uint16_t)(mi_lo - lo); D->mi_len = (uint8_t)mi_len;                 // lane 0 = record 0
      D->has_cb = any_cb ? 1 : 0; D->cb_off = (uint16_t)(fc_cb_lo - fc_lo); D->cb_len = (uint8_t)fc_cb_len;
      D->has_rx = rx_empty ? 0 : 1; D->rx_len = (uint8_t)ulen; D->flags = (uint8_t)((r1_neg ? 1 : 0) | (r2_neg ? 2 : 0));
      const uint32_t nm = P.prefix_len + 1 + mi_len;
      // every int tag holds a depth <= 128 (<= 64 records per wavefront): 4 bytes each
      const uint32_t size = 32 + nm + 1 + (C + 1) / 2 + C + (3 + P.rg_len + 1) + (3 + mi_len + 1) + 3 * (4 + 4 + 7) +
                            (P.per_base_tags ? 4 * (8 + 2 * C) + 4 * (3 + C + 1) : 0) + (any_cb ? 3 + fc_cb_len + 1 : 0) + (rx_empty ? 0 : 3 + ulen + 1);
      D->rec_size = size;
      D->valid = 1;
      P.rec_sizes[slot] = (uint64_t)size + 4;
    }
  }
  s_cons = 1;
  flush_stats();
  PH(8)
  }
}

// -----------------------------------------------------------------------------------------------------
// k_call_full — the columns (and UMI characters) the fast path could not establish, compacted by the
// family kernels into N_LISTS append lists: one lane per item, every lane busy with the full
// log-sum-exp chain (call_full, base_builder.rs:1023-1054) on the glibc-compatible device libm.
// -----------------------------------------------------------------------------------------------------
struct FullParams {
  const FullItem* items; const uint32_t* count; uint32_t cap;
  const DeviceTables* T; const DeviceTables* TU;
  uint32_t min_reads; uint8_t min_cons_bq;
  uint8_t* col_code; uint8_t* col_qual; uint16_t* col_err;
  char* rx_base; uint32_t rx_stride;   // consensus-UMI characters: rx_base + slot * rx_stride + index
  uint16_t* col_depth; uint32_t min_input_bq;   // observation items (k_split_cols): the column's depth is counted here
  // direct records (FullItem.dest & FULL_DEST_DIRECT): the column is patched in the record k_split_cols wrote
  uint8_t* out; const SlotDesc* slot_desc; uint32_t* slot_err; uint32_t rg_len, per_base_tags;
};

// A column of a directly written record (fastpath.h: SlotDesc): the sequence nibble is OR-ed into the word that holds its byte (k_split_cols
// left it 0; the other nibble of the byte may be patched by another lane), quality, cd (observation items: the depth is counted here) and
// ce entries are stored, and the column's errors are added to the record's count for k_fix_ce.
__device__ __forceinline__ void full_patch_direct(const FullParams& P, uint64_t dest, uint32_t ob, uint32_t oq, uint32_t err, uint32_t depth, bool with_depth) {
  const uint32_t slot = (uint32_t)(dest >> 16), p = (uint32_t)dest & 0xFFFFu;
  const SlotDesc sd = P.slot_desc[slot];
  uint8_t* const rec = P.out + sd.out_off;
  const uint32_t Lc = sd.lc_seq & 0xFFFFu, o_seq = sd.lc_seq >> 16;
  const uint32_t o_qual = o_seq + ((Lc + 1u) >> 1), o_cd = o_qual + Lc + 3u + P.rg_len + 1u + 15u + 8u, o_ce = o_cd + 2u * Lc + 8u;
  const uintptr_t a = (uintptr_t)(rec + o_seq + (p >> 1));
  atomicOr((unsigned int*)(a & ~(uintptr_t)3), ob << (8u * (uint32_t)(a & 3) + ((p & 1u) ? 0u : 4u)));
  rec[o_qual + p] = (uint8_t)oq;
  const uint32_t e16 = err < 32767u ? err : 32767u;
  if (P.per_base_tags) {
    if (with_depth) { const uint16_t d16 = (uint16_t)depth; __builtin_memcpy(rec + o_cd + 2u * p, &d16, 2); }
    const uint16_t w = (uint16_t)e16; __builtin_memcpy(rec + o_ce + 2u * p, &w, 2);
  }
  if (e16) atomicAdd(&P.slot_err[slot], e16);
}
struct DirCast { __host__ __device__ __forceinline__ uint64_t operator()(const uint32_t& v) const { return (uint64_t)v; } };
typedef hipcub::TransformInputIterator<uint64_t, DirCast, const uint32_t*> DirSizeIn;
// a chunk's record offsets: its exclusive scan + the bytes of the chunks before; the last family leaves the next chunk's base
__global__ void k_dir_carry(uint64_t* __restrict__ off, const uint32_t* __restrict__ size, uint32_t n, uint64_t* __restrict__ base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t b = base[0], local = off[i];
  off[i] = local + b;
  if (i == n - 1) base[1] = b + local + size[i];
}
// cE of the directly written records whose columns had errors: f32(total errors) / f32(total depth) (vanilla_caller.rs:1805-1808); the
// value sits 11 bytes into the cD cM cE block that follows the RG tag
__global__ void k_fix_ce(const SlotDesc* __restrict__ slot_desc, const uint32_t* __restrict__ slot_err, uint32_t n_slots, uint8_t* __restrict__ out, uint32_t rg_len) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_slots) return;
  const uint32_t e = slot_err[i];
  if (!e) return;
  const SlotDesc sd = slot_desc[i];
  const uint32_t Lc = sd.lc_seq & 0xFFFFu, o_seq = sd.lc_seq >> 16;
  const uint32_t o_t3 = o_seq + ((Lc + 1u) >> 1) + Lc + 3u + rg_len + 1u;
  const float rate = sd.sum_depth > 0 ? (float)e / (float)sd.sum_depth : 0.0f;
  const uint32_t u = __float_as_uint(rate);
  uint8_t* const q = out + sd.out_off + o_t3 + 11u;
  q[0] = (uint8_t)u; q[1] = (uint8_t)(u >> 8); q[2] = (uint8_t)(u >> 16); q[3] = (uint8_t)(u >> 24);
}
// The merge of a batch in which some families left the split pipeline (or were deferred): the directly written records of family g lie at
// dir_off[g] in `dir`; they move to their place in the final stream (the scan of ALL record sizes), a wavefront per family.
__global__ __launch_bounds__(256) void k_dir_copy(const SplitOut* __restrict__ split_out, const uint64_t* __restrict__ dir_off, const uint64_t* __restrict__ out_off,
                                                  const uint64_t* __restrict__ sizes, const uint8_t* __restrict__ dir, uint8_t* __restrict__ out, uint32_t n_grp) {
  const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), lane = threadIdx.x & 63;
  if (g >= n_grp) return;
  if (split_out[g].status != 1) return;
  const uint64_t bytes = sizes[3 * (size_t)g] + sizes[3 * (size_t)g + 1] + sizes[3 * (size_t)g + 2];
  const uint8_t* const src = dir + dir_off[g];
  uint8_t* const dst = out + out_off[3 * (size_t)g];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint64_t i = 16ull * lane;
  for (; i + 16 <= bytes; i += 1024) { u32x4 v; __builtin_memcpy(&v, src + i, 16); __builtin_memcpy(dst + i, &v, 16); }
  const uint64_t whole = bytes & ~15ull;
  if (lane < (uint32_t)(bytes - whole)) dst[whole + lane] = src[whole + lane];
}

// Round 5: two phases per workgroup.  Phase 1 (a thread per item): the sums of an observation item, then the TABLE answer of a column that
// shows one base (try_unanimous_fast_path, base_builder.rs:883-994) — most observation items of k_split_cols's packed pass are such columns
// (one base seen fewer times than its gate-free depth) and end here.  What needs the log-sum-exp chain (call_full: several hundred f64
// instructions through the libm port) is appended to a list in LDS, and phase 2 runs it with the list's entries side by side in the
// lanes: a wavefront pays for call_full only when it holds such columns, and then with all its lanes busy — before, one such item in
// 64 kept the whole wavefront in the chain.
__global__ __launch_bounds__(256) void k_call_full(FullParams P) {
  __shared__ double sLL[256][4];
  __shared__ unsigned long long sDest[256];
  __shared__ uint32_t sObs[256];         // observation counts, a byte each, in the order of sLL; bit 31 of sKind: an observation item (depth counted here)
  __shared__ uint32_t sKind[256];
  __shared__ uint32_t sCnt;
  __shared__ double sTab[94][2];         // {correct, error_per_alt} by quality: an observation item reads a pair per observation (from global memory that is a dependent ~1 us chain per item)
  const uint32_t list = blockIdx.y;
  uint32_t cnt = P.count[list];
  if (cnt > P.cap) cnt = P.cap;
  if (blockIdx.x * blockDim.x >= cnt) return;      // (the grid covers every list's capacity: most workgroups have nothing)
  if (threadIdx.x == 0) sCnt = 0;
  if (threadIdx.x < 94) { sTab[threadIdx.x][0] = P.T->t.correct[threadIdx.x]; sTab[threadIdx.x][1] = P.T->t.error_per_alt[threadIdx.x]; }
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // what a decided column leaves (vanilla_caller.rs:1711-1748); `is_obs`: an observation item — its depth is written too
  auto finish = [&](uint64_t dest, int bi, uint8_t q, uint32_t o0, uint32_t o1, uint32_t o2, uint32_t o3, bool is_obs) {
    const uint32_t depth = o0 + o1 + o2 + o3;
    const uint32_t err = depth - (bi == 0 ? o0 : bi == 1 ? o1 : bi == 2 ? o2 : bi == 3 ? o3 : 0u);
    const uint8_t code = bi >= 0 ? (uint8_t)(1u << bi) : 15;
    uint8_t ob, oq;
    if (depth < P.min_reads) { ob = 15; oq = 0; }
    else if (q < P.min_cons_bq) { ob = 15; oq = FGX_MIN_PHRED; }
    else { ob = code; oq = q; }
    if (dest & FULL_DEST_DIRECT) { full_patch_direct(P, dest, ob, oq, err, depth, is_obs); return; }
    P.col_code[dest] = ob; P.col_qual[dest] = oq; P.col_err[dest] = (uint16_t)(err < 32767u ? err : 32767u);
    if (is_obs) P.col_depth[dest] = (uint16_t)depth;
  };
  auto push = [&](uint64_t dest, const double* ll, uint32_t obs, uint32_t kind) {
    const uint32_t k = atomicAdd(&sCnt, 1u);
    sLL[k][0] = ll[0]; sLL[k][1] = ll[1]; sLL[k][2] = ll[2]; sLL[k][3] = ll[3];
    sDest[k] = dest; sObs[k] = obs; sKind[k] = kind;
  };
  if (i < cnt) {
    FullItem it = P.items[(size_t)list * P.cap + i];
    if (it.chains == FULL_ITEM_CONT) { /* observations 16 .. of the column whose head item sits before it (or padding) */ }
    else if (it.chains & FULL_ITEM_OBS) {
      // a column k_split_cols did not answer: its observations in file order, 16 bits each — the four-lane Kahan sum of
      // ConsensusBaseBuilder::add (base_builder.rs:836-868), then the call.  More than 16 of them: the following items hold the rest.
      const uint32_t m = it.chains & 0xFF;
      ColumnAcc acc;
      acc.reset();
      const ConsensusTables& TT = P.T->t;
      for (uint32_t k0 = 0; k0 < m; k0 += 16) {
        uint16_t ob16[16];
        if (k0 == 0) __builtin_memcpy(ob16, it.ll, 32);
        else __builtin_memcpy(ob16, P.items[(size_t)list * P.cap + i + (k0 >> 4)].ll, 32);
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) {
          if (k0 + j < m) {
            const uint32_t o = ob16[j], q = o & 0xFF, code = (o >> 8) & 15;
            const int bl = bam::code_to_lane((uint8_t)code);
            if (bl != 255 && q >= P.min_input_bq) { const uint32_t qq = q < 93 ? q : 93; acc.add(bl, sTab[qq][0], sTab[qq][1]); }
          }
        }
      }
      int bi = -1;
      uint8_t q = FGX_MIN_PHRED;
      if (acc.contributions() == 0 || unanimous_fast_path(TT, acc.s, acc.obs, &bi, &q)) finish(it.dest, bi, q, acc.obs[0], acc.obs[1], acc.obs[2], acc.obs[3], true);
      else push(it.dest, acc.s, acc.obs[0] | (acc.obs[1] << 8) | (acc.obs[2] << 16) | (acc.obs[3] << 24), 0x80000000u);
    } else {
      if (it.chains) {   // chains by order of appearance (k_simplex_wave2 / k_simplex_seg): the bases are sorted out here, once per item
        const uint32_t b1 = it.chains & 15, b2 = (it.chains >> 4) & 15, b3 = (it.chains >> 8) & 15;
        const double c1 = it.ll[0], c2 = it.ll[1], c3 = it.ll[2], cR = it.ll[3];
        const uint32_t n1 = it.obs & 0xFF, n2 = (it.obs >> 8) & 0xFF, n3 = (it.obs >> 16) & 0xFF, nR = it.obs >> 24;
        uint32_t packed = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const uint32_t code = 1u << k;
          it.ll[k] = code == b1 ? c1 : code == b2 ? c2 : code == b3 ? c3 : cR;
          packed |= (code == b1 ? n1 : code == b2 ? n2 : code == b3 ? n3 : nR) << (8 * k);
        }
        it.obs = packed;
      }
      push(it.dest, it.ll, it.obs, 0u);       // (the kernels that send these have run their gates already)
    }
  }
  __syncthreads();
  const uint32_t todo = sCnt;
  if (threadIdx.x < todo) {
    const uint32_t k = threadIdx.x;
    const uint64_t dest = sDest[k];
    const double ll[4] = {sLL[k][0], sLL[k][1], sLL[k][2], sLL[k][3]};
    const uint32_t ob = sObs[k];
    const bool is_rx = (dest >> 63) != 0;
    const ConsensusTables& T = is_rx ? P.TU->t : P.T->t;
    int bi;
    uint8_t q;
    call_full(T, ll, &bi, &q);
    if (is_rx) {
      const uint64_t d = dest & ~(1ull << 63);
      P.rx_base[(size_t)(d >> 8) * P.rx_stride + (d & 0xFF)] = bi >= 0 ? "ACGT"[bi] : 'N';
    } else finish(dest, bi, q, ob & 0xFF, (ob >> 8) & 0xFF, (ob >> 16) & 0xFF, ob >> 24, (sKind[k] >> 31) != 0);
  }
}

// -----------------------------------------------------------------------------------------------------
// pass B: one wavefront per consensus read
// -----------------------------------------------------------------------------------------------------
// byte j of an integer tag `ab:<c|C|S>:v` (smallest type, signed first; v <= 32767 here)
__device__ __forceinline__ uint8_t int_tag_byte(uint32_t j, char a, char b, uint32_t v) {
  return j == 0 ? (uint8_t)a : j == 1 ? (uint8_t)b : j == 2 ? (uint8_t)(v <= 127 ? 'c' : v <= 255 ? 'C' : 'S') : j == 3 ? (uint8_t)v : (uint8_t)(v >> 8);
}

// One wavefront serialises one consensus record (block_size prefix, 32-byte core, name, packed bases, quals, tags
// `RG cD cM cE [cd ce] MI [CB] RX` — vanilla_caller.rs:1767-1881).  The kernel is bound by memory round trips per
// wavefront, not by bytes: so (hot path, consensus <= 192 columns and short names/tags) EVERY load of the record is
// issued before the first store — one wait instead of one per field — descriptor fields are scalar loads, and every
// field group is one full-wave store in which each lane computes the byte it owns (fixed header bytes included).
#ifndef FGX_EMIT_FLAT
#define FGX_EMIT_FLAT 1   /* k_emit's small fields as straight-line code (0: the nested conditionals of rounds 2 - 4, for measurements) */
#endif
struct EmitCtx {
  uint8_t* q; const uint8_t* first; const uint8_t* code; const uint8_t* cq; const uint16_t* cd; const uint16_t* ce;
  uint32_t Lc, name_len, mi_len, mi_off, flag, rec_size;
};

// unaligned global dwords (gfx950 takes them in one instruction)
__device__ __forceinline__ uint32_t gld32u(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint2 gld64u(const uint8_t* p) { uint2 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void gst32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void gst16u(uint8_t* p, uint32_t v) { const uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }

__device__ __forceinline__ void emit_core(uint8_t* q, uint32_t lane, const EmitCtx& X) {
  // block_size + fixed core: ref_id -1, pos -1, l_read_name, mapq 0, bin 4680, n_cigar_op 0, flag, l_seq, next_ref -1, next_pos -1, tlen 0
  if (lane < 36) {
    const uint32_t dw = lane >> 2;
    const uint32_t v = dw == 0 ? X.rec_size : dw == 3 ? ((X.name_len + 1) | (4680u << 16)) : dw == 4 ? (X.flag << 16) : dw == 5 ? X.Lc : dw == 8 ? 0u : 0xFFFFFFFFu;
    q[lane] = (uint8_t)(v >> (8 * (lane & 3)));
  }
}
__device__ __forceinline__ uint8_t cdcmce_byte(uint32_t lane, uint32_t n_cd, uint32_t n_cm, uint32_t maxd, uint32_t mind, float rate) {
  if (lane < n_cd) return int_tag_byte(lane, 'c', 'D', maxd);
  if (lane < n_cd + n_cm) return int_tag_byte(lane - n_cd, 'c', 'M', mind);
  const uint32_t j = lane - n_cd - n_cm, u = __float_as_uint(rate);
  return j == 0 ? 'c' : j == 1 ? 'E' : j == 2 ? 'f' : (uint8_t)(u >> (8 * (j - 3)));
}

// any length: field after field (one load → store round trip per 64 bytes)
__device__ void emit_generic(const EmitParams& P, const EndDesc& D, const EmitCtx& X, uint32_t lane) {
  uint8_t* q = X.q;
  const uint32_t Lc = X.Lc, name_len = X.name_len, mi_len = X.mi_len, mi_off = X.mi_off;
  uint32_t maxd = 0, mind = 0xFFFFFFFFu, sumd = 0, sume = 0;
  for (uint32_t i = lane; i < Lc; i += 64) { uint32_t d = X.cd[i], e = X.ce[i]; maxd = d > maxd ? d : maxd; mind = d < mind ? d : mind; sumd += d; sume += e; }
  emit_core(q, lane, X);
  q += 36;
  for (uint32_t i = lane; i < name_len + 1; i += 64) {
    uint8_t ch;
    if (i < P.prefix_len) ch = (uint8_t)P.prefix[i];
    else if (i == P.prefix_len) ch = ':';
    else if (i < name_len) ch = X.first[mi_off + (i - P.prefix_len - 1)];
    else ch = 0;
    q[i] = ch;
  }
  q += name_len + 1;
  for (uint32_t i = lane; i < (Lc + 1) / 2; i += 64) {
    uint8_t hi = X.code[2 * i], lo = (2 * i + 1 < Lc) ? X.code[2 * i + 1] : 0;
    q[i] = (uint8_t)((hi << 4) | lo);
  }
  q += (Lc + 1) / 2;
  for (uint32_t i = lane; i < Lc; i += 64) q[i] = X.cq[i];
  q += Lc;
  for (uint32_t i = lane; i < 3 + P.rg_len + 1; i += 64) q[i] = i == 0 ? 'R' : i == 1 ? 'G' : i == 2 ? 'Z' : i - 3 < P.rg_len ? (uint8_t)P.rg[i - 3] : 0;
  q += 3 + P.rg_len + 1;
  maxd = wave_max(maxd); mind = wave_min(mind); sumd = wave_sum(sumd); sume = wave_sum(sume);
  if (Lc == 0) { maxd = 0; mind = 0; }
  const float ce_rate = sumd > 0 ? (float)sume / (float)sumd : 0.0f;
  const uint32_t n_cd = 3 + int_tag_width(maxd), n_cm = 3 + int_tag_width(mind);
  if (lane < n_cd + n_cm + 7) q[lane] = cdcmce_byte(lane, n_cd, n_cm, maxd, mind, ce_rate);
  q += n_cd + n_cm + 7;
  if (P.per_base_tags) {
    for (int pass = 0; pass < 2; pass++) {
      const uint16_t* src = pass == 0 ? X.cd : X.ce;
      for (uint32_t i = lane; i < 8 + 2 * Lc; i += 64) {
        uint8_t v;
        if (i < 8) v = i == 0 ? 'c' : i == 1 ? (pass == 0 ? 'd' : 'e') : i == 2 ? 'B' : i == 3 ? 's' : (uint8_t)(Lc >> (8 * (i - 4)));
        else { const uint32_t k = i - 8; const uint16_t w = src[k >> 1]; v = (k & 1) ? (uint8_t)(w >> 8) : (uint8_t)w; }
        q[i] = v;
      }
      q += 8 + 2 * Lc;
    }
  }
  for (uint32_t i = lane; i < 3 + mi_len + 1; i += 64) q[i] = i == 0 ? (uint8_t)P.tag0 : i == 1 ? (uint8_t)P.tag1 : i == 2 ? 'Z' : i - 3 < mi_len ? X.first[mi_off + i - 3] : 0;
  q += 3 + mi_len + 1;
  if (D.has_cb) {
    const uint8_t* fk = P.blob + D.kept_off;
    const uint32_t cb_len = D.cb_len, cb_off = D.cb_off;
    for (uint32_t i = lane; i < 3 + cb_len + 1; i += 64) q[i] = i == 0 ? (uint8_t)P.cell0 : i == 1 ? (uint8_t)P.cell1 : i == 2 ? 'Z' : i - 3 < cb_len ? fk[cb_off + i - 3] : 0;
    q += 3 + cb_len + 1;
  }
  if (D.has_rx) {
    const uint32_t rx_len = D.rx_len;
    for (uint32_t i = lane; i < 3 + rx_len + 1; i += 64) q[i] = i == 0 ? 'R' : i == 1 ? 'X' : i == 2 ? 'Z' : i - 3 < rx_len ? (uint8_t)D.rx[i - 3] : 0;
  }
}

// One wavefront per FAMILY: its (up to three) records one after the other.  A third of the slots is empty on paired data (the
// fragment slot), and a wavefront that only finds `valid == 0` still costs a launch and a memory round trip; the descriptor
// carries blob OFFSETS, so the record's strings are one dependent load away instead of two (2.35 → 2.20 ms per 2 M records).
// (Assembling the record through LDS — whole-record image with byte writes, or dword-staged column arrays with dword payload
// copies — was measured three times, rounds 1 and 2: 3.4 – 4.0 ms.  The wave's lifetime is a chain of memory round trips, and
// every LDS hop adds one; registers-only streaming below is the fastest form found.)
// A record in two halves: everything it reads (emit_load: every load of the record issued back to back, nothing waited for), and
// the reductions + stores (emit_store).  k_emit issues the loads of ALL the family's records before the first store: a wavefront's
// life is a chain of memory round trips, and the records' round trips now run side by side instead of one after the other.
struct EmitLoads {
  EmitCtx X;
  uint2 cw; uint32_t qw, dw[2], ew[2], ao[2], so, qo, j3;
  uint8_t nb, rgb, mib, cbb, rxb;
  uint32_t cb_len, rx_len;
  bool has_cb, has_rx, generic;
};
__device__ __forceinline__ void emit_load(const EmitParams& P, const EndDesc& D, uint64_t out_off, uint32_t lane, EmitLoads& R) {
  // the descriptor's fields come out of LDS into vector registers; they are the same in every lane, and saying so (readfirstlane)
  // turns every address below into scalar base + 32-bit lane offset
  EmitCtx& X = R.X;
  X.q = P.out + (out_off - P.out_base);
  X.Lc = uni(D.cons_len);
  X.first = P.blob + uniform_u64(D.first_off);
  X.mi_len = uni(D.mi_len); X.mi_off = uni(D.mi_off);
  X.name_len = P.prefix_len + 1 + X.mi_len;
  X.rec_size = uni(D.rec_size);
  X.flag = bam::F_UNMAPPED;
  const uint32_t d_type = uni(D.type);
  if (d_type == 1) X.flag |= bam::F_PAIRED | bam::F_FIRST | bam::F_MATE_UNMAPPED;
  else if (d_type == 2) X.flag |= bam::F_PAIRED | bam::F_LAST | bam::F_MATE_UNMAPPED;
  const uint64_t col_off = uniform_u64(D.col_off);
  X.code = P.col_code + col_off; X.cq = P.col_qual + col_off; X.cd = P.col_depth + col_off; X.ce = P.col_err + col_off;
  const uint32_t Lc = X.Lc, name_len = X.name_len, mi_len = X.mi_len, mi_off = X.mi_off;
  R.has_cb = uni(D.has_cb) != 0; R.has_rx = uni(D.has_rx) != 0;
  R.cb_len = R.has_cb ? uni(D.cb_len) : 0; R.rx_len = R.has_rx ? uni(D.rx_len) : 0;
  R.generic = Lc > 192 || Lc < 8 || name_len + 1 > 64 || P.rg_len + 4 > 64 || R.cb_len + 4 > 64;
  if (R.generic) return;                                       // (any length: emit_generic, field after field)

  // Payloads move as (unaligned) dwords: lane l owns bytes [4l, 4l + 4) of a field, and the lane past the last whole dword
  // takes the field's LAST four bytes instead (an overlapping store of the same values) — no byte-granular tail.
  const uint32_t seq_bytes = (Lc + 1) / 2;
  R.so = min(4 * lane, seq_bytes - 4);                                                  // sequence: 4 output bytes = 8 columns
  R.qo = min(4 * lane, Lc - 4);                                                         // qualities: 4 columns
  R.cw = gld64u(X.code + 2 * R.so);                                                     // (column Lc may be read: one byte of slack)
  R.qw = gld32u(X.cq + R.qo);
#pragma unroll
  for (int t = 0; t < 2; t++) {                                                         // per-base arrays: 4 bytes = 2 columns
    R.ao[t] = min(4 * (lane + 64 * t), 2 * Lc - 4);
    R.dw[t] = gld32u((const uint8_t*)X.cd + R.ao[t]); R.ew[t] = gld32u((const uint8_t*)X.ce + R.ao[t]);
  }
  const uint32_t j3 = lane >= 3 ? lane - 3 : 0;
  R.j3 = j3;
  const uint32_t ni = lane > P.prefix_len ? lane - P.prefix_len - 1 : 0;
  const uint8_t pfx = (uint8_t)P.prefix[lane < P.prefix_len ? lane : 0];                       // d_strings keeps 16 bytes of slack
  const uint8_t nmb = X.first[mi_off + (ni < mi_len ? ni : mi_len)];                            // index mi_len is the tag's NUL
#if FGX_EMIT_FLAT
  {   // (nmb is the tag's NUL from lane name_len on: two one-level selects)
    const uint8_t colon_or_mi = lane == P.prefix_len ? (uint8_t)':' : nmb;
    R.nb = lane < P.prefix_len ? pfx : colon_or_mi;
  }
#else
  R.nb = lane < P.prefix_len ? pfx : lane == P.prefix_len ? (uint8_t)':' : lane < name_len ? nmb : (uint8_t)0;
#endif
  R.rgb = (uint8_t)P.rg[j3 < P.rg_len ? j3 : 0];
  R.mib = X.first[mi_off + (j3 < mi_len ? j3 : mi_len)];
  const uint8_t* fk = R.has_cb ? P.blob + uniform_u64(D.kept_off) + uni(D.cb_off) : X.first;
  R.cbb = fk[j3 < R.cb_len ? j3 : 0];
  R.rxb = (uint8_t)D.rx[j3 < FAST_RX_CAP ? j3 : 0];
}
__device__ __forceinline__ void emit_store(const EmitParams& P, const EndDesc& D, uint32_t lane, const EmitLoads& R) {
  const EmitCtx& X = R.X;
  if (R.generic) { emit_generic(P, D, X, lane); return; }
  const uint32_t Lc = X.Lc, name_len = X.name_len, mi_len = X.mi_len, seq_bytes = (Lc + 1) / 2, j3 = R.j3;
  const uint32_t so = R.so, qo = R.qo, cb_len = R.cb_len, rx_len = R.rx_len;
  const bool has_cb = R.has_cb, has_rx = R.has_rx;
  // ---- cD / cM / cE (vanilla_caller.rs:1800-1810): max / min depth, Σerrors / Σdepth as f32 ---------------------
  // every column counted once: a lane's low half is a duplicate when its offset was pulled back to the field's last dword
  uint32_t maxd = 0, mind = 0xFFFFFFFFu, sumd = 0, sume = 0;
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const uint32_t nat = 4 * (lane + 64 * t);
    const bool in = nat < 2 * Lc, lo_own = in && nat == R.ao[t];
    const uint32_t dl = R.dw[t] & 0xFFFF, dh = R.dw[t] >> 16, el = R.ew[t] & 0xFFFF, eh = R.ew[t] >> 16;
    if (lo_own) { maxd = dl > maxd ? dl : maxd; mind = dl < mind ? dl : mind; sumd += dl; sume += el; }
    if (in) { maxd = dh > maxd ? dh : maxd; mind = dh < mind ? dh : mind; sumd += dh; sume += eh; }
  }
  maxd = wave_max(maxd); mind = wave_min(mind); sumd = wave_sum(sumd); sume = wave_sum(sume);
  maxd = uni(maxd); mind = uni(mind); sumd = uni(sumd); sume = uni(sume);   // (every lane holds the totals: the tag widths below, and with them every later address, are scalar)
  const float ce_rate = sumd > 0 ? (float)sume / (float)sumd : 0.0f;
  const uint32_t n_cd = 3 + int_tag_width(maxd), n_cm = 3 + int_tag_width(mind);

  // ---- stores -----------------------------------------------------------------------------------------------------------
  uint8_t* q = X.q;
#if FGX_EMIT_FLAT
  // The small fields are chains of "lane k holds byte k" choices over wave-uniform values.  Written as nested conditionals they compile into
  // nested exec-mask regions (the kernel executed more scalar instructions than vector ones: 611 against 530 per family); written as below
  // — uniform words built by the scalar unit, a lane's byte taken with one shift, one-level selects — they are straight-line code.
  const uint32_t l3 = lane < 3u ? lane : 3u, sh3 = 8u * l3;                             // (a 24-bit header word >> sh3: its byte for lanes 0 - 2, 0 from lane 3 on)
  auto ztag = [&](uint32_t c3, uint32_t len, uint32_t body) -> uint32_t {               // byte `lane` of the tag  XY:Z:<len bytes> NUL
    const uint32_t u = (lane - 3u < len) ? body : 0u;                                    // (unsigned: false for lanes 0 - 2)
    return (c3 >> sh3) | u;
  };
  {   // block_size + fixed core: ref_id -1, pos -1, l_read_name, mapq 0, bin 4680, n_cigar_op 0, flag, l_seq, next_ref -1, next_pos -1, tlen 0
    uint32_t v = 0xFFFFFFFFu;
    if (lane == 0) v = X.rec_size;
    if (lane == 3) v = (name_len + 1) | (4680u << 16);
    if (lane == 4) v = X.flag << 16;
    if (lane == 5) v = Lc;
    if (lane == 8) v = 0u;
    if (lane < 9) gst32u(q + 4 * lane, v);
  }
#else
  if (lane < 9) {   // block_size + fixed core: ref_id -1, pos -1, l_read_name, mapq 0, bin 4680, n_cigar_op 0, flag, l_seq, next_ref -1, next_pos -1, tlen 0
    const uint32_t v = lane == 0 ? X.rec_size : lane == 3 ? ((name_len + 1) | (4680u << 16)) : lane == 4 ? (X.flag << 16) : lane == 5 ? Lc : lane == 8 ? 0u : 0xFFFFFFFFu;
    gst32u(q + 4 * lane, v);
  }
#endif
  q += 36;
  if (lane < name_len + 1) q[lane] = R.nb;
  q += name_len + 1;
  if (4 * lane < seq_bytes) {   // eight columns → four bytes, high nibble first; a column past the end packs as 0
    const uint32_t c0 = 2 * so;
    uint32_t lo4 = R.cw.x, hi4 = R.cw.y;                                               // codes of columns c0..c0+3 / c0+4..c0+7, one byte each
    if (c0 + 7 >= Lc) hi4 &= 0x00FFFFFFu;                                              // (only column c0 + 7 can be past the end: Lc odd)
    const uint32_t b0 = ((lo4 << 4) | (lo4 >> 8)) & 0xFF, b1 = ((lo4 >> 12) | (lo4 >> 24)) & 0xFF;
    const uint32_t b2 = ((hi4 << 4) | (hi4 >> 8)) & 0xFF, b3 = ((hi4 >> 12) | (hi4 >> 24)) & 0xFF;
    gst32u(q + so, b0 | (b1 << 8) | (b2 << 16) | (b3 << 24));
  }
  q += seq_bytes;
  if (4 * lane < Lc) gst32u(q + qo, R.qw);
  q += Lc;
#if FGX_EMIT_FLAT
  {
    const uint32_t b = ztag('R' | ('G' << 8) | ('Z' << 16), P.rg_len, R.rgb);
    if (lane < 3 + P.rg_len + 1) q[lane] = (uint8_t)b;
  }
  q += 3 + P.rg_len + 1;
  {   // cD cM cE and, when asked for, the header of the cd array right behind them: one store.  Four uniform 64-bit words (an integer tag
      // is `ab` + its type + one or two value bytes, cE is `cEf` + the four bytes of the rate, the array header `cdBs` + its count)
    const uint32_t n3 = n_cd + n_cm + 7, nh = P.per_base_tags ? 8u : 0u;
    auto int_word = [](uint32_t a, uint32_t b, uint32_t v) -> unsigned long long {
      const uint32_t ty = v <= 127 ? (uint32_t)'c' : v <= 255 ? (uint32_t)'C' : (uint32_t)'S';
      return (unsigned long long)(a | (b << 8) | (ty << 16)) | ((unsigned long long)v << 24);
    };
    const unsigned long long w_cd = int_word('c', 'D', maxd), w_cm = int_word('c', 'M', mind);
    const unsigned long long w_ce = (unsigned long long)('c' | ('E' << 8) | ('f' << 16)) | ((unsigned long long)__float_as_uint(ce_rate) << 24);
    const unsigned long long w_hd = (unsigned long long)('c' | ('d' << 8) | ('B' << 16) | ('s' << 24)) | ((unsigned long long)Lc << 32);
    unsigned long long w = w_hd;
    uint32_t k = lane - n3;
    if (lane < n3) { w = w_ce; k = lane - n_cd - n_cm; }
    if (lane < n_cd + n_cm) { w = w_cm; k = lane - n_cd; }
    if (lane < n_cd) { w = w_cd; k = lane; }
    const uint32_t b = (uint32_t)(w >> (8u * (k & 7u)));
    if (lane < n3 + nh) q[lane] = (uint8_t)b;
    q += n3;
  }
#else
  if (lane < 3 + P.rg_len + 1) q[lane] = lane == 0 ? 'R' : lane == 1 ? 'G' : lane == 2 ? 'Z' : j3 < P.rg_len ? R.rgb : (uint8_t)0;
  q += 3 + P.rg_len + 1;
  {   // cD cM cE and, when asked for, the header of the cd array right behind them: one store
    const uint32_t n3 = n_cd + n_cm + 7, nh = P.per_base_tags ? 8u : 0u;
    if (lane < n3 + nh) {
      const uint32_t i = lane - n3;
      q[lane] = lane < n3 ? cdcmce_byte(lane, n_cd, n_cm, maxd, mind, ce_rate)
                          : (uint8_t)(i == 0 ? 'c' : i == 1 ? 'd' : i == 2 ? 'B' : i == 3 ? 's' : (Lc >> (8 * (i - 4))));
    }
    q += n3;
  }
#endif
  if (P.per_base_tags) {
    q += 8;
#pragma unroll
    for (int t = 0; t < 2; t++) if (4 * (lane + 64 * t) < 2 * Lc) gst32u(q + R.ao[t], R.dw[t]);
    q += 2 * Lc;
    if (lane < 2) gst32u(q + 4 * lane, lane == 0 ? ('c' | ('e' << 8) | ('B' << 16) | ('s' << 24)) : Lc);
    q += 8;
#pragma unroll
    for (int t = 0; t < 2; t++) if (4 * (lane + 64 * t) < 2 * Lc) gst32u(q + R.ao[t], R.ew[t]);
    q += 2 * Lc;
  }
#if FGX_EMIT_FLAT
  {
    const uint32_t b = ztag((uint32_t)(uint8_t)P.tag0 | ((uint32_t)(uint8_t)P.tag1 << 8) | ('Z' << 16), mi_len, R.mib);
    if (lane < 3 + mi_len + 1) q[lane] = (uint8_t)b;
  }
  q += 3 + mi_len + 1;
  if (has_cb) {
    const uint32_t b = ztag((uint32_t)(uint8_t)P.cell0 | ((uint32_t)(uint8_t)P.cell1 << 8) | ('Z' << 16), cb_len, R.cbb);
    if (lane < 3 + cb_len + 1) q[lane] = (uint8_t)b;
    q += 3 + cb_len + 1;
  }
  if (has_rx) {
    const uint32_t b = ztag('R' | ('X' << 8) | ('Z' << 16), rx_len, R.rxb);
    if (lane < 3 + rx_len + 1) q[lane] = (uint8_t)b;
  }
  (void)j3;
#else
  if (lane < 3 + mi_len + 1) q[lane] = lane == 0 ? (uint8_t)P.tag0 : lane == 1 ? (uint8_t)P.tag1 : lane == 2 ? 'Z' : j3 < mi_len ? R.mib : (uint8_t)0;
  q += 3 + mi_len + 1;
  if (has_cb) { if (lane < 3 + cb_len + 1) q[lane] = lane == 0 ? (uint8_t)P.cell0 : lane == 1 ? (uint8_t)P.cell1 : lane == 2 ? 'Z' : j3 < cb_len ? R.cbb : (uint8_t)0; q += 3 + cb_len + 1; }
  if (has_rx) { if (lane < 3 + rx_len + 1) q[lane] = lane == 0 ? 'R' : lane == 1 ? 'X' : lane == 2 ? 'Z' : j3 < rx_len ? R.rxb : (uint8_t)0; }
#endif
}

// ---- round 6: BOTH records of a pair family at once — record R1 in lanes 0 - 31, record R2 in lanes 32 - 63 -----------------------------------
// k_emit executed 450 vector + 357 scalar instructions per family for two records written one after the other (profiles/r05y_pmc_5M_families.json):
// most of them for the small fields (a store per field, a lane per byte, a dozen of 64 lanes busy) and for wave-uniform address arithmetic that is
// redone per record.  Here the two records share every instruction.  A lane takes EIGHT columns of its half's record — 8 B of codes -> 4 B of
// packed bases, 8 B of qualities, 16 B each of cd / ce: 19 lanes of a half for 150 columns, the group of the last lane pulled back so that it ends
// with the record (an overlapping store of the same values; its duplicate columns are masked out of the sums) —, the small fields are a lane per
// byte of the half's record, and every address is ONE scalar base (record R1, the scratch columns of R1) + a 32-bit lane offset: R2 follows R1
// in the output, and its columns follow R1's in the scratch.  cD / cM / cE: the four reductions stop at the half (five DPP steps).
// Taken when both records are there, 8 <= Lc <= 256 and every string field fits 32 lanes; returns false (nothing touched) otherwise.
#ifndef FGX_EMIT_PAIR
#define FGX_EMIT_PAIR 1
#endif
typedef unsigned short em_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t em_pkmax(uint32_t a, uint32_t b) { em_u16x2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = __builtin_elementwise_max(x, y); uint32_t r; __builtin_memcpy(&r, &x, 4); return r; }
__device__ __forceinline__ uint32_t em_pkmin(uint32_t a, uint32_t b) { em_u16x2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = __builtin_elementwise_min(x, y); uint32_t r; __builtin_memcpy(&r, &x, 4); return r; }
__device__ __forceinline__ uint32_t em_sum2(uint32_t pair, uint32_t acc) { em_u16x2 x; const em_u16x2 one = {1, 1}; __builtin_memcpy(&x, &pair, 4); return __builtin_amdgcn_udot2(x, one, acc, false); }   // acc + both 16-bit halves
// reductions over each HALF of the wavefront: the four row-local steps of FGX_WAVE_REDUCE, then row_bcast:15 into rows 1 and 3 — lane 31 holds the
// result of lanes 0 - 31, lane 63 that of lanes 32 - 63
#define FGX_HALF_REDUCE(v, idn, OP) do { \
    v = OP(v, wave_dpp<0xB1, 0xF>(idn, v)); v = OP(v, wave_dpp<0x4E, 0xF>(idn, v)); v = OP(v, wave_dpp<0x141, 0xF>(idn, v)); v = OP(v, wave_dpp<0x140, 0xF>(idn, v)); \
    v = OP(v, wave_dpp<0x142, 0xA>(idn, v)); } while (0)
__device__ __forceinline__ bool emit_pair(const EmitParams& P, const EndDesc* D /* [3]: slots F, R1, R2 in LDS */, uint64_t oo1, uint64_t oo2, uint32_t lane) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const EndDesc& D1 = D[1];
  const EndDesc& D2 = D[2];
  const uint32_t Lc1 = uni(D1.cons_len), Lc2 = uni(D2.cons_len);
  const uint32_t mi_len = uni(D1.mi_len), mi_off = uni(D1.mi_off);
  const uint64_t first_off = uniform_u64(D1.first_off);
  const uint64_t col1 = uniform_u64(D1.col_off), col2 = uniform_u64(D2.col_off);
  const bool hcb1 = uni(D1.has_cb) != 0, hcb2 = uni(D2.has_cb) != 0, hrx1 = uni(D1.has_rx) != 0, hrx2 = uni(D2.has_rx) != 0;
  const uint32_t cbl1 = hcb1 ? uni(D1.cb_len) : 0u, cbl2 = hcb2 ? uni(D2.cb_len) : 0u, rxl1 = hrx1 ? uni(D1.rx_len) : 0u, rxl2 = hrx2 ? uni(D2.rx_len) : 0u;
  const uint32_t name_len = P.prefix_len + 1u + mi_len, rg_len = P.rg_len;
  const bool ok = Lc1 >= 8u && Lc1 <= 256u && Lc2 >= 8u && Lc2 <= 256u && name_len + 1u <= 32u && rg_len + 4u <= 32u && mi_len + 4u <= 32u &&
                  cbl1 + 4u <= 32u && cbl2 + 4u <= 32u && rxl1 + 4u <= 32u && rxl2 + 4u <= 32u && rxl1 <= (uint32_t)FAST_RX_CAP && rxl2 <= (uint32_t)FAST_RX_CAP &&
                  uniform_u64(D2.first_off) == first_off && uni(D2.mi_len) == mi_len && uni(D2.mi_off) == mi_off &&
                  col2 >= col1 && col2 - col1 < (1ull << 30) && oo2 > oo1 && oo2 - oo1 < (1ull << 30);
  if (!ok) return false;
  const uint32_t l = lane & 31u;
  const bool hb = lane >= 32u;
  const uint32_t Lc = hb ? Lc2 : Lc1;
  const uint32_t dcol = hb ? (uint32_t)(col2 - col1) : 0u, dq = hb ? (uint32_t)(oo2 - oo1) : 0u;
  const uint32_t ty1 = uni(D1.type), ty2 = uni(D2.type), rs1 = uni(D1.rec_size), rs2 = uni(D2.rec_size);   // (uniform reads by every lane, then the half's pick)
  const uint32_t d_type = hb ? ty2 : ty1, rec_size = hb ? rs2 : rs1;
  const bool hcb = hb ? hcb2 : hcb1, hrx = hb ? hrx2 : hrx1;
  const uint32_t cb_len = hb ? cbl2 : cbl1, rx_len = hb ? rxl2 : rxl1;
  // ---- loads: everything the two records read, before the first store ---------------------------------------------------------------------
  const uint32_t n0 = 8u * l;
  const bool pay = n0 < Lc;                                                             // this lane holds columns of its record
  const uint32_t c0 = min(n0, Lc - 8u);                                                 // first column of the lane's group (the last group is pulled back)
  const uint32_t cs = min(n0, ((Lc + 1u) & ~1u) - 8u);                                  // ... of its group of packed bases: even (column Lc may be read: slack)
  const uint8_t* const code = P.col_code + col1;
  const uint8_t* const cq = P.col_qual + col1;
  const uint8_t* const cdb = (const uint8_t*)(P.col_depth + col1);
  const uint8_t* const ceb = (const uint8_t*)(P.col_err + col1);
  const uint2 cw = gld64u(code + (dcol + cs));
  const uint2 qw = gld64u(cq + (dcol + c0));
  u32x4 dv, ev;
  __builtin_memcpy(&dv, cdb + 2u * (dcol + c0), 16);
  __builtin_memcpy(&ev, ceb + 2u * (dcol + c0), 16);
  const uint8_t* const first = P.blob + first_off;
  const uint32_t j3 = l >= 3u ? l - 3u : 0u;
  const uint32_t ni = l > P.prefix_len ? l - P.prefix_len - 1u : 0u;
  const uint8_t pfx = (uint8_t)P.prefix[l < P.prefix_len ? l : 0u];                     // d_strings keeps 16 bytes of slack
  const uint8_t nmb = first[mi_off + (ni < mi_len ? ni : mi_len)];                      // index mi_len is the tag's NUL
  const uint8_t rgb = (uint8_t)P.rg[j3 < rg_len ? j3 : 0u];
  const uint8_t mib = first[mi_off + (j3 < mi_len ? j3 : mi_len)];
  uint8_t cbb = 0;
  if (hcb1 || hcb2) {                                                                   // (wave-uniform; the half without the tag reads byte 0 of the family's first record)
    const uint64_t k1 = hcb1 ? uniform_u64(D1.kept_off) + uni(D1.cb_off) : first_off, k2 = hcb2 ? uniform_u64(D2.kept_off) + uni(D2.cb_off) : first_off;
    const uint8_t* const fk = P.blob + (hb ? k2 : k1);
    cbb = fk[j3 < cb_len ? j3 : 0u];
  }
  const uint8_t rxb = (uint8_t)D[hb ? 2 : 1].rx[j3 < (uint32_t)FAST_RX_CAP ? j3 : 0u];
  // ---- cD / cM / cE (vanilla_caller.rs:1800-1810): max / min depth, sum of errors / sum of depths as f32, per half -------------------------
  uint32_t maxd, mind, sumd = 0, sume = 0;
  {
    const uint32_t skip16 = 16u * (n0 - c0);                                            // bits of duplicate columns at the low end of a pulled-back group
    const uint32_t d4[4] = {dv.x, dv.y, dv.z, dv.w}, e4[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int32_t sb = (int32_t)skip16 - 32 * k;
      const uint32_t m = sb <= 0 ? 0xFFFFFFFFu : sb == 16 ? 0xFFFF0000u : 0u;
      sumd = em_sum2(d4[k] & m, sumd); sume = em_sum2(e4[k] & m, sume);
    }
    const uint32_t mx = em_pkmax(em_pkmax(d4[0], d4[1]), em_pkmax(d4[2], d4[3])), mn = em_pkmin(em_pkmin(d4[0], d4[1]), em_pkmin(d4[2], d4[3]));   // (duplicates do not move a maximum)
    maxd = max(mx & 0xFFFFu, mx >> 16); mind = min(mn & 0xFFFFu, mn >> 16);
    if (!pay) { maxd = 0u; mind = 0xFFFFFFFFu; sumd = 0u; sume = 0u; }
  }
  FGX_HALF_REDUCE(maxd, 0u, wr_max); FGX_HALF_REDUCE(mind, 0xFFFFFFFFu, wr_min); FGX_HALF_REDUCE(sumd, 0u, wr_add); FGX_HALF_REDUCE(sume, 0u, wr_add);
  {   // (lane 31 / lane 63 hold the halves' results; every lane reads both)
    const uint32_t mx0 = rlane(maxd, 31), mx1 = rlane(maxd, 63), mn0 = rlane(mind, 31), mn1 = rlane(mind, 63);
    const uint32_t sd0 = rlane(sumd, 31), sd1 = rlane(sumd, 63), se0 = rlane(sume, 31), se1 = rlane(sume, 63);
    maxd = hb ? mx1 : mx0; mind = hb ? mn1 : mn0; sumd = hb ? sd1 : sd0; sume = hb ? se1 : se0;
  }
  const float ce_rate = sumd > 0u ? (float)sume / (float)sumd : 0.0f;
  const uint32_t n_cd = 3u + int_tag_width(maxd), n_cm = 3u + int_tag_width(mind);
  // ---- where the fields of the half's record lie (offsets from record R1's first byte) ----------------------------------------------------
  const uint32_t seq_bytes = (Lc + 1u) >> 1;
  const uint32_t o_seq = dq + 36u + name_len + 1u, o_qual = o_seq + seq_bytes, o_rg = o_qual + Lc, o_t3 = o_rg + 3u + rg_len + 1u;
  const uint32_t n3 = n_cd + n_cm + 7u, nh = P.per_base_tags ? 8u : 0u;
  const uint32_t o_cd = o_t3 + n3 + 8u, o_ceh = o_cd + 2u * Lc, o_ce = o_ceh + 8u;
  const uint32_t o_mi = P.per_base_tags ? o_ce + 2u * Lc : o_t3 + n3;
  const uint32_t o_cb = o_mi + 3u + mi_len + 1u, o_rx = o_cb + (hcb ? 3u + cb_len + 1u : 0u);
  uint8_t* const q = P.out + (oo1 - P.out_base);
  const uint32_t l3 = l < 3u ? l : 3u, sh3 = 8u * l3;                                   // (a 24-bit header word >> sh3: its byte for lanes 0 - 2 of the half, 0 from lane 3 on)
  auto ztag = [&](uint32_t c3, uint32_t len, uint32_t body) -> uint32_t { const uint32_t u = (l - 3u < len) ? body : 0u; return (c3 >> sh3) | u; };   // byte l of  XY:Z:<len bytes> NUL
  // ---- stores ---------------------------------------------------------------------------------------------------------------------------------
  {   // block_size + fixed core: ref_id -1, pos -1, l_read_name, mapq 0, bin 4680, n_cigar_op 0, flag, l_seq, next_ref -1, next_pos -1, tlen 0
    uint32_t flag = bam::F_UNMAPPED;
    if (d_type == 1u) flag |= bam::F_PAIRED | bam::F_FIRST | bam::F_MATE_UNMAPPED;
    else if (d_type == 2u) flag |= bam::F_PAIRED | bam::F_LAST | bam::F_MATE_UNMAPPED;
    uint32_t v = 0xFFFFFFFFu;
    if (l == 0u) v = rec_size;
    if (l == 3u) v = (name_len + 1u) | (4680u << 16);
    if (l == 4u) v = flag << 16;
    if (l == 5u) v = Lc;
    if (l == 8u) v = 0u;
    if (l < 9u) gst32u(q + (dq + 4u * l), v);
  }
  if (l < name_len + 1u) {
    const uint8_t colon_or_mi = l == P.prefix_len ? (uint8_t)':' : nmb;                 // (nmb is the tag's NUL from lane name_len on)
    q[dq + 36u + l] = l < P.prefix_len ? pfx : colon_or_mi;
  }
  if (pay) {
    // eight columns -> four bytes, high nibble first; a column past the end packs as 0 (only column cs + 7 can be: Lc odd)
    uint32_t lo4 = cw.x, hi4 = cw.y;
    if (cs + 7u >= Lc) hi4 &= 0x00FFFFFFu;
    const uint32_t t = (lo4 << 4) | (lo4 >> 8), u = (hi4 << 4) | (hi4 >> 8);            // bytes 0 and 2: (code << 4) | next code
    gst32u(q + (o_seq + (cs >> 1)), __builtin_amdgcn_perm(u, t, 0x06040200u));
    __builtin_memcpy(q + (o_qual + c0), &qw, 8);
  }
  {
    const uint32_t b = ztag('R' | ('G' << 8) | ('Z' << 16), rg_len, rgb);
    if (l < 3u + rg_len + 1u) q[o_rg + l] = (uint8_t)b;
  }
  {   // cD cM cE and, when asked for, the header of the cd array right behind them: one store
    auto int_word = [](uint32_t a, uint32_t b, uint32_t v) -> unsigned long long {
      const uint32_t ty = v <= 127u ? (uint32_t)'c' : v <= 255u ? (uint32_t)'C' : (uint32_t)'S';
      return (unsigned long long)(a | (b << 8) | (ty << 16)) | ((unsigned long long)v << 24);
    };
    const unsigned long long w_cd = int_word('c', 'D', maxd), w_cm = int_word('c', 'M', mind);
    const unsigned long long w_ce = (unsigned long long)('c' | ('E' << 8) | ('f' << 16)) | ((unsigned long long)__float_as_uint(ce_rate) << 24);
    const unsigned long long w_hd = (unsigned long long)('c' | ('d' << 8) | ('B' << 16) | ('s' << 24)) | ((unsigned long long)Lc << 32);
    unsigned long long w = w_hd;
    uint32_t k = l - n3;
    if (l < n3) { w = w_ce; k = l - n_cd - n_cm; }
    if (l < n_cd + n_cm) { w = w_cm; k = l - n_cd; }
    if (l < n_cd) { w = w_cd; k = l; }
    const uint32_t b = (uint32_t)(w >> (8u * (k & 7u)));
    if (l < n3 + nh) q[o_t3 + l] = (uint8_t)b;
  }
  if (P.per_base_tags) {
    if (pay) __builtin_memcpy(q + (o_cd + 2u * c0), &dv, 16);
    if (l < 2u) gst32u(q + (o_ceh + 4u * l), l == 0u ? ('c' | ('e' << 8) | ('B' << 16) | ('s' << 24)) : Lc);
    if (pay) __builtin_memcpy(q + (o_ce + 2u * c0), &ev, 16);
  }
  {
    const uint32_t b = ztag((uint32_t)(uint8_t)P.tag0 | ((uint32_t)(uint8_t)P.tag1 << 8) | ('Z' << 16), mi_len, mib);
    if (l < 3u + mi_len + 1u) q[o_mi + l] = (uint8_t)b;
  }
  if (hcb1 || hcb2) {
    const uint32_t b = ztag((uint32_t)(uint8_t)P.cell0 | ((uint32_t)(uint8_t)P.cell1 << 8) | ('Z' << 16), cb_len, cbb);
    if (hcb && l < 3u + cb_len + 1u) q[o_cb + l] = (uint8_t)b;
  }
  if (hrx1 || hrx2) {
    const uint32_t b = ztag('R' | ('X' << 8) | ('Z' << 16), rx_len, rxb);
    if (hrx && l < 3u + rx_len + 1u) q[o_rx + l] = (uint8_t)b;
  }
  return true;
}
#ifndef FGX_EMIT_OCC
#define FGX_EMIT_OCC 7   /* wavefronts per SIMD the register allocation of k_emit aims at */
#endif
// One wavefront per FAMILY: its (up to three) records.  A third of the slots is empty on paired data (the fragment slot), and a
// wavefront that only finds `valid == 0` still costs a launch and a memory round trip; the descriptor carries blob OFFSETS, so the
// record's strings are one dependent load away instead of two.
// (Assembling the record through LDS — whole-record image with byte writes, or dword-staged column arrays with dword payload
// copies — was measured three times, rounds 1 and 2: 3.4 – 4.0 ms against 2.2 ms per 2 M records; registers-only streaming it is.)
__global__ __launch_bounds__(256, FGX_EMIT_OCC) void k_emit(EmitParams P) {
  // the family's (up to) three descriptors, copied once with 16-byte loads: every field read below is an LDS read — as global
  // loads, the valid flags and then each record's fields were dependent memory round trips of their own
  __shared__ __align__(16) EndDesc sD[4][3];
  static_assert(sizeof(EndDesc) == 96, "EndDesc is copied as six 16-byte pieces");
  uint32_t fam = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const uint32_t lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
  if (P.fam_list) {   // the merge of a direct-records batch: only the families that left the split pipeline have descriptors
    if (fam >= P.n_fam) return;
    fam = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.fam_list[fam]);
  }
  const uint32_t s0 = P.slot0 + 3 * fam;
  if (s0 >= P.slot_end) return;
  // (every kernel argument the records need is asked for here, in one batch: fetched where first used they were five separate
  // scalar-load round trips along the way)
#if defined(FGX_WAVEMU)
#define K_EMIT_PIN(x) ((void)(x))
#else
#define K_EMIT_PIN(x) asm volatile("" :: "s"(x))
#endif
  K_EMIT_PIN(P.blob); K_EMIT_PIN(P.out); K_EMIT_PIN(P.out_base); K_EMIT_PIN(P.col_code); K_EMIT_PIN(P.col_qual); K_EMIT_PIN(P.col_depth);
  K_EMIT_PIN(P.col_err); K_EMIT_PIN(P.prefix); K_EMIT_PIN(P.prefix_len); K_EMIT_PIN(P.rg); K_EMIT_PIN(P.rg_len);
#undef K_EMIT_PIN
  const uint32_t ns = P.slot_end - s0 < 3 ? P.slot_end - s0 : 3;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 piece = ((const u32x4*)&P.ends[s0])[lane < 6 * ns ? lane : 0u];
  const uint64_t oo = P.out_off[s0 + (lane < ns ? lane : 0u)];   // (both loads leave before either is waited for)
  if (lane < 6 * ns) ((u32x4*)&sD[wv][0])[lane] = piece;
  wave_sync();
  const bool v0 = uni(sD[wv][0].valid) != 0, v1 = ns > 1 && uni(sD[wv][1].valid) != 0, v2 = ns > 2 && uni(sD[wv][2].valid) != 0;
  // (the record's output offset as a SCALAR: with it in a vector register every store of the record formed a 64-bit vector
  // address of its own — a third of the kernel's vector instructions were address adds and register moves)
  auto off_of = [&](int k) -> uint64_t {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(oo >> 32), k) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)oo, k);
  };
  if (v0) { EmitLoads R0; emit_load(P, sD[wv][0], off_of(0), lane, R0); emit_store(P, sD[wv][0], lane, R0); }   // a fragment record: families of single reads
#if FGX_EMIT_PAIR
  if (!v0 && v1 && v2 && emit_pair(P, &sD[wv][0], off_of(1), off_of(2), lane)) return;   // (round 6) the usual pair family: both records side by side in the halves of the wavefront
#endif
  // the two records of a pair family: both records' loads, then both records' stores
  EmitLoads R1, R2;
  if (v1) emit_load(P, sD[wv][1], off_of(1), lane, R1);
  if (v2) emit_load(P, sD[wv][2], off_of(2), lane, R2);
  if (v1) emit_store(P, sD[wv][1], lane, R1);
  if (v2) emit_store(P, sD[wv][2], lane, R2);
}

// -----------------------------------------------------------------------------------------------------
// k_emit_duplex — one wavefront per duplex consensus record: the A/B strand combine of duplex_consensus
// (duplex_caller.rs:931-1108) over the two single-strand column segments, then the record of duplex_read_into
// (:1118-1405): tags MI [CB] RG aD aE aM [ac ad ae aq] bD bE bM [bc bd be bq] cD cE cM RX.
// -----------------------------------------------------------------------------------------------------
// records the fast writers (k_emit_duplex_fast / k_emit_codec_fast, below) take; the per-field kernels skip them
constexpr uint32_t DUP_SLOTS = 2;     // duplex: up to 256 positions

__device__ __forceinline__ bool small_tags(uint32_t name_len, uint32_t mi_len, uint32_t cb_len, uint32_t rg_len, uint32_t rx_len) {
  return name_len + 1 <= 64 && mi_len + 4 <= 64 && cb_len + 4 <= 64 && rg_len + 4 <= 64 && rx_len + 4 <= 64;
}
__device__ __forceinline__ bool duplex_fast_ok(const DuplexDesc& D, uint32_t prefix_len, uint32_t rg_len) {
  return D.len <= 128 * DUP_SLOTS && small_tags(prefix_len + 1 + D.mi_len, D.mi_len, D.has_cb ? D.cb_len : 0, rg_len, D.has_rx ? D.rx_len : 0);
}
__device__ __forceinline__ bool codec_fast_ok(const CodecDesc& D, uint32_t prefix_len, uint32_t rg_len) {
  return small_tags(prefix_len + 1 + D.mi_len, D.mi_len, D.has_cb ? D.cb_len : 0, rg_len, D.has_rx ? D.rx_len : 0);     // any length: written in windows of 256 positions
}

struct DCol { uint32_t ca, qa, ea, da, cb, qb, eb, db, oc, oq, oe; };
__device__ __forceinline__ uint32_t obs_sum(uint32_t o) { return (o & 0xFF) + ((o >> 8) & 0xFF) + ((o >> 16) & 0xFF) + (o >> 24); }
__device__ __forceinline__ uint32_t obs_of_code(uint32_t o, uint32_t code) {   // count of the base with 4-bit code 1/2/4/8
  return code == 1 ? (o & 0xFF) : code == 2 ? ((o >> 8) & 0xFF) : code == 4 ? ((o >> 16) & 0xFF) : code == 8 ? (o >> 24) : 0u;
}
__device__ __forceinline__ uint32_t cap_q(int32_t v) { return v < 2 ? 2u : v > 93 ? 93u : (uint32_t)v; }

__global__ __launch_bounds__(256) void k_emit_duplex(DuplexEmitParams P) {
  const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(P.slot0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
  const uint32_t lane = threadIdx.x & 63;
  if (slot >= P.slot_end) return;
  const DuplexDesc& D = P.ends[slot];
  if (!D.valid || duplex_fast_ok(D, P.prefix_len, P.rg_len)) return;     // k_emit_duplex_fast writes those
  uint8_t* q = P.out + P.out_off[slot];
  const uint32_t L = D.len;
  const bool has_ba = D.has_ba != 0;
  const uint64_t a_off = D.a_off, b_off = D.b_off;
  const uint8_t* first = P.blob + P.rec_off[D.first_rec];
  const uint32_t mi_len = D.mi_len, mi_off = D.mi_off, name_len = P.prefix_len + 1 + mi_len;
  const bool has_cb = D.has_cb != 0, has_rx = D.has_rx != 0;
  const uint32_t cb_len = has_cb ? D.cb_len : 0, rx_len = has_rx ? D.rx_len : 0;
  uint32_t flag = bam::F_UNMAPPED | bam::F_PAIRED | bam::F_MATE_UNMAPPED | (D.type == 1 ? bam::F_FIRST : bam::F_LAST);
  // one position of the record: both strands' single-strand calls and the duplex call
  auto col = [&](uint32_t i) {
    DCol c;
    const uint32_t oa = P.col_obs[a_off + i];
    c.ca = P.col_code[a_off + i]; c.qa = P.col_qual[a_off + i]; c.ea = P.col_err[a_off + i]; c.da = obs_sum(oa);
    if (!has_ba) { c.cb = 15; c.qb = 0; c.eb = 0; c.db = 0; c.oc = c.ca; c.oq = c.qa; c.oe = c.ea; return c; }
    const uint32_t ob = P.col_obs[b_off + i];
    c.cb = P.col_code[b_off + i]; c.qb = P.col_qual[b_off + i]; c.eb = P.col_err[b_off + i]; c.db = obs_sum(ob);
    uint32_t rb, rq;
    if (c.ca == c.cb) { rb = c.ca; rq = cap_q((int32_t)c.qa + (int32_t)c.qb); }
    else if (c.qa > c.qb) { rb = c.ca; rq = cap_q((int32_t)c.qa - (int32_t)c.qb); }
    else if (c.qb > c.qa) { rb = c.cb; rq = cap_q((int32_t)c.qb - (int32_t)c.qa); }
    else { rb = c.ca; rq = FGX_MIN_PHRED; }
    const bool nocall = c.ca == 15 || c.cb == 15 || rq == FGX_MIN_PHRED;
    c.oc = nocall ? 15u : rb; c.oq = nocall ? (uint32_t)FGX_MIN_PHRED : rq;
    // errors: source reads of both strands that disagree with the raw duplex base (N never counts)
    const uint32_t agree = obs_of_code(oa, rb) + obs_of_code(ob, rb);
    c.oe = rb == 15 ? 0u : (c.da + c.db) - agree;
    return c;
  };
  // ---- reductions for aD aM aE / bD bM bE / cD cM cE -------------------------------------------------------------------------
  uint32_t amax = 0, amin = 0xFFFFFFFFu, asd = 0, ase = 0, bmax = 0, bmin = 0xFFFFFFFFu, bsd = 0, bse = 0, cmax = 0, cmin = 0xFFFFFFFFu, csd = 0, cse = 0;
  for (uint32_t i = lane; i < L; i += 64) {
    const DCol c = col(i);
    amax = c.da > amax ? c.da : amax; amin = c.da < amin ? c.da : amin; asd += c.da; ase += c.ea;
    bmax = c.db > bmax ? c.db : bmax; bmin = c.db < bmin ? c.db : bmin; bsd += c.db; bse += c.eb;
    const uint32_t t = c.da + c.db;
    cmax = t > cmax ? t : cmax; cmin = t < cmin ? t : cmin; csd += t; cse += c.oe;
  }
  amax = wave_max(amax); amin = wave_min(amin); asd = wave_sum(asd); ase = wave_sum(ase);
  bmax = wave_max(bmax); bmin = wave_min(bmin); bsd = wave_sum(bsd); bse = wave_sum(bse);
  cmax = wave_max(cmax); cmin = wave_min(cmin); csd = wave_sum(csd); cse = wave_sum(cse);
  if (L == 0) { amin = 0; bmin = 0; cmin = 0; }
  if (!has_ba) { bmax = 0; bmin = 0; bsd = 0; bse = 0; }
  const float a_rate = asd ? (float)ase / (float)asd : 0.0f, b_rate = bsd ? (float)bse / (float)bsd : 0.0f, c_rate = csd ? (float)cse / (float)csd : 0.0f;

  // ---- block_size + core, name, bases, quals ------------------------------------------------------------------------------------
  if (lane < 36) {
    const uint32_t dw = lane >> 2;
    const uint32_t v = dw == 0 ? D.rec_size : dw == 3 ? ((name_len + 1) | (4680u << 16)) : dw == 4 ? (flag << 16) : dw == 5 ? L : dw == 8 ? 0u : 0xFFFFFFFFu;
    q[lane] = (uint8_t)(v >> (8 * (lane & 3)));
  }
  q += 36;
  for (uint32_t i = lane; i < name_len + 1; i += 64)
    q[i] = i < P.prefix_len ? (uint8_t)P.prefix[i] : i == P.prefix_len ? (uint8_t)':' : i < name_len ? first[mi_off + (i - P.prefix_len - 1)] : (uint8_t)0;
  q += name_len + 1;
  for (uint32_t i = lane; i < (L + 1) / 2; i += 64) {
    const uint32_t hi = col(2 * i).oc, lo = 2 * i + 1 < L ? col(2 * i + 1).oc : 0u;
    q[i] = (uint8_t)((hi << 4) | lo);
  }
  q += (L + 1) / 2;
  for (uint32_t i = lane; i < L; i += 64) q[i] = (uint8_t)col(i).oq;
  q += L;
  // ---- tags -----------------------------------------------------------------------------------------------------------------------
  auto z_tag = [&](char t0, char t1, const uint8_t* src, uint32_t n) {     // Z tag from a byte string in global memory
    for (uint32_t i = lane; i < 3 + n + 1; i += 64) q[i] = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'Z' : i - 3 < n ? src[i - 3] : (uint8_t)0;
    q += 3 + n + 1;
  };
  auto scalar_tags = [&](char s, uint32_t dmax, float rate, uint32_t dmin) {   // <s>D int, <s>E float, <s>M int: 4 + 7 + 4 bytes
    if (lane < 15) {
      uint8_t v;
      if (lane < 4) v = int_tag_byte(lane, s, 'D', dmax);
      else if (lane < 11) { const uint32_t j = lane - 4, u = __float_as_uint(rate); v = j == 0 ? (uint8_t)s : j == 1 ? (uint8_t)'E' : j == 2 ? (uint8_t)'f' : (uint8_t)(u >> (8 * (j - 3))); }
      else v = int_tag_byte(lane - 11, s, 'M', dmin);
      q[lane] = v;
    }
    q += 15;
  };
  auto per_base = [&](char s, bool b_side) {      // <s>c bases, <s>d depths, <s>e errors, <s>q quals of one strand
    for (uint32_t i = lane; i < 3 + L + 1; i += 64) {
      uint8_t v = i == 0 ? (uint8_t)s : i == 1 ? (uint8_t)'c' : i == 2 ? (uint8_t)'Z' : (uint8_t)0;
      if (i >= 3 && i - 3 < L) { const DCol c = col(i - 3); v = bam::code_to_ascii((uint8_t)(b_side ? c.cb : c.ca)); }
      q[i] = v;
    }
    q += 3 + L + 1;
    for (int pass = 0; pass < 2; pass++) {
      for (uint32_t i = lane; i < 8 + 2 * L; i += 64) {
        uint8_t v;
        if (i < 8) v = i == 0 ? (uint8_t)s : i == 1 ? (uint8_t)(pass == 0 ? 'd' : 'e') : i == 2 ? (uint8_t)'B' : i == 3 ? (uint8_t)'s' : (uint8_t)(L >> (8 * (i - 4)));
        else { const uint32_t k = i - 8; const DCol c = col(k >> 1); const uint32_t w = pass == 0 ? (b_side ? c.db : c.da) : (b_side ? c.eb : c.ea); v = (k & 1) ? (uint8_t)(w >> 8) : (uint8_t)w; }
        q[i] = v;
      }
      q += 8 + 2 * L;
    }
    for (uint32_t i = lane; i < 3 + L + 1; i += 64) {
      uint8_t v = i == 0 ? (uint8_t)s : i == 1 ? (uint8_t)'q' : i == 2 ? (uint8_t)'Z' : (uint8_t)0;
      if (i >= 3 && i - 3 < L) { const DCol c = col(i - 3); const uint32_t qq = (b_side ? c.qb : c.qa) + 33; v = (uint8_t)(qq > 255 ? 255 : qq); }
      q[i] = v;
    }
    q += 3 + L + 1;
  };
  z_tag('M', 'I', first + mi_off, mi_len);
  if (has_cb) z_tag(P.cell0, P.cell1, P.blob + P.rec_off[D.cb_rec] + D.cb_off, cb_len);
  z_tag('R', 'G', (const uint8_t*)P.rg, P.rg_len);
  scalar_tags('a', amax, a_rate, amin);
  if (P.per_base_tags) per_base('a', false);
  scalar_tags('b', bmax, b_rate, bmin);
  if (P.per_base_tags && has_ba) per_base('b', true);
  scalar_tags('c', cmax, c_rate, cmin);
  if (has_rx) z_tag('R', 'X', (const uint8_t*)D.rx, rx_len);
}

// -----------------------------------------------------------------------------------------------------
// k_emit_codec — one wavefront per CODEC molecule: orient and pad the two single-strand consensi
// (codec_caller.rs:955-968, 1272-1314), combine them position by position (:1331-1512), apply the quality
// masks (:1526-1561), turn the result into R1's orientation and write the fragment record (:1590-1757):
// tags RG MI cD cM cE aD aM aE bD bM bE [ad bd ae be ac bc aq bq] [CB] RX.
// -----------------------------------------------------------------------------------------------------
struct CCol { uint32_t b1, q1, d1, e1, b2, q2, d2, e2, ob, oq, oe; bool pad1, pad2, dup, dis; };   // bases as 4-bit codes; padN = lower-case 'n' padding

__global__ __launch_bounds__(256) void k_emit_codec(CodecEmitParams P) {
  const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(P.slot0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
  const uint32_t lane = threadIdx.x & 63;
  if (slot >= P.slot_end) return;
  const CodecDesc& D = P.ends[slot];
  if (!D.valid || codec_fast_ok(D, P.prefix_len, P.rg_len)) return;      // k_emit_codec_fast writes those
  uint8_t* q = P.out + P.out_off[slot];
  const uint32_t C = D.cons_len, l1 = D.l1, l2 = D.l2;
  const uint64_t s1 = D.s1_off, s2 = D.s2_off;
  const bool r1_neg = D.flags & 1, r2_neg = (D.flags & 2) != 0;
  const uint8_t* first = P.blob + P.rec_off[D.first_rec];
  const uint32_t mi_len = D.mi_len, mi_off = D.mi_off, name_len = P.prefix_len + 1 + mi_len;
  const bool has_cb = D.has_cb != 0, has_rx = D.has_rx != 0;
  const uint32_t cb_len = has_cb ? D.cb_len : 0, rx_len = has_rx ? D.rx_len : 0;
  // Position f of the record (R1's orientation).  Both strands are brought to reference orientation (the reverse strand's
  // consensus reverse-complemented), the negative-strand one right-aligned by padding on the left, combined, and the whole
  // thing reverse-complemented again when R1 is the negative strand.
  auto col = [&](uint32_t f) {
    CCol c;
    const uint32_t i = r1_neg ? C - 1 - f : f;
    auto strand = [&](uint64_t off, uint32_t len, bool rc, bool pad_left, uint32_t& b, uint32_t& qq, uint32_t& d, uint32_t& e, bool& pad) {
      const uint32_t shift = pad_left ? C - len : 0;
      pad = i < shift || i - shift >= len;
      b = 15; qq = 0; d = 0; e = 0;
      if (!pad) {
        const uint32_t j = i - shift, k = rc ? len - 1 - j : j;
        b = P.col_code[off + k]; qq = P.col_qual[off + k]; d = P.col_depth[off + k]; e = P.col_err[off + k];
        if (rc) b = comp_code((uint8_t)b);
      }
    };
    strand(s1, l1, r1_neg, r1_neg, c.b1, c.q1, c.d1, c.e1, c.pad1);
    strand(s2, l2, !r1_neg, r2_neg, c.b2, c.q2, c.d2, c.e2, c.pad2);
    const bool ha = !c.pad1 && c.b1 != 15, hb = !c.pad2 && c.b2 != 15;
    c.dup = ha && hb; c.dis = false;
    uint32_t fb, fq, depth, err;
    if (ha && hb) {
      uint32_t rb, rq;
      if (c.b1 == c.b2) { rb = c.b1; const uint32_t sm = c.q1 + c.q2; rq = sm < 93 ? sm : 93; }
      else if (c.q1 > c.q2) { c.dis = true; rb = c.b1; rq = c.q1 - c.q2; if (rq < FGX_MIN_PHRED) rq = FGX_MIN_PHRED; }
      else if (c.q2 > c.q1) { c.dis = true; rb = c.b2; rq = c.q2 - c.q1; if (rq < FGX_MIN_PHRED) rq = FGX_MIN_PHRED; }
      else { c.dis = true; rb = c.b1; rq = FGX_MIN_PHRED; }
      if (rq == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = rb; fq = rq; }
      const uint32_t de = c.b1 == c.b2 ? c.e1 + c.e2 : c.b1 == rb ? c.e1 + (c.d2 > c.e2 ? c.d2 - c.e2 : 0) : c.e2 + (c.d1 > c.e1 ? c.d1 - c.e1 : 0);
      err = de < 32767 ? de : 32767;
      depth = c.d1 + c.d2;
    } else if (ha) { if (c.q1 == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = c.b1; fq = c.q1; } depth = c.d1; err = c.e1; }
    else if (hb) { if (c.q2 == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = c.b2; fq = c.q2; } depth = c.d2; err = c.e2; }
    else { fb = 15; fq = FGX_MIN_PHRED; depth = 0; const uint32_t de = c.e1 + c.e2; err = de < 32767 ? de : 32767; }
    if ((!c.pad1 && c.b1 == 15) || (!c.pad2 && c.b2 == 15)) { fb = 15; fq = FGX_MIN_PHRED; }     // an upper-case N on either strand
    // quality masks, on the reference-orientation index: outer bases first, then single-strand stretches
    if (P.has_outer && P.outer_len > 0 && (i < P.outer_len || C - 1 - i < P.outer_len)) fq = P.outer_qual;
    if (P.has_ss && (!ha || !hb)) fq = P.ss_qual;
    (void)depth;
    c.ob = r1_neg ? (uint32_t)comp_code((uint8_t)fb) : fb; c.oq = fq; c.oe = err;
    if (r1_neg) { c.b1 = comp_code((uint8_t)c.b1); c.b2 = comp_code((uint8_t)c.b2); }
    return c;
  };
  // ---- reductions: cD cM cE / aD aM aE / bD bM bE, and the duplex counters -----------------------------------------------------
  uint32_t amax = 0, amin = 0xFFFFFFFFu, asd = 0, ase = 0, bmax = 0, bmin = 0xFFFFFFFFu, bsd = 0, bse = 0, cmax = 0, cmin = 0xFFFFFFFFu, csd = 0, cse = 0;
  uint32_t n_dup = 0, n_dis = 0;
  for (uint32_t f = lane; f < C; f += 64) {
    const CCol c = col(f);
    amax = c.d1 > amax ? c.d1 : amax; amin = c.d1 < amin ? c.d1 : amin; asd += c.d1; ase += c.e1;
    bmax = c.d2 > bmax ? c.d2 : bmax; bmin = c.d2 < bmin ? c.d2 : bmin; bsd += c.d2; bse += c.e2;
    const uint32_t t = c.d1 + c.d2;
    cmax = t > cmax ? t : cmax; cmin = t < cmin ? t : cmin; csd += t; cse += c.oe;
    n_dup += c.dup ? 1 : 0; n_dis += c.dis ? 1 : 0;
  }
  amax = wave_max(amax); amin = wave_min(amin); asd = wave_sum(asd); ase = wave_sum(ase);
  bmax = wave_max(bmax); bmin = wave_min(bmin); bsd = wave_sum(bsd); bse = wave_sum(bse);
  cmax = wave_max(cmax); cmin = wave_min(cmin); csd = wave_sum(csd); cse = wave_sum(cse);
  n_dup = wave_sum(n_dup); n_dis = wave_sum(n_dis);
  if (C == 0) { amin = 0; bmin = 0; cmin = 0; }
  if (lane == 0) {
    unsigned long long* st = P.stats + (size_t)(blockIdx.x & (STAT_SLOTS - 1)) * 32;
    atomicAdd(&st[24], (unsigned long long)C);
    if (n_dup) atomicAdd(&st[25], (unsigned long long)n_dup);
    if (n_dis) atomicAdd(&st[26], (unsigned long long)n_dis);
  }
  const float a_rate = asd ? (float)ase / (float)asd : 0.0f, b_rate = bsd ? (float)bse / (float)bsd : 0.0f, c_rate = csd ? (float)cse / (float)csd : 0.0f;

  // ---- block_size + core (flag: unmapped fragment), name, bases, quals -------------------------------------------------------------
  if (lane < 36) {
    const uint32_t dw = lane >> 2;
    const uint32_t v = dw == 0 ? D.rec_size : dw == 3 ? ((name_len + 1) | (4680u << 16)) : dw == 4 ? ((uint32_t)bam::F_UNMAPPED << 16) : dw == 5 ? C : dw == 8 ? 0u : 0xFFFFFFFFu;
    q[lane] = (uint8_t)(v >> (8 * (lane & 3)));
  }
  q += 36;
  for (uint32_t i = lane; i < name_len + 1; i += 64)
    q[i] = i < P.prefix_len ? (uint8_t)P.prefix[i] : i == P.prefix_len ? (uint8_t)':' : i < name_len ? first[mi_off + (i - P.prefix_len - 1)] : (uint8_t)0;
  q += name_len + 1;
  for (uint32_t i = lane; i < (C + 1) / 2; i += 64) {
    const uint32_t hi = col(2 * i).ob, lo = 2 * i + 1 < C ? col(2 * i + 1).ob : 0u;
    q[i] = (uint8_t)((hi << 4) | lo);
  }
  q += (C + 1) / 2;
  for (uint32_t i = lane; i < C; i += 64) q[i] = (uint8_t)col(i).oq;
  q += C;
  // ---- tags ----------------------------------------------------------------------------------------------------------------------------
  auto z_tag = [&](char t0, char t1, const uint8_t* src, uint32_t n) {
    for (uint32_t i = lane; i < 3 + n + 1; i += 64) q[i] = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'Z' : i - 3 < n ? src[i - 3] : (uint8_t)0;
    q += 3 + n + 1;
  };
  auto scalar_tags = [&](char s, uint32_t dmax, uint32_t dmin, float rate) {   // <s>D int, <s>M int, <s>E float: 4 + 4 + 7 bytes
    if (lane < 15) {
      uint8_t v;
      if (lane < 4) v = int_tag_byte(lane, s, 'D', dmax);
      else if (lane < 8) v = int_tag_byte(lane - 4, s, 'M', dmin);
      else { const uint32_t j = lane - 8, u = __float_as_uint(rate); v = j == 0 ? (uint8_t)s : j == 1 ? (uint8_t)'E' : j == 2 ? (uint8_t)'f' : (uint8_t)(u >> (8 * (j - 3))); }
      q[lane] = v;
    }
    q += 15;
  };
  z_tag('R', 'G', (const uint8_t*)P.rg, P.rg_len);
  z_tag('M', 'I', first + mi_off, mi_len);
  scalar_tags('c', cmax, cmin, c_rate);
  scalar_tags('a', amax, amin, a_rate);
  scalar_tags('b', bmax, bmin, b_rate);
  if (P.per_base_tags) {
    for (int arr = 0; arr < 4; arr++) {        // ad bd ae be
      const char t0 = (arr & 1) ? 'b' : 'a', t1 = arr < 2 ? 'd' : 'e';
      for (uint32_t i = lane; i < 8 + 2 * C; i += 64) {
        uint8_t v;
        if (i < 8) v = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'B' : i == 3 ? (uint8_t)'s' : (uint8_t)(C >> (8 * (i - 4)));
        else { const uint32_t k = i - 8; const CCol c = col(k >> 1); const uint32_t w = arr == 0 ? c.d1 : arr == 1 ? c.d2 : arr == 2 ? c.e1 : c.e2; v = (k & 1) ? (uint8_t)(w >> 8) : (uint8_t)w; }
        q[i] = v;
      }
      q += 8 + 2 * C;
    }
    for (int str = 0; str < 4; str++) {        // ac bc aq bq
      const char t0 = (str & 1) ? 'b' : 'a', t1 = str < 2 ? 'c' : 'q';
      for (uint32_t i = lane; i < 3 + C + 1; i += 64) {
        uint8_t v = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'Z' : (uint8_t)0;
        if (i >= 3 && i - 3 < C) {
          const CCol c = col(i - 3);
          const bool pd = (str & 1) ? c.pad2 : c.pad1;
          if (str < 2) v = pd ? (uint8_t)'n' : bam::code_to_ascii((uint8_t)((str & 1) ? c.b2 : c.b1));
          else { const uint32_t qq = ((str & 1) ? c.q2 : c.q1) + 33; v = (uint8_t)(qq > 255 ? 255 : qq); }
        }
        q[i] = v;
      }
      q += 3 + C + 1;
    }
  }
  if (has_cb) z_tag(P.cell0, P.cell1, P.blob + P.rec_off[D.cb_rec] + D.cb_off, cb_len);
  if (has_rx) z_tag('R', 'X', (const uint8_t*)D.rx, rx_len);
}

// -----------------------------------------------------------------------------------------------------
// Fast record writers for duplex and CODEC.  The per-field writers above wait on a memory round trip for every
// 64 bytes they produce and re-evaluate the strand combine for every field; here each lane owns PAIRS of
// positions (2·lane, 2·lane+1, then +128 per slot), loads all of them in one sweep, evaluates the combine once
// per position into three packed registers, and then only stores: a string field is two byte stores per slot, an
// int16 array four, packed bases one — no cross-lane traffic, no waits.  Records that do not fit the register
// budget (or have unusually long names / tags) are left to the per-field kernels, which skip what is done here.
// -----------------------------------------------------------------------------------------------------
// shared field writers (q advances; every lane calls them)
// (The small fields are written as straight-line code — uniform words built by the scalar unit, a lane's byte taken with one shift,
// one-level selects —: nested conditionals over the lane number compile into nested exec-mask regions, and the record writers were bound by
// scalar instructions, profiles/r04_experiments.md.)
struct FieldWriterFlat {
  uint8_t* q; uint32_t lane;
  __device__ __forceinline__ uint32_t sh3() const { return 8u * (lane < 3u ? lane : 3u); }   // (a 24-bit header word >> sh3: its byte for lanes 0 - 2, 0 from lane 3 on)
  // one store: bytes [0, n) with n <= 64, byte i = f(i)
  template <class F> __device__ __forceinline__ void small(uint32_t n, F f) { const uint8_t b = f(lane); if (lane < n) q[lane] = b; q += n; }
  __device__ __forceinline__ void z_small(char t0, char t1, const uint8_t* src, uint32_t n) {       // Z tag, n + 4 <= 64
    const uint32_t k = lane >= 3 ? lane - 3 : 0;
    const uint32_t b = src[k < n ? k : (n ? n - 1 : 0)];
    const uint32_t c3 = (uint32_t)(uint8_t)t0 | ((uint32_t)(uint8_t)t1 << 8) | ((uint32_t)'Z' << 16);
    const uint32_t u = (lane - 3u < n) ? b : 0u;                                                      // (unsigned: false for lanes 0 - 2)
    const uint32_t v = (c3 >> sh3()) | u;
    if (lane < 3 + n + 1) q[lane] = (uint8_t)v;
    q += 3 + n + 1;
  }
  __device__ __forceinline__ void scalars(char s, bool m_second, uint32_t dmax, uint32_t dmin, float rate) {   // <s>D <s>E <s>M (duplex) or <s>D <s>M <s>E (CODEC)
    // three uniform words: an integer tag here is `<s>D` + its type + ONE value byte (4 bytes), the rate `<s>Ef` + its four bytes (7)
    auto int_word = [&](uint32_t b, uint32_t v) -> unsigned long long {
      const uint32_t ty = v <= 127 ? (uint32_t)'c' : v <= 255 ? (uint32_t)'C' : (uint32_t)'S';
      return (unsigned long long)((uint32_t)(uint8_t)s | (b << 8) | (ty << 16) | ((v & 0xFFu) << 24));
    };
    const unsigned long long wD = int_word('D', dmax), wM = int_word('M', dmin);
    const unsigned long long wE = (unsigned long long)((uint32_t)(uint8_t)s | ((uint32_t)'E' << 8) | ((uint32_t)'f' << 16)) | ((unsigned long long)__float_as_uint(rate) << 24);
    const unsigned long long w2 = m_second ? wM : wE, w3 = m_second ? wE : wM;
    const uint32_t n2 = m_second ? 4u : 7u;
    unsigned long long w = wD;
    uint32_t k = lane;
    if (lane >= 4u) { w = w2; k = lane - 4u; }
    if (lane >= 4u + n2) { w = w3; k = lane - 4u - n2; }
    const uint32_t v = (uint32_t)(w >> (8u * (k & 7u)));
    if (lane < 15) q[lane] = (uint8_t)v;
    q += 15;
  }
  __device__ __forceinline__ void header3(char t0, char t1) {                                        // `xyZ`
    const uint32_t c3 = (uint32_t)(uint8_t)t0 | ((uint32_t)(uint8_t)t1 << 8) | ((uint32_t)'Z' << 16);
    const uint32_t v = c3 >> sh3();
    if (lane < 3) q[lane] = (uint8_t)v;
    q += 3;
  }
  __device__ __forceinline__ void header8(char t0, char t1, uint32_t L) {                            // `xyBs` + the count
    const unsigned long long w = (unsigned long long)((uint32_t)(uint8_t)t0 | ((uint32_t)(uint8_t)t1 << 8) | ((uint32_t)'B' << 16) | ((uint32_t)'s' << 24)) | ((unsigned long long)L << 32);
    const uint32_t v = (uint32_t)(w >> (8u * (lane & 7u)));
    if (lane < 8) q[lane] = (uint8_t)v;
    q += 8;
  }
  __device__ __forceinline__ void core(uint32_t rec_size, uint32_t name_len, uint32_t flag, uint32_t L) {
    uint32_t v = 0xFFFFFFFFu;
    if (lane == 0) v = rec_size;
    if (lane == 3) v = (name_len + 1) | (4680u << 16);
    if (lane == 4) v = flag << 16;
    if (lane == 5) v = L;
    if (lane == 8) v = 0u;
    if (lane < 9) gst32u(q + 4 * lane, v);
    q += 36;
  }
};

// the same fields as nested conditionals (rounds 2 - 4): k_emit_codec_fast keeps them — the straight-line forms cost it four registers and with
// them a wavefront per SIMD (79 -> 83; 0.54 -> 0.53 G raw reads/s, profiles/r04_experiments.md)
struct FieldWriterNested {
  uint8_t* q; uint32_t lane;
  // one store: bytes [0, n) with n <= 64, byte i = f(i)
  template <class F> __device__ __forceinline__ void small(uint32_t n, F f) { if (lane < n) q[lane] = f(lane); q += n; }
  __device__ __forceinline__ void z_small(char t0, char t1, const uint8_t* src, uint32_t n) {       // Z tag, n + 4 <= 64
    const uint32_t k = lane >= 3 ? lane - 3 : 0;
    const uint8_t b = src[k < n ? k : (n ? n - 1 : 0)];
    small(3 + n + 1, [&](uint32_t i) { return i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'Z' : k < n ? b : (uint8_t)0; });
  }
  __device__ __forceinline__ void scalars(char s, bool m_second, uint32_t dmax, uint32_t dmin, float rate) {   // <s>D <s>E <s>M (duplex) or <s>D <s>M <s>E (CODEC)
    uint8_t v;
    const uint32_t u = __float_as_uint(rate);
    if (m_second) {
      if (lane < 4) v = int_tag_byte(lane, s, 'D', dmax);
      else if (lane < 8) v = int_tag_byte(lane - 4, s, 'M', dmin);
      else { const uint32_t j = lane - 8; v = j == 0 ? (uint8_t)s : j == 1 ? (uint8_t)'E' : j == 2 ? (uint8_t)'f' : (uint8_t)(u >> (8 * ((j - 3) & 3))); }
    } else {
      if (lane < 4) v = int_tag_byte(lane, s, 'D', dmax);
      else if (lane < 11) { const uint32_t j = lane - 4; v = j == 0 ? (uint8_t)s : j == 1 ? (uint8_t)'E' : j == 2 ? (uint8_t)'f' : (uint8_t)(u >> (8 * ((j - 3) & 3))); }
      else v = int_tag_byte(lane - 11, s, 'M', dmin);
    }
    if (lane < 15) q[lane] = v;
    q += 15;
  }
  __device__ __forceinline__ void header3(char t0, char t1) { small(3, [&](uint32_t i) { return i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : (uint8_t)'Z'; }); }
  __device__ __forceinline__ void header8(char t0, char t1, uint32_t L) {
    small(8, [&](uint32_t i) { return i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'B' : i == 3 ? (uint8_t)'s' : (uint8_t)(L >> (8 * ((i - 4) & 3))); });
  }
  __device__ __forceinline__ void core(uint32_t rec_size, uint32_t name_len, uint32_t flag, uint32_t L) {
    if (lane < 36) {
      const uint32_t dw = lane >> 2;
      const uint32_t v = dw == 0 ? rec_size : dw == 3 ? ((name_len + 1) | (4680u << 16)) : dw == 4 ? (flag << 16) : dw == 5 ? L : dw == 8 ? 0u : 0xFFFFFFFFu;
      q[lane] = (uint8_t)(v >> (8 * (lane & 3)));
    }
    q += 36;
  }
};

template <uint32_t SLOTS, class FW, class B>      // string field of L bytes after a 3-byte `xyZ` header and before a NUL
__device__ __forceinline__ void put_string(FW& W, char t0, char t1, uint32_t L, B byte_of) {
  W.header3(t0, t1);
  // a lane's two positions are neighbours in the record: ONE 16-bit store (gfx950 takes them unaligned) instead of two byte stores
#pragma unroll
  for (uint32_t t = 0; t < SLOTS; t++) {
    const uint32_t p = 128 * t + 2 * W.lane;
    if (p + 1 < L) gst16u(W.q + p, (uint32_t)(uint8_t)byte_of(t, 0) | ((uint32_t)(uint8_t)byte_of(t, 1) << 8));
    else if (p < L) W.q[p] = byte_of(t, 0);
  }
  if (W.lane == 0) W.q[L] = 0;
  W.q += L + 1;
}
template <uint32_t SLOTS, class FW, class V>      // B:s array of L int16 values
__device__ __forceinline__ void put_i16(FW& W, char t0, char t1, uint32_t L, V val_of) {
  W.header8(t0, t1, L);
  // a lane's two int16 values are four consecutive bytes: one (unaligned) dword store instead of four byte stores
#pragma unroll
  for (uint32_t t = 0; t < SLOTS; t++) {
    const uint32_t p = 128 * t + 2 * W.lane;
    if (p + 1 < L) gst32u(W.q + 2 * p, ((uint32_t)val_of(t, 0) & 0xFFFFu) | ((uint32_t)val_of(t, 1) << 16));
    else if (p < L) gst16u(W.q + 2 * p, (uint32_t)val_of(t, 0));
  }
  W.q += 2 * L;
}
template <uint32_t SLOTS, class FW, class C, class Q>   // 4-bit packed bases, then qualities
__device__ __forceinline__ void put_seq_qual(FW& W, uint32_t L, C code_of, Q qual_of) {
#pragma unroll
  for (uint32_t t = 0; t < SLOTS; t++) {
    const uint32_t p = 128 * t + 2 * W.lane;
    if (p < L) W.q[p >> 1] = (uint8_t)((code_of(t, 0) << 4) | (p + 1 < L ? code_of(t, 1) : 0u));
  }
  W.q += (L + 1) / 2;
#pragma unroll
  for (uint32_t t = 0; t < SLOTS; t++) {
    const uint32_t p = 128 * t + 2 * W.lane;
    if (p + 1 < L) gst16u(W.q + p, ((uint32_t)qual_of(t, 0) & 0xFFu) | (((uint32_t)qual_of(t, 1) & 0xFFu) << 8));
    else if (p < L) W.q[p] = (uint8_t)qual_of(t, 0);
  }
  W.q += L;
}

// (round 6) how many valid records the fast writers refuse — a thread per slot, one atomic per wavefront: the per-field writers (k_emit_duplex / k_emit_codec)
// are launched only when the count is not 0.  (Counting inside the fast writers cost them a wavefront per SIMD: 71 -> 73 / 79 -> 89 VGPRs.)
__global__ __launch_bounds__(256) void k_count_slow_duplex(const DuplexDesc* __restrict__ ends, uint32_t slot0, uint32_t slot_end, uint32_t prefix_len, uint32_t rg_len, uint32_t* __restrict__ n_slow) {
  const uint32_t slot = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
  bool slow = false;
  if (slot < slot_end) { const DuplexDesc& D = ends[slot]; slow = D.valid && !duplex_fast_ok(D, prefix_len, rg_len); }
  const unsigned long long m = __ballot(slow);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(n_slow, (uint32_t)__popcll(m));
}
__global__ __launch_bounds__(256) void k_count_slow_codec(const CodecDesc* __restrict__ ends, uint32_t slot0, uint32_t slot_end, uint32_t prefix_len, uint32_t rg_len, uint32_t* __restrict__ n_slow) {
  const uint32_t slot = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
  bool slow = false;
  if (slot < slot_end) { const CodecDesc& D = ends[slot]; slow = D.valid && !codec_fast_ok(D, prefix_len, rg_len); }
  const unsigned long long m = __ballot(slow);
  if (m && (threadIdx.x & 63) == 0) atomicAdd(n_slow, (uint32_t)__popcll(m));
}
__global__ __launch_bounds__(256) void k_emit_duplex_fast(DuplexEmitParams P) {
  const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(P.slot0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
  const uint32_t lane = threadIdx.x & 63;
  if (slot >= P.slot_end) return;
  const DuplexDesc& D = P.ends[slot];
  if (!D.valid || !duplex_fast_ok(D, P.prefix_len, P.rg_len)) return;
  const uint32_t L = D.len, lastp = L ? L - 1 : 0;
  const bool has_ba = D.has_ba != 0;
  const uint64_t a_off = D.a_off, b_off = has_ba ? D.b_off : D.a_off;
  const uint8_t* first = P.blob + P.rec_off[D.first_rec];
  const uint32_t mi_len = D.mi_len, mi_off = D.mi_off, name_len = P.prefix_len + 1 + mi_len;
  const bool has_cb = D.has_cb != 0, has_rx = D.has_rx != 0;
  const uint32_t cb_len = has_cb ? D.cb_len : 0, rx_len = has_rx ? D.rx_len : 0;
  const uint32_t flag = bam::F_UNMAPPED | bam::F_PAIRED | bam::F_MATE_UNMAPPED | (D.type == 1 ? bam::F_FIRST : bam::F_LAST);
  // ---- every load of the record (indices clamped into the record's own segments, so unconditional) -----------------------
  uint32_t ca[DUP_SLOTS][2], qa[DUP_SLOTS][2], ea[DUP_SLOTS][2], oa[DUP_SLOTS][2], cb[DUP_SLOTS][2], qb[DUP_SLOTS][2], eb[DUP_SLOTS][2], ob[DUP_SLOTS][2];
#pragma unroll
  for (uint32_t t = 0; t < DUP_SLOTS; t++) {
    // (round 6) a lane's two neighbouring positions with ONE load per array and strand (2 + 2 + 4 + 8 bytes) instead of one per position: 16 vector memory
    // instructions per lane where there were 32.  The pair's first position is clamped into the record; its second may lie one column past the end
    // (the scratch arrays carry slack) — such a position is masked out of the statistics and written by no field writer.
    const uint32_t pb = 128 * t + 2 * lane, pc = pb < lastp ? pb : lastp;
    auto ld2 = [&](uint64_t off, uint32_t* c2, uint32_t* q2, uint32_t* e2, uint32_t* o2) {
      uint16_t cw, qw; uint32_t ew; unsigned long long ow;
      __builtin_memcpy(&cw, P.col_code + off, 2); __builtin_memcpy(&qw, P.col_qual + off, 2);
      __builtin_memcpy(&ew, (const uint8_t*)P.col_err + 2 * off, 4); __builtin_memcpy(&ow, (const uint8_t*)P.col_obs + 4 * off, 8);
      c2[0] = cw & 0xFFu; c2[1] = cw >> 8; q2[0] = qw & 0xFFu; q2[1] = qw >> 8; e2[0] = ew & 0xFFFFu; e2[1] = ew >> 16; o2[0] = (uint32_t)ow; o2[1] = (uint32_t)(ow >> 32);
    };
    ld2(a_off + pc, ca[t], qa[t], ea[t], oa[t]);
    ld2(b_off + pc, cb[t], qb[t], eb[t], ob[t]);
  }
  const uint32_t k3 = lane >= 3 ? lane - 3 : 0, ni = lane > P.prefix_len ? lane - P.prefix_len - 1 : 0;
  const uint8_t pfx = (uint8_t)P.prefix[lane < P.prefix_len ? lane : 0], nmb = first[mi_off + (ni < mi_len ? ni : mi_len)];
  // ---- the duplex call of each position, packed: w0 = ca | cb<<4 | oc<<8 | qa<<16 | qb<<24 ; w1 = da | db<<8 | ea<<16 | eb<<24 ; w2 = oq | oe<<8
  uint32_t w0[DUP_SLOTS][2], w1[DUP_SLOTS][2], w2[DUP_SLOTS][2];
  uint32_t amax = 0, amin = 0xFFFFFFFFu, asd = 0, ase = 0, bmax = 0, bmin = 0xFFFFFFFFu, bsd = 0, bse = 0, cmax = 0, cmin = 0xFFFFFFFFu, csd = 0, cse = 0;
#pragma unroll
  for (uint32_t t = 0; t < DUP_SLOTS; t++)
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
      const uint32_t da = obs_sum(oa[t][k]);
      uint32_t db = 0, xcb = 15, xqb = 0, xeb = 0, oc = ca[t][k], oq = qa[t][k], oe = ea[t][k];
      if (has_ba) {
        db = obs_sum(ob[t][k]); xcb = cb[t][k]; xqb = qb[t][k]; xeb = eb[t][k];
        const uint32_t xa = ca[t][k], xq = qa[t][k];
        uint32_t rb, rq;
        if (xa == xcb) { rb = xa; rq = cap_q((int32_t)xq + (int32_t)xqb); }
        else if (xq > xqb) { rb = xa; rq = cap_q((int32_t)xq - (int32_t)xqb); }
        else if (xqb > xq) { rb = xcb; rq = cap_q((int32_t)xqb - (int32_t)xq); }
        else { rb = xa; rq = FGX_MIN_PHRED; }
        const bool nocall = xa == 15 || xcb == 15 || rq == FGX_MIN_PHRED;
        oc = nocall ? 15u : rb; oq = nocall ? (uint32_t)FGX_MIN_PHRED : rq;
        oe = rb == 15 ? 0u : (da + db) - (obs_of_code(oa[t][k], rb) + obs_of_code(ob[t][k], rb));
      }
      w0[t][k] = ca[t][k] | (xcb << 4) | (oc << 8) | (qa[t][k] << 16) | (xqb << 24);
      w1[t][k] = da | (db << 8) | (ea[t][k] << 16) | (xeb << 24);
      w2[t][k] = oq | (oe << 8);
      if (128 * t + 2 * lane + k < L) {
        amax = da > amax ? da : amax; amin = da < amin ? da : amin; asd += da; ase += ea[t][k];
        bmax = db > bmax ? db : bmax; bmin = db < bmin ? db : bmin; bsd += db; bse += xeb;
        const uint32_t tt = da + db;
        cmax = tt > cmax ? tt : cmax; cmin = tt < cmin ? tt : cmin; csd += tt; cse += oe;
      }
    }
  amax = wave_max(amax); amin = wave_min(amin); asd = wave_sum(asd); ase = wave_sum(ase);
  bmax = wave_max(bmax); bmin = wave_min(bmin); bsd = wave_sum(bsd); bse = wave_sum(bse);
  cmax = wave_max(cmax); cmin = wave_min(cmin); csd = wave_sum(csd); cse = wave_sum(cse);
  if (L == 0) { amin = 0; bmin = 0; cmin = 0; }
  if (!has_ba) { bmax = 0; bmin = 0; bsd = 0; bse = 0; }
  const float a_rate = asd ? (float)ase / (float)asd : 0.0f, b_rate = bsd ? (float)bse / (float)bsd : 0.0f, c_rate = csd ? (float)cse / (float)csd : 0.0f;
  // ---- stores ------------------------------------------------------------------------------------------------------------------
  FieldWriterFlat W{P.out + P.out_off[slot], lane};
  W.core(D.rec_size, name_len, flag, L);
  W.small(name_len + 1, [&](uint32_t i) { const uint8_t x = i == P.prefix_len ? (uint8_t)':' : nmb, y = i < P.prefix_len ? pfx : x; return i < name_len ? y : (uint8_t)0; });   // (three one-level selects)
  put_seq_qual<DUP_SLOTS>(W, L, [&](uint32_t t, uint32_t k) { return (w0[t][k] >> 8) & 15; }, [&](uint32_t t, uint32_t k) { return w2[t][k] & 0xFF; });
  W.z_small('M', 'I', first + mi_off, mi_len);
  if (has_cb) W.z_small(P.cell0, P.cell1, P.blob + P.rec_off[D.cb_rec] + D.cb_off, cb_len);
  W.z_small('R', 'G', (const uint8_t*)P.rg, P.rg_len);
  W.scalars('a', false, amax, amin, a_rate);
  if (P.per_base_tags) {
    put_string<DUP_SLOTS>(W, 'a', 'c', L, [&](uint32_t t, uint32_t k) { return bam::code_to_ascii((uint8_t)(w0[t][k] & 15)); });
    put_i16<DUP_SLOTS>(W, 'a', 'd', L, [&](uint32_t t, uint32_t k) { return w1[t][k] & 0xFF; });
    put_i16<DUP_SLOTS>(W, 'a', 'e', L, [&](uint32_t t, uint32_t k) { return (w1[t][k] >> 16) & 0xFF; });
    put_string<DUP_SLOTS>(W, 'a', 'q', L, [&](uint32_t t, uint32_t k) { return (uint8_t)(((w0[t][k] >> 16) & 0xFF) + 33); });
  }
  W.scalars('b', false, bmax, bmin, b_rate);
  if (P.per_base_tags && has_ba) {
    put_string<DUP_SLOTS>(W, 'b', 'c', L, [&](uint32_t t, uint32_t k) { return bam::code_to_ascii((uint8_t)((w0[t][k] >> 4) & 15)); });
    put_i16<DUP_SLOTS>(W, 'b', 'd', L, [&](uint32_t t, uint32_t k) { return (w1[t][k] >> 8) & 0xFF; });
    put_i16<DUP_SLOTS>(W, 'b', 'e', L, [&](uint32_t t, uint32_t k) { return w1[t][k] >> 24; });
    put_string<DUP_SLOTS>(W, 'b', 'q', L, [&](uint32_t t, uint32_t k) { return (uint8_t)((w0[t][k] >> 24) + 33); });
  }
  W.scalars('c', false, cmax, cmin, c_rate);
  if (has_rx) W.z_small('R', 'X', (const uint8_t*)D.rx, rx_len);
}

__global__ __launch_bounds__(256) void k_emit_codec_fast(CodecEmitParams P) {
  const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(P.slot0 + ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
  const uint32_t lane = threadIdx.x & 63;
  if (slot >= P.slot_end) return;
  const CodecDesc& D = P.ends[slot];
  if (!D.valid || !codec_fast_ok(D, P.prefix_len, P.rg_len)) return;
  const uint32_t C = D.cons_len, l1 = D.l1, l2 = D.l2;
  const uint64_t s1 = D.s1_off, s2 = D.s2_off;
  const bool r1_neg = D.flags & 1, r2_neg = (D.flags & 2) != 0;
  const uint8_t* first = P.blob + P.rec_off[D.first_rec];
  const uint32_t mi_len = D.mi_len, mi_off = D.mi_off, name_len = P.prefix_len + 1 + mi_len;
  const bool has_cb = D.has_cb != 0, has_rx = D.has_rx != 0;
  const uint32_t cb_len = has_cb ? D.cb_len : 0, rx_len = has_rx ? D.rx_len : 0;
  const uint32_t ni = lane > P.prefix_len ? lane - P.prefix_len - 1 : 0;
  const uint8_t pfx = (uint8_t)P.prefix[lane < P.prefix_len ? lane : 0], nmb = first[mi_off + (ni < mi_len ? ni : mi_len)];
  // strand geometry in reference orientation: strand 1 is reverse-complemented and right-aligned when R1 is the negative read,
  // strand 2 is reverse-complemented when R1 is NOT negative and right-aligned when R2 is
  const uint32_t sh1 = r1_neg ? C - l1 : 0, sh2 = r2_neg ? C - l2 : 0;
  const bool rc1 = r1_neg, rc2 = !r1_neg;
  // field layout of the record (every offset is known up front, so the position-indexed fields can be written window by window)
  uint8_t* const rec = P.out + P.out_off[slot];
  uint8_t* const q_seq = rec + 36 + name_len + 1;
  uint8_t* const q_qual = q_seq + (C + 1) / 2;
  uint8_t* const q_rg = q_qual + C;
  uint8_t* const q_scal = q_rg + (3 + P.rg_len + 1) + (3 + mi_len + 1);
  uint8_t* const q_arr = q_scal + 45;                       // ad bd ae be: 8 + 2C each
  uint8_t* const q_str = q_arr + 4 * (8 + 2 * C);           // ac bc aq bq: 3 + C + 1 each
  uint8_t* const q_tail = P.per_base_tags ? q_str + 4 * (3 + C + 1) : q_arr;
  uint32_t amax = 0, amin = 0xFFFFFFFFu, asd = 0, ase = 0, bmax = 0, bmin = 0xFFFFFFFFu, bsd = 0, bse = 0, cmax = 0, cmin = 0xFFFFFFFFu, csd = 0, cse = 0;
  uint32_t n_dup = 0, n_dis = 0;
  for (uint32_t win = 0; win < C; win += 256) {             // 256 positions per sweep: 4 per lane, all loads of the sweep in flight together
    uint32_t vb1[4], vq1[4], vd1[4], ve1[4], vb2[4], vq2[4], vd2[4], ve2[4];
    bool vp1[4], vp2[4];
    uint32_t vi[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t f = win + 128 * (u >> 1) + 2 * lane + (u & 1);
      const uint32_t i = f < C ? (r1_neg ? C - 1 - f : f) : 0;
      vi[u] = i;
      vp1[u] = i < sh1 || i - sh1 >= l1; vp2[u] = i < sh2 || i - sh2 >= l2;
      const uint32_t j1 = vp1[u] ? 0 : i - sh1, j2 = vp2[u] ? 0 : i - sh2;
      const uint32_t k1 = l1 ? (rc1 ? l1 - 1 - j1 : j1) : 0, k2 = l2 ? (rc2 ? l2 - 1 - j2 : j2) : 0;
      vb1[u] = P.col_code[s1 + k1]; vq1[u] = P.col_qual[s1 + k1]; vd1[u] = P.col_depth[s1 + k1]; ve1[u] = P.col_err[s1 + k1];
      vb2[u] = P.col_code[s2 + k2]; vq2[u] = P.col_qual[s2 + k2]; vd2[u] = P.col_depth[s2 + k2]; ve2[u] = P.col_err[s2 + k2];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; u++) {
      const uint32_t f = win + 128 * (u >> 1) + 2 * lane + (u & 1), i = vi[u];
      const bool inrec = f < C, pad1 = vp1[u], pad2 = vp2[u];
      uint32_t b1 = vb1[u], q1 = vq1[u], d1 = vd1[u], e1 = ve1[u], b2 = vb2[u], q2 = vq2[u], d2 = vd2[u], e2 = ve2[u];
      if (rc1) b1 = comp_code((uint8_t)b1);
      if (rc2) b2 = comp_code((uint8_t)b2);
      if (pad1) { b1 = 15; q1 = 0; d1 = 0; e1 = 0; }
      if (pad2) { b2 = 15; q2 = 0; d2 = 0; e2 = 0; }
      const bool ha = !pad1 && b1 != 15, hb = !pad2 && b2 != 15;
      bool dis = false;
      uint32_t fb, fq, err;
      if (ha && hb) {
        uint32_t rb, rq;
        if (b1 == b2) { rb = b1; const uint32_t sm = q1 + q2; rq = sm < 93 ? sm : 93; }
        else if (q1 > q2) { dis = true; rb = b1; rq = q1 - q2; if (rq < FGX_MIN_PHRED) rq = FGX_MIN_PHRED; }
        else if (q2 > q1) { dis = true; rb = b2; rq = q2 - q1; if (rq < FGX_MIN_PHRED) rq = FGX_MIN_PHRED; }
        else { dis = true; rb = b1; rq = FGX_MIN_PHRED; }
        if (rq == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = rb; fq = rq; }
        const uint32_t de = b1 == b2 ? e1 + e2 : b1 == rb ? e1 + (d2 > e2 ? d2 - e2 : 0) : e2 + (d1 > e1 ? d1 - e1 : 0);
        err = de < 32767 ? de : 32767;
      } else if (ha) { if (q1 == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = b1; fq = q1; } err = e1; }
      else if (hb) { if (q2 == FGX_MIN_PHRED) { fb = 15; fq = FGX_MIN_PHRED; } else { fb = b2; fq = q2; } err = e2; }
      else { fb = 15; fq = FGX_MIN_PHRED; const uint32_t de = e1 + e2; err = de < 32767 ? de : 32767; }
      if ((!pad1 && b1 == 15) || (!pad2 && b2 == 15)) { fb = 15; fq = FGX_MIN_PHRED; }
      if (P.has_outer && P.outer_len > 0 && (i < P.outer_len || C - 1 - i < P.outer_len)) fq = P.outer_qual;
      if (P.has_ss && (!ha || !hb)) fq = P.ss_qual;
      if (r1_neg) { fb = comp_code((uint8_t)fb); b1 = comp_code((uint8_t)b1); b2 = comp_code((uint8_t)b2); }
      if (inrec) {
        amax = d1 > amax ? d1 : amax; amin = d1 < amin ? d1 : amin; asd += d1; ase += e1;
        bmax = d2 > bmax ? d2 : bmax; bmin = d2 < bmin ? d2 : bmin; bsd += d2; bse += e2;
        const uint32_t tt = d1 + d2;
        cmax = tt > cmax ? tt : cmax; cmin = tt < cmin ? tt : cmin; csd += tt; cse += err;
        n_dup += (ha && hb) ? 1 : 0; n_dis += dis ? 1 : 0;
        q_qual[f] = (uint8_t)fq;
        if (P.per_base_tags) {
          uint8_t* a = q_arr + 8 + 2 * f;
          a[0] = (uint8_t)d1; a[1] = 0;
          a += 8 + 2 * C; a[0] = (uint8_t)d2; a[1] = 0;
          a += 8 + 2 * C; a[0] = (uint8_t)e1; a[1] = 0;
          a += 8 + 2 * C; a[0] = (uint8_t)e2; a[1] = 0;
          uint8_t* z = q_str + 3 + f;
          z[0] = pad1 ? (uint8_t)'n' : bam::code_to_ascii((uint8_t)b1);
          z += 3 + C + 1; z[0] = pad2 ? (uint8_t)'n' : bam::code_to_ascii((uint8_t)b2);
          z += 3 + C + 1; z[0] = (uint8_t)(q1 + 33);
          z += 3 + C + 1; z[0] = (uint8_t)(q2 + 33);
        }
      }
      vb1[u] = fb;       // keep the duplex base for the nibble packing below
    }
#pragma unroll
    for (uint32_t h = 0; h < 2; h++) {   // packed bases: this lane's two positions of each half-window share one byte
      const uint32_t f = win + 128 * h + 2 * lane;
      if (f < C) q_seq[f >> 1] = (uint8_t)((vb1[2 * h] << 4) | (f + 1 < C ? vb1[2 * h + 1] : 0u));
    }
  }
  amax = wave_max(amax); amin = wave_min(amin); asd = wave_sum(asd); ase = wave_sum(ase);
  bmax = wave_max(bmax); bmin = wave_min(bmin); bsd = wave_sum(bsd); bse = wave_sum(bse);
  cmax = wave_max(cmax); cmin = wave_min(cmin); csd = wave_sum(csd); cse = wave_sum(cse);
  n_dup = wave_sum(n_dup); n_dis = wave_sum(n_dis);
  if (C == 0) { amin = 0; bmin = 0; cmin = 0; }
  if (lane == 0) {
    unsigned long long* st = P.stats + (size_t)(blockIdx.x & (STAT_SLOTS - 1)) * 32;
    atomicAdd(&st[24], (unsigned long long)C);
    if (n_dup) atomicAdd(&st[25], (unsigned long long)n_dup);
    if (n_dis) atomicAdd(&st[26], (unsigned long long)n_dis);
  }
  const float a_rate = asd ? (float)ase / (float)asd : 0.0f, b_rate = bsd ? (float)bse / (float)bsd : 0.0f, c_rate = csd ? (float)cse / (float)csd : 0.0f;
  // ---- the fields that are not indexed by position ----------------------------------------------------------------------------
  FieldWriterNested W{rec, lane};
  W.core(D.rec_size, name_len, (uint32_t)bam::F_UNMAPPED, C);
  W.small(name_len + 1, [&](uint32_t i) { return i < P.prefix_len ? pfx : i == P.prefix_len ? (uint8_t)':' : i < name_len ? nmb : (uint8_t)0; });
  W.q = q_rg;
  W.z_small('R', 'G', (const uint8_t*)P.rg, P.rg_len);
  W.z_small('M', 'I', first + mi_off, mi_len);
  W.scalars('c', true, cmax, cmin, c_rate);
  W.scalars('a', true, amax, amin, a_rate);
  W.scalars('b', true, bmax, bmin, b_rate);
  if (P.per_base_tags) {
    if (lane < 32) {          // the four array headers and the four string headers + terminators
      const uint32_t r = lane >> 3, i = lane & 7;
      const char t0 = (r & 1) ? 'b' : 'a', t1 = r < 2 ? 'd' : 'e';
      q_arr[(size_t)r * (8 + 2 * C) + i] = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : i == 2 ? (uint8_t)'B' : i == 3 ? (uint8_t)'s' : (uint8_t)(C >> (8 * ((i - 4) & 3)));
    } else if (lane < 48) {
      const uint32_t r = (lane - 32) >> 2, i = (lane - 32) & 3;
      const char t0 = (r & 1) ? 'b' : 'a', t1 = r < 2 ? 'c' : 'q';
      uint8_t* z = q_str + (size_t)r * (3 + C + 1);
      if (i < 3) z[i] = i == 0 ? (uint8_t)t0 : i == 1 ? (uint8_t)t1 : (uint8_t)'Z'; else z[3 + C] = 0;
    }
  }
  W.q = q_tail;
  if (has_cb) W.z_small(P.cell0, P.cell1, P.blob + P.rec_off[D.cb_rec] + D.cb_off, cb_len);
  if (has_rx) W.z_small('R', 'X', (const uint8_t*)D.rx, rx_len);
}

#include "simplex_wave2.inc"
#include "simplex_seg.inc"
#include "simplex_split.inc"
#include "simplex_deep.inc"

// Upper bound on the consensus columns a batch can produce: a family yields at most three ends, each no
// longer than its longest read, and l_seq <= (block_size - 33) * 2 / 3.
// Per family: the scratch-column bound, and a descriptor {byte offset of its first record, bytes up to the end of its last record,
// record count} — with it k_simplex_wave2 starts loading the family's bytes one memory round trip earlier (it needs neither
// grp_first nor rec_off to know where they are; it checks afterwards that every record lies inside that span).
// (+ how many records sit in families of at most 32 records: what decides between the two heads of the simplex launch chain)
__global__ void k_col_bound(const uint32_t* __restrict__ grp_first, const uint32_t* __restrict__ rec_len, const uint64_t* __restrict__ rec_off, uint32_t n_grp,
                            uint64_t* __restrict__ bound, uint32_t max_ends, uint4* __restrict__ fam_desc, unsigned long long* __restrict__ small_recs, uint64_t* __restrict__ sizes_tail) {
  uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  {
    uint32_t mine = 0;
    if (g < n_grp) { const uint32_t n = grp_first[g + 1] - grp_first[g]; mine = (grp_first[g + 1] >= grp_first[g] && n <= 64u) ? n : 0u; }
    // one global atomic per WORKGROUP (a wavefront's sum goes through LDS first): 78 k same-address atomics per 5 M families were most of this kernel's 1 ms
    __shared__ unsigned int s_tot;
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    const uint32_t tot = wave_sum(mine);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&s_tot, tot);
    __syncthreads();
    if (threadIdx.x == 0 && s_tot) atomicAdd(small_recs, (unsigned long long)s_tot);
  }
  if (g >= n_grp) return;
  if (g == n_grp - 1) { bound[n_grp] = 0; *sizes_tail = 0; }  // (both scans of the step run over one element more, a zero: their last element is the total)
  uint32_t a = grp_first[g], b = grp_first[g + 1], mx = 0;
  {   // the family's longest record: four lengths per load where the table allows it (a thread-per-family loop of single loads cost 1 ms per 80 M records)
    uint32_t r = a;
    if (((uintptr_t)rec_len & 15u) == 0) {
      for (; r < b && (r & 3u); r++) { const uint32_t l = rec_len[r]; mx = l > mx ? l : mx; }
      for (; r + 4 <= b; r += 4) {
        const uint4 v = *(const uint4*)(rec_len + r);
        const uint32_t m01 = v.x > v.y ? v.x : v.y, m23 = v.z > v.w ? v.z : v.w, m4 = m01 > m23 ? m01 : m23;
        mx = m4 > mx ? m4 : mx;
      }
    }
    for (; r < b; r++) { const uint32_t l = rec_len[r]; mx = l > mx ? l : mx; }
  }
  uint32_t lb = mx > 33 ? (mx - 33) * 2 / 3 + 1 : 1;
  uint32_t ends = (b - a) < max_ends ? (b - a) : max_ends;
  bound[g] = (uint64_t)ends * lb + COL_BOUND_PAD;
  uint64_t lo = 0;
  uint32_t span = 0xFFFFFFFFu;                               // no usable span (empty group, descending or far-apart records)
  if (b > a) {
    lo = rec_off[a];
    const uint64_t hi = rec_off[b - 1] + rec_len[b - 1];
    if (hi >= lo && hi - lo < 0xFFFFFFFFull) span = (uint32_t)(hi - lo);
  }
  fam_desc[g] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), span, b - a);
}

__global__ void k_reduce_stats(const unsigned long long* __restrict__ slots, unsigned long long* __restrict__ out) {
  uint32_t k = threadIdx.x;   // one thread per counter; the slots' entries 28 .. 31 are diagnostics (k_split_finish: families per build) and go to out[40 ..]
  if (k >= 32) return;
  unsigned long long v = 0;
  for (int i = 0; i < STAT_SLOTS; i++) v += slots[(size_t)i * 32 + k];
  out[k < FGX_STATS_LEN ? k : 40 + (k - FGX_STATS_LEN)] = v;
}

}  // namespace

// -----------------------------------------------------------------------------------------------------
// host driver
// -----------------------------------------------------------------------------------------------------
// diagnostics of a batch (fgx_debug_last_chain: bench.py's share_of_8 block): kernel launches and host synchronisations of FastPath::run_once
#define FGX_SYNC(st) do { last_host_syncs++; hip_check(hipStreamSynchronize(st), "sync"); } while (0)
void FastPath::release() {
  for (DevBuf* b : {&d_ends, &d_sizes, &d_offsets, &d_code, &d_qual, &d_depth, &d_err, &d_misc, &d_deferred, &d_out, &d_scan_tmp, &d_strings, &d_obs, &d_retry2,
                    &d_retry, &d_bound, &d_colbase, &d_statslots, &d_full_items, &d_full_count, &d_retry_old, &d_w2img, &d_famdesc, &d_fwimg,
                    &d_split_rec, &d_split_fam, &d_split_out, &d_route, &d_s2img, &d_dir_size, &d_dir_off, &d_dir_base, &d_slot_desc, &d_slot_err, &d_out2, &d_scan_tmp2, &d_big, &d_deep_sizes, &d_deep_row0, &d_deep_rows, &d_deep_fams, &d_deep_out, &d_deep_out2, &d_mflag, &d_mu, &d_mt, &d_mslot, &d_mcontigs})
    b->free_();
  for (int i = 0; i < 4; i++) if (ev[i]) { (void)hipEventDestroy(ev[i]); ev[i] = nullptr; }
  if (s2) {
    (void)hipStreamDestroy(s2); s2 = nullptr;
    for (int i = 0; i < MAX_CHUNKS; i++) { if (ev_chunk[i]) { (void)hipEventDestroy(ev_chunk[i]); ev_chunk[i] = nullptr; } if (ev_cols[i]) { (void)hipEventDestroy(ev_cols[i]); ev_cols[i] = nullptr; } }
    if (ev_fin) { (void)hipEventDestroy(ev_fin); ev_fin = nullptr; }
    if (ev_sample) { (void)hipEventDestroy(ev_sample); ev_sample = nullptr; }
  }
}

// The columns whose call needs the log-sum-exp chain wait in append lists for k_call_full.  Their room is a fraction of the column
// bound (1/8: clean libraries need under 1 %); when a batch fills EVERY list — a noisy library: a few per cent of disagreeing bases
// at depth 8 — the batch is run again with twice the room (and the caller keeps the larger fraction for its next batches) instead
// of handing the families that found no room to the host path.
int FastPath::run(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len,
                  uint32_t n_rec, const uint32_t* d_grp_first, uint32_t n_grp, FastResult* res) {
  if (!pool_init) {   // (test knobs: a tiny pool makes a small batch exhaust it)
    if (const char* e = getenv("FGX_POOL_DIV")) { const long v = atol(e); if (v >= 1 && v <= (1 << 20)) pool_div = (uint32_t)v; }
    if (const char* e = getenv("FGX_POOL_SLACK")) { const long v = atol(e); if (v >= 1 && v <= 4096) pool_slack = (uint32_t)v; }
    pool_init = true;
  }
  for (;;) {
    const int rc = run_once(c, d_blob, blob_len, d_rec_off, d_rec_len, n_rec, d_grp_first, n_grp, res);
    if (rc != RUN_AGAIN_LARGER_POOL) {
      if (const char* e = fgx_knob("FGX_S2_DEBUG")) if (atoi(e)) {   // (development: the packed pass's counters, cumulative over the process)
        uint32_t h[64];
        if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_s2_dbg), sizeof(h)) == hipSuccess) {
          fprintf(stderr, "[s2 packed] runs tried %u finished %u (columns for k_call_full %u) no room %u shape refused %u", h[0], h[1], h[2], h[3], h[42]);
          fprintf(stderr, "\n");
        }
      }
      return rc;
    }
  }
}

int FastPath::run_once(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len,
                       uint32_t n_rec, const uint32_t* d_grp_first, uint32_t n_grp, FastResult* res) {
  const fgx_options& o = c->opt;
  const bool duplex = o.caller_kind == FGX_CALLER_DUPLEX, codec = o.caller_kind == FGX_CALLER_CODEC;
  hipStream_t s = c->stream;
  memset(res, 0, sizeof(*res));
  last_launches = 0; last_host_syncs = 0;
  if (n_grp == 0) return 0;
  const uint32_t n_slots = 3 * n_grp;   // simplex: Fragment, R1, R2 of each family; duplex: slot 0 unused, R1, R2
  d_ends.reserve((size_t)n_slots * (duplex ? sizeof(DuplexDesc) : codec ? sizeof(CodecDesc) : sizeof(EndDesc)));
  d_sizes.reserve(((size_t)n_slots + 1) * 8);            // (+ 1: a zero behind the last size, so that the scan's last element IS the total — one small copy fewer per step)
  d_offsets.reserve(((size_t)n_slots + 1) * 8);
  d_deferred.reserve((size_t)n_grp * 4);
  // misc: [0..28) stats, [28] col_cursor, [29] n_deferred (u32 in low half), [30] valid count
  d_misc.reserve(48 * 8);   // ... [31] n_retry, [32] n_retry_old (k_simplex_wave2 → k_family_wave<0>); [40 .. 44) diagnostics (k_reduce_stats)
  hip_check(hipMemsetAsync(d_misc.p, 0, 48 * 8, s), "memset");
  // strings: prefix | rg
  std::string strs = c->prefix + c->rg;
  d_strings.reserve(strs.size() + 16);
  if (!strs.empty()) hip_check(hipMemcpyAsync(d_strings.p, strs.data(), strs.size(), hipMemcpyHostToDevice, s), "H2D strings");
  // column scratch: deterministic per-family slots from an exclusive scan of the column bound
  (void)n_rec;
  for (int i = 0; i < 4; i++) if (!ev[i]) hip_check(hipEventCreate(&ev[i]), "hipEventCreate");
  unsigned long long* misc = d_misc.as<unsigned long long>();
  d_bound.reserve(((size_t)n_grp + 1) * 8); d_colbase.reserve(((size_t)n_grp + 1) * 8);
  d_statslots.reserve((size_t)STAT_SLOTS * 32 * 8);
  hip_check(hipMemsetAsync(d_statslots.p, 0, (size_t)STAT_SLOTS * 32 * 8, s), "memset");
  d_famdesc.reserve((size_t)n_grp * 16);
  hip_check(hipEventRecord(c->ev0, s), "event");
  hip_check(hipEventRecord(ev[0], s), "event");   // the family stage: record kernel / column bound, scan, family kernels, k_call_full
  // Simplex without --trim: which head of the launch chain?  Shallow families (the mean family fits a quarter of a wave's LDS
  // slice) start at k_simplex_seg<4>; everything else at the split pipeline (k_split_parse + k_split_cols, simplex_split.inc),
  // whose record kernel also leaves what k_col_bound would (column bound, byte-span descriptor).
  static const bool use_v2 = [] { const char* e = fgx_knob("FGX_V2"); return !(e && e[0] == '0'); }();
  static const bool use_seg = [] { const char* e = fgx_knob("FGX_SEG"); return !(e && e[0] == '0'); }();
  const bool use_split_env = [] { const char* e = getenv("FGX_SPLIT"); return !(e && e[0] == '0'); }();   // (read per batch: tests switch it inside one process)
  uint32_t seg_bytes = 11776;   // 4 wavefronts x 11776 B + the static tables = 3 workgroups per CU
  if (const char* e = fgx_knob("FGX_SEG_BYTES")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 4096 && v <= 32768) seg_bytes = v & ~63u; }
  const double mean_span = (double)blob_len / (double)n_grp + 48.0;   // mean bytes of a family + alignment / read-ahead slack
  // Methylation-aware mode (simplex, a reference set, no --trim): the streaming kernels of simplex_deep.inc are the whole pipeline — every
  // family on their list, the reference lookup / counts / normalisation in k_deep_cols<1>, MM / ML / cu / ct behind the standard record
  // (k_meth_sizes, k_meth_tail); a family outside their shape is deferred to the general path, which knows the mode.
  const bool meth_dev = !duplex && !codec && !o.trim && o.methylation_mode != FGX_METHYLATION_DISABLED && c->genome != nullptr;
  last_meth_device = 0;
  const bool simplex_v2 = !duplex && !codec && !o.trim && use_v2 && !meth_dev;
  const bool seg4 = simplex_v2 && use_seg && mean_span + (16 * 8 + 64) <= seg_bytes / 4;
  // the split pipeline's geometry: families per wavefront of the record kernel (as many as fill its 64 lanes on average), chunks, families per chunk
  const double mean_recs = (double)n_rec / (double)n_grp;
  auto split_geometry = [&](uint32_t& fpw, uint32_t& n_chunks, uint32_t& chunk_fam_raw, uint32_t& chunk_fam) {
    fpw = mean_recs >= 1.0 ? (uint32_t)(64.0 / mean_recs) : 16u;
    fpw = fpw < 1u ? 1u : fpw > 16u ? 16u : fpw;
    if (const char* e = fgx_knob("FGX_SPLIT_FPW")) { const int v = atoi(e); if (v >= 1 && v <= 32) fpw = (uint32_t)v; }   // (measurement knob)
    static const int chunks_env = [] { const char* e = getenv("FGX_SPLIT_CHUNKS"); return e ? atoi(e) : 0; }();      // (measurement knob)
    // chunks of at least ~150 000 families (a column kernel of 37 500 workgroups: 20 rounds of the chip's resident workgroups, so that its tail — the
    // last round runs with a part of the chip — stays a few per cent), eight at most: one rank's share of an 8-way strong-scaling run (625 000
    // families) took 8 chunks of 78 000 until round 6 and spent a third of its step in launch gaps and kernel tails (bench.py share_of_8)
    n_chunks = chunks_env >= 1 ? (uint32_t)chunks_env : std::min<uint32_t>(8u, std::max<uint32_t>(1u, n_grp / 150000u));
    if (n_chunks > (uint32_t)MAX_CHUNKS - 1) n_chunks = MAX_CHUNKS - 1;   // (the last event marks where the second stream starts)
    chunk_fam_raw = (n_grp + n_chunks - 1) / n_chunks;
    chunk_fam = ((chunk_fam_raw + 4 * fpw - 1) / (4 * fpw)) * (4 * fpw);      // whole workgroups of both kernels per chunk
  };
  auto split_streams = [&]() {
    if (!s2) { create_compute_stream(&s2); for (int i = 0; i < MAX_CHUNKS; i++) { hip_check(hipEventCreateWithFlags(&ev_chunk[i], hipEventDisableTiming), "hipEventCreate"); hip_check(hipEventCreateWithFlags(&ev_cols[i], hipEventDisableTiming), "hipEventCreate"); } hip_check(hipEventCreateWithFlags(&ev_fin, hipEventDisableTiming), "hipEventCreate"); hip_check(hipEventCreateWithFlags(&ev_sample, hipEventDisableTiming), "hipEventCreate"); }
  };
  auto split_parse_params = [&](FastParams& PK) {
    memset(&PK, 0, sizeof(PK));
    PK.blob = d_blob; PK.rec_off = d_rec_off; PK.rec_len = d_rec_len; PK.grp_first = d_grp_first; PK.blob_len = blob_len;
    PK.min_reads = o.min_reads; PK.max_reads = o.max_reads; PK.overlap = o.overlapping_consensus;
    PK.tag0 = o.tag[0]; PK.tag1 = o.tag[1]; PK.cell0 = o.cell_tag[0]; PK.cell1 = o.cell_tag[1];
    PK.prefix_len = (uint32_t)c->prefix.size();
    PK.split_rec = d_split_rec.as<SplitRec>(); PK.split_fam = d_split_fam.as<SplitFam>();
  };
  // (round 6) The record kernel of the FIRST chunk starts before the column bounds are counted and scanned: it needs nothing of them, and the host
  // waits for its first families anyway (their tile strides pick the column kernel's build).  Before, the second stream started behind
  // k_col_bound + scan + a host synchronisation: 0.37 ms into the step.  Launched when the batch can take the split pipeline at all; should the count of
  // small families then say otherwise (below), its descriptors are simply not used.
  bool early_parse = false;
  {
    const bool direct_env0 = [] { const char* e = getenv("FGX_DIRECT"); return e && e[0] == '1'; }();
    static const bool early_env = [] { const char* e = fgx_knob("FGX_S2_EARLY"); return !(e && e[0] == '0'); }();      // (measurement knob)
    if (simplex_v2 && use_split_env && !seg4 && !direct_env0 && early_env) {
      d_split_rec.reserve((size_t)n_rec * sizeof(SplitRec) + 64);
      d_split_fam.reserve((size_t)n_grp * sizeof(SplitFam) + 64);
      split_streams();
      uint32_t fpw, n_chunks, chunk_fam_raw, chunk_fam;
      split_geometry(fpw, n_chunks, chunk_fam_raw, chunk_fam);
      FastParams PK;
      split_parse_params(PK);
      hip_check(hipEventRecord(ev_chunk[MAX_CHUNKS - 1], s), "event");      // (behind what the stream holds: the batch's buffers, the memsets)
      hip_check(hipStreamWaitEvent(s2, ev_chunk[MAX_CHUNKS - 1], 0), "wait");
      const uint32_t gb = (uint32_t)std::min<uint64_t>(chunk_fam, n_grp);
      const uint64_t waves = ((uint64_t)gb + fpw - 1) / fpw;
      do { last_launches++; hipLaunchKernelGGL(k_split_parse, dim3((uint32_t)((waves + 1) / 2)), dim3(128), 0, s2, PK, 0u, gb, fpw, (uint64_t*)nullptr, 3u, (uint4*)nullptr); } while (0);
      hip_check(hipGetLastError(), "k_split_parse launch (first chunk, early)");
      hip_check(hipEventRecord(ev_chunk[0], s2), "event");
      early_parse = true;
    }
  }
  do { last_launches++; hipLaunchKernelGGL(k_col_bound, dim3((n_grp + 1023) / 1024), dim3(1024), 0, s, d_grp_first, d_rec_len, d_rec_off, n_grp, d_bound.as<uint64_t>(), duplex ? 4u : codec ? 2u : 3u,
                     d_famdesc.as<uint4>(), misc + 35, d_sizes.as<uint64_t>() + n_slots); } while (0);
  {
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_bound.as<uint64_t>(), d_colbase.as<uint64_t>(), (int)n_grp + 1, s);
    d_scan_tmp.reserve(tb);
    hip_check(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp.p, tb, d_bound.as<uint64_t>(), d_colbase.as<uint64_t>(), (int)n_grp + 1, s), "scan bound");
  }
  // (round 6: every small device-to-host copy costs the stream ~25 us, timeline of the 625 000-family step — so the total is the scan's own last element)
  uint64_t col_total = 0;
  hip_check(hipMemcpyAsync(&col_total, d_colbase.as<uint64_t>() + n_grp, 8, hipMemcpyDeviceToHost, s), "D2H");
  unsigned long long small_recs = 0;
  hip_check(hipMemcpyAsync(&small_recs, misc + 35, 8, hipMemcpyDeviceToHost, s), "D2H");
  FGX_SYNC(s);
  // The split pipeline is the faster head for every family it keeps (depth 8: 25.7 against 37 ms per 5 M families) — since round 4 that is
  // every family of up to 64 records in the common shape (an end of more than 16 reads sends its disagreeing columns to k_call_full as
  // several items); larger families it only measures and hands to k_family.  Split when at least half of the reads sit in families of
  // at most 64 records (FGX_SPLIT=0 / 2: never / always).
  const int split_mode = [] { const char* e = getenv("FGX_SPLIT"); return e ? atoi(e) : 1; }();
  const bool use_split = simplex_v2 && use_split_env && !seg4 && (split_mode == 2 || (double)small_recs >= 0.5 * (double)n_rec);
  // Direct records (fastpath.h): the split pipeline's column kernel writes the consensus records itself (FGX_DIRECT=1).  Byte-identical on
  // the GPU (tools/direct_check.py, tests/test_gpu_direct_records.py), but NOT the default: measured on 5 M depth-8 families the column kernel
  // pays for the emission what k_emit cost as a kernel of its own (k_split_cols 2.82 -> 3.82 ms per chunk, k_split_parse 0.94 -> 1.41 ms with the
  // size prediction; step 33.7 -> 40.2 ms with the merge, ~34 ms without: profiles/r04_experiments.md) — vector-instruction issue is what the
  // stage is short of, and the record's constant bytes cost as many instructions written from here as from there.
  const bool direct_env = [] { const char* e = getenv("FGX_DIRECT"); return e && e[0] == '1'; }();   // (read per batch: tools/direct_check.py switches it between two runs of one process)
  const bool direct = use_split && direct_env && !direct_off && !meth_dev;
  last_direct = 0;
  if (early_parse && !use_split) hip_check(hipStreamWaitEvent(s, ev_chunk[0], 0), "wait");   // (its descriptors are not used; nothing of this batch may still run when the call returns)
  last_routed = 0; last_big_families = 0; last_deep_families = 0; last_packed_families = 0; last_classic_families = 0; last_split_build = 0; last_first_stage_retries = 0;
  uint64_t col_cap = col_total + 64;
  uint64_t dir_cap = 0;
  if (direct) {
    // room for the records: 6 bytes per column of the (generous) column bound covers sequence + qualities + cd + ce twice over for
    // simulate-shaped reads; names and tags come on top.  A batch that needs more says so (dir_flags[1]) and runs again with what it asked for.
    dir_cap = std::max<uint64_t>(col_cap * 6 + (uint64_t)n_slots * 64 + 4096, dir_cap_min);
    d_out.reserve(dir_cap + 64);
    d_dir_size.reserve((size_t)n_grp * 4 + 64); d_dir_off.reserve((size_t)n_grp * 8 + 64); d_dir_base.reserve((MAX_CHUNKS + 2) * 8);
    d_slot_desc.reserve((size_t)n_slots * sizeof(SlotDesc)); d_slot_err.reserve((size_t)n_slots * 4);
    hip_check(hipMemsetAsync(d_dir_size.p, 0, (size_t)n_grp * 4, s), "memset");
    hip_check(hipMemsetAsync(d_dir_base.p, 0, (MAX_CHUNKS + 2) * 8, s), "memset");
    hip_check(hipMemsetAsync(d_slot_err.p, 0, (size_t)n_slots * 4, s), "memset");
    hip_check(hipMemsetAsync(d_sizes.p, 0, ((size_t)n_slots + 1) * 8, s), "memset");
  }
  d_code.reserve(col_cap + 64); d_qual.reserve(col_cap + 64); d_err.reserve(col_cap * 2 + 64);   // (+ slack: k_emit reads whole dwords)
  if (meth_dev) { d_mflag.reserve(col_cap + 64); d_mu.reserve(col_cap * 2 + 64); d_mt.reserve(col_cap * 2 + 64); d_mslot.reserve((size_t)n_slots * sizeof(MethSlot) + 64); }
  if (duplex) d_obs.reserve(col_cap * 4); else d_depth.reserve(col_cap * 2 + 64);

  FastParams P;
  memset(&P, 0, sizeof(P));
  P.blob = d_blob; P.rec_off = d_rec_off; P.rec_len = d_rec_len; P.grp_first = d_grp_first;
  P.blob_len = blob_len;
  P.g0 = 0;
  P.T = c->d_tables.as<DeviceTables>(); P.TU = c->d_umi_tables.as<DeviceTables>();
  if (!d_fwimg.p) {   // the tables of a caller never change: one image per FastPath (k_family_wave's LDS block)
    FwLds* img = new FwLds;
    build_fw_image(*img, c->h_tables.t);
    d_fwimg.reserve(sizeof(FwLds));
    hip_check(hipMemcpyAsync(d_fwimg.p, img, sizeof(FwLds), hipMemcpyHostToDevice, s), "H2D fw image");
    FGX_SYNC(s);
    delete img;
  }
  P.fw_image = d_fwimg.p;
  P.min_reads = o.min_reads; P.max_reads = o.max_reads;
  P.min_input_bq = o.min_input_base_quality; P.min_cons_bq = o.min_consensus_base_quality;
  {   // k_split_cols's sum-free observation step: from how many agreeing observations a column is the cap whatever their qualities (gate_core.h)
    static const int nosum_env = [] { const char* e = fgx_knob("FGX_S2_NOSUM"); return e ? atoi(e) : 1; }();      // (measurement knob: 0 = every end through the f32 sums)
    static const int packed_env = [] { const char* e = getenv("FGX_S2_PACKED"); return e ? atoi(e) : 1; }();   // (measurement knob: 0 = the 64-column passes only)
    static const int s2dbg_env = [] { const char* e = fgx_knob("FGX_S2_DEBUG"); return e ? atoi(e) : 0; }();
    P.s2_packed = (packed_env ? 1u : 0u) | ((packed_env && s2dbg_env) ? 2u : 0u);
    P.s2_nsafe = nosum_env ? unanimous_cap_depth(c->h_tables.t, (uint32_t)o.min_input_base_quality & 0xFFu, 64u) : FGX_NEVER_CAP;
  }
  P.trim = o.trim; P.overlap = o.overlapping_consensus;
  if (duplex) {   // single-strand caller of the duplex caller (duplex_caller.rs:474-489): min_reads 1, min consensus base quality 2
    P.min_reads = 1; P.max_reads = -1; P.min_cons_bq = FGX_MIN_PHRED;
    P.dmin_total = o.duplex_min_reads[0]; P.dmin_xy = o.duplex_min_reads[1]; P.dmin_yx = o.duplex_min_reads[2];
    P.dmax_reads = o.duplex_max_reads_per_strand;
    P.col_obs = d_obs.as<uint32_t>(); P.dends = d_ends.as<DuplexDesc>();
  }
  if (codec) {    // single-strand caller of the CODEC caller (codec_caller.rs:374-397): min_reads 1, no cap, min consensus base quality 0
    P.min_reads = 1; P.max_reads = -1; P.min_cons_bq = 0; P.trim = 0; P.overlap = 0;
    P.cends = d_ends.as<CodecDesc>(); P.cmin_reads = o.codec_min_reads_per_strand; P.cmax_reads = o.codec_max_reads_per_strand;
    P.cmin_duplex_len = o.codec_min_duplex_length;
  }
  P.per_base_tags = o.produce_per_base_tags; P.track_rejects = o.track_rejects;
  P.tag0 = (duplex || codec) ? 'M' : o.tag[0]; P.tag1 = (duplex || codec) ? 'I' : o.tag[1]; P.cell0 = o.cell_tag[0]; P.cell1 = o.cell_tag[1];
  P.prefix_len = (uint32_t)c->prefix.size(); P.rg_len = (uint32_t)c->rg.size();
  P.ends = d_ends.as<EndDesc>(); P.rec_sizes = d_sizes.as<uint64_t>();
  P.col_code = d_code.as<uint8_t>(); P.col_qual = d_qual.as<uint8_t>(); P.col_depth = d_depth.as<uint16_t>(); P.col_err = d_err.as<uint16_t>();
  if (meth_dev) {
    const GenomeRef* gr = c->genome.get();
    const uint32_t n_ref = (uint32_t)gr->off.size();
    d_mcontigs.reserve((size_t)(n_ref + 1) * 16 + 64);
    if (n_ref) {
      hip_check(hipMemcpyAsync(d_mcontigs.p, gr->off.data(), (size_t)n_ref * 8, hipMemcpyHostToDevice, s), "H2D contig offsets");
      hip_check(hipMemcpyAsync(d_mcontigs.as<uint64_t>() + n_ref, gr->len.data(), (size_t)n_ref * 8, hipMemcpyHostToDevice, s), "H2D contig lengths");
    }
    P.meth_mode = o.methylation_mode; P.n_ref = n_ref; P.genome = (const uint8_t*)gr->d_genome.p;
    P.contig_off = d_mcontigs.as<uint64_t>(); P.contig_len = d_mcontigs.as<uint64_t>() + n_ref;
    P.meth_flag = d_mflag.as<uint8_t>(); P.meth_u = d_mu.as<uint16_t>(); P.meth_t = d_mt.as<uint16_t>();
  }
  P.col_base = d_colbase.as<uint64_t>();
  if (direct) {
    P.dir_size = d_dir_size.as<uint32_t>(); P.dir_off = d_dir_off.as<uint64_t>(); P.out = d_out.as<uint8_t>(); P.out_cap = dir_cap;
    P.out_off = d_offsets.as<uint64_t>(); P.slot_desc = d_slot_desc.as<SlotDesc>(); P.slot_err = d_slot_err.as<uint32_t>();
    P.strings = d_strings.as<char>(); P.dir_flags = (uint32_t*)(misc + 36);
  }
  P.stats = d_statslots.as<unsigned long long>(); P.n_deferred = (uint32_t*)(misc + 29);
  P.deferred = d_deferred.as<uint32_t>();
  d_retry.reserve((size_t)n_grp * 4);
  P.retry = d_retry.as<uint32_t>(); P.n_retry = (uint32_t*)(misc + 31);
  P.group_list = nullptr;
  // append lists for the columns that need call_full: room for 1/8 of the column bound (overflow → general path)
  uint64_t full_total = col_cap / pool_div + (uint64_t)N_LISTS * pool_slack;
  uint32_t full_cap = (uint32_t)std::min<uint64_t>(full_total / N_LISTS, 0x7FFFFFFFull);
  d_full_items.reserve((size_t)full_cap * N_LISTS * sizeof(FullItem));
  d_full_count.reserve((size_t)N_LISTS * 4);
  hip_check(hipMemsetAsync(d_full_count.p, 0, (size_t)N_LISTS * 4, s), "memset");
  P.full_items = d_full_items.as<FullItem>(); P.full_count = d_full_count.as<uint32_t>(); P.full_cap = full_cap;
  uint32_t wave_bytes = duplex ? lds_wave_bytes_duplex : codec ? lds_wave_bytes_codec : lds_wave_bytes;
  if (const char* e = fgx_knob("FGX_WAVE_BYTES")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1024 && v <= 22016) wave_bytes = v & ~15u; }   // tuning knob
  P.lds_wave_bytes = wave_bytes;
  P.lds_tile_bytes = lds_tile_bytes_large;

  // Wave-per-family launches over growing LDS slices: everything first, then only the groups whose records did not fit
  // (long-tail families: 6 KB holds ~18 records of 150 bp, 12 KB ~36, 22 KB all 64 a wavefront can take).
  {
    const uint32_t stages[3] = {wave_bytes, 12288u, 22016u};
    if (!lds_attr_set) {
      hip_check(hipFuncSetAttribute((const void*)k_family_wave<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 22016), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_family_wave<0>: the device refused the dynamic LDS size");
      hip_check(hipFuncSetAttribute((const void*)k_family_wave<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 22016), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_family_wave<1>: the device refused the dynamic LDS size");
      hip_check(hipFuncSetAttribute((const void*)k_family_wave<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 22016), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_family_wave<2>: the device refused the dynamic LDS size");
      lds_attr_set = true;
    }
    d_retry2.reserve((size_t)n_grp * 4);
    uint32_t* lists[2] = {d_retry.as<uint32_t>(), d_retry2.as<uint32_t>()};
    // simplex families of more than 64 records: no wavefront-per-family kernel takes them — the first kernel that sees one puts it on
    // k_family's list (round 3 walked them through k_simplex_wave2 and three k_family_wave<0> launches first: 2 ms per 1 M long-tail families)
    uint32_t* d_cnt_big = (uint32_t*)(misc + 37);
    uint32_t h_route_seen = 0, h_big_seen = 0;
    bool split_counts_seen = false;
    // (not with direct records: their merge pass walks ONE list of the families that left the split pipeline — the route list)
    if (!duplex && !codec && !direct) { d_big.reserve((size_t)n_grp * 4); P.big = d_big.as<uint32_t>(); P.n_big = d_cnt_big; }
    uint32_t* d_cnt = (uint32_t*)(misc + 31);
    uint32_t n_cur = meth_dev ? 0u : n_grp;       // (methylation-aware mode: no wavefront-per-family kernel runs)
    const uint32_t* cur_list = nullptr;
    int out_list = 0;
    // Simplex, no --trim: k_simplex_wave2 (two-accumulator column loop) takes the families of the common record shape over the same
    // growing LDS slices; what is outside its shape is collected in `retry_old` and goes through the k_family_wave<0> launches below.
    uint32_t n_v2 = n_grp;
    const uint32_t* v2_list = nullptr;
    last_split_chunks = 0;
    if (use_split) {
      // k_split_cols over growing LDS slices (4 / 2 / 1 / 1 wavefronts per workgroup); what it does not take is collected in
      // `route` and starts the k_simplex_wave2 chain below
        if (!s2_attr_set) {
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<0, 0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<0, 0, 0, 0>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<160, 80, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<160, 80, 0, 0>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<0, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<0, 0, 0, 1>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<160, 80, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<160, 80, 0, 1>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<0, 0, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<0, 0, 0, 2>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<160, 80, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<160, 80, 0, 2>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<0, 0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<0, 0, 1, 0>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_split_cols<160, 80, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_split_cols<160, 80, 1, 0>: the device refused the dynamic LDS size");
        s2_attr_set = true;
      }
      if (!d_s2img.p) {   // the tables of a caller never change: one image per FastPath
        S2Image* img = new S2Image;
        build_s2_image(*img, c->h_tables.t);
        d_s2img.reserve(sizeof(S2Image));
        hip_check(hipMemcpyAsync(d_s2img.p, img, sizeof(S2Image), hipMemcpyHostToDevice, s), "H2D s2 image");
        FGX_SYNC(s);
        delete img;
      }
      d_route.reserve((size_t)n_grp * 4);
      uint32_t* d_cnt_route = (uint32_t*)(misc + 33);
      P.s2_image = d_s2img.p; P.fam_desc = d_famdesc.as<uint4>();
      P.split_rec = d_split_rec.as<SplitRec>(); P.split_fam = d_split_fam.as<SplitFam>();
      d_split_out.reserve((size_t)n_grp * sizeof(SplitOut));
      P.split_out = d_split_out.as<SplitOut>();
      P.route = d_route.as<uint32_t>(); P.n_route = d_cnt_route;
      // ---- the record kernel, chunk by chunk on a second stream: chunk i + 1 (waiting on memory most of its time) runs under the column
      //      kernel of chunk i (bound by vector-instruction issue) ----------------------------------------------------------------------
      d_split_rec.reserve((size_t)n_rec * sizeof(SplitRec) + 64);
      d_split_fam.reserve((size_t)n_grp * sizeof(SplitFam) + 64);
      P.split_rec = d_split_rec.as<SplitRec>(); P.split_fam = d_split_fam.as<SplitFam>();
      split_streams();
      FastParams PK;
      split_parse_params(PK);
      if (direct) { PK.dir_size = P.dir_size; PK.min_input_bq = o.min_input_base_quality; PK.per_base_tags = o.produce_per_base_tags; PK.rg_len = (uint32_t)c->rg.size(); }
      uint32_t fpw, n_chunks, chunk_fam;
      {
        uint32_t chunk_fam_raw;
        split_geometry(fpw, n_chunks, chunk_fam_raw, chunk_fam);
        last_split_chunks = n_chunks;
        dir_chunks_run = (n_grp + chunk_fam_raw - 1) / chunk_fam_raw;     // chunks that hold families (the rounding of chunk_fam can leave the last ones empty)
      }
      hip_check(hipEventRecord(ev_chunk[MAX_CHUNKS - 1], s), "event");      // (the second stream starts where this one is: buffers, memsets)
      hip_check(hipStreamWaitEvent(s2, ev_chunk[MAX_CHUNKS - 1], 0), "wait");
      size_t dir_scan_bytes = 0;
      if (direct) {
        const DirSizeIn in(d_dir_size.as<uint32_t>(), DirCast());
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, dir_scan_bytes, in, d_dir_off.as<uint64_t>(), (int)std::min<uint64_t>(chunk_fam, n_grp), s2);
        d_scan_tmp2.reserve(dir_scan_bytes + 64);
      }
      auto launch_parse = [&](uint32_t ci) {
        const uint32_t ga = ci * chunk_fam, gb = std::min<uint64_t>((uint64_t)ga + chunk_fam, n_grp);
        if (ga >= gb) return;
        const uint64_t waves = ((uint64_t)(gb - ga) + fpw - 1) / fpw;
        do { last_launches++; hipLaunchKernelGGL(k_split_parse, dim3((uint32_t)((waves + 1) / 2)), dim3(128), 0, s2, PK, ga, gb, fpw, (uint64_t*)nullptr, 3u, (uint4*)nullptr); } while (0);
        hip_check(hipGetLastError(), "k_split_parse launch");
        if (direct) {   // the chunk's record offsets: exclusive scan of the predicted family sizes, carried on from the chunk before
          const DirSizeIn in(d_dir_size.as<uint32_t>() + ga, DirCast());
          hip_check(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp2.p, dir_scan_bytes, in, d_dir_off.as<uint64_t>() + ga, (int)(gb - ga), s2), "scan of the record sizes");
          do { last_launches++; hipLaunchKernelGGL(k_dir_carry, dim3((gb - ga + 255) / 256), dim3(256), 0, s2, d_dir_off.as<uint64_t>() + ga, d_dir_size.as<uint32_t>() + ga, gb - ga,
                             d_dir_base.as<uint64_t>() + ci); } while (0);
          hip_check(hipGetLastError(), "k_dir_carry launch");
        }
        hip_check(hipEventRecord(ev_chunk[ci], s2), "event");
      };
      if (!early_parse) launch_parse(0);
      // the tile strides of a sample of families decide which build of k_split_cols goes first: 64 families spread evenly over the first chunk (one strided
      // copy; round 5 took the batch's FIRST 64 — the head of a coordinate-sorted file need not look like the rest)
      SplitFam fam_sample[64];
      const uint32_t n_sample = n_grp < 64u ? n_grp : 64u;
      const uint32_t sample_span = (uint32_t)std::min<uint64_t>(chunk_fam, n_grp), sample_stride = std::max<uint32_t>(1u, sample_span / 64u);
      hip_check(hipMemcpy2DAsync(fam_sample, sizeof(SplitFam), d_split_fam.p, (size_t)sample_stride * sizeof(SplitFam), sizeof(SplitFam), n_sample, hipMemcpyDeviceToHost, s2), "D2H sample");
      hip_check(hipEventRecord(ev_sample, s2), "event");
      // (round 5) the record kernel is bound by HBM bandwidth (it reads the whole blob: 4.2 TB/s alone on the chip), and a column kernel
      // beside it gets what is left — with every chunk's record kernel queued up front, the first column kernel took 6.9 ms instead of
      // 1.7 and the two streams added up like one (gpurun_out timeline, profiles/r05_timeline_*.txt).  Paced: chunk k + 2 is parsed
      // when the column kernel of chunk k has finished, i.e. under the column kernel of chunk k + 1.
      static const int pace_env = [] { const char* e = fgx_knob("FGX_S2_PACE"); return e ? atoi(e) : 1; }();      // (measurement knob: 0 = all record kernels up front)
      const bool paced = pace_env != 0 && n_chunks > 2;
      for (uint32_t ci = 1; ci < (paced ? 2u : n_chunks); ci++) launch_parse(ci);
      do { last_host_syncs++; hip_check(hipEventSynchronize(ev_sample), "sync"); } while (0);
      // LDS slice of the first launch: the MEAN family's tile (rows of 160 + 80 bytes) + room for its k_call_full items, at least the
      // 4352 bytes of a 16-record family — a deeper library starts at the slice its families need instead of failing the first launch
      // as a whole (depth 12: every family took two launches, 32 ms per 1 M families)
      static const uint32_t s2_bytes_env = [] { const char* e = fgx_knob("FGX_S2_BYTES"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 2048 && v <= 32768 ? (v & ~15) : 0); }();
      static const uint32_t s2_wpb_env = [] { const char* e = fgx_knob("FGX_S2_WPB"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 4 ? v : 0); }();
      // (round 5: the packed pass sends every column it does not answer itself to k_call_full — also the one-base columns of too few
      // observations, which run_cols's gates answer — and keeps an 8-byte descriptor per such column at the top of the slice: 56 bytes per
      // column, ~14 columns per depth-8 family of `simulate` data; 5632 bytes x 4 wavefronts still leave a CU six workgroups)
      bool s2_packed_on = !direct && P.s2_packed != 0 && P.s2_nsafe != FGX_NEVER_CAP && P.min_reads <= P.s2_nsafe && ((uint32_t)P.min_input_bq & 0xFFu) <= 128u;   // (what run_cols_packed asks of the caller's options)
      // which of the two first-stage kernels the batch needs: by the sampled families (a family of the other kind still finds its way: the
      // packed kernel without a partner hands it to the next launch, the partner kernel alone IS the classic kernel)
      uint32_t n_pk = 0;
      if (s2_packed_on) for (uint32_t i = 0; i < n_sample; i++) {
        const SplitFam& F = fam_sample[i];
        auto ok = [&](uint32_t m) { return m < 2u || (m >= P.s2_nsafe && m <= S2_PACKED_MAX_ROWS); };
        n_pk += (ok(F.m_a) && ok(F.m_b) && ((F.len_a + 7u) >> 3) + ((F.len_b + 7u) >> 3) <= 64u) ? 1u : 0u;
      }
      static const int s2_partner_env = [] { const char* e = fgx_knob("FGX_S2_PARTNER"); return e ? atoi(e) : -1; }();   // (measurement knob: 1 = always both kernels, 0 = never)
      if (s2_packed_on && n_sample && 100ull * n_pk < (unsigned long long)n_sample) s2_packed_on = false;               // under 1 % of its shape: the classic kernel alone
      const bool s2_partner = s2_packed_on && (s2_partner_env >= 0 ? s2_partner_env != 0 : 100ull * n_pk < 99ull * n_sample);
      last_split_build = s2_packed_on ? (s2_partner ? 2u : 1u) : 0u;
      uint32_t s2_bytes0 = s2_packed_on ? 5632 : 4352;
      {
        const uint32_t mean_need = (uint32_t)(mean_recs + 0.999) * 240u + 16u + (s2_packed_on ? 1680u : 400u);
        if (mean_need > s2_bytes0) s2_bytes0 = std::min<uint32_t>((mean_need + 15u) & ~15u, 17408u);
      }
      if (s2_bytes_env) s2_bytes0 = s2_bytes_env;
      const uint32_t s2_wpb = s2_wpb_env ? s2_wpb_env : (s2_bytes0 <= 6528u ? 4u : s2_bytes0 <= 13056u ? 2u : 1u);
      // rows of 160 + 80 bytes (reads up to 160 bases) have their own build: member rows at immediate offsets in the column loop
      static const int s2_fixed_env = [] { const char* e = fgx_knob("FGX_S2_FIXED"); return e ? atoi(e) : -1; }();   // (measurement knob: 0 / 1)
      uint32_t n160 = 0;
      for (uint32_t i = 0; i < n_sample; i++) n160 += (fam_sample[i].qs == 160 && fam_sample[i].ss == 80) ? 1u : 0u;
      const bool s2_fixed = s2_fixed_env >= 0 ? s2_fixed_env != 0 : 2 * n160 > n_sample;
      struct S2Stage { uint32_t bytes, wpb; bool fixed; };
      std::vector<S2Stage> st2;
      // (reads of up to 160 bases: every slice size in the build with the strides as immediates, then ONE launch of the generic build for
      // the families with other strides; round 3 ran the generic build from the second slice on — a long-tail batch spends most of its
      // column time there)
      if (s2_fixed) {
        st2.push_back({s2_bytes0, s2_wpb, true});
        st2.push_back({2 * s2_bytes0 > 8704u ? 2 * s2_bytes0 : 8704u, 2u, true});
        st2.push_back({17408u, 1u, true});
        st2.push_back({34816u, 1u, true});
        st2.push_back({34816u, 1u, false});
      } else {
        st2.push_back({s2_bytes0, s2_wpb, false});
        st2.push_back({2 * s2_bytes0 > 8704u ? 2 * s2_bytes0 : 8704u, 2u, false});
        st2.push_back({17408u, 1u, false});
        st2.push_back({34816u, 1u, false});
      }
      uint32_t n_s2 = n_grp;
      const uint32_t* s2_list = nullptr;
      int s2_out = 0;
      for (size_t ci = 0; ci < st2.size() && n_s2; ci++) {
        if (ci > 0 && st2[ci - 1].fixed == st2[ci].fixed && st2[ci].bytes <= st2[ci - 1].bytes) continue;
        const bool last = ci + 1 == st2.size();
        hip_check(hipMemsetAsync(d_cnt, 0, 4, s), "memset");
        FastParams PS = P;
        PS.group_list = s2_list; PS.lds_wave_bytes = st2[ci].bytes;
        static const bool later_packed = [] { const char* e = fgx_knob("FGX_S2_LATER_PACKED"); return !(e && e[0] == '0'); }();   // (measurement knob: 0 = the classic build in the later stages, as in round 5)
        PS.s2_partner = (s2_partner || (ci > 0 && later_packed && s2_packed_on)) ? 1u : 0u;
        PS.retry = last ? nullptr : lists[s2_out]; PS.n_retry = d_cnt;
        if (st2[ci].bytes > 65536u) continue;                 // (more than the attribute limit set above: the family goes down the chain)
        const uint32_t wpb = std::min<uint32_t>(st2[ci].wpb, 65536u / st2[ci].bytes);   // wpb x slice within the 64 KiB requested for k_split_cols
        const size_t lds = (size_t)wpb * st2[ci].bytes;
        auto launch_cols = [&](uint32_t g_first, uint32_t count) {
          PS.g0 = g_first;
          const dim3 grid((count + wpb - 1) / wpb), block(64 * wpb);
          if (direct) {
            if (st2[ci].fixed) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<160, 80, 1, 0>), grid, block, lds, s, PS, count); } while (0);
            else do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<0, 0, 1, 0>), grid, block, lds, s, PS, count); } while (0);
          } else if (s2_packed_on && (ci == 0 || later_packed)) {
            // (round 5) the first stage as two launches over the same families: the packed pass for the families of its shape, run_cols for the rest;
            // (round 6) the later stages — the families that need larger LDS slices: ends of 9 .. 31 rows — likewise, always as the pair (their lists are mixed)
            const bool pair = s2_partner || ci > 0;
            if (st2[ci].fixed) {
              do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<160, 80, 0, 1>), grid, block, lds, s, PS, count); } while (0);
              if (pair) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<160, 80, 0, 2>), grid, block, lds, s, PS, count); } while (0);
            } else {
              do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<0, 0, 0, 1>), grid, block, lds, s, PS, count); } while (0);
              if (pair) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<0, 0, 0, 2>), grid, block, lds, s, PS, count); } while (0);
            }
          } else if (st2[ci].fixed) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<160, 80, 0, 0>), grid, block, lds, s, PS, count); } while (0);
          else do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_split_cols<0, 0, 0, 0>), grid, block, lds, s, PS, count); } while (0);
        };
        if (ci == 0) {   // the first stage takes the families in file order, chunk by chunk behind the record kernel
          for (uint32_t k = 0; k < n_chunks; k++) {
            const uint32_t ga = k * chunk_fam, gb = (uint32_t)std::min<uint64_t>((uint64_t)ga + chunk_fam, n_grp);
            if (ga >= gb) break;
            hip_check(hipStreamWaitEvent(s, ev_chunk[k], 0), "wait");
            launch_cols(ga, gb - ga);
            // the chunk's EndDescs / record sizes / counters (a thread per family: waits on memory, few instructions) on the second
            // stream, under the next chunk's column kernel
            hip_check(hipEventRecord(ev_cols[k], s), "event");
            hip_check(hipStreamWaitEvent(s2, ev_cols[k], 0), "wait");
            FastParams PF = P;
            PF.group_list = nullptr; PF.g0 = ga;
            do { last_launches++; hipLaunchKernelGGL(k_split_finish, dim3((gb - ga + 255) / 256), dim3(256), 0, s2, PF, gb - ga); } while (0);
            if (paced && k + 2 < n_chunks) launch_parse(k + 2);
          }
        } else {
          launch_cols(0u, n_s2);
          FastParams PF = P;                       // the families this (larger-slice) launch finished: the list it was given
          PF.group_list = s2_list; PF.g0 = 0;
          do { last_launches++; hipLaunchKernelGGL(k_split_finish, dim3((n_s2 + 255) / 256), dim3(256), 0, s, PF, n_s2); } while (0);
        }
        hip_check(hipGetLastError(), "k_split_cols launch");
        // (round 6: the route / big counters come along — after the LAST stage they are final, and two host synchronisations of their own go away;
        // the three are words 31, 33 and 37 of `misc`: ONE copy of its words 31 .. 37)
        unsigned long long h_cnt7[7] = {0, 0, 0, 0, 0, 0, 0};
        static_assert(sizeof(h_cnt7) == 56, "misc words 31 .. 37");
        hip_check(hipMemcpyAsync(h_cnt7, misc + 31, sizeof(h_cnt7), hipMemcpyDeviceToHost, s), "D2H");
        split_counts_seen = true;
        FGX_SYNC(s);
        const uint32_t n_next = (uint32_t)h_cnt7[0];
        h_route_seen = (uint32_t)h_cnt7[2]; h_big_seen = (uint32_t)h_cnt7[6];
        if (ci == 0) last_first_stage_retries = PS.retry ? n_next : 0u;
        s2_list = lists[s2_out]; n_s2 = PS.retry ? n_next : 0; s2_out ^= 1;
      }
      hip_check(hipEventRecord(ev_fin, s2), "event");          // the per-chunk k_split_finish launches
      hip_check(hipStreamWaitEvent(s, ev_fin, 0), "wait");
      uint32_t n_route = h_route_seen;
      if (!split_counts_seen) {
        hip_check(hipMemcpyAsync(&n_route, d_cnt_route, 4, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
      }
      n_v2 = n_route; v2_list = d_route.as<uint32_t>();
      last_routed = n_route;
      static const bool s2_verbose = [] { const char* e = fgx_knob("FGX_S2_VERBOSE"); return e && e[0] == '1'; }();
      if (s2_verbose) fprintf(stderr, "[fgx] split pipeline: %u families, %u routed to k_simplex_wave2\n", n_grp, n_route);
    }
    if (simplex_v2) {
      // Launch chain: k_simplex_seg<4> / <2> (4 / 2 families per wavefront, while the mean family fits a quarter / half of the
      // wave's LDS) → k_simplex_wave2 over the growing slices.  A family that does not fit a launch moves to the next one.
        if (!v2_attr_set) {
        hip_check(hipFuncSetAttribute((const void*)k_simplex_wave2, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 22016), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_simplex_wave2: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_simplex_seg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 32768), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_simplex_seg<2>: the device refused the dynamic LDS size");
        hip_check(hipFuncSetAttribute((const void*)k_simplex_seg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WAVES_PER_BLOCK * 32768), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_simplex_seg<4>: the device refused the dynamic LDS size");
        v2_attr_set = true;
      }
      if (!d_w2img.p) {   // the tables of a caller never change: one image per FastPath
        W2Lds img;
        build_w2_image(img, c->h_tables.t);
        d_w2img.reserve(sizeof(W2Lds));
        hip_check(hipMemcpyAsync(d_w2img.p, &img, sizeof(W2Lds), hipMemcpyHostToDevice, s), "H2D w2 image");
        FGX_SYNC(s);             // (`img` is on this stack frame)
      }
      P.w2_image = d_w2img.p;
      P.fam_desc = d_famdesc.as<uint4>();
      struct Stage { int fam_per_wave; uint32_t bytes; uint32_t wpb; };
      std::vector<Stage> chain;
      static const uint32_t seg_wpb = [] { const char* e = fgx_knob("FGX_SEG_WPB"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= WAVES_PER_BLOCK ? v : WAVES_PER_BLOCK); }();   // (measurement knob)
      if (use_seg && mean_span + (16 * 8 + 64) <= seg_bytes / 4) chain.push_back({4, seg_bytes, seg_wpb});
      // (two families per wavefront measured slower than one at depth 8 — 14.9 vs 10.7 ms per 1 M families: the column phase costs
      // the same per family and a CU holds 12 instead of 20 wavefronts; kept behind FGX_SEG2=1 for experiments)
      static const bool use_seg2 = [] { const char* e = fgx_knob("FGX_SEG2"); return e && e[0] == '1'; }();
      if (use_seg && use_seg2 && mean_span + (32 * 8 + 64) <= seg_bytes / 2) chain.push_back({2, seg_bytes, (uint32_t)WAVES_PER_BLOCK});
      // (Two workgroup-cooperative variants were built, verified byte-identical and measured slower — the record phases of four
      // families on ONE wavefront while the other three wait at a barrier, 11.9 ms, and the same as a producer / consumer pipeline
      // in persistent workgroups, 19.3 ms, against 10.2 ms per 1 M depth-8 families here: they execute 20 % fewer vector and 43 %
      // fewer scalar instructions, but a lone wavefront retires an instruction every ~30 cycles, and the CU is fed by the number of
      // independent wavefronts, not by lane utilisation.  profiles/r02c_pmc_1M_blk.json, r02d_pmc_1M_pipe.json; HISTORY.md §4.)
      // wavefronts per workgroup of k_simplex_wave2's first launch: the LDS of a workgroup is freed when its SLOWEST wavefront is
      // done, so small workgroups keep more wavefronts running (FGX_W2_WPB: measurement knob)
      static const uint32_t w2_wpb = [] { const char* e = fgx_knob("FGX_W2_WPB"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= WAVES_PER_BLOCK ? v : FGX_W2_WPB_DEFAULT); }();
      for (int st = 0; st < 3; st++) if (st == 0 || stages[st] > stages[st - 1]) chain.push_back({1, stages[st], st == 0 ? w2_wpb : st == 1 ? 2u : 1u});
      d_retry_old.reserve((size_t)n_grp * 4);
      uint32_t* d_cnt_old = (uint32_t*)(misc + 32);
      int v2_out = 0;
      bool v2_ran = false;
      for (size_t ci = 0; ci < chain.size() && n_v2; ci++) {
        v2_ran = true;
        const Stage& S = chain[ci];
        const bool last = ci + 1 == chain.size();
        hip_check(hipMemsetAsync(d_cnt, 0, 4, s), "memset");
        FastParams PS = P;
        PS.group_list = v2_list; PS.lds_wave_bytes = S.bytes;
        PS.retry = last ? nullptr : lists[v2_out]; PS.n_retry = d_cnt;
        PS.retry_old = d_retry_old.as<uint32_t>(); PS.n_retry_old = d_cnt_old;
        const uint32_t fpb = S.wpb * (uint32_t)S.fam_per_wave;      // families per workgroup
        const dim3 grid((n_v2 + fpb - 1) / fpb), block(64 * S.wpb);
        const size_t lds = (size_t)S.wpb * S.bytes;
        if (S.fam_per_wave == 4) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_simplex_seg<4>), grid, block, lds, s, PS, n_v2); } while (0);
        else if (S.fam_per_wave == 2) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_simplex_seg<2>), grid, block, lds, s, PS, n_v2); } while (0);
        else do { last_launches++; hipLaunchKernelGGL(k_simplex_wave2, grid, block, lds, s, PS, n_v2); } while (0);
        hip_check(hipGetLastError(), "k_simplex launch");
        uint32_t n_next = 0;
        hip_check(hipMemcpyAsync(&n_next, d_cnt, 4, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
        v2_list = lists[v2_out]; n_v2 = PS.retry ? n_next : 0; v2_out ^= 1;
      }
      uint32_t n_old = 0;
      if (v2_ran) {   // (no kernel of the chain has run — the split pipeline kept every family —: nothing to read, no synchronisation)
        hip_check(hipMemcpyAsync(&n_old, d_cnt_old, 4, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
      }
      n_cur = n_old; cur_list = d_retry_old.as<uint32_t>();
      out_list = 0;
    }
    for (int st = 0; st < 3 && n_cur; st++) {
      if (st > 0 && stages[st] <= stages[st - 1]) continue;
      const bool last = st == 2;
      hip_check(hipMemsetAsync(d_cnt, 0, 4, s), "memset");
      FastParams PS = P;
      PS.group_list = cur_list; PS.lds_wave_bytes = stages[st];
      PS.retry = (last && (duplex || codec)) ? nullptr : lists[out_list]; PS.n_retry = d_cnt;
      // fewer wavefronts per workgroup as the slices grow: the LDS a workgroup asks for is what limits the wavefronts a CU holds
      // (4 x 22 KB = one workgroup = 4 waves per CU; 1 x 22 KB = six workgroups = 6 waves)
      static const uint32_t fw_wpb = [] { const char* e = fgx_knob("FGX_FW_WPB"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= WAVES_PER_BLOCK ? v : WAVES_PER_BLOCK); }();   // (measurement knob)
      const uint32_t wpb = st == 0 ? fw_wpb : st == 1 ? 2u : 1u;
      const dim3 grid((n_cur + wpb - 1) / wpb), block(64 * wpb);
      const size_t lds = (size_t)wpb * stages[st];
      if (codec) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_family_wave<2>), grid, block, lds, s, PS, n_cur); } while (0);
      else if (duplex) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_family_wave<1>), grid, block, lds, s, PS, n_cur); } while (0);
      else do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_family_wave<0>), grid, block, lds, s, PS, n_cur); } while (0);
      hip_check(hipGetLastError(), "k_family_wave launch");
      uint32_t n_next = 0;
      hip_check(hipMemcpyAsync(&n_next, d_cnt, 4, hipMemcpyDeviceToHost, s), "D2H");
      FGX_SYNC(s);
      cur_list = lists[out_list]; n_cur = PS.retry ? n_next : 0; out_list ^= 1;
    }
    uint32_t n_big = 0;
    if (!duplex && !codec) {
      if (split_counts_seen && last_routed == 0) n_big = h_big_seen;   // (no kernel after the split stages has run: what their last synchronisation read is final)
      else {
        hip_check(hipMemcpyAsync(&n_big, d_cnt_big, 4, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
      }
    }
    if (meth_dev) {
      d_big.reserve((size_t)n_grp * 4);
      do { last_launches++; hipLaunchKernelGGL(k_iota, dim3((n_grp + 255) / 256), dim3(256), 0, s, d_big.as<uint32_t>(), n_grp); } while (0);
      n_big = n_grp;
    }
    last_big_families = n_big;
    last_deep_families = 0;
    // ---- deep families: k_deep_parse + k_deep_cols (simplex_deep.inc) take the big list; what is outside their shape goes on to k_family ----
    const bool use_deep = [] { const char* e = getenv("FGX_DEEP"); return !(e && e[0] == '0'); }();   // (read per batch)
    const uint32_t* big_list = d_big.as<uint32_t>();
    if (n_big && (use_deep || meth_dev) && !o.trim) {
      if (!d_s2img.p) {   // (the column kernel's LDS tables: the split pipeline's image)
        S2Image* img = new S2Image;
        build_s2_image(*img, c->h_tables.t);
        d_s2img.reserve(sizeof(S2Image));
        hip_check(hipMemcpyAsync(d_s2img.p, img, sizeof(S2Image), hipMemcpyHostToDevice, s), "H2D s2 image");
        FGX_SYNC(s);
        delete img;
      }
      P.s2_image = d_s2img.p;
      uint32_t* d_cnt_deep = (uint32_t*)(misc + 38);
      // one pass of the streaming kernels over a family list; returns how many families it handed on (to `out_list`)
      // `size_class`: 0 = the wavefront-sized build of the record kernel (families of up to 64 records), 1 = two wavefronts (up to 128), 2 = four (up to DEEP_MAX)
      auto deep_pass = [&](const uint32_t* list, uint32_t n_list, int size_class, uint32_t* out_list) -> uint32_t {
        d_deep_sizes.reserve((size_t)n_list * 8 + 64); d_deep_row0.reserve((size_t)n_list * 8 + 64);
        do { last_launches++; hipLaunchKernelGGL(k_deep_sizes, dim3((n_list + 255) / 256), dim3(256), 0, s, list, n_list, d_grp_first, d_deep_sizes.as<uint64_t>()); } while (0);
        size_t tb = 0;
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, d_deep_sizes.as<uint64_t>(), d_deep_row0.as<uint64_t>(), (int)n_list, s);
        d_scan_tmp.reserve(tb);
        hip_check(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp.p, tb, d_deep_sizes.as<uint64_t>(), d_deep_row0.as<uint64_t>(), (int)n_list, s), "scan of the deep families' records");
        uint64_t lastr[2] = {0, 0};
        hip_check(hipMemcpyAsync(&lastr[0], d_deep_row0.as<uint64_t>() + (n_list - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
        hip_check(hipMemcpyAsync(&lastr[1], d_deep_sizes.as<uint64_t>() + (n_list - 1), 8, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
        const uint64_t n_rows = lastr[0] + lastr[1];
        if (n_rows > (uint64_t)n_rec) throw std::runtime_error("device pipeline: " + std::to_string(n_rows) + " rows for the streaming kernels, more than the batch has records");
        d_deep_rows.reserve((size_t)n_rows * sizeof(DeepRow) + 64); d_deep_fams.reserve((size_t)n_list * sizeof(DeepFam) + 64);
        hip_check(hipMemsetAsync(d_cnt_deep, 0, 4, s), "memset");
        DeepParams DP;
        DP.list = list; DP.n_list = n_list; DP.row0 = d_deep_row0.as<uint64_t>();
        DP.rows = d_deep_rows.as<DeepRow>(); DP.fams = d_deep_fams.as<DeepFam>(); DP.out_list = out_list; DP.n_out = d_cnt_deep;
        FastParams PD = P;
        PD.group_list = nullptr;
        if (size_class == 0) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deep_parse<64, 64>), dim3(n_list), dim3(64), 0, s, PD, DP); } while (0);
        else if (size_class == 1) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deep_parse<128, 128>), dim3(n_list), dim3(128), 0, s, PD, DP); } while (0);
        else do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deep_parse<256, DEEP_MAX>), dim3(n_list), dim3(256), 0, s, PD, DP); } while (0);
        hip_check(hipGetLastError(), "k_deep_parse launch");
        if (meth_dev) do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deep_cols<1>), dim3((n_list + 3) / 4), dim3(256), 0, s, PD, DP); } while (0);
        else do { last_launches++; hipLaunchKernelGGL(HIP_KERNEL_NAME(k_deep_cols<0>), dim3((n_list + 3) / 4), dim3(256), 0, s, PD, DP); } while (0);
        hip_check(hipGetLastError(), "k_deep_cols launch");
        uint32_t n_left = 0;
        hip_check(hipMemcpyAsync(&n_left, d_cnt_deep, 4, hipMemcpyDeviceToHost, s), "D2H");
        FGX_SYNC(s);
        return n_left;
      };
      d_deep_out.reserve((size_t)n_big * 4 + 64);
      if (meth_dev) {
        // every family: the wavefront-sized build of the record kernel first, the workgroup-sized one for the families above 64 records;
        // what neither takes is on the deferred list (the general path knows the mode)
        const uint32_t n_large = deep_pass(d_big.as<uint32_t>(), n_big, 0, d_deep_out.as<uint32_t>());
        if (n_large) {
          d_deep_out2.reserve((size_t)n_large * 4 + 64);
          (void)deep_pass(d_deep_out.as<uint32_t>(), n_large, 2, d_deep_out2.as<uint32_t>());
        }
        last_deep_families = n_big;
        last_meth_device = n_grp; n_big = 0;
      } else {
        // (round 6) two wavefronts per family first: a family of 65 .. 128 records — every deep family of a 2 .. 50-pair long tail — kept a quarter to a
        // half of the 256 threads of the large build busy in the record kernel; what has more records (or is not the kernels' shape) goes on to the large build
        const uint32_t n_mid = deep_pass(d_big.as<uint32_t>(), n_big, 1, d_deep_out.as<uint32_t>());
        uint32_t n_left = 0;
        big_list = d_deep_out.as<uint32_t>();
        if (n_mid) {
          d_deep_out2.reserve((size_t)n_mid * 4 + 64);
          n_left = deep_pass(d_deep_out.as<uint32_t>(), n_mid, 2, d_deep_out2.as<uint32_t>());
          big_list = d_deep_out2.as<uint32_t>();
        }
        last_deep_families = n_big - n_left;
        n_big = n_left;
      }
    }
    // more than 64 records (the list the first kernels filled), or more bytes than the largest slice (what the chain left): one workgroup per family
    for (int pass = 0; pass < 2 && !duplex && !codec; pass++) {
      const uint32_t cnt = pass == 0 ? n_cur : n_big;
      if (!cnt) continue;
      FastParams P2 = P;
      P2.group_list = pass == 0 ? cur_list : big_list; P2.retry = nullptr; P2.n_retry = nullptr;
      if (const char* e = fgx_knob("FGX_LDS_TILE_LARGE")) { uint32_t v = (uint32_t)atoi(e); if (v >= 16384 && v <= 163840) lds_tile_bytes_large = v & ~15u; P2.lds_tile_bytes = lds_tile_bytes_large; }   // tuning knob
      hip_check(hipFuncSetAttribute((const void*)k_family, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile_bytes_large), "hipFuncSetAttribute(MaxDynamicSharedMemorySize) for k_family: the device refused the dynamic LDS size");
      do { last_launches++; hipLaunchKernelGGL(k_family, dim3(cnt), dim3(NT), lds_tile_bytes_large, s, P2); } while (0);
      hip_check(hipGetLastError(), "k_family (large) launch");
    }
  }
  uint64_t n_full = 0;
  {   // dense call_full pass over the compacted lists
    std::vector<uint32_t> counts(N_LISTS);
    hip_check(hipMemcpyAsync(counts.data(), d_full_count.p, (size_t)N_LISTS * 4, hipMemcpyDeviceToHost, s), "D2H");
    FGX_SYNC(s);
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
    for (uint32_t v : counts) { uint32_t vv = v < full_cap ? v : full_cap; mx = vv > mx ? vv : mx; mn = v < mn ? v : mn; n_full += vv; }
    if (mn >= full_cap && pool_div > 1) {
      // every list is full: some family found no room (and was deferred).  Twice the room, if the device has it, and the batch again.
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const uint64_t want = (col_cap / (pool_div / 2) + (uint64_t)N_LISTS * pool_slack) * sizeof(FullItem);
      if (want < (uint64_t)free_b + (uint64_t)d_full_items.cap) {
        pool_div /= 2;
        static const bool verbose = [] { const char* e = fgx_knob("FGX_S2_VERBOSE"); return e && e[0] == '1'; }();
        if (verbose) fprintf(stderr, "[fgx] call_full pool exhausted: the batch again with 1/%u of the column bound\n", pool_div);
        return RUN_AGAIN_LARGER_POOL;
      }
    }
    if (mx) {
      FullParams F;
      memset(&F, 0, sizeof(F));
      F.items = d_full_items.as<FullItem>(); F.count = d_full_count.as<uint32_t>(); F.cap = full_cap;
      F.T = P.T; F.TU = P.TU; F.min_reads = P.min_reads; F.min_cons_bq = P.min_cons_bq;
      F.col_code = P.col_code; F.col_qual = P.col_qual; F.col_err = P.col_err;
      F.col_depth = P.col_depth; F.min_input_bq = P.min_input_bq;
      F.out = P.out; F.slot_desc = P.slot_desc; F.slot_err = P.slot_err; F.rg_len = P.rg_len; F.per_base_tags = P.per_base_tags;
      if (duplex) { F.rx_base = (char*)P.dends + offsetof(DuplexDesc, rx); F.rx_stride = sizeof(DuplexDesc); }
      else if (codec) { F.rx_base = (char*)P.cends + offsetof(CodecDesc, rx); F.rx_stride = sizeof(CodecDesc); }
      else { F.rx_base = (char*)P.ends + offsetof(EndDesc, rx); F.rx_stride = sizeof(EndDesc); }
      do { last_launches++; hipLaunchKernelGGL(k_call_full, dim3((mx + 255) / 256, N_LISTS), dim3(256), 0, s, F); } while (0);
      hip_check(hipGetLastError(), "k_call_full launch");
    }
  }
  if (meth_dev) {   // the methylation tags' share of the record sizes: the consensus bases are final now
    do { last_launches++; hipLaunchKernelGGL(k_meth_sizes, dim3((n_slots + 3) / 4), dim3(256), 0, s, P, n_slots, d_mslot.as<MethSlot>()); } while (0);
    hip_check(hipGetLastError(), "k_meth_sizes launch");
  }
  // ---- direct records: cE of the records whose columns had errors; did every prediction hold; is there anything to merge? ----------
  bool dir_pure = false;
  uint32_t dir_routed = 0;
  uint64_t dir_total = 0;
  if (direct) {
    do { last_launches++; hipLaunchKernelGGL(k_fix_ce, dim3((n_slots + 255) / 256), dim3(256), 0, s, d_slot_desc.as<SlotDesc>(), d_slot_err.as<uint32_t>(), n_slots, d_out.as<uint8_t>(), P.rg_len); } while (0);
    hip_check(hipGetLastError(), "k_fix_ce launch");
    unsigned long long h_flags = 0, h_def = 0, h_route = 0;
    hip_check(hipMemcpyAsync(&h_flags, misc + 36, 8, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipMemcpyAsync(&h_def, misc + 29, 8, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipMemcpyAsync(&h_route, misc + 33, 8, hipMemcpyDeviceToHost, s), "D2H");
    hip_check(hipMemcpyAsync(&dir_total, d_dir_base.as<uint64_t>() + dir_chunks_run, 8, hipMemcpyDeviceToHost, s), "D2H");
    FGX_SYNC(s);
    static const bool dir_verbose = [] { const char* e = fgx_knob("FGX_S2_VERBOSE"); return e && e[0] == '1'; }();
    if ((uint32_t)h_flags != 0) {          // a family's records are not what k_split_parse predicted: nothing of the batch can be trusted to be in place
      fprintf(stderr, "[fgx] direct records: %u families differ in size from the prediction; this caller goes back to the column scratch (please report)\n", (uint32_t)h_flags);
      direct_off = true;
      return RUN_AGAIN_LARGER_POOL;
    }
    if ((uint32_t)(h_flags >> 32) != 0) {  // more record bytes than the first estimate of the room: the exact amount is known now
      dir_cap_min = dir_total + dir_total / 16 + 4096;
      if (dir_verbose) fprintf(stderr, "[fgx] direct records: the batch needs %llu bytes of output, %llu were at hand: again\n", (unsigned long long)dir_total, (unsigned long long)dir_cap);
      return RUN_AGAIN_LARGER_POOL;
    }
    dir_routed = (uint32_t)h_route;
    dir_pure = dir_routed == 0 && (uint32_t)h_def == 0;
    last_direct = dir_pure ? 1 : 2;
    if (dir_verbose) fprintf(stderr, "[fgx] direct records: %llu bytes in place, %u families left the split pipeline, %u deferred\n", (unsigned long long)dir_total, dir_routed, (uint32_t)h_def);
  }
  hip_check(hipEventRecord(ev[1], s), "event");

  uint64_t out_len = dir_total;
  uint8_t* out_ptr = d_out.as<uint8_t>();
  if (!dir_pure) {
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_sizes.as<uint64_t>(), d_offsets.as<uint64_t>(), (int)n_slots + 1, s);
    d_scan_tmp.reserve(tmp_bytes);
    hip_check(hipcub::DeviceScan::ExclusiveSum(d_scan_tmp.p, tmp_bytes, d_sizes.as<uint64_t>(), d_offsets.as<uint64_t>(), (int)n_slots + 1, s), "scan");

    // total output size = last offset + last size
    // total output size = the scan's element behind the last slot (d_sizes[n_slots] is 0)
    uint64_t out_total = 0;
    hip_check(hipMemcpyAsync(&out_total, d_offsets.as<uint64_t>() + n_slots, 8, hipMemcpyDeviceToHost, s), "D2H");
    FGX_SYNC(s);
    out_len = out_total;
    if (out_len > (1ull << 40)) throw std::runtime_error("device pipeline: the scan of the record sizes gives " + std::to_string(out_len) + " bytes of output (a record size is corrupt)");
    if (direct) { d_out2.reserve(out_len + 16); out_ptr = d_out2.as<uint8_t>(); }   // the merged stream: d_out holds the directly written records
    else { d_out.reserve(out_len + 16); out_ptr = d_out.as<uint8_t>(); }
  }

  EmitParams E;
  memset(&E, 0, sizeof(E));
  E.blob = d_blob; E.rec_off = d_rec_off; E.ends = d_ends.as<EndDesc>(); E.out_off = d_offsets.as<uint64_t>(); E.out = out_ptr;
  E.out_base = 0; E.slot0 = 0; E.slot_end = n_slots;
  E.col_code = P.col_code; E.col_qual = P.col_qual; E.col_depth = P.col_depth; E.col_err = P.col_err;
  E.prefix = d_strings.as<char>(); E.prefix_len = P.prefix_len; E.rg = d_strings.as<char>() + P.prefix_len; E.rg_len = P.rg_len;
  E.per_base_tags = P.per_base_tags; E.tag0 = P.tag0; E.tag1 = P.tag1; E.cell0 = P.cell0; E.cell1 = P.cell1;
  hip_check(hipEventRecord(ev[2], s), "event");
  // (round 6) duplex / CODEC: the per-field record writer (k_emit_duplex / k_emit_codec: any length, any tag size) used to run over EVERY slot behind the fast
  // writer and find nothing to do (0.97 / 0.33 ms per step of wavefronts that load a descriptor and leave); a counting kernel (a thread per slot) now says
  // how many records the fast writer refuses, and the per-field kernel is launched — after the batch's last synchronisation — only when there are any
  DuplexEmitParams DE_late;
  CodecEmitParams CE_late;
  int late_emit = 0;
  if (codec) {
    CodecEmitParams CE;
    memset(&CE, 0, sizeof(CE));
    CE.blob = d_blob; CE.rec_off = d_rec_off; CE.ends = d_ends.as<CodecDesc>(); CE.out_off = d_offsets.as<uint64_t>(); CE.out = d_out.as<uint8_t>();
    CE.slot0 = 0; CE.slot_end = n_slots;
    CE.col_code = P.col_code; CE.col_qual = P.col_qual; CE.col_depth = P.col_depth; CE.col_err = P.col_err;
    CE.prefix = E.prefix; CE.prefix_len = E.prefix_len; CE.rg = E.rg; CE.rg_len = E.rg_len;
    CE.per_base_tags = P.per_base_tags; CE.cell0 = P.cell0; CE.cell1 = P.cell1;
    CE.has_outer = o.codec_has_outer_bases_qual; CE.outer_qual = o.codec_outer_bases_qual; CE.outer_len = o.codec_outer_bases_length;
    CE.has_ss = o.codec_has_single_strand_qual; CE.ss_qual = o.codec_single_strand_qual;
    CE.stats = d_statslots.as<unsigned long long>();
    CE.n_slow = (uint32_t*)(misc + 44);
    CE_late = CE; late_emit = 2;
    do { last_launches++; hipLaunchKernelGGL(k_count_slow_codec, dim3((n_slots + 255) / 256), dim3(256), 0, s, CE.ends, 0u, n_slots, CE.prefix_len, CE.rg_len, CE.n_slow); } while (0);
    do { last_launches++; hipLaunchKernelGGL(k_emit_codec_fast, dim3((n_slots + 3) / 4), dim3(256), 0, s, CE); } while (0);
  } else if (duplex) {
    DuplexEmitParams DE;
    memset(&DE, 0, sizeof(DE));
    DE.blob = d_blob; DE.rec_off = d_rec_off; DE.ends = d_ends.as<DuplexDesc>(); DE.out_off = d_offsets.as<uint64_t>(); DE.out = d_out.as<uint8_t>();
    DE.slot0 = 0; DE.slot_end = n_slots;
    DE.col_code = P.col_code; DE.col_qual = P.col_qual; DE.col_err = P.col_err; DE.col_obs = P.col_obs;
    DE.prefix = E.prefix; DE.prefix_len = E.prefix_len; DE.rg = E.rg; DE.rg_len = E.rg_len;
    DE.per_base_tags = P.per_base_tags; DE.cell0 = P.cell0; DE.cell1 = P.cell1;
    DE.n_slow = (uint32_t*)(misc + 44);
    DE_late = DE; late_emit = 1;
    do { last_launches++; hipLaunchKernelGGL(k_count_slow_duplex, dim3((n_slots + 255) / 256), dim3(256), 0, s, DE.ends, 0u, n_slots, DE.prefix_len, DE.rg_len, DE.n_slow); } while (0);
    do { last_launches++; hipLaunchKernelGGL(k_emit_duplex_fast, dim3((n_slots + 3) / 4), dim3(256), 0, s, DE); } while (0);
  } else if (direct) {
    if (!dir_pure) {
      // the merge: the families that left the split pipeline are written by k_emit from their descriptors, the directly written ones move
      // to their place in the final stream
      if (dir_routed) {
        E.fam_list = d_route.as<uint32_t>(); E.n_fam = dir_routed;
        do { last_launches++; hipLaunchKernelGGL(k_emit, dim3((dir_routed + 3) / 4), dim3(256), 0, s, E); } while (0);
      }
      do { last_launches++; hipLaunchKernelGGL(k_dir_copy, dim3((n_grp + 3) / 4), dim3(256), 0, s, d_split_out.as<SplitOut>(), d_dir_off.as<uint64_t>(), d_offsets.as<uint64_t>(),
                         d_sizes.as<uint64_t>(), d_out.as<uint8_t>(), out_ptr, n_grp); } while (0);
    }
  } else do { last_launches++; hipLaunchKernelGGL(k_emit, dim3((n_grp + 3) / 4), dim3(256), 0, s, E); } while (0);   // one wavefront per family (slots 3g .. 3g + 2)
  hip_check(hipGetLastError(), "k_emit launch");
  if (meth_dev) {
    do { last_launches++; hipLaunchKernelGGL(k_meth_tail, dim3((n_slots + 3) / 4), dim3(256), 0, s, P, n_slots, d_mslot.as<MethSlot>(), d_offsets.as<uint64_t>(), out_ptr); } while (0);
    hip_check(hipGetLastError(), "k_meth_tail launch");
  }
  hip_check(hipEventRecord(ev[3], s), "event");
  hip_check(hipEventRecord(c->ev1, s), "event");
  do { last_launches++; hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, s, d_statslots.as<unsigned long long>(), misc); } while (0);
  unsigned long long h_misc[46];
  hip_check(hipMemcpyAsync(h_misc, misc, sizeof(h_misc), hipMemcpyDeviceToHost, s), "D2H");
  FGX_SYNC(s);
  if (late_emit && (uint32_t)h_misc[44] != 0) {     // records the fast writer left: the per-field kernel, then the counters again (the CODEC writer counts bases)
    if (late_emit == 2) do { last_launches++; hipLaunchKernelGGL(k_emit_codec, dim3((n_slots + 3) / 4), dim3(256), 0, s, CE_late); } while (0);
    else do { last_launches++; hipLaunchKernelGGL(k_emit_duplex, dim3((n_slots + 3) / 4), dim3(256), 0, s, DE_late); } while (0);
    hip_check(hipGetLastError(), "per-field record writer launch");
    do { last_launches++; hipLaunchKernelGGL(k_reduce_stats, dim3(1), dim3(64), 0, s, d_statslots.as<unsigned long long>(), misc); } while (0);
    hip_check(hipMemcpyAsync(h_misc, misc, sizeof(h_misc), hipMemcpyDeviceToHost, s), "D2H");
    FGX_SYNC(s);
  }
  float ms = 0;
  hip_check(hipEventElapsedTime(&ms, c->ev0, c->ev1), "elapsed");

  res->d_out = out_ptr;
  res->out_len = out_len;
  res->count = h_misc[1];   // every consensus read of a fast-path family is one record
  for (int i = 0; i < FGX_STATS_LEN; i++) res->stats[i] = h_misc[i];
  res->n_deferred = (uint32_t)(h_misc[29] & 0xFFFFFFFFull);
  res->d_deferred = d_deferred.as<uint32_t>();
  res->d_out_off = d_offsets.as<uint64_t>();
  res->d_slot_size = d_sizes.as<uint64_t>();
  res->n_slots = n_slots;
  res->ms_kernels = ms;
  float msf = 0, mse = 0;
  hip_check(hipEventElapsedTime(&msf, ev[0], ev[1]), "elapsed");
  hip_check(hipEventElapsedTime(&mse, ev[2], ev[3]), "elapsed");
  res->ms_k_family = msf; res->ms_k_emit = mse;
  last_packed_families = h_misc[40]; last_classic_families = h_misc[41];
  res->cols_used = h_misc[28];
  res->full_items = n_full;
  return 0;
}

}  // namespace fgx

#if FGX_PHASE_TIMING
extern "C" int fgx_debug_phase_cycles(unsigned long long* out16, int reset) {
  unsigned long long h[64 * 16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fgx::g_phase), sizeof(h)) != hipSuccess) return 1;
  for (int i = 0; i < 16; i++) { out16[i] = 0; for (int b = 0; b < 64; b++) out16[i] += h[b * 16 + i]; }
  if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(fgx::g_phase), h, sizeof(h)) != hipSuccess) return 1; }
  return 0;
}
#endif
