// fastpath.h — declarations for the device-resident simplex pipeline (fastpath.hip).
#pragma once
#include "engine.h"

namespace fgx {

constexpr int FAST_RX_CAP = 48;   // longest RX value handled on the device (longer → general path)

struct EndDesc {            // one consensus read (a family end), written by the family kernels, consumed by k_emit
  uint64_t col_off;         // first column in the per-position arrays
  uint64_t first_off;       // blob offset (record body) of the first record of the MI group: MI value → read name + MI tag
  uint64_t kept_off;        // blob offset of the first retained source read: cell-barcode tag
                            // (offsets, not record indices: k_emit goes from the descriptor straight to the bytes, one dependent
                            // memory round trip less per record)
  uint32_t cons_len;
  uint32_t rec_size;        // BAM block_size of the record to emit
  uint16_t mi_off, cb_off;
  uint8_t mi_len, cb_len, rx_len, type;   // type: 0 fragment, 1 R1, 2 R2
  uint8_t has_cb, has_rx, valid, meth;   // meth (methylation-aware mode): bit 0 the call was annotated (cu / ct, MM / ML follow RX), bit 1 the MM / ML strand is the top strand
  char rx[FAST_RX_CAP];
};
// methylation-aware mode: what k_meth_sizes found for a record — the bytes of the standard record (the methylation tags follow them), ML entries, characters of the MM entries
struct MethSlot { uint32_t std_size, n_ml, mm_len, _pad; };

struct DuplexDesc {         // one duplex consensus record (slot 3g+1 = R1, 3g+2 = R2), written by k_family_wave<1>, consumed by k_emit_duplex
  uint64_t a_off, b_off;    // first column of the AB-side strand and of the BA-side strand (b = a when a single strand is passed through)
  uint32_t len;             // record length: the paired span of both strands, or the whole lone strand
  uint32_t first_rec;       // first paired record of the molecule (MI value → read name + MI tag, minus the /A|/B suffix)
  uint32_t cb_rec;          // first /A (else /B) record: cell-barcode source
  uint32_t rec_size;
  uint16_t mi_off, cb_off;
  uint8_t mi_len, cb_len, rx_len, type;
  uint8_t has_cb, has_rx, valid, has_ba;
  char rx[FAST_RX_CAP];
};

struct CodecDesc {          // one CODEC consensus record (slot 3g+1), written by k_family_wave<2>, consumed by k_emit_codec
  uint64_t s1_off, s2_off;  // R1-strand and R2-strand single-strand column segments (read orientation)
  uint32_t l1, l2, cons_len;
  uint32_t first_rec, cb_rec, rec_size;
  uint16_t mi_off, cb_off;
  uint8_t mi_len, cb_len, rx_len, flags;   // flags: bit 0 = longest R1 on the reverse strand, bit 1 = longest R2 on the reverse strand
  uint8_t has_cb, has_rx, valid, _pad;
  char rx[FAST_RX_CAP];
};

struct FullItem {           // a column (or UMI character) whose call needs the full log-sum-exp chain
  uint64_t dest;            // bit 63 clear: scratch column index; set: (slot << 8 | char index) of an RX character
  double ll[4];             // chains == 0: log-likelihoods of A, C, G, T; else: Kahan chains 1, 2, 3 and R as the kernel holds them
  uint32_t obs;             // observation counts, one byte each, in the order of ll
  uint32_t chains;          // 0, or the one-hot BAM codes of chains 1 | 2 << 4 | 3 << 8 (0 = chain not opened); every other base reads chain R
};
constexpr int N_LISTS = 1024;
// FullItem.chains with this bit set: the column did not stay with one base — `ll` holds its observations instead of chains,
// 16-bit each (4-bit code in consensus orientation << 8 | quality), chains & 0xFF of them in file order (at most 16);
// k_call_full accumulates them with the four-lane Kahan loop (base_builder.rs:836-868) and calls the column
constexpr uint32_t FULL_ITEM_OBS = 0x80000000u;
// A column of more than 16 observations (an end of up to 64 reads) takes ceil(m / 16) CONSECUTIVE items of one list: the head item as above
// with m = chains & 0xFF, then continuation items — chains == FULL_ITEM_CONT, observations 16 .. in `ll` again — which k_call_full's own
// threads skip.  (A family's items stay in one list when it holds such a column: k_split_cols pads a list's tail with continuation items.)
constexpr uint32_t FULL_ITEM_CONT = 0x40000000u;

// ---- split simplex pipeline (simplex_split.inc): k_split_parse leaves one SplitRec per record and one SplitFam per family;
// k_split_cols (one wavefront per family) reads them instead of parsing, pairing and walking tags itself --------------------------
struct SplitRec {           // 32 bytes
  uint32_t body_off;        // record body offset minus the body offset of the family's first record
  uint16_t seq_rel;         // SEQ offset inside the body (qualities follow at + (l_seq + 1) / 2)
  uint16_t l_seq;
  uint16_t clip;            // bases past the mate's start (raw-bam/overlap.rs:181-357), <= l_seq
  uint16_t ov_off1, ov_off2, ov_cnt;   // an R1 with a mate: first shared base in this read / in the mate, number of shared bases (0: none)
  uint8_t slot, mate_slot;  // row of this read / of its mate in the family's LDS tile (rows of end A first, then end B, file order inside an end)
  uint8_t bits;             // end type (0 fragment, 1 R1, 2 R2) | reverse strand << 2 | has the cell tag << 3
  uint8_t cb_len;
  uint16_t cb_rel, mi_rel;  // tag values: offsets inside the body
  uint8_t mi_len, rx_len;
  uint16_t rx_rel;
  uint32_t _pad;
};
struct SplitFam {           // 32 bytes
  uint32_t flags;           // bit 0: k_split_cols takes the family (else it goes to k_simplex_wave2 untouched)
  uint32_t need_bytes;      // LDS bytes of the family's tile: rows x (qs + ss)
  uint16_t qs, ss;          // row strides of the quality / sequence tiles (multiples of 16)
  uint8_t n, m_a, m_b, type_a;   // records; rows of end A / end B; type of end A (0 fragment, 1 R1) — end B is R2
  uint8_t rev_a, rev_b;     // strand of the rows of end A / end B (a regular end has one)
  uint8_t rx_len, rx_all;   // rx_all 1: every record carries the same RX value of rx_len bytes; 0: no record has one
  uint16_t len_a, len_b;    // longest read of end A / end B (reverse ends: THE read length)
  uint32_t inv_cpr;         // ceil(2^24 / chunks per row): row of flat chunk t = (t * inv_cpr) >> 24
  uint16_t qc, sc;          // 16-byte chunks per row: qualities, sequence
};
struct SplitOut {           // 32 bytes: what k_split_cols decided for a family; k_split_finish (a thread per family) turns it into the
                            // EndDescs, record sizes and counters
  uint8_t status;           // 0: not finished by k_split_cols (another kernel's family); 1: finished; 2: fewer records than --min-reads
  uint8_t ne;               // consensus reads: 0, 1 (fragment) or 2 (R1 + R2)
  uint8_t type_a;           // end A: 0 fragment, 1 R1
  uint8_t fk_a, fk_b;       // first retained record of end A / end B (index inside the family): cell-barcode source
  uint8_t kept_a, kept_b;   // retained reads per end (the UMI rule: one read = its bytes, several = normalised)
  uint8_t n;                // records of the family
  uint16_t lc_a, lc_b;      // consensus lengths
  uint32_t rej;             // rejected reads: insufficient | zero length << 8 | orphan << 16
  uint32_t ov_agree, ov_dis, ov_corr;   // overlapping-bases counters
  uint32_t depth_stats;     // direct records: max | min << 8 of the per-base depths of end A, the same of end B << 16 (the cD / cM tags; at most 16 reads per end)
};
static_assert(sizeof(SplitRec) == 32 && sizeof(SplitFam) == 32 && sizeof(SplitOut) == 32, "split descriptors are moved as two 16-byte pieces");

// ---- direct records (round 4): k_split_cols writes the consensus records of its families itself ------------------------------------
// k_split_parse predicts the exact bytes of every family's records (final lengths at the reads' 3' ends, consensus lengths, tag
// lengths); a per-chunk exclusive scan turns them into output offsets before the chunk's column kernel starts; the column lanes store
// nibbles / qualities / cd / ce straight into the records and the family's wavefront writes header, name and tags.  The columns that
// wait for k_call_full are patched in the records by that kernel (nibble OR, quality byte, cd / ce entries, the record's error count),
// and k_fix_ce rewrites the cE value of the records that had errors.  No column scratch, no EndDesc, no k_emit for these families.
struct SlotDesc {           // one directly written record (slot 3g + type): what k_call_full / k_fix_ce need to patch it
  uint64_t out_off;         // offset of the record (its block_size prefix) in the output buffer
  uint32_t lc_seq;          // consensus length | offset of SEQ inside the record (from the prefix) << 16
  uint32_t sum_depth;       // sum of the per-base depths: the denominator of cE
};
constexpr uint64_t FULL_DEST_DIRECT = 1ull << 62;   // FullItem.dest: DIRECT | slot << 16 | column

struct FastParams {
  const uint8_t* blob; const uint64_t* rec_off; const uint32_t* rec_len; const uint32_t* grp_first;
  uint32_t g0;
  const DeviceTables* T; const DeviceTables* TU;
  uint32_t min_reads; int64_t max_reads;
  char tag0, tag1, cell0, cell1;   // (dword-aligned: the kernels read them with scalar loads; at odd offsets they came through vector memory)
  uint32_t min_input_bq, min_cons_bq, trim, overlap, per_base_tags, track_rejects;   // (one dword each: adjacent bytes were fetched as one unaligned vector load)
  uint32_t prefix_len, rg_len;
  EndDesc* ends; uint64_t* rec_sizes;
  uint8_t* col_code; uint8_t* col_qual; uint16_t* col_depth; uint16_t* col_err;
  const uint64_t* col_base;        // per-group first scratch column (exclusive scan of the column bound)
  unsigned long long* stats;
  uint32_t* deferred; uint32_t* n_deferred;
  const uint32_t* group_list;      // nullptr: group = g0 + blockIdx.x
  uint32_t* retry; uint32_t* n_retry;   // families needing more LDS than this launch provides (nullptr: defer them)
  uint32_t* retry_old; uint32_t* n_retry_old;   // k_simplex_wave2: families outside its record shape → k_family_wave<0>
  const void* w2_image;            // k_simplex_wave2: image of its LDS tables (W2Lds, simplex_wave2.inc)
  const void* fw_image;            // k_family_wave: image of its LDS tables (FwLds, fastpath.hip)
  SplitRec* split_rec; SplitFam* split_fam; SplitOut* split_out;   // split simplex pipeline: k_split_parse → k_split_cols → k_split_finish
  uint32_t* route; uint32_t* n_route;         // k_split_cols: families it does not take → k_simplex_wave2
  uint32_t* big; uint32_t* n_big;             // families of more than 64 records (no wavefront-per-family kernel takes them): straight to k_family's list (may be null)
  const void* s2_image;            // k_split_cols: image of its LDS tables (S2Lds, simplex_split.inc)
  const uint4* fam_desc;           // per family {first record offset lo, hi, bytes to the end of the last record (~0: none), records} (k_col_bound)
  uint64_t blob_len;               // records must end inside the blob (checked before the family's bytes are staged)
  // direct records (k_split_parse predicts, k_split_cols<.., .., 1> writes)
  uint32_t* dir_size;              // per family: bytes of its consensus records (block_size prefixes included); 0: none / not this pipeline's
  const uint64_t* dir_off;         // per family: offset of its first record in `out` (exclusive scan of dir_size)
  uint8_t* out; uint64_t out_cap;  // the output buffer
  uint64_t* out_off;               // per slot (3g + type): offset of the record in `out` (what the scan of rec_sizes gives the scratch path)
  SlotDesc* slot_desc; uint32_t* slot_err;   // per slot: patch descriptor; errors counted by k_call_full
  const char* strings;             // read-name prefix | read group id
  uint32_t* dir_flags;             // [0] families whose records differ in size from the prediction, [1] records past out_cap
  uint32_t lds_tile_bytes;
  uint32_t lds_wave_bytes;
  uint32_t s2_partner;             // k_split_cols<.., 1>: a <.., 2> launch over the same families follows (else it hands the families that are not its shape to the next launch)
  uint32_t s2_packed;              // k_split_cols: 1 = ends of at least s2_nsafe rows try the packed pass first (round 5; FGX_S2_PACKED=0: measurements)
  uint32_t s2_nsafe;               // k_split_cols: unanimous_cap_depth(tables, min_input_bq) — agreeing observations from which a column is the cap for certain (gate_core.h)
  FullItem* full_items; uint32_t* full_count; uint32_t full_cap;   // N_LISTS append lists of `full_cap` items each
  // methylation-aware mode on the streaming kernels (simplex_deep.inc; meth_mode = FGX_METHYLATION_*, 0: off): the genome of fgx_set_reference
  // (contig i = [contig_off[i], + contig_len[i])), and per scratch column: the reference shows a cytosine of the call's strand | unconverted |
  // converted source bases (methylation.rs:193-242)
  uint32_t meth_mode, n_ref;
  const uint8_t* genome; const uint64_t* contig_off; const uint64_t* contig_len;
  uint8_t* meth_flag; uint16_t* meth_u; uint16_t* meth_t;
  // duplex (k_family_wave<1>)
  uint32_t dmin_total, dmin_xy, dmin_yx; int64_t dmax_reads;
  uint32_t* col_obs;               // per column: observation counts of A,C,G,T, one byte each
  DuplexDesc* dends;
  // CODEC (k_family_wave<2>)
  CodecDesc* cends; uint32_t cmin_reads, cmin_duplex_len; int64_t cmax_reads;
};

struct EmitParams {
  const uint8_t* blob; const uint64_t* rec_off; const EndDesc* ends; const uint64_t* out_off; uint8_t* out;
  uint64_t out_base; uint32_t slot0, slot_end;
  const uint8_t* col_code; const uint8_t* col_qual; const uint16_t* col_depth; const uint16_t* col_err;
  const char* prefix; uint32_t prefix_len; const char* rg; uint32_t rg_len;
  uint8_t per_base_tags; char tag0, tag1, cell0, cell1;
  const uint32_t* fam_list; uint32_t n_fam;   // non-null: a wavefront per LISTED family (direct records: the families that left the split pipeline)
};

struct DuplexEmitParams {
  const uint8_t* blob; const uint64_t* rec_off; const DuplexDesc* ends; const uint64_t* out_off; uint8_t* out;
  uint32_t slot0, slot_end;
  const uint8_t* col_code; const uint8_t* col_qual; const uint16_t* col_err; const uint32_t* col_obs;
  const char* prefix; uint32_t prefix_len; const char* rg; uint32_t rg_len;
  uint8_t per_base_tags; char cell0, cell1;
  uint32_t* n_slow;                // k_count_slow_duplex counts the valid records the fast writer leaves to the per-field kernel here (0: that kernel is not launched at all)
};

struct CodecEmitParams {
  const uint8_t* blob; const uint64_t* rec_off; const CodecDesc* ends; const uint64_t* out_off; uint8_t* out;
  uint32_t slot0, slot_end;
  const uint8_t* col_code; const uint8_t* col_qual; const uint16_t* col_depth; const uint16_t* col_err;
  const char* prefix; uint32_t prefix_len; const char* rg; uint32_t rg_len;
  uint8_t per_base_tags; char cell0, cell1;
  uint8_t has_outer, outer_qual, has_ss, ss_qual; uint32_t outer_len;
  unsigned long long* stats;       // slot-spread counters: [24] consensus bases, [25] duplex bases, [26] disagreeing duplex bases
  uint32_t* n_slow;                // k_count_slow_codec counts the valid records the fast writer leaves to the per-field kernel here
};

struct FastResult {
  const uint8_t* d_out; uint64_t out_len; uint64_t count;
  uint64_t stats[FGX_STATS_LEN];
  uint32_t n_deferred; const uint32_t* d_deferred;
  const uint64_t* d_out_off;     // byte offset of each of the 3*n_grp slots in d_out
  const uint64_t* d_slot_size;   // bytes of each slot (block_size prefix included; 0 = no record)
  uint32_t n_slots;
  double ms_kernels, ms_k_family, ms_k_emit; uint64_t cols_used; uint64_t full_items;
};

struct FastPath {
  DevBuf d_ends, d_sizes, d_offsets, d_code, d_qual, d_depth, d_err, d_misc, d_deferred, d_out, d_scan_tmp, d_strings;
  uint32_t lds_tile_bytes = 12288;        // first launch: tiles of the common small families
  uint32_t lds_tile_bytes_large = 65536;  // workgroup-per-family kernel: raw records + unpacked base / quality tiles of one big family (2 workgroups per CU)
  DevBuf d_retry, d_bound, d_colbase, d_statslots, d_full_items, d_full_count, d_obs, d_retry2, d_retry_old;
  DevBuf d_w2img;                         // W2Lds image, built from the caller's tables at the first batch
  DevBuf d_famdesc;                       // k_col_bound's family descriptors
  DevBuf d_fwimg;                         // FwLds image (k_family_wave)
  DevBuf d_split_rec, d_split_fam, d_split_out, d_route, d_s2img;   // split simplex pipeline
  DevBuf d_big;                            // simplex families of more than 64 records: k_family's list, filled by the first kernel that sees them
  uint32_t last_big_families = 0;          // ... in the last batch
  uint32_t last_deep_families = 0;         // ... of which k_deep_parse + k_deep_cols (simplex_deep.inc) took
  DevBuf d_deep_sizes, d_deep_row0, d_deep_rows, d_deep_fams, d_deep_out, d_deep_out2;
  DevBuf d_mflag, d_mu, d_mt, d_mslot, d_mcontigs;   // methylation-aware mode: per-column annotation, per-slot tag sizes, the contig table
  uint32_t last_meth_device = 0;           // families of the last batch that the device pipeline decided in the methylation-aware mode
  uint32_t last_routed = 0;                // families the split pipeline handed to the k_simplex_wave2 chain in the last batch
  uint64_t last_packed_families = 0, last_classic_families = 0;   // families of the last batch finished by k_split_cols's packed build / by its classic builds (k_split_finish counts them)
  uint32_t last_split_build = 0;           // first-stage build of the last batch: 0 classic alone (or no split pipeline), 1 packed alone, 2 packed + partner launch
  uint32_t last_first_stage_retries = 0;   // families the first stage handed to the next launch
  uint32_t last_launches = 0, last_host_syncs = 0;   // kernel launches (library scans not counted) / host synchronisations of the last run_once
  DevBuf d_dir_size, d_dir_off, d_dir_base, d_slot_desc, d_slot_err, d_out2, d_scan_tmp2;   // direct records (simplex_split.inc, fastpath.h)
  bool direct_off = false;                // a batch whose predicted record sizes did not hold: this caller stays on the scratch path (diagnostics: last_direct)
  uint64_t dir_cap_min = 0;               // output room a batch asked for beyond the first estimate
  uint32_t dir_chunks_run = 0;            // chunks of the last direct batch that held families: d_dir_base[dir_chunks_run] = bytes of all directly written records
  int last_direct = 0;                    // 0: the last batch went through the column scratch + k_emit; 1: records written directly; 2: directly + merge
  uint32_t lds_wave_bytes = 6144;         // wave-per-family kernel: LDS copy of one family's raw records
  uint32_t lds_wave_bytes_duplex = 8704;  // duplex molecules carry both strands (config 3: 24 records x ~330 B)
  uint32_t lds_wave_bytes_codec = 5120;   // CODEC (config 5: 8 records x ~570 B): its kernel needs 71 VGPRs, so the smaller slice buys a sixth wave per SIMD (+3 %)
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // split simplex pipeline: the record kernel runs chunk by chunk on a second stream, under the column kernel of the chunk before
  uint32_t pool_slack = 256; bool pool_init = false;
  uint32_t pool_div = 8;                  // k_call_full's append lists hold 1 / pool_div of the column bound (halved when a batch exhausts them)
  static constexpr int RUN_AGAIN_LARGER_POOL = -77;   // (also: the batch again without / with more room for the direct records)
  int run_once(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
               const uint32_t* d_grp_first, uint32_t n_grp, FastResult* res);
  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: one flag per FastPath (= per caller = per device), not per process
  bool lds_attr_set = false, s2_attr_set = false, v2_attr_set = false;
  static constexpr int MAX_CHUNKS = 65;
  uint32_t last_split_chunks = 0;         // chunks the record / column pipeline ran the last batch in (0: another head of the chain); diagnostics
  hipStream_t s2 = nullptr;
  hipEvent_t ev_chunk[MAX_CHUNKS] = {}, ev_cols[MAX_CHUNKS] = {}, ev_fin = nullptr, ev_sample = nullptr;
  int run(fgx_caller* c, const uint8_t* d_blob, uint64_t blob_len, const uint64_t* d_rec_off, const uint32_t* d_rec_len, uint32_t n_rec,
          const uint32_t* d_grp_first, uint32_t n_grp, FastResult* res);
  void release();
};

}  // namespace fgx
