// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Plain C++17 restatement (host, scalar, real glibc libm) of the reference's Phred /
// log-probability helpers and of the per-column `ConsensusBaseBuilder`.
//
//   crates/fgumi-consensus/src/phred.rs:13-42, 72-400
//   crates/fgumi-consensus/src/base_builder.rs:289-327, 336-387, 403-527, 595-773, 775-1081
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code,
// and only as the checker.  Must be built with -O2 -ffp-contract=off -fno-fast-math so that
// no a*b+c is fused (the reference is compiled for baseline x86-64, no FMA contraction).
//
// Parity status: pinned against every known-answer vector the reference's own unit tests
// hold for this layer (replayed by tests/test_oracle_*.py).  Whole-BAM
// parity against the real Rust binary is unpinned here: no Rust toolchain in the image.
#pragma once
#include <cfloat>
#include <cmath>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace orc {

using std::size_t;

constexpr double LN_10 = 2.30258509299404568401799145468436421;
constexpr double LN_TWO = 0.693147180559945309417232121458176568;
constexpr double LOG1PEXP_ZERO = LN_TWO;                 // phred.rs:24
constexpr double LN_FOUR_THIRDS = 0.2876820724517809;    // phred.rs:27
constexpr double LN_ONE = 0.0;
constexpr uint8_t MIN_PHRED = 2;                         // fgumi-dna/src/lib.rs:24
constexpr uint8_t MAX_PHRED = 93;                        // phred.rs:36
constexpr uint8_t NO_CALL_BASE = 'N';
constexpr uint8_t NO_CALL_BASE_LOWER = 'n';
constexpr double PHRED_PRECISION = 0.001;                // phred.rs:39
static const double MAX_PHRED_AS_LN_ERROR = -(double)MAX_PHRED * LN_10 / 10.0;  // phred.rs:42
static const double NEG_INF = -std::numeric_limits<double>::infinity();
static const double POS_INF = std::numeric_limits<double>::infinity();

// phred.rs:73-75
inline double phred_to_ln_error_prob(uint8_t phred) { return -(double)phred * LN_10 / 10.0; }

// phred.rs:127-143
inline uint8_t ln_prob_to_phred(double ln_prob) {
  if (ln_prob < MAX_PHRED_AS_LN_ERROR) return MAX_PHRED;
  double phred = std::floor(-10.0 * ln_prob / LN_10 + PHRED_PRECISION);
  if (std::isnan(phred)) return 0;  // Rust: NaN.clamp(..) stays NaN, `as u8` saturates to 0
  if (phred < (double)MIN_PHRED) phred = (double)MIN_PHRED;
  if (phred > (double)MAX_PHRED) phred = (double)MAX_PHRED;
  return (uint8_t)phred;
}

// phred.rs:156-166
inline double log1pexp(double x) {
  if (x <= -37.0) return std::exp(x);
  if (x <= 18.0) return std::log1p(std::exp(x));
  if (x <= 33.3) return x + std::exp(-x);
  return x;
}

// phred.rs:176-189
inline double ln_one_minus_exp(double x) {
  if (x >= 0.0) return NEG_INF;
  if (x >= -LN_TWO) return std::log(-std::expm1(x));
  return std::log1p(-std::exp(x));
}

struct OracleError {
  const char* what;
};

// phred.rs:211-223
inline double ln_a_minus_b(double a, double b) {
  if (std::isinf(b) && b < 0.0) return a;
  if (std::fabs(a - b) < DBL_EPSILON) return NEG_INF;
  if (a < b) throw OracleError{"Subtraction will be less than zero."};
  return a + ln_one_minus_exp(b - a);
}

// phred.rs:308-335
inline double ln_sum_exp(double ln_a, double ln_b) {
  if (std::isinf(ln_a) && ln_a < 0.0) return ln_b;
  if (std::isinf(ln_b) && ln_b < 0.0) return ln_a;
  if (ln_a == ln_b) return ln_a + LOG1PEXP_ZERO;
  if (ln_b < ln_a) { double t = ln_a; ln_a = ln_b; ln_b = t; }
  return ln_a + log1pexp(ln_b - ln_a);
}

// phred.rs:256-275
inline double ln_error_prob_two_trials(double ln_p1, double ln_p2) {
  if (ln_p1 < ln_p2) { double t = ln_p1; ln_p1 = ln_p2; ln_p2 = t; }
  if (ln_p1 - ln_p2 >= 6.0) return ln_p1;
  double term1 = ln_sum_exp(ln_p1, ln_p2);
  double term2 = LN_FOUR_THIRDS + ln_p1 + ln_p2;
  return ln_a_minus_b(term1, term2);
}

// phred.rs:357-384
inline double ln_sum_exp_array(const double* values, size_t n) {
  bool all_neg_inf = true;
  for (size_t i = 0; i < n; i++) if (!(values[i] == NEG_INF)) all_neg_inf = false;
  if (n == 0 || all_neg_inf) return NEG_INF;
  double min_value = POS_INF;
  size_t min_index = 0;
  for (size_t i = 0; i < n; i++) {
    if (values[i] < min_value) { min_index = i; min_value = values[i]; }
  }
  double sum = min_value;
  for (size_t i = 0; i < n; i++) if (i != min_index) sum = ln_sum_exp(sum, values[i]);
  return sum;
}

inline double ln_normalize(double v, double norm) { return v - norm; }   // phred.rs:391
inline double ln_not(double x) { return ln_one_minus_exp(x); }            // phred.rs:398
inline double phred_to_ln_correct_prob(uint8_t p) { return ln_one_minus_exp(phred_to_ln_error_prob(p)); }

// ------------------------------------------------------------------------------------
// base_builder.rs
// ------------------------------------------------------------------------------------
constexpr double UNANIMOUS_MARGIN_HEADROOM = 16.0;                  // :312
constexpr double UNANIMOUS_UNIT_ROUNDOFF = DBL_EPSILON / 2.0;       // :319
constexpr double FGBIO_TIE_EPSILON = 1.0 / 4503599627370496.0;      // :408
constexpr uint32_t TIE_TOLERANCE_ULPS = 4;                          // :401
static const uint8_t DNA_BASES[4] = {'A', 'C', 'G', 'T'};

inline int base_to_index(uint8_t b) {  // BASE_TO_INDEX :323-334
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 255;
  }
}

// AdjustedProbabilityTables::compute :349-370
struct AdjustedTables {
  double correct[94];
  double error_per_alt[94];
  explicit AdjustedTables(uint8_t post) {
    double ln_error_post = phred_to_ln_error_prob(post);
    double ln_three = std::log(3.0);
    for (int q = 0; q <= MAX_PHRED; q++) {
      double ln_error_seq = phred_to_ln_error_prob((uint8_t)q);
      double adjusted = ln_error_prob_two_trials(ln_error_post, ln_error_seq);
      correct[q] = ln_not(adjusted);
      error_per_alt[q] = adjusted - ln_three;
    }
  }
};

// :595-601
inline uint8_t unanimous_quality_from_gap(double gap, double ln_error_pre_umi) {
  double v[4] = {0.0, -gap, -gap, -gap};
  double ln_sum = ln_sum_exp_array(v, 4);
  double ln_posterior = ln_normalize(0.0, ln_sum);
  double ln_consensus_error = ln_not(ln_posterior);
  double ln_final = ln_error_prob_two_trials(ln_error_pre_umi, ln_consensus_error);
  return ln_prob_to_phred(ln_final);
}

// :676-679
inline double consensus_error(double gap) {
  double e = 3.0 * std::exp(-gap);
  return e / (1.0 + e);
}

// :700-703
inline double unanimous_margin(double w, double l, double cerr_lower_bound) {
  return UNANIMOUS_MARGIN_HEADROOM * UNANIMOUS_UNIT_ROUNDOFF * (std::fabs(w) + std::fabs(l)) /
         cerr_lower_bound;
}

// build_unanimous_gap_thresholds :615-656, build_unanimous_gap_tables :743-754
struct GapTables {
  double thresholds[94];
  double cerr_min[94];
  size_t cap;
  explicit GapTables(uint8_t pre) {
    const double MAX_GAP = 256.0;
    const int ITER = 64;
    double ln_pre = phred_to_ln_error_prob(pre);
    uint8_t q0 = unanimous_quality_from_gap(0.0, ln_pre);
    uint8_t qmax = unanimous_quality_from_gap(MAX_GAP, ln_pre);
    for (int q = 0; q <= MAX_PHRED; q++) {
      thresholds[q] = POS_INF;
      if (q0 >= q) { thresholds[q] = 0.0; continue; }
      if (qmax < q) continue;
      double too_small = 0.0, wide_enough = MAX_GAP;
      for (int it = 0; it < ITER; it++) {
        double mid = 0.5 * (too_small + wide_enough);
        if (unanimous_quality_from_gap(mid, ln_pre) >= q) wide_enough = mid; else too_small = mid;
      }
      thresholds[q] = wide_enough;
    }
    cap = (size_t)ln_prob_to_phred(ln_pre);
    for (int q = 0; q <= MAX_PHRED; q++) cerr_min[q] = 0.0;
    for (size_t q = 0; q < cap && q < 94; q++) cerr_min[q] = consensus_error(thresholds[q + 1]);
  }
};

enum class TieRule { UlpRelative = 1, FgbioCompat = 0 };

// fgbio_unique_max_index :448-470 ; returns -1 for None
inline int fgbio_unique_max_index(const double* ll) {
  double max = -DBL_MAX;
  int max_index = -1;
  bool assigned = false;
  for (int i = 0; i < 4; i++) {
    double v = ll[i];
    if (std::isnan(v)) continue;
    if (!assigned || v > max) { max = v; max_index = i; assigned = true; }
    else if (std::fabs(v - max) <= FGBIO_TIE_EPSILON) max_index = -1;
  }
  if (!assigned || max_index < 0) return -1;
  return max_index;
}

// approx 0.5.1 `ulps_eq!(a, b, epsilon = 0.0, max_ulps = 4)` for f64
inline bool ulps_eq0(double a, double b, uint32_t max_ulps) {
  double diff = (a > b) ? (a - b) : (b - a);
  if (diff <= 0.0) return true;  // abs_diff_eq with epsilon 0 (false for NaN)
  auto signum = [](double x) { return std::isnan(x) ? NAN : (std::signbit(x) ? -1.0 : 1.0); };
  double sa = signum(a), sb = signum(b);
  if (sa != sb) return false;  // also NaN
  uint64_t ia, ib;
  std::memcpy(&ia, &a, 8);
  std::memcpy(&ib, &b, 8);
  return (ia <= ib) ? (ib - ia <= max_ulps) : (ia - ib <= max_ulps);
}

// unique_max_index :487-516
inline int unique_max_index(const double* ll) {
  double max = NEG_INF;
  int max_index = -1;
  for (int i = 0; i < 4; i++) if (ll[i] > max) { max = ll[i]; max_index = i; }
  if (max_index < 0 || !std::isfinite(max)) return -1;
  int tied = 0;
  for (int i = 0; i < 4; i++) if (ulps_eq0(ll[i], max, TIE_TOLERANCE_ULPS)) tied++;
  return tied == 1 ? max_index : -1;
}

// Process-wide table caches, one slot per Phred value, built once on first use — the reference keeps both table
// sets in `OnceLock` caches of 256 slots (`adjusted_tables`, base_builder.rs:384-387; `unanimous_gap_tables`, :756-773):
// a caller object per batch of 50 MI groups must not pay for the 64-step bisections again.
template <class T>
inline const T& cached_tables(uint8_t key) {
  static std::once_flag once[256];
  static T* slot[256];
  std::call_once(once[key], [key] { slot[key] = new T(key); });
  return *slot[key];
}
inline const AdjustedTables& adjusted_tables(uint8_t post) { return cached_tables<AdjustedTables>(post); }
inline const GapTables& unanimous_gap_tables(uint8_t pre) { return cached_tables<GapTables>(pre); }

struct ConsensusBaseBuilder {
  double likelihoods[4];
  double compensations[4];
  uint32_t observations[4];
  TieRule tie_rule = TieRule::FgbioCompat;
  const AdjustedTables& adj;
  const GapTables& gap;
  double ln_error_pre_umi;

  ConsensusBaseBuilder(uint8_t pre, uint8_t post) : adj(adjusted_tables(post)), gap(unanimous_gap_tables(pre)) {
    ln_error_pre_umi = phred_to_ln_error_prob(pre);
    reset();
  }
  void reset() {
    for (int i = 0; i < 4; i++) { likelihoods[i] = LN_ONE; compensations[i] = 0.0; observations[i] = 0; }
  }
  // :836-868
  void add(uint8_t base, uint8_t qual) {
    int idx = base_to_index(base);
    if (idx == 255) return;
    int q = qual < MAX_PHRED ? qual : MAX_PHRED;
    double ln_correct = adj.correct[q];
    double ln_err = adj.error_per_alt[q];
    for (int lane = 0; lane < 4; lane++) {
      double v = (lane == idx) ? ln_correct : ln_err;
      double y = v - compensations[lane];
      double t = likelihoods[lane] + y;
      compensations[lane] = (t - likelihoods[lane]) - y;
      likelihoods[lane] = t;
    }
    observations[idx] += 1;
  }
  uint32_t contributions() const { return observations[0] + observations[1] + observations[2] + observations[3]; }
  uint32_t observations_for_base(uint8_t base) const {
    int idx = base_to_index(base);
    return idx == 255 ? 0 : observations[idx];
  }
  // :883-994 ; returns true and sets (base, qual) when the fast path answers
  bool try_unanimous_fast_path(uint8_t& base, uint8_t& qual) const {
    int observed = -1, n_obs = 0;
    for (int i = 0; i < 4; i++) {
      if (observations[i] > 0) { n_obs++; observed = i; if (n_obs > 1) return false; }
    }
    if (observed < 0) return false;
    double w = likelihoods[observed];
    double l = likelihoods[(observed + 1) % 4];
    double g = w - l;
    if (!(std::isfinite(g) && g > DBL_EPSILON)) return false;
    size_t cap = gap.cap;
    double cap_threshold = gap.thresholds[cap];
    if (g >= cap_threshold) {
      double delta = unanimous_margin(w, l, 1.0);
      double cerr_at_cap = gap.cerr_min[cap - 1];
      if (g - cap_threshold >= LN_TWO && delta < 0.5 * cerr_at_cap) {
        base = DNA_BASES[observed]; qual = (uint8_t)cap; return true;
      }
      return false;
    }
    // partition_point(|t| t <= gap) - 1
    size_t lo = 0, hi = 94;
    while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (gap.thresholds[mid] <= g) lo = mid + 1; else hi = mid; }
    size_t q = lo - 1;
    double margin = unanimous_margin(w, l, gap.cerr_min[q]);
    if (g - gap.thresholds[q] > margin && gap.thresholds[q + 1] - g > margin) {
      base = DNA_BASES[observed]; qual = (uint8_t)q; return true;
    }
    return false;
  }
  // :1023-1054
  void call_full(uint8_t& base, uint8_t& qual) const {
    double ln_sum = ln_sum_exp_array(likelihoods, 4);
    int max_idx = (tie_rule == TieRule::FgbioCompat) ? fgbio_unique_max_index(likelihoods)
                                                     : unique_max_index(likelihoods);
    if (max_idx < 0) { base = NO_CALL_BASE; qual = MIN_PHRED; return; }
    double ln_post = ln_normalize(likelihoods[max_idx], ln_sum);
    double ln_cons_err = ln_not(ln_post);
    double ln_final = ln_error_prob_two_trials(ln_error_pre_umi, ln_cons_err);
    base = DNA_BASES[max_idx];
    qual = ln_prob_to_phred(ln_final);
  }
  // :1002-1014
  void call(uint8_t& base, uint8_t& qual) const {
    if (contributions() == 0) { base = NO_CALL_BASE; qual = MIN_PHRED; return; }
    if (try_unanimous_fast_path(base, qual)) return;
    call_full(base, qual);
  }
};

}  // namespace orc
