// consensus_math.h — the per-column arithmetic of the consensus caller, written once for
// MI355X device code and for the host-side table builders of the library.
//
// Semantics follow the reference exactly (bit-exact contract):
//   crates/fgumi-consensus/src/phred.rs:73-75, 127-143, 156-189, 211-223, 256-275, 308-384
//   crates/fgumi-consensus/src/base_builder.rs:349-370 (adjusted tables), 448-516 (tie rules),
//       595-601, 615-656, 676-679, 700-703, 743-754 (gap tables), 836-868 (add), 883-994 (fast
//       path), 1002-1054 (call / call_full)
// Transcendentals come from glibc_libm.h (bit-exact glibc 2.35), never from ocml.
// Compile with -ffp-contract=off.
#pragma once
#include "glibc_libm.h"

namespace fgx {

#define FGX_LN_10 2.30258509299404568401799145468436421
#define FGX_LN_2 0.693147180559945309417232121458176568
#define FGX_LN_FOUR_THIRDS 0.2876820724517809
#define FGX_DBL_EPSILON 2.220446049250313e-16
#define FGX_MAX_PHRED 93
#define FGX_MIN_PHRED 2

FGX_HD double m_inf() { return fgx_asdouble(0x7ff0000000000000ULL); }
FGX_HD double m_neg_inf() { return fgx_asdouble(0xfff0000000000000ULL); }
FGX_HD bool m_isnan(double x) { return x != x; }
FGX_HD bool m_isinf(double x) { return (fgx_asuint64(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
FGX_HD bool m_isfinite(double x) { return (fgx_asuint64(x) & 0x7ff0000000000000ULL) != 0x7ff0000000000000ULL; }
FGX_HD double m_fabs(double x) { return fgx_asdouble(fgx_asuint64(x) & 0x7fffffffffffffffULL); }
FGX_HD double m_floor(double x) { return __builtin_floor(x); }

FGX_HD double phred_to_ln_error_prob(uint8_t phred) { return -(double)phred * FGX_LN_10 / 10.0; }

FGX_HD uint8_t ln_prob_to_phred(double ln_prob) {
  const double max_as_ln = -(double)FGX_MAX_PHRED * FGX_LN_10 / 10.0;
  if (ln_prob < max_as_ln) return FGX_MAX_PHRED;
  double phred = m_floor(-10.0 * ln_prob / FGX_LN_10 + 0.001);
  if (m_isnan(phred)) return 0;
  if (phred < (double)FGX_MIN_PHRED) phred = (double)FGX_MIN_PHRED;
  if (phred > (double)FGX_MAX_PHRED) phred = (double)FGX_MAX_PHRED;
  return (uint8_t)phred;
}

FGX_HD double log1pexp(double x) {
  if (x <= -37.0) return g_exp(x);
  if (x <= 18.0) return g_log1p(g_exp(x));
  if (x <= 33.3) return x + g_exp(-x);
  return x;
}

FGX_HD double ln_one_minus_exp(double x) {
  if (x >= 0.0) return m_neg_inf();
  if (x >= -FGX_LN_2) return g_log(-g_expm1(x));
  return g_log1p(-g_exp(x));
}

// `ok` is cleared where the reference panics (a < b by >= EPSILON; unreachable in production).
FGX_HD double ln_a_minus_b(double a, double b, bool* ok) {
  if (m_isinf(b) && b < 0.0) return a;
  if (m_fabs(a - b) < FGX_DBL_EPSILON) return m_neg_inf();
  if (a < b) { if (ok) *ok = false; return fgx_asdouble(0x7ff8000000000000ULL); }
  return a + ln_one_minus_exp(b - a);
}

FGX_HD double ln_sum_exp(double ln_a, double ln_b) {
  if (m_isinf(ln_a) && ln_a < 0.0) return ln_b;
  if (m_isinf(ln_b) && ln_b < 0.0) return ln_a;
  if (ln_a == ln_b) return ln_a + FGX_LN_2;
  if (ln_b < ln_a) { double t = ln_a; ln_a = ln_b; ln_b = t; }
  return ln_a + log1pexp(ln_b - ln_a);
}

FGX_HD double ln_error_prob_two_trials(double ln_p1, double ln_p2, bool* ok) {
  if (ln_p1 < ln_p2) { double t = ln_p1; ln_p1 = ln_p2; ln_p2 = t; }
  if (ln_p1 - ln_p2 >= 6.0) return ln_p1;
  double term1 = ln_sum_exp(ln_p1, ln_p2);
  double term2 = FGX_LN_FOUR_THIRDS + ln_p1 + ln_p2;
  return ln_a_minus_b(term1, term2, ok);
}

FGX_HD double ln_sum_exp_array4(const double* v) {
  const double ninf = m_neg_inf();
  if (v[0] == ninf && v[1] == ninf && v[2] == ninf && v[3] == ninf) return ninf;
  double min_value = m_inf();
  int min_index = 0;
  for (int i = 0; i < 4; i++)
    if (v[i] < min_value) { min_index = i; min_value = v[i]; }
  double sum = min_value;
  for (int i = 0; i < 4; i++)
    if (i != min_index) sum = ln_sum_exp(sum, v[i]);
  return sum;
}

// fgbio MathUtil.maxWithIndex(requireUniqueMaximum = true); -1 = None
FGX_HD int fgbio_unique_max_index(const double* ll) {
  double max = -1.7976931348623157e308;
  int max_index = -1;
  bool assigned = false;
  for (int i = 0; i < 4; i++) {
    double v = ll[i];
    if (m_isnan(v)) continue;
    if (!assigned || v > max) { max = v; max_index = i; assigned = true; }
    else if (m_fabs(v - max) <= FGX_DBL_EPSILON) max_index = -1;
  }
  if (!assigned || max_index < 0) return -1;
  return max_index;
}

FGX_HD bool ulps_eq0(double a, double b, uint32_t max_ulps) {
  double diff = (a > b) ? (a - b) : (b - a);
  if (diff <= 0.0) return true;
  if (m_isnan(a) || m_isnan(b)) return false;
  bool sa = (fgx_asuint64(a) >> 63) != 0, sb = (fgx_asuint64(b) >> 63) != 0;
  if (sa != sb) return false;
  uint64_t ia = fgx_asuint64(a), ib = fgx_asuint64(b);
  return (ia <= ib) ? (ib - ia <= max_ulps) : (ia - ib <= max_ulps);
}

FGX_HD int ulp_unique_max_index(const double* ll) {
  double max = m_neg_inf();
  int max_index = -1;
  for (int i = 0; i < 4; i++)
    if (ll[i] > max) { max = ll[i]; max_index = i; }
  if (max_index < 0 || !m_isfinite(max)) return -1;
  int tied = 0;
  for (int i = 0; i < 4; i++)
    if (ulps_eq0(ll[i], max, 4)) tied++;
  return tied == 1 ? max_index : -1;
}

// ---- tables ------------------------------------------------------------------------------
// One set per (pre-UMI, post-UMI) pair.  Built on the host by build_tables() with the same
// functions the kernels run, uploaded once; 3 KB, read through the scalar/constant path.
struct ConsensusTables {
  double correct[94];        // AdjustedProbabilityTables::correct
  double error_per_alt[94];  // AdjustedProbabilityTables::error_per_alt
  double thresholds[94];     // UnanimousGapTables::thresholds
  double cerr_min[94];       // UnanimousGapTables::cerr_min
  double ln_error_pre_umi;
  uint32_t cap;
  uint32_t tie_rule;         // 0 FgbioCompat, 1 UlpRelative
  // Search hint for the gap → quality bracket lookup: qguess[k] = number of thresholds <= k/8, so a gap in
  // [k/8, (k+1)/8) has its upper bound at or just above qguess[k] (derived data, not a reference table).
  uint8_t qguess[256];
  double cap_threshold;      // thresholds[cap]
  double half_cerr_at_cap;   // 0.5 * cerr_min[cap - 1], the cap-region budget of unanimous_fast_path
  // margin_scale[q] >= (1 / cerr_min[q]) * (1 + 2^-51): `x * margin_scale[q]` is an upper bound of the reference's
  // `x / cerr_min[q]` (unanimous_margin), so a gate that holds with it holds with the exact margin; only when it fails
  // does a kernel pay for the f64 division (derived data, not a reference table)
  double margin_scale[94];
};

FGX_HD uint8_t unanimous_quality_from_gap(double gap, double ln_error_pre_umi) {
  double v[4] = {0.0, -gap, -gap, -gap};
  double ln_sum = ln_sum_exp_array4(v);
  double ln_posterior = 0.0 - ln_sum;
  double ln_consensus_error = ln_one_minus_exp(ln_posterior);
  double ln_final = ln_error_prob_two_trials(ln_error_pre_umi, ln_consensus_error, nullptr);
  return ln_prob_to_phred(ln_final);
}

FGX_HD double consensus_error(double gap) {
  double e = 3.0 * g_exp(-gap);
  return e / (1.0 + e);
}

FGX_HD double unanimous_margin(double w, double l, double cerr_lower_bound) {
  return 16.0 * (FGX_DBL_EPSILON / 2.0) * (m_fabs(w) + m_fabs(l)) / cerr_lower_bound;
}

inline void build_tables(ConsensusTables& t, uint8_t pre, uint8_t post, uint32_t tie_rule) {
  double ln_error_post = phred_to_ln_error_prob(post);
  double ln_three = g_log(3.0);
  for (int q = 0; q <= FGX_MAX_PHRED; q++) {
    double ln_error_seq = phred_to_ln_error_prob((uint8_t)q);
    double adjusted = ln_error_prob_two_trials(ln_error_post, ln_error_seq, nullptr);
    t.correct[q] = ln_one_minus_exp(adjusted);
    t.error_per_alt[q] = adjusted - ln_three;
  }
  double ln_pre = phred_to_ln_error_prob(pre);
  t.ln_error_pre_umi = ln_pre;
  uint8_t q0 = unanimous_quality_from_gap(0.0, ln_pre);
  uint8_t qmax = unanimous_quality_from_gap(256.0, ln_pre);
  for (int q = 0; q <= FGX_MAX_PHRED; q++) {
    t.thresholds[q] = m_inf();
    if (q0 >= q) { t.thresholds[q] = 0.0; continue; }
    if (qmax < q) continue;
    double too_small = 0.0, wide_enough = 256.0;
    for (int it = 0; it < 64; it++) {
      double mid = 0.5 * (too_small + wide_enough);
      if (unanimous_quality_from_gap(mid, ln_pre) >= q) wide_enough = mid; else too_small = mid;
    }
    t.thresholds[q] = wide_enough;
  }
  t.cap = ln_prob_to_phred(ln_pre);
  for (int q = 0; q <= FGX_MAX_PHRED; q++) t.cerr_min[q] = 0.0;
  for (uint32_t q = 0; q < t.cap && q < 94; q++) t.cerr_min[q] = consensus_error(t.thresholds[q + 1]);
  t.tie_rule = tie_rule;
  for (int k = 0; k < 256; k++) {
    double edge = (double)k * 0.125;
    int n = 0;
    for (int q = 0; q <= FGX_MAX_PHRED; q++) if (t.thresholds[q] <= edge) n++;
    t.qguess[k] = (uint8_t)n;
  }
  for (int q = 0; q <= FGX_MAX_PHRED; q++) {
    const double c = t.cerr_min[q];
    t.margin_scale[q] = c > 0.0 ? (1.0 / c) * (1.0 + 4.0 * FGX_DBL_EPSILON) : m_inf();
  }
  t.cap_threshold = t.thresholds[t.cap < 94 ? t.cap : 93];
  t.half_cerr_at_cap = 0.5 * t.cerr_min[t.cap ? t.cap - 1 : 0];
}

// single_input_consensus_quals (vanilla_caller.rs:469-501)
inline void build_single_input_quals(uint8_t* out94, uint8_t pre, uint8_t post) {
  uint8_t lab = pre < post ? pre : post;
  double ln_lab = phred_to_ln_error_prob(lab);
  for (int q = 0; q <= FGX_MAX_PHRED; q++) {
    uint8_t adj = ln_prob_to_phred(ln_error_prob_two_trials(phred_to_ln_error_prob((uint8_t)q), ln_lab, nullptr));
    out94[q] = adj < FGX_MAX_PHRED ? adj : FGX_MAX_PHRED;
  }
}

// ---- one column -----------------------------------------------------------------------------
// Kahan-compensated f64 accumulation of the four base log-likelihoods, reads added serially in
// file order (summation order is observable through the one-ULP fgbio tie rule).
struct ColumnAcc {
  double s[4], c[4];
  uint32_t obs[4];
  FGX_HD void reset() {
    for (int i = 0; i < 4; i++) { s[i] = 0.0; c[i] = 0.0; obs[i] = 0; }
  }
  // idx in 0..3 (A,C,G,T); q already min(qual, 93)
  FGX_HD void add(int idx, double ln_correct, double ln_err) {
#pragma unroll
    for (int lane = 0; lane < 4; lane++) {
      double v = (lane == idx) ? ln_correct : ln_err;
      double y = v - c[lane];
      double t = s[lane] + y;
      c[lane] = (t - s[lane]) - y;
      s[lane] = t;
      obs[lane] += (lane == idx) ? 1u : 0u;     // no dynamic register indexing (would spill to scratch)
    }
  }
  FGX_HD uint32_t obs_of(int idx) const { return idx == 0 ? obs[0] : idx == 1 ? obs[1] : idx == 2 ? obs[2] : idx == 3 ? obs[3] : 0u; }
  FGX_HD uint32_t contributions() const { return obs[0] + obs[1] + obs[2] + obs[3]; }
};

// The same accumulation with Kahan chains BY ORDER OF APPEARANCE instead of one chain per base: chain 1 = the first base seen,
// chains 2 / 3 = the second / third distinct base, chain R = every base not seen yet.  base_builder.rs:836-868 adds `correct`
// to the observed lane and `error_per_alt` to the other three, lane by lane, in read order — so the lanes of the bases not yet
// observed receive identical additions and hold bit-identical (sum, compensation) pairs: one chain stands for all of them, a
// new base opens its chain as a copy of chain R, and once three bases have been seen chain R IS the fourth base.  A column
// that shows a single base (99 % of them) costs two chains = 8 f64 operations per observation instead of 16; the lanes come
// out bit-identical to ColumnAcc's.  `code` is the BAM 4-bit code in read orientation; only the one-hot codes A C G T count.
struct ChainAcc {
  double s1, c1, sR, cR, s2, c2, s3, c3;
  uint32_t st;                 // 0: nothing seen; a one-hot code: that base only, so far; 16: several bases
  uint32_t b1, b2, b3;         // several bases: codes of chains 1..3, 0 = not opened
  uint32_t n1, n2, n3, nR;     // observations per chain
  FGX_HD void reset() { s1 = c1 = sR = cR = s2 = c2 = s3 = c3 = 0.0; st = b1 = b2 = b3 = n1 = n2 = n3 = nR = 0; }
  static FGX_HD void kahan(double& s, double& c, double v) { const double y = v - c; const double t = s + y; c = (t - s) - y; s = t; }
  FGX_HD void add(bool valid, uint32_t code, double ln_correct, double ln_err) {
    const uint32_t sc = st | code;
    const bool hot = valid && sc == code;              // nothing seen yet, or this very base only
    if (hot) { st = sc; n1++; kahan(s1, c1, ln_correct); kahan(sR, cR, ln_err); }
    if (valid && !hot) {
      if (st != 16) { b1 = st; st = 16; }
      if (code != b1 && code != b2 && code != b3) {
        if (b2 == 0) { b2 = code; s2 = sR; c2 = cR; }
        else if (b3 == 0) { b3 = code; s3 = sR; c3 = cR; }
      }
      const bool h1 = code == b1, h2 = code == b2, h3 = code == b3;
      kahan(s1, c1, h1 ? ln_correct : ln_err);
      kahan(s2, c2, h2 ? ln_correct : ln_err);          // (chains 2 / 3 hold nothing of value until opened: overwritten then)
      kahan(s3, c3, h3 ? ln_correct : ln_err);
      kahan(sR, cR, (h1 || h2 || h3) ? ln_err : ln_correct);
      n1 += h1; n2 += h2; n3 += h3; nR += !(h1 || h2 || h3);
    }
  }
  // lanes of A, C, G, T (codes 1, 2, 4, 8); bases never seen read chain R
  FGX_HD void finish(double* ll, uint32_t* obs) const {
    const uint32_t f1 = st != 16 ? st : b1;
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
      const uint32_t code = 1u << i;
      ll[i] = code == f1 ? s1 : code == b2 ? s2 : code == b3 ? s3 : sR;
      obs[i] = code == f1 ? n1 : code == b2 ? n2 : code == b3 ? n3 : nR;
    }
  }
};

// try_unanimous_fast_path; returns true when the table answer is established.
FGX_HD bool unanimous_fast_path(const ConsensusTables& T, const double* ll, const uint32_t* obs, int* base_idx, uint8_t* qual) {
  int observed = -1, n_obs = 0;
  for (int i = 0; i < 4; i++)
    if (obs[i] > 0) { n_obs++; observed = i; }
  if (n_obs != 1) return false;
  double w = ll[observed];
  double l = ll[(observed + 1) & 3];
  double gap = w - l;
  if (!(m_isfinite(gap) && gap > FGX_DBL_EPSILON)) return false;
  uint32_t cap = T.cap;
  double cap_threshold = T.thresholds[cap];
  if (gap >= cap_threshold) {
    double delta = unanimous_margin(w, l, 1.0);
    double cerr_at_cap = T.cerr_min[cap - 1];
    if (gap - cap_threshold >= FGX_LN_2 && delta < 0.5 * cerr_at_cap) { *base_idx = observed; *qual = (uint8_t)cap; return true; }
    return false;
  }
  uint32_t lo = 0, hi = 94;
  while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (T.thresholds[mid] <= gap) lo = mid + 1; else hi = mid; }
  uint32_t q = lo - 1;
  double margin = unanimous_margin(w, l, T.cerr_min[q]);
  if (gap - T.thresholds[q] > margin && T.thresholds[q + 1] - gap > margin) { *base_idx = observed; *qual = (uint8_t)q; return true; }
  return false;
}

// call_full; base_idx = -1 for the ('N', 2) tie no-call
FGX_HD void call_full(const ConsensusTables& T, const double* ll, int* base_idx, uint8_t* qual) {
  double ln_sum = ln_sum_exp_array4(ll);
  int max_idx = T.tie_rule == 0 ? fgbio_unique_max_index(ll) : ulp_unique_max_index(ll);
  if (max_idx < 0) { *base_idx = -1; *qual = FGX_MIN_PHRED; return; }
  double ln_post = ll[max_idx] - ln_sum;
  double ln_cons_err = ln_one_minus_exp(ln_post);
  double ln_final = ln_error_prob_two_trials(T.ln_error_pre_umi, ln_cons_err, nullptr);
  *base_idx = max_idx;
  *qual = ln_prob_to_phred(ln_final);
}

// call() without the call_full leg: true when the answer is established (no observations, or the
// unanimous fast path's sufficient condition holds); false → the column needs call_full.
FGX_HD bool column_call_fast(const ConsensusTables& T, const double* ll, const uint32_t* obs, int* base_idx, uint8_t* qual) {
  if (obs[0] + obs[1] + obs[2] + obs[3] == 0) { *base_idx = -1; *qual = FGX_MIN_PHRED; return true; }
  return unanimous_fast_path(T, ll, obs, base_idx, qual);
}

// ConsensusBaseBuilder::call
FGX_HD void column_call(const ConsensusTables& T, const double* ll, const uint32_t* obs, int* base_idx, uint8_t* qual) {
  if (obs[0] + obs[1] + obs[2] + obs[3] == 0) { *base_idx = -1; *qual = FGX_MIN_PHRED; return; }
  if (unanimous_fast_path(T, ll, obs, base_idx, qual)) return;
  call_full(T, ll, base_idx, qual);
}

}  // namespace fgx
