// gate_core.h — the unanimous-column gates of the simplex kernels as host + device functions: what k_simplex_wave2 / k_simplex_seg /
// k_deep_cols decide with on exact (Kahan) sums, and what k_split_cols decides with on approximate (f32) sums.  One source for the
// kernels (fastpath.hip) and for tests/devemu, where the same functions run on the host against the oracle's ConsensusBaseBuilder
// (tests/test_gate_core.py): "a column the approximate gate answers gets the reference's answer" is checked there on millions of columns.
#pragma once
#include <cstdint>
#include "consensus_math.h"

namespace fgx {

// What the unanimous-column gate reads, kept in LDS (1.8 KB; a CU's LDS decides how many workgroups it holds)
struct GateTables {
  double thresholds[98];     // UnanimousGapTables::thresholds; [94..98) repeat [93] (the bracket probe reads up to b + 3 unclamped)
  double margin_scale[94];   // upper bound of 1 / cerr_min[q] (ConsensusTables::margin_scale)
  uint8_t qguess[256];
  double cap_threshold, half_cerr_at_cap;
  uint32_t cap, _pad;
};
// (host and device) the tables of a caller, as the kernels' LDS images hold them
FGX_HD void fill_gate_tables(GateTables& G, const ConsensusTables& t) {
  for (uint32_t i = 0; i < 98; i++) G.thresholds[i] = t.thresholds[i < 93 ? i : 93];
  for (uint32_t i = 0; i < 94; i++) G.margin_scale[i] = t.margin_scale[i];
  for (uint32_t i = 0; i < 256; i++) G.qguess[i] = t.qguess[i];
  G.cap_threshold = t.cap_threshold; G.half_cerr_at_cap = t.half_cerr_at_cap; G.cap = t.cap; G._pad = 0;
}
struct CallConst { double cap_threshold, half_cerr_at_cap; uint32_t cap; };

// The {correct, error_per_alt} table of k_split_cols's hot loop, rounded to f32 (round to nearest: |x~ - x| <= 2^-24 |x|), row i for the
// quality byte i (qualities above 93 take row 93).  correct[0] is ln 0: the f32 table holds -10^30 in its place (the loop adds
// pair * {1, 0}: an infinite entry times 0 would be a NaN).  A column that really observes such a quality (--min-input-base-quality 0) sums
// to about -10^30: its gap is negative, no gate answers, and the column travels to k_call_full with its observations like every other
// undecided one.
FGX_HD void fill_pairs_f32(float (*pairf)[2], uint32_t rows, const ConsensusTables& t) {
  for (uint32_t i = 0; i < rows; i++) {
    const uint32_t q = i < 93 ? i : 93;
    pairf[i][0] = (float)t.correct[q]; pairf[i][1] = (float)t.error_per_alt[q];
    for (int k = 0; k < 2; k++) if (!m_isfinite((double)pairf[i][k])) pairf[i][k] = -1.0e30f;
  }
}

// The unanimous-column gate of try_unanimous_fast_path (base_builder.rs:883-994) for a column whose observations all show
// one base: w = Kahan sum of correct[q], l = Kahan sum of error_per_alt[q].  Same decisions as column_call_fast_lds;
// `cerr_min` (global memory) is read only by the columns the division-free bound cannot decide.
FGX_HD bool unanimous_call_lds(const GateTables& T, const double* __restrict__ cerr_min, const CallConst& K, double w, double l, uint32_t* qual) {
  const double gap = w - l;
  if (!(m_isfinite(gap) && gap > FGX_DBL_EPSILON)) return false;
  // unanimous_margin(w, l, c) = 16 * (EPSILON / 2) * (|w| + |l|) / c, evaluated left to right: `num / c`
  const double num = 16.0 * (FGX_DBL_EPSILON / 2.0) * (m_fabs(w) + m_fabs(l));
  if (gap >= K.cap_threshold) {
    if (gap - K.cap_threshold >= FGX_LN_2 && num < K.half_cerr_at_cap) { *qual = K.cap; return true; }   // (num / 1.0 == num)
    return false;
  }
  uint32_t lo;
  double thq, thq1;
  bool probed = false;
  if (gap < 32.0) {
    const uint32_t b = T.qguess[(uint32_t)(gap * 8.0)];
    if (b >= 1) {
      // thresholds[0 .. b) <= gap for certain: count how many of the next four are, then read the bracket itself (two more LDS
      // reads instead of selecting among five registers: selects are vector instructions, LDS reads are not)
      const double* tb = &T.thresholds[b];
      const uint32_t c = (tb[0] <= gap) + (tb[1] <= gap) + (tb[2] <= gap) + (tb[3] <= gap);
      if (c < 4) { lo = b + c; thq = tb[(int)c - 1]; thq1 = tb[c]; probed = true; }
    }
  }
  if (!probed) {
    uint32_t a = 0, hi = 94;
    while (a < hi) { uint32_t mid = a + (hi - a) / 2; if (T.thresholds[mid] <= gap) a = mid + 1; else hi = mid; }
    lo = a; thq = T.thresholds[lo - 1]; thq1 = T.thresholds[lo];
  }
  const uint32_t q = lo - 1;
  const double below = gap - thq, above = thq1 - gap;
  const double ub = num * T.margin_scale[q];              // >= the exact margin: passing with it is passing the reference's gate
  if (below > ub && above > ub) { *qual = q; return true; }
  const double margin = num / cerr_min[q];                // the reference's margin, for the columns the bound could not decide
  if (below > margin && above > margin) { *qual = q; return true; }
  return false;
}

// The unanimous gate on APPROXIMATE sums.  k_split_cols's hot loop adds a column's single-base observations in f32 — w~ = sum of
// fl32(correct[q_i]), l~ = sum of fl32(error_per_alt[q_i]), n plain additions in file order — where the reference holds the f64 Kahan sums
// w, l (base_builder.rs:836-868).  Every addend of a sum has the same sign (logarithms of probabilities), so with u = 2^-24
//     |w~ - w| <= ((n - 1) u / (1 - (n - 1) u) (1 + u) + u) |W| + 2^-51 |W|  <=  (n + 1) u |w~| (1 + 10^-3)      (n <= 1000)
// and the same for l: the reference's gap lies within  B = (n + 2) u (|w~| + |l~|)  of gap~ = w~ - l~, and its |w| + |l| within a factor
// 1 + 10^-4 of ours.  try_unanimous_fast_path (base_builder.rs:883-994) answers q when the gap clears both edges of its bracket by the
// margin 16 (eps / 2) (|w| + |l|) / cerr_min[q], the cap when gap - thresholds[cap] >= ln 2 and the margin numerator is below half the
// cap's consensus error, and defers to call_full otherwise.  Here: the SAME answer whenever gap~ clears the edges by margin~ (1 + 10^-4)
// + B (then the reference's gap is in the same bracket and clears its own margin: it answers q too), the cap when gap~ - B clears
// thresholds[cap] + ln 2 — and `false` in every other case, which sends the column's OBSERVATIONS to k_call_full: it redoes the sums as
// the reference does (four-lane Kahan), applies the reference's gate and, failing that, call_full.  A `false` too many costs time, never
// a byte.  (margin_scale[q] >= 1 / cerr_min[q], consensus_math.h.)
FGX_HD bool unanimous_call_approx(const GateTables& T, const CallConst& K, float wf, float lf, uint32_t n, uint32_t* qual) {
  const double w = (double)wf, l = (double)lf;
  const double gap = w - l, mag = m_fabs(w) + m_fabs(l);
  if (!m_isfinite(gap)) return false;
  const double B = (double)(n + 2u) * 5.9604644775390625e-08 * mag;          // (n + 2) 2^-24 (|w~| + |l~|)
  const double num = 16.0 * (FGX_DBL_EPSILON / 2.0) * mag * 1.0001;
  if (!(gap - B > FGX_DBL_EPSILON)) return false;
  if (gap + B >= K.cap_threshold) {
    if ((gap - B) - K.cap_threshold >= FGX_LN_2 && num < K.half_cerr_at_cap) { *qual = K.cap; return true; }
    return false;
  }
  uint32_t lo;
  double thq, thq1;
  bool probed = false;
  if (gap < 32.0) {
    const uint32_t b = T.qguess[(uint32_t)(gap * 8.0)];
    if (b >= 1) {
      const double* tb = &T.thresholds[b];
      const uint32_t c = (tb[0] <= gap) + (tb[1] <= gap) + (tb[2] <= gap) + (tb[3] <= gap);
      if (c < 4) { lo = b + c; thq = tb[(int)c - 1]; thq1 = tb[c]; probed = true; }
    }
  }
  if (!probed) {
    uint32_t a = 0, hi = 94;
    while (a < hi) { uint32_t mid = a + (hi - a) / 2; if (T.thresholds[mid] <= gap) a = mid + 1; else hi = mid; }
    lo = a; thq = T.thresholds[lo - 1]; thq1 = T.thresholds[lo];
  }
  const uint32_t q = lo - 1;
  const double below = gap - thq, above = thq1 - gap;
  const double ub = num * T.margin_scale[q] + B;
  if (below > ub && above > ub) { *qual = q; return true; }
  return false;
}

// The cap answer of unanimous_call_approx, decided in f32.  With gf = fl32(w~ - l~), mf = fl32(|w~| + |l~|) (each within 2^-24 of the real
// value) and M >= n:   gf - mf kB >= thf  and  mf kN < hcf,   kB >= (M + 3) 2^-24 (1 + 2^-20),  thf >= (thresholds[cap] + ln 2)(1 + 2^-20),
// kN >= 16 (eps / 2) 1.0001 (1 + 2^-20),  hcf <= half_cerr_at_cap (1 - 2^-20)
// imply the f64 tests of unanimous_call_approx's cap branch with room to spare (the (M + 3) takes the rounding of gf, every other f32 rounding
// is 2^-24 against factors of 1 + 2^-20; the f64 roundings are 2^-53): `true` here means that function returns (true, cap).  Sums that are
// not finite give NaN comparisons, i.e. `false`.
struct S2PreGate { float kB, thf, hcf; };
FGX_HD S2PreGate s2_pregate_consts(const GateTables& T, uint32_t m) {
  S2PreGate g;
  g.kB = (float)(m + 3u) * 5.9604644775390625e-08f * 1.000002f;
  g.thf = (float)(T.cap_threshold + FGX_LN_2) * 1.000002f;
  g.hcf = (float)T.half_cerr_at_cap * 0.999998f;
  return g;
}
FGX_HD bool s2_cap_pregate(const S2PreGate& g, float wf, float lf) {
  const float gf = wf - lf, mf = __builtin_fabsf(wf) + __builtin_fabsf(lf);
  constexpr float kN = 0x1p-49f * 1.0001f * 1.000002f;
  return (gf - mf * g.kB >= g.thf) && (mf * kN < g.hcf);
}


// The cap WITHOUT sums (round 5: k_split_cols's hot loop).  A single-base column of n observations whose qualities all lie in
// [min_bq, 93] (bytes above 93 count as 93, base_builder.rs:836-868) has the reference's gap w - l >= n dmin - (Kahan error), dmin = the
// smallest correct[q] - error_per_alt[q] of that range.  try_unanimous_fast_path (base_builder.rs:883-994) answers the cap as soon as
// gap - thresholds[cap] >= ln 2 and 16 (eps / 2) (|w| + |l|) < cerr_min[cap - 1] / 2 — so from
//     n_safe = the smallest n with  n dmin >= thresholds[cap] + ln 2 + slack
// observations on EVERY such column is (base, cap), whatever the qualities are: the loop then needs the OR of the bases and a count, no
// table, no sum.  `slack` covers the two Kahan sums (each within 2 eps sum|x| + O(n eps^2) of its real value; 4 eps n_max mag is generous)
// and the roundings of the gate's own subtraction; the margin budget is checked for the deepest column the caller will ask about (n_max).
// Returns FGX_NEVER_CAP when the table has a non-finite entry in the range (quality 0: ln 0), dmin <= 0, or the budget fails: the caller
// then keeps to the sums.  The tests run the oracle's ConsensusBaseBuilder over columns at and above n_safe (tests/test_gate_core.py).
#define FGX_NEVER_CAP 0xFFFFFFFFu
FGX_HD uint32_t unanimous_cap_depth(const ConsensusTables& t, uint32_t min_bq, uint32_t n_max) {
  const uint32_t lo = min_bq < 93u ? min_bq : 93u;
  double dmin = 1.0e300, mag = 0.0;
  for (uint32_t q = lo; q <= 93u; q++) {
    const double c = t.correct[q], e = t.error_per_alt[q];
    if (!m_isfinite(c) || !m_isfinite(e)) return FGX_NEVER_CAP;
    const double d = c - e, a = m_fabs(c) + m_fabs(e);
    dmin = d < dmin ? d : dmin; mag = a > mag ? a : mag;
  }
  const double need = t.cap_threshold + FGX_LN_2;
  if (!(dmin > 0.0) || !m_isfinite(need) || t.cap < 2u || t.cap > 93u) return FGX_NEVER_CAP;
  const double kahan = 4.0 * FGX_DBL_EPSILON * (double)n_max * mag;
  if (!(16.0 * (FGX_DBL_EPSILON / 2.0) * (double)n_max * mag * 1.000001 + kahan < t.half_cerr_at_cap)) return FGX_NEVER_CAP;
  for (uint32_t n = 1; n <= n_max; n++)
    if ((double)n * dmin * (1.0 - 1.0e-12) - kahan - 1.0e-9 >= need) return n;
  return FGX_NEVER_CAP;
}


}  // namespace fgx
