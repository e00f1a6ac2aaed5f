// Per-particle arithmetic of the importance-weighted reduction (K4), shared by the kernels of zs_iw.hip and by the
// host-side sanitizer test (tests/host_math/zs_host_math.hip).  See zs_iw.hip for the layout and the O(K) leave-one-out.
#pragma once
#include "zs_common.h"
#include "../../include/zs_hip.h"

namespace zs {

struct IwRow {  // row-level scalars, uniform across the lanes that own the row
  float m1, m2, S, S2, sumL, logS;
  int jstar;
  float invK, invKm1;
};

// per-particle outputs given the row scalars.
// Learning signal of VIMCO: signal_j = LME(l) - LME(l with l_j -> sub_j) = log(S) - log(S - e_j + t_j),
// t_j = exp(sub_j - m1).  Formed as -log1p((t_j - e_j)/S) it carries no cancellation between two
// ~|log w|-sized numbers (the fp32 reference loses ~1e-5 absolute there).  For the arg-max particle,
// when it dominates the row (S < 2), S - e_j would cancel instead: there the sum over the other
// particles S2 (taken relative to the second maximum m2) is used directly.
// `e` = exp(l - m1), which the callers' row sums already formed
ZS_HD void iw_particle_e(const IwRow& r, float l, float lq, float e, int j, int estimator,
                         float& wt, float& cost_term, float& cq) {
  wt = e / r.S;
  cost_term = -wt * l;
  cq = wt;
  if (estimator == ZS_IW_VIMCO) {
    const float sub = (r.sumL - l) * r.invKm1;
    float signal;
    if (j != r.jstar || r.S >= 2.0f) {
      signal = -log1pf((expf(sub - r.m1) - e) / r.S);
    } else {
      const float sx = r.S2 + expf(sub - r.m2);
      signal = (r.logS - logf(sx)) + (r.m1 - r.m2);
    }
    cost_term -= lq * signal;
    cq = wt - signal;
  }
}
// The same with the quarter-rate instructions only (K4 with thousands of datapoints, where the precise expf / log1pf / two
// divisions per particle set the pace): 1 / S once per lane (`invS`), exp as v_exp_f32 of x * log2(e), log1p as a cubic for
// |u| < 2^-5 and v_log_f32(1 + u) otherwise.  Relative error of the exponentials 6e-8 * |x| (the particles that matter have
// |x| < ~15), of log1p < 3e-7 for small and 6e-8 / |u| <= 2e-6 for large arguments: checked against the float64 truth by the
// same gate as the precise form (tests/test_cabi.py::test_hip_iw_reduce_lane_groups, tests/host_math).
ZS_HD float log1p_fast(float u) {
  const float small = u * (1.0f - u * (0.5f - u * 0.33333334f));
  const float big = ln_fast(1.0f + u);
  return (u < 0.03125f && u > -0.03125f) ? small : big;
}
ZS_HD void iw_particle_fast(const IwRow& r, float invS, float l, float lq, float e, int j, int estimator,
                            float& wt, float& cost_term, float& cq) {
  wt = e * invS;
  cost_term = -wt * l;
  cq = wt;
  if (estimator == ZS_IW_VIMCO) {
    const float sub = (r.sumL - l) * r.invKm1;
    float signal;
    if (j != r.jstar || r.S >= 2.0f) {
      signal = -log1p_fast((exp_fast(sub - r.m1) - e) * invS);
    } else {
      const float sx = r.S2 + expf(sub - r.m2);
      signal = (r.logS - logf(sx)) + (r.m1 - r.m2);
    }
    cost_term -= lq * signal;
    cq = wt - signal;
  }
}
ZS_HD void iw_particle(const IwRow& r, float l, float lq, int j, int estimator,
                                            float& wt, float& cost_term, float& cq) {
  iw_particle_e(r, l, lq, expf(l - r.m1), j, estimator, wt, cost_term, cq);
}

// One datapoint on one wavefront, lane = particle (K <= 64): the row scalars by DPP reductions (zs_common.h), then the per-particle terms
// (the body of k_iw_reduce_wave; also the tail of the fused generator-side objective, zs_iwpersist.h).  `l` = log w of this
// lane's particle (-inf on lanes >= K), `lq` its log q.  Writes the two coefficient rows (scaled) and, from lane 0, the
// per-datapoint cost / bound when the pointers are given; returns the datapoint's cost (uniform across the wave).
__device__ __forceinline__ float iw_wave_row(float l, float lq, bool on, int lane, int K, int estimator, float scale,
                                             int64_t b, float* __restrict__ cost_b, float* __restrict__ bound_b,
                                             float* __restrict__ coef_p, float* __restrict__ coef_q) {
  // (the arithmetic of the lane-group kernel, zs_iw.hip: v_exp_f32 exponentials, one reciprocal of S per lane, S2 / log S only for
  //  the rows whose arg-max particle reads them -- held to the float64 truth by the same gates as the precise form, tests/host_math
  //  and tests/test_cabi.py::test_hip_iw_reduce; round 5: this wave is the tail of the fused objective, where its ~400 extra
  //  instructions of expf / log1pf / two divisions per lane sat on the critical path of every launch)
  IwRow r;
  r.m1 = wave_max_all(l);
  const unsigned long long hit = __ballot(on && l == r.m1);
  r.jstar = hit ? (int)__ffsll((long long)hit) - 1 : 0;
  r.m2 = wave_max_all((on && lane != r.jstar) ? l : -INFINITY);
  const float e = on ? exp_fast(l - r.m1) : 0.f;
  r.S = wave_sum_all(e);
  r.sumL = wave_sum_all(on ? l : 0.f);
  r.S2 = 0.f;
  r.logS = 0.f;
  if (estimator == ZS_IW_VIMCO && r.S < 2.0f) {                // (wave-uniform)
    r.S2 = wave_sum_all((on && lane != r.jstar) ? expf(l - r.m2) : 0.f);
    r.logS = logf(r.S);
  }
  r.invK = 1.0f / (float)K;
  r.invKm1 = K > 1 ? 1.0f / (float)(K - 1) : 0.f;
  const float invS = 1.0f / r.S;
  float wt = 0.f, ct = 0.f, cq = 0.f;
  if (on) iw_particle_fast(r, invS, l, lq, e, lane, estimator, wt, ct, cq);
  const float cost = wave_sum_all(ct);
  if (on) {
    if (coef_p) coef_p[b * K + lane] = -wt * scale;
    if (coef_q) coef_q[b * K + lane] = cq * scale;
  }
  if (lane == 0) {
    if (cost_b) cost_b[b] = cost;
    if (bound_b) bound_b[b] = ln_fast(r.S * r.invK) + r.m1;  // log(mean(exp(x - max))) + max, utils.py:18 (v_log_f32: 1 ulp of log2)
  }
  return cost;
}


// ================================================================================================ IW1's batch mean (zs_iwpersist.h)
// Host-callable on purpose: tests/host_math checks the fixed-point mean against a long-double mean over twenty-odd orders of
// magnitude (small costs, mixed signs, B = 1 .. 32 768).
// ------------------------------------------------------------------------------------------------ accumulator layout
// acc[0]        word A, total                      acc[32]       (unused)
// acc[1 .. 16]  word A, shards                     acc[33 .. 48] word B, shards (no total: the finisher sums them)
// Both words: bits 63 / 62 / 61 = sticky NaN / +inf / -inf (A total only), bits [S, 61) = workgroups counted, bits [0, S) = sum of
// (value + BIAS) over the datapoints counted, S = 61 - ZS_IW1_CNT_BITS, BIAS = 2^bias_bits, bias_bits = S - 1 - cb, cb = ceil(log2 R).
#define ZS_IW1_CNT_BITS 10          // up to 1023 workgroups (the grid is one per CU)
#define ZS_IW1_S (61 - ZS_IW1_CNT_BITS)
#define ZS_IW1_BOUND_BITS 24        // |cost| < 2^24 per datapoint, else the mean is NaN
#define ZS_IW1_B_OFF 32
// (ZS_IW1_POISON_WORD, include/zs_hip.h: raised by a watcher that gave up; the host re-zeroes the accumulator)
#define ZS_IW1_FLAG_NAN (1ull << 63)
#define ZS_IW1_FLAG_PINF (1ull << 62)
#define ZS_IW1_FLAG_NINF (1ull << 61)

struct Iw1Fixed {
  long long a, b;       // round(cost * 2^s1) and round(residual * 2^bias_bits)
  unsigned flags;       // 4 = NaN (or out of range), 2 = +inf, 1 = -inf
};
__host__ __device__ __forceinline__ int iw1_bias_bits(int cb) { return ZS_IW1_S - 1 - cb; }
// cost -> its two fixed-point words (exact: the product and the residual are representable in double)
__host__ __device__ __forceinline__ Iw1Fixed iw1_fixed(float cost, int cb) {
  Iw1Fixed o = {0, 0, 0u};
  const int bias_bits = iw1_bias_bits(cb), s1 = bias_bits - ZS_IW1_BOUND_BITS;
  if (cost != cost) { o.flags = 4u; return o; }
  if (cost == INFINITY) { o.flags = 2u; return o; }
  if (cost == -INFINITY) { o.flags = 1u; return o; }
  const double c = (double)cost;
  if (!(fabs(c) < (double)(1ull << ZS_IW1_BOUND_BITS))) { o.flags = 4u; return o; }
  const double scaled = s1 >= 0 ? c * (double)(1ull << s1) : c / (double)(1ull << -s1);
  const double ra = rint(scaled);
  o.a = (long long)ra;
  o.b = (long long)rint((scaled - ra) * (double)(1ull << bias_bits));
  return o;
}
// the mean from the two completed sums (biases removed)
__host__ __device__ __forceinline__ float iw1_mean(long long sum_a, long long sum_b, unsigned long long flags, int cb, int64_t R) {
  if ((flags & ZS_IW1_FLAG_NAN) || ((flags & ZS_IW1_FLAG_PINF) && (flags & ZS_IW1_FLAG_NINF))) return __builtin_nanf("");
  if (flags & ZS_IW1_FLAG_PINF) return INFINITY;
  if (flags & ZS_IW1_FLAG_NINF) return -INFINITY;
  const int bias_bits = iw1_bias_bits(cb), s1 = bias_bits - ZS_IW1_BOUND_BITS;
  // (scalings by powers of two: exact, one instruction each -- this runs once per launch, at its very end, on the critical path)
  const double v = (double)sum_a + __builtin_ldexp((double)sum_b, -bias_bits);
  const double total = __builtin_ldexp(v, -s1);
  return (float)(total / (double)R);
}

}  // namespace zs
