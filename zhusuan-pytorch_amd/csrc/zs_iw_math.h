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

// One datapoint on one wavefront, lane = particle (K <= 64): the row scalars by butterflies, then the per-particle terms
// (the body of k_iw_reduce_wave; also the tail of the fused generator-side objective, zs_iwfused.hip).  `l` = log w of this
// lane's particle (-inf on lanes >= K), `lq` its log q.  Writes the two coefficient rows (scaled) and, from lane 0, the
// per-datapoint cost / bound when the pointers are given; returns the datapoint's cost (uniform across the wave).
__device__ __forceinline__ float iw_wave_row(float l, float lq, bool on, int lane, int K, int estimator, float scale,
                                             int64_t b, float* __restrict__ cost_b, float* __restrict__ bound_b,
                                             float* __restrict__ coef_p, float* __restrict__ coef_q) {
  IwRow r;
  r.m1 = wave_max(l);
  const unsigned long long hit = __ballot(on && l == r.m1);
  r.jstar = hit ? (int)__ffsll((long long)hit) - 1 : 0;
  r.m2 = wave_max((on && lane != r.jstar) ? l : -INFINITY);
  const float e = on ? expf(l - r.m1) : 0.f;
  r.S = wave_sum(e);
  r.sumL = wave_sum(on ? l : 0.f);
  r.S2 = 0.f;
  if (estimator == ZS_IW_VIMCO) r.S2 = wave_sum((on && lane != r.jstar) ? expf(l - r.m2) : 0.f);
  r.logS = logf(r.S);
  r.invK = 1.0f / (float)K;
  r.invKm1 = K > 1 ? 1.0f / (float)(K - 1) : 0.f;
  float wt = 0.f, ct = 0.f, cq = 0.f;
  if (on) iw_particle(r, l, lq, lane, estimator, wt, ct, cq);
  const float cost = wave_sum(ct);
  if (on) {
    if (coef_p) coef_p[b * K + lane] = -wt * scale;
    if (coef_q) coef_q[b * K + lane] = cq * scale;
  }
  if (lane == 0) {
    if (cost_b) cost_b[b] = cost;
    if (bound_b) bound_b[b] = logf(r.S * r.invK) + r.m1;  // log(mean(exp(x - max))) + max, utils.py:18
  }
  return cost;
}


}  // namespace zs
