// Per-element arithmetic of the Logistic / Uniform forms of the shared kernels (zs_sample_tile.h), __host__ __device__ like
// zs_common.h's helpers so that the host-side sanitizer test (tests/host_math/zs_host_math.hip) runs the same code.
#pragma once
#include "zs_common.h"

namespace zs {

// Logistic draw from a uniform u in (0, 1) (logistic.py:64-66): eps = log u - log(1 - u); its own standard log-density
// -eps - 2 softplus(-eps) is log u + log(1 - u): the two logarithms serve both.
ZS_HD void logistic_draw(float u, float& eps, float& dens) {
  const float a = ln_fast(u), b = ln_fast(1.0f - u);
  eps = a - b;
  dens = a + b;
}

// -(standard Logistic log-density) at t = diff * inv_scale:  t + 2 softplus(-t) = |t| + 2 log1p(exp(-|t|))
// (logistic.py:81-82; the form is even in t, so neither exponential can overflow)
ZS_HD float logistic_neg_lp_term(float diff, float inv_scale) {
  const float at = __builtin_fabsf(diff * inv_scale);
  const float e = exp2_fast(at * -1.44269504088896341f);
  return at + 2.0f * ZS_LN2 * log2_fast(1.0f + e);
}

// Backward of the Logistic log-density for one element, g = the row's incoming gradient, u = diff / scale, h = tanh(u / 2):
//   gx = -g h / scale,   a += g h / scale (d loc),   b += g (h u - 1) / scale (d scale)
ZS_HD void logistic_ksum_elem(float g, float diff, float inv_scale, float& gx, float& a, float& b) {
  const float u = diff * inv_scale;
  const float e = exp2_fast(__builtin_fabsf(u) * -1.44269504088896341f);
  float h = (1.0f - e) * rcp_fast(1.0f + e);           // tanh(|u| / 2)
  h = u < 0.f ? -h : h;
  const float gh = g * h * inv_scale;
  gx = -gh;
  a += gh;
  b += g * (h * u - 1.0f) * inv_scale;
}

// torch Uniform.log_prob's support test: lb * ub with lb = low <= x, ub = high > x (uniform.py:78-81)
ZS_HD bool uniform_inside(float x, float low, float high) { return (low <= x) && (high > x); }

}  // namespace zs
