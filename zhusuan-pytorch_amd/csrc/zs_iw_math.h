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
ZS_HD void iw_particle(const IwRow& r, float l, float lq, int j, int estimator,
                                            float& wt, float& cost_term, float& cq) {
  iw_particle_e(r, l, lq, expf(l - r.m1), j, estimator, wt, cost_term, cq);
}


}  // namespace zs
