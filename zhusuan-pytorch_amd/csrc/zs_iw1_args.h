// IW1: the argument block of the fused generator-side objective (zs_iwpersist.h; filled by zs_bernoulli.hip).  The struct is the
// kernel's ONLY argument: it starts the kernel-argument segment, and the once-per-datapoint code reads its cold fields from there.
#pragma once
#include "zs_common.h"
#include "../../include/zs_hip.h"

namespace zs {

struct Iw1Args {
  // Bernoulli term: p (probabilities or logits) [K, R, D4 float4], x rows [R, D4] (shared by the particles) or [K * R, D4]
  const float4* p;
  const float4* x;
  int x_full;
  int64_t R;
  int D4;
  int K;
  // Normal term of a given value (optional): z [K, R, Dz4 float4], mean / scale [R, Dz4] or one scalar each.  Without the term
  // (has_z == 0) the three pointers are redirected to readable memory (p) with Dz4 = 1: the kernel loads unconditionally.
  const float4* z;
  const float* pmu;
  const float* psg;
  int has_z, pmu_scalar, psg_scalar, psg_is_logstd;
  int Dz4;
  // ready-made rows of further generator nodes (optional) and log q: K-fastest [R, ld]
  const float* rows_a;
  int64_t ld_a;
  const float* logq;
  int64_t ld_q;
  int estimator;
  // outputs
  float* lp_x;        // [R, K] row sums of the Bernoulli term
  float* lp_z;        // [R, K] row sums of the Normal term (optional)
  float* cost_b;
  float* bound_b;
  float* coef_p;      // [R, K]
  float* coef_q;      // [R, K]
  float scale;
  float* mean_cost;
  unsigned long long* acc;   // the batch mean's accumulator words (zs_iw_math.h): shard words of the two sums; zero between launches
  int cb, bound_bits, sharded;      // (bound_bits, sharded: reserved -- the release kernel's mean has one form)
  float inv_B;
  int variant;        // (reserved: 0.  tools/lab's timing variants read it)
};

// layout of the accumulator for R datapoints
__host__ __device__ __forceinline__ int iw1_cb(int64_t R) {
  int cb = 0;
  while ((1ll << cb) < R) ++cb;
  return cb;
}

#define ZS_IW1_SHARDS 16      // shard words per sum: workgroup g adds to shard g mod 16 (256 same-address atomics arriving together
                              // serialise in L2, ~11 ns each: MI355X_MICROARCH.md "dequeue")

}  // namespace zs
