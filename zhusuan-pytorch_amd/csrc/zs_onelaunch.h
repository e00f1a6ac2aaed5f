// Shared pieces of the one-launch kernels (zs_logjoint.hip: LJ1, MS1; zs_layers.hip: PL1, CS1, AB1, PR1): per-precision element
// math, double-precision reductions, and the cross-workgroup hand-off they all end with.
#pragma once
#include "zs_common.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

// ---------------------------------------------------------------- element math, per precision
template <typename T>
struct Mth;
template <>
struct Mth<float> {
  static __device__ __forceinline__ float sigma_of(float v, bool ls) { return ls ? expf(v) : v; }
  // log(sigma) and sigma^-2 as the fp32 kernels form them (v_log_f32 / v_exp_f32; zs_normal.hip)
  static __device__ __forceinline__ void parts(float s, float& logstd, float& prec) {
    const float l2 = log2_fast(s);
    logstd = l2 * ZS_LN2;
    prec = exp2_fast(-2.0f * l2);
  }
  static __device__ __forceinline__ float normal_term(float d, float logstd, float prec) { return normal_lp_term(d, logstd, prec); }
  static __device__ __forceinline__ float bern_term(float p, float x) { return bern_lp2_term(p, x) * ZS_LN2; }
  static __device__ __forceinline__ float bern_dp(float p, float x) { return zs::bern_dp(p, x); }
  static __device__ __forceinline__ float sigmoid(float l) { return sigmoid_fast(l); }
  static __device__ __forceinline__ float sqrt_n(int64_t n) { return sqrtf((float)n); }
};
template <>
struct Mth<double> {
  static __device__ __forceinline__ double sigma_of(double v, bool ls) { return ls ? exp(v) : v; }
  static __device__ __forceinline__ void parts(double s, double& logstd, double& prec) {
    logstd = log(s);
    prec = exp(-2.0 * logstd);
  }
  static __device__ __forceinline__ double normal_term(double d, double logstd, double prec) {
    return (-0.91893853320467274178 - logstd) - 0.5 * prec * (d * d);
  }
  static __device__ __forceinline__ double bern_term(double p, double x) {
    return x * log(p + 1e-8) + (1.0 - x) * log((1.0 - p) + 1e-8);
  }
  static __device__ __forceinline__ double bern_dp(double p, double x) { return x / (p + 1e-8) - (1.0 - x) / ((1.0 - p) + 1e-8); }
  static __device__ __forceinline__ double sigmoid(double l) { return 1.0 / (1.0 + exp(-l)); }
  static __device__ __forceinline__ double sqrt_n(int64_t n) { return sqrt((double)n); }
};

template <typename T>
struct alignas(4 * sizeof(T)) V4 { T v[4]; };

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  return v;
}
// sum over a 256-thread workgroup; valid in every thread.  `sh`: 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// ---------------------------------------------------------------- cross-workgroup hand-off without a release fence
// "Every workgroup writes partial results, the LAST one to arrive combines them" needs the partials to be visible across
// CUs and XCDs (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by other CUs' stores).  An
// agent-scope RELEASE on the ticket does that by writing back the XCD's whole dirty L2 (buffer_wbl2: 1.7 us clean, 6.5 us
// with 16 KB freshly dirtied, per workgroup) -- it was most of these kernels' time.  The cheaper valid form
// (MI355X_MICROARCH.md, inter-workgroup visibility: "ONE lane of each storing workgroup adds to one counter; the workgroup
// whose add came last consumes"):
//   producer  every byte of the hand-off is stored WRITE-THROUGH (relaxed agent-scope atomic store = global_store ... sc1);
//             every storing wave waits for its stores (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds to the ticket
//             with a RELAXED agent-scope atomic;
//   consumer  the workgroup whose add returned count - 1: either it reads the hand-off with sc1 loads only (relaxed agent
//             atomic loads; at most a couple per thread: they are issued one after the other), after a workgroup barrier
//             behind the adding lane -- or that lane runs ONE agent-scope acquire (buffer_inv sc1: invalidates this CU's
//             L1), waits for it, and after a barrier the workgroup reads with plain loads (many per thread, batched).
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ void store_wt(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T load_wt(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ticket_take(unsigned* t) {
  return __hip_atomic_fetch_add(t, 1u, ZS_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed; acq_rel in the strict build, zs_common.h)
}
__device__ __forceinline__ void ticket_return(unsigned* t) { __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // namespace
