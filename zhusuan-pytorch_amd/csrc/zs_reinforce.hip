// Scalar epilogues of the ELBO.
// R1: score-function (REINFORCE) epilogue of the ELBO in one launch (include/zs_hip.h; reference
// zhusuan/variational/elbo.py:163-238).  The operands are the per-datapoint (or already batch-reduced) log-joints, so n
// is small -- one 1024-thread workgroup walks them twice (mean of the learning signal, then the cost); the moving
// mean and the step counter are device state that the kernel itself updates, which makes the objective capturable in
// a hipGraph.  Sums are accumulated in double and combined in a fixed order: deterministic.
#include "zs_common.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

template <typename T>
__global__ __launch_bounds__(1024) void k_reinforce(const T* __restrict__ logp, const T* __restrict__ logq,
                                                    const T* __restrict__ baseline, int64_t Pb, int64_t n, int vr, int do_mean,
                                                    float decay, float* __restrict__ moving_mean, int32_t* __restrict__ local_step,
                                                    T* __restrict__ signal, T* __restrict__ cost, T* __restrict__ resid) {
  __shared__ double sh[16];
  __shared__ float sh_mm;
  const bool has_b = vr && baseline != nullptr;
  float mm = 0.f;
  if (vr) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
      T l = logp[i] - logq[i];
      if (has_b) l -= baseline[Pb == 1 ? 0 : i];
      s += (double)l;
    }
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) {
      const float bc = do_mean ? (float)(s / (double)n) : (float)s;   // !do_mean: n == 1 (checked by the host side)
      float m = *moving_mean;
      m -= (m - bc) * (1.0f - decay);                                   // elbo.py:221
      const int32_t st = *local_step + 1;                               // elbo.py:222
      const float bias = 1.0f - powf(decay, (float)st);                 // elbo.py:223
      m /= bias;                                                        // elbo.py:224 (in place, kept)
      *moving_mean = m;
      *local_step = st;
      sh_mm = m;
    }
    __syncthreads();
    mm = sh_mm;
  }
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const T lp = logp[i], lq = logq[i];
    T l = lp - lq;
    T bcost = (T)0;
    if (has_b) {
      l -= baseline[Pb == 1 ? 0 : i];
      if (resid) resid[i] = l;
      bcost = (T)0.5 * l * l;                                           // elbo.py:210-212
    }
    if (vr) l -= (T)mm;                                                 // elbo.py:225
    if (signal) signal[i] = l;
    const T c = -(lp + l * lq) + bcost;                                 // elbo.py:228
    if (do_mean) acc += (double)c;
    else cost[i] = c;
  }
  if (do_mean) {
    acc = block_sum_1024(acc, sh);
    if (threadIdx.x == 0) cost[0] = (T)(acc / (double)n);
  }
}

// ---- long vectors (n > R1_ONE_BLOCK_MAX): the same epilogue over many workgroups.
// With l0_i = logp_i - logq_i [- baseline_i] the cost is  c_i = -(logp_i + (l0_i - mm) * logq_i) + b_i
//                                                             = -(logp_i + l0_i * logq_i) + b_i  +  mm * logq_i,
// so ONE pass suffices although the moving mean mm depends on every element: each workgroup adds up S1 = sum l0 (for the
// moving mean), S2 = sum(-(logp + l0 * logq) + b) and S3 = sum logq (in double: the two products cancel to the small
// (l0 - mm) * logq) and writes l0 / resid; the last workgroup to arrive (ticket; hand-off as in zs_onelaunch.h: partials
// written through, relaxed ticket) updates the moving mean and writes  cost = (S2 + mm * S3) / n.  The learning signal the
// backward pass needs is l0 - mm: a second, element-wise launch subtracts mm in place (reading it from the module buffer
// the first launch has just written) -- two launches of ~6 us at 10^6 elements instead of one 1024-thread workgroup's 300 us.
constexpr int R1_ONE_BLOCK_MAX = 16384;
constexpr int R1_BLOCK = 256;
constexpr unsigned R1_MAX_BLOCKS = 2048;

__device__ __forceinline__ double block_sum_r1(double v, double* sh) {       // 256 threads; valid in every thread
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename T>
__global__ __launch_bounds__(R1_BLOCK) void k_reinforce_many(const T* __restrict__ logp, const T* __restrict__ logq,
                                                             const T* __restrict__ baseline, int64_t Pb, int64_t n, int vr,
                                                             int do_mean, float decay, float* __restrict__ moving_mean,
                                                             int32_t* __restrict__ local_step, T* __restrict__ signal,
                                                             T* __restrict__ cost, T* __restrict__ resid, double* __restrict__ ws,
                                                             unsigned* __restrict__ ticket) {
  __shared__ double sh[4];
  __shared__ bool last;
  const bool has_b = vr && baseline != nullptr;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * R1_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * R1_BLOCK) {
    const T lp = logp[i], lq = logq[i];
    T l = lp - lq;
    T bcost = (T)0;
    if (has_b) {
      l -= baseline[Pb == 1 ? 0 : i];
      if (resid) resid[i] = l;
      bcost = (T)0.5 * l * l;                                           // elbo.py:210-212
    }
    if (signal) signal[i] = l;                                          // l0: the second launch subtracts the moving mean
    if (do_mean) {
      s1 += (double)l;
      s2 += -((double)lp + (double)l * (double)lq) + (double)bcost;
      s3 += (double)lq;
    } else {
      cost[i] = -(lp + l * lq) + bcost;                                 // (no variance reduction here: vr needs do_mean or n == 1)
    }
  }
  if (!do_mean) return;
  s1 = block_sum_r1(s1, sh);
  s2 = block_sum_r1(s2, sh);
  s3 = block_sum_r1(s3, sh);
  if (threadIdx.x == 0) {
    __hip_atomic_store(ws + 3 * blockIdx.x + 0, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // written through
    __hip_atomic_store(ws + 3 * blockIdx.x + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ws + 3 * blockIdx.x + 2, s3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = __hip_atomic_fetch_add(ticket, 1u, ZS_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  double t1 = 0.0, t2 = 0.0, t3 = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += R1_BLOCK) {
    t1 += __hip_atomic_load(ws + 3 * b + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t2 += __hip_atomic_load(ws + 3 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t3 += __hip_atomic_load(ws + 3 * b + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t1 = block_sum_r1(t1, sh);
  t2 = block_sum_r1(t2, sh);
  t3 = block_sum_r1(t3, sh);
  if (threadIdx.x == 0) {
    double mm = 0.0;
    if (vr) {
      const float bc = (float)(t1 / (double)n);
      float m = *moving_mean;
      m -= (m - bc) * (1.0f - decay);                                   // elbo.py:221
      const int32_t st = *local_step + 1;                               // elbo.py:222
      const float bias = 1.0f - powf(decay, (float)st);                 // elbo.py:223
      m /= bias;                                                        // elbo.py:224 (in place, kept)
      *moving_mean = m;
      *local_step = st;
      mm = (double)m;
    }
    cost[0] = (T)((t2 + mm * t3) / (double)n);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// signal_i -= moving_mean (elbo.py:225), after k_reinforce_many has updated the buffer
template <typename T>
__global__ __launch_bounds__(R1_BLOCK) void k_reinforce_center(T* __restrict__ signal, int64_t n, const float* __restrict__ moving_mean) {
  const T mm = (T)moving_mean[0];
  for (int64_t i = (int64_t)blockIdx.x * R1_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * R1_BLOCK) signal[i] -= mm;
}

template <typename T>
int reinforce(const T* logp, const T* logq, const T* baseline, int64_t Pb, int64_t n, int vr, int do_mean, double decay,
              float* moving_mean, int32_t* local_step, T* signal, T* cost, T* resid, double* workspace, int64_t workspace_len,
              uint32_t* ticket, void* stream) {
  if (n < 0 || Pb < 1) return ZS_EINVAL;
  if (baseline && Pb != 1 && Pb != n) return ZS_EINVAL;
  if (vr && !do_mean && n > 1) return ZS_EINVAL;      // the moving mean is a single number (elbo.py:221)
  if (n == 0) return 0;
  if (!logp || !logq || !cost) return ZS_EINVAL;
  if (vr && (!moving_mean || !local_step)) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (n > R1_ONE_BLOCK_MAX && workspace && ticket) {
    // many workgroups (a thread takes ~8 elements before it strides)
    int64_t nb = (n + R1_BLOCK * 8 - 1) / (R1_BLOCK * 8);
    if (nb > (int64_t)R1_MAX_BLOCKS) nb = R1_MAX_BLOCKS;
    if (workspace_len < 3 * nb) return ZS_EINVAL;
    ZS_LAUNCH(KID_REINFORCE, (k_reinforce_many<T>), dim3((unsigned)nb), dim3(R1_BLOCK), st, logp, logq, baseline, Pb, n, vr, do_mean,
              (float)decay, moving_mean, local_step, signal, cost, resid, workspace, (unsigned*)ticket);
    ZS_CHECK_LAUNCH();
    if (vr && do_mean && signal) {
      ZS_LAUNCH(KID_REINFORCE, (k_reinforce_center<T>), dim3((unsigned)nb), dim3(R1_BLOCK), st, signal, n, (const float*)moving_mean);
      ZS_CHECK_LAUNCH();
    }
    return 0;
  }
  ZS_LAUNCH(KID_REINFORCE, (k_reinforce<T>), dim3(1), dim3(n >= 1024 ? 1024 : (n > 64 ? 256 : 64)), st, logp, logq,
            baseline, Pb, n, vr, do_mean, (float)decay, moving_mean, local_step, signal, cost, resid);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int zs_reinforce_f32(const float* logp, const float* logq, const float* baseline, int64_t Pb, int64_t n,
                                int variance_reduction, int do_mean, double decay, float* moving_mean, int32_t* local_step,
                                float* signal, float* cost, float* resid, double* workspace, int64_t workspace_len,
                                uint32_t* ticket, void* stream) {
  return reinforce<float>(logp, logq, baseline, Pb, n, variance_reduction, do_mean, decay, moving_mean, local_step, signal, cost, resid,
                          workspace, workspace_len, ticket, stream);
}
extern "C" int zs_reinforce_f64(const double* logp, const double* logq, const double* baseline, int64_t Pb, int64_t n,
                                int variance_reduction, int do_mean, double decay, float* moving_mean, int32_t* local_step,
                                double* signal, double* cost, double* resid, double* workspace, int64_t workspace_len,
                                uint32_t* ticket, void* stream) {
  return reinforce<double>(logp, logq, baseline, Pb, n, variance_reduction, do_mean, decay, moving_mean, local_step, signal, cost,
                           resid, workspace, workspace_len, ticket, stream);
}

// ---------------------------------------------------------------- S1: out = sum_t coef_t * sum_i rows_t[i]
namespace {

template <typename T>
struct Terms {
  const T* r[ZS_MAX_TERMS];
  int64_t n[ZS_MAX_TERMS];
  double c[ZS_MAX_TERMS];
};

template <typename T>
__global__ __launch_bounds__(1024) void k_scalar_objective(Terms<T> t, T* __restrict__ out, T* __restrict__ coef_out) {
  __shared__ double sh[16];
  double acc = 0.0;                                        // one weighted accumulation, ONE block reduction for all terms
#pragma unroll
  for (int j = 0; j < ZS_MAX_TERMS; ++j) {
    if (t.r[j] == nullptr) continue;                       // uniform
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < t.n[j]; i += blockDim.x) s += (double)t.r[j][i];
    acc += t.c[j] * s;
    if (coef_out && threadIdx.x == 0) coef_out[j] = (T)t.c[j];
  }
  acc = block_sum_1024(acc, sh);
  if (threadIdx.x == 0) out[0] = (T)acc;
}

template <typename T>
int scalar_objective(const Terms<T>& t, T* out, T* coef_out, void* stream) {
  int64_t nmax = 0;
  bool any = false;
  for (int j = 0; j < ZS_MAX_TERMS; ++j) {
    if (!t.r[j]) continue;
    if (t.n[j] < 0) return ZS_EINVAL;
    any = true;
    nmax = t.n[j] > nmax ? t.n[j] : nmax;
  }
  if (!out || !any) return ZS_EINVAL;
  ZS_LAUNCH(KID_SCALAR_OBJECTIVE, (k_scalar_objective<T>), dim3(1), dim3(nmax >= 1024 ? 1024 : (nmax > 64 ? 256 : 64)),
            (hipStream_t)stream, t, out, coef_out);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

#define ZS_TERMS(T) Terms<T> t = {{r0, r1, r2, r3, r4, r5}, {n0, n1, n2, n3, n4, n5}, {c0, c1, c2, c3, c4, c5}}
extern "C" int zs_scalar_objective_f32(const float* r0, int64_t n0, double c0, const float* r1, int64_t n1, double c1, const float* r2,
                                       int64_t n2, double c2, const float* r3, int64_t n3, double c3, const float* r4, int64_t n4,
                                       double c4, const float* r5, int64_t n5, double c5, float* out, float* coef_out, void* stream) {
  ZS_TERMS(float);
  return scalar_objective<float>(t, out, coef_out, stream);
}
extern "C" int zs_scalar_objective_f64(const double* r0, int64_t n0, double c0, const double* r1, int64_t n1, double c1,
                                       const double* r2, int64_t n2, double c2, const double* r3, int64_t n3, double c3,
                                       const double* r4, int64_t n4, double c4, const double* r5, int64_t n5, double c5, double* out,
                                       double* coef_out, void* stream) {
  ZS_TERMS(double);
  return scalar_objective<double>(t, out, coef_out, stream);
}
