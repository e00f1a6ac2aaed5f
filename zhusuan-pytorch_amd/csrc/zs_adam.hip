// A1: Adam update of up to 32 parameter tensors in one launch (include/zs_hip.h).
//
// The callers' optimizer is torch.optim.Adam(model.parameters(), lr) with its defaults (reference examples
// variational_autoencoder/vae_mnist.py:104, iwae.py:141, bayesian_neural_nets/bnn_vi.py:135).  PyTorch's fused
// multi-tensor Adam splits the tensors into 64 K-element chunks, one workgroup each: the 1.35 M parameters of the VAE / IWAE
// models are 21 workgroups on a 256-CU chip (43 us per step).  Here the tensors form ONE flat index space
// [starts[s], starts[s+1]) -- parameters and gradients are read where they live through a pointer table passed by value
// (the gradients of a data-parallel bucket are consecutive slices of one buffer, zhusuan/dataparallel.py; a single
// process's are one tensor per parameter), both moments are flat --, a thread owns four consecutive elements and finds
// their tensor by bisection, and the 1/world factor of the gradient mean is folded into the read.  HBM-bound: 28 bytes per
// parameter.
//
// The step count is device state (the call is hipGraph-capturable): every workgroup reads it first; the last workgroup
// to finish -- found with a ticket, as in zs_iw.hip -- writes the incremented value back, so no workgroup can see the
// new count early.
#include <stdlib.h>
#include "zs_common.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

template <typename T>
struct alignas(16) Vec4 { T v[4]; };

template <typename T>
struct Tensors {
  T* param[ZS_ADAM_MAX_TENSORS];
  const T* grad[ZS_ADAM_MAX_TENSORS];          // NULL: no gradient this step (read as zero)
  int64_t start[ZS_ADAM_MAX_TENSORS + 1];      // start[n_tensors] = n
  int n_tensors;
};
template <typename T>
__device__ __forceinline__ int tensor_of(const Tensors<T>& ts, int64_t i) {
  int lo = 0, hi = ts.n_tensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (ts.start[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <typename T, bool VEC>
__global__ __launch_bounds__(1024) void k_adam_step(const Tensors<T> ts, T* __restrict__ m, T* __restrict__ v,
                                                    int64_t* __restrict__ steps, unsigned* __restrict__ ticket, int64_t n, double lr,
                                                    double beta1, double beta2, double eps, double grad_scale,
                                                    const double* __restrict__ hyper) {
  // Per-TENSOR step counts (torch.optim.Adam keeps one per parameter and leaves a parameter without a gradient alone:
  // no moment decay, no movement, no step).  Bias corrections once per workgroup: thread 2s forms lr / (1 - beta1^t_s),
  // thread 2s + 1 forms 1 / sqrt(1 - beta2^t_s); beta^t by squaring (<= 62 double multiplies).  A workgroup has at least
  // 64 threads = 2 * ZS_ADAM_MAX_TENSORS (checked on the host).
  __shared__ T c_step[ZS_ADAM_MAX_TENSORS], c_isq[ZS_ADAM_MAX_TENSORS];
  const int64_t groups = (n + 3) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // VEC: every tensor starts at a multiple of 4 and is 16-byte aligned (checked on the host): one tensor per group of four
  // elements, one 16-byte load per operand.  The loads of a thread's NEXT group are in flight while it works on the current
  // one, and those of its FIRST group while the workgroup forms the bias corrections below (the step counts are a memory
  // round trip of their own: at the BNN's 1 503 parameters the kernel is that chain of round trips and nothing else).
  struct Group { Vec4<T> p, g, m, v; int s; int64_t off; bool live; };
  auto fetch = [&](int64_t gi) {
    Group q;
    q.s = tensor_of(ts, gi << 2);
    q.off = (gi << 2) - ts.start[q.s];
    q.live = ts.grad[q.s] != nullptr;                 // no gradient this step: the tensor is left alone
    const T* gp = q.live ? ts.grad[q.s] + q.off : m + (gi << 2);      // (an address that is valid either way: no branch between loads)
    q.p = *reinterpret_cast<const Vec4<T>*>(ts.param[q.s] + q.off);
    q.m = *reinterpret_cast<const Vec4<T>*>(m + (gi << 2));
    q.v = *reinterpret_cast<const Vec4<T>*>(v + (gi << 2));
    q.g = *reinterpret_cast<const Vec4<T>*>(gp);
    return q;
  };
  // element form (tensors of any length and alignment): four consecutive elements, each finds its own tensor
  struct Scalars { T p[4], g[4], m[4], v[4]; T* dst[4]; int s[4]; bool live[4]; };
  auto fetch_scalars = [&](int64_t gi) {
    Scalars q;
    const int64_t i0 = gi << 2;
    const int cnt = n - i0 < 4 ? (int)(n - i0) : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = i0 + (j < cnt ? j : 0);   // clamped: unconditional loads
      const int s = tensor_of(ts, i);
      const int64_t off = i - ts.start[s];
      q.s[j] = s;
      q.dst[j] = ts.param[s] + off;
      q.live[j] = j < cnt && ts.grad[s] != nullptr;
      const T* gp = ts.grad[s] ? ts.grad[s] + off : m + i;
      q.p[j] = *q.dst[j]; q.m[j] = m[i]; q.v[j] = v[i]; q.g[j] = *gp;
    }
    return q;
  };
  Group cur;
  Scalars curS;
  bool have = VEC && g < groups;
  int64_t gS = g;
  if (have) cur = fetch(g);
  if (!VEC && gS < groups) curS = fetch_scalars(gS);
  if (hyper) { lr = hyper[0]; beta1 = hyper[1]; beta2 = hyper[2]; eps = hyper[3]; }     // device-resident: graph replays see updates
  if ((int)threadIdx.x < 2 * ts.n_tensors) {
    const int s = threadIdx.x >> 1;
    int64_t t = steps[s] + 1;
    double b = (threadIdx.x & 1) ? beta2 : beta1, pw = 1.0;
    for (; t > 0; t >>= 1, b *= b)
      if (t & 1) pw *= b;
    if (threadIdx.x & 1) c_isq[s] = (T)(1.0 / sqrt(1.0 - pw));
    else c_step[s] = (T)(lr / (1.0 - pw));
  }
  __syncthreads();
  const T b1w = (T)(1.0 - beta1), b2 = (T)beta2, b2w = (T)(1.0 - beta2), e = (T)eps, gs = (T)grad_scale;
  auto update = [&](T& pp, T& mm, T& vv, T gg, T ss, T iq) {
    const T gr = gs * gg;
    mm = mm + b1w * (gr - mm);                               // exp_avg.lerp_(grad, 1 - beta1)
    vv = b2 * vv + b2w * (gr * gr);                          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const T denom = sqrt(vv) * iq + e;                       // sqrt(v) / sqrt(bc2) + eps
    pp = pp - ss * (mm / denom);                             // param.addcdiv_(exp_avg, denom, -lr / bc1)
  };
  if (VEC) {
    while (have) {
      const int64_t g2 = g + stride;
      const bool have2 = g2 < groups;
      Group nxt;
      if (have2) nxt = fetch(g2);
      if (cur.live) {
        const T ss = c_step[cur.s], iq = c_isq[cur.s];
#pragma unroll
        for (int j = 0; j < 4; ++j) update(cur.p.v[j], cur.m.v[j], cur.v.v[j], cur.g.v[j], ss, iq);
        *reinterpret_cast<Vec4<T>*>(ts.param[cur.s] + cur.off) = cur.p;
        *reinterpret_cast<Vec4<T>*>(m + (g << 2)) = cur.m;
        *reinterpret_cast<Vec4<T>*>(v + (g << 2)) = cur.v;
      }
      if (have2) cur = nxt;
      g = g2;
      have = have2;
    }
  } else {
    bool haveS = gS < groups;
    while (haveS) {
      const int64_t i0 = gS << 2, g2 = gS + stride;
      const bool have2 = g2 < groups;
      Scalars nxt;
      if (have2) nxt = fetch_scalars(g2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        T pp = curS.p[j], mm = curS.m[j], vv = curS.v[j];
        update(pp, mm, vv, curS.g[j], c_step[curS.s[j]], c_isq[curS.s[j]]);
        if (curS.live[j]) { *curS.dst[j] = pp; m[i0 + j] = mm; v[i0 + j] = vv; }
      }
      if (have2) curS = nxt;
      gS = g2;
      haveS = have2;
    }
  }
  // Every workgroup has read the step counts -- the values have come back and gone into LDS -- before it takes its
  // ticket, and the last one to take a ticket publishes the new counts.  Relaxed atomics on purpose: an agent-scope
  // release here is an L2 write-back per workgroup (the XCDs' L2s are not coherent with each other): 56 instead of 20 us
  // for a 1 315-workgroup launch over the VAE / IWAE parameters; nothing but the ticket itself is communicated between
  // workgroups.
  __syncthreads();
  __shared__ unsigned last;
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, ZS_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (last) {
    if ((int)threadIdx.x < ts.n_tensors && ts.grad[threadIdx.x]) steps[threadIdx.x] += 1;
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T>
int adam_step(T* const* param_ptrs, const T* const* grad_ptrs, const int64_t* starts, int n_tensors, T* m, T* v, int64_t* step,
              uint32_t* ticket, int64_t n, double lr, double beta1, double beta2, double eps, double grad_scale,
              const double* hyper, void* stream) {
  if (n < 0 || !(lr >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return ZS_EINVAL;
  if (n_tensors < 1 || !param_ptrs || !grad_ptrs || !starts) return ZS_EINVAL;
  if (n_tensors > ZS_ADAM_MAX_TENSORS) return ZS_ENOTSUP;
  if (n == 0) return 0;
  if (!m || !v || !step || !ticket) return ZS_EINVAL;
  if (starts[0] != 0 || starts[n_tensors] != n) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  bool vec = (n & 3) == 0 && !(((uintptr_t)m | (uintptr_t)v) & (A - 1));
  Tensors<T> ts;
  memset(&ts, 0, sizeof(ts));
  ts.n_tensors = n_tensors;
  for (int i = 0; i < n_tensors; ++i) {
    if (starts[i + 1] <= starts[i] || !param_ptrs[i]) return ZS_EINVAL;       // non-empty, ascending
    ts.param[i] = param_ptrs[i];
    ts.grad[i] = grad_ptrs[i];
    ts.start[i] = starts[i];
    vec = vec && (starts[i] & 3) == 0 && !(((uintptr_t)param_ptrs[i] | (uintptr_t)grad_ptrs[i]) & (A - 1));
  }
  ts.start[n_tensors] = n;
  // few, fat workgroups: every workgroup ends with an atomic on the one ticket word, and same-address atomics retire at
  // 6-10 ns each.  Measured for the 1.35 M parameters of the VAE / IWAE models (37.7 MB, one flat tensor): 1 315 workgroups
  // of 256 threads 19.3 us, 512 x 256: 11.1, 512 x 1024: 8.6, 256 x 1024 (one per CU): 7.6-7.9 us = 60 % of the HBM
  // roofline; 5.2 M parameters: 80 %; 168 M parameters: 1024 x 1024 73 % against 67 % with 256.
  static const int grid_env = env_knob("ZS_ADAM_GRID", 0);      // experiments only (zs_common.h)
  static const int block_env = env_knob("ZS_ADAM_BLOCK", 0);
  const unsigned block = block_env >= 64 ? (unsigned)block_env : 1024u;      // >= 2 * ZS_ADAM_MAX_TENSORS threads (bias corrections)
  const unsigned grid = grid_for((n + 3) / 4, (int)block, grid_env > 0 ? (unsigned)grid_env : (n > (int64_t(1) << 24) ? 1024u : 256u));
  if (vec)
    ZS_LAUNCH(KID_ADAM, (k_adam_step<T, true>), dim3(grid), dim3(block), (hipStream_t)stream, ts, m, v, step, (unsigned*)ticket, n,
              lr, beta1, beta2, eps, grad_scale, hyper);
  else
    ZS_LAUNCH(KID_ADAM, (k_adam_step<T, false>), dim3(grid), dim3(block), (hipStream_t)stream, ts, m, v, step, (unsigned*)ticket, n,
              lr, beta1, beta2, eps, grad_scale, hyper);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int zs_adam_step_f32(float* const* param_ptrs, const float* const* grad_ptrs, const int64_t* starts, int n_tensors,
                                float* exp_avg, float* exp_avg_sq, int64_t* step, uint32_t* ticket, int64_t n, double lr,
                                double beta1, double beta2, double eps, double grad_scale, const double* hyper, void* stream) {
  return adam_step<float>(param_ptrs, grad_ptrs, starts, n_tensors, exp_avg, exp_avg_sq, step, ticket, n, lr, beta1, beta2, eps,
                          grad_scale, hyper, stream);
}
extern "C" int zs_adam_step_f64(double* const* param_ptrs, const double* const* grad_ptrs, const int64_t* starts, int n_tensors,
                                double* exp_avg, double* exp_avg_sq, int64_t* step, uint32_t* ticket, int64_t n, double lr,
                                double beta1, double beta2, double eps, double grad_scale, const double* hyper, void* stream) {
  return adam_step<double>(param_ptrs, grad_ptrs, starts, n_tensors, exp_avg, exp_avg_sq, step, ticket, n, lr, beta1, beta2, eps,
                           grad_scale, hyper, stream);
}
