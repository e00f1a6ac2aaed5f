// Importance-weighted reductions (K4) and RNG utilities for gfx950.
//
// Layout: log-joint rows are K-fastest ([B, ld], K particles of one datapoint contiguous), so for
// K <= 64 one wavefront owns one datapoint with lane == particle: max / sum / arg-max are
// __shfl_xor butterflies and nothing touches LDS.  For K > 64 one 256-thread workgroup owns a
// datapoint, lanes stride over K and the per-wave partials are combined through LDS.
//
// VIMCO (importance_weighted_objective.py:152-191) needs, for every particle j, the
// log-mean-exp of the row with entry j replaced by the mean of the others.  The reference
// materialises a [B, K, K] tensor; here it is O(K) per row:
//   j != argmax : max stays m1, sum = (S - e_j) + exp(sub_j - m1)      (S - e_j >= 1: no cancellation)
//   j == argmax : max becomes m2 (second largest), sum = S2 + exp(sub_j - m2) with
//                 S2 = sum_{i != j} exp(l_i - m2) accumulated separately (exact, no subtraction).
#include <stdlib.h>
#include "zs_common.h"
#include "zs_iw_math.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

// Extras of zs_iw_objective (all zero / NULL for zs_iw_reduce): second log-joint term, coefficient scale and the
// deterministic batch mean (per-workgroup partials + last-workgroup-done ticket).
struct IwExt {
  const float* logp_b;
  int64_t ld_b;
  float scale;
  float* mean_cost;
  float* partials;
  unsigned* ticket;
  float inv_B;
};

// Called by every thread of a workgroup after its rows are done: `block_cost` = the workgroup's cost sum (valid in
// thread 0).  The last workgroup to arrive adds the partials in index order with one wavefront.
__device__ __forceinline__ void iw_finish_mean(const IwExt& ext, float block_cost) {
  __shared__ bool last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(ext.partials + blockIdx.x, block_cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release: the partial above is visible to whoever observes this increment; acquire: the last arrival sees all
    const unsigned t = __hip_atomic_fetch_add(ext.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    float s = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 64)
      s += __hip_atomic_load(ext.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written by other workgroups
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      ext.mean_cost[0] = s * ext.inv_B;
      // hand the ticket back at zero for the next launch that uses this workspace (atomic, agent scope: the next
      // launch may run on another CU / XCD)
      __hip_atomic_store(ext.ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- K <= 64: one wave per datapoint, lane = particle
__global__ __launch_bounds__(256) void k_iw_reduce_wave(
    const float* __restrict__ logp, int64_t ld_p, const float* __restrict__ logq, int64_t ld_q,
    int64_t B, int K, int estimator, float* __restrict__ cost_b, float* __restrict__ bound_b,
    float* __restrict__ coef_p, float* __restrict__ coef_q, IwExt ext) {
  __shared__ float wave_cost[4];
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const bool on = lane < K;
  float my_cost = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < B; b += nwaves) {
    float lq = 0.f, l = -INFINITY;
    if (on) {
      // three loads issued together (a branch in front of the optional third would make it wait for the first two: one
      // more cold-cache round trip inside the step); without a second term the first is simply read twice
      const float* __restrict__ second = ext.logp_b ? ext.logp_b + b * ext.ld_b : logp + b * ld_p;
      lq = logq[b * ld_q + lane];
      float lp = logp[b * ld_p + lane];
      const float lb = second[lane];
      if (ext.logp_b) lp += lb;                                   // (a + b) - q, rounded like the reference's add
      l = lp - lq;
    }
    my_cost += iw_wave_row(l, lq, on, lane, K, estimator, ext.scale, b, cost_b, bound_b, coef_p, coef_q);
  }
  if (ext.mean_cost) {
    if (lane == 0) wave_cost[threadIdx.x >> 6] = my_cost;
    __syncthreads();
    iw_finish_mean(ext, (wave_cost[0] + wave_cost[1]) + (wave_cost[2] + wave_cost[3]));
  }
}

// ---- K <= 64, many datapoints: LPR (8 / 16) lanes per datapoint, lane q owns particles q, q + LPR, ... (NI of them, in
// registers), 64 / LPR datapoints per wave.  The wave-per-datapoint kernel above spends most of its time in six 64-lane
// butterflies per row (ds_bpermute) with 50 of 64 lanes holding a particle; here a reduction is 2-4 DPP instructions
// (quad_perm, row_half_mirror, row_mirror: plain VALU, no LDS) after NI local steps, and every lane slot but
// LPR * NI - K per row is a particle.  Needs B * LPR / 64 waves to fill the chip: the host picks LPR from B.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
#define ZS_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define ZS_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define ZS_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within 8
#define ZS_DPP_MIRROR 0x140       // lane i <-> 15 - i within 16
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += dpp_f<ZS_DPP_XOR1>(v);
  v += dpp_f<ZS_DPP_XOR2>(v);
  if (LPR >= 8) v += dpp_f<ZS_DPP_HALF_MIRROR>(v);
  if (LPR >= 16) v += dpp_f<ZS_DPP_MIRROR>(v);
  return v;
}
template <int LPR>
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, dpp_f<ZS_DPP_XOR1>(v));
  v = fmaxf(v, dpp_f<ZS_DPP_XOR2>(v));
  if (LPR >= 8) v = fmaxf(v, dpp_f<ZS_DPP_HALF_MIRROR>(v));
  if (LPR >= 16) v = fmaxf(v, dpp_f<ZS_DPP_MIRROR>(v));
  return v;
}
template <int LPR>
__device__ __forceinline__ int group_min(int v) {
  int t;
  t = dpp_i<ZS_DPP_XOR1>(v); v = t < v ? t : v;
  t = dpp_i<ZS_DPP_XOR2>(v); v = t < v ? t : v;
  if (LPR >= 8) { t = dpp_i<ZS_DPP_HALF_MIRROR>(v); v = t < v ? t : v; }
  if (LPR >= 16) { t = dpp_i<ZS_DPP_MIRROR>(v); v = t < v ? t : v; }
  return v;
}

template <int LPR, int NI>
__global__ __launch_bounds__(256) void k_iw_reduce_group(
    const float* __restrict__ logp, int64_t ld_p, const float* __restrict__ logq, int64_t ld_q,
    int64_t B, int K, int estimator, float* __restrict__ cost_b, float* __restrict__ bound_b,
    float* __restrict__ coef_p, float* __restrict__ coef_q, IwExt ext) {
  __shared__ float wave_cost[4];
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int rw = lane / LPR, q = lane % LPR;
  const int64_t items = (B + RPW - 1) / RPW;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float my_cost = 0.f;
  for (int64_t it = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); it < items; it += nwaves) {
    const int64_t b = it * RPW + rw;
    const bool row_on = b < B;
    const int64_t bc = row_on ? b : B - 1;                   // clamped: every lane loads and takes part in the reductions
    float l[NI], lq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = q + i * LPR, kc = k < K ? k : 0;
      lq[i] = logq[bc * ld_q + kc];
      l[i] = logp[bc * ld_p + kc];
    }
    if (ext.logp_b) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int k = q + i * LPR, kc = k < K ? k : 0;
        l[i] += ext.logp_b[bc * ext.ld_b + kc];                // (a + b) - q, rounded like the reference's add
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool on = q + i * LPR < K;
      l[i] = on ? l[i] - lq[i] : -INFINITY;
      mx = fmaxf(mx, l[i]);
    }
    IwRow r;
    r.m1 = group_max<LPR>(mx);
    int jl = 0x7fffffff;
#pragma unroll
    for (int i = NI - 1; i >= 0; --i)
      if (q + i * LPR < K && l[i] == r.m1) jl = q + i * LPR;    // lowest particle index holding the maximum
    jl = group_min<LPR>(jl);
    r.jstar = jl == 0x7fffffff ? 0 : jl;
    float m2 = -INFINITY, sl = 0.f, s = 0.f, e[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = q + i * LPR;
      const bool on = k < K;
      if (k != r.jstar) m2 = fmaxf(m2, l[i]);
      sl += on ? l[i] : 0.f;
      e[i] = on ? exp_fast(l[i] - r.m1) : 0.f;          // (v_exp_f32: see iw_particle_fast)
      s += e[i];
    }
    r.m2 = group_max<LPR>(m2);
    r.sumL = group_sum<LPR>(sl);
    r.S = group_sum<LPR>(s);
    r.S2 = 0.f;
    r.logS = 0.f;
    if (estimator == ZS_IW_VIMCO && r.S < 2.0f) {             // the only rows whose arg-max particle reads S2 / logS
      float s2 = 0.f;                                          // (uniform within the lane group: the DPP partners are active)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int k = q + i * LPR;
        s2 += (k < K && k != r.jstar) ? expf(l[i] - r.m2) : 0.f;
      }
      r.S2 = group_sum<LPR>(s2);
      r.logS = logf(r.S);
    }
    r.invK = 1.0f / (float)K;
    r.invKm1 = K > 1 ? 1.0f / (float)(K - 1) : 0.f;
    const float invS = 1.0f / r.S;                           // one division per lane instead of two per particle
    float ct = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = q + i * LPR;
      if (k < K) {
        float wt, c1, cq;
        iw_particle_fast(r, invS, l[i], lq[i], e[i], k, estimator, wt, c1, cq);
        ct += c1;
        if (row_on) {
          if (coef_p) coef_p[b * K + k] = -wt * ext.scale;
          if (coef_q) coef_q[b * K + k] = cq * ext.scale;
        }
      }
    }
    const float cost = group_sum<LPR>(ct);
    if (row_on && q == 0) {
      my_cost += cost;
      if (cost_b) cost_b[b] = cost;
      if (bound_b) bound_b[b] = logf(r.S * r.invK) + r.m1;     // log(mean(exp(x - max))) + max, utils.py:18
    }
  }
  if (ext.mean_cost) {
    my_cost = wave_sum(my_cost);
    if (lane == 0) wave_cost[threadIdx.x >> 6] = my_cost;
    __syncthreads();
    iw_finish_mean(ext, (wave_cost[0] + wave_cost[1]) + (wave_cost[2] + wave_cost[3]));
  }
}

// ---- any K: one 256-thread workgroup per datapoint, LDS-staged partials
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ int block_min_int(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o, ZS_WAVE);
    v = t < v ? t : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int a = sh[0] < sh[1] ? sh[0] : sh[1], c = sh[2] < sh[3] ? sh[2] : sh[3];
  return a < c ? a : c;
}

__global__ __launch_bounds__(256) void k_iw_reduce_block(
    const float* __restrict__ logp, int64_t ld_p, const float* __restrict__ logq, int64_t ld_q,
    int64_t B, int64_t K, int estimator, float* __restrict__ cost_b, float* __restrict__ bound_b,
    float* __restrict__ coef_p, float* __restrict__ coef_q, IwExt ext) {
  __shared__ float shf[4];
  __shared__ int shi[4];
  float my_cost = 0.f;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const float* __restrict__ pp = logp + b * ld_p;
    const float* __restrict__ pb = ext.logp_b ? ext.logp_b + b * ext.ld_b : nullptr;
    const float* __restrict__ qq = logq + b * ld_q;
#define ZS_LW(k) ((pb ? pp[k] + pb[k] : pp[k]) - qq[k])
    // pass 1: max, sum of l
    float mx = -INFINITY, sl = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const float l = ZS_LW(k);
      mx = fmaxf(mx, l);
      sl += l;
    }
    IwRow r;
    r.m1 = block_max(mx, shf);
    r.sumL = block_sum(sl, shf);
    // pass 2: first arg-max, S
    int jm = 0x7fffffff;
    float s = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const float l = ZS_LW(k);
      if (l == r.m1 && (int)k < jm) jm = (int)k;
      s += expf(l - r.m1);
    }
    r.jstar = block_min_int(jm, shi);
    r.S = block_sum(s, shf);
    r.m2 = -INFINITY;
    r.S2 = 0.f;
    if (estimator == ZS_IW_VIMCO) {
      float m2 = -INFINITY;
      for (int64_t k = threadIdx.x; k < K; k += 256)
        if ((int)k != r.jstar) m2 = fmaxf(m2, ZS_LW(k));
      r.m2 = block_max(m2, shf);
      float s2 = 0.f;
      for (int64_t k = threadIdx.x; k < K; k += 256)
        if ((int)k != r.jstar) s2 += expf(ZS_LW(k) - r.m2);
      r.S2 = block_sum(s2, shf);
    }
    r.logS = logf(r.S);
    r.invK = 1.0f / (float)K;
    r.invKm1 = K > 1 ? 1.0f / (float)(K - 1) : 0.f;
    float ct = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const float lq = qq[k];
      const float l = (pb ? pp[k] + pb[k] : pp[k]) - lq;
      float wt, c1, cq;
      iw_particle(r, l, lq, (int)k, estimator, wt, c1, cq);
      ct += c1;
      if (coef_p) coef_p[b * K + k] = -wt * ext.scale;
      if (coef_q) coef_q[b * K + k] = cq * ext.scale;
    }
    const float cost = block_sum(ct, shf);
    my_cost += cost;
    if (threadIdx.x == 0) {
      if (cost_b) cost_b[b] = cost;
      if (bound_b) bound_b[b] = logf(r.S * r.invK) + r.m1;
    }
  }
#undef ZS_LW
  if (ext.mean_cost) iw_finish_mean(ext, my_cost);
}

__global__ __launch_bounds__(256) void k_philox_normal(float* __restrict__ out, int64_t N, uint64_t seed,
                                                       uint64_t call, const uint64_t* __restrict__ rs) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  const int64_t groups = (N + 3) / 4;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
    const float4 n = philox_normal4((uint64_t)g, call, seed);
    const int64_t i = g * 4;
    if (i + 3 < N && ((((uintptr_t)out) & 15u) == 0)) {
      reinterpret_cast<float4*>(out)[g] = n;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < N) out[i + j] = f4_get(n, j);
    }
  }
}

}  // namespace

static int iw_launch(int kid, const float* logp, int64_t ld_p, const float* logq, int64_t ld_q, int64_t B, int64_t K, int estimator,
                     float* cost_b, float* bound_b, float* coef_p, float* coef_q, const IwExt& ext, int64_t workspace_len,
                     void* stream) {
  if (B < 0 || K < 1 || ld_p < K || ld_q < K) return ZS_EINVAL;
  if (estimator != ZS_IW_SGVB && estimator != ZS_IW_VIMCO) return ZS_EINVAL;
  if (estimator == ZS_IW_VIMCO && K < 2) return ZS_EINVAL;
  if (K > 0x7fffffff) return ZS_ENOTSUP;
  if (ext.logp_b && ext.ld_b < K) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!logp || !logq) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // K <= 64 and enough datapoints to fill the chip with fewer lanes per datapoint: the lane-group kernel
  static const int lpr_env = env_knob("ZS_K4_LPR", -1);      // experiments only (zs_common.h)
  // measured at K = 50 (tools/kernel_sweep.py, VIMCO): B = 20 971: wave kernel 14.3 us, 16 lanes 11.8, 8 lanes 11.1, 4 lanes
  // 13.9; B = 83 886: 46.6 / 33.5 / 31.2 / 33.6 us (the precise exp / log1p / divisions of the particles set the pace from
  // there); B = 2 621: launch floor either way
  int lpr = K > 64 || B < 4096 ? 0 : (B < 16384 ? 16 : 8);
  if (lpr_env >= 0 && K <= 64) lpr = lpr_env;
  if (lpr == 8 || lpr == 16) {
    const int rpw = 64 / lpr;
    const unsigned grid = grid_for((B + rpw - 1) / rpw, 4);
    if (ext.mean_cost && (!ext.partials || !ext.ticket || workspace_len < (int64_t)grid)) return ZS_EINVAL;
    const int ni = (int)((K + lpr - 1) / lpr);
#define ZS_IW_GROUP(L, N)                                                                                                    \
  ZS_LAUNCH(kid, (k_iw_reduce_group<L, N>), dim3(grid), dim3(256), st, logp, ld_p, logq, ld_q, B, (int)K, estimator, cost_b, \
            bound_b, coef_p, coef_q, ext)
    if (lpr == 8) {
      if (ni <= 2) ZS_IW_GROUP(8, 2); else if (ni <= 4) ZS_IW_GROUP(8, 4); else if (ni <= 7) ZS_IW_GROUP(8, 7); else ZS_IW_GROUP(8, 8);
    } else {
      if (ni <= 1) ZS_IW_GROUP(16, 1); else if (ni <= 2) ZS_IW_GROUP(16, 2); else ZS_IW_GROUP(16, 4);
    }
#undef ZS_IW_GROUP
    ZS_CHECK_LAUNCH();
    return 0;
  }
  const unsigned grid = K <= 64 ? grid_for(B, 4) : grid_for(B, 1);
  if (ext.mean_cost && (!ext.partials || !ext.ticket || workspace_len < (int64_t)grid)) return ZS_EINVAL;
  if (K <= 64)
    ZS_LAUNCH(kid, k_iw_reduce_wave, dim3(grid), dim3(256), st, logp, ld_p, logq, ld_q, B, (int)K, estimator, cost_b, bound_b, coef_p,
              coef_q, ext);
  else
    ZS_LAUNCH(kid, k_iw_reduce_block, dim3(grid), dim3(256), st, logp, ld_p, logq, ld_q, B, K, estimator, cost_b, bound_b, coef_p,
              coef_q, ext);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_iw_reduce_f32(const float* logp, int64_t ld_p, const float* logq, int64_t ld_q, int64_t B,
                                int64_t K, int estimator, float* cost_b, float* bound_b, float* coef_p,
                                float* coef_q, void* stream) {
  const IwExt ext = {nullptr, 0, 1.0f, nullptr, nullptr, nullptr, 0.f};
  return iw_launch(KID_IW_REDUCE, logp, ld_p, logq, ld_q, B, K, estimator, cost_b, bound_b, coef_p, coef_q, ext, 0, stream);
}

extern "C" int zs_iw_objective_f32(const float* logp_a, int64_t ld_a, const float* logp_b, int64_t ld_b, const float* logq,
                                   int64_t ld_q, int64_t B, int64_t K, int estimator, int want_mean, float* cost_b,
                                   float* bound_b, float* coef, float* mean_cost, float* workspace, int64_t workspace_len,
                                   uint32_t* ticket, void* stream) {
  if (want_mean && !mean_cost) return ZS_EINVAL;
  if (B < 0 || K < 1) return ZS_EINVAL;
  const float inv_B = B > 0 ? 1.0f / (float)B : 0.f;
  const IwExt ext = {logp_b, ld_b, want_mean ? inv_B : 1.0f, want_mean ? mean_cost : nullptr, workspace, ticket, inv_B};
  return iw_launch(KID_IW_OBJECTIVE, logp_a, ld_a, logq, ld_q, B, K, estimator, cost_b, bound_b, coef, coef ? coef + B * K : nullptr, ext,
                   workspace_len, stream);
}

// log_mean_exp over K-fastest rows: the IW reduction with logq == 0 and only the bound requested
namespace {
__global__ __launch_bounds__(256) void k_lme_block(const float* __restrict__ x, int64_t ld, int64_t B, int64_t K,
                                                   float* __restrict__ out) {
  __shared__ float shf[4];
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const float* __restrict__ xx = x + b * ld;
    float mx = -INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, xx[k]);
    const float m = block_max(mx, shf);
    float s = 0.f;
    for (int64_t k = threadIdx.x; k < K; k += 256) s += exp_fast(xx[k] - m);
    const float S = block_sum(s, shf);
    if (threadIdx.x == 0) out[b] = ln_fast(S / (float)K) + m;
  }
}
__global__ __launch_bounds__(256) void k_lme_wave(const float* __restrict__ x, int64_t ld, int64_t B, int K,
                                                  float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < B; b += nwaves) {
    const float l = lane < K ? x[b * ld + lane] : -INFINITY;
    const float m = wave_max(l);
    const float S = wave_sum(lane < K ? exp_fast(l - m) : 0.f);
    if (lane == 0) out[b] = ln_fast(S / (float)K) + m;
  }
}
}  // namespace

extern "C" int zs_log_mean_exp_f32(const float* x, int64_t ld, int64_t B, int64_t K, float* out, void* stream) {
  if (B < 0 || K < 1 || ld < K) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!x || !out) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (K <= 64)
    ZS_LAUNCH(KID_LME, k_lme_wave, dim3(grid_for(B, 4)), dim3(256), st, x, ld, B, (int)K, out);
  else
    ZS_LAUNCH(KID_LME, k_lme_block, dim3(grid_for(B, 1)), dim3(256), st, x, ld, B, K, out);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_philox_normal_f32(float* out, int64_t N, uint64_t seed, uint64_t offset,
                                    const uint64_t* rng_state, void* stream) {
  if (N < 0) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!out) return ZS_EINVAL;
  ZS_LAUNCH(KID_PHILOX, k_philox_normal, dim3(grid_for((N + 3) / 4, 256)), dim3(256), (hipStream_t)stream, out,
                     N, seed, offset, rng_state);
  ZS_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ per-kernel timing (bench.py)
#include <mutex>
#include <string.h>
#include <vector>
namespace {
struct ProfState {
  bool on = false;
  std::mutex mu;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[zs::KID_COUNT];
};
ProfState& prof_state() {
  static ProfState s;
  return s;
}
const char* const kKernelNames[zs::KID_COUNT] = {
    "zs_normal_sample_logprob_f32", "zs_normal_sample_logprob_bwd_f32", "zs_normal_logprob_f32",
    "zs_normal_logprob_bwd_f32", "zs_normal_logprob_bwd_ksum_f32", "zs_bernoulli_logprob_f32",
    "zs_bernoulli_logprob_bwd_f32", "zs_bernoulli_logits_logprob_f32", "zs_bernoulli_logits_logprob_bwd_f32",
    "zs_bernoulli_sample_f32", "zs_iw_reduce_f32", "zs_log_mean_exp_f32", "zs_philox_normal_f32",
    "zs_logistic_sample_logprob_f32", "zs_logistic_sample_logprob_bwd_f32", "zs_logistic_logprob_f32",
    "zs_logistic_logprob_bwd_f32", "zs_uniform_sample_f32", "zs_uniform_logprob_f32", "zs_philox_uniform_f32",
    "zs_reinforce_f32", "zs_iw_objective_f32", "zs_scalar_objective_f32", "zs_adam_step_f32",
    "zs_logistic_logprob_bwd_ksum_f32", "zs_logjoint_scalar_f32", "zs_logjoint_scalar_bwd_f32",
    "zs_normal_sample_logprob_multi_f32", "zs_normal_sample_logprob_multi_bwd_f32", "zs_particle_linear_f32",
    "zs_particle_linear_bwd_f32", "zs_column_sum_f32", "zs_dense_act_bwd_f32", "zs_particle_rmse_f32", "zs_particle_mlp_f32",
    "zs_particle_mlp_bwd_f32", "zs_bernoulli_iw_objective_f32", "zs_bernoulli_iw_objective_bwd_f32",
    "zs_normal_sample_logprob_pair_f32", "zs_bernoulli_logprob_bwd_x_f32"};
void prof_clear(ProfState& s) {
  for (int k = 0; k < zs::KID_COUNT; ++k) {
    for (auto& p : s.ev[k]) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
    s.ev[k].clear();
  }
}
}  // namespace

bool zs::prof_begin_launch(int kid, hipEvent_t* start, hipEvent_t* stop) {
  ProfState& s = prof_state();
  if (!s.on || kid < 0 || kid >= zs::KID_COUNT) return false;
  std::lock_guard<std::mutex> g(s.mu);
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return false;
  if (hipEventCreate(&b) != hipSuccess) {
    (void)hipEventDestroy(a);
    return false;
  }
  s.ev[kid].push_back(std::make_pair(a, b));
  *start = a;
  *stop = b;
  return true;
}

extern "C" int zs_prof_enable(int on) {
  ProfState& s = prof_state();
  std::lock_guard<std::mutex> g(s.mu);
  if (on) prof_clear(s);
  s.on = on != 0;
  return 0;
}

extern "C" int zs_prof_kernel_id(const char* entry_point) {
  if (!entry_point) return ZS_EINVAL;
  const size_t n = strlen(entry_point);
  if (n < 5) return ZS_EINVAL;
  for (int k = 0; k < zs::KID_COUNT; ++k) {   // the _f32 and _f64 forms of an entry point share one id
    const char* name = kKernelNames[k];
    if (strlen(name) == n && strncmp(entry_point, name, n - 4) == 0 &&
        (strcmp(entry_point + n - 4, "_f32") == 0 || strcmp(entry_point + n - 4, "_f64") == 0))
      return k;
  }
  return ZS_EINVAL;
}

extern "C" int zs_prof_query(int kernel_id, double* total_ms, double* min_ms, double* max_ms, int64_t* count) {
  ProfState& s = prof_state();
  if (kernel_id < 0 || kernel_id >= zs::KID_COUNT) return ZS_EINVAL;
  std::lock_guard<std::mutex> g(s.mu);
  double tot = 0.0, mn = 1e30, mx = 0.0;
  int64_t n = 0;
  for (auto& p : s.ev[kernel_id]) {
    hipError_t e = hipEventSynchronize(p.second);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p.first, p.second);
    if (e != hipSuccess) return (int)e;
    tot += ms;
    mn = ms < mn ? ms : mn;
    mx = ms > mx ? ms : mx;
    ++n;
  }
  if (total_ms) *total_ms = tot;
  if (min_ms) *min_ms = n ? mn : 0.0;
  if (max_ms) *max_ms = mx;
  if (count) *count = n;
  return 0;
}

extern "C" int64_t zs_prof_durations(int kernel_id, double* out_ms, int64_t capacity) {
  ProfState& s = prof_state();
  if (kernel_id < 0 || kernel_id >= zs::KID_COUNT || capacity < 0 || (capacity > 0 && !out_ms)) return ZS_EINVAL;
  std::lock_guard<std::mutex> g(s.mu);
  int64_t n = 0;
  for (auto& p : s.ev[kernel_id]) {
    if (n < capacity) {
      if (hipEventSynchronize(p.second) != hipSuccess) return ZS_EINVAL;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.first, p.second) != hipSuccess) return ZS_EINVAL;
      out_ms[n] = ms;
    }
    ++n;
  }
  return n;
}

#define ZS_STR_(x) #x
#define ZS_STR(x) ZS_STR_(x)
extern "C" int zs_abi_version(void) { return ZS_ABI_VERSION; }
extern "C" const char* zs_build_info(void) { return "libzs_hip gfx950 ABI " ZS_STR(ZS_ABI_VERSION) ", " ZS_BUILD_KIND ZS_HANDOFF_KIND; }

extern "C" const char* zs_error_string(int code) {
  if (code == 0) return "success";
  if (code == ZS_EINVAL) return "zs: invalid argument (null pointer, non-dividing period, or bad size)";
  if (code == ZS_ENOTSUP) return "zs: unsupported configuration";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "zs: unknown error";
}
