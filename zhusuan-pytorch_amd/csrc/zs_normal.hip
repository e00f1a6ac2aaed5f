// Normal distribution kernels (K1, K2) for gfx950.  See include/zs_hip.h for the contract and
// the reference lines each entry point replaces.  HBM-bound streaming kernels: 16-byte
// per-lane accesses, G = min(D/4, 64) lanes cooperate on one row of D elements, row sums by
// wavefront shuffles, no LDS except for the cross-wave K-slice reduction of the backward.
#include "zs_common.h"
#include "zs_sample_tile.h"
#include "../../include/zs_hip.h"
#include <stdlib.h>


using namespace zs;

namespace {

// ------------------------------------------------------------------------------------
// K1 forward, rows of up to 256 elements (D4 <= 64): a wave owns `rpw` parameter rows (G = D4 lanes
// each) and walks a chunk of the K particles, so log(sigma) and sigma^-2 are computed once per lane and
// reused for every particle.  z is written 16 B per lane.  Row sums are LDS-staged: every lane parks its
// 4-element partial in LDS each particle; after KB particles the wave reads the partials back transposed
// (lane = (row, particle)), adds the G partials of a row and writes log q coalesced along the particle
// axis of the K-fastest result.  (A per-particle __shfl_down tree over the 10-lane groups costs 26 % of
// the kernel: tools/k1_variants.hip.)  NT: non-temporal stores of z for tensors that cannot stay in the
// 256 MB Infinity Cache (streaming-write rate 4.1 -> 5.3 TB/s at 0.7 GB).
// ------------------------------------------------------------------------------------
#define ZS_K1_KB 32
#define ZS_K1_LDW 65

template <bool HAS_EPS, bool HAS_LP, bool NT>
__global__ __launch_bounds__(256) void k_normal_sample_smallrow(
    const float4* __restrict__ mu, const float4* __restrict__ sigma, const float4* __restrict__ eps,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs, float4* __restrict__ z, float* __restrict__ lp,
    int64_t K, int64_t R, int D4, int G, int rpw, int64_t kchunk, int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used) {
  __shared__ float stage[4][ZS_K1_KB * ZS_K1_LDW];
  float* __restrict__ st = stage[threadIdx.x >> 6];
  if (rs) { seed = rs[0]; call += rs[1]; }
  publish_rng(rng_used, seed, call);
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw;
  const int64_t k_tiles = (K + kchunk - 1) / kchunk;
  const int64_t total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < total; t += nwaves) {
    int64_t kt, rt;
    divmod(t, row_tiles, kt, rt);
    const int64_t rbase = rt * rpw;
    const int64_t r = rbase + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = r * D4 + lig;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s = make_float4(1.f, 1.f, 1.f, 1.f);
    if (on) {
      m = mu[m4];
      s = sigma_of(sigma[m4], ls);
    }
    // per-lane row constants, computed once and reused for every particle of the chunk:
    // rowc = sum_j (c - log sigma_j)  (normal.py:121-124),  hp_j = 0.5 * exp(-2 log sigma_j)
    float rowc = 0.f, hp[4];
    {
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float l2 = log2_fast(sv[j]);
        rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2;
        hp[j] = 0.5f * exp2_fast(-2.0f * l2);
      }
    }
    const int64_t k0 = kt * kchunk;
    const int64_t k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;                      // running address: no 64-bit multiplies in the loop
    for (int64_t kb0 = k0; kb0 < k1; kb0 += ZS_K1_KB) {
      const int kb = (int)((k1 - kb0 < ZS_K1_KB) ? (k1 - kb0) : ZS_K1_KB);
      for (int kk = 0; kk < kb; ++kk, g += M4) {
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
          if (HAS_EPS) e = eps[g];
          else e = philox_normal4((uint64_t)g, call, seed);
        }
        float4 zz;
        zz.x = mul_add_2round(m.x, s.x, e.x);
        zz.y = mul_add_2round(m.y, s.y, e.y);
        zz.z = mul_add_2round(m.z, s.z, e.z);
        zz.w = mul_add_2round(m.w, s.w, e.w);
        if (on) {
          if (NT) {
            const zs_f4v v = {zz.x, zz.y, zz.z, zz.w};
            __builtin_nontemporal_store(v, reinterpret_cast<zs_f4v*>(&z[g]));
          } else {
            z[g] = zz;
          }
        }
        if (HAS_LP) {
          const float d0 = zz.x - m.x, d1 = zz.y - m.y, d2 = zz.z - m.z, d3 = zz.w - m.w;
          st[kk * ZS_K1_LDW + lane] =
              rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3));
        }
      }
      if (HAS_LP) {
        // same-wave LDS hand-off (the slice is private to this wave, so no workgroup barrier): a wave-scope
        // release/acquire fence pair orders the ds_write's above before the ds_read's below for the memory model, and
        // wave_barrier keeps the compiler from moving either across it.  Both compile to no instructions beyond the
        // s_waitcnt lgkmcnt the reads need anyway.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nout = rpw * kb;
        for (int o = lane; o < nout; o += 64) {
          const int q = o / kb, kk = o - q * kb;
          const float* __restrict__ src = st + kk * ZS_K1_LDW + q * G;
          float sum = 0.f;
          for (int j = 0; j < G; ++j) sum += src[j];
          if (rbase + q < R) lp[(kb0 + kk) * sk + (rbase + q) * sr] = sum;
        }
        // ... and the reads before the next batch of particles overwrites the slice
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
}

// K1 forward, long rows (D4 > 64): WPR waves per (k, r) row, lanes stride over the row.  WPR = 1: one wave per row
// (many rows).  WPR = 4: the whole workgroup on one row -- for the few long rows of a weight-matrix node (BNN:
// K = 10 rows of 700 elements; with one wave per row each lane walked three dependent load -> generate -> store rounds:
// 6.1 us for 28 KB), partials combined through LDS.
template <bool HAS_EPS, bool HAS_LP, int WPR>
__global__ __launch_bounds__(256) void k_normal_sample_longrow(
    const float4* __restrict__ mu, const float4* __restrict__ sigma, const float4* __restrict__ eps,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs, float4* __restrict__ z, float* __restrict__ lp,
    int64_t K, int64_t R, int D4, int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used) {
  __shared__ float part[4];
  if (rs) { seed = rs[0]; call += rs[1]; }
  publish_rng(rng_used, seed, call);
  const int lane = threadIdx.x & 63;
  constexpr int TPR = 64 * WPR;                       // threads per row
  const int tin = WPR == 1 ? lane : (int)threadIdx.x;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t rows = K * R;
  const int64_t stride = (int64_t)gridDim.x * (4 / WPR);
  for (int64_t row = (int64_t)blockIdx.x * (4 / WPR) + (WPR == 1 ? (threadIdx.x >> 6) : 0); row < rows; row += stride) {
    int64_t k, r;
    divmod(row, R, k, r);
    float acc = 0.f;
    for (int c = tin; c < D4; c += TPR) {
      const int64_t m4 = r * D4 + c;
      const int64_t g = k * M4 + m4;
      const float4 m = mu[m4], s = sigma_of(sigma[m4], ls);
      float4 e;
      if (HAS_EPS) e = eps[g];
      else e = philox_normal4((uint64_t)g, call, seed);
      float4 zz;
      zz.x = mul_add_2round(m.x, s.x, e.x);
      zz.y = mul_add_2round(m.y, s.y, e.y);
      zz.z = mul_add_2round(m.z, s.z, e.z);
      zz.w = mul_add_2round(m.w, s.w, e.w);
      z[g] = zz;
      if (HAS_LP) {
        const float sv[4] = {s.x, s.y, s.z, s.w};
        const float dv[4] = {zz.x - m.x, zz.y - m.y, zz.z - m.z, zz.w - m.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float l2 = log2_fast(sv[j]);
          acc += normal_lp_term(dv[j], l2 * ZS_LN2, exp2_fast(-2.0f * l2));
        }
      }
    }
    if (HAS_LP) {
      acc = wave_sum(acc);
      if (WPR == 1) {
        if (lane == 0) lp[k * sk + r * sr] = acc;
      } else {                                           // `row` is uniform across the workgroup: barriers are safe
        if (lane == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) lp[k * sk + r * sr] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
      }
    }
  }
}

// K1 forward, any D / any alignment: one thread per (k, r) row, serial over the row.
template <bool HAS_EPS>
__global__ __launch_bounds__(256) void k_normal_sample_serial(
    const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ eps,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs, float* __restrict__ z, float* __restrict__ lp,
    int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  publish_rng(rng_used, seed, call);
  const int64_t rows = K * R, M = R * D;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    int64_t k, r;
    divmod(row, R, k, r);
    float acc = 0.f;
    float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t have = -1;
    for (int64_t d = 0; d < D; ++d) {
      const int64_t m = r * D + d, i = k * M + m;
      float e;
      if (HAS_EPS) {
        e = eps[i];
      } else {
        if ((i >> 2) != have) {
          have = i >> 2;
          n4 = philox_normal4((uint64_t)have, call, seed);
        }
        e = f4_get(n4, (int)(i & 3));
      }
      const float mm = mu[m], s = sigma_of(sigma[m], ls);
      const float zz = mul_add_2round(mm, s, e);
      z[i] = zz;
      if (lp) {
        float l2 = log2_fast(s);
        acc += normal_lp_term(zz - mm, l2 * ZS_LN2, exp2_fast(-2.0f * l2));
      }
    }
    if (lp) lp[k * sk + r * sr] = acc;
  }
}

// K1 forward for rows that cannot use 16-byte accesses (D % 4 != 0 or unaligned), D >= 8: one wave per
// (k, r) row, one element per lane and pass, wavefront row sum.  (BNN weight matrix [1, 51]: the per-thread
// serial kernel above took 21 us for ten rows of 51 elements.)
template <bool HAS_EPS>
__global__ __launch_bounds__(256) void k_normal_sample_waverow(
    const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ eps,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs, float* __restrict__ z, float* __restrict__ lp,
    int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  publish_rng(rng_used, seed, call);
  const int lane = threadIdx.x & 63;
  const int64_t rows = K * R, M = R * D;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += nwaves) {
    int64_t k, r;
    divmod(row, R, k, r);
    float acc = 0.f;
    for (int64_t d = lane; d < D; d += 64) {
      const int64_t m = r * D + d, i = k * M + m;
      float e;
      if (HAS_EPS) e = eps[i];
      else e = f4_get(philox_normal4((uint64_t)(i >> 2), call, seed), (int)(i & 3));
      const float mm = mu[m], s = sigma_of(sigma[m], ls);
      const float zz = mul_add_2round(mm, s, e);
      z[i] = zz;
      if (lp) {
        const float l2 = log2_fast(s);
        acc += normal_lp_term(zz - mm, l2 * ZS_LN2, exp2_fast(-2.0f * l2));
      }
    }
    if (lp) {
      acc = wave_sum(acc);
      if (lane == 0) lp[k * sk + r * sr] = acc;
    }
  }
}

// K2 forward, same fallback shape: one wave per row, element-wise periodic addressing.
__global__ __launch_bounds__(256) void k_normal_logprob_waverow(
    const float* __restrict__ x, int64_t Px, const float* __restrict__ mu, int64_t Pm,
    const float* __restrict__ sigma, int64_t Ps, float* __restrict__ lp,
    int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, bool ls) {
  const int lane = threadIdx.x & 63;
  const int64_t rows = K * R;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += nwaves) {
    float acc = 0.f;
    for (int64_t d = lane; d < D; d += 64) {
      const int64_t i = row * D + d;
      const float s = sigma_of(sigma[Ps == 1 ? 0 : mod_fast(i, Ps)], ls);
      const float l2 = log2_fast(s);
      acc += normal_lp_term(x[Px == 1 ? 0 : mod_fast(i, Px)] - mu[Pm == 1 ? 0 : mod_fast(i, Pm)], l2 * ZS_LN2,
                            exp2_fast(-2.0f * l2));
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      int64_t k, r;
      divmod(row, R, k, r);
      lp[k * sk + r * sr] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------
// K1 backward (reparameterised): thread tile = 64 parameter float4 groups x 4 K-slices,
// K-slice partials combined through LDS.
// ------------------------------------------------------------------------------------
template <bool HAS_EPS, bool HAS_GZ, bool HAS_GLP, int NS>
__global__ __launch_bounds__(64 * NS) void k_normal_sample_bwd(
    const float4* __restrict__ sigma, const float4* __restrict__ eps, uint64_t seed, uint64_t call,
    const uint64_t* __restrict__ rs, const float4* __restrict__ gz, const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float4* __restrict__ gmu, float4* __restrict__ gsigma, int64_t K, int64_t M4, int D4, bool ls) {
  __shared__ float4 red_a[NS][64];
  __shared__ float4 red_b[NS][64];
  __shared__ float red_g[NS][64];
  if (rs) { seed = rs[0]; call += rs[1]; }
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t m4 = (int64_t)blockIdx.x * 64 + lane;
  const bool on = m4 < M4;
  float4 am = make_float4(0.f, 0.f, 0.f, 0.f), as = am;
  float gl = 0.f;
  // The slice index is the same for the whole wavefront: with it in a scalar register the particle loop runs on scalar
  // counters, and the three streams (gz, eps, glp) and the Philox group index advance by ONE 64-bit add each per particle --
  // the index arithmetic of the per-lane form (k * M4 + m4 and kn * gsk + r * gsr with 64-bit multiplies, the clamp of the
  // prefetch index, the loop predicate) was 28 of the loop's 95 VALU instructions, and the kernel is bound by instruction issue.
  const int64_t ks = __builtin_amdgcn_readfirstlane(slice);
  if (on && ks < K) {
    const int64_t r = (int64_t)((uint64_t)m4 >> 31 ? m4 / D4 : (int64_t)((uint32_t)m4 / (uint32_t)D4));
    // rolling prefetch: the loads of particle k + NS are in flight while the draw of particle k is regenerated (which
    // operands exist is compiled in: a branch between two loads would make the second wait for the first)
    float4 gn = make_float4(0.f, 0.f, 0.f, 0.f), en = gn;
    float gln = 0.f;
    const float4* __restrict__ gzp = HAS_GZ ? gz + (ks * M4 + m4) : nullptr;
    const float4* __restrict__ epp = (HAS_GZ && HAS_EPS) ? eps + (ks * M4 + m4) : nullptr;
    const float* __restrict__ glpp = HAS_GLP ? glp + (ks * gsk + r * gsr) : nullptr;
    uint64_t g = (uint64_t)(ks * M4 + m4);                 // Philox group of (particle k, this lane)
    if (HAS_GZ) gn = *gzp;
    if (HAS_GZ && HAS_EPS) en = *epp;
    if (HAS_GLP) gln = *glpp;
    const int64_t zstep = (int64_t)NS * M4, lstep = (int64_t)NS * gsk;
    for (int64_t k = ks; k < K; k += NS) {                  // uniform: scalar registers
      const float4 gv = gn;
      float4 e = en;
      gl += gln;
      const bool more = k + NS < K;                         // last iteration: a harmless re-read of the same particle
      const int64_t zs = more ? zstep : 0, ls2 = more ? lstep : 0;
      if (HAS_GZ) { gzp += zs; gn = *gzp; }
      if (HAS_GZ && HAS_EPS) { epp += zs; en = *epp; }
      if (HAS_GLP) { glpp += ls2; gln = *glpp; }
      if (HAS_GZ) {
        if (!HAS_EPS) e = philox_normal4(g, call, seed);
        am.x += gv.x; am.y += gv.y; am.z += gv.z; am.w += gv.w;
        as.x += gv.x * e.x; as.y += gv.y * e.y; as.z += gv.z * e.z; as.w += gv.w * e.w;
      }
      g += (uint64_t)zstep;
    }
  }
  red_a[slice][lane] = am;
  red_b[slice][lane] = as;
  red_g[slice][lane] = gl;
  __syncthreads();
  if (slice == 0 && on) {
    float4 a = red_a[0][lane], b = red_b[0][lane];
    float g = red_g[0][lane];
#pragma unroll
    for (int s = 1; s < NS; ++s) {
      const float4 a2 = red_a[s][lane], b2 = red_b[s][lane];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      g += red_g[s][lane];
    }
    const float4 s = sigma_of(sigma[m4], ls);
    gmu[m4] = a;
    // d/d sigma = sum_k gz*eps - (sum_k glp)/sigma ;  d/d logstd = sigma * that = sigma * sum_k gz*eps - sum_k glp
    gsigma[m4] = ls ? make_float4(b.x * s.x - g, b.y * s.y - g, b.z * s.z - g, b.w * s.w - g)
                    : make_float4(b.x - g / s.x, b.y - g / s.y, b.z - g / s.z, b.w - g / s.w);
  }
}

// K1 backward for rows without 16-byte access (any D / alignment; the BNN's [1, 51] weight matrix): the same tiling as
// the vector kernel -- 64 elements x 4 K-slices per workgroup, slices combined through LDS -- with scalar accesses (a
// thread per element walking all K particles took 6.3 us for 51 elements x 10 particles: ten dependent rounds of load,
// generate, accumulate).
__global__ __launch_bounds__(256) void k_normal_sample_bwd_serial(
    const float* __restrict__ sigma, const float* __restrict__ eps, uint64_t seed, uint64_t call,
    const uint64_t* __restrict__ rs, const float* __restrict__ gz, const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float* __restrict__ gmu, float* __restrict__ gsigma, int64_t K, int64_t M, int64_t D, bool ls) {
  __shared__ float red[3][4][64];
  if (rs) { seed = rs[0]; call += rs[1]; }
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 64 + lane;
  const bool on = m < M;
  float a = 0.f, b = 0.f, g = 0.f;
  if (on) {
    const int64_t r = m / D;
    for (int64_t k = slice; k < K; k += 4) {
      const int64_t i = k * M + m;
      if (gz) {
        float e;
        if (eps) e = eps[i];
        else e = f4_get(philox_normal4((uint64_t)(i >> 2), call, seed), (int)(i & 3));
        const float gv = gz[i];
        a += gv;
        b += gv * e;
      }
      if (glp) g += glp[k * gsk + r * gsr];
    }
  }
  red[0][slice][lane] = a;
  red[1][slice][lane] = b;
  red[2][slice][lane] = g;
  __syncthreads();
  if (slice == 0 && on) {
    a = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    b = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
    g = (red[2][0][lane] + red[2][1][lane]) + (red[2][2][lane] + red[2][3][lane]);
    gmu[m] = a;
    const float sg = sigma_of(sigma[m], ls);
    gsigma[m] = ls ? b * sg - g : b - g / sg;
  }
}

// ------------------------------------------------------------------------------------
// K2 forward: log-prob of a given value, periodic operands.
// Row-period form: operand row index = row % prow (prow = period / D rows), or scalar (prow = 0).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld_row4(const float4* __restrict__ p, int64_t row, int64_t prow, int D4, int c) {
  if (prow == 0) {
    const float v = *reinterpret_cast<const float*>(p);
    return make_float4(v, v, v, v);
  }
  return p[mod_fast(row, prow) * D4 + c];
}

__global__ __launch_bounds__(256) void k_normal_logprob_rows(
    const float4* __restrict__ x, int64_t xr, const float4* __restrict__ mu, int64_t mr,
    const float4* __restrict__ sigma, int64_t sr_, float* __restrict__ lp,
    int64_t K, int64_t R, int D4, int G, int rpw, int p2, int64_t sk, int64_t sr, bool ls) {
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t rows = K * R;
  const int64_t tiles = (rows + rpw - 1) / rpw;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < tiles; t += nwaves) {
    const int64_t row = t * rpw + rw;
    const bool on = lane_on && row < rows;
    float acc = 0.f;
    if (on) {
      for (int c = lig; c < D4; c += G) {
        const float4 xv = ld_row4(x, row, xr, D4, c);
        const float4 m = ld_row4(mu, row, mr, D4, c);
        const float4 s = sigma_of(ld_row4(sigma, row, sr_, D4, c), ls);
        const float sv[4] = {s.x, s.y, s.z, s.w};
        const float dv[4] = {xv.x - m.x, xv.y - m.y, xv.z - m.z, xv.w - m.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float l2 = log2_fast(sv[j]);
          acc += normal_lp_term(dv[j], l2 * ZS_LN2, exp2_fast(-2.0f * l2));
        }
      }
    }
    acc = group_sum_down(acc, lig, G, p2);
    if (on && lig == 0) {
      int64_t k, r;
    divmod(row, R, k, r);
      lp[k * sk + r * sr] = acc;
    }
  }
}

// K2 forward, few long rows (D4 > 64, fewer rows than SIMDs -- a weight-matrix node's prior, K rows of 700 elements): the
// whole workgroup on one row, every lane one or two 16-byte pieces, partials combined through LDS (one wave per row
// walked three dependent rounds of loads: 6.3 us for 28 KB).
// MS / SS: mu / sigma given as one scalar (decided on the host: a run-time scalar-or-row branch between two loads makes
// the compiler wait for the first load before it issues the second, and the three loads of a piece went out one after
// the other -- 6.9 us; with the operand classes compiled in they go out together).
template <bool MS, bool SS>
__global__ __launch_bounds__(256) void k_normal_logprob_blockrow(
    const float4* __restrict__ x, int64_t xr, const float4* __restrict__ mu, int64_t mr,
    const float4* __restrict__ sigma, int64_t sr_, float* __restrict__ lp,
    int64_t K, int64_t R, int D4, int64_t sk, int64_t sr, bool ls) {
  __shared__ float part[4];
  const int64_t rows = K * R;
  float ms = 0.f, l2s = 0.f, ps = 1.f;
  if (MS) ms = *reinterpret_cast<const float*>(mu);
  if (SS) {
    l2s = log2_fast(sigma_of(*reinterpret_cast<const float*>(sigma), ls));
    ps = exp2_fast(-2.0f * l2s);
  }
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {       // uniform across the workgroup
    const float4* __restrict__ xrow = x + mod_fast(row, xr) * D4;
    const float4* __restrict__ mrow = MS ? nullptr : mu + mod_fast(row, mr) * D4;
    const float4* __restrict__ srow = SS ? nullptr : sigma + mod_fast(row, sr_) * D4;
    float acc = 0.f;
    for (int c = threadIdx.x; c < D4; c += 256) {
      const float4 xv = xrow[c];
      const float4 m = MS ? make_float4(ms, ms, ms, ms) : mrow[c];
      float4 s = make_float4(1.f, 1.f, 1.f, 1.f);
      if (!SS) s = sigma_of(srow[c], ls);
      const float sv[4] = {s.x, s.y, s.z, s.w};
      const float dv[4] = {xv.x - m.x, xv.y - m.y, xv.z - m.z, xv.w - m.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (SS) {
          acc += normal_lp_term(dv[j], l2s * ZS_LN2, ps);
        } else {
          const float l2 = log2_fast(sv[j]);
          acc += normal_lp_term(dv[j], l2 * ZS_LN2, exp2_fast(-2.0f * l2));
        }
      }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      int64_t k, r;
      divmod(row, R, k, r);
      lp[k * sk + r * sr] = (part[0] + part[1]) + (part[2] + part[3]);
    }
    __syncthreads();
  }
}

// K2 forward, every operand either full-size or a scalar: rows with no index arithmetic at all.
__global__ __launch_bounds__(256) void k_normal_logprob_full(
    const float4* __restrict__ x, int x_scalar, const float4* __restrict__ mu, int mu_scalar,
    const float4* __restrict__ sigma, int sg_scalar, float* __restrict__ lp,
    int64_t rows, int64_t R, int D4, int G, int rpw, int p2, int64_t sk, int64_t sr, bool ls) {
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t tiles = (rows + rpw - 1) / rpw;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  float xs = 0.f, ms = 0.f, ss = 1.f, ls_s = 0.f, hp_s = 0.5f;
  if (x_scalar) xs = *reinterpret_cast<const float*>(x);
  if (mu_scalar) ms = *reinterpret_cast<const float*>(mu);
  if (sg_scalar) {
    ss = sigma_of(*reinterpret_cast<const float*>(sigma), ls);
    const float l2 = log2_fast(ss);
    ls_s = l2 * ZS_LN2;
    hp_s = 0.5f * exp2_fast(-2.0f * l2);
  }
  for (int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < tiles; t += nwaves) {
    const int64_t row = t * rpw + rw;
    const bool on = lane_on && row < rows;
    float acc = 0.f;
    if (on) {
      const int64_t base = row * D4;
      for (int c = lig; c < D4; c += G) {
        const float4 xv = x_scalar ? make_float4(xs, xs, xs, xs) : x[base + c];
        const float4 mv = mu_scalar ? make_float4(ms, ms, ms, ms) : mu[base + c];
        const float dv[4] = {xv.x - mv.x, xv.y - mv.y, xv.z - mv.z, xv.w - mv.w};
        if (sg_scalar) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc += (ZS_NEG_HALF_LOG_2PI - ls_s) - hp_s * (dv[j] * dv[j]);
        } else {
          const float4 sv4 = sigma_of(sigma[base + c], ls);
          const float sv[4] = {sv4.x, sv4.y, sv4.z, sv4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float l2 = log2_fast(sv[j]);
            acc += normal_lp_term(dv[j], l2 * ZS_LN2, exp2_fast(-2.0f * l2));
          }
        }
      }
    }
    acc = group_sum_down(acc, lig, G, p2);
    if (on && lig == 0) {
      int64_t k, r;
    divmod(row, R, k, r);
      lp[k * sk + r * sr] = acc;
    }
  }
}

// any D / alignment / period: one thread per row, element-wise modulo addressing
__global__ __launch_bounds__(256) void k_normal_logprob_serial(
    const float* __restrict__ x, int64_t Px, const float* __restrict__ mu, int64_t Pm,
    const float* __restrict__ sigma, int64_t Ps, float* __restrict__ lp,
    int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, bool ls) {
  const int64_t rows = K * R;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows;
       row += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int64_t d = 0; d < D; ++d) {
      const int64_t i = row * D + d;
      const float s = sigma_of(sigma[mod_fast(i, Ps)], ls);
      const float l2 = log2_fast(s);
      acc += normal_lp_term(x[mod_fast(i, Px)] - mu[mod_fast(i, Pm)], l2 * ZS_LN2, exp2_fast(-2.0f * l2));
    }
    int64_t k, r;
    divmod(row, R, k, r);
    lp[k * sk + r * sr] = acc;
  }
}

// K2 backward, element-wise partials (any shape; scalar accesses, coalesced over i)
__global__ __launch_bounds__(256) void k_normal_logprob_bwd_elem(
    const float* __restrict__ x, int64_t Px, const float* __restrict__ mu, int64_t Pm,
    const float* __restrict__ sigma, int64_t Ps, const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float* __restrict__ gx, float* __restrict__ gmu, float* __restrict__ gsigma,
    int64_t N, int64_t R, int64_t D, bool ls) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / D;
    int64_t k, r;
    divmod(row, R, k, r);
    const float g = glp[k * gsk + r * gsr];
    const float s = sigma_of(sigma[mod_fast(i, Ps)], ls);
    const float diff = x[mod_fast(i, Px)] - mu[mod_fast(i, Pm)];
    const float prec = exp2_fast(-2.0f * log2_fast(s));
    const float t = g * prec * diff;
    if (gx) gx[i] = -t;
    if (gmu) gmu[i] = t;
    if (gsigma) gsigma[i] = ls ? g * (prec * diff * diff - 1.0f) : g * (prec * diff * diff - 1.0f) / s;
  }
}


inline bool period_ok_rows(int64_t P, int64_t D, int64_t N) {
  return P == 1 || (P >= D && P % D == 0 && N % P == 0);
}

}  // namespace

// ======================================================================== C ABI
extern "C" int zs_normal_sample_logprob_f32(const float* mu, const float* sigma, const float* eps,
                                            uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                            float* z, float* lp,
                                            int64_t K, int64_t M, int64_t D,
                                            int64_t sk, int64_t sr, int sigma_is_logstd, uint64_t* rng_used,
                                            void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!mu || !sigma || !z) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = M / D;
  const bool vec = (D % 4 == 0) && aligned16(mu) && aligned16(sigma) && aligned16(z) && (!eps || aligned16(eps));
  if (vec) {
    const int D4 = (int)(D / 4);
    const float4 *m4 = (const float4*)mu, *s4 = (const float4*)sigma, *e4 = (const float4*)eps;
    // eps handed in (the parity route): the flat-plane given-stream kernel in its SAMPLE mode (zs_sample_tile.h) -- against the
    // row-per-lane-group kernel below: 131 k rows 59 -> 68 %, 1 M rows 68 -> 70-71 %, 4.2 M rows 68 -> 73 % of the roofline
    static const int given_tile_env = env_knob("ZS_K1_GIVEN_TILE", 1);      // experiments only: 0 = the row-per-lane-group kernel
    if (eps && given_tile_env) {
      const K1Tile g = k1_tile(K, R, D4, true);
      if (g.ok) {
        const bool big = (double)K * (double)M * 8.0 > 268435456.0;
        if (big)
          ZS_LAUNCH_SMEM(KID_NORMAL_SAMPLE, (k_logprob_tile<D_NORMAL, true, true>), dim3(g.grid), dim3(g.threads), g.smem, st, e4, m4, s4, lp, (uint32_t)K, R,
                         (uint32_t)D4, (uint32_t)(R * D4), g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, ls, (float4*)z);
        else
          ZS_LAUNCH_SMEM(KID_NORMAL_SAMPLE, (k_logprob_tile<D_NORMAL, false, true>), dim3(g.grid), dim3(g.threads), g.smem, st, e4, m4, s4, lp, (uint32_t)K, R,
                         (uint32_t)D4, (uint32_t)(R * D4), g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, ls, (float4*)z);
        ZS_CHECK_LAUNCH();
        return 0;
      }
    }
    const K1Tile kt_ = eps ? K1Tile() : k1_tile(K, R, D4, lp != nullptr);
    if (kt_.ok) {
      static const int nt_env = env_knob("ZS_K1_NT", -1);
      const bool nt = nt_env >= 0 ? nt_env != 0 : (double)K * (double)M * 4.0 > 268435456.0;   // z cannot stay in the Infinity Cache
#define ZS_LAUNCH_TILE(L, T)                                                                                  \
  ZS_LAUNCH_SMEM(KID_NORMAL_SAMPLE, (k_sample_tile<D_NORMAL, L, T>), dim3(kt_.grid), dim3(kt_.threads), kt_.smem, st, \
                 m4, s4, seed, offset, rng_state, (float4*)z, lp, (uint32_t)K, R, (uint32_t)D4,                \
                 (uint32_t)(R * D4), kt_.kchunk, kt_.KB, kt_.n_ptiles, kt_.total, sk, sr, ls, rng_used,    \
                 (float4*)nullptr, false, 0u)
      if (nt) { if (lp) ZS_LAUNCH_TILE(true, true); else ZS_LAUNCH_TILE(false, true); }
      else    { if (lp) ZS_LAUNCH_TILE(true, false); else ZS_LAUNCH_TILE(false, false); }
#undef ZS_LAUNCH_TILE
    } else if (D4 <= 64) {
      RowMap rm = row_map(D4);
      const int64_t row_tiles = (R + rm.rpw - 1) / rm.rpw;
      // enough waves to fill the chip (256 CUs x 8 waves) before reusing parameters across K
      int64_t want = 256 * 8;
      int64_t kt = (want + row_tiles - 1) / row_tiles;
      if (kt < 1) kt = 1;
      if (kt > K) kt = K;
      const int64_t kchunk = (K + kt - 1) / kt;
      const int64_t total = row_tiles * ((K + kchunk - 1) / kchunk);
      const unsigned grid = grid_for(total, 4);
#define ZS_LAUNCH_SMALL(E, L, T)                                                                          \
  ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_smallrow<E, L, T>), dim3(grid), dim3(256), st, m4, s4, e4, \
            seed, offset, rng_state, (float4*)z, lp, K, R, D4, rm.G, rm.rpw, kchunk, sk, sr, ls, rng_used)
      static const int nt_env2 = env_knob("ZS_K1_NT", -1);
      const bool nt = nt_env2 >= 0 ? nt_env2 != 0 : (double)K * (double)M * 4.0 > 268435456.0;   // z cannot stay in the Infinity Cache
      if (nt) {
        if (eps) { if (lp) ZS_LAUNCH_SMALL(true, true, true); else ZS_LAUNCH_SMALL(true, false, true); }
        else     { if (lp) ZS_LAUNCH_SMALL(false, true, true); else ZS_LAUNCH_SMALL(false, false, true); }
      } else {
        if (eps) { if (lp) ZS_LAUNCH_SMALL(true, true, false); else ZS_LAUNCH_SMALL(true, false, false); }
        else     { if (lp) ZS_LAUNCH_SMALL(false, true, false); else ZS_LAUNCH_SMALL(false, false, false); }
      }
#undef ZS_LAUNCH_SMALL
    } else {
      const bool few = K * R < 2048;                      // fewer rows than the chip has SIMDs: a workgroup per row
      const unsigned grid = few ? grid_for(K * R, 1) : grid_for(K * R, 4);
#define ZS_LAUNCH_LONG(E, L, W)                                                                           \
  ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_longrow<E, L, W>), dim3(grid), dim3(256), st, m4, s4, e4, seed, \
                     offset, rng_state, (float4*)z, lp, K, R, D4, sk, sr, ls, rng_used)
      if (few) {
        if (eps) { if (lp) ZS_LAUNCH_LONG(true, true, 4); else ZS_LAUNCH_LONG(true, false, 4); }
        else     { if (lp) ZS_LAUNCH_LONG(false, true, 4); else ZS_LAUNCH_LONG(false, false, 4); }
      } else {
        if (eps) { if (lp) ZS_LAUNCH_LONG(true, true, 1); else ZS_LAUNCH_LONG(true, false, 1); }
        else     { if (lp) ZS_LAUNCH_LONG(false, true, 1); else ZS_LAUNCH_LONG(false, false, 1); }
      }
#undef ZS_LAUNCH_LONG
    }
  } else if (D >= 8) {
    const unsigned grid = grid_for(K * R, 4);
    if (eps)
      ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_waverow<true>), dim3(grid), dim3(256), st, mu, sigma, eps, seed,
                offset, rng_state, z, lp, K, R, D, sk, sr, ls, rng_used);
    else
      ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_waverow<false>), dim3(grid), dim3(256), st, mu, sigma, eps, seed,
                offset, rng_state, z, lp, K, R, D, sk, sr, ls, rng_used);
  } else {
    const unsigned grid = grid_for(K * R, 256);
    if (eps)
      ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_serial<true>), dim3(grid), dim3(256), st, mu, sigma, eps, seed,
                         offset, rng_state, z, lp, K, R, D, sk, sr, ls, rng_used);
    else
      ZS_LAUNCH(KID_NORMAL_SAMPLE, (k_normal_sample_serial<false>), dim3(grid), dim3(256), st, mu, sigma, eps, seed,
                         offset, rng_state, z, lp, K, R, D, sk, sr, ls, rng_used);
  }
  ZS_CHECK_LAUNCH();
  return 0;
}


// Whether zs_normal_sample_logprob_pair_f32 is ONE launch for this shape (16-byte aligned operands assumed): callers that gain
// nothing from two launches behind one call (a second draw that other kernels batch) ask first.
extern "C" int zs_normal_sample_pair_one_launch(int64_t K, int64_t M, int64_t D, int want_lp) {
  if (K < 1 || M < 1 || D < 1 || (M % D) != 0 || (D % 4) != 0 || 2 * K > 0x7fffffff) return 0;
  return k1_tile(2 * K, M / D, (int)(D / 4), want_lp != 0).ok ? 1 : 0;
}

// Two independent draws of K particles each (Philox call ids offset and offset + 1) -- what the objectives do with every
// latent (stochastic_tensor.py:115-127, then elbo.py:122 / importance_weighted_objective.py:85) -- as ONE launch when the
// flat-plane kernel takes the shape, else as the two launches it stands for.  Results are bit for bit those of two calls of
// zs_normal_sample_logprob_f32: z [2 K, M] (first K particles: the first draw), lp element (k, r) of draw j at
// lp[(j K + k) * sk + r * sr].
extern "C" int zs_normal_sample_logprob_pair_f32(const float* mu, const float* sigma, uint64_t seed, uint64_t offset,
                                                 const uint64_t* rng_state, float* z, float* lp, int64_t K, int64_t M,
                                                 int64_t D, int64_t sk, int64_t sr, int sigma_is_logstd, uint64_t* rng_used,
                                                 void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!mu || !sigma || !z) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = M / D;
  if ((D % 4 == 0) && aligned16(mu) && aligned16(sigma) && aligned16(z) && 2 * K <= 0x7fffffff) {
    const int D4 = (int)(D / 4);
    const K1Tile kt_ = k1_tile(2 * K, R, D4, lp != nullptr);
    if (kt_.ok) {
      const bool nt = 2.0 * (double)K * (double)M * 4.0 > 268435456.0;
#define ZS_LAUNCH_TILE2(L, T)                                                                                 \
  ZS_LAUNCH_SMEM(KID_NORMAL_SAMPLE_PAIR, (k_sample_tile<D_NORMAL, L, T>), dim3(kt_.grid), dim3(kt_.threads), kt_.smem, st, \
                 (const float4*)mu, (const float4*)sigma, seed, offset, rng_state, (float4*)z, lp, (uint32_t)(2 * K), R,  \
                 (uint32_t)D4, (uint32_t)(R * D4), kt_.kchunk, kt_.KB, kt_.n_ptiles, kt_.total, sk, sr, ls, rng_used,     \
                 (float4*)nullptr, false, (uint32_t)K)
      if (nt) { if (lp) ZS_LAUNCH_TILE2(true, true); else ZS_LAUNCH_TILE2(false, true); }
      else    { if (lp) ZS_LAUNCH_TILE2(true, false); else ZS_LAUNCH_TILE2(false, false); }
#undef ZS_LAUNCH_TILE2
      ZS_CHECK_LAUNCH();
      return 0;
    }
  }
  int rc = zs_normal_sample_logprob_f32(mu, sigma, nullptr, seed, offset, rng_state, z, lp, K, M, D, sk, sr, sigma_is_logstd, rng_used,
                                        stream);
  if (rc != 0) return rc;
  return zs_normal_sample_logprob_f32(mu, sigma, nullptr, seed, offset + 1, rng_state, z + K * M, lp ? lp + K * sk : nullptr, K, M, D,
                                      sk, sr, sigma_is_logstd, nullptr, stream);
}

extern "C" int zs_normal_sample_logprob_bwd_f32(const float* sigma, const float* eps, uint64_t seed,
                                                uint64_t offset, const uint64_t* rng_state,
                                                const float* gz, const float* glp,
                                                int64_t gsk, int64_t gsr, float* gmu, float* gsigma,
                                                int64_t K, int64_t M, int64_t D, int sigma_is_logstd, void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!sigma || !gmu || !gsigma) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (D % 4 == 0) && aligned16(sigma) && aligned16(gmu) && aligned16(gsigma) &&
                   (!eps || aligned16(eps)) && (!gz || aligned16(gz));
  if (vec) {
    const int64_t M4 = M / 4;
    const unsigned grid = (unsigned)((M4 + 63) / 64);
    // K-slices per workgroup: 4, or 16 when the parameter plane gives fewer than 256 workgroups and there are particles to
    // share out (the config shapes inside a training step: fewer dependent rounds of cold-cache loads per wave)
    static const int wide_env = env_knob("ZS_K1_BWD_WIDE", -1);     // experiments only
    const bool wide = wide_env >= 0 ? wide_env != 0 : (grid < 256 && K >= 16);
#define ZS_LAUNCH_K1_BWD(E, G, L)                                                                                             \
  do {                                                                                                                        \
    if (wide)                                                                                                                 \
      ZS_LAUNCH(KID_NORMAL_SAMPLE_BWD, (k_normal_sample_bwd<E, G, L, 16>), dim3(grid), dim3(1024), st, (const float4*)sigma,  \
                (const float4*)eps, seed, offset, rng_state, (const float4*)gz, glp, gsk, gsr, (float4*)gmu, (float4*)gsigma, \
                K, M4, (int)(D / 4), ls);                                                                                     \
    else                                                                                                                      \
      ZS_LAUNCH(KID_NORMAL_SAMPLE_BWD, (k_normal_sample_bwd<E, G, L, 4>), dim3(grid), dim3(256), st, (const float4*)sigma,    \
                (const float4*)eps, seed, offset, rng_state, (const float4*)gz, glp, gsk, gsr, (float4*)gmu, (float4*)gsigma, \
                K, M4, (int)(D / 4), ls);                                                                                     \
  } while (0)
    if (gz) {
      if (eps) { if (glp) ZS_LAUNCH_K1_BWD(true, true, true); else ZS_LAUNCH_K1_BWD(true, true, false); }
      else     { if (glp) ZS_LAUNCH_K1_BWD(false, true, true); else ZS_LAUNCH_K1_BWD(false, true, false); }
    } else {
      if (glp) ZS_LAUNCH_K1_BWD(false, false, true); else ZS_LAUNCH_K1_BWD(false, false, false);
    }
#undef ZS_LAUNCH_K1_BWD
  } else {
    if ((M + 63) / 64 > 0x7fffffffll) return ZS_ENOTSUP;
    ZS_LAUNCH(KID_NORMAL_SAMPLE_BWD, k_normal_sample_bwd_serial, dim3((unsigned)((M + 63) / 64)), dim3(256), st, sigma, eps, seed,
                       offset, rng_state, gz, glp, gsk, gsr, gmu, gsigma, K, M, D, ls);
  }
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_f32(const float* x, int64_t Px, const float* mu, int64_t Pm,
                                     const float* sigma, int64_t Ps, float* lp,
                                     int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, int sigma_is_logstd,
                                     void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !lp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (D % 4 == 0) && period_ok_rows(Px, D, N) && period_ok_rows(Pm, D, N) &&
                   period_ok_rows(Ps, D, N) && (Px == 1 || aligned16(x)) && (Pm == 1 || aligned16(mu)) &&
                   (Ps == 1 || aligned16(sigma));
  if (vec) {
    const int D4 = (int)(D / 4);
    RowMap rm = row_map(D4);
    const int64_t rows = K * R;
    const int64_t tiles = (rows + rm.rpw - 1) / rm.rpw;
    const bool simple = (Px == N || Px == 1) && (Pm == N || Pm == 1) && (Ps == N || Ps == 1);
    if (K > 1 && D4 <= 64 && Px == N && Pm == R * D && Ps == R * D) {
      launch_logprob_krep<D_NORMAL>(KID_NORMAL_LOGPROB, x, mu, sigma, lp, K, R, D4, sk, sr, ls, st);
    } else if (simple) {
      ZS_LAUNCH(KID_NORMAL_LOGPROB, k_normal_logprob_full, dim3(grid_for(tiles, 4)), dim3(256), st, (const float4*)x,
                (int)(Px == 1 && N != 1), (const float4*)mu, (int)(Pm == 1 && N != 1), (const float4*)sigma,
                (int)(Ps == 1 && N != 1), lp, rows, R, D4, rm.G, rm.rpw, rm.p2, sk, sr, ls);
    } else if (D4 > 64 && rows < 2048 && Px != 1 && (Pm != 1 || Ps == 1)) {
#define ZS_LAUNCH_BLOCKROW(MSC, SSC)                                                                              \
  ZS_LAUNCH(KID_NORMAL_LOGPROB, (k_normal_logprob_blockrow<MSC, SSC>), dim3((unsigned)rows), dim3(256), st,        \
            (const float4*)x, Px / D, (const float4*)mu, Pm == 1 ? 1 : Pm / D, (const float4*)sigma,             \
            Ps == 1 ? 1 : Ps / D, lp, K, R, D4, sk, sr, ls)
      if (Pm == 1) ZS_LAUNCH_BLOCKROW(true, true);
      else if (Ps == 1) ZS_LAUNCH_BLOCKROW(false, true);
      else ZS_LAUNCH_BLOCKROW(false, false);
#undef ZS_LAUNCH_BLOCKROW
    } else {
      ZS_LAUNCH(KID_NORMAL_LOGPROB, k_normal_logprob_rows, dim3(grid_for(tiles, 4)), dim3(256), st, (const float4*)x,
                Px == 1 ? 0 : Px / D, (const float4*)mu, Pm == 1 ? 0 : Pm / D, (const float4*)sigma,
                Ps == 1 ? 0 : Ps / D, lp, K, R, D4, rm.G, rm.rpw, rm.p2, sk, sr, ls);
    }
  } else if (D >= 8) {
    ZS_LAUNCH(KID_NORMAL_LOGPROB, k_normal_logprob_waverow, dim3(grid_for(K * R, 4)), dim3(256), st, x, Px, mu, Pm,
              sigma, Ps, lp, K, R, D, sk, sr, ls);
  } else {
    ZS_LAUNCH(KID_NORMAL_LOGPROB, k_normal_logprob_serial, dim3(grid_for(K * R, 256)), dim3(256), st, x, Px, mu, Pm,
                       sigma, Ps, lp, K, R, D, sk, sr, ls);
  }
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_bwd_f32(const float* x, int64_t Px, const float* mu, int64_t Pm,
                                         const float* sigma, int64_t Ps, const float* glp, int64_t gsk,
                                         int64_t gsr, float* gx, float* gmu, float* gsigma,
                                         int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_LOGPROB_BWD, k_normal_logprob_bwd_elem, dim3(grid_for(N, 256)), dim3(256), (hipStream_t)stream, x,
                     Px, mu, Pm, sigma, Ps, glp, gsk, gsr, gx, gmu, gsigma, N, R, D, ls);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_bwd_ksum_f32(const float* x, const float* mu, const float* sigma,
                                              const float* glp, int64_t gsk, int64_t gsr, float* gx,
                                              float* gmu, float* gsigma, int64_t K, int64_t R, int64_t D,
                                              int sigma_is_logstd, void* stream) {
  const bool ls = sigma_is_logstd != 0;
  if (K < 1 || R < 0 || D < 1) return ZS_EINVAL;
  const int64_t M = R * D;
  if (M == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  launch_logprob_bwd_ksum<D_NORMAL>(KID_NORMAL_LOGPROB_BWD_KSUM, x, mu, sigma, glp, gsk, gsr, gx, gmu, gsigma, K, R, D, ls, st);
  ZS_CHECK_LAUNCH();
  return 0;
}
