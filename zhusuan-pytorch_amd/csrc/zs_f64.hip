// float64 entry points of the C ABI (include/zs_hip.h).  The reference accepts float64 parameters
// (zhusuan/distributions/utils.py:5, test/distributions/utils.py:test_dtype_2parameter); none of
// BASELINE.json's configs uses them, so these are straightforward one-thread-per-row / one-workgroup-per-
// datapoint kernels with libm-grade double math: correct and coalesced where D = 1, not tuned.
// Draws: the same Philox4x32-10 + Box-Muller stream as the fp32 kernels (24-bit uniforms), widened to double.
#include "zs_common.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

constexpr double kC = -0.91893853320467274178;  // -0.5*log(2*pi)
constexpr double kEps = 1e-8;                   // bernoulli.py:94

__device__ __forceinline__ double normal_term(double x, double mu, double sigma) {
  const double logstd = log(sigma);             // normal.py:121
  const double precision = exp(-2.0 * logstd);  // normal.py:123
  const double d = x - mu;
  return (kC - logstd) - 0.5 * precision * (d * d);
}
__device__ __forceinline__ double bern_term(double p, double x) {
  return x * log(p + kEps) + (1.0 - x) * log((1.0 - p) + kEps);
}
__device__ __forceinline__ double sig_of(double v, bool ls) { return ls ? exp(v) : v; }   // Normal(logstd=...), normal.py:56
__device__ __forceinline__ double sigmoid_d(double l) { return 1.0 / (1.0 + exp(-l)); }
__device__ __forceinline__ double mul_add_2round(double m, double s, double e) {
#pragma clang fp contract(off)   // two roundings like the reference's separate mul and add (normal.py:105)
  const double prod = s * e;
  return m + prod;
}
__device__ __forceinline__ double eps_at(const double* eps, int64_t i, uint64_t seed, uint64_t call) {
  if (eps) return eps[i];
  return (double)f4_get(philox_normal4((uint64_t)(i >> 2), call, seed), (int)(i & 3));
}

__global__ __launch_bounds__(256) void k64_normal_sample(const double* __restrict__ mu, const double* __restrict__ sigma,
                                                         const double* __restrict__ eps, uint64_t seed, uint64_t call,
                                                         const uint64_t* __restrict__ rs, double* __restrict__ z,
                                                         double* __restrict__ lp, int64_t K, int64_t R, int64_t D,
                                                         int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  if (rng_used && blockIdx.x == 0 && threadIdx.x == 0) { rng_used[0] = seed; rng_used[1] = call; }
  const int64_t rows = K * R, M = R * D;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    int64_t k, r;
    divmod(row, R, k, r);
    double acc = 0.0;
    for (int64_t d = 0; d < D; ++d) {
      const int64_t m = r * D + d, i = k * M + m;
      const double sg = sig_of(sigma[m], ls);
      const double zz = mul_add_2round(mu[m], sg, eps_at(eps, i, seed, call));
      z[i] = zz;
      if (lp) acc += normal_term(zz, mu[m], sg);
    }
    if (lp) lp[k * sk + r * sr] = acc;
  }
}

__global__ __launch_bounds__(256) void k64_normal_sample_bwd(const double* __restrict__ sigma, const double* __restrict__ eps,
                                                             uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs,
                                                             const double* __restrict__ gz, const double* __restrict__ glp,
                                                             int64_t gsk, int64_t gsr, double* __restrict__ gmu,
                                                             double* __restrict__ gsigma, int64_t K, int64_t M, int64_t D, bool ls) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = m / D;
    double a = 0.0, b = 0.0, g = 0.0;
    for (int64_t k = 0; k < K; ++k) {
      const int64_t i = k * M + m;
      if (gz) {
        a += gz[i];
        b += gz[i] * eps_at(eps, i, seed, call);
      }
      if (glp) g += glp[k * gsk + r * gsr];
    }
    gmu[m] = a;
    const double sg = sig_of(sigma[m], ls);
    gsigma[m] = ls ? b * sg - g : b - g / sg;
  }
}

__global__ __launch_bounds__(256) void k64_normal_logprob(const double* __restrict__ x, int64_t Px, const double* __restrict__ mu,
                                                          int64_t Pm, const double* __restrict__ sigma, int64_t Ps,
                                                          double* __restrict__ lp, int64_t K, int64_t R, int64_t D,
                                                          int64_t sk, int64_t sr, bool ls) {
  const int64_t rows = K * R;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int64_t d = 0; d < D; ++d) {
      const int64_t i = row * D + d;
      acc += normal_term(x[mod_fast(i, Px)], mu[mod_fast(i, Pm)], sig_of(sigma[mod_fast(i, Ps)], ls));
    }
    int64_t k, r;
    divmod(row, R, k, r);
    lp[k * sk + r * sr] = acc;
  }
}

__device__ __forceinline__ void normal_partials(double x, double mu, double sigma, double g, double& gx, double& gm, double& gs,
                                                bool ls) {
  sigma = sig_of(sigma, ls);
  const double prec = exp(-2.0 * log(sigma));
  const double d = x - mu;
  const double t = g * prec * d;
  gx = -t;
  gm = t;
  gs = ls ? g * (prec * d * d - 1.0) : g * (prec * d * d - 1.0) / sigma;   // d/d logstd = sigma * d/d sigma
}

__global__ __launch_bounds__(256) void k64_normal_logprob_bwd(const double* __restrict__ x, int64_t Px, const double* __restrict__ mu,
                                                              int64_t Pm, const double* __restrict__ sigma, int64_t Ps,
                                                              const double* __restrict__ glp, int64_t gsk, int64_t gsr,
                                                              double* __restrict__ gx, double* __restrict__ gmu,
                                                              double* __restrict__ gsigma, int64_t N, int64_t R, int64_t D, bool ls) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row, dd, k, r;
    divmod(i, D, row, dd);
    divmod(row, R, k, r);
    double a, b, c;
    normal_partials(x[mod_fast(i, Px)], mu[mod_fast(i, Pm)], sigma[mod_fast(i, Ps)], glp[k * gsk + r * gsr], a, b, c, ls);
    if (gx) gx[i] = a;
    if (gmu) gmu[i] = b;
    if (gsigma) gsigma[i] = c;
  }
}

__global__ __launch_bounds__(256) void k64_normal_logprob_bwd_ksum(const double* __restrict__ x, const double* __restrict__ mu,
                                                                   const double* __restrict__ sigma, const double* __restrict__ glp,
                                                                   int64_t gsk, int64_t gsr, double* __restrict__ gx,
                                                                   double* __restrict__ gmu, double* __restrict__ gsigma,
                                                                   int64_t K, int64_t M, int64_t D, bool ls,
                                                                   const double* __restrict__ gscale, int64_t gss) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = m / D;
    const double gs = gscale ? gscale[r * gss] : 1.0;
    double sa = 0.0, sb = 0.0;
    for (int64_t k = 0; k < K; ++k) {
      double a, b, c;
      normal_partials(x[k * M + m], mu[m], sigma[m], gscale ? glp[k * gsk + r * gsr] * gs : glp[k * gsk + r * gsr], a, b, c, ls);
      if (gx) gx[k * M + m] = a;
      sa += b;
      sb += c;
    }
    if (gmu) gmu[m] = sa;
    if (gsigma) gsigma[m] = sb;
  }
}

template <bool LOGITS>
__global__ __launch_bounds__(256) void k64_bern_logprob(const double* __restrict__ p, const double* __restrict__ x, int64_t Px,
                                                        double* __restrict__ lp, double* __restrict__ probs_out, int64_t K,
                                                        int64_t R, int64_t D, int64_t sk, int64_t sr) {
  const int64_t rows = K * R;
  for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int64_t d = 0; d < D; ++d) {
      const int64_t i = row * D + d;
      double pv = p[i];
      if (LOGITS) {
        pv = sigmoid_d(pv);
        if (probs_out) probs_out[i] = pv;
      }
      acc += bern_term(pv, x[mod_fast(i, Px)]);
    }
    int64_t k, r;
    divmod(row, R, k, r);
    lp[k * sk + r * sr] = acc;
  }
}

template <bool LOGITS>
__global__ __launch_bounds__(256) void k64_bern_logprob_bwd(const double* __restrict__ p, const double* __restrict__ x, int64_t Px,
                                                            const double* __restrict__ glp, int64_t gsk, int64_t gsr,
                                                            double* __restrict__ gp, int64_t N, int64_t R, int64_t D,
                                                            const double* __restrict__ gscale, int64_t gss) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row, dd, k, r;
    divmod(i, D, row, dd);
    divmod(row, R, k, r);
    double pv = p[i], scale = 1.0;
    if (LOGITS) {
      pv = sigmoid_d(pv);
      scale = pv * (1.0 - pv);
    }
    const double xv = x[mod_fast(i, Px)];
    double g = glp[k * gsk + r * gsr];
    if (gscale) g *= gscale[r * gss];
    gp[i] = g * (xv / (pv + kEps) - (1.0 - xv) / ((1.0 - pv) + kEps)) * scale;
  }
}

__global__ __launch_bounds__(256) void k64_bern_sample(const double* __restrict__ p, int64_t Pp, double* __restrict__ out, int64_t N,
                                                       uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const Philox4 r = philox4x32_10((uint64_t)(i >> 2), call, seed);
    const uint32_t w = (i & 3) == 0 ? r.x : ((i & 3) == 1 ? r.y : ((i & 3) == 2 ? r.z : r.w));
    out[i] = (double)u01(w) < p[mod_fast(i, Pp)] ? 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(256) void k64_philox_normal(double* __restrict__ out, int64_t N, uint64_t seed, uint64_t call,
                                                         const uint64_t* __restrict__ rs) {
  if (rs) { seed = rs[0]; call += rs[1]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = eps_at(nullptr, i, seed, call);
}

// ---- importance-weighted reduction, one 256-thread workgroup per datapoint (any K)
__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  return v;
}
__device__ __forceinline__ double wmax_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, ZS_WAVE));
  return v;
}
__device__ __forceinline__ double bsum_d(double v, double* sh) {
  v = wsum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double bmax_d(double v, double* sh) {
  v = wmax_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__device__ __forceinline__ int bmin_i(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o, ZS_WAVE);
    v = t < v ? t : v;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const int a = sh[0] < sh[1] ? sh[0] : sh[1], c = sh[2] < sh[3] ? sh[2] : sh[3];
  return a < c ? a : c;
}

struct IwExt64 {   // extras of zs_iw_objective_f64 (zs_iw.hip: IwExt)
  const double* logp_b;
  int64_t ld_b;
  const double* logp_c;   // a third term, added last: ((a + b) + c) - q (zs_bernoulli_iw_objective_f64)
  int64_t ld_c;
  double scale;
  double* mean_cost;
  double* partials;
  unsigned* ticket;
  double inv_B;
};

__global__ __launch_bounds__(256) void k64_iw_reduce(const double* __restrict__ logp, int64_t ld_p, const double* __restrict__ logq,
                                                     int64_t ld_q, int64_t B, int64_t K, int estimator, double* __restrict__ cost_b,
                                                     double* __restrict__ bound_b, double* __restrict__ coef_p,
                                                     double* __restrict__ coef_q, IwExt64 ext) {
  __shared__ double shf[4];
  __shared__ int shi[4];
  __shared__ bool last;
  double my_cost = 0.0;
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const double* __restrict__ pp = logp + b * ld_p;
    const double* __restrict__ pb = ext.logp_b ? ext.logp_b + b * ext.ld_b : nullptr;
    const double* __restrict__ pc = ext.logp_c ? ext.logp_c + b * ext.ld_c : nullptr;
    const double* __restrict__ qq = logq + b * ld_q;
#define ZS_LP(k) (pc ? (pb ? pp[k] + pb[k] : pp[k]) + pc[k] : (pb ? pp[k] + pb[k] : pp[k]))
#define ZS_LW(k) (ZS_LP(k) - qq[k])
    double mx = -INFINITY, sl = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const double l = ZS_LW(k);
      mx = fmax(mx, l);
      sl += l;
    }
    const double m1 = bmax_d(mx, shf), sumL = bsum_d(sl, shf);
    int jm = 0x7fffffff;
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const double l = ZS_LW(k);
      if (l == m1 && (int)k < jm) jm = (int)k;
      s += exp(l - m1);
    }
    const int jstar = bmin_i(jm, shi);
    const double S = bsum_d(s, shf);
    double m2 = -INFINITY, S2 = 0.0;
    if (estimator == ZS_IW_VIMCO) {
      double t = -INFINITY;
      for (int64_t k = threadIdx.x; k < K; k += 256)
        if ((int)k != jstar) t = fmax(t, ZS_LW(k));
      m2 = bmax_d(t, shf);
      double s2 = 0.0;
      for (int64_t k = threadIdx.x; k < K; k += 256)
        if ((int)k != jstar) s2 += exp(ZS_LW(k) - m2);
      S2 = bsum_d(s2, shf);
    }
    const double logS = log(S), invKm1 = K > 1 ? 1.0 / (double)(K - 1) : 0.0;
    double ct = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += 256) {
      const double lq = qq[k], l = ZS_LP(k) - lq;
      const double e = exp(l - m1), wt = e / S;
      double c1 = -wt * l, cq = wt;
      if (estimator == ZS_IW_VIMCO) {
        const double sub = (sumL - l) * invKm1;
        double signal;
        if ((int)k != jstar || S >= 2.0) signal = -log1p((exp(sub - m1) - e) / S);
        else signal = (logS - log(S2 + exp(sub - m2))) + (m1 - m2);
        c1 -= lq * signal;
        cq = wt - signal;
      }
      ct += c1;
      if (coef_p) coef_p[b * K + k] = -wt * ext.scale;
      if (coef_q) coef_q[b * K + k] = cq * ext.scale;
    }
    const double cost = bsum_d(ct, shf);
    my_cost += cost;
    if (threadIdx.x == 0) {
      if (cost_b) cost_b[b] = cost;
      if (bound_b) bound_b[b] = log(S / (double)K) + m1;
    }
  }
#undef ZS_LW
#undef ZS_LP
  if (ext.mean_cost) {   // deterministic batch mean: the last workgroup to arrive adds the partials in index order
    if (threadIdx.x == 0) {
      ext.partials[blockIdx.x] = my_cost;
      __threadfence();
      last = (atomicAdd(ext.ticket, 1u) == gridDim.x - 1);
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
      __threadfence();
      double s = 0.0;
      for (unsigned i = threadIdx.x; i < gridDim.x; i += 64)
        s += __hip_atomic_load(ext.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s = wsum_d(s);
      if (threadIdx.x == 0) {
        ext.mean_cost[0] = s * ext.inv_B;
        *ext.ticket = 0u;
      }
    }
  }
}


__global__ __launch_bounds__(256) void k64_lme(const double* __restrict__ x, int64_t ld, int64_t B, int64_t K, double* __restrict__ out) {
  __shared__ double shf[4];
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    const double* __restrict__ xx = x + b * ld;
    double mx = -INFINITY;
    for (int64_t k = threadIdx.x; k < K; k += 256) mx = fmax(mx, xx[k]);
    const double m = bmax_d(mx, shf);
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += 256) s += exp(xx[k] - m);
    const double S = bsum_d(s, shf);
    if (threadIdx.x == 0) out[b] = log(S / (double)K) + m;
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int zs_normal_sample_logprob_f64(const double* mu, const double* sigma, const double* eps, uint64_t seed,
                                            uint64_t offset, const uint64_t* rng_state, double* z, double* lp, int64_t K,
                                            int64_t M, int64_t D, int64_t sk, int64_t sr, int sigma_is_logstd,
                                            uint64_t* rng_used, void* stream) {
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!mu || !sigma || !z) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_SAMPLE, k64_normal_sample, dim3(grid_for(K * (M / D), 256)), dim3(256), ST, mu, sigma, eps, seed, offset,
            rng_state, z, lp, K, M / D, D, sk, sr, sigma_is_logstd != 0, rng_used);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_sample_logprob_pair_f64(const double* mu, const double* sigma, uint64_t seed, uint64_t offset,
                                                 const uint64_t* rng_state, double* z, double* lp, int64_t K, int64_t M,
                                                 int64_t D, int64_t sk, int64_t sr, int sigma_is_logstd, uint64_t* rng_used,
                                                 void* stream) {
  const int rc = zs_normal_sample_logprob_f64(mu, sigma, nullptr, seed, offset, rng_state, z, lp, K, M, D, sk, sr, sigma_is_logstd,
                                              rng_used, stream);
  if (rc != 0 || M <= 0) return rc;
  return zs_normal_sample_logprob_f64(mu, sigma, nullptr, seed, offset + 1, rng_state, z + K * M, lp ? lp + K * sk : nullptr, K, M, D,
                                      sk, sr, sigma_is_logstd, nullptr, stream);
}

extern "C" int zs_normal_sample_logprob_bwd_f64(const double* sigma, const double* eps, uint64_t seed, uint64_t offset,
                                                const uint64_t* rng_state, const double* gz, const double* glp, int64_t gsk,
                                                int64_t gsr, double* gmu, double* gsigma, int64_t K, int64_t M, int64_t D,
                                                int sigma_is_logstd, void* stream) {
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!sigma || !gmu || !gsigma) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_SAMPLE_BWD, k64_normal_sample_bwd, dim3(grid_for(M, 256)), dim3(256), ST, sigma, eps, seed, offset, rng_state,
            gz, glp, gsk, gsr, gmu, gsigma, K, M, D, sigma_is_logstd != 0);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_f64(const double* x, int64_t Px, const double* mu, int64_t Pm, const double* sigma, int64_t Ps,
                                     double* lp, int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, int sigma_is_logstd,
                                     void* stream) {
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !lp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_LOGPROB, k64_normal_logprob, dim3(grid_for(K * R, 256)), dim3(256), ST, x, Px, mu, Pm, sigma, Ps, lp, K, R, D,
            sk, sr, sigma_is_logstd != 0);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_bwd_f64(const double* x, int64_t Px, const double* mu, int64_t Pm, const double* sigma,
                                         int64_t Ps, const double* glp, int64_t gsk, int64_t gsr, double* gx, double* gmu,
                                         double* gsigma, int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream) {
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_LOGPROB_BWD, k64_normal_logprob_bwd, dim3(grid_for(N, 256)), dim3(256), ST, x, Px, mu, Pm, sigma, Ps, glp, gsk,
            gsr, gx, gmu, gsigma, N, R, D, sigma_is_logstd != 0);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_normal_logprob_bwd_ksum_f64(const double* x, const double* mu, const double* sigma, const double* glp,
                                              int64_t gsk, int64_t gsr, double* gx, double* gmu, double* gsigma, int64_t K,
                                              int64_t R, int64_t D, int sigma_is_logstd, void* stream) {
  if (K < 1 || R < 0 || D < 1) return ZS_EINVAL;
  const int64_t M = R * D;
  if (M == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  ZS_LAUNCH(KID_NORMAL_LOGPROB_BWD_KSUM, k64_normal_logprob_bwd_ksum, dim3(grid_for(M, 256)), dim3(256), ST, x, mu, sigma, glp, gsk,
            gsr, gx, gmu, gsigma, K, M, D, sigma_is_logstd != 0, (const double*)nullptr, (int64_t)0);
  ZS_CHECK_LAUNCH();
  return 0;
}

static int bern_fwd64(bool logits, const double* p, const double* x, int64_t Px, double* lp, double* probs_out, int64_t K,
                      int64_t R, int64_t D, int64_t sk, int64_t sr, void* stream) {
  if (K < 1 || R < 0 || D < 1 || Px < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!p || !x || !lp) return ZS_EINVAL;
  if (N % Px) return ZS_EINVAL;
  if (logits)
    ZS_LAUNCH(KID_BERN_LOGITS_LOGPROB, (k64_bern_logprob<true>), dim3(grid_for(K * R, 256)), dim3(256), ST, p, x, Px, lp, probs_out,
              K, R, D, sk, sr);
  else
    ZS_LAUNCH(KID_BERN_LOGPROB, (k64_bern_logprob<false>), dim3(grid_for(K * R, 256)), dim3(256), ST, p, x, Px, lp, probs_out, K, R,
              D, sk, sr);
  ZS_CHECK_LAUNCH();
  return 0;
}
static int bern_bwd64(bool logits, const double* p, const double* x, int64_t Px, const double* glp, int64_t gsk, int64_t gsr,
                      double* gp, int64_t K, int64_t R, int64_t D, void* stream, const double* gscale = nullptr, int64_t gss = 0) {
  if (K < 1 || R < 0 || D < 1 || Px < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!p || !x || !glp || !gp) return ZS_EINVAL;
  if (N % Px) return ZS_EINVAL;
  if (logits)
    ZS_LAUNCH(KID_BERN_LOGITS_LOGPROB_BWD, (k64_bern_logprob_bwd<true>), dim3(grid_for(N, 256)), dim3(256), ST, p, x, Px, glp, gsk,
              gsr, gp, N, R, D, gscale, gss);
  else
    ZS_LAUNCH(KID_BERN_LOGPROB_BWD, (k64_bern_logprob_bwd<false>), dim3(grid_for(N, 256)), dim3(256), ST, p, x, Px, glp, gsk, gsr,
              gp, N, R, D, gscale, gss);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_bernoulli_logprob_f64(const double* p, const double* x, int64_t Px, double* lp, int64_t K, int64_t R,
                                        int64_t D, int64_t sk, int64_t sr, void* stream) {
  return bern_fwd64(false, p, x, Px, lp, nullptr, K, R, D, sk, sr, stream);
}
extern "C" int zs_bernoulli_logprob_bwd_f64(const double* p, const double* x, int64_t Px, const double* glp, int64_t gsk,
                                            int64_t gsr, double* gp, int64_t K, int64_t R, int64_t D, void* stream) {
  return bern_bwd64(false, p, x, Px, glp, gsk, gsr, gp, K, R, D, stream);
}
extern "C" int zs_bernoulli_logits_logprob_f64(const double* logits, const double* x, int64_t Px, double* lp, double* probs_out,
                                               int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, void* stream) {
  return bern_fwd64(true, logits, x, Px, lp, probs_out, K, R, D, sk, sr, stream);
}
extern "C" int zs_bernoulli_logits_logprob_bwd_f64(const double* logits, const double* x, int64_t Px, const double* glp,
                                                   int64_t gsk, int64_t gsr, double* glogits, int64_t K, int64_t R, int64_t D,
                                                   void* stream) {
  return bern_bwd64(true, logits, x, Px, glp, gsk, gsr, glogits, K, R, D, stream);
}

extern "C" int zs_bernoulli_sample_f64(const double* p, int64_t Pp, double* out, int64_t N, uint64_t seed, uint64_t offset,
                                       const uint64_t* rng_state, void* stream) {
  if (N < 0 || Pp < 1) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!p || !out) return ZS_EINVAL;
  ZS_LAUNCH(KID_BERN_SAMPLE, k64_bern_sample, dim3(grid_for(N, 256)), dim3(256), ST, p, Pp, out, N, seed, offset, rng_state);
  ZS_CHECK_LAUNCH();
  return 0;
}

static int iw_launch64(int kid, const double* logp, int64_t ld_p, const double* logq, int64_t ld_q, int64_t B, int64_t K, int estimator,
                       double* cost_b, double* bound_b, double* coef_p, double* coef_q, const IwExt64& ext, int64_t workspace_len,
                       void* stream) {
  if (B < 0 || K < 1 || ld_p < K || ld_q < K) return ZS_EINVAL;
  if (estimator != ZS_IW_SGVB && estimator != ZS_IW_VIMCO) return ZS_EINVAL;
  if (estimator == ZS_IW_VIMCO && K < 2) return ZS_EINVAL;
  if (K > 0x7fffffff) return ZS_ENOTSUP;
  if (ext.logp_b && ext.ld_b < K) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!logp || !logq) return ZS_EINVAL;
  const unsigned grid = grid_for(B, 1);
  if (ext.mean_cost && (!ext.partials || !ext.ticket || workspace_len < (int64_t)grid)) return ZS_EINVAL;
  ZS_LAUNCH(kid, k64_iw_reduce, dim3(grid), dim3(256), ST, logp, ld_p, logq, ld_q, B, K, estimator, cost_b, bound_b, coef_p, coef_q, ext);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_iw_reduce_f64(const double* logp, int64_t ld_p, const double* logq, int64_t ld_q, int64_t B, int64_t K,
                                int estimator, double* cost_b, double* bound_b, double* coef_p, double* coef_q, void* stream) {
  const IwExt64 ext = {nullptr, 0, nullptr, 0, 1.0, nullptr, nullptr, nullptr, 0.0};
  return iw_launch64(KID_IW_REDUCE, logp, ld_p, logq, ld_q, B, K, estimator, cost_b, bound_b, coef_p, coef_q, ext, 0, stream);
}

extern "C" int zs_iw_objective_f64(const double* logp_a, int64_t ld_a, const double* logp_b, int64_t ld_b, const double* logq,
                                   int64_t ld_q, int64_t B, int64_t K, int estimator, int want_mean, double* cost_b,
                                   double* bound_b, double* coef, double* mean_cost, double* workspace, int64_t workspace_len,
                                   uint32_t* ticket, void* stream) {
  if (want_mean && !mean_cost) return ZS_EINVAL;
  if (B < 0 || K < 1) return ZS_EINVAL;
  const double inv_B = B > 0 ? 1.0 / (double)B : 0.0;
  const IwExt64 ext = {logp_b, ld_b, nullptr, 0, want_mean ? inv_B : 1.0, want_mean ? mean_cost : nullptr, workspace, ticket, inv_B};
  return iw_launch64(KID_IW_OBJECTIVE, logp_a, ld_a, logq, ld_q, B, K, estimator, cost_b, bound_b, coef, coef ? coef + B * K : nullptr, ext,
                     workspace_len, stream);
}

__global__ __launch_bounds__(256) void k64_mean_of(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
  s = bsum_d(s, sh);
  if (threadIdx.x == 0) out[0] = s / (double)n;
}

// IW1 (include/zs_hip.h) in float64: composed from the plain kernels of this file -- the Bernoulli row sums, the Normal
// term's row sums, the IW reduction over their sum and the batch mean are separate launches here (no benchmark configuration uses float64).
extern "C" int zs_bernoulli_iw_objective_f64(const double* p, int from_logits, const double* x, int64_t Px, int64_t K, int64_t R,
                                             int64_t D, const double* z, const double* pmu, int64_t Pm, const double* psigma,
                                             int64_t Ps, int64_t Dz, int psigma_is_logstd, const double* rows_a, int64_t ld_a,
                                             const double* logq, int64_t ld_q, int estimator, int want_mean, double* lp_x,
                                             double* lp_z, double* cost_b, double* bound_b, double* coef, double* mean_cost,
                                             uint64_t* acc, void* stream) {
  if (K < 1 || R < 0 || D < 1 || Px < 1 || ld_q < K || (rows_a && ld_a < K)) return ZS_EINVAL;
  if (estimator != ZS_IW_SGVB && estimator != ZS_IW_VIMCO) return ZS_EINVAL;
  if (estimator == ZS_IW_VIMCO && K < 2) return ZS_EINVAL;
  if (want_mean && (!mean_cost || !acc)) return ZS_EINVAL;
  if (R == 0) return 0;
  if (!p || !x || !logq || !lp_x || !cost_b) return ZS_EINVAL;
  if (z && (!pmu || !psigma || !lp_z || Dz < 1 || (Pm != 1 && Pm != R * Dz) || (Ps != 1 && Ps != R * Dz))) return ZS_EINVAL;
  if (Px != R * D && Px != K * R * D) return ZS_ENOTSUP;
  int rc = bern_fwd64(from_logits != 0, p, x, Px, lp_x, nullptr, K, R, D, 1, K, stream);
  if (rc != 0) return rc;
  if (z) {
    rc = zs_normal_logprob_f64(z, K * R * Dz, pmu, Pm, psigma, Ps, lp_z, K, R, Dz, 1, K, psigma_is_logstd, stream);
    if (rc != 0) return rc;
  }
  // the terms that exist, in the reference's order of addition: rows_a, lp_z, lp_x
  const double* t[3] = {nullptr, nullptr, nullptr};
  int64_t ld[3] = {K, K, K};
  int n = 0;
  if (rows_a) { t[n] = rows_a; ld[n++] = ld_a; }
  if (z) { t[n] = lp_z; ld[n++] = K; }
  t[n] = lp_x; ld[n++] = K;
  const double inv_B = 1.0 / (double)R;
  const IwExt64 ext = {t[1], ld[1], t[2], ld[2], want_mean ? inv_B : 1.0, nullptr, nullptr, nullptr, inv_B};
  rc = iw_launch64(KID_BERN_IW_OBJECTIVE, t[0], ld[0], logq, ld_q, R, K, estimator, cost_b, bound_b, coef, coef ? coef + R * K : nullptr,
                   ext, 0, stream);
  if (rc != 0 || !want_mean) return rc;
  ZS_LAUNCH(KID_BERN_IW_OBJECTIVE, k64_mean_of, dim3(1), dim3(256), ST, cost_b, R, mean_cost);       // one workgroup, fixed order
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_bernoulli_iw_objective_bwd_f64(const double* p, int from_logits, const double* x, int64_t Px, int64_t K,
                                                 int64_t R, int64_t D, const double* coef, const double* gout,
                                                 int64_t gout_stride, double* gp, const double* zq, const double* qmu,
                                                 const double* qsigma, int64_t Dq, int qsigma_is_logstd, double* gqmu,
                                                 double* gqsigma, void* stream) {
  if (K < 1 || R < 0 || D < 1 || Px < 1 || gout_stride < 0) return ZS_EINVAL;
  if (R == 0) return 0;
  if (!coef || !gout) return ZS_EINVAL;
  if (gp) {
    const int rc = bern_bwd64(from_logits != 0, p, x, Px, coef, 1, K, gp, K, R, D, stream, gout, gout_stride);
    if (rc != 0) return rc;
  }
  if (zq) {
    if (!qmu || !qsigma || Dq < 1 || !gqmu || !gqsigma) return ZS_EINVAL;
    ZS_LAUNCH(KID_BERN_IW_OBJECTIVE_BWD, k64_normal_logprob_bwd_ksum, dim3(grid_for(R * Dq, 256)), dim3(256), ST, zq, qmu, qsigma,
              coef + R * K, (int64_t)1, K, (double*)nullptr, gqmu, gqsigma, K, R * Dq, Dq, qsigma_is_logstd != 0, gout, gout_stride);
    ZS_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int zs_log_mean_exp_f64(const double* x, int64_t ld, int64_t B, int64_t K, double* out, void* stream) {
  if (B < 0 || K < 1 || ld < K) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!x || !out) return ZS_EINVAL;
  ZS_LAUNCH(KID_LME, k64_lme, dim3(grid_for(B, 1)), dim3(256), ST, x, ld, B, K, out);
  ZS_CHECK_LAUNCH();
  return 0;
}

extern "C" int zs_philox_normal_f64(double* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                    void* stream) {
  if (N < 0) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!out) return ZS_EINVAL;
  ZS_LAUNCH(KID_PHILOX, k64_philox_normal, dim3(grid_for(N, 256)), dim3(256), ST, out, N, seed, offset, rng_state);
  ZS_CHECK_LAUNCH();
  return 0;
}
