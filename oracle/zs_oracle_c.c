/*
 * zs_oracle_c.c -- plain-C CPU restatement of the hot path behind the SAME C ABI as the HIP
 * library (include/zs_hip.h), operating on HOST pointers.  TEST INFRASTRUCTURE ONLY:
 *
 *   - tests/ load it next to libzs_hip.so and call both through one ctypes wrapper, so the
 *     GPU parity tests compare the two libraries entry point by entry point;
 *   - the CPU test-suite injects it as the kernel library of the zhusuan package to exercise
 *     the host logic (Python -> C-ABI marshalling, autograd formulas) without a GPU;
 *   - the shipped package never loads it by itself.
 *
 * Parity is pinned: tests/test_oracle_c_golden.py checks these functions against the golden
 * fixtures generated from the real reference (tests/golden/gen_golden.py) and against
 * oracle/zs_oracle.py.
 *
 * Arithmetic is fp32 in the reference's op order (paths relative to /root/reference):
 *   Normal     zhusuan/distributions/normal.py:89-126
 *   Bernoulli  zhusuan/distributions/bernoulli.py:46-50,84-95
 *   group sum  zhusuan/distributions/base.py:175-176, framework/stochastic_tensor.py:160-181
 *   IW / VIMCO zhusuan/variational/importance_weighted_objective.py:16-25,123-132,152-191
 *   LME        zhusuan/utils.py:6-21
 * Serial loops, no threads, no SIMD intrinsics: clarity over speed.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/zs_hip.h"

#define C_NORM (-0.91893853320467274178f) /* -0.5*log(2*pi) */
#define BERN_EPS 1e-8f

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox4(uint64_t group, uint64_t call, uint64_t seed, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2 = (uint32_t)call, c3 = (uint32_t)(call >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static float u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * 5.9604644775390625e-08f; }
static void philox_normal4(uint64_t group, uint64_t call, uint64_t seed, float n[4]) {
  uint32_t r[4];
  philox4(group, call, seed, r);
  const double two_pi = 6.283185307179586476925;
  double u0 = u01(r[0]), u1 = u01(r[1]), u2 = u01(r[2]), u3 = u01(r[3]);
  double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
  n[0] = (float)(ra * cos(two_pi * u1));
  n[1] = (float)(ra * sin(two_pi * u1));
  n[2] = (float)(rb * cos(two_pi * u3));
  n[3] = (float)(rb * sin(two_pi * u3));
}
static float eps_at(const float* eps, int64_t i, uint64_t seed, uint64_t call) {
  if (eps) return eps[i];
  float n[4];
  philox_normal4((uint64_t)(i >> 2), call, seed, n);
  return n[i & 3];
}

/* one element of the Normal log-density, normal.py:121-124 */
static float normal_term(float x, float mu, float sigma) {
  float logstd = logf(sigma);
  float precision = expf(-2.0f * logstd);
  float d = x - mu;
  return (C_NORM - logstd) - 0.5f * precision * (d * d);
}
static float bern_term(float p, float x) { /* bernoulli.py:94 */
  return x * logf(p + BERN_EPS) + (1.0f - x) * logf((1.0f - p) + BERN_EPS);
}
static float sigmoidf_(float l) { return 1.0f / (1.0f + expf(-l)); }

int zs_abi_version(void) { return ZS_ABI_VERSION; }
const char* zs_error_string(int code) {
  if (code == 0) return "success";
  if (code == ZS_EINVAL) return "zs(oracle): invalid argument";
  if (code == ZS_ENOTSUP) return "zs(oracle): unsupported configuration";
  return "zs(oracle): unknown error";
}

/* ------------------------------------------------------------------ K1 */
#define RNG_STATE(rs, seed, offset) do { if (rs) { seed = (rs)[0]; offset += (rs)[1]; } } while (0)

int zs_normal_sample_logprob_f32(const float* mu, const float* sigma, const float* eps, uint64_t seed,
                                 uint64_t offset, const uint64_t* rng_state, float* z, float* lp, int64_t K,
                                 int64_t M, int64_t D, int64_t sk, int64_t sr, void* stream) {
  (void)stream;
  RNG_STATE(rng_state, seed, offset);
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!mu || !sigma || !z) return ZS_EINVAL;
  const int64_t R = M / D;
  for (int64_t k = 0; k < K; ++k)
    for (int64_t r = 0; r < R; ++r) {
      float acc = 0.f;
      for (int64_t d = 0; d < D; ++d) {
        const int64_t m = r * D + d, i = k * M + m;
        const float e = eps_at(eps, i, seed, offset);
        const float prod = sigma[m] * e; /* separate mul and add: two roundings, normal.py:105 */
        const float zz = mu[m] + prod;
        z[i] = zz;
        acc += normal_term(zz, mu[m], sigma[m]);
      }
      if (lp) lp[k * sk + r * sr] = acc;
    }
  return 0;
}

int zs_normal_sample_logprob_bwd_f32(const float* sigma, const float* eps, uint64_t seed, uint64_t offset,
                                     const uint64_t* rng_state, const float* gz, const float* glp, int64_t gsk,
                                     int64_t gsr, float* gmu, float* gsigma, int64_t K, int64_t M, int64_t D,
                                     void* stream) {
  (void)stream;
  RNG_STATE(rng_state, seed, offset);
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!sigma || !gmu || !gsigma) return ZS_EINVAL;
  for (int64_t m = 0; m < M; ++m) {
    const int64_t r = m / D;
    float a = 0.f, b = 0.f, g = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      const int64_t i = k * M + m;
      if (gz) {
        a += gz[i];
        b += gz[i] * eps_at(eps, i, seed, offset);
      }
      if (glp) g += glp[k * gsk + r * gsr];
    }
    gmu[m] = a;
    gsigma[m] = b - g / sigma[m];
  }
  return 0;
}

/* ------------------------------------------------------------------ K2 */
int zs_normal_logprob_f32(const float* x, int64_t Px, const float* mu, int64_t Pm, const float* sigma,
                          int64_t Ps, float* lp, int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr,
                          void* stream) {
  (void)stream;
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !lp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  for (int64_t k = 0; k < K; ++k)
    for (int64_t r = 0; r < R; ++r) {
      float acc = 0.f;
      for (int64_t d = 0; d < D; ++d) {
        const int64_t i = (k * R + r) * D + d;
        acc += normal_term(x[i % Px], mu[i % Pm], sigma[i % Ps]);
      }
      lp[k * sk + r * sr] = acc;
    }
  return 0;
}

static void normal_partials(float x, float mu, float sigma, float g, float* gx, float* gmu, float* gsig) {
  const float prec = expf(-2.0f * logf(sigma));
  const float d = x - mu;
  const float t = g * prec * d;
  *gx = -t;
  *gmu = t;
  *gsig = g * (prec * d * d - 1.0f) / sigma;
}

int zs_normal_logprob_bwd_f32(const float* x, int64_t Px, const float* mu, int64_t Pm, const float* sigma,
                              int64_t Ps, const float* glp, int64_t gsk, int64_t gsr, float* gx, float* gmu,
                              float* gsigma, int64_t K, int64_t R, int64_t D, void* stream) {
  (void)stream;
  if (K < 1 || R < 0 || D < 1 || Px < 1 || Pm < 1 || Ps < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  for (int64_t i = 0; i < N; ++i) {
    const int64_t row = i / D, k = row / R, r = row % R;
    float a, b, c;
    normal_partials(x[i % Px], mu[i % Pm], sigma[i % Ps], glp[k * gsk + r * gsr], &a, &b, &c);
    if (gx) gx[i] = a;
    if (gmu) gmu[i] = b;
    if (gsigma) gsigma[i] = c;
  }
  return 0;
}

int zs_normal_logprob_bwd_ksum_f32(const float* x, const float* mu, const float* sigma, const float* glp,
                                   int64_t gsk, int64_t gsr, float* gx, float* gmu, float* gsigma, int64_t K,
                                   int64_t R, int64_t D, void* stream) {
  (void)stream;
  if (K < 1 || R < 0 || D < 1) return ZS_EINVAL;
  const int64_t M = R * D;
  if (M == 0) return 0;
  if (!x || !mu || !sigma || !glp) return ZS_EINVAL;
  for (int64_t m = 0; m < M; ++m) {
    const int64_t r = m / D;
    float sa = 0.f, sb = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      float a, b, c;
      normal_partials(x[k * M + m], mu[m], sigma[m], glp[k * gsk + r * gsr], &a, &b, &c);
      if (gx) gx[k * M + m] = a;
      sa += b;
      sb += c;
    }
    if (gmu) gmu[m] = sa;
    if (gsigma) gsigma[m] = sb;
  }
  return 0;
}

/* ------------------------------------------------------------------ K3 / K5 */
static int bern_fwd(const float* p, int logits, const float* x, int64_t Px, float* lp, float* probs_out,
                    int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr) {
  if (K < 1 || R < 0 || D < 1 || Px < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!p || !x || !lp) return ZS_EINVAL;
  if (N % Px) return ZS_EINVAL;
  for (int64_t k = 0; k < K; ++k)
    for (int64_t r = 0; r < R; ++r) {
      float acc = 0.f;
      for (int64_t d = 0; d < D; ++d) {
        const int64_t i = (k * R + r) * D + d;
        float pv = p[i];
        if (logits) {
          pv = sigmoidf_(pv);
          if (probs_out) probs_out[i] = pv;
        }
        acc += bern_term(pv, x[i % Px]);
      }
      lp[k * sk + r * sr] = acc;
    }
  return 0;
}
static int bern_bwd(const float* p, int logits, const float* x, int64_t Px, const float* glp, int64_t gsk,
                    int64_t gsr, float* gp, int64_t K, int64_t R, int64_t D) {
  if (K < 1 || R < 0 || D < 1 || Px < 1) return ZS_EINVAL;
  const int64_t N = K * R * D;
  if (N == 0) return 0;
  if (!p || !x || !glp || !gp) return ZS_EINVAL;
  if (N % Px) return ZS_EINVAL;
  for (int64_t i = 0; i < N; ++i) {
    const int64_t row = i / D, k = row / R, r = row % R;
    float pv = p[i], scale = 1.0f;
    if (logits) {
      pv = sigmoidf_(pv);
      scale = pv * (1.0f - pv);
    }
    const float xv = x[i % Px];
    gp[i] = glp[k * gsk + r * gsr] * (xv / (pv + BERN_EPS) - (1.0f - xv) / ((1.0f - pv) + BERN_EPS)) * scale;
  }
  return 0;
}
int zs_bernoulli_logprob_f32(const float* p, const float* x, int64_t Px, float* lp, int64_t K, int64_t R,
                             int64_t D, int64_t sk, int64_t sr, void* stream) {
  (void)stream;
  return bern_fwd(p, 0, x, Px, lp, NULL, K, R, D, sk, sr);
}
int zs_bernoulli_logprob_bwd_f32(const float* p, const float* x, int64_t Px, const float* glp, int64_t gsk,
                                 int64_t gsr, float* gp, int64_t K, int64_t R, int64_t D, void* stream) {
  (void)stream;
  return bern_bwd(p, 0, x, Px, glp, gsk, gsr, gp, K, R, D);
}
int zs_bernoulli_logits_logprob_f32(const float* logits, const float* x, int64_t Px, float* lp,
                                    float* probs_out, int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr,
                                    void* stream) {
  (void)stream;
  return bern_fwd(logits, 1, x, Px, lp, probs_out, K, R, D, sk, sr);
}
int zs_bernoulli_logits_logprob_bwd_f32(const float* logits, const float* x, int64_t Px, const float* glp,
                                        int64_t gsk, int64_t gsr, float* glogits, int64_t K, int64_t R,
                                        int64_t D, void* stream) {
  (void)stream;
  return bern_bwd(logits, 1, x, Px, glp, gsk, gsr, glogits, K, R, D);
}
int zs_bernoulli_sample_f32(const float* p, int64_t Pp, float* out, int64_t N, uint64_t seed, uint64_t offset,
                            const uint64_t* rng_state, void* stream) {
  (void)stream;
  RNG_STATE(rng_state, seed, offset);
  if (N < 0 || Pp < 1) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!p || !out) return ZS_EINVAL;
  for (int64_t i = 0; i < N; ++i) {
    uint32_t r[4];
    philox4((uint64_t)(i >> 2), offset, seed, r);
    out[i] = u01(r[i & 3]) < p[i % Pp] ? 1.0f : 0.0f;
  }
  return 0;
}

/* ------------------------------------------------------------------ K4 */
static float lme_row(const float* v, int64_t n) { /* zhusuan/utils.py:17-18 */
  float mx = v[0];
  for (int64_t i = 1; i < n; ++i) mx = v[i] > mx ? v[i] : mx;
  float s = 0.f;
  for (int64_t i = 0; i < n; ++i) s += expf(v[i] - mx);
  return logf(s / (float)n) + mx;
}

int zs_iw_reduce_f32(const float* logp, int64_t ld_p, const float* logq, int64_t ld_q, int64_t B, int64_t K,
                     int estimator, float* cost_b, float* bound_b, float* coef_p, float* coef_q, void* stream) {
  (void)stream;
  if (B < 0 || K < 1 || ld_p < K || ld_q < K) return ZS_EINVAL;
  if (estimator != ZS_IW_SGVB && estimator != ZS_IW_VIMCO) return ZS_EINVAL;
  if (estimator == ZS_IW_VIMCO && K < 2) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!logp || !logq) return ZS_EINVAL;
  float* l = (float*)malloc(sizeof(float) * (size_t)K * 2);
  if (!l) return ZS_ENOTSUP;
  float* tmp = l + K;
  for (int64_t b = 0; b < B; ++b) {
    const float* pp = logp + b * ld_p;
    const float* qq = logq + b * ld_q;
    float mx = -INFINITY, sumL = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      l[k] = pp[k] - qq[k];
      mx = l[k] > mx ? l[k] : mx;
      sumL += l[k];
    }
    float S = 0.f;
    for (int64_t k = 0; k < K; ++k) S += expf(l[k] - mx);
    const float bound = logf(S / (float)K) + mx;
    float cost = 0.f;
    for (int64_t j = 0; j < K; ++j) {
      const float wt = expf(l[j] - mx) / S; /* compute_iw_term, :21-24 */
      float cq = wt;
      cost -= wt * l[j];
      if (estimator == ZS_IW_VIMCO) {
        /* column j of the [K, K] matrix of :184-185: l with entry j replaced by the mean of the others */
        for (int64_t i = 0; i < K; ++i) tmp[i] = l[i];
        tmp[j] = (sumL - l[j]) / (float)(K - 1);
        const float signal = bound - lme_row(tmp, K); /* :186-187 */
        cost -= qq[j] * signal;                       /* :188 */
        cq = wt - signal;
      }
      if (coef_p) coef_p[b * K + j] = -wt;
      if (coef_q) coef_q[b * K + j] = cq;
    }
    if (cost_b) cost_b[b] = cost;
    if (bound_b) bound_b[b] = bound;
  }
  free(l);
  return 0;
}

int zs_log_mean_exp_f32(const float* x, int64_t ld, int64_t B, int64_t K, float* out, void* stream) {
  (void)stream;
  if (B < 0 || K < 1 || ld < K) return ZS_EINVAL;
  if (B == 0) return 0;
  if (!x || !out) return ZS_EINVAL;
  for (int64_t b = 0; b < B; ++b) out[b] = lme_row(x + b * ld, K);
  return 0;
}

int zs_philox_normal_f32(float* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                         void* stream) {
  (void)stream;
  RNG_STATE(rng_state, seed, offset);
  if (N < 0) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!out) return ZS_EINVAL;
  for (int64_t i = 0; i < N; ++i) out[i] = eps_at(NULL, i, seed, offset);
  return 0;
}

/* timing hooks exist only in the HIP library */
int zs_prof_enable(int on) { (void)on; return ZS_ENOTSUP; }
int zs_prof_kernel_id(const char* entry_point) { (void)entry_point; return ZS_ENOTSUP; }
int zs_prof_query(int kernel_id, double* total_ms, double* min_ms, double* max_ms, int64_t* count) {
  (void)kernel_id; (void)total_ms; (void)min_ms; (void)max_ms; (void)count;
  return ZS_ENOTSUP;
}

/* raw Philox words, for the known-answer test of the generator itself */
void zs_oracle_philox4x32_10(uint64_t group, uint64_t call, uint64_t seed, uint32_t out[4]) {
  philox4(group, call, seed, out);
}
