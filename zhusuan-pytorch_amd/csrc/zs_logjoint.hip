// One-launch pieces for the launch-bound configurations (include/zs_hip.h: LJ1, MS1, PL1).
//
// At the VAE (B = 512) and BNN (B = 512, K = 10) shapes every kernel of the step occupies ~4 us of the stream whatever it
// computes, so what counts is the NUMBER of launches.  The reference walks the nodes of a BayesianNet in Python loops
// (ELBO.log_joint, zhusuan/variational/elbo.py:58-79; the re-read of every latent, elbo.py:122); round 2 had one launch
// per node and direction.  Here:
//   LJ1  all log-probs of all nodes + their weighted sum = the scalar objective: one launch forward, one backward
//        (pointer table in the kernel arguments, as the Adam update does for 32 tensors);
//   MS1  the fused sample + log-density (K1) of several Normal nodes: one launch forward, one backward;
//   PL1  the BNN caller's particle-batched dense layer (bias column, 1/sqrt(n), ReLU fused): one launch each way.
// All three are HBM-trivial (a few hundred KB): they are written for few dependent rounds of loads and for deterministic
// sums (fixed combination order), not for bandwidth.  Templated on float / double.
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
  static __device__ __forceinline__ float rsqrt_n(int64_t n) { return sqrtf((float)n); }
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
  static __device__ __forceinline__ double rsqrt_n(int64_t n) { return sqrt((double)n); }
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

// ================================================================ LJ1
constexpr int LJ_BLOCK = 256;
constexpr unsigned LJ_MAX_BLOCKS = 2048;      // 3 doubles of workspace per workgroup: 6144 <= ZS_LJ_WORKSPACE

template <typename T>
struct LJTerm {
  const T *x, *a, *b;
  T *gx, *ga, *gb;
  int64_t n, px, pa, pb;
  double coef;
  int family;
  unsigned block0, nblocks;   // the element-wise workgroups of this term
  unsigned vec;               // bit o: operand o (0 = x, 1 = a, 2 = b) can be read / written 4 elements at a time
};
struct LJFold {                // backward: the gradient of an operand of period 1 < P < n, one thread per element of it
  int term, operand;
  unsigned block0, nblocks;
};
template <typename T>
struct LJArgs {
  LJTerm<T> t[ZS_LJ_MAX_TERMS];
  LJFold f[3 * ZS_LJ_MAX_TERMS];
  int n_terms, n_folds;
  unsigned n_elem_blocks, n_blocks;
};

template <typename T>
__device__ __forceinline__ void lj_load4(const T* __restrict__ p, int64_t P, int64_t n, bool vec, int64_t i0, int cnt, T v[4]) {
  if (P == 1) {
    const T s = p[0];
    v[0] = v[1] = v[2] = v[3] = s;
    return;
  }
  if (vec && cnt == 4) {                       // P % 4 == 0 and 4-element aligned base: the group does not wrap
    const int64_t j = (P == n) ? i0 : mod_fast(i0, P);
    const V4<T> q = *reinterpret_cast<const V4<T>*>(p + j);
    v[0] = q.v[0]; v[1] = q.v[1]; v[2] = q.v[2]; v[3] = q.v[3];
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = i0 + (j < cnt ? j : 0);  // clamped: unconditional loads
    v[j] = p[(P == n) ? i : mod_fast(i, P)];
  }
}
template <typename T>
__device__ __forceinline__ void lj_store4(T* __restrict__ p, bool vec, int64_t i0, int cnt, const T v[4]) {
  if (vec && cnt == 4) {
    V4<T> q;
    q.v[0] = v[0]; q.v[1] = v[1]; q.v[2] = v[2]; q.v[3] = v[3];
    *reinterpret_cast<V4<T>*>(p + i0) = q;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < cnt) p[i0 + j] = v[j];
}

// which term owns this workgroup (uniform; the table lives in the kernel-argument segment: scalar loads)
template <typename T>
__device__ __forceinline__ int lj_term_of(const LJArgs<T>& A, unsigned blk) {
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (A.t[j].nblocks && blk >= A.t[j].block0) ti = j;
  return __builtin_amdgcn_readfirstlane(ti);
}

// log-density terms of 4 consecutive elements
template <typename T>
__device__ __forceinline__ T lj_terms4(int family, const T x[4], const T a[4], const T b[4], int cnt) {
  typedef Mth<T> M;
  T s = (T)0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    T t;
    if (family == ZS_LJ_ROWS) {
      t = x[j];
    } else if (family == ZS_LJ_NORMAL || family == ZS_LJ_NORMAL_LOGSTD) {
      const T sg = M::sigma_of(b[j], family == ZS_LJ_NORMAL_LOGSTD);
      T logstd, prec;
      M::parts(sg, logstd, prec);
      t = M::normal_term(x[j] - a[j], logstd, prec);
    } else {
      const T p = family == ZS_LJ_BERNOULLI_LOGITS ? M::sigmoid(a[j]) : a[j];
      t = M::bern_term(p, x[j]);
    }
    if (j < cnt) s += t;
  }
  return s;
}

template <typename T>
__global__ __launch_bounds__(LJ_BLOCK) void k_logjoint_fwd(const LJArgs<T> A, T* __restrict__ out, double* __restrict__ ws,
                                                           unsigned* __restrict__ ticket) {
  __shared__ double sh[4];
  __shared__ bool last;
  const int ti = lj_term_of(A, blockIdx.x);
  const LJTerm<T>& t = A.t[ti];
  const int family = t.family;
  const int64_t n = t.n, groups = (n + 3) >> 2;
  const bool vx = t.vec & 1u, va = t.vec & 2u, vb = t.vec & 4u;
  const bool has_a = family != ZS_LJ_ROWS, has_b = family == ZS_LJ_NORMAL || family == ZS_LJ_NORMAL_LOGSTD;
  double acc = 0.0;
  for (int64_t g = (int64_t)(blockIdx.x - t.block0) * LJ_BLOCK + threadIdx.x; g < groups; g += (int64_t)t.nblocks * LJ_BLOCK) {
    const int64_t i0 = g << 2;
    const int cnt = n - i0 < 4 ? (int)(n - i0) : 4;
    T x[4], a[4] = {(T)0, (T)0, (T)0, (T)0}, b[4] = {(T)1, (T)1, (T)1, (T)1};
    lj_load4(t.x, t.px, n, vx, i0, cnt, x);
    if (has_a) lj_load4(t.a, t.pa, n, va, i0, cnt, a);
    if (has_b) lj_load4(t.b, t.pb, n, vb, i0, cnt, b);
    acc += (double)lj_terms4<T>(family, x, a, b, cnt);
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    __hip_atomic_store(ws + blockIdx.x, t.coef * acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // release: the partial is visible to whoever observes this increment; acquire: the last arrival sees all of them
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (tk == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    double s = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 64)
      s += __hip_atomic_load(ws + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = wave_sum_d(s);
    if (threadIdx.x == 0) {
      out[0] = (T)s;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// d term / d (x, a, b) of one element, times gc = g * coef
template <typename T>
__device__ __forceinline__ void lj_partials(int family, T x, T a, T b, T gc, T& dx, T& da, T& db) {
  typedef Mth<T> M;
  if (family == ZS_LJ_NORMAL || family == ZS_LJ_NORMAL_LOGSTD) {
    const bool ls = family == ZS_LJ_NORMAL_LOGSTD;
    const T sg = M::sigma_of(b, ls);
    T logstd, prec;
    M::parts(sg, logstd, prec);
    const T d = x - a;
    const T u = gc * prec * d;
    dx = -u;
    da = u;
    const T v = gc * (prec * d * d - (T)1);
    db = ls ? v : v / sg;                       // d/d log std = sigma * d/d sigma
  } else {                                      // Bernoulli: d/d probs, or d/d logits = d/dp * p * (1 - p)
    dx = (T)0;
    db = (T)0;
    if (family == ZS_LJ_BERNOULLI_LOGITS) {
      const T p = M::sigmoid(a);
      da = gc * M::bern_dp(p, x) * p * ((T)1 - p);
    } else {
      da = gc * M::bern_dp(a, x);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(LJ_BLOCK) void k_logjoint_bwd(const LJArgs<T> A, const T* __restrict__ gout, T* __restrict__ gcoef,
                                                           double* __restrict__ ws, unsigned* __restrict__ ticket) {
  __shared__ double sh[4];
  __shared__ bool last;
  const T g = gout[0];
  if (blockIdx.x < A.n_elem_blocks) {
    // ---- element-wise role: full-size gradients are written here, gradients of scalar operands leave as partials
    const int ti = lj_term_of(A, blockIdx.x);
    const LJTerm<T>& t = A.t[ti];
    const int family = t.family;
    const int64_t n = t.n, groups = (n + 3) >> 2;
    const bool vx = t.vec & 1u, va = t.vec & 2u, vb = t.vec & 4u;
    const bool has_b = family == ZS_LJ_NORMAL || family == ZS_LJ_NORMAL_LOGSTD;
    const T gc = (T)(t.coef * (double)g);
    const bool full_x = t.gx && t.px == n, full_a = t.ga && t.pa == n, full_b = t.gb && t.pb == n;
    const bool sc_x = t.gx && t.px == 1 && n > 1, sc_a = t.ga && t.pa == 1 && n > 1, sc_b = t.gb && t.pb == 1 && n > 1;
    double sx = 0.0, sa = 0.0, sb = 0.0;
    for (int64_t gi = (int64_t)(blockIdx.x - t.block0) * LJ_BLOCK + threadIdx.x; gi < groups; gi += (int64_t)t.nblocks * LJ_BLOCK) {
      const int64_t i0 = gi << 2;
      const int cnt = n - i0 < 4 ? (int)(n - i0) : 4;
      T x[4], a[4], b[4] = {(T)1, (T)1, (T)1, (T)1}, dx[4], da[4], db[4];
      lj_load4(t.x, t.px, n, vx, i0, cnt, x);
      lj_load4(t.a, t.pa, n, va, i0, cnt, a);
      if (has_b) lj_load4(t.b, t.pb, n, vb, i0, cnt, b);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lj_partials<T>(family, x[j], a[j], b[j], gc, dx[j], da[j], db[j]);
        if (j < cnt) { sx += (double)dx[j]; sa += (double)da[j]; sb += (double)db[j]; }
      }
      if (full_x) lj_store4(t.gx, vx, i0, cnt, dx);
      if (full_a) lj_store4(t.ga, va, i0, cnt, da);
      if (full_b) lj_store4(t.gb, vb, i0, cnt, db);
    }
    if (sc_x) sx = block_sum_256(sx, sh);
    if (sc_a) sa = block_sum_256(sa, sh);
    if (sc_b) sb = block_sum_256(sb, sh);
    if (threadIdx.x == 0) {
      if (sc_x) __hip_atomic_store(ws + 3 * blockIdx.x + 0, sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sc_a) __hip_atomic_store(ws + 3 * blockIdx.x + 1, sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sc_b) __hip_atomic_store(ws + 3 * blockIdx.x + 2, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (A.n_folds > 0) {
    // ---- fold role: operand of period 1 < P < n: thread j adds the contributions of elements j, j + P, j + 2P, ... in order
    int fi = 0;
    for (int j = 1; j < A.n_folds; ++j)
      if (blockIdx.x >= A.f[j].block0) fi = j;
    fi = __builtin_amdgcn_readfirstlane(fi);
    const LJFold& f = A.f[fi];
    const LJTerm<T>& t = A.t[f.term];
    const int family = t.family;
    const T gc = (T)(t.coef * (double)g);
    const int64_t n = t.n;
    const int64_t P = f.operand == 0 ? t.px : (f.operand == 1 ? t.pa : t.pb);
    T* __restrict__ dst = f.operand == 0 ? t.gx : (f.operand == 1 ? t.ga : t.gb);
    const bool has_b = family == ZS_LJ_NORMAL || family == ZS_LJ_NORMAL_LOGSTD;
    for (int64_t j = (int64_t)(blockIdx.x - f.block0) * LJ_BLOCK + threadIdx.x; j < P; j += (int64_t)f.nblocks * LJ_BLOCK) {
      T acc = (T)0;
      for (int64_t i = j; i < n; i += P) {
        const T x = t.x[t.px == n ? i : mod_fast(i, t.px)];
        const T a = t.a[t.pa == n ? i : mod_fast(i, t.pa)];
        const T b = has_b ? t.b[t.pb == n ? i : mod_fast(i, t.pb)] : (T)1;
        T dx, da, db;
        lj_partials<T>(family, x, a, b, gc, dx, da, db);
        acc += f.operand == 0 ? dx : (f.operand == 1 ? da : db);
      }
      dst[j] = acc;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = (tk == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    for (int ti = 0; ti < A.n_terms; ++ti) {
      const LJTerm<T>& t = A.t[ti];
      if (gcoef && threadIdx.x == 0) gcoef[ti] = (T)(t.coef * (double)g);
      if (t.family == ZS_LJ_ROWS || t.n <= 1) continue;
      for (int o = 0; o < 3; ++o) {
        T* dst = o == 0 ? t.gx : (o == 1 ? t.ga : t.gb);
        const int64_t P = o == 0 ? t.px : (o == 1 ? t.pa : t.pb);
        if (!dst || P != 1) continue;
        double s = 0.0;
        for (unsigned i = threadIdx.x; i < t.nblocks; i += 64)
          s += __hip_atomic_load(ws + 3 * (t.block0 + i) + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = wave_sum_d(s);
        if (threadIdx.x == 0) dst[0] = (T)s;
      }
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <typename T>
bool lj_aligned(const void* p) { return (((uintptr_t)p) & (4 * sizeof(T) - 1)) == 0; }

// validates the host table and lays the workgroups out; `backward`: ZS_LJ_ROWS terms get no workgroups
template <typename T>
int lj_build(const zs_lj_term* terms, int n_terms, bool backward, LJArgs<T>& A) {
  if (!terms || n_terms < 1 || n_terms > ZS_LJ_MAX_TERMS) return n_terms > ZS_LJ_MAX_TERMS ? ZS_ENOTSUP : ZS_EINVAL;
  memset(&A, 0, sizeof(A));
  A.n_terms = n_terms;
  int64_t want[ZS_LJ_MAX_TERMS];
  int64_t total = 0;
  for (int i = 0; i < n_terms; ++i) {
    const zs_lj_term& s = terms[i];
    LJTerm<T>& t = A.t[i];
    if (s.family < ZS_LJ_ROWS || s.family > ZS_LJ_BERNOULLI_LOGITS || s.n < 0) return ZS_EINVAL;
    const bool rows = s.family == ZS_LJ_ROWS, normal = s.family == ZS_LJ_NORMAL || s.family == ZS_LJ_NORMAL_LOGSTD;
    t.family = s.family;
    t.n = s.n;
    t.coef = s.coef;
    t.x = (const T*)s.x; t.a = (const T*)s.a; t.b = (const T*)s.b;
    t.px = rows ? s.n : s.px; t.pa = s.pa; t.pb = s.pb;
    want[i] = 0;
    if (s.n == 0) continue;
    if (!s.x || t.px < 1 || s.n % t.px) return ZS_EINVAL;
    if (!rows && (!s.a || s.pa < 1 || s.n % s.pa)) return ZS_EINVAL;
    if (normal && (!s.b || s.pb < 1 || s.n % s.pb)) return ZS_EINVAL;
    if (rows) { t.pa = t.pb = 1; }
    if (!normal) t.pb = 1;
    if (backward) {
      t.gx = (T*)s.gx; t.ga = (T*)s.ga; t.gb = (T*)s.gb;
      if (rows) { t.gx = t.ga = t.gb = nullptr; }
      if (!normal) {
        if (t.gx) return ZS_ENOTSUP;            // gradient w.r.t. the Bernoulli observation is not provided
        t.gb = nullptr;
      }
      if (rows || !(t.gx || t.ga || t.gb)) continue;     // nothing to compute element-wise for this term
    }
    t.vec = 0;
    if ((t.px % 4) == 0 && lj_aligned<T>(t.x) && (!backward || !t.gx || lj_aligned<T>(t.gx))) t.vec |= 1u;
    if (!rows && (t.pa % 4) == 0 && lj_aligned<T>(t.a) && (!backward || !t.ga || lj_aligned<T>(t.ga))) t.vec |= 2u;
    if (normal && (t.pb % 4) == 0 && lj_aligned<T>(t.b) && (!backward || !t.gb || lj_aligned<T>(t.gb))) t.vec |= 4u;
    // a thread takes ~4 groups of 4 elements before it strides
    want[i] = (((s.n + 3) / 4) + LJ_BLOCK * 4 - 1) / (LJ_BLOCK * 4);
    total += want[i];
  }
  int64_t fold_want[3 * ZS_LJ_MAX_TERMS];
  int64_t fold_total = 0;
  if (backward) {
    for (int i = 0; i < n_terms; ++i) {
      const LJTerm<T>& t = A.t[i];
      if (t.family == ZS_LJ_ROWS || t.n == 0) continue;
      for (int o = 0; o < 3; ++o) {
        const T* dst = o == 0 ? t.gx : (o == 1 ? t.ga : t.gb);
        const int64_t P = o == 0 ? t.px : (o == 1 ? t.pa : t.pb);
        if (!dst || P == 1 || P == t.n) continue;
        LJFold& f = A.f[A.n_folds];
        f.term = i;
        f.operand = o;
        fold_want[A.n_folds] = (P + LJ_BLOCK - 1) / LJ_BLOCK;
        fold_total += fold_want[A.n_folds];
        ++A.n_folds;
      }
    }
  }
  // cap the grid (the workspace holds 3 doubles per element-wise workgroup); scale the shares down, at least 1 each
  const int64_t cap_e = fold_total ? LJ_MAX_BLOCKS / 2 : LJ_MAX_BLOCKS, cap_f = LJ_MAX_BLOCKS / 2;
  unsigned blk = 0;
  for (int i = 0; i < n_terms; ++i) {
    if (!want[i]) continue;
    int64_t nb = total > cap_e ? (want[i] * cap_e) / total : want[i];
    if (nb < 1) nb = 1;
    A.t[i].block0 = blk;
    A.t[i].nblocks = (unsigned)nb;
    blk += (unsigned)nb;
  }
  A.n_elem_blocks = blk;
  for (int j = 0; j < A.n_folds; ++j) {
    int64_t nb = fold_total > cap_f ? (fold_want[j] * cap_f) / fold_total : fold_want[j];
    if (nb < 1) nb = 1;
    A.f[j].block0 = blk;
    A.f[j].nblocks = (unsigned)nb;
    blk += (unsigned)nb;
  }
  A.n_blocks = blk;
  return 0;
}

template <typename T>
int logjoint_fwd(const zs_lj_term* terms, int n_terms, T* out, double* ws, int64_t ws_len, uint32_t* ticket, void* stream) {
  LJArgs<T> A;
  const int rc = lj_build<T>(terms, n_terms, false, A);
  if (rc) return rc;
  if (!out || !ws || !ticket) return ZS_EINVAL;
  if (A.n_blocks == 0) {        // every term is empty: the objective is 0 (one tiny launch keeps the call asynchronous)
    A.t[0].block0 = 0; A.t[0].nblocks = 1; A.n_blocks = A.n_elem_blocks = 1;
  }
  if (ws_len < (int64_t)A.n_blocks + 8) return ZS_EINVAL;
  ZS_LAUNCH(KID_LOGJOINT, (k_logjoint_fwd<T>), dim3(A.n_blocks), dim3(LJ_BLOCK), (hipStream_t)stream, A, out, ws, (unsigned*)ticket);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int logjoint_bwd(const zs_lj_term* terms, int n_terms, const T* gout, T* gcoef, double* ws, int64_t ws_len, uint32_t* ticket,
                 void* stream) {
  LJArgs<T> A;
  const int rc = lj_build<T>(terms, n_terms, true, A);
  if (rc) return rc;
  if (!gout || !ws || !ticket) return ZS_EINVAL;
  if (A.n_blocks == 0) {
    if (!gcoef) return 0;
    A.n_blocks = 1;              // only the per-term scalars are wanted: one workgroup that owns neither elements nor a fold job
  }
  if (ws_len < 3 * (int64_t)A.n_blocks + 8) return ZS_EINVAL;
  ZS_LAUNCH(KID_LOGJOINT_BWD, (k_logjoint_bwd<T>), dim3(A.n_blocks), dim3(LJ_BLOCK), (hipStream_t)stream, A, gout, gcoef, ws,
            (unsigned*)ticket);
  ZS_CHECK_LAUNCH();
  return 0;
}

// ================================================================ MS1
template <typename T>
struct MSTerm {
  const T *mu, *sigma, *eps;
  T *z, *lp;
  int64_t K, M, D, R, sk, sr;
  uint64_t offset;
  int ls;
  int64_t row0;               // forward: first wavefront (= row) of this term
  const T *gz, *glp;
  int64_t gsk, gsr;
  T *gmu, *gsigma;
  unsigned block0, nblocks;   // backward: workgroups of this term (256 parameter elements each)
};
template <typename T>
struct MSArgs {
  MSTerm<T> t[ZS_MS_MAX_TERMS];
  int n_terms;
  int64_t total_rows;
};

template <typename T>
__device__ __forceinline__ T ms_mul_add_2round(T m, T s, T e) {
#pragma clang fp contract(off)   // z = mean + std * eps rounds twice like the reference's separate mul and add (normal.py:105)
  const T prod = s * e;
  return m + prod;
}
template <typename T>
__device__ __forceinline__ T wave_sum_t(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  return v;
}

// one wavefront per (term, particle, row): lanes walk the Philox groups (4 consecutive flat elements) that touch the row
template <typename T>
__global__ __launch_bounds__(256) void k_normal_sample_multi(const MSArgs<T> A, uint64_t seed, const uint64_t* __restrict__ rs,
                                                             uint64_t* __restrict__ rng_used) {
  typedef Mth<T> Mh;
  uint64_t base = 0;
  if (rs) { seed = rs[0]; base = rs[1]; }
  if (rng_used && blockIdx.x == 0 && threadIdx.x == 0) { rng_used[0] = seed; rng_used[1] = base; }
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= A.total_rows) return;
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (w >= A.t[j].row0) ti = j;
  ti = __builtin_amdgcn_readfirstlane(ti);         // a wavefront serves ONE term: its Philox call id is wave-uniform
  const MSTerm<T>& t = A.t[ti];
  int64_t k, r;
  divmod(w - t.row0, t.R, k, r);
  const uint64_t call = base + t.offset;
  const bool ls = t.ls != 0;
  const int64_t start = k * t.M + r * t.D, end = start + t.D;
  T acc = (T)0;
  for (int64_t g = (start >> 2) + lane; g <= ((end - 1) >> 2); g += 64) {
    const int64_t i0 = g << 2;
    float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!t.eps) n4 = philox_normal4((uint64_t)g, call, seed);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = i0 + j;
      if (i < start || i >= end) continue;
      const int64_t m = i - k * t.M;
      const T e = t.eps ? t.eps[i] : (T)f4_get(n4, j);
      const T mu = t.mu[m], sg = Mh::sigma_of(t.sigma[m], ls);
      const T zz = ms_mul_add_2round<T>(mu, sg, e);
      t.z[i] = zz;
      if (t.lp) {
        T logstd, prec;
        Mh::parts(sg, logstd, prec);
        acc += Mh::normal_term(zz - mu, logstd, prec);
      }
    }
  }
  if (t.lp) {
    acc = wave_sum_t<T>(acc);
    if (lane == 0) t.lp[k * t.sk + r * t.sr] = acc;
  }
}

// backward: one thread per parameter element, the K particles in order; workgroups belong to ONE term (uniform call id)
template <typename T>
__global__ __launch_bounds__(256) void k_normal_sample_multi_bwd(const MSArgs<T> A, uint64_t seed, const uint64_t* __restrict__ rs) {
  typedef Mth<T> Mh;
  uint64_t base = 0;
  if (rs) { seed = rs[0]; base = rs[1]; }
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (A.t[j].nblocks && blockIdx.x >= A.t[j].block0) ti = j;
  ti = __builtin_amdgcn_readfirstlane(ti);
  const MSTerm<T>& t = A.t[ti];
  const int64_t m = (int64_t)(blockIdx.x - t.block0) * 256 + threadIdx.x;
  if (m >= t.M) return;
  const uint64_t call = base + t.offset;
  const int64_t r = m / t.D;
  T a = (T)0, b = (T)0, g = (T)0;
  for (int64_t k = 0; k < t.K; ++k) {
    const int64_t i = k * t.M + m;
    if (t.gz) {
      const T gzv = t.gz[i];
      const T e = t.eps ? t.eps[i] : (T)f4_get(philox_normal4((uint64_t)(i >> 2), call, seed), (int)(i & 3));
      a += gzv;
      b += gzv * e;
    }
    if (t.glp) g += t.glp[k * t.gsk + r * t.gsr];
  }
  t.gmu[m] = a;
  const T sg = Mh::sigma_of(t.sigma[m], t.ls != 0);
  t.gsigma[m] = t.ls ? b * sg - g : b - g / sg;
}

template <typename T>
int ms_build(const zs_ms_term* terms, int n_terms, bool backward, MSArgs<T>& A, unsigned& grid) {
  if (!terms || n_terms < 1) return ZS_EINVAL;
  if (n_terms > ZS_MS_MAX_TERMS) return ZS_ENOTSUP;
  memset(&A, 0, sizeof(A));
  A.n_terms = n_terms;
  int64_t rows = 0;
  unsigned blk = 0;
  for (int i = 0; i < n_terms; ++i) {
    const zs_ms_term& s = terms[i];
    MSTerm<T>& t = A.t[i];
    if (s.K < 1 || s.M < 0 || s.D < 1 || (s.M % s.D) != 0) return ZS_EINVAL;
    t.mu = (const T*)s.mu; t.sigma = (const T*)s.sigma; t.eps = (const T*)s.eps;
    t.z = (T*)s.z; t.lp = (T*)s.lp;
    t.K = s.K; t.M = s.M; t.D = s.D; t.R = s.M / s.D; t.sk = s.lp_stride_k; t.sr = s.lp_stride_r;
    t.offset = s.offset;
    t.ls = s.sigma_is_logstd;
    t.row0 = rows;
    t.block0 = blk;
    if (s.M == 0) continue;
    if (!backward) {
      if (!s.mu || !s.sigma || !s.z) return ZS_EINVAL;
      rows += s.K * t.R;
    } else {
      if (!s.sigma || !s.gmu || !s.gsigma) return ZS_EINVAL;
      t.gz = (const T*)s.gz; t.glp = (const T*)s.glp; t.gsk = s.glp_stride_k; t.gsr = s.glp_stride_r;
      t.gmu = (T*)s.gmu; t.gsigma = (T*)s.gsigma;
      t.nblocks = (unsigned)((s.M + 255) / 256);
      blk += t.nblocks;
    }
  }
  if (rows > (int64_t(1) << 31) || blk > (1u << 30)) return ZS_ENOTSUP;
  A.total_rows = rows;
  grid = backward ? blk : (unsigned)((rows + 3) / 4);
  return 0;
}

template <typename T>
int ms_fwd(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rs, uint64_t* rng_used, void* stream) {
  MSArgs<T> A;
  unsigned grid = 0;
  const int rc = ms_build<T>(terms, n_terms, false, A, grid);
  if (rc) return rc;
  if (grid == 0) grid = 1;                       // still publishes rng_used
  ZS_LAUNCH(KID_NORMAL_SAMPLE_MULTI, (k_normal_sample_multi<T>), dim3(grid), dim3(256), (hipStream_t)stream, A, seed, rs, rng_used);
  ZS_CHECK_LAUNCH();
  return 0;
}
template <typename T>
int ms_bwd(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rs, void* stream) {
  MSArgs<T> A;
  unsigned grid = 0;
  const int rc = ms_build<T>(terms, n_terms, true, A, grid);
  if (rc) return rc;
  if (grid == 0) return 0;
  ZS_LAUNCH(KID_NORMAL_SAMPLE_MULTI_BWD, (k_normal_sample_multi_bwd<T>), dim3(grid), dim3(256), (hipStream_t)stream, A, seed, rs);
  ZS_CHECK_LAUNCH();
  return 0;
}

// ================================================================ PL1
constexpr int PL_BT = 64;        // batch rows per tile
constexpr int PL_LDS_FLOATS = 15360;   // 60 KB of fp32 (30 K doubles would not fit: the double twin halves the limits)

__host__ __device__ __forceinline__ int pl_odd(int v) { return v | 1; }     // odd leading dimension: conflict-free columns

// forward: workgroup = (tile of PL_BT rows, particle k); w[k] and the h tile are staged in LDS; outputs of the tile are one
// contiguous run of PL_BT * n_out values: consecutive lanes take consecutive (b, o) pairs -> coalesced stores
template <typename T>
__global__ __launch_bounds__(256) void k_particle_linear(const T* __restrict__ h, int64_t hsk, const T* __restrict__ w,
                                                         T* __restrict__ out, int B, int n_in, int n_out, int relu, int ntiles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* ws = reinterpret_cast<T*>(smem_raw);
  const int WS = pl_odd(n_in + 1), HS = pl_odd(n_in);
  T* hs = ws + n_out * WS;
  const int k = blockIdx.x / ntiles, bt = blockIdx.x - k * ntiles;
  const int b0 = bt * PL_BT, nb = B - b0 < PL_BT ? B - b0 : PL_BT;
  const T* __restrict__ wk = w + (int64_t)k * n_out * (n_in + 1);
  for (int e = threadIdx.x; e < n_out * (n_in + 1); e += 256) {
    const int o = e / (n_in + 1), i = e - o * (n_in + 1);
    ws[o * WS + i] = wk[e];
  }
  const T* __restrict__ hk = h + (int64_t)k * hsk + (int64_t)b0 * n_in;
  for (int e = threadIdx.x; e < nb * n_in; e += 256) {
    const int r = e / n_in, c = e - r * n_in;
    hs[r * HS + c] = hk[e];
  }
  __syncthreads();
  const T p = Mth<T>::rsqrt_n(n_in + 1);                           // torch.sqrt(torch.as_tensor(h.shape[2])), bnn_vi.py:42
  T* __restrict__ ok = out + ((int64_t)k * B + b0) * n_out;
  for (int e = threadIdx.x; e < nb * n_out; e += 256) {
    const int b = e / n_out, o = e - b * n_out;
    const T* __restrict__ hr = hs + b * HS;
    const T* __restrict__ wr = ws + o * WS;
    T acc = (T)0;
    for (int i = 0; i < n_in; ++i) acc += hr[i] * wr[i];
    acc += wr[n_in];                                                // the appended column of ones (bnn_vi.py:40)
    acc = acc / p;
    if (relu) acc = acc > (T)0 ? acc : (T)0;
    ok[e] = acc;
  }
}

// backward, two roles in one launch:
//   blockIdx <  n_gh : (tile, k)  gh tile = gpre tile x w[k] / p          (skipped when gh == NULL)
//   blockIdx >= n_gh : (chunk of 256 / slices weight elements of particle k) gw = sum over ALL rows b of gpre[b, o] * [h | 1][b, i] / p,
//                      the rows split over `slices` thread groups whose partial sums are added in slice order
template <typename T>
__global__ __launch_bounds__(256) void k_particle_linear_bwd(const T* __restrict__ h, int64_t hsk, const T* __restrict__ w,
                                                             const T* __restrict__ out, const T* __restrict__ gout,
                                                             T* __restrict__ gh, T* __restrict__ gw, int B, int n_in, int n_out,
                                                             int relu, int ntiles, unsigned n_gh, int slices, int chunks) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  const T p = Mth<T>::rsqrt_n(n_in + 1);
  const int GS = pl_odd(n_out), HS = pl_odd(n_in);
  if (blockIdx.x < n_gh) {
    const int WS = n_in + 1;                                  // read along i by consecutive lanes
    T* ws = smem;
    T* gs = ws + n_out * WS;
    const int k = blockIdx.x / ntiles, bt = blockIdx.x - k * ntiles;
    const int b0 = bt * PL_BT, nb = B - b0 < PL_BT ? B - b0 : PL_BT;
    const T* __restrict__ wk = w + (int64_t)k * n_out * (n_in + 1);
    for (int e = threadIdx.x; e < n_out * (n_in + 1); e += 256) ws[e] = wk[e];
    const int64_t ob = ((int64_t)k * B + b0) * n_out;
    for (int e = threadIdx.x; e < nb * n_out; e += 256) {
      const int r = e / n_out, c = e - r * n_out;
      T g = gout[ob + e];
      if (relu && !(out[ob + e] > (T)0)) g = (T)0;
      gs[r * GS + c] = g;
    }
    __syncthreads();
    T* __restrict__ ghk = gh + ((int64_t)k * B + b0) * n_in;
    for (int e = threadIdx.x; e < nb * n_in; e += 256) {
      const int b = e / n_in, i = e - b * n_in;
      const T* __restrict__ gr = gs + b * GS;
      T acc = (T)0;
      for (int o = 0; o < n_out; ++o) acc += gr[o] * ws[o * WS + i];
      ghk[e] = acc / p;
    }
    return;
  }
  // ---- gw role
  const int per = 256 / slices;                               // weight elements per workgroup
  const int blk = blockIdx.x - n_gh;
  const int k = blk / chunks, ch = blk - k * chunks;
  const int nW = n_out * (n_in + 1);
  const int sl = threadIdx.x / per, el = ch * per + (threadIdx.x - sl * per);
  const bool live = el < nW;
  const int o = live ? el / (n_in + 1) : 0, i = live ? el - o * (n_in + 1) : 0;
  T* gs = smem;                       // [PL_BT][GS]
  T* hs = gs + PL_BT * GS;            // [PL_BT][HS]
  T* red = hs + PL_BT * HS;           // [256] slice partials
  const T* __restrict__ hk = h + (int64_t)k * hsk;
  T acc = (T)0;
  for (int b0 = 0; b0 < B; b0 += PL_BT) {
    const int nb = B - b0 < PL_BT ? B - b0 : PL_BT;
    const int64_t ob = ((int64_t)k * B + b0) * n_out;
    __syncthreads();
    for (int e = threadIdx.x; e < nb * n_out; e += 256) {
      const int r = e / n_out, c = e - r * n_out;
      T g = gout[ob + e];
      if (relu && !(out[ob + e] > (T)0)) g = (T)0;
      gs[r * GS + c] = g;
    }
    for (int e = threadIdx.x; e < nb * n_in; e += 256) {
      const int r = e / n_in, c = e - r * n_in;
      hs[r * HS + c] = hk[(int64_t)b0 * n_in + e];
    }
    __syncthreads();
    if (live) {
      if (i < n_in) {
        for (int b = sl; b < nb; b += slices) acc += gs[b * GS + o] * hs[b * HS + i];
      } else {
        for (int b = sl; b < nb; b += slices) acc += gs[b * GS + o];
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && live) {
    T s = (T)0;
    for (int j = 0; j < slices; ++j) s += red[j * per + threadIdx.x];      // slice order: deterministic
    gw[(int64_t)k * nW + el] = s / p;
  }
}

template <typename T>
bool pl_fits(int64_t n_in, int64_t n_out) {
  if (n_in < 1 || n_out < 1 || n_in > 255 || n_out > 256) return false;
  const int64_t lim = PL_LDS_FLOATS * (int64_t)sizeof(float) / (int64_t)sizeof(T);
  const int64_t fwd = n_out * pl_odd((int)n_in + 1) + PL_BT * pl_odd((int)n_in);
  const int64_t bwd_a = n_out * (n_in + 1) + PL_BT * pl_odd((int)n_out);
  const int64_t bwd_b = PL_BT * pl_odd((int)n_out) + PL_BT * pl_odd((int)n_in) + 256;
  return fwd <= lim && bwd_a <= lim && bwd_b <= lim;
}

template <typename T>
int particle_linear(const T* h, int64_t hsk, const T* w, T* out, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu,
                    void* stream) {
  if (K < 0 || B < 0 || n_in < 1 || n_out < 1 || (hsk != 0 && hsk != B * n_in)) return ZS_EINVAL;
  if (!pl_fits<T>(n_in, n_out) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0 || B == 0) return 0;
  if (!h || !w || !out) return ZS_EINVAL;
  const int ntiles = (int)((B + PL_BT - 1) / PL_BT);
  if ((int64_t)ntiles * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  const size_t smem = sizeof(T) * (size_t)(n_out * pl_odd((int)n_in + 1) + PL_BT * pl_odd((int)n_in));
  ZS_LAUNCH_SMEM(KID_PARTICLE_LINEAR, (k_particle_linear<T>), dim3((unsigned)(ntiles * K)), dim3(256), smem, (hipStream_t)stream, h,
                 hsk, w, out, (int)B, (int)n_in, (int)n_out, relu, ntiles);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int particle_linear_bwd(const T* h, int64_t hsk, const T* w, const T* out, const T* gout, T* gh, T* gw, int64_t K, int64_t B,
                        int64_t n_in, int64_t n_out, int relu, void* stream) {
  if (K < 0 || B < 0 || n_in < 1 || n_out < 1 || (hsk != 0 && hsk != B * n_in)) return ZS_EINVAL;
  if (!pl_fits<T>(n_in, n_out) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0) return 0;
  if (!gw) return ZS_EINVAL;
  if (B > 0 && (!h || !w || !gout || (relu && !out))) return ZS_EINVAL;
  const int ntiles = (int)((B + PL_BT - 1) / PL_BT);
  const unsigned n_gh = gh ? (unsigned)(ntiles * K) : 0u;
  const int nW = (int)(n_out * (n_in + 1));
  // few weight elements (the last layer: n_out = 1): split the batch loop over up to 8 thread groups
  int slices = 1;
  while (slices < 8 && nW * slices * 2 <= 256) slices *= 2;
  const int per = 256 / slices, chunks = (nW + per - 1) / per;
  if ((int64_t)n_gh + (int64_t)chunks * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  const size_t sm_a = sizeof(T) * (size_t)(n_out * (n_in + 1) + PL_BT * pl_odd((int)n_out));
  const size_t sm_b = sizeof(T) * (size_t)(PL_BT * pl_odd((int)n_out) + PL_BT * pl_odd((int)n_in) + 256);
  const size_t smem = sm_a > sm_b ? sm_a : sm_b;
  ZS_LAUNCH_SMEM(KID_PARTICLE_LINEAR_BWD, (k_particle_linear_bwd<T>), dim3(n_gh + (unsigned)(chunks * K)), dim3(256), smem,
                 (hipStream_t)stream, h, hsk, w, out, gout, gh, gw, (int)B, (int)n_in, (int)n_out, relu, ntiles, n_gh, slices, chunks);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int zs_logjoint_scalar_f32(const zs_lj_term* terms, int n_terms, float* out, double* workspace, int64_t workspace_len,
                                      uint32_t* ticket, void* stream) {
  return logjoint_fwd<float>(terms, n_terms, out, workspace, workspace_len, ticket, stream);
}
extern "C" int zs_logjoint_scalar_f64(const zs_lj_term* terms, int n_terms, double* out, double* workspace, int64_t workspace_len,
                                      uint32_t* ticket, void* stream) {
  return logjoint_fwd<double>(terms, n_terms, out, workspace, workspace_len, ticket, stream);
}
extern "C" int zs_logjoint_scalar_bwd_f32(const zs_lj_term* terms, int n_terms, const float* gout, float* gcoef, double* workspace,
                                          int64_t workspace_len, uint32_t* ticket, void* stream) {
  return logjoint_bwd<float>(terms, n_terms, gout, gcoef, workspace, workspace_len, ticket, stream);
}
extern "C" int zs_logjoint_scalar_bwd_f64(const zs_lj_term* terms, int n_terms, const double* gout, double* gcoef, double* workspace,
                                          int64_t workspace_len, uint32_t* ticket, void* stream) {
  return logjoint_bwd<double>(terms, n_terms, gout, gcoef, workspace, workspace_len, ticket, stream);
}
extern "C" int zs_normal_sample_logprob_multi_f32(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                                  uint64_t* rng_used, void* stream) {
  return ms_fwd<float>(terms, n_terms, seed, rng_state, rng_used, stream);
}
extern "C" int zs_normal_sample_logprob_multi_f64(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                                  uint64_t* rng_used, void* stream) {
  return ms_fwd<double>(terms, n_terms, seed, rng_state, rng_used, stream);
}
extern "C" int zs_normal_sample_logprob_multi_bwd_f32(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                                      void* stream) {
  return ms_bwd<float>(terms, n_terms, seed, rng_state, stream);
}
extern "C" int zs_normal_sample_logprob_multi_bwd_f64(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                                      void* stream) {
  return ms_bwd<double>(terms, n_terms, seed, rng_state, stream);
}
extern "C" int zs_particle_linear_f32(const float* h, int64_t h_stride_k, const float* w, float* out, int64_t K, int64_t B,
                                      int64_t n_in, int64_t n_out, int relu, void* stream) {
  return particle_linear<float>(h, h_stride_k, w, out, K, B, n_in, n_out, relu, stream);
}
extern "C" int zs_particle_linear_f64(const double* h, int64_t h_stride_k, const double* w, double* out, int64_t K, int64_t B,
                                      int64_t n_in, int64_t n_out, int relu, void* stream) {
  return particle_linear<double>(h, h_stride_k, w, out, K, B, n_in, n_out, relu, stream);
}
extern "C" int zs_particle_linear_bwd_f32(const float* h, int64_t h_stride_k, const float* w, const float* out, const float* gout,
                                          float* gh, float* gw, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu,
                                          void* stream) {
  return particle_linear_bwd<float>(h, h_stride_k, w, out, gout, gh, gw, K, B, n_in, n_out, relu, stream);
}
extern "C" int zs_particle_linear_bwd_f64(const double* h, int64_t h_stride_k, const double* w, const double* out,
                                          const double* gout, double* gh, double* gw, int64_t K, int64_t B, int64_t n_in,
                                          int64_t n_out, int relu, void* stream) {
  return particle_linear_bwd<double>(h, h_stride_k, w, out, gout, gh, gw, K, B, n_in, n_out, relu, stream);
}
