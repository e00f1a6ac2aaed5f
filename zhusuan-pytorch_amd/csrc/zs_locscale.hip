// Logistic and Uniform kernels for gfx950 (the reference's two other hand-written
// samplers, zhusuan/distributions/logistic.py and uniform.py).  Contract: include/zs_hip.h.
//
// HBM-bound streaming work, one code path for float and double (template parameter T):
//   * a thread owns 4 consecutive flat elements = one Philox4x32 group = one 16-byte (fp32) access when the
//     operands allow it (VEC), scalar accesses otherwise;
//   * row sums ("span" kernel): a 256-thread workgroup owns a span of whole rows (<= 2048 elements), parks the
//     per-element terms in LDS and then adds each row with G lanes -- the global accesses stay fully coalesced
//     for ANY row length D (40, 51, 784 ...), unlike a lane-group-per-row mapping;
//   * rows longer than the span: one workgroup per row, register accumulation + block reduction;
//   * D == 1 (no fold) with a contiguous result: terms go straight from registers to the result.
#include "zs_common.h"
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

constexpr int kSpan = 2048;

template <typename T>
struct alignas(sizeof(T) * 4) V4 {
  T v[4];
};

// ---------------------------------------------------------------- math in T
__device__ __forceinline__ float t_log(float x) { return log2_fast(x) * ZS_LN2; }
__device__ __forceinline__ double t_log(double x) { return log(x); }
__device__ __forceinline__ float t_exp(float x) { return exp_fast(x); }
__device__ __forceinline__ double t_exp(double x) { return exp(x); }
__device__ __forceinline__ float t_log1p(float e) { return log2_fast(1.0f + e) * ZS_LN2; }  // e in (0, 1]
__device__ __forceinline__ double t_log1p(double e) { return log1p(e); }
__device__ __forceinline__ float t_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return fabs(x); }
__device__ __forceinline__ float t_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return fmax(a, b); }

template <typename T>
__device__ __forceinline__ T mul_add_2round(T a, T b, T e) {
#pragma clang fp contract(off)   // loc + scale * eps rounds twice like the reference (logistic.py:66, uniform.py:70)
  const T prod = b * e;
  return a + prod;
}

// Logistic log-density of x (logistic.py:81-82): -t - 2*softplus(-t) - log(scale), t = (x - loc)/scale,
// softplus(-t) = max(-t, 0) + log1p(exp(-|t|)).
template <typename T>
__device__ __forceinline__ T logistic_term(T x, T loc, T scale) {
  const T t = (x - loc) / scale;
  const T e = t_exp(-t_abs(t));
  const T sp = t_max(-t, (T)0) + t_log1p(e);
  return (-t - (T)2 * sp) - t_log(scale);
}

__device__ __forceinline__ uint32_t philox_word(const Philox4& r, int j) {
  return j == 0 ? r.x : (j == 1 ? r.y : (j == 2 ? r.z : r.w));
}

// ---------------------------------------------------------------- operand access
template <typename T, bool VEC>
__device__ __forceinline__ void ld4(const T* __restrict__ p, int64_t i0, int n, T out[4]) {
  if (VEC) {
    const V4<T> v = *reinterpret_cast<const V4<T>*>(p + i0);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v.v[j];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = j < n ? p[i0 + j] : (T)0;
  }
}
template <typename T, bool VEC>
__device__ __forceinline__ void st4(T* __restrict__ p, int64_t i0, int n, const T in[4]) {
  if (VEC) {
    V4<T> v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v.v[j] = in[j];
    *reinterpret_cast<V4<T>*>(p + i0) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < n) p[i0 + j] = in[j];
  }
}
// periodic operand a[i % P]; VEC callers guarantee P % 4 == 0 (or P == 1) and i0 % 4 == 0
template <typename T, bool VEC>
__device__ __forceinline__ void ldp4(const T* __restrict__ p, int64_t P, int64_t i0, int n, T out[4], T pad) {
  if (P == 1) {
    const T v = p[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v;
    return;
  }
  if (VEC) {
    const int64_t idx = i0 < P ? i0 : mod_fast(i0, P);
    ld4<T, true>(p, idx, 4, out);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = j < n ? p[(i0 + j) < P ? (i0 + j) : mod_fast(i0 + j, P)] : pad;
  }
}
// uniform (0,1) draws for flat elements i0 .. i0+3: supplied, or words of the Philox group(s)
template <typename T, bool VEC>
__device__ __forceinline__ void draw4(const T* __restrict__ u, int64_t i0, int n, uint64_t seed, uint64_t call, T out[4]) {
  if (u) {
    ld4<T, VEC>(u, i0, n, out);
    if (!VEC) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j >= n) out[j] = (T)0.5;
    }
    return;
  }
  if (VEC || (i0 & 3) == 0) {
    const Philox4 r = philox4x32_10((uint64_t)(i0 >> 2), call, seed);
    out[0] = (T)u01(r.x); out[1] = (T)u01(r.y); out[2] = (T)u01(r.z); out[3] = (T)u01(r.w);
  } else {
    const Philox4 a = philox4x32_10((uint64_t)(i0 >> 2), call, seed);
    const Philox4 b = philox4x32_10((uint64_t)(i0 >> 2) + 1, call, seed);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int w = (int)(i0 & 3) + j;
      out[j] = (T)u01(w < 4 ? philox_word(a, w) : philox_word(b, w - 4));
    }
  }
}

// ---------------------------------------------------------------- functors: per-4-element work
template <typename T>
struct LogisticSampleF {   // L1
  const T* loc; const T* scale; const T* u; uint64_t seed, call; const uint64_t* rs; T* z; int64_t M;
  __device__ void prepare() { if (rs) { seed = rs[0]; call += rs[1]; } }
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool want) const {
    T a[4], b[4], uu[4], zz[4];
    ldp4<T, VEC>(loc, M, i0, n, a, (T)0);
    ldp4<T, VEC>(scale, M, i0, n, b, (T)1);
    draw4<T, VEC>(u, i0, n, seed, call, uu);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T lu = t_log(uu[j]), l1 = t_log((T)1 - uu[j]);
      zz[j] = mul_add_2round(a[j], b[j], lu - l1);
      // log-density of the fresh sample: -eps - 2*softplus(-eps) = log(u) + log(1-u)
      if (want) t[j] = (lu + l1) - t_log(b[j]);
    }
    st4<T, VEC>(z, i0, n, zz);
  }
};

template <typename T>
struct LogisticLogProbF {   // L2
  const T* x; int64_t Px; const T* loc; int64_t Pm; const T* scale; int64_t Ps;
  __device__ void prepare() {}
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool) const {
    T xv[4], a[4], b[4];
    ldp4<T, VEC>(x, Px, i0, n, xv, (T)0);
    ldp4<T, VEC>(loc, Pm, i0, n, a, (T)0);
    ldp4<T, VEC>(scale, Ps, i0, n, b, (T)1);
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = logistic_term(xv[j], a[j], b[j]);
  }
};

template <typename T>
struct UniformLogProbF {   // U2
  const T* x; int64_t Px; const T* low; int64_t Pl; const T* high; int64_t Ph;
  __device__ void prepare() {}
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool) const {
    T xv[4], lo[4], hi[4];
    ldp4<T, VEC>(x, Px, i0, n, xv, (T)0);
    ldp4<T, VEC>(low, Pl, i0, n, lo, (T)0);
    ldp4<T, VEC>(high, Ph, i0, n, hi, (T)1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool inside = (lo[j] <= xv[j]) && (hi[j] > xv[j]);   // torch Uniform.log_prob: lb * ub
      t[j] = (inside ? (T)0 : (T)(-INFINITY)) - t_log(hi[j] - lo[j]);
    }
  }
};

template <typename T>
struct UniformSampleF {   // U1
  const T* low; int64_t Pl; const T* high; int64_t Ph; const T* u; uint64_t seed, call; const uint64_t* rs;
  T* out; T* cache; int reparam;
  __device__ void prepare() { if (rs) { seed = rs[0]; call += rs[1]; } }
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    T lo[4], hi[4], uu[4], o[4], c[4];
    ldp4<T, VEC>(low, Pl, i0, n, lo, (T)0);
    ldp4<T, VEC>(high, Ph, i0, n, hi, (T)1);
    draw4<T, VEC>(u, i0, n, seed, call, uu);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T w = hi[j] - lo[j];
      c[j] = reparam ? uu[j] : mul_add_2round(lo[j], uu[j], w);   // uniform.py:63-67
      o[j] = mul_add_2round(lo[j], c[j], w);                      // uniform.py:70
    }
    st4<T, VEC>(out, i0, n, o);
    if (cache) st4<T, VEC>(cache, i0, n, c);
  }
};

template <typename T>
struct PhiloxUniformF {
  uint64_t seed, call; const uint64_t* rs; T* out;
  __device__ void prepare() { if (rs) { seed = rs[0]; call += rs[1]; } }
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    T uu[4];
    draw4<T, VEC>(nullptr, i0, n, seed, call, uu);
    st4<T, VEC>(out, i0, n, uu);
  }
};

template <typename T>
struct LogisticLogProbBwdF {   // element-wise partials of L2
  const T* x; int64_t Px; const T* loc; int64_t Pm; const T* scale; int64_t Ps;
  const T* glp; int64_t gsk, gsr; T* gx; T* gloc; T* gscale; int64_t R, D;
  __device__ void prepare() {}
  template <bool VEC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    T xv[4], a[4], b[4], o1[4], o2[4], o3[4];
    ldp4<T, VEC>(x, Px, i0, n, xv, (T)0);
    ldp4<T, VEC>(loc, Pm, i0, n, a, (T)0);
    ldp4<T, VEC>(scale, Ps, i0, n, b, (T)1);
    int64_t row, dd;
    divmod(i0, D, row, dd);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < n) {
        int64_t k, r;
        divmod(row, R, k, r);
        const T g = glp[k * gsk + r * gsr];
        const T t = (xv[j] - a[j]) / b[j];
        const T e = t_exp(-t_abs(t));
        T h = ((T)1 - e) / ((T)1 + e);            // tanh(|t|/2)
        h = t < (T)0 ? -h : h;
        const T gh = g * h / b[j];
        o1[j] = -gh;
        o2[j] = gh;
        o3[j] = g * (h * t - (T)1) / b[j];
        if (++dd == D) { dd = 0; ++row; }
      } else {
        o1[j] = o2[j] = o3[j] = (T)0;
      }
    }
    if (gx) st4<T, VEC>(gx, i0, n, o1);
    if (gloc) st4<T, VEC>(gloc, i0, n, o2);
    if (gscale) st4<T, VEC>(gscale, i0, n, o3);
  }
};

// ---------------------------------------------------------------- kernels
__device__ __forceinline__ int pad_idx(int e) { return e + (e >> 5); }

// Row sums over spans of whole rows (D <= span).  rpb rows per workgroup pass, G lanes add one row.
template <typename T, typename F, bool VEC>
__global__ __launch_bounds__(256) void k_span_rows(F f, T* __restrict__ lp, int64_t rows, int64_t R, int64_t D, int rpb,
                                                   int lgG, int64_t sk, int64_t sr, int direct) {
  __shared__ T term[kSpan + kSpan / 32 + 1];
  f.prepare();
  const bool want = lp != nullptr;
  const int G = 1 << lgG;
  const int64_t tiles = (rows + rpb - 1) / rpb;
  for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int64_t row0 = tile * rpb;
    const int nrows = (int)((rows - row0 < rpb) ? (rows - row0) : rpb);
    const int64_t e0 = row0 * D;
    const int n = nrows * (int)D;
    for (int e = threadIdx.x * 4; e < n; e += 1024) {
      T t[4] = {(T)0, (T)0, (T)0, (T)0};
      const int cnt = n - e < 4 ? n - e : 4;
      f.template eval<VEC>(e0 + e, cnt, t, want);
      if (!want) continue;
      if (direct) {
        st4<T, VEC>(lp, e0 + e, cnt, t);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < cnt) term[pad_idx(e + j)] = t[j];
      }
    }
    if (want && !direct) {
      __syncthreads();
      const int q0 = threadIdx.x >> lgG, g = threadIdx.x & (G - 1);
      for (int q = q0; q < nrows; q += 256 >> lgG) {
        T acc = (T)0;
        const int base = q * (int)D;
        for (int d = g; d < (int)D; d += G) acc += term[pad_idx(base + d)];
        for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, ZS_WAVE);
        if (g == 0) {
          int64_t k, r;
          divmod(row0 + q, R, k, r);
          lp[k * sk + r * sr] = acc;
        }
      }
      __syncthreads();
    }
  }
}

// Rows longer than the span: one workgroup per row.
template <typename T, typename F, bool VEC>
__global__ __launch_bounds__(256) void k_long_rows(F f, T* __restrict__ lp, int64_t rows, int64_t R, int64_t D, int64_t sk,
                                                   int64_t sr) {
  __shared__ T part[4];
  f.prepare();
  const bool want = lp != nullptr;
  for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
    const int64_t e0 = row * D;
    T acc = (T)0;
    for (int64_t e = (int64_t)threadIdx.x * 4; e < D; e += 1024) {
      T t[4] = {(T)0, (T)0, (T)0, (T)0};
      const int cnt = D - e < 4 ? (int)(D - e) : 4;
      f.template eval<VEC>(e0 + e, cnt, t, want);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < cnt) acc += t[j];
    }
    if (want) {
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, ZS_WAVE);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
      __syncthreads();
      if (threadIdx.x == 0) {
        int64_t k, r;
        divmod(row, R, k, r);
        lp[k * sk + r * sr] = (part[0] + part[1]) + (part[2] + part[3]);
      }
    }
  }
}

template <typename T, typename F, bool VEC>
__global__ __launch_bounds__(256) void k_elem(F f, int64_t N) {
  f.prepare();
  const int64_t groups = (N + 3) >> 2;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = g << 2;
    f.template eval<VEC>(i0, N - i0 < 4 ? (int)(N - i0) : 4);
  }
}

// L1 backward: workgroup = 64 groups of 4 parameters x 4 K-slices, slices combined through LDS.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_logistic_sample_bwd(const T* __restrict__ scale, const T* __restrict__ u, uint64_t seed,
                                                             uint64_t call, const uint64_t* __restrict__ rs,
                                                             const T* __restrict__ gz, const T* __restrict__ glp, int64_t gsk,
                                                             int64_t gsr, T* __restrict__ gloc, T* __restrict__ gscale, int64_t K,
                                                             int64_t M, int64_t D) {
  __shared__ T red[3][4][4][64];
  if (rs) { seed = rs[0]; call += rs[1]; }
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t m0 = ((int64_t)blockIdx.x * 64 + lane) * 4;
  const int n = M - m0 < 4 ? (int)(M - m0) : 4;   // <= 0: lane off
  T a[4] = {(T)0, (T)0, (T)0, (T)0}, b[4] = {(T)0, (T)0, (T)0, (T)0}, gl[4] = {(T)0, (T)0, (T)0, (T)0};
  if (n > 0) {
    int64_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (m0 + j) / D;
    for (int64_t k = slice; k < K; k += 4) {
      const int64_t i0 = k * M + m0;
      if (gz) {
        T gv[4], uu[4];
        ld4<T, VEC>(gz, i0, n, gv);
        draw4<T, VEC>(u, i0, n, seed, call, uu);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const T eps = t_log(uu[j]) - t_log((T)1 - uu[j]);
          a[j] += gv[j];
          b[j] += gv[j] * eps;
        }
      }
      if (glp) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < n) gl[j] += glp[k * gsk + r[j] * gsr];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][slice][j][lane] = a[j];
    red[1][slice][j][lane] = b[j];
    red[2][slice][j][lane] = gl[j];
  }
  __syncthreads();
  if (slice == 0 && n > 0) {
    T sc[4], o1[4], o2[4];
    ld4<T, VEC>(scale, m0, n, sc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      T sa = (T)0, sb = (T)0, sg = (T)0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        sa += red[0][s][j][lane];
        sb += red[1][s][j][lane];
        sg += red[2][s][j][lane];
      }
      o1[j] = sa;
      o2[j] = j < n ? sb - sg / sc[j] : (T)0;
    }
    st4<T, VEC>(gloc, m0, n, o1);
    st4<T, VEC>(gscale, m0, n, o2);
  }
}

// ---------------------------------------------------------------- host side
inline bool al(const void* p, size_t bytes) { return p == nullptr || (((uintptr_t)p) & (bytes - 1)) == 0; }
inline bool per4(int64_t P) { return P == 1 || (P & 3) == 0; }

template <typename T, typename F>
int launch_rows(int kid, F f, bool vec_ok, T* lp, int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, hipStream_t st) {
  const int64_t rows = K * R, N = rows * D;
  vec_ok = vec_ok && (N & 3) == 0 && al(lp, sizeof(T));
  if (D > kSpan) {
    const bool vec = vec_ok && (D & 3) == 0;
    const dim3 grid(grid_for(rows, 1, 256u * 32u));
    if (vec) ZS_LAUNCH(kid, (k_long_rows<T, F, true>), grid, dim3(256), st, f, lp, rows, R, D, sk, sr);
    else ZS_LAUNCH(kid, (k_long_rows<T, F, false>), grid, dim3(256), st, f, lp, rows, R, D, sk, sr);
    return 0;
  }
  int span = kSpan;                        // smaller spans for small problems: more workgroups than CUs
  while (span > 256 && N / span < 1024) span >>= 1;
  if (span < D) span = (int)D;
  int rpb = span / (int)D;
  if (rpb >= 4) rpb &= ~3;                 // spans start on a multiple of 4 elements
  if (rpb > rows) rpb = (int)rows;
  const bool direct = lp != nullptr && D == 1 && sr == 1 && (K == 1 || sk == R);
  const bool vec = vec_ok && ((((int64_t)rpb * D) & 3) == 0 || rpb >= rows) && (!direct || al(lp, sizeof(T) * 4));
  int lgG = 0;
  while (lgG < 6 && (2 << lgG) * rpb <= 256 && (1 << lgG) < D) ++lgG;
  const dim3 grid(grid_for((rows + rpb - 1) / rpb, 1, 256u * 32u));
  if (vec) ZS_LAUNCH(kid, (k_span_rows<T, F, true>), grid, dim3(256), st, f, lp, rows, R, D, rpb, lgG, sk, sr, (int)direct);
  else ZS_LAUNCH(kid, (k_span_rows<T, F, false>), grid, dim3(256), st, f, lp, rows, R, D, rpb, lgG, sk, sr, (int)direct);
  return 0;
}

template <typename T, typename F>
int launch_elem(int kid, F f, bool vec_ok, int64_t N, hipStream_t st) {
  const dim3 grid(grid_for((N + 3) / 4, 256));
  if (vec_ok && (N & 3) == 0) ZS_LAUNCH(kid, (k_elem<T, F, true>), grid, dim3(256), st, f, N);
  else ZS_LAUNCH(kid, (k_elem<T, F, false>), grid, dim3(256), st, f, N);
  return 0;
}

template <typename T>
int logistic_sample(const T* loc, const T* scale, const T* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, T* z,
                    T* lp, int64_t K, int64_t M, int64_t D, int64_t sk, int64_t sr, void* stream) {
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!loc || !scale || !z) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = (M & 3) == 0 && al(loc, A) && al(scale, A) && al(u, A) && al(z, A);
  LogisticSampleF<T> f = {loc, scale, u, seed, offset, rng_state, z, M};
  launch_rows<T>(KID_LOGISTIC_SAMPLE, f, vec, lp, K, M / D, D, sk, sr, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int logistic_sample_bwd(const T* scale, const T* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, const T* gz,
                        const T* glp, int64_t gsk, int64_t gsr, T* gloc, T* gscale, int64_t K, int64_t M, int64_t D,
                        void* stream) {
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!scale || !gloc || !gscale) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = (M & 3) == 0 && al(scale, A) && al(u, A) && al(gz, A) && al(gloc, A) && al(gscale, A);
  const dim3 grid((unsigned)((M + 255) / 256));
  if (vec)
    ZS_LAUNCH(KID_LOGISTIC_SAMPLE_BWD, (k_logistic_sample_bwd<T, true>), grid, dim3(256), (hipStream_t)stream, scale, u, seed,
              offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D);
  else
    ZS_LAUNCH(KID_LOGISTIC_SAMPLE_BWD, (k_logistic_sample_bwd<T, false>), grid, dim3(256), (hipStream_t)stream, scale, u, seed,
              offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D);
  ZS_CHECK_LAUNCH();
  return 0;
}

// shared argument checks of the [K, R, D] problems with three periodic operands
inline int check3(int64_t K, int64_t R, int64_t D, int64_t P1, int64_t P2, int64_t P3, int64_t* N) {
  if (K < 1 || R < 0 || D < 1 || P1 < 1 || P2 < 1 || P3 < 1) return ZS_EINVAL;
  *N = K * R * D;
  return 0;
}

template <typename T>
int logistic_logprob(const T* x, int64_t Px, const T* loc, int64_t Pm, const T* scale, int64_t Ps, T* lp, int64_t K, int64_t R,
                     int64_t D, int64_t sk, int64_t sr, void* stream) {
  int64_t N;
  if (check3(K, R, D, Px, Pm, Ps, &N)) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!x || !loc || !scale || !lp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = per4(Px) && per4(Pm) && per4(Ps) && (Px == 1 || al(x, A)) && (Pm == 1 || al(loc, A)) && (Ps == 1 || al(scale, A));
  LogisticLogProbF<T> f = {x, Px, loc, Pm, scale, Ps};
  launch_rows<T>(KID_LOGISTIC_LOGPROB, f, vec, lp, K, R, D, sk, sr, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int logistic_logprob_bwd(const T* x, int64_t Px, const T* loc, int64_t Pm, const T* scale, int64_t Ps, const T* glp, int64_t gsk,
                         int64_t gsr, T* gx, T* gloc, T* gscale, int64_t K, int64_t R, int64_t D, void* stream) {
  int64_t N;
  if (check3(K, R, D, Px, Pm, Ps, &N)) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!x || !loc || !scale || !glp) return ZS_EINVAL;
  if (N % Px || N % Pm || N % Ps) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = per4(Px) && per4(Pm) && per4(Ps) && (Px == 1 || al(x, A)) && (Pm == 1 || al(loc, A)) &&
                   (Ps == 1 || al(scale, A)) && al(gx, A) && al(gloc, A) && al(gscale, A);
  LogisticLogProbBwdF<T> f = {x, Px, loc, Pm, scale, Ps, glp, gsk, gsr, gx, gloc, gscale, R, D};
  launch_elem<T>(KID_LOGISTIC_LOGPROB_BWD, f, vec, N, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int uniform_sample(const T* low, int64_t Pl, const T* high, int64_t Ph, const T* u, uint64_t seed, uint64_t offset,
                   const uint64_t* rng_state, T* out, T* cache, int64_t N, int reparam, void* stream) {
  if (N < 0 || Pl < 1 || Ph < 1) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!low || !high || !out) return ZS_EINVAL;
  if (N % Pl || N % Ph) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = per4(Pl) && per4(Ph) && (Pl == 1 || al(low, A)) && (Ph == 1 || al(high, A)) && al(u, A) && al(out, A) &&
                   al(cache, A);
  UniformSampleF<T> f = {low, Pl, high, Ph, u, seed, offset, rng_state, out, cache, reparam};
  launch_elem<T>(KID_UNIFORM_SAMPLE, f, vec, N, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int uniform_logprob(const T* x, int64_t Px, const T* low, int64_t Pl, const T* high, int64_t Ph, T* lp, int64_t K, int64_t R,
                    int64_t D, int64_t sk, int64_t sr, void* stream) {
  int64_t N;
  if (check3(K, R, D, Px, Pl, Ph, &N)) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!x || !low || !high || !lp) return ZS_EINVAL;
  if (N % Px || N % Pl || N % Ph) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = per4(Px) && per4(Pl) && per4(Ph) && (Px == 1 || al(x, A)) && (Pl == 1 || al(low, A)) && (Ph == 1 || al(high, A));
  UniformLogProbF<T> f = {x, Px, low, Pl, high, Ph};
  launch_rows<T>(KID_UNIFORM_LOGPROB, f, vec, lp, K, R, D, sk, sr, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int philox_uniform(T* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state, void* stream) {
  if (N < 0) return ZS_EINVAL;
  if (N == 0) return 0;
  if (!out) return ZS_EINVAL;
  PhiloxUniformF<T> f = {seed, offset, rng_state, out};
  launch_elem<T>(KID_PHILOX_UNIFORM, f, al(out, sizeof(T) * 4), N, (hipStream_t)stream);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

#define ZS_LOCSCALE_ENTRY(SFX, T)                                                                                                  \
  extern "C" int zs_logistic_sample_logprob##SFX(const T* loc, const T* scale, const T* u, uint64_t seed, uint64_t offset,         \
                                                 const uint64_t* rng_state, T* z, T* lp, int64_t K, int64_t M, int64_t D,          \
                                                 int64_t sk, int64_t sr, void* stream) {                                           \
    return logistic_sample<T>(loc, scale, u, seed, offset, rng_state, z, lp, K, M, D, sk, sr, stream);                             \
  }                                                                                                                                \
  extern "C" int zs_logistic_sample_logprob_bwd##SFX(const T* scale, const T* u, uint64_t seed, uint64_t offset,                   \
                                                     const uint64_t* rng_state, const T* gz, const T* glp, int64_t gsk,            \
                                                     int64_t gsr, T* gloc, T* gscale, int64_t K, int64_t M, int64_t D,             \
                                                     void* stream) {                                                               \
    return logistic_sample_bwd<T>(scale, u, seed, offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D, stream);            \
  }                                                                                                                                \
  extern "C" int zs_logistic_logprob##SFX(const T* x, int64_t Px, const T* loc, int64_t Pm, const T* scale, int64_t Ps, T* lp,     \
                                          int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, void* stream) {                 \
    return logistic_logprob<T>(x, Px, loc, Pm, scale, Ps, lp, K, R, D, sk, sr, stream);                                            \
  }                                                                                                                                \
  extern "C" int zs_logistic_logprob_bwd##SFX(const T* x, int64_t Px, const T* loc, int64_t Pm, const T* scale, int64_t Ps,        \
                                              const T* glp, int64_t gsk, int64_t gsr, T* gx, T* gloc, T* gscale, int64_t K,        \
                                              int64_t R, int64_t D, void* stream) {                                                \
    return logistic_logprob_bwd<T>(x, Px, loc, Pm, scale, Ps, glp, gsk, gsr, gx, gloc, gscale, K, R, D, stream);                   \
  }                                                                                                                                \
  extern "C" int zs_uniform_sample##SFX(const T* low, int64_t Pl, const T* high, int64_t Ph, const T* u, uint64_t seed,            \
                                        uint64_t offset, const uint64_t* rng_state, T* out, T* cache, int64_t N, int reparam,      \
                                        void* stream) {                                                                            \
    return uniform_sample<T>(low, Pl, high, Ph, u, seed, offset, rng_state, out, cache, N, reparam, stream);                       \
  }                                                                                                                                \
  extern "C" int zs_uniform_logprob##SFX(const T* x, int64_t Px, const T* low, int64_t Pl, const T* high, int64_t Ph, T* lp,       \
                                         int64_t K, int64_t R, int64_t D, int64_t sk, int64_t sr, void* stream) {                  \
    return uniform_logprob<T>(x, Px, low, Pl, high, Ph, lp, K, R, D, sk, sr, stream);                                              \
  }                                                                                                                                \
  extern "C" int zs_philox_uniform##SFX(T* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state,              \
                                        void* stream) {                                                                            \
    return philox_uniform<T>(out, N, seed, offset, rng_state, stream);                                                             \
  }

ZS_LOCSCALE_ENTRY(_f32, float)
ZS_LOCSCALE_ENTRY(_f64, double)
