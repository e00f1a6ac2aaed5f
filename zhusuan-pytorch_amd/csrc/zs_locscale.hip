// Logistic and Uniform kernels for gfx950 (the reference's two other hand-written
// samplers, zhusuan/distributions/logistic.py and uniform.py).  Contract: include/zs_hip.h.
//
// HBM-bound streaming work, one code path for float and double (template parameter T):
//   * a thread owns 4 consecutive flat elements = one Philox4x32 group = one 16-byte (fp32) access when the
//     operands allow it (VEC), scalar accesses otherwise;
//   * row sums ("wave tile" kernel): a wavefront owns a tile of whole rows (<= 1024 elements: rr consecutive rows
//     r for kk consecutive particles k), parks the per-element terms in its private LDS slice, adds each row with
//     G lanes and writes the results in OUTPUT order -- global reads stay 16 B per lane for ANY row length D
//     (40, 51, 784 ...), K-fastest results leave as runs of kk consecutive floats, and there is no workgroup
//     barrier (same-wave LDS traffic is ordered);
//   * rows longer than a tile: one workgroup per row, register accumulation + block reduction;
//   * D == 1 (no fold) with a contiguous result: terms go straight from registers to the result.
#include "zs_common.h"
#include "zs_sample_tile.h"
#include <type_traits>
#include "../../include/zs_hip.h"

using namespace zs;

namespace {

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
    // unconditional (clamped) loads: a guard per element would serialise them behind branches
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = p[i0 + (j < n ? j : 0)];
  }
}
// (Non-temporal stores were measured and rejected for these kernels: the strided 160-byte segments of a wave tile only
// merge into full lines in L2 when the stores are cacheable -- 58 % -> 42 % of the roofline at 1.4 GB with the hint -- and
// the two-output Uniform sampler gained nothing.)
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
// `im` / `RD`: index of i0 inside the [R*D] parameter plane and the plane size, known to the row kernels for free
// (RD == 0: unknown); operands repeated over the particles (P == RD) then need no division.
template <typename T, bool VEC>
__device__ __forceinline__ void ldp4(const T* __restrict__ p, int64_t P, int64_t i0, int n, T out[4], T pad, int64_t im = 0,
                                     int64_t RD = 0) {
  if (P == 1) {
    const T v = p[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v;
    return;
  }
  if (VEC) {
    const int64_t idx = i0 < P ? i0 : (P == RD ? im : mod_fast(i0, P));
    ld4<T, true>(p, idx, 4, out);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = j < n ? p[(i0 + j) < P ? (i0 + j) : mod_fast(i0 + j, P)] : pad;
  }
}
// Operand classes: how a periodic operand is addressed is decided on the host and compiled in, because ANY branch
// between two global loads (even a uniform one) makes the compiler wait for the first before issuing the second.
enum { C_GEN = 0, C_FULL = 1, C_PLANE = 2, C_SCALAR = 3 };   // generic a[i % P] | a[i] | a[i % (R*D)] = a[im] | a[0]
template <typename T, bool VEC, int CLS>
__device__ __forceinline__ void ldc4(const T* __restrict__ p, int64_t P, int64_t i0, int n, T out[4], T pad, int64_t im,
                                     int64_t RD) {
  if (CLS == C_SCALAR) {
    const T v = p[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v;
  } else if (CLS == C_FULL) {
    ld4<T, VEC>(p, i0, n, out);
  } else if (CLS == C_PLANE) {
    ld4<T, VEC>(p, im, n, out);
  } else {
    ldp4<T, VEC>(p, P, i0, n, out, pad, im, RD);
  }
}
inline int operand_class(int64_t P, int64_t N, int64_t RD) {
  return P == N ? C_FULL : (P == 1 ? C_SCALAR : (P == RD ? C_PLANE : C_GEN));
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
// Row functors expose load() (global reads only) and finish() (arithmetic + stores) separately so that the tile
// kernel can issue the reads of all its groups before the first dependent instruction.
template <typename T>
struct Regs3 {
  T a[4], b[4], c[4];
};
template <typename T, bool HAS_U, int CP = C_GEN>
struct LogisticSampleF {   // L1.  CP: C_PLANE inside the tile kernel (im known), C_GEN elsewhere
  const T* loc; const T* scale; const T* u; uint64_t seed, call; const uint64_t* rs; T* z; int64_t M; uint64_t* used;

  __device__ void prepare() {
    if (rs) { seed = rs[0]; call += rs[1]; }
    // resolved ids for the backward call, which may run after the caller has advanced the live rng_state
    if (used && blockIdx.x == 0 && threadIdx.x == 0) { used[0] = seed; used[1] = call; }
  }
  template <int ACC>
  __device__ __forceinline__ void load(int64_t i0, int n, int64_t im, int64_t RD, Regs3<T>& r) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    ldc4<T, VEC, CP>(loc, M, i0, n, r.a, (T)0, im, RD);
    ldc4<T, VEC, CP>(scale, M, i0, n, r.b, (T)1, im, RD);
    if (HAS_U) ld4<T, VEC>(u, i0, n, r.c);
  }
  template <int ACC>
  __device__ __forceinline__ void finish(int64_t i0, int n, Regs3<T>& r, T t[4], bool want) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    T zz[4];
    if (!HAS_U) draw4<T, VEC>(nullptr, i0, n, seed, call, r.c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const T lu = t_log(r.c[j]), l1 = t_log((T)1 - r.c[j]);
      zz[j] = mul_add_2round(r.a[j], r.b[j], lu - l1);
      // log-density of the fresh sample: -eps - 2*softplus(-eps) = log(u) + log(1-u)
      if (want) t[j] = (lu + l1) - t_log(r.b[j]);
    }
    st4<T, VEC>(z, i0, n, zz);
  }
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool want, int64_t im, int64_t RD) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    Regs3<T> r;
    load<ACC>(i0, n, im, RD, r);
    finish<ACC>(i0, n, r, t, want);
  }
};

template <typename T, int CX = C_GEN, int CP = C_GEN>
struct LogisticLogProbF {   // L2
  const T* x; int64_t Px; const T* loc; int64_t Pm; const T* scale; int64_t Ps;
  __device__ void prepare() {}
  template <int ACC>
  __device__ __forceinline__ void load(int64_t i0, int n, int64_t im, int64_t RD, Regs3<T>& r) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    ldc4<T, VEC, CX>(x, Px, i0, n, r.c, (T)0, im, RD);
    ldc4<T, VEC, CP>(loc, Pm, i0, n, r.a, (T)0, im, RD);
    ldc4<T, VEC, CP>(scale, Ps, i0, n, r.b, (T)1, im, RD);
  }
  template <int ACC>
  __device__ __forceinline__ void finish(int64_t, int, Regs3<T>& r, T t[4], bool) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = logistic_term(r.c[j], r.a[j], r.b[j]);
  }
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool want, int64_t im, int64_t RD) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    Regs3<T> r;
    load<ACC>(i0, n, im, RD, r);
    finish<ACC>(i0, n, r, t, want);
  }
};

template <typename T, int CX = C_GEN, int CP = C_GEN>
struct UniformLogProbF {   // U2
  const T* x; int64_t Px; const T* low; int64_t Pl; const T* high; int64_t Ph;
  __device__ void prepare() {}
  template <int ACC>
  __device__ __forceinline__ void load(int64_t i0, int n, int64_t im, int64_t RD, Regs3<T>& r) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    ldc4<T, VEC, CX>(x, Px, i0, n, r.c, (T)0, im, RD);
    ldc4<T, VEC, CP>(low, Pl, i0, n, r.a, (T)0, im, RD);
    ldc4<T, VEC, CP>(high, Ph, i0, n, r.b, (T)1, im, RD);
  }
  template <int ACC>
  __device__ __forceinline__ void finish(int64_t, int, Regs3<T>& r, T t[4], bool) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool inside = (r.a[j] <= r.c[j]) && (r.b[j] > r.c[j]);   // torch Uniform.log_prob: lb * ub
      t[j] = (inside ? (T)0 : (T)(-INFINITY)) - t_log(r.b[j] - r.a[j]);
    }
  }
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n, T t[4], bool want, int64_t im, int64_t RD) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
    Regs3<T> r;
    load<ACC>(i0, n, im, RD, r);
    finish<ACC>(i0, n, r, t, want);
  }
};

template <typename T>
struct UniformSampleF {   // U1
  const T* low; int64_t Pl; const T* high; int64_t Ph; const T* u; uint64_t seed, call; const uint64_t* rs;
  T* out; T* cache; int reparam;
  __device__ void prepare() { if (rs) { seed = rs[0]; call += rs[1]; } }
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
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
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
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
  template <int ACC>
  __device__ __forceinline__ void eval(int64_t i0, int n) const {
    constexpr bool VEC = ACC != 0;
    (void)VEC;
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

constexpr int kTile = 1024;                         // elements of one wave tile
constexpr int kTileRows = 256;                      // at most this many rows per tile
constexpr int kTileLds = kTile + kTile / 32 + kTileRows;   // padded terms + the row results

// Row sums over wave tiles.  Tile (rt, kt) = rows r0 .. r0+rr-1 of particles k0 .. k0+kk-1, i.e. kk segments of rr*D
// contiguous elements; element e of the tile is (seg, off) = divmod(e, rr*D).  Tile-local row q = seg*rr + off/D lies
// at term[q*D .. q*D+D).  Results are written in the order (rrow, seg): runs of kk consecutive k for the K-fastest
// layout, runs of rr consecutive r for the row-major one (kk == 1).
// MODE 0: scalar accesses, one LDS slot per element; 1: 16-byte accesses, one slot per element;
// 2: 16-byte accesses and D % 4 == 0: a group of 4 elements never straddles a row, so only its partial sum is parked
// (4x less LDS traffic, 4x fewer adds in the row pass).
template <typename T, typename F, int MODE>
__global__ __launch_bounds__(256) void k_wave_rows(F f, T* __restrict__ lp, int64_t K, int64_t R, int D, int rr, int kk, int lgG,
                                                   int64_t sk, int64_t sr, int direct) {
  constexpr bool VEC = MODE != 0, PART = MODE == 2;
  constexpr int ACC = VEC ? 1 : 0;
  __shared__ T lds[4][kTileLds];
  T* __restrict__ term = lds[threadIdx.x >> 6];
  T* __restrict__ res = term + (kTile + kTile / 32);
  f.prepare();
  const bool want = lp != nullptr;
  const int lane = threadIdx.x & 63;
  const int G = 1 << lgG;
  const int seg_len = rr * D;                 // <= kTile
  const int64_t RD = R * (int64_t)D;
  const int64_t rtiles = (R + rr - 1) / rr, ktiles = (K + kk - 1) / kk;
  const int64_t tiles = rtiles * ktiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  // this lane's (up to) 4 groups of a tile: (segment, offset) and LDS slot are the same for every tile
  int segj[4], offj[4], ldsj[4];
  int64_t soj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane * 4 + 256 * j;
    segj[j] = (int)((uint32_t)e / (uint32_t)seg_len);
    offj[j] = e - segj[j] * seg_len;
    soj[j] = segj[j] * RD + offj[j];
    ldsj[j] = PART ? pad_idx(e >> 2) : pad_idx(e);   // e % 4 == 0: the four slots e .. e+3 never straddle a pad
  }
  const int DD = PART ? D >> 2 : D, SL = PART ? seg_len >> 2 : seg_len;   // row / segment length in LDS slots
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += nwaves) {
    int64_t kt, rt;
    divmod(t, rtiles, kt, rt);
    const int64_t r0 = rt * rr, k0 = kt * kk;
    const int nr = (int)((R - r0 < rr) ? (R - r0) : rr);   // valid rows per segment
    const int nk = (int)((K - k0 < kk) ? (K - k0) : kk);   // valid segments
    const int seg_valid = nr * D;
    const int64_t im0 = r0 * D;
    const int64_t base = k0 * RD + im0;
    Regs3<T> rg[4];
    int cnt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // All global reads of the tile first, and UNCONDITIONALLY: a group outside the tile reads the tile's first
      // elements instead.  (A divergent guard around the loads makes the compiler wait for each group's data before
      // it issues the next group's loads: 34 % -> 69 % of the HBM roofline in tools/rows_variants.hip.)
      const int c = seg_valid - offj[j];
      cnt[j] = (segj[j] < nk && c > 0) ? (c < 4 ? c : 4) : 0;
      const bool on = cnt[j] > 0;
      f.template load<ACC>(base + (on ? soj[j] : 0), on ? cnt[j] : (VEC ? 4 : 1), im0 + (on ? offj[j] : 0), RD, rg[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cnt[j] <= 0) continue;
      T tv[4] = {(T)0, (T)0, (T)0, (T)0};
      f.template finish<ACC>(base + soj[j], cnt[j], rg[j], tv, want);
      if (!want) continue;
      if (direct) {
        st4<T, VEC>(lp, base + soj[j], cnt[j], tv);
      } else if (PART) {
        term[ldsj[j]] = (tv[0] + tv[1]) + (tv[2] + tv[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < cnt[j]) term[ldsj[j] + q] = tv[q];
      }
    }
    if (want && !direct) {
      // same-wave LDS hand-off: the DS queue of a wave is in order, no workgroup barrier needed
      __builtin_amdgcn_wave_barrier();
      const int nrows = nk * nr;              // <= kTileRows
      const int g = lane & (G - 1);
      for (int w = lane >> lgG; w < nrows; w += 64 >> lgG) {     // w = rrow * nk + seg: output order, k fastest
        const int rrow = (int)((uint32_t)w / (uint32_t)nk), seg = w - rrow * nk;
        const int b0 = seg * SL + rrow * DD;
        T acc = (T)0;
        for (int d = g; d < DD; d += G) acc += term[pad_idx(b0 + d)];
        for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, ZS_WAVE);
        if (g == 0) res[w] = acc;
      }
      __builtin_amdgcn_wave_barrier();
      for (int o = lane; o < nrows; o += 64) {
        const int rrow = (int)((uint32_t)o / (uint32_t)nk), seg = o - rrow * nk;
        lp[(k0 + seg) * sk + (r0 + rrow) * sr] = res[o];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Rows longer than a tile: one workgroup per row.
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
      f.template eval<(VEC ? 1 : 0)>(e0 + e, cnt, t, want, 0, 0);
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

template <typename T, typename F, int ACC>
__global__ __launch_bounds__(256) void k_elem(F f, int64_t N) {
  f.prepare();
  const int64_t groups = (N + 3) >> 2;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i0 = g << 2;
    f.template eval<ACC>(i0, N - i0 < 4 ? (int)(N - i0) : 4);
  }
}

// L1 backward: workgroup = 64 groups of 4 parameters x 4 K-slices, slices combined through LDS.
// SAME_ROW (D % 4 == 0): the four elements of a group share their row, so the incoming log-prob gradient is ONE unguarded
// load per particle, issued together with the gz loads (four guarded gathers serialised the loop: 45 -> see DESIGN 4a).
template <typename T, bool VEC, bool SAME_ROW>
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
    const int64_t ks = __builtin_amdgcn_readfirstlane(slice);      // wave-uniform: the particle loop runs on scalar counters
    for (int64_t k = ks; k < K; k += 4) {
      const int64_t i0 = k * M + m0;
      T grow = (T)0;
      if (SAME_ROW && glp) grow = glp[k * gsk + r[0] * gsr];
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
      if (SAME_ROW) {
#pragma unroll
        for (int j = 0; j < 4; ++j) gl[j] += grow;
      } else if (glp) {
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
  const int64_t rows = K * R, RD = R * D, N = rows * D;
  vec_ok = vec_ok && (N & 3) == 0;
  if (D > kTile) {
    const bool vec = vec_ok && (D & 3) == 0;
    const dim3 grid(grid_for(rows, 1, 256u * 32u));
    if (vec) ZS_LAUNCH(kid, (k_long_rows<T, F, true>), grid, dim3(256), st, f, lp, rows, R, D, sk, sr);
    else ZS_LAUNCH(kid, (k_long_rows<T, F, false>), grid, dim3(256), st, f, lp, rows, R, D, sk, sr);
    return 0;
  }
  // tile geometry: rr rows x kk particles (see k_wave_rows)
  const int d = (int)D;
  const int rr0 = (d & 3) == 0 ? 1 : ((d & 1) == 0 ? 2 : 4);      // smallest rr with rr*D % 4 == 0
  int rr = 0, kk = 1;
  if (lp != nullptr && K > 1 && sk == 1) {                        // K-fastest result: runs of kk particles per row
    rr = rr0 * ((32 + rr0 * d - 1) / (rr0 * d));                  // segments of >= 32 contiguous elements
    if (rr > R) rr = (int)R;
    int64_t k2 = kTile / (rr * d);
    if (k2 > kTileRows / rr) k2 = kTileRows / rr;
    if (k2 > K) k2 = K;
    kk = (int)k2;
    if (kk < 8) rr = 0;
  }
  if (rr == 0) {                                                  // row-major tiles
    kk = 1;
    rr = kTile / d;
    if (rr > kTileRows) rr = kTileRows;
    if (rr >= rr0) rr -= rr % rr0;
    if (rr > R) rr = (int)R;
    // small problems: more tiles than wave slots
    while (rr >= 2 * rr0 && rr * d >= 512 && ((R + rr - 1) / rr) * K < 2048) rr = ((rr / 2) / rr0) * rr0;
  }
  const bool direct = lp != nullptr && D == 1 && sr == 1 && (K == 1 || sk == R);
  const bool vec = vec_ok && (K == 1 || (RD & 3) == 0) && ((((int64_t)rr * D) & 3) == 0 || rr >= R) &&
                   (!direct || al(lp, sizeof(T) * 4));
  const bool part = vec && (d & 3) == 0 && !direct && lp != nullptr;
  const int nrows = rr * kk, dd = part ? d >> 2 : d;
  int lgG = 0;
  while (lgG < 6 && (64 >> (lgG + 1)) >= nrows && (2 << lgG) <= dd) ++lgG;
  const int64_t tiles = ((R + rr - 1) / rr) * ((K + kk - 1) / kk);
  const dim3 grid(grid_for(tiles, 4, 256u * 16u));
  if (part)
    ZS_LAUNCH(kid, (k_wave_rows<T, F, 2>), grid, dim3(256), st, f, lp, K, R, d, rr, kk, lgG, sk, sr, (int)direct);
  else if (vec)
    ZS_LAUNCH(kid, (k_wave_rows<T, F, 1>), grid, dim3(256), st, f, lp, K, R, d, rr, kk, lgG, sk, sr, (int)direct);
  else
    ZS_LAUNCH(kid, (k_wave_rows<T, F, 0>), grid, dim3(256), st, f, lp, K, R, d, rr, kk, lgG, sk, sr, (int)direct);
  return 0;
}

template <typename T, typename F>
int launch_elem(int kid, F f, bool vec_ok, int64_t N, hipStream_t st) {
  const dim3 grid(grid_for((N + 3) / 4, 256));
  if (vec_ok && (N & 3) == 0) {
    ZS_LAUNCH(kid, (k_elem<T, F, 1>), grid, dim3(256), st, f, N);
  } else {
    ZS_LAUNCH(kid, (k_elem<T, F, 0>), grid, dim3(256), st, f, N);
  }
  return 0;
}

template <typename T>
int logistic_sample(const T* loc, const T* scale, const T* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, T* z,
                    T* lp, int64_t K, int64_t M, int64_t D, int64_t sk, int64_t sr, uint64_t* rng_used, void* stream) {
  if (K < 1 || M < 0 || D < 1 || (M % D) != 0) return ZS_EINVAL;
  if (M == 0) return 0;
  if (!loc || !scale || !z) return ZS_EINVAL;
  const size_t A = sizeof(T) * 4;
  const bool vec = (M & 3) == 0 && al(loc, A) && al(scale, A) && al(u, A) && al(z, A);
  if constexpr (std::is_same<T, float>::value) {
    // u handed in (the parity route): the flat-plane given-stream kernel in its SAMPLE mode -- against the generic wave-tile path
    // below: 131 k rows 47 -> 67 %, 1 M rows 53 -> 70-72 %, 4.2 M rows 53-58 -> 73 % of the roofline (>= 62)
    static const int given_tile_env = env_knob("ZS_K1_GIVEN_TILE", 1);      // experiments only: 0 = the generic path
    if (u && given_tile_env && vec && (D & 3) == 0) {
      const int D4 = (int)(D / 4);
      const int64_t R = M / D;
      const K1Tile g = k1_tile(K, R, D4, true);
      if (g.ok) {
        hipStream_t st = (hipStream_t)stream;
        const bool big = (double)K * (double)M * 8.0 > 268435456.0;
        if (big)
          ZS_LAUNCH_SMEM(KID_LOGISTIC_SAMPLE, (k_logprob_tile<D_LOGISTIC, true, true>), dim3(g.grid), dim3(g.threads), g.smem, st,
                         (const float4*)u, (const float4*)loc, (const float4*)scale, lp, (uint32_t)K, R, (uint32_t)D4, (uint32_t)(R * D4),
                         g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, false, (float4*)z);
        else
          ZS_LAUNCH_SMEM(KID_LOGISTIC_SAMPLE, (k_logprob_tile<D_LOGISTIC, false, true>), dim3(g.grid), dim3(g.threads), g.smem, st,
                         (const float4*)u, (const float4*)loc, (const float4*)scale, lp, (uint32_t)K, R, (uint32_t)D4, (uint32_t)(R * D4),
                         g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, false, (float4*)z);
        ZS_CHECK_LAUNCH();
        return 0;
      }
    }
    // in-kernel Philox, fp32, rows of up to 256 elements: the fused flat-plane kernel shared with Normal (zs_sample_tile.h)
    if (!u && vec && (D & 3) == 0) {
      const int D4 = (int)(D / 4);
      const int64_t R = M / D;
      const K1Tile g = k1_tile(K, R, D4, lp != nullptr);
      if (g.ok) {
        hipStream_t st = (hipStream_t)stream;
        const bool nt = (double)K * (double)M * 4.0 > 268435456.0;   // z cannot stay in the Infinity Cache
#define ZS_LAUNCH_LTILE(L, NTF)                                                                                          \
  ZS_LAUNCH_SMEM(KID_LOGISTIC_SAMPLE, (k_sample_tile<D_LOGISTIC, L, NTF>), dim3(g.grid), dim3(g.threads), g.smem, st,     \
                 (const float4*)loc, (const float4*)scale, seed, offset, rng_state, (float4*)z, lp, (uint32_t)K, R,      \
                 (uint32_t)D4, (uint32_t)(R * D4), g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, false, rng_used,     \
                 (float4*)nullptr, false, 0u)
        if (nt) { if (lp) ZS_LAUNCH_LTILE(true, true); else ZS_LAUNCH_LTILE(false, true); }
        else    { if (lp) ZS_LAUNCH_LTILE(true, false); else ZS_LAUNCH_LTILE(false, false); }
#undef ZS_LAUNCH_LTILE
        ZS_CHECK_LAUNCH();
        return 0;
      }
    }
  }
  // D <= kTile: the tile kernel knows the index inside the [R*D] parameter plane (C_PLANE); the long-row kernel does not
  if (D <= kTile) {
    if (u) {
      LogisticSampleF<T, true, C_PLANE> f = {loc, scale, u, seed, offset, rng_state, z, M, rng_used};
      launch_rows<T>(KID_LOGISTIC_SAMPLE, f, vec, lp, K, M / D, D, sk, sr, (hipStream_t)stream);
    } else {
      LogisticSampleF<T, false, C_PLANE> f = {loc, scale, u, seed, offset, rng_state, z, M, rng_used};
      launch_rows<T>(KID_LOGISTIC_SAMPLE, f, vec, lp, K, M / D, D, sk, sr, (hipStream_t)stream);
    }
  } else if (u) {
    LogisticSampleF<T, true> f = {loc, scale, u, seed, offset, rng_state, z, M, rng_used};
    launch_rows<T>(KID_LOGISTIC_SAMPLE, f, vec, lp, K, M / D, D, sk, sr, (hipStream_t)stream);
  } else {
    LogisticSampleF<T, false> f = {loc, scale, u, seed, offset, rng_state, z, M, rng_used};
    launch_rows<T>(KID_LOGISTIC_SAMPLE, f, vec, lp, K, M / D, D, sk, sr, (hipStream_t)stream);
  }
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
  if (vec && (D & 3) == 0)
    ZS_LAUNCH(KID_LOGISTIC_SAMPLE_BWD, (k_logistic_sample_bwd<T, true, true>), grid, dim3(256), (hipStream_t)stream, scale, u,
              seed, offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D);
  else if (vec)
    ZS_LAUNCH(KID_LOGISTIC_SAMPLE_BWD, (k_logistic_sample_bwd<T, true, false>), grid, dim3(256), (hipStream_t)stream, scale, u,
              seed, offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D);
  else
    ZS_LAUNCH(KID_LOGISTIC_SAMPLE_BWD, (k_logistic_sample_bwd<T, false, false>), grid, dim3(256), (hipStream_t)stream, scale, u,
              seed, offset, rng_state, gz, glp, gsk, gsr, gloc, gscale, K, M, D);
  ZS_CHECK_LAUNCH();
  return 0;
}

// shared argument checks of the [K, R, D] problems with three periodic operands
inline int check3(int64_t K, int64_t R, int64_t D, int64_t P1, int64_t P2, int64_t P3, int64_t* N) {
  if (K < 1 || R < 0 || D < 1 || P1 < 1 || P2 < 1 || P3 < 1) return ZS_EINVAL;
  *N = K * R * D;
  return 0;
}

// The value streams in full while both parameters share one class (the cases the callers produce: parameters of the
// full shape, repeated over the particles, or scalar); anything else, and rows longer than a tile (no plane index),
// takes the generic a[i % P] functor.
#define ZS_DISPATCH_CLASSES(FUNCTOR, KID, x, Px, a, Pa, b, Pb)                                                       \
  do {                                                                                                               \
    const int cx = operand_class(Px, N, R * D), ca = operand_class(Pa, N, R * D), cb = operand_class(Pb, N, R * D);  \
    const hipStream_t st__ = (hipStream_t)stream;                                                                    \
    if (D <= kTile && cx == C_FULL && ca == cb && ca == C_FULL) {                                                    \
      FUNCTOR<T, C_FULL, C_FULL> f = {x, Px, a, Pa, b, Pb};                                                          \
      launch_rows<T>(KID, f, vec, lp, K, R, D, sk, sr, st__);                                                        \
    } else if (D <= kTile && cx == C_FULL && ca == cb && ca == C_PLANE) {                                            \
      FUNCTOR<T, C_FULL, C_PLANE> f = {x, Px, a, Pa, b, Pb};                                                         \
      launch_rows<T>(KID, f, vec, lp, K, R, D, sk, sr, st__);                                                        \
    } else if (D <= kTile && cx == C_FULL && ca == cb && ca == C_SCALAR) {                                           \
      FUNCTOR<T, C_FULL, C_SCALAR> f = {x, Px, a, Pa, b, Pb};                                                        \
      launch_rows<T>(KID, f, vec, lp, K, R, D, sk, sr, st__);                                                        \
    } else {                                                                                                         \
      FUNCTOR<T> f = {x, Px, a, Pa, b, Pb};                                                                          \
      launch_rows<T>(KID, f, vec, lp, K, R, D, sk, sr, st__);                                                        \
    }                                                                                                                \
  } while (0)

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
  if constexpr (sizeof(T) == 4) {
    // parameters [R, D] shared by the K particles, rows of up to 256 elements: the K2 mapping (per-lane 1/scale and
    // log scale formed once, value rows streamed past them)
    if (vec && K > 1 && D % 4 == 0 && D / 4 <= 64 && Px == N && Pm == R * D && Ps == R * D) {
      launch_logprob_krep<D_LOGISTIC>(KID_LOGISTIC_LOGPROB, (const float*)x, (const float*)loc, (const float*)scale, (float*)lp, K, R,
                                      (int)(D / 4), sk, sr, false, (hipStream_t)stream);
      ZS_CHECK_LAUNCH();
      return 0;
    }
  }
  ZS_DISPATCH_CLASSES(LogisticLogProbF, KID_LOGISTIC_LOGPROB, x, Px, loc, Pm, scale, Ps);
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

// L2 backward reduced over the K particles, parameters [R, D] repeated over them: the float32 form is the kernel shared
// with Normal (zs_sample_tile.h); float64 (untuned, like every fp64 twin): a thread per parameter element.
template <typename T>
__global__ __launch_bounds__(256) void k_logistic_logprob_bwd_ksum_serial(
    const T* __restrict__ x, const T* __restrict__ loc, const T* __restrict__ scale, const T* __restrict__ glp, int64_t gsk,
    int64_t gsr, T* __restrict__ gx, T* __restrict__ gloc, T* __restrict__ gscale, int64_t K, int64_t M, int64_t D) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = m / D;
    const T a = loc[m], b = scale[m];
    T sl = (T)0, ss = (T)0;
    for (int64_t k = 0; k < K; ++k) {
      const T g = glp[k * gsk + r * gsr];
      const T t = (x[k * M + m] - a) / b;
      const T e = t_exp(-t_abs(t));
      T h = ((T)1 - e) / ((T)1 + e);            // tanh(|t|/2)
      h = t < (T)0 ? -h : h;
      const T gh = g * h / b;
      sl += gh;
      ss += g * (h * t - (T)1) / b;
      if (gx) gx[k * M + m] = -gh;
    }
    if (gloc) gloc[m] = sl;
    if (gscale) gscale[m] = ss;
  }
}

template <typename T>
int logistic_logprob_bwd_ksum(const T* x, const T* loc, const T* scale, const T* glp, int64_t gsk, int64_t gsr, T* gx, T* gloc,
                              T* gscale, int64_t K, int64_t R, int64_t D, void* stream) {
  if (K < 1 || R < 0 || D < 1) return ZS_EINVAL;
  const int64_t M = R * D;
  if (M == 0) return 0;
  if (!x || !loc || !scale || !glp) return ZS_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if constexpr (std::is_same<T, float>::value) {
    launch_logprob_bwd_ksum<D_LOGISTIC>(KID_LOGISTIC_LOGPROB_BWD_KSUM, x, loc, scale, glp, gsk, gsr, gx, gloc, gscale, K, R, D, false, st);
  } else {
    ZS_LAUNCH(KID_LOGISTIC_LOGPROB_BWD_KSUM, (k_logistic_logprob_bwd_ksum_serial<T>), dim3(grid_for(M, 256)), dim3(256), st, x, loc, scale,
              glp, gsk, gsr, gx, gloc, gscale, K, M, D);
  }
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
  if constexpr (std::is_same<T, float>::value) {
    // in-kernel Philox, bounds of one period shared by the K = N / P repetitions: the flat-plane kernel of Normal / Logistic
    // (zs_sample_tile.h) with two store streams and no density (scalar loop control, one store instruction per output and
    // particle, non-temporal beyond the Infinity Cache)
    if (!u && vec && Pl == Ph && Pl >= 4 && (N & 3) == 0) {
      const int64_t K = N / Pl;
      const K1Tile g = k1_tile(K, Pl / 4, 1, false);
      if (g.ok) {
        hipStream_t st = (hipStream_t)stream;
        const bool nt = (double)N * (cache ? 8.0 : 4.0) > 268435456.0;
        if (nt)
          ZS_LAUNCH_SMEM(KID_UNIFORM_SAMPLE, (k_sample_tile<D_UNIFORM, false, true>), dim3(g.grid), dim3(g.threads), 0, st,
                         (const float4*)low, (const float4*)high, seed, offset, rng_state, (float4*)out, (float*)nullptr, (uint32_t)K,
                         Pl / 4, 1u, (uint32_t)(Pl / 4), g.kchunk, g.KB, g.n_ptiles, g.total, (int64_t)0, (int64_t)0, false,
                         (uint64_t*)nullptr, (float4*)cache, reparam != 0, 0u);
        else
          ZS_LAUNCH_SMEM(KID_UNIFORM_SAMPLE, (k_sample_tile<D_UNIFORM, false, false>), dim3(g.grid), dim3(g.threads), 0, st,
                         (const float4*)low, (const float4*)high, seed, offset, rng_state, (float4*)out, (float*)nullptr, (uint32_t)K,
                         Pl / 4, 1u, (uint32_t)(Pl / 4), g.kchunk, g.KB, g.n_ptiles, g.total, (int64_t)0, (int64_t)0, false,
                         (uint64_t*)nullptr, (float4*)cache, reparam != 0, 0u);
        ZS_CHECK_LAUNCH();
        return 0;
      }
    }
  }
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
  if constexpr (sizeof(T) == 4) {
    if (vec && K > 1 && D % 4 == 0 && D / 4 <= 64 && Px == N && Pl == R * D && Ph == R * D) {   // bounds shared by the particles
      launch_logprob_krep<D_UNIFORM>(KID_UNIFORM_LOGPROB, (const float*)x, (const float*)low, (const float*)high, (float*)lp, K, R,
                                     (int)(D / 4), sk, sr, false, (hipStream_t)stream);
      ZS_CHECK_LAUNCH();
      return 0;
    }
  }
  ZS_DISPATCH_CLASSES(UniformLogProbF, KID_UNIFORM_LOGPROB, x, Px, low, Pl, high, Ph);
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
                                                 int64_t sk, int64_t sr, uint64_t* rng_used, void* stream) {                       \
    return logistic_sample<T>(loc, scale, u, seed, offset, rng_state, z, lp, K, M, D, sk, sr, rng_used, stream);                   \
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
  extern "C" int zs_logistic_logprob_bwd_ksum##SFX(const T* x, const T* loc, const T* scale, const T* glp, int64_t gsk,             \
                                                   int64_t gsr, T* gx, T* gloc, T* gscale, int64_t K, int64_t R, int64_t D,          \
                                                   void* stream) {                                                                   \
    return logistic_logprob_bwd_ksum<T>(x, loc, scale, glp, gsk, gsr, gx, gloc, gscale, K, R, D, stream);                            \
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
