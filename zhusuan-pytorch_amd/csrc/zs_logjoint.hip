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
  return __hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ticket_return(unsigned* t) { __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ================================================================ LJ1
// Launch layout: every term owns a run of workgroups (a table in the kernel arguments, as zs_adam.hip has for 32 tensors).
// A thread handles LJ_U groups of 4 consecutive elements per round and issues ALL their loads before any arithmetic: at
// these sizes a load is a 1-2 us round trip and the duration of the kernel is the number of DEPENDENT rounds, not its
// bytes.  The loops are compiled per (family, operand pattern): inside them there is no branch between two loads (a
// uniform branch there makes the compiler wait for the first load before it issues the second):
//   vector form   every non-scalar operand is 16-byte aligned with a period that is a multiple of 4 (and at least one
//                 operand has full size): one dwordx4 load per operand and group; a scalar operand's load is redirected
//                 to a full-size operand's address (harmless) and its value selected from a register;
//   element form  anything else: one element per thread and load, index i, i % P or 0 chosen by a select.
// Indices are 32-bit (n < 2^31 per term; larger problems are not launch-bound and take the per-node kernels).
constexpr int LJ_BLOCK = 256;
constexpr int LJ_U = 4;                      // groups (vector form) / elements (element form) a thread has in flight
constexpr unsigned LJ_MAX_BLOCKS = 2048;      // 3 doubles of workspace per workgroup: 6144 <= ZS_LJ_WORKSPACE
enum { CLS_NONE = 0, CLS_SCALAR = 1, CLS_FULL = 2, CLS_PERIODIC = 3 };

template <typename T>
struct LJTerm {
  const T *x, *a, *b;
  const T* dummy;             // vector form: a full-size, aligned operand (where the loads of scalar operands are pointed)
  T *gx, *ga, *gb;
  uint32_t n, px, pa, pb;
  double coef;
  int family;
  unsigned block0, nblocks;   // the element-wise workgroups of this term
  unsigned char cx, ca, cb;   // operand classes (CLS_*)
  unsigned char vec, periodic;   // vector form; some operand is CLS_PERIODIC (the vector loop then computes i % P)
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

// ---- per-family element arithmetic
template <typename T, int FAM>
struct Fam {
  static constexpr bool has_a = FAM != ZS_LJ_ROWS;
  static constexpr bool has_b = FAM == ZS_LJ_NORMAL || FAM == ZS_LJ_NORMAL_LOGSTD;
  static __device__ __forceinline__ T term(T x, T a, T b) {
    typedef Mth<T> M;
    if (FAM == ZS_LJ_ROWS) return x;
    if (has_b) {
      const T sg = M::sigma_of(b, FAM == ZS_LJ_NORMAL_LOGSTD);
      T logstd, prec;
      M::parts(sg, logstd, prec);
      return M::normal_term(x - a, logstd, prec);
    }
    return M::bern_term(FAM == ZS_LJ_BERNOULLI_LOGITS ? M::sigmoid(a) : a, x);
  }
  // d term / d (x, a, b) times gc = g * coef
  static __device__ __forceinline__ void partials(T x, T a, T b, T gc, T& dx, T& da, T& db) {
    typedef Mth<T> M;
    if (has_b) {
      const bool ls = FAM == ZS_LJ_NORMAL_LOGSTD;
      const T sg = M::sigma_of(b, ls);
      T logstd, prec;
      M::parts(sg, logstd, prec);
      const T d = x - a;
      const T u = gc * prec * d;
      dx = -u;
      da = u;
      const T v = gc * (prec * d * d - (T)1);
      db = ls ? v : v / sg;                     // d/d log std = sigma * d/d sigma
    } else {                                    // Bernoulli: d/d probs, or d/d logits = d/dp * p * (1 - p)
      dx = (T)0;
      db = (T)0;
      if (FAM == ZS_LJ_BERNOULLI_LOGITS) {
        const T p = M::sigmoid(a);
        da = gc * M::bern_dp(p, x) * p * ((T)1 - p);
      } else {
        da = gc * M::bern_dp(a, x);
      }
    }
  }
};

__device__ __forceinline__ uint32_t lj_idx(unsigned cls, uint32_t i, uint32_t P) {
  return cls == CLS_FULL ? i : (cls == CLS_PERIODIC ? i % P : 0u);
}

// which term owns this workgroup (uniform; the table lives in the kernel-argument segment: scalar loads)
template <typename T>
__device__ __forceinline__ int lj_term_of(const LJArgs<T>& A, unsigned blk) {
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (A.t[j].nblocks && blk >= A.t[j].block0) ti = j;
  return __builtin_amdgcn_readfirstlane(ti);
}

// ---- forward loops: the sum of this thread's terms
template <typename T, int FAM, bool PER>
__device__ __forceinline__ double lj_fwd_vec(const LJTerm<T>& t, unsigned blk) {
  typedef Fam<T, FAM> F;
  const uint32_t n = t.n, ng = n >> 2, stride = t.nblocks * LJ_BLOCK;
  const bool sx = t.cx == CLS_SCALAR, sa = t.ca == CLS_SCALAR, sb = t.cb == CLS_SCALAR;
  const T xs = t.x[0], as = F::has_a ? t.a[0] : (T)0, bs = F::has_b ? t.b[0] : (T)1;
  const T* __restrict__ px = sx ? t.dummy : t.x;
  const T* __restrict__ pa = (!F::has_a || sa) ? t.dummy : t.a;
  const T* __restrict__ pb = (!F::has_b || sb) ? t.dummy : t.b;
  const bool mx = PER && t.cx == CLS_PERIODIC, ma = PER && t.ca == CLS_PERIODIC, mb = PER && t.cb == CLS_PERIODIC;
  double acc = 0.0;
  for (uint32_t g0 = blk * LJ_BLOCK + threadIdx.x; g0 < ng; g0 += stride * LJ_U) {
    V4<T> x[LJ_U], a[LJ_U], b[LJ_U];
    bool live[LJ_U];
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      const uint32_t g = g0 + u * stride;
      live[u] = g < ng;
      const uint32_t i0 = (live[u] ? g : ng - 1) << 2;           // clamped: unconditional loads
      x[u] = *reinterpret_cast<const V4<T>*>(px + (mx ? i0 % t.px : i0));
      if (F::has_a) a[u] = *reinterpret_cast<const V4<T>*>(pa + (ma ? i0 % t.pa : i0));
      if (F::has_b) b[u] = *reinterpret_cast<const V4<T>*>(pb + (mb ? i0 % t.pb : i0));
    }
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      T s = (T)0;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s += F::term(sx ? xs : x[u].v[j], F::has_a ? (sa ? as : a[u].v[j]) : (T)0, F::has_b ? (sb ? bs : b[u].v[j]) : (T)1);
      acc += live[u] ? (double)s : 0.0;
    }
  }
  // the n % 4 elements after the last full group: the first threads of the term's first workgroup
  if (blk == 0 && threadIdx.x < (n & 3u)) {
    const uint32_t i = (ng << 2) + threadIdx.x;
    acc += (double)F::term(t.x[lj_idx(t.cx, i, t.px)], F::has_a ? t.a[lj_idx(t.ca, i, t.pa)] : (T)0,
                           F::has_b ? t.b[lj_idx(t.cb, i, t.pb)] : (T)1);
  }
  return acc;
}
template <typename T, int FAM>
__device__ __forceinline__ double lj_fwd_elem(const LJTerm<T>& t, unsigned blk) {
  typedef Fam<T, FAM> F;
  const uint32_t n = t.n, stride = t.nblocks * LJ_BLOCK;
  double acc = 0.0;
  for (uint32_t e0 = blk * LJ_BLOCK + threadIdx.x; e0 < n; e0 += stride * LJ_U) {
    T x[LJ_U], a[LJ_U], b[LJ_U];
    bool live[LJ_U];
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      const uint32_t e = e0 + u * stride;
      live[u] = e < n;
      const uint32_t i = live[u] ? e : n - 1;
      x[u] = t.x[lj_idx(t.cx, i, t.px)];
      a[u] = F::has_a ? t.a[lj_idx(t.ca, i, t.pa)] : (T)0;
      b[u] = F::has_b ? t.b[lj_idx(t.cb, i, t.pb)] : (T)1;
    }
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) acc += live[u] ? (double)F::term(x[u], a[u], b[u]) : 0.0;
  }
  return acc;
}
template <typename T, int FAM>
__device__ __forceinline__ double lj_fwd_family(const LJTerm<T>& t, unsigned blk) {
  if (!t.vec) return lj_fwd_elem<T, FAM>(t, blk);
  return t.periodic ? lj_fwd_vec<T, FAM, true>(t, blk) : lj_fwd_vec<T, FAM, false>(t, blk);
}

template <typename T>
__global__ __launch_bounds__(LJ_BLOCK) void k_logjoint_fwd(const LJArgs<T> A, T* __restrict__ out, double* __restrict__ ws,
                                                           unsigned* __restrict__ ticket) {
  __shared__ double sh[4];
  __shared__ bool last;
  const int ti = lj_term_of(A, blockIdx.x);
  const LJTerm<T>& t = A.t[ti];
  const unsigned blk = blockIdx.x - t.block0;
  double acc = 0.0;
  if (t.n > 0) {
    switch (t.family) {                          // uniform
      case ZS_LJ_ROWS: acc = lj_fwd_family<T, ZS_LJ_ROWS>(t, blk); break;
      case ZS_LJ_NORMAL: acc = lj_fwd_family<T, ZS_LJ_NORMAL>(t, blk); break;
      case ZS_LJ_NORMAL_LOGSTD: acc = lj_fwd_family<T, ZS_LJ_NORMAL_LOGSTD>(t, blk); break;
      case ZS_LJ_BERNOULLI: acc = lj_fwd_family<T, ZS_LJ_BERNOULLI>(t, blk); break;
      default: acc = lj_fwd_family<T, ZS_LJ_BERNOULLI_LOGITS>(t, blk); break;
    }
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    store_wt(ws + blockIdx.x, t.coef * acc);      // the only byte handed over: written through by the lane that signals
    drain_stores();
    last = (ticket_take(ticket) == gridDim.x - 1);
  }
  __syncthreads();
  if (last) {                                   // (uniform) one round of sc1 loads for up to 256 partials, fixed combination order
    double s = 0.0;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += LJ_BLOCK) s += load_wt(ws + i);
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) {
      out[0] = (T)s;
      ticket_return(ticket);
    }
  }
}

// ---- backward loops: full-size gradients are written, the sums for scalar operands returned through (sx, sa, sb)
template <typename T, int FAM, bool PER>
__device__ __forceinline__ void lj_bwd_vec(const LJTerm<T>& t, unsigned blk, T gc, double& rx, double& ra, double& rb) {
  typedef Fam<T, FAM> F;
  const uint32_t n = t.n, ng = n >> 2, stride = t.nblocks * LJ_BLOCK;
  const bool sx = t.cx == CLS_SCALAR, sa = t.ca == CLS_SCALAR, sb = t.cb == CLS_SCALAR;
  const T xs = t.x[0], as = t.a[0], bs = F::has_b ? t.b[0] : (T)1;
  const T* __restrict__ px = sx ? t.dummy : t.x;
  const T* __restrict__ pa = sa ? t.dummy : t.a;
  const T* __restrict__ pb = (!F::has_b || sb) ? t.dummy : t.b;
  const bool mx = PER && t.cx == CLS_PERIODIC, ma = PER && t.ca == CLS_PERIODIC, mb = PER && t.cb == CLS_PERIODIC;
  const bool full_x = F::has_b && t.gx && t.cx == CLS_FULL, full_a = t.ga && t.ca == CLS_FULL,
             full_b = F::has_b && t.gb && t.cb == CLS_FULL;
  for (uint32_t g0 = blk * LJ_BLOCK + threadIdx.x; g0 < ng; g0 += stride * LJ_U) {
    V4<T> x[LJ_U], a[LJ_U], b[LJ_U];
    uint32_t i0s[LJ_U];
    bool live[LJ_U];
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      const uint32_t g = g0 + u * stride;
      live[u] = g < ng;
      const uint32_t i0 = (live[u] ? g : ng - 1) << 2;
      i0s[u] = i0;
      x[u] = *reinterpret_cast<const V4<T>*>(px + (mx ? i0 % t.px : i0));
      a[u] = *reinterpret_cast<const V4<T>*>(pa + (ma ? i0 % t.pa : i0));
      if (F::has_b) b[u] = *reinterpret_cast<const V4<T>*>(pb + (mb ? i0 % t.pb : i0));
    }
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      V4<T> dx, da, db;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        F::partials(sx ? xs : x[u].v[j], sa ? as : a[u].v[j], F::has_b ? (sb ? bs : b[u].v[j]) : (T)1, gc, dx.v[j], da.v[j], db.v[j]);
        if (live[u]) { rx += (double)dx.v[j]; ra += (double)da.v[j]; rb += (double)db.v[j]; }
      }
      if (live[u]) {
        if (full_x) *reinterpret_cast<V4<T>*>(t.gx + i0s[u]) = dx;
        if (full_a) *reinterpret_cast<V4<T>*>(t.ga + i0s[u]) = da;
        if (full_b) *reinterpret_cast<V4<T>*>(t.gb + i0s[u]) = db;
      }
    }
  }
  if (blk == 0 && threadIdx.x < (n & 3u)) {
    const uint32_t i = (ng << 2) + threadIdx.x;
    T dx, da, db;
    F::partials(t.x[lj_idx(t.cx, i, t.px)], t.a[lj_idx(t.ca, i, t.pa)], F::has_b ? t.b[lj_idx(t.cb, i, t.pb)] : (T)1, gc, dx, da, db);
    rx += (double)dx; ra += (double)da; rb += (double)db;
    if (full_x) t.gx[i] = dx;
    if (full_a) t.ga[i] = da;
    if (full_b) t.gb[i] = db;
  }
}
template <typename T, int FAM>
__device__ __forceinline__ void lj_bwd_elem(const LJTerm<T>& t, unsigned blk, T gc, double& rx, double& ra, double& rb) {
  typedef Fam<T, FAM> F;
  const uint32_t n = t.n, stride = t.nblocks * LJ_BLOCK;
  const bool full_x = F::has_b && t.gx && t.cx == CLS_FULL, full_a = t.ga && t.ca == CLS_FULL,
             full_b = F::has_b && t.gb && t.cb == CLS_FULL;
  for (uint32_t e0 = blk * LJ_BLOCK + threadIdx.x; e0 < n; e0 += stride * LJ_U) {
    T x[LJ_U], a[LJ_U], b[LJ_U];
    uint32_t is[LJ_U];
    bool live[LJ_U];
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      const uint32_t e = e0 + u * stride;
      live[u] = e < n;
      const uint32_t i = live[u] ? e : n - 1;
      is[u] = i;
      x[u] = t.x[lj_idx(t.cx, i, t.px)];
      a[u] = t.a[lj_idx(t.ca, i, t.pa)];
      b[u] = F::has_b ? t.b[lj_idx(t.cb, i, t.pb)] : (T)1;
    }
#pragma unroll
    for (int u = 0; u < LJ_U; ++u) {
      T dx, da, db;
      F::partials(x[u], a[u], b[u], gc, dx, da, db);
      if (live[u]) {
        rx += (double)dx; ra += (double)da; rb += (double)db;
        if (full_x) t.gx[is[u]] = dx;
        if (full_a) t.ga[is[u]] = da;
        if (full_b) t.gb[is[u]] = db;
      }
    }
  }
}
template <typename T, int FAM>
__device__ __forceinline__ void lj_bwd_family(const LJTerm<T>& t, unsigned blk, T gc, double& rx, double& ra, double& rb) {
  if (!t.vec) lj_bwd_elem<T, FAM>(t, blk, gc, rx, ra, rb);
  else if (t.periodic) lj_bwd_vec<T, FAM, true>(t, blk, gc, rx, ra, rb);
  else lj_bwd_vec<T, FAM, false>(t, blk, gc, rx, ra, rb);
}
// one element's contribution to the gradient of operand `o` (fold role)
template <typename T, int FAM>
__device__ __forceinline__ T lj_fold_sum(const LJTerm<T>& t, int o, uint32_t j, uint32_t P, T gc) {
  typedef Fam<T, FAM> F;
  T acc = (T)0;
  for (uint32_t i = j; i < t.n; i += P) {
    T dx, da, db;
    F::partials(t.x[lj_idx(t.cx, i, t.px)], t.a[lj_idx(t.ca, i, t.pa)], F::has_b ? t.b[lj_idx(t.cb, i, t.pb)] : (T)1, gc, dx, da, db);
    acc += o == 0 ? dx : (o == 1 ? da : db);
  }
  return acc;
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
    const unsigned blk = blockIdx.x - t.block0;
    const T gc = (T)(t.coef * (double)g);
    double sx = 0.0, sa = 0.0, sb = 0.0;
    switch (t.family) {                          // uniform (ZS_LJ_ROWS terms own no workgroups here)
      case ZS_LJ_NORMAL: lj_bwd_family<T, ZS_LJ_NORMAL>(t, blk, gc, sx, sa, sb); break;
      case ZS_LJ_NORMAL_LOGSTD: lj_bwd_family<T, ZS_LJ_NORMAL_LOGSTD>(t, blk, gc, sx, sa, sb); break;
      case ZS_LJ_BERNOULLI: lj_bwd_family<T, ZS_LJ_BERNOULLI>(t, blk, gc, sx, sa, sb); break;
      default: lj_bwd_family<T, ZS_LJ_BERNOULLI_LOGITS>(t, blk, gc, sx, sa, sb); break;
    }
    const bool sc_x = t.gx && t.cx == CLS_SCALAR && t.n > 1, sc_a = t.ga && t.ca == CLS_SCALAR && t.n > 1,
               sc_b = t.gb && t.cb == CLS_SCALAR && t.n > 1;
    if (sc_x) sx = block_sum_256(sx, sh);
    if (sc_a) sa = block_sum_256(sa, sh);
    if (sc_b) sb = block_sum_256(sb, sh);
    if (threadIdx.x == 0) {                      // the hand-off: written through by the lane that takes the ticket below
      if (sc_x) store_wt(ws + 3 * blockIdx.x + 0, sx);
      if (sc_a) store_wt(ws + 3 * blockIdx.x + 1, sa);
      if (sc_b) store_wt(ws + 3 * blockIdx.x + 2, sb);
    }
  } else if (A.n_folds > 0) {
    // ---- fold role: operand of period 1 < P < n: thread j adds the contributions of elements j, j + P, j + 2P, ... in order
    int fi = 0;
    for (int j = 1; j < A.n_folds; ++j)
      if (blockIdx.x >= A.f[j].block0) fi = j;
    fi = __builtin_amdgcn_readfirstlane(fi);
    const LJFold& f = A.f[fi];
    const LJTerm<T>& t = A.t[f.term];
    const T gc = (T)(t.coef * (double)g);
    const uint32_t P = f.operand == 0 ? t.px : (f.operand == 1 ? t.pa : t.pb);
    T* __restrict__ dst = f.operand == 0 ? t.gx : (f.operand == 1 ? t.ga : t.gb);
    for (uint32_t j = (blockIdx.x - f.block0) * LJ_BLOCK + threadIdx.x; j < P; j += f.nblocks * LJ_BLOCK) {
      T v;
      switch (t.family) {
        case ZS_LJ_NORMAL: v = lj_fold_sum<T, ZS_LJ_NORMAL>(t, f.operand, j, P, gc); break;
        case ZS_LJ_NORMAL_LOGSTD: v = lj_fold_sum<T, ZS_LJ_NORMAL_LOGSTD>(t, f.operand, j, P, gc); break;
        case ZS_LJ_BERNOULLI: v = lj_fold_sum<T, ZS_LJ_BERNOULLI>(t, f.operand, j, P, gc); break;
        default: v = lj_fold_sum<T, ZS_LJ_BERNOULLI_LOGITS>(t, f.operand, j, P, gc); break;
      }
      dst[j] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    drain_stores();
    last = (ticket_take(ticket) == gridDim.x - 1);
  }
  __syncthreads();
  if (last && threadIdx.x < 64) {
    for (int ti = 0; ti < A.n_terms; ++ti) {
      const LJTerm<T>& t = A.t[ti];
      if (gcoef && threadIdx.x == 0) gcoef[ti] = (T)(t.coef * (double)g);
      if (t.family == ZS_LJ_ROWS || t.n <= 1) continue;
      for (int o = 0; o < 3; ++o) {
        T* dst = o == 0 ? t.gx : (o == 1 ? t.ga : t.gb);
        const unsigned cls = o == 0 ? t.cx : (o == 1 ? t.ca : t.cb);
        if (!dst || cls != CLS_SCALAR) continue;
        double s = 0.0;
        for (unsigned i = threadIdx.x; i < t.nblocks; i += 64) s += load_wt(ws + 3 * (t.block0 + i) + o);
        s = wave_sum_d(s);
        if (threadIdx.x == 0) dst[0] = (T)s;
      }
    }
    if (threadIdx.x == 0) ticket_return(ticket);
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
    if (s.n >= (int64_t(1) << 31)) return ZS_ENOTSUP;        // 32-bit indices: this path is for the launch-bound sizes
    const bool rows = s.family == ZS_LJ_ROWS, normal = s.family == ZS_LJ_NORMAL || s.family == ZS_LJ_NORMAL_LOGSTD;
    t.family = s.family;
    t.n = (uint32_t)s.n;
    t.coef = s.coef;
    t.x = (const T*)s.x; t.a = (const T*)s.a; t.b = (const T*)s.b;
    const int64_t px = rows ? s.n : s.px, pa = rows ? 1 : s.pa, pb = normal ? s.pb : 1;
    want[i] = 0;
    if (s.n == 0) continue;
    if (!s.x || px < 1 || s.n % px) return ZS_EINVAL;
    if (!rows && (!s.a || pa < 1 || s.n % pa)) return ZS_EINVAL;
    if (normal && (!s.b || pb < 1 || s.n % pb)) return ZS_EINVAL;
    t.px = (uint32_t)px; t.pa = (uint32_t)pa; t.pb = (uint32_t)pb;
    const auto cls = [&](int64_t P) { return (unsigned char)(P == s.n ? CLS_FULL : (P == 1 ? CLS_SCALAR : CLS_PERIODIC)); };
    t.cx = cls(px);
    t.ca = rows ? (unsigned char)CLS_NONE : cls(pa);
    t.cb = normal ? cls(pb) : (unsigned char)CLS_NONE;
    if (backward) {
      t.gx = (T*)s.gx; t.ga = (T*)s.ga; t.gb = (T*)s.gb;
      if (rows) { t.gx = t.ga = t.gb = nullptr; }
      if (!normal) {
        if (t.gx) return ZS_ENOTSUP;            // gradient w.r.t. the Bernoulli observation is not provided
        t.gb = nullptr;
      }
      if (rows || !(t.gx || t.ga || t.gb)) continue;     // nothing to compute element-wise for this term
    }
    // vector form: every present non-scalar operand (and every full-size gradient) 4-element aligned with a period that is a
    // multiple of 4, and at least one full-size operand to point the scalar operands' loads at
    bool vec = s.n >= 4;
    const T* dummy = nullptr;
    const struct { const T* p; const T* g; unsigned char c; int64_t P; } ops[3] = {
        {t.x, t.gx, t.cx, px}, {t.a, t.ga, t.ca, pa}, {t.b, t.gb, t.cb, pb}};
    for (int o = 0; o < 3; ++o) {
      if (ops[o].c == CLS_NONE || ops[o].c == CLS_SCALAR) continue;
      if ((ops[o].c == CLS_PERIODIC && (ops[o].P % 4)) || !lj_aligned<T>(ops[o].p)) vec = false;
      if (ops[o].c == CLS_FULL) {
        if (backward && ops[o].g && !lj_aligned<T>(ops[o].g)) vec = false;
        if (!dummy && lj_aligned<T>(ops[o].p)) dummy = ops[o].p;
      }
      if (ops[o].c == CLS_PERIODIC) t.periodic = 1;
    }
    if (!dummy) vec = false;
    t.vec = vec ? 1 : 0;
    t.dummy = dummy;
    // one round per thread (LJ_U groups of 4 elements, or LJ_U elements) until the grid cap makes it stride
    const int64_t items = vec ? (s.n >> 2) : s.n;
    want[i] = (items + LJ_BLOCK * LJ_U - 1) / (LJ_BLOCK * LJ_U);
    if (want[i] < 1) want[i] = 1;
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
        const unsigned cls = o == 0 ? t.cx : (o == 1 ? t.ca : t.cb);
        const int64_t P = o == 0 ? t.px : (o == 1 ? t.pa : t.pb);
        if (!dst || cls != CLS_PERIODIC) continue;
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
  int wpr;                    // forward: wavefronts per row (1, 2 or 4: a row's Philox groups in ONE round of lanes)
  int64_t wave0;              // forward: first wavefront of this term (a multiple of 4: rows never straddle workgroups)
  const T *gz, *glp;
  int64_t gsk, gsr;
  T *gmu, *gsigma;
  int ks;                     // backward: particle slices per parameter element (power of two <= 16)
  unsigned block0, nblocks;   // backward: workgroups of this term (256 / ks parameter elements each)
};
template <typename T>
struct MSArgs {
  MSTerm<T> t[ZS_MS_MAX_TERMS];
  int n_terms;
  int64_t total_waves;
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

// forward: `wpr` wavefronts per (term, particle, row); lanes take the Philox groups (4 consecutive flat elements) that touch
// the row -- for the shapes this kernel is for (rows of up to ~1000 elements) every lane has at most one group: ONE round
// of parameter loads, the draw, one round of stores.  A workgroup's four wavefronts belong to one term (its Philox call id
// is wave-uniform, the contract of philox4x32_10) and to whole rows (partial row sums meet in LDS).
template <typename T>
__global__ __launch_bounds__(256) void k_normal_sample_multi(const MSArgs<T> A, uint64_t seed, const uint64_t* __restrict__ rs,
                                                             uint64_t* __restrict__ rng_used) {
  typedef Mth<T> Mh;
  __shared__ T part[4];
  uint64_t base = 0;
  if (rs) { seed = rs[0]; base = rs[1]; }
  if (rng_used && blockIdx.x == 0 && threadIdx.x == 0) { rng_used[0] = seed; rng_used[1] = base; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t w0 = (int64_t)blockIdx.x * 4;
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (A.t[j].M > 0 && w0 >= A.t[j].wave0) ti = j;
  ti = __builtin_amdgcn_readfirstlane(ti);
  const MSTerm<T>& t = A.t[ti];
  const int wpr = t.wpr;
  const int64_t row = (w0 - t.wave0 + wv) / wpr;
  const int sub = (int)((w0 - t.wave0 + wv) - row * wpr);
  const bool active = t.M > 0 && row < t.K * t.R;
  T acc = (T)0;
  int64_t k = 0, r = 0;
  if (active) {
    divmod(row, t.R, k, r);
    const uint64_t call = base + t.offset;
    const bool ls = t.ls != 0;
    const int64_t start = k * t.M + r * t.D, end = start + t.D;
    for (int64_t g = (start >> 2) + sub * 64 + lane; g <= ((end - 1) >> 2); g += 64 * wpr) {
      const int64_t i0 = g << 2;
      float4 n4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!t.eps) n4 = philox_normal4((uint64_t)g, call, seed);
      T mu[4], sg[4], e[4];
      bool in[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {              // all loads first (clamped, unconditional)
        const int64_t i = i0 + j;
        in[j] = i >= start && i < end;
        const int64_t ic = in[j] ? i : start;
        const int64_t m = ic - k * t.M;
        mu[j] = t.mu[m];
        sg[j] = t.sigma[m];
        e[j] = t.eps ? t.eps[ic] : (T)f4_get(n4, j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const T s = Mh::sigma_of(sg[j], ls);
        const T zz = ms_mul_add_2round<T>(mu[j], s, e[j]);
        if (in[j]) t.z[i0 + j] = zz;
        T logstd, prec;
        Mh::parts(s, logstd, prec);
        if (in[j]) acc += Mh::normal_term(zz - mu[j], logstd, prec);
      }
    }
    acc = wave_sum_t<T>(acc);
  }
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (active && t.lp && sub == 0 && lane == 0) {
    T s = part[wv];
    for (int j = 1; j < wpr; ++j) s += part[wv + j];         // the row's wavefronts, in order
    t.lp[k * t.sk + r * t.sr] = s;
  }
}

// backward: thread = (parameter element, particle slice); the slices of an element meet in LDS and are added in slice order.
// (One thread per element walking all K particles was K dependent rounds of loads + K Philox regenerations in sequence.)
template <typename T>
__global__ __launch_bounds__(256) void k_normal_sample_multi_bwd(const MSArgs<T> A, uint64_t seed, const uint64_t* __restrict__ rs) {
  typedef Mth<T> Mh;
  __shared__ T red[3][256];
  uint64_t base = 0;
  if (rs) { seed = rs[0]; base = rs[1]; }
  int ti = 0;
  for (int j = 1; j < A.n_terms; ++j)
    if (A.t[j].nblocks && blockIdx.x >= A.t[j].block0) ti = j;
  ti = __builtin_amdgcn_readfirstlane(ti);
  const MSTerm<T>& t = A.t[ti];
  const int KS = t.ks, MP = 256 / KS;
  const int ks = threadIdx.x / MP, mp = threadIdx.x - ks * MP;
  const int64_t m = (int64_t)(blockIdx.x - t.block0) * MP + mp;
  const bool live = m < t.M;
  const uint64_t call = base + t.offset;
  const int64_t mc = live ? m : 0, r = mc / t.D;
  T a = (T)0, b = (T)0, g = (T)0;
  for (int64_t k = ks; k < t.K; k += KS) {
    const int64_t i = k * t.M + mc;
    if (t.gz) {
      const T gzv = t.gz[i];
      const T e = t.eps ? t.eps[i] : (T)f4_get(philox_normal4((uint64_t)(i >> 2), call, seed), (int)(i & 3));
      a += gzv;
      b += gzv * e;
    }
    if (t.glp) g += t.glp[k * t.gsk + r * t.gsr];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  red[2][threadIdx.x] = g;
  __syncthreads();
  if (ks == 0 && live) {
    for (int j = 1; j < KS; ++j) {
      a += red[0][j * MP + mp];
      b += red[1][j * MP + mp];
      g += red[2][j * MP + mp];
    }
    t.gmu[m] = a;
    const T sg = Mh::sigma_of(t.sigma[m], t.ls != 0);
    t.gsigma[m] = t.ls ? b * sg - g : b - g / sg;
  }
}

template <typename T>
int ms_build(const zs_ms_term* terms, int n_terms, bool backward, MSArgs<T>& A, unsigned& grid) {
  if (!terms || n_terms < 1) return ZS_EINVAL;
  if (n_terms > ZS_MS_MAX_TERMS) return ZS_ENOTSUP;
  memset(&A, 0, sizeof(A));
  A.n_terms = n_terms;
  int64_t waves = 0;
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
    t.wave0 = waves;
    t.block0 = blk;
    t.wpr = 1;
    t.ks = 1;
    if (s.M == 0) continue;
    if (!backward) {
      if (!s.mu || !s.sigma || !s.z) return ZS_EINVAL;
      t.wpr = s.D > 512 ? 4 : (s.D > 256 ? 2 : 1);            // (D + 3) / 4 + 1 groups on 64 * wpr lanes
      const int64_t wv = s.K * t.R * t.wpr;
      waves += (wv + 3) / 4 * 4;
    } else {
      if (!s.sigma || !s.gmu || !s.gsigma) return ZS_EINVAL;
      t.gz = (const T*)s.gz; t.glp = (const T*)s.glp; t.gsk = s.glp_stride_k; t.gsr = s.glp_stride_r;
      t.gmu = (T*)s.gmu; t.gsigma = (T*)s.gsigma;
      int ks = 1;
      while (ks < 16 && ks < s.K) ks *= 2;
      if (s.M >= 65536) ks = 1;                               // plenty of elements: one thread per element
      t.ks = ks;
      const int64_t mp = 256 / ks;
      t.nblocks = (unsigned)((s.M + mp - 1) / mp);
      blk += t.nblocks;
    }
  }
  if (waves > (int64_t(1) << 31) || blk > (1u << 30)) return ZS_ENOTSUP;
  A.total_waves = waves;
  grid = backward ? blk : (unsigned)(waves / 4);
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
// Workgroup = (tile of `bt` batch rows, particle k); bt is 64, 32 or 16 -- the smallest that still leaves >= ~256
// workgroups, so that a K = 10, B = 512 layer (80 tiles of 64 rows) spreads over the chip instead of 80 of its 256 CUs.
// Tiles of h / gout / out are CONTIGUOUS in memory: they are staged into LDS as flat copies, four elements per load when the
// tile is 16-byte aligned (one round of loads for the whole tile; a per-element (row, column) split costs an integer division
// per element and the 8-deep batches the compiler forms made four dependent rounds of it).
constexpr int PL_BT_MAX = 64;
constexpr int PL_LDS_FLOATS = 15360;   // 60 KB of fp32 (the double twin: half as many elements)

__host__ __device__ __forceinline__ int pl_odd(int v) { return v | 1; }     // odd leading dimension: conflict-free columns

// flat copy of `n` elements global -> LDS (vectorised when `vec`: src 4-element aligned; dst is)
template <typename T>
__device__ __forceinline__ void pl_stage(T* __restrict__ dst, const T* __restrict__ src, int n, bool vec) {
  if (vec) {
    const int n4 = n >> 2;
    for (int e = threadIdx.x; e < n4; e += 256) reinterpret_cast<V4<T>*>(dst)[e] = reinterpret_cast<const V4<T>*>(src)[e];
    for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) dst[e] = src[e];
  } else {
    for (int e = threadIdx.x; e < n; e += 256) dst[e] = src[e];
  }
}
// the same with the ReLU mask applied on the way: dst = (out > 0) ? gout : 0
template <typename T>
__device__ __forceinline__ void pl_stage_gpre(T* __restrict__ dst, const T* __restrict__ gout, const T* __restrict__ out, int n,
                                              bool vec, bool relu) {
  if (vec) {
    const int n4 = n >> 2;
    for (int e = threadIdx.x; e < n4; e += 256) {
      V4<T> g = reinterpret_cast<const V4<T>*>(gout)[e];
      if (relu) {
        const V4<T> o = reinterpret_cast<const V4<T>*>(out)[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) g.v[j] = o.v[j] > (T)0 ? g.v[j] : (T)0;
      }
      reinterpret_cast<V4<T>*>(dst)[e] = g;
    }
    for (int e = (n4 << 2) + threadIdx.x; e < n; e += 256) dst[e] = (!relu || out[e] > (T)0) ? gout[e] : (T)0;
  } else {
    for (int e = threadIdx.x; e < n; e += 256) dst[e] = (!relu || out[e] > (T)0) ? gout[e] : (T)0;
  }
}
template <typename T>
__host__ __device__ __forceinline__ bool pl_al(const void* p) { return (((uintptr_t)p) & (4 * sizeof(T) - 1)) == 0; }

// forward: w[k] (padded rows: lanes of a wavefront differ in the output unit o) and the h tile (flat) are staged in LDS;
// the outputs of a tile are one contiguous run of nb * n_out values: consecutive lanes take consecutive (b, o) pairs ->
// coalesced stores
template <typename T>
__global__ __launch_bounds__(256) void k_particle_linear(const T* __restrict__ h, int64_t hsk, const T* __restrict__ w,
                                                         T* __restrict__ out, int B, int n_in, int n_out, int relu, int ntiles,
                                                         int bt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T* hs = reinterpret_cast<T*>(smem_raw);                      // [bt][n_in] flat (16-byte aligned: vector stores)
  const int WS = pl_odd(n_in + 1);
  T* ws = hs + ((bt * n_in + 3) & ~3);                          // [n_out][WS]
  const int k = blockIdx.x / ntiles, tile = blockIdx.x - k * ntiles;
  const int b0 = tile * bt, nb = B - b0 < bt ? B - b0 : bt;
  const T* __restrict__ hk = h + (int64_t)k * hsk + (int64_t)b0 * n_in;
  pl_stage<T>(hs, hk, nb * n_in, pl_al<T>(hk));
  const T* __restrict__ wk = w + (int64_t)k * n_out * (n_in + 1);
  for (int e = threadIdx.x; e < n_out * (n_in + 1); e += 256) {
    const int o = e / (n_in + 1), i = e - o * (n_in + 1);
    ws[o * WS + i] = wk[e];
  }
  __syncthreads();
  const T p = Mth<T>::rsqrt_n(n_in + 1);                           // torch.sqrt(torch.as_tensor(h.shape[2])), bnn_vi.py:42
  T* __restrict__ ok = out + ((int64_t)k * B + b0) * n_out;
  for (int e = threadIdx.x; e < nb * n_out; e += 256) {
    const int b = e / n_out, o = e - b * n_out;
    const T* __restrict__ hr = hs + b * n_in;
    const T* __restrict__ wr = ws + o * WS;
    T acc = (T)0;
#pragma unroll 8
    for (int i = 0; i < n_in; ++i) acc += hr[i] * wr[i];
    acc += wr[n_in];                                                // the appended column of ones (bnn_vi.py:40)
    acc = acc / p;
    if (relu) acc = acc > (T)0 ? acc : (T)0;
    ok[e] = acc;
  }
}

// backward: workgroup = (tile, particle k), as in the forward kernel.  Each workgroup stages its tile of
// gpre = gout * (out > 0), its tile of h and (for gh) w[k] ONCE -- one round of loads -- then
//   - writes its tile of gh = gpre x w[k] / p (when wanted), and
//   - writes the tile's PARTIAL weight gradient part[k, tile, o, i] = sum_{b in tile} gpre[b, o] * [h | 1][b, i];
// the last workgroup of particle k to finish (a ticket per particle) adds the partials of all tiles in tile order and writes
// gw[k] = sum / p: deterministic, one launch, every workgroup busy for one short round (the first version gave each weight
// element one thread that walked all B rows: 10-30 workgroups of 8 dependent rounds, 62 us at B = 512).
template <typename T>
__global__ __launch_bounds__(256) void k_particle_linear_bwd(const T* __restrict__ h, int64_t hsk, const T* __restrict__ w,
                                                             const T* __restrict__ out, const T* __restrict__ gout,
                                                             T* __restrict__ gh, T* __restrict__ gw, T* __restrict__ part,
                                                             unsigned* __restrict__ tickets, int B, int n_in, int n_out, int relu,
                                                             int ntiles, int bt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ bool last;
  const T p = Mth<T>::rsqrt_n(n_in + 1);
  const int WS = n_in + 1, nW = n_out * (n_in + 1);
  T* gs = reinterpret_cast<T*>(smem_raw);                     // [bt][n_out] flat
  T* hs = gs + ((bt * n_out + 3) & ~3);                        // [bt][n_in] flat
  T* ws = hs + ((bt * n_in + 3) & ~3);                         // [n_out][n_in + 1] flat (only when gh is wanted)
  const int k = blockIdx.x / ntiles, tile = blockIdx.x - k * ntiles;
  const int b0 = tile * bt, nb = B - b0 < bt ? B - b0 : bt;
  const int64_t ob = ((int64_t)k * B + b0) * n_out;
  pl_stage_gpre<T>(gs, gout + ob, out + ob, nb * n_out, pl_al<T>(gout + ob) && pl_al<T>(out + ob), relu != 0);
  const T* __restrict__ hk = h + (int64_t)k * hsk + (int64_t)b0 * n_in;
  pl_stage<T>(hs, hk, nb * n_in, pl_al<T>(hk));
  if (gh) {
    const T* __restrict__ wk = w + (int64_t)k * nW;
    pl_stage<T>(ws, wk, nW, pl_al<T>(wk));
  }
  __syncthreads();
  if (gh) {
    T* __restrict__ ghk = gh + ((int64_t)k * B + b0) * n_in;
    for (int e = threadIdx.x; e < nb * n_in; e += 256) {
      const int b = e / n_in, i = e - b * n_in;
      const T* __restrict__ gr = gs + b * n_out;
      T acc = (T)0;
#pragma unroll 8
      for (int o = 0; o < n_out; ++o) acc += gr[o] * ws[o * WS + i];      // (8 independent LDS read pairs in flight)
      ghk[e] = acc / p;
    }
  }
  T* __restrict__ pk = part + ((int64_t)k * ntiles + tile) * nW;
  for (int e = threadIdx.x; e < nW; e += 256) {
    const int o = e / (n_in + 1), i = e - o * (n_in + 1);
    T acc = (T)0;
    if (i < n_in) {
#pragma unroll 8
      for (int b = 0; b < nb; ++b) acc += gs[b * n_out + o] * hs[b * n_in + i];
    } else {
#pragma unroll 8
      for (int b = 0; b < nb; ++b) acc += gs[b * n_out + o];
    }
    store_wt(pk + e, acc);                       // written through: no release fence below (see the hand-off note at the top)
  }
  drain_stores();                                // every storing wave waits for its own stores ...
  __syncthreads();                               // ... before the lane that signals for all of them takes the ticket
  if (threadIdx.x == 0) {
    last = (ticket_take(tickets + k) == (unsigned)ntiles - 1u);
    if (last) {
      // the last arrival reads ntiles partials per weight element: ONE agent acquire by this lane (invalidates this CU's L1),
      // waited for, then PLAIN loads behind the barrier -- sc1 (atomic) loads are issued one after the other: 8 tiles = 8
      // dependent round trips, 7 us of the first version's 18 at B = 512 and 50 us at B = 4096
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      drain_stores();
    }
  }
  __syncthreads();
  if (last) {
    // four consecutive weight elements per thread and load (the partials of a tile are nW contiguous values, 16-byte aligned
    // when nW % 4 == 0), 16 tiles in flight: ntiles = 64 (B = 4096) is 4 rounds instead of 3 x 8 rounds of single floats
    const T* __restrict__ p0 = part + (int64_t)k * ntiles * nW;
    T* __restrict__ gk = gw + (int64_t)k * nW;
    const bool v4 = (nW & 3) == 0 && pl_al<T>(p0) && pl_al<T>(gk);
    const int nv = v4 ? (nW >> 2) : 0;
    for (int e = threadIdx.x; e < nv; e += 256) {
      V4<T> s = {{(T)0, (T)0, (T)0, (T)0}};
#pragma unroll 16
      for (int t = 0; t < ntiles; ++t) {                                    // tile order: deterministic
        const V4<T> q = reinterpret_cast<const V4<T>*>(p0 + (int64_t)t * nW)[e];
        s.v[0] += q.v[0]; s.v[1] += q.v[1]; s.v[2] += q.v[2]; s.v[3] += q.v[3];
      }
      s.v[0] /= p; s.v[1] /= p; s.v[2] /= p; s.v[3] /= p;
      reinterpret_cast<V4<T>*>(gk)[e] = s;
    }
    for (int e = (nv << 2) + threadIdx.x; e < nW; e += 256) {
      T s = (T)0;
#pragma unroll 16
      for (int t = 0; t < ntiles; ++t) s += p0[(int64_t)t * nW + e];
      gk[e] = s / p;
    }
    if (threadIdx.x == 0) ticket_return(tickets + k);
  }
}

// rows per tile: the largest of 64 / 32 / 16 that still leaves at least `want` workgroups (never below 16).  Forward: 256
// (every CU busy: K = 10, B = 512 -> 320 tiles of 16 rows, 4.4 us against 8.0 with 80 tiles of 64).  Backward: 128 -- every
// tile costs a hand-off (write-through partials, a ticket, a share of the last arrival's reduction): measured at K = 10,
// B = 512, 13 -> 50: 16 rows 12.6 us, 32 rows 10.4, 64 rows 11.6; B = 4096: 68.6 / 37.2 / 25.2
inline int pl_tile_rows(int64_t K, int64_t B, int64_t want) {
  int bt = PL_BT_MAX;
  while (bt > 16 && K * ((B + bt - 1) / bt) < want) bt >>= 1;
  return bt;
}
template <typename T>
bool pl_fits(int64_t n_in, int64_t n_out) {
  if (n_in < 1 || n_out < 1 || n_in > 255 || n_out > 256) return false;
  const int64_t lim = PL_LDS_FLOATS * (int64_t)sizeof(float) / (int64_t)sizeof(T);
  const int64_t fwd = PL_BT_MAX * n_in + 4 + n_out * pl_odd((int)n_in + 1);
  const int64_t bwd = PL_BT_MAX * n_out + PL_BT_MAX * n_in + 8 + n_out * (n_in + 1);
  return fwd <= lim && bwd <= lim;
}

template <typename T>
int particle_linear(const T* h, int64_t hsk, const T* w, T* out, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu,
                    void* stream) {
  if (K < 0 || B < 0 || n_in < 1 || n_out < 1 || (hsk != 0 && hsk != B * n_in)) return ZS_EINVAL;
  if (!pl_fits<T>(n_in, n_out) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0 || B == 0) return 0;
  if (!h || !w || !out) return ZS_EINVAL;
  const int bt = pl_tile_rows(K, B, 256);
  const int ntiles = (int)((B + bt - 1) / bt);
  if ((int64_t)ntiles * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  const size_t smem = sizeof(T) * (size_t)(((bt * n_in + 3) & ~3) + n_out * pl_odd((int)n_in + 1));
  ZS_LAUNCH_SMEM(KID_PARTICLE_LINEAR, (k_particle_linear<T>), dim3((unsigned)(ntiles * K)), dim3(256), smem, (hipStream_t)stream, h,
                 hsk, w, out, (int)B, (int)n_in, (int)n_out, relu, ntiles, bt);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int particle_linear_bwd(const T* h, int64_t hsk, const T* w, const T* out, const T* gout, T* gh, T* gw, int64_t K, int64_t B,
                        int64_t n_in, int64_t n_out, int relu, T* workspace, int64_t workspace_len, uint32_t* tickets,
                        void* stream) {
  if (K < 0 || B < 0 || n_in < 1 || n_out < 1 || (hsk != 0 && hsk != B * n_in)) return ZS_EINVAL;
  if (!pl_fits<T>(n_in, n_out) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0) return 0;
  if (!gw) return ZS_EINVAL;
  const int64_t nW = n_out * (n_in + 1);
  if (B == 0) {                                   // no rows: the weight gradient is zero
    const hipError_t e = hipMemsetAsync(gw, 0, sizeof(T) * (size_t)(K * nW), (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
  }
  if (!h || !w || !gout || (relu && !out)) return ZS_EINVAL;
  static const int bt_env = env_knob("ZS_PL_BWD_BT", 0);          // experiments only (zs_common.h)
  const int bt = (bt_env == 16 || bt_env == 32 || bt_env == 64) ? bt_env : pl_tile_rows(K, B, 128);
  const int ntiles = (int)((B + bt - 1) / bt);
  if ((int64_t)ntiles * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  if (!workspace || !tickets || workspace_len < K * ntiles * nW) return ZS_EINVAL;
  const size_t smem = sizeof(T) * (size_t)(((bt * n_out + 3) & ~3) + ((bt * n_in + 3) & ~3) + nW);
  ZS_LAUNCH_SMEM(KID_PARTICLE_LINEAR_BWD, (k_particle_linear_bwd<T>), dim3((unsigned)(ntiles * K)), dim3(256), smem,
                 (hipStream_t)stream, h, hsk, w, out ? out : gout, gout, gh, gw, workspace, (unsigned*)tickets, (int)B, (int)n_in,
                 (int)n_out, relu, ntiles, bt);
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
                                          float* workspace, int64_t workspace_len, uint32_t* tickets, void* stream) {
  return particle_linear_bwd<float>(h, h_stride_k, w, out, gout, gh, gw, K, B, n_in, n_out, relu, workspace, workspace_len, tickets,
                                    stream);
}
extern "C" int zs_particle_linear_bwd_f64(const double* h, int64_t h_stride_k, const double* w, const double* out,
                                          const double* gout, double* gh, double* gw, int64_t K, int64_t B, int64_t n_in,
                                          int64_t n_out, int relu, double* workspace, int64_t workspace_len, uint32_t* tickets,
                                          void* stream) {
  return particle_linear_bwd<double>(h, h_stride_k, w, out, gout, gh, gw, K, B, n_in, n_out, relu, workspace, workspace_len, tickets,
                                     stream);
}

// ================================================================ CS1: column sums of a row-major matrix
// out[c] = sum_r x[r, c] -- the bias gradient of a dense layer (grad_bias = grad_output.sum(0)), the one reduction of the
// callers' nn.Linear stack that is not a GEMM: torch's generic reduce kernel takes 12.4 us for the [12 800, 500] gradients of
// the IWAE step (7 of them per step: 10 % of the step) where the bytes need 3-4.  Lanes run along the columns (16 bytes
// each when the row length allows: a wavefront reads 1 KB of one row), the four wavefronts of a workgroup and the
// workgroups of a column tile split the rows; the row-chunk partials meet in LDS, then in a workspace whose last arrival
// (a ticket per column tile; fence-free hand-off, see the top of this file) adds them in chunk order: deterministic.
namespace {
constexpr int CS_MAX_CHUNKS = 128;
#ifndef CS_U
#define CS_U 8                   // rows a wavefront has in flight (16: no faster, measured)
#endif

// ACT != 0 (AB1, below): x is the gradient w.r.t. a dense layer's ACTIVATED output y; the kernel forms the gradient w.r.t.
// the pre-activation on the way (ReLU: g * [y > 0]; sigmoid: g * y * (1 - y)), writes it to `gpre` (which may be x itself)
// and sums THAT by columns: the activation's backward pass and the bias gradient in the one pass over the gradient.
template <typename T, int ACT>
__device__ __forceinline__ T act_bwd(T g, T y) {
  if (ACT == ZS_ACT_RELU) return y > (T)0 ? g : (T)0;
  if (ACT == ZS_ACT_SIGMOID) return g * ((T)1 - y) * y;          // (torch's sigmoid_backward: grad * (1 - y) * y)
  return g;
}

template <typename T, int V, int ACT = 0>      // V = 4: dwordx4 lanes (cols % 4 == 0, aligned), V = 1: one column per lane
__global__ __launch_bounds__(256) void k_column_sum(const T* x, T* __restrict__ out, T* __restrict__ part,
                                                    unsigned* __restrict__ tickets, int64_t rows, int cols, int nchunks,
                                                    int64_t rows_per_chunk, const T* __restrict__ y = nullptr, T* gpre = nullptr) {
  __shared__ T red[4][64 * V];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ctile = blockIdx.x / nchunks, chunk = blockIdx.x - ctile * nchunks;
  const int ncol_tile = 64 * V;
  const int c0 = ctile * ncol_tile + lane * V;
  const bool on = c0 < cols;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
  T acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = (T)0;
  if (on) {
    const T* p = x + c0;
    // a wavefront takes rows r0 + wv, r0 + wv + 4, ...: CS_U of them in flight (8 KB per wavefront with 16-byte lanes),
    // clamped to the chunk's last row so that the loads are unconditional; a clamped row's contribution is dropped
    for (int64_t r = r0 + wv; r < r1; r += 4 * CS_U) {
      T v[CS_U][V], a[ACT ? CS_U : 1][V];
      bool live[CS_U];
#pragma unroll
      for (int u = 0; u < CS_U; ++u) {
        const int64_t ru = r + 4 * u;
        live[u] = ru < r1;
        const int64_t rc = live[u] ? ru : r;
        if (V == 4) {
          const V4<T> q = *reinterpret_cast<const V4<T>*>(p + rc * cols);
#pragma unroll
          for (int j = 0; j < V; ++j) v[u][j] = q.v[j];
          if (ACT) {
            const V4<T> qa = *reinterpret_cast<const V4<T>*>(y + c0 + rc * cols);
#pragma unroll
            for (int j = 0; j < V; ++j) a[u][j] = qa.v[j];
          }
        } else {
          v[u][0] = p[rc * cols];
          if (ACT) a[u][0] = y[c0 + rc * cols];
        }
      }
      if (ACT) {
#pragma unroll
        for (int u = 0; u < CS_U; ++u) {
#pragma unroll
          for (int j = 0; j < V; ++j) v[u][j] = act_bwd<T, ACT>(v[u][j], a[u][j]);
          if (live[u]) {
            T* o = gpre + c0 + (r + 4 * u) * cols;
            if (V == 4) {
              V4<T> q;
#pragma unroll
              for (int j = 0; j < V; ++j) q.v[j] = v[u][j];
              *reinterpret_cast<V4<T>*>(o) = q;
            } else {
              o[0] = v[u][0];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < CS_U; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += live[u] ? v[u][j] : (T)0;
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[wv][lane * V + j] = acc[j];
  __syncthreads();
  T* __restrict__ pk = part + ((int64_t)ctile * nchunks + chunk) * ncol_tile;
  if (threadIdx.x < ncol_tile) {                       // (V = 4: all 256 threads; V = 1: the first wavefront)
    const int c = threadIdx.x;
    const T s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (nchunks == 1) {
      if (ctile * ncol_tile + c < cols) out[ctile * ncol_tile + c] = s;
    } else {
      store_wt(pk + c, s);
    }
  }
  if (nchunks == 1) return;
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    last = ticket_take(tickets + ctile) == (unsigned)nchunks - 1u;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      drain_stores();
    }
  }
  __syncthreads();
  if (last) {
    // wavefront wv adds the partials of chunks wv, wv + 4, ... (V columns per lane, 16 loads in flight); the four slices meet
    // in LDS and are added in slice order: a fixed order of additions whatever the arrival order was
    const T* __restrict__ p0 = part + (int64_t)ctile * nchunks * ncol_tile + lane * V;
    T s[V];
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = (T)0;
#pragma unroll 16
    for (int t = wv; t < nchunks; t += 4) {
      if (V == 4) {
        const V4<T> q = *reinterpret_cast<const V4<T>*>(p0 + (int64_t)t * ncol_tile);
#pragma unroll
        for (int j = 0; j < V; ++j) s[j] += q.v[j];
      } else {
        s[0] += p0[(int64_t)t * ncol_tile];
      }
    }
    __syncthreads();                                   // (uniform: `last` is a workgroup-wide flag)
#pragma unroll
    for (int j = 0; j < V; ++j) red[wv][lane * V + j] = s[j];
    __syncthreads();
    if (threadIdx.x < ncol_tile) {
      const int c = threadIdx.x;
      if (ctile * ncol_tile + c < cols) out[ctile * ncol_tile + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
    if (threadIdx.x == 0) ticket_return(tickets + ctile);
  }
}

template <typename T>
int column_sum(const T* x, T* out, int64_t rows, int64_t cols, T* workspace, int64_t workspace_len, uint32_t* tickets,
               int64_t n_tickets, void* stream, int act = ZS_ACT_NONE, const T* y = nullptr, T* gpre = nullptr) {
  if (rows < 0 || cols < 0) return ZS_EINVAL;
  if (act != ZS_ACT_NONE && act != ZS_ACT_RELU && act != ZS_ACT_SIGMOID) return ZS_EINVAL;
  if (cols == 0) return 0;
  if (!out) return ZS_EINVAL;
  if (cols > (1 << 24)) return ZS_ENOTSUP;
  if (rows == 0) {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(T) * (size_t)cols, (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
  }
  if (!x) return ZS_EINVAL;
  if (act != ZS_ACT_NONE && (!y || !gpre)) return ZS_EINVAL;
  const bool v4 = (cols % 4) == 0 && pl_al<T>(x) && (act == ZS_ACT_NONE || (pl_al<T>(y) && pl_al<T>(gpre)));
  const int ncol_tile = v4 ? 256 : 64;
  const int64_t ctiles = (cols + ncol_tile - 1) / ncol_tile;
  // row chunks: enough workgroups to fill the chip (~512), at least 32 rows each, at most CS_MAX_CHUNKS per column tile
  int64_t nchunks = (512 + ctiles - 1) / ctiles;
  if (nchunks > CS_MAX_CHUNKS) nchunks = CS_MAX_CHUNKS;
  if (nchunks > (rows + 31) / 32) nchunks = (rows + 31) / 32;
  if (nchunks < 1) nchunks = 1;
  int64_t rpc = (rows + nchunks - 1) / nchunks;
  rpc = (rpc + 3) / 4 * 4;                               // whole groups of four rows (one per wavefront)
  nchunks = (rows + rpc - 1) / rpc;
  if (nchunks > 1 && (!workspace || !tickets || workspace_len < ctiles * nchunks * ncol_tile || n_tickets < ctiles)) return ZS_EINVAL;
  if (ctiles * nchunks > (int64_t(1) << 30)) return ZS_ENOTSUP;
  const dim3 grid((unsigned)(ctiles * nchunks));
#define ZS_CS_LAUNCH(KID, VV, AA)                                                                                            \
  ZS_LAUNCH(KID, (k_column_sum<T, VV, AA>), grid, dim3(256), (hipStream_t)stream, x, out, workspace, (unsigned*)tickets, rows, \
            (int)cols, (int)nchunks, rpc, y, gpre)
  if (act == ZS_ACT_RELU) {
    if (v4) ZS_CS_LAUNCH(KID_DENSE_ACT_BWD, 4, ZS_ACT_RELU); else ZS_CS_LAUNCH(KID_DENSE_ACT_BWD, 1, ZS_ACT_RELU);
  } else if (act == ZS_ACT_SIGMOID) {
    if (v4) ZS_CS_LAUNCH(KID_DENSE_ACT_BWD, 4, ZS_ACT_SIGMOID); else ZS_CS_LAUNCH(KID_DENSE_ACT_BWD, 1, ZS_ACT_SIGMOID);
  } else {
    if (v4) ZS_CS_LAUNCH(KID_COLUMN_SUM, 4, 0); else ZS_CS_LAUNCH(KID_COLUMN_SUM, 1, 0);
  }
#undef ZS_CS_LAUNCH
  ZS_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int zs_column_sum_f32(const float* x, float* out, int64_t rows, int64_t cols, float* workspace, int64_t workspace_len,
                                 uint32_t* tickets, int64_t n_tickets, void* stream) {
  return column_sum<float>(x, out, rows, cols, workspace, workspace_len, tickets, n_tickets, stream);
}
extern "C" int zs_column_sum_f64(const double* x, double* out, int64_t rows, int64_t cols, double* workspace, int64_t workspace_len,
                                 uint32_t* tickets, int64_t n_tickets, void* stream) {
  return column_sum<double>(x, out, rows, cols, workspace, workspace_len, tickets, n_tickets, stream);
}

// ================================================================ AB1: activation backward + bias gradient of a dense layer
// gpre = g * act'(y), out[c] = sum_r gpre[r, c] in one pass (k_column_sum<., ., ACT>): what the backward of
// `act(linear(x))` needs before its two GEMMs.  torch runs threshold_backward / sigmoid_backward (read g, y; write gpre) and then
// a reduction that reads gpre again.
extern "C" int zs_dense_act_bwd_f32(const float* g, const float* y, int act, float* gpre, float* gbias, int64_t rows, int64_t cols,
                                    float* workspace, int64_t workspace_len, uint32_t* tickets, int64_t n_tickets, void* stream) {
  if (act == ZS_ACT_NONE) return ZS_EINVAL;
  return column_sum<float>(g, gbias, rows, cols, workspace, workspace_len, tickets, n_tickets, stream, act, y, gpre);
}
extern "C" int zs_dense_act_bwd_f64(const double* g, const double* y, int act, double* gpre, double* gbias, int64_t rows,
                                    int64_t cols, double* workspace, int64_t workspace_len, uint32_t* tickets, int64_t n_tickets,
                                    void* stream) {
  if (act == ZS_ACT_NONE) return ZS_EINVAL;
  return column_sum<double>(g, gbias, rows, cols, workspace, workspace_len, tickets, n_tickets, stream, act, y, gpre);
}

// ================================================================ PR1: RMSE of the particle-mean prediction
// out = sqrt(mean_b (y[b] - mean_k pred[k, b])^2) -- the diagnostic the BNN caller evaluates in every forward pass
// (examples/bayesian_neural_nets/bnn_vi.py:84-87: mean over particles, sub, pow, mean, sqrt: five launches at a size where a
// launch is the cost).  Lanes run along b (coalesced rows of pred), K loads per lane in flight; squared errors are summed in
// double.  Up to 4096 datapoints one workgroup does it all; beyond, workgroups of 1024 datapoints hand their partial sums
// to the last arrival (fence-free hand-off, top of this file), added in workgroup order: deterministic.
namespace {
constexpr int PR_BLOCK = 1024;
constexpr int PR_ONE_BLOCK_MAX = 4096;
constexpr int PR_MAX_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(PR_BLOCK) void k_particle_rmse(const T* __restrict__ pred, const T* __restrict__ y, T* __restrict__ out,
                                                            int64_t K, int64_t B, double* __restrict__ ws, unsigned* __restrict__ ticket) {
  __shared__ double sh[PR_BLOCK / 64];
  __shared__ bool last;
  const T invK = (T)1 / (T)K;
  double acc = 0.0;
  for (int64_t b = (int64_t)blockIdx.x * PR_BLOCK + threadIdx.x; b < B; b += (int64_t)gridDim.x * PR_BLOCK) {
    T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
    int64_t k = 0;
    for (; k + 4 <= K; k += 4) {                      // four rows in flight; one fixed order of additions
      const T v0 = pred[k * B + b], v1 = pred[(k + 1) * B + b], v2 = pred[(k + 2) * B + b], v3 = pred[(k + 3) * B + b];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; k < K; ++k) s0 += pred[k * B + b];
    const T d = y[b] - ((s0 + s1) + (s2 + s3)) * invK;
    acc += (double)d * (double)d;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < PR_BLOCK / 64; ++w) tot += sh[w];
  }
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) out[0] = (T)sqrt(tot / (double)B);
    return;
  }
  if (threadIdx.x == 0) {
    store_wt(ws + blockIdx.x, tot);
    drain_stores();
    last = ticket_take(ticket) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  double s = 0.0;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += PR_BLOCK) s += load_wt(ws + i);    // (<= PR_MAX_BLOCKS: one load per thread)
  s = wave_sum_d(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < PR_BLOCK / 64; ++w) t2 += sh[w];
    out[0] = (T)sqrt(t2 / (double)B);
    ticket_return(ticket);
  }
}

template <typename T>
int particle_rmse(const T* pred, const T* y, T* out, int64_t K, int64_t B, double* workspace, int64_t workspace_len, uint32_t* ticket,
                  void* stream) {
  if (K < 0 || B < 0 || !out) return ZS_EINVAL;
  // (K == 0 or B == 0: torch's mean of nothing is NaN; the kernel's 0 * (1 / 0) and 0 / 0 produce it)
  if (B > 0 && (!y || (K > 0 && !pred))) return ZS_EINVAL;
  int64_t nb = B <= PR_ONE_BLOCK_MAX ? 1 : (B + PR_BLOCK - 1) / PR_BLOCK;
  if (nb > PR_MAX_BLOCKS) nb = PR_MAX_BLOCKS;
  if (nb > 1 && (!workspace || !ticket || workspace_len < nb)) return ZS_EINVAL;
  ZS_LAUNCH(KID_PARTICLE_RMSE, (k_particle_rmse<T>), dim3((unsigned)nb), dim3(PR_BLOCK), (hipStream_t)stream, pred, y, out, K, B, workspace,
            (unsigned*)ticket);
  ZS_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int zs_particle_rmse_f32(const float* pred, const float* y, float* out, int64_t K, int64_t B, double* workspace,
                                    int64_t workspace_len, uint32_t* ticket, void* stream) {
  return particle_rmse<float>(pred, y, out, K, B, workspace, workspace_len, ticket, stream);
}
extern "C" int zs_particle_rmse_f64(const double* pred, const double* y, double* out, int64_t K, int64_t B, double* workspace,
                                    int64_t workspace_len, uint32_t* ticket, void* stream) {
  return particle_rmse<double>(pred, y, out, K, B, workspace, workspace_len, ticket, stream);
}
