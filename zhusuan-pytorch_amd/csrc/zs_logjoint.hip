// One-launch log-joint and multi-node sampler for the launch-bound configurations (include/zs_hip.h: LJ1, MS1).
//
// At the VAE (B = 512) and BNN (B = 512, K = 10) shapes every kernel of the step occupies ~4 us of the stream whatever it
// computes, so what counts is the NUMBER of launches.  The reference walks the nodes of a BayesianNet in Python loops
// (ELBO.log_joint, zhusuan/variational/elbo.py:58-79; the re-read of every latent, elbo.py:122); round 2 had one launch
// per node and direction.  Here:
//   LJ1  all log-probs of all nodes + their weighted sum = the scalar objective: one launch forward, one backward
//        (pointer table in the kernel arguments, as the Adam update does for 32 tensors);
//   MS1  the fused sample + log-density (K1) of several Normal nodes: one launch forward, one backward.
// (The callers' layers that got the same treatment -- PL1, CS1, AB1, PR1 -- live in zs_layers.hip.)
// Both are HBM-trivial (a few hundred KB): they are written for few dependent rounds of loads and for deterministic
// sums (fixed combination order), not for bandwidth.  Templated on float / double.
#include "zs_onelaunch.h"

namespace {

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

// Backward only: a Bernoulli term whose OBSERVATION wants a gradient (bernoulli.py:94 is differentiable in `sample`) runs as a
// family of its own -- two more logarithms per element that the ordinary term (observed data) must not pay for.
constexpr int LJ_BERNOULLI_GX = ZS_LJ_BERNOULLI_LOGITS + 1, LJ_BERNOULLI_LOGITS_GX = ZS_LJ_BERNOULLI_LOGITS + 2;

// ---- per-family element arithmetic
template <typename T, int FAM>
struct Fam {
  static constexpr bool has_a = FAM != ZS_LJ_ROWS;
  static constexpr bool has_b = FAM == ZS_LJ_NORMAL || FAM == ZS_LJ_NORMAL_LOGSTD;
  static constexpr bool bern_gx = FAM == LJ_BERNOULLI_GX || FAM == LJ_BERNOULLI_LOGITS_GX;
  static constexpr bool bern_logits = FAM == ZS_LJ_BERNOULLI_LOGITS || FAM == LJ_BERNOULLI_LOGITS_GX;
  static constexpr bool has_gx = has_b || bern_gx;          // families whose first operand can receive a gradient
  static __device__ __forceinline__ T term(T x, T a, T b) {
    typedef Mth<T> M;
    if (FAM == ZS_LJ_ROWS) return x;
    if (has_b) {
      const T sg = M::sigma_of(b, FAM == ZS_LJ_NORMAL_LOGSTD);
      T logstd, prec;
      M::parts(sg, logstd, prec);
      return M::normal_term(x - a, logstd, prec);
    }
    return M::bern_term(bern_logits ? M::sigmoid(a) : a, x);
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
      db = (T)0;
      const T p = bern_logits ? M::sigmoid(a) : a;
      da = gc * M::bern_dp(p, x) * (bern_logits ? p * ((T)1 - p) : (T)1);
      dx = bern_gx ? gc * log_ratio_any(p) : (T)0;          // d/dx = log(p + 1e-8) - log(1 - p + 1e-8)
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
  const bool full_x = F::has_gx && t.gx && t.cx == CLS_FULL, full_a = t.ga && t.ca == CLS_FULL,
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
  const bool full_x = F::has_gx && t.gx && t.cx == CLS_FULL, full_a = t.ga && t.ca == CLS_FULL,
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
      case LJ_BERNOULLI_GX: lj_bwd_family<T, LJ_BERNOULLI_GX>(t, blk, gc, sx, sa, sb); break;
      case LJ_BERNOULLI_LOGITS_GX: lj_bwd_family<T, LJ_BERNOULLI_LOGITS_GX>(t, blk, gc, sx, sa, sb); break;
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
        case LJ_BERNOULLI_GX: v = lj_fold_sum<T, LJ_BERNOULLI_GX>(t, f.operand, j, P, gc); break;
        case LJ_BERNOULLI_LOGITS_GX: v = lj_fold_sum<T, LJ_BERNOULLI_LOGITS_GX>(t, f.operand, j, P, gc); break;
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
        t.gb = nullptr;
        if (t.gx) t.family = s.family == ZS_LJ_BERNOULLI ? LJ_BERNOULLI_GX : LJ_BERNOULLI_LOGITS_GX;     // (see Fam)
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
  const T *gz, *gz2, *glp;
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
    if (t.gz || t.gz2) {
      // two gradient tensors for one sample (the draw used by the model AND by its own prior / log-joint term): added here
      // instead of by an autograd accumulation launch in front of this kernel
      const T gzv = (t.gz ? t.gz[i] : (T)0) + (t.gz2 ? t.gz2[i] : (T)0);
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
      t.gz = (const T*)s.gz; t.gz2 = (const T*)s.gz2; t.glp = (const T*)s.glp; t.gsk = s.glp_stride_k; t.gsr = s.glp_stride_r;
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
