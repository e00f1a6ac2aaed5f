// The callers' layers of the launch-bound configurations as one launch each (include/zs_hip.h): caller-side glue, not part of
// the distribution / objective path.
//   PL1  the BNN caller's particle-batched dense layer (bias column, 1/sqrt(n), ReLU fused): one launch each way;
//   CS1  column sums = the bias gradient of a dense layer;  AB1  the same pass with the activation's backward folded in;
//   PR1  the BNN caller's RMSE diagnostic.
// Written for few dependent rounds of loads and deterministic sums (fixed combination order); templated on float / double.
#include "zs_onelaunch.h"

namespace {

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

// the last arrival's reduction: gk[e] = (sum over the tiles t, in tile order, of p0[t * tile_stride + e]) / p for e < nW.
// Four consecutive weight elements per thread and load when everything is 16-byte aligned, 16 tiles in flight: ntiles = 64
// (B = 4096) is 4 rounds instead of 3 x 8 rounds of single floats.  Deterministic.
template <typename T>
__device__ __forceinline__ void pl_reduce_tiles(const T* __restrict__ p0, int64_t tile_stride, T* __restrict__ gk, int nW, int ntiles,
                                                T p) {
  const bool v4 = (tile_stride & 3) == 0 && pl_al<T>(p0) && pl_al<T>(gk);
  const int nv = v4 ? (nW >> 2) : 0;
  for (int e = threadIdx.x; e < nv; e += 256) {
    V4<T> s = {{(T)0, (T)0, (T)0, (T)0}};
#pragma unroll 16
    for (int t = 0; t < ntiles; ++t) {
      const V4<T> q = reinterpret_cast<const V4<T>*>(p0 + (int64_t)t * tile_stride)[e];
      s.v[0] += q.v[0]; s.v[1] += q.v[1]; s.v[2] += q.v[2]; s.v[3] += q.v[3];
    }
    s.v[0] /= p; s.v[1] /= p; s.v[2] /= p; s.v[3] /= p;
    reinterpret_cast<V4<T>*>(gk)[e] = s;
  }
  for (int e = (nv << 2) + threadIdx.x; e < nW; e += 256) {
    T s = (T)0;
#pragma unroll 16
    for (int t = 0; t < ntiles; ++t) s += p0[(int64_t)t * tile_stride + e];
    gk[e] = s / p;
  }
}

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
  const T p = Mth<T>::sqrt_n(n_in + 1);                           // torch.sqrt(torch.as_tensor(h.shape[2])), bnn_vi.py:42
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
  const T p = Mth<T>::sqrt_n(n_in + 1);
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
    store_wt(pk + e, acc);                       // written through: no release fence below (hand-off note in zs_onelaunch.h)
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
    pl_reduce_tiles<T>(part + (int64_t)k * ntiles * nW, nW, gw + (int64_t)k * nW, nW, ntiles, p);
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

// ================================================================ PM1: the whole particle-batched network in one launch each way
// The BNN caller's network is a loop over PL1 layers (bnn_vi.py:27-48); at its shapes ([13, 50, 1]: 751 weights per particle)
// every layer is a launch and, backward, a cross-workgroup hand-off of its own.  PM1 walks all layers inside ONE workgroup per
// (tile of batch rows, particle): forward, the weights of every layer and the input tile are staged in LDS in one round of
// loads and the activations ping-pong between two LDS buffers (each layer's output also goes to global memory: backward needs
// it); backward, the tile's activations, the incoming gradient and the weights are staged once, the gradient walks the layers
// in reverse inside LDS, every layer's PARTIAL weight gradient goes to the tile's slab of the workspace, and ONE hand-off (a
// ticket per particle) lets the last tile of a particle add up the slabs of all layers in tile order.  ReLU after every layer
// but the last, as the caller has it.  Same arithmetic, in the same order, as a chain of PL1 launches (bit-identical outputs).
constexpr int PM_MAX_LAYERS = ZS_PM_MAX_LAYERS;
template <typename T>
struct PMArgs {
  const T* w[PM_MAX_LAYERS];
  T* out[PM_MAX_LAYERS];
  T* gw[PM_MAX_LAYERS];
  int n[PM_MAX_LAYERS + 1];      // widths: n[0] inputs, n[l + 1] outputs of layer l
  int woff[PM_MAX_LAYERS];       // LDS offset of layer l's staged weights (forward: rows padded to an odd length; backward: flat)
  int ioff[PM_MAX_LAYERS];       // backward: LDS offset of the tile of layer l's INPUT (x for l = 0, the activation of layer l - 1)
  int poff[PM_MAX_LAYERS];       // offset of layer l's partial weight gradient inside a tile's workspace slab (multiples of 4)
  int L, maxw, slab;
};
__host__ __device__ __forceinline__ int pm_pad4(int v) { return (v + 3) & ~3; }

template <typename T>
__global__ __launch_bounds__(256) void k_particle_mlp(const PMArgs<T> A, const T* __restrict__ x, int64_t xsk, int B, int ntiles,
                                                      int bt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int AS = pm_pad4(bt * A.maxw);
  T* cur = reinterpret_cast<T*>(smem_raw);         // [bt][width] flat: the input of the layer at hand
  T* nxt = cur + AS;
  T* wsm = nxt + AS;
  const int k = blockIdx.x / ntiles, tile = blockIdx.x - k * ntiles;
  const int b0 = tile * bt, nb = B - b0 < bt ? B - b0 : bt;
  const T* __restrict__ xk = x + (int64_t)k * xsk + (int64_t)b0 * A.n[0];
  pl_stage<T>(cur, xk, nb * A.n[0], pl_al<T>(xk));
#pragma unroll
  for (int l = 0; l < PM_MAX_LAYERS; ++l) {
    if (l >= A.L) break;
    const int n_in = A.n[l], n_out = A.n[l + 1], WS = pl_odd(n_in + 1);
    const T* __restrict__ wk = A.w[l] + (int64_t)k * n_out * (n_in + 1);
    T* __restrict__ wl = wsm + A.woff[l];
    for (int e = threadIdx.x; e < n_out * (n_in + 1); e += 256) {
      const int o = e / (n_in + 1), i = e - o * (n_in + 1);
      wl[o * WS + i] = wk[e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < PM_MAX_LAYERS; ++l) {
    if (l >= A.L) break;
    const int n_in = A.n[l], n_out = A.n[l + 1], WS = pl_odd(n_in + 1);
    const bool relu = l < A.L - 1;
    const T p = Mth<T>::sqrt_n(n_in + 1);
    const T* __restrict__ wl = wsm + A.woff[l];
    T* __restrict__ ok = A.out[l] + ((int64_t)k * B + b0) * n_out;
    for (int e = threadIdx.x; e < nb * n_out; e += 256) {
      const int b = e / n_out, o = e - b * n_out;
      const T* __restrict__ hr = cur + b * n_in;
      const T* __restrict__ wr = wl + o * WS;
      T acc = (T)0;
#pragma unroll 8
      for (int i = 0; i < n_in; ++i) acc += hr[i] * wr[i];
      acc += wr[n_in];
      acc = acc / p;
      if (relu) acc = acc > (T)0 ? acc : (T)0;
      ok[e] = acc;
      nxt[e] = acc;
    }
    __syncthreads();
    T* t = cur; cur = nxt; nxt = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_particle_mlp_bwd(const PMArgs<T> A, const T* __restrict__ x, int64_t xsk,
                                                          const T* __restrict__ gout, T* __restrict__ gx, T* __restrict__ part,
                                                          unsigned* __restrict__ tickets, int B, int ntiles, int bt) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  __shared__ bool last;
  const int AS = pm_pad4(bt * A.maxw);
  T* cur = reinterpret_cast<T*>(smem_raw);         // gradient w.r.t. the OUTPUT of the layer at hand (ReLU mask applied), [bt][n_out]
  T* nxt = cur + AS;
  T* sm = nxt + AS;                                 // inputs of all layers, then the weights
  const int k = blockIdx.x / ntiles, tile = blockIdx.x - k * ntiles;
  const int b0 = tile * bt, nb = B - b0 < bt ? B - b0 : bt;
  const int L = A.L;
  {                                                 // one round of loads: gradient tile, every layer's input tile, the weights
    const int nL = A.n[L];
    const T* __restrict__ gk = gout + ((int64_t)k * B + b0) * nL;
    pl_stage<T>(cur, gk, nb * nL, pl_al<T>(gk));
    const T* __restrict__ xk = x + (int64_t)k * xsk + (int64_t)b0 * A.n[0];
    pl_stage<T>(sm + A.ioff[0], xk, nb * A.n[0], pl_al<T>(xk));
#pragma unroll
    for (int l = 1; l < PM_MAX_LAYERS; ++l) {
      if (l >= L) break;
      const T* __restrict__ ak = A.out[l - 1] + ((int64_t)k * B + b0) * A.n[l];
      pl_stage<T>(sm + A.ioff[l], ak, nb * A.n[l], pl_al<T>(ak));
    }
#pragma unroll
    for (int l = 0; l < PM_MAX_LAYERS; ++l) {
      if (l >= L) break;
      if (l == 0 && !gx) continue;                  // the first layer's weights are only needed for the input gradient
      const int nW = A.n[l + 1] * (A.n[l] + 1);
      const T* __restrict__ wk = A.w[l] + (int64_t)k * nW;
      pl_stage<T>(sm + A.woff[l], wk, nW, pl_al<T>(wk));
    }
  }
  __syncthreads();
  T* __restrict__ slab = part + ((int64_t)k * ntiles + tile) * A.slab;
#pragma unroll
  for (int ll = 0; ll < PM_MAX_LAYERS; ++ll) {
    const int l = L - 1 - ll;
    if (l < 0) break;
    const int n_in = A.n[l], n_out = A.n[l + 1], nW = n_out * (n_in + 1), WS = n_in + 1;
    const T p = Mth<T>::sqrt_n(n_in + 1);
    const T* __restrict__ in = sm + A.ioff[l];
    T* __restrict__ pk = slab + A.poff[l];
    for (int e = threadIdx.x; e < nW; e += 256) {   // the tile's partial weight gradient (unscaled: / p in the reduction)
      const int o = e / (n_in + 1), i = e - o * (n_in + 1);
      T acc = (T)0;
      if (i < n_in) {
#pragma unroll 8
        for (int b = 0; b < nb; ++b) acc += cur[b * n_out + o] * in[b * n_in + i];
      } else {
#pragma unroll 8
        for (int b = 0; b < nb; ++b) acc += cur[b * n_out + o];
      }
      store_wt(pk + e, acc);
    }
    if (l > 0 || gx) {                              // gradient w.r.t. the layer's input; below a hidden layer: its ReLU mask
      const T* __restrict__ wl = sm + A.woff[l];
      T* __restrict__ gxk = gx ? gx + ((int64_t)k * B + b0) * n_in : nullptr;
      for (int e = threadIdx.x; e < nb * n_in; e += 256) {
        const int b = e / n_in, i = e - b * n_in;
        const T* __restrict__ gr = cur + b * n_out;
        T acc = (T)0;
#pragma unroll 8
        for (int o = 0; o < n_out; ++o) acc += gr[o] * wl[o * WS + i];
        acc = acc / p;
        if (l > 0) nxt[e] = in[e] > (T)0 ? acc : (T)0;
        else gxk[e] = acc;
      }
    }
    __syncthreads();
    T* t = cur; cur = nxt; nxt = t;
  }
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    last = (ticket_take(tickets + k) == (unsigned)ntiles - 1u);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      drain_stores();
    }
  }
  __syncthreads();
  if (last) {
    const T* __restrict__ p0 = part + (int64_t)k * ntiles * A.slab;
#pragma unroll
    for (int l = 0; l < PM_MAX_LAYERS; ++l) {
      if (l >= L) break;
      const int nW = A.n[l + 1] * (A.n[l] + 1);
      pl_reduce_tiles<T>(p0 + A.poff[l], A.slab, A.gw[l] + (int64_t)k * nW, nW, ntiles, Mth<T>::sqrt_n(A.n[l] + 1));
    }
    if (threadIdx.x == 0) ticket_return(tickets + k);
  }
}

// admission: every width within PL1's limits and both directions' LDS layouts (at the largest tile) within 60 KB
template <typename T>
bool pm_build(const zs_pm_layer* layers, int L, bool backward, bool want_gx, PMArgs<T>& A, int64_t& lds_fwd, int64_t& lds_bwd) {
  if (L < 1 || L > PM_MAX_LAYERS || !layers) return false;
  memset(&A, 0, sizeof(A));
  A.L = L;
  int maxw = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t n_in = layers[l].n_in, n_out = layers[l].n_out;
    if (n_in < 1 || n_out < 1 || n_in > 255 || n_out > 256) return false;
    if (l > 0 && n_in != layers[l - 1].n_out) return false;
    A.n[l] = (int)n_in;
    A.n[l + 1] = (int)n_out;
    if (n_in > maxw) maxw = (int)n_in;
    if (n_out > maxw) maxw = (int)n_out;
    A.w[l] = static_cast<const T*>(layers[l].w);
    A.out[l] = static_cast<T*>(layers[l].out);
    A.gw[l] = static_cast<T*>(layers[l].gw);
  }
  A.maxw = maxw;
  const int bt = PL_BT_MAX;
  int64_t wf = 0, wb = 0, in = 0, slab = 0;
  for (int l = 0; l < L; ++l) {
    const int nW = A.n[l + 1] * (A.n[l] + 1);
    if (!backward) { A.woff[l] = (int)wf; }
    wf += pm_pad4(A.n[l + 1] * pl_odd(A.n[l] + 1));
    A.ioff[l] = (int)in;
    in += pm_pad4(bt * A.n[l]);
    A.poff[l] = (int)slab;
    slab += pm_pad4(nW);
  }
  if (backward) {
    for (int l = 0; l < L; ++l) {
      A.woff[l] = (int)(in + wb);
      if (l > 0 || want_gx) wb += pm_pad4(A.n[l + 1] * (A.n[l] + 1));
    }
  }
  A.slab = (int)slab;
  int64_t wall = 0;
  for (int l = 0; l < L; ++l) wall += pm_pad4(A.n[l + 1] * (A.n[l] + 1));
  lds_fwd = 2 * (int64_t)pm_pad4(bt * maxw) + wf;
  lds_bwd = 2 * (int64_t)pm_pad4(bt * maxw) + in + wall;        // (admission counts the first layer's weights: one rule either way)
  const int64_t lim = PL_LDS_FLOATS * (int64_t)sizeof(float) / (int64_t)sizeof(T);
  return lds_fwd <= lim && lds_bwd <= lim;
}

template <typename T>
int particle_mlp(const T* x, int64_t xsk, const zs_pm_layer* layers, int L, int64_t K, int64_t B, void* stream) {
  if (K < 0 || B < 0 || !layers || L < 1) return ZS_EINVAL;
  if (L > PM_MAX_LAYERS) return ZS_ENOTSUP;
  for (int l = 0; l < L; ++l)
    if (layers[l].n_in < 1 || layers[l].n_out < 1 || (l > 0 && layers[l].n_in != layers[l - 1].n_out)) return ZS_EINVAL;
  if (xsk != 0 && xsk != B * layers[0].n_in) return ZS_EINVAL;
  PMArgs<T> A;
  int64_t lf, lb;
  if (!pm_build<T>(layers, L, false, false, A, lf, lb) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0 || B == 0) return 0;
  if (!x) return ZS_EINVAL;
  for (int l = 0; l < L; ++l)
    if (!A.w[l] || !A.out[l]) return ZS_EINVAL;
  const int bt = pl_tile_rows(K, B, 256);
  const int ntiles = (int)((B + bt - 1) / bt);
  if ((int64_t)ntiles * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  int64_t wf = 0;
  for (int l = 0; l < L; ++l) wf += pm_pad4(A.n[l + 1] * pl_odd(A.n[l] + 1));
  const size_t smem = sizeof(T) * (size_t)(2 * pm_pad4(bt * A.maxw) + wf);
  ZS_LAUNCH_SMEM(KID_PARTICLE_MLP, (k_particle_mlp<T>), dim3((unsigned)(ntiles * K)), dim3(256), smem, (hipStream_t)stream, A, x, xsk,
                 (int)B, ntiles, bt);
  ZS_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int particle_mlp_bwd(const T* x, int64_t xsk, const zs_pm_layer* layers, int L, const T* gout, T* gx, int64_t K, int64_t B,
                     T* workspace, int64_t workspace_len, uint32_t* tickets, void* stream) {
  if (K < 0 || B < 0 || !layers || L < 1) return ZS_EINVAL;
  if (L > PM_MAX_LAYERS) return ZS_ENOTSUP;
  for (int l = 0; l < L; ++l)
    if (layers[l].n_in < 1 || layers[l].n_out < 1 || (l > 0 && layers[l].n_in != layers[l - 1].n_out)) return ZS_EINVAL;
  if (xsk != 0 && xsk != B * layers[0].n_in) return ZS_EINVAL;
  PMArgs<T> A;
  int64_t lf, lb;
  if (!pm_build<T>(layers, L, true, gx != nullptr, A, lf, lb) || B > (1 << 24) || K > (1 << 20)) return ZS_ENOTSUP;
  if (K == 0) return 0;
  for (int l = 0; l < L; ++l)
    if (!A.gw[l]) return ZS_EINVAL;
  if (B == 0) {                                   // no rows: the weight gradients are zero
    for (int l = 0; l < L; ++l) {
      const hipError_t e = hipMemsetAsync(A.gw[l], 0, sizeof(T) * (size_t)(K * A.n[l + 1] * (A.n[l] + 1)), (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
    }
    return 0;
  }
  if (!x || !gout) return ZS_EINVAL;
  for (int l = 0; l < L; ++l)
    if (!A.w[l] || (l < L - 1 && !A.out[l])) return ZS_EINVAL;
  const int bt = pl_tile_rows(K, B, 128);
  const int ntiles = (int)((B + bt - 1) / bt);
  if ((int64_t)ntiles * K > (int64_t(1) << 30)) return ZS_ENOTSUP;
  if (!workspace || !tickets || workspace_len < K * ntiles * (int64_t)A.slab) return ZS_EINVAL;
  // LDS offsets were laid out for the largest tile (input tiles of PL_BT_MAX rows): valid for any bt <= PL_BT_MAX
  int64_t in = 0, wb = 0;
  for (int l = 0; l < L; ++l) {
    in += pm_pad4(PL_BT_MAX * A.n[l]);
    if (l > 0 || gx) wb += pm_pad4(A.n[l + 1] * (A.n[l] + 1));
  }
  const size_t smem = sizeof(T) * (size_t)(2 * pm_pad4(bt * A.maxw) + in + wb);
  ZS_LAUNCH_SMEM(KID_PARTICLE_MLP_BWD, (k_particle_mlp_bwd<T>), dim3((unsigned)(ntiles * K)), dim3(256), smem, (hipStream_t)stream, A,
                 x, xsk, gout, gx, workspace, (unsigned*)tickets, (int)B, ntiles, bt);
  ZS_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int zs_particle_mlp_f32(const float* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, int64_t K, int64_t B,
                                   void* stream) {
  return particle_mlp<float>(x, x_stride_k, layers, n_layers, K, B, stream);
}
extern "C" int zs_particle_mlp_f64(const double* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, int64_t K, int64_t B,
                                   void* stream) {
  return particle_mlp<double>(x, x_stride_k, layers, n_layers, K, B, stream);
}
extern "C" int zs_particle_mlp_bwd_f32(const float* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, const float* gout,
                                       float* gx, int64_t K, int64_t B, float* workspace, int64_t workspace_len, uint32_t* tickets,
                                       void* stream) {
  return particle_mlp_bwd<float>(x, x_stride_k, layers, n_layers, gout, gx, K, B, workspace, workspace_len, tickets, stream);
}
extern "C" int zs_particle_mlp_bwd_f64(const double* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, const double* gout,
                                       double* gx, int64_t K, int64_t B, double* workspace, int64_t workspace_len, uint32_t* tickets,
                                       void* stream) {
  return particle_mlp_bwd<double>(x, x_stride_k, layers, n_layers, gout, gx, K, B, workspace, workspace_len, tickets, stream);
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
// (a ticket per column tile; fence-free hand-off, zs_onelaunch.h) adds them in chunk order: deterministic.
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
  // (experiments only, zs_common.h.  More, smaller chunks do not pay: the last arrival's reduction grows with them -- CS1 at
  // [12 800, 500]: 128 / 256 / 512 chunks per column tile = 8.1 / 14.4 / 19.2 us, AB1 17.9 / 23.0 / 27.5 us)
  static const int wgs_env = env_knob("ZS_CS_WGS", 0), chunks_env = env_knob("ZS_CS_CHUNKS", 0), minrows_env = env_knob("ZS_CS_MINROWS", 0);
  const int64_t want_wgs = wgs_env > 0 ? wgs_env : 512, max_chunks = chunks_env > 0 ? chunks_env : CS_MAX_CHUNKS;
  const int64_t min_rows = minrows_env > 0 ? minrows_env : 32;
  int64_t nchunks = (want_wgs + ctiles - 1) / ctiles;
  if (nchunks > max_chunks) nchunks = max_chunks;
  if (nchunks > (rows + min_rows - 1) / min_rows) nchunks = (rows + min_rows - 1) / min_rows;
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
// to the last arrival (fence-free hand-off, zs_onelaunch.h), added in workgroup order: deterministic.
namespace {
constexpr int PR_BLOCK = 1024;
constexpr int PR_ONE_BLOCK_MAX = 4096;
constexpr int PR_MAX_BLOCKS = 1024;

template <typename T>
__global__ __launch_bounds__(PR_BLOCK) void k_particle_rmse(const T* __restrict__ pred, const T* __restrict__ y, T* __restrict__ out,
                                                            int64_t K, int64_t B, double* __restrict__ ws, unsigned* __restrict__ ticket) {
  __shared__ double sh[PR_BLOCK / 64];
  __shared__ bool last;
  const T Kf = (T)K;
  double acc = 0.0;
  for (int64_t b = (int64_t)blockIdx.x * PR_BLOCK + threadIdx.x; b < B; b += (int64_t)gridDim.x * PR_BLOCK) {
    T s0 = (T)0, s1 = (T)0, s2 = (T)0, s3 = (T)0;
    int64_t k = 0;
    for (; k + 4 <= K; k += 4) {                      // four rows in flight; one fixed order of additions
      const T v0 = pred[k * B + b], v1 = pred[(k + 1) * B + b], v2 = pred[(k + 2) * B + b], v3 = pred[(k + 3) * B + b];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; k < K; ++k) s0 += pred[k * B + b];
    const T d = y[b] - ((s0 + s1) + (s2 + s3)) / Kf;         // torch.mean: the sum divided by K
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
  // (K == 0 or B == 0: torch's mean of nothing is NaN; the kernel's 0 / 0 produces it)
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
