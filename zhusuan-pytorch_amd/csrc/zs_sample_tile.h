// Fused sample + log-density kernel of the location-scale families (Normal: K1, Logistic: L1) and its launch geometry,
// shared by zs_normal.hip and zs_locscale.hip.  See the comment on k_sample_tile.
#pragma once
#include <stdlib.h>

#include "zs_common.h"
#include "zs_locscale_math.h"

namespace zs {

enum { D_NORMAL = 0, D_LOGISTIC = 1, D_UNIFORM = 2 };

// round-to-nearest mul / add that the compiler may not contract into an FMA: the sample
// z = mean + std * eps must round twice like the reference's separate mul and add
// (normal.py:105) so that z is bit-identical for identical eps.
__device__ __forceinline__ float mul_add_2round(float m, float s, float e) {
#pragma clang fp contract(off)   // HIP's __fmul_rn/__fadd_rn are plain * and + and would still fuse
  const float prod = s * e;
  return m + prod;
}

// `sigma` operand given as log(sigma) (Normal(logstd=...), normal.py:56: std = exp(logstd)): the kernels form sigma
// themselves instead of the caller launching an exp (and a multiply in backward).  Precise expf: it runs once per
// parameter, not per particle.  d/d logstd = sigma * d/d sigma.
__device__ __forceinline__ float sigma_of(float v, bool is_logstd) { return is_logstd ? expf(v) : v; }
__device__ __forceinline__ float4 sigma_of(float4 v, bool is_logstd) {
  if (is_logstd) { v.x = expf(v.x); v.y = expf(v.y); v.z = expf(v.z); v.w = expf(v.w); }
  return v;
}
// resolved Philox ids of a draw, written once per launch for the backward call (which may run after the caller has
// advanced the live rng_state)
__device__ __forceinline__ void publish_rng(uint64_t* __restrict__ rng_used, uint64_t seed, uint64_t call) {
  if (rng_used && blockIdx.x == 0 && threadIdx.x == 0) { rng_used[0] = seed; rng_used[1] = call; }
}

// ------------------------------------------------------------------------------------
// Fused sample + log-density, flat-plane tiling (the kernel the benchmark shapes run; DIST = D_NORMAL: K1, D_LOGISTIC: L1,
// location-scale families that differ only in how a standard draw and its density are formed): a workgroup of NW waves owns TB = 64*NW
// CONSECUTIVE float4 groups of the [R, D4] parameter plane, with NW chosen on the host so that TB is a multiple of D4
// (whole rows; D = 40 -> D4 = 10 -> 5 waves = 320 lanes = 32 rows).  Against the row-per-lane-group mapping above this
// keeps all 64 lanes of every wave busy (10-lane groups fill only 60 of 64: the kernel is VALU-bound on the
// generator, so idle lanes are lost throughput) and makes every z store of a wave one contiguous 1 KB segment.
// The particle loop runs on scalar counters and a uniform base pointer per particle (lane offset in a 32-bit VGPR), so
// per particle and lane the VALU work is the generator, the sample, the density and one 64-bit counter increment.
// Row sums: each lane parks its partial per particle in LDS ([KB][TB+1], odd leading dimension: conflict-free both
// ways); after KB particles the workgroup reads them back transposed, adds the D4 partials of a row and writes log q
// coalesced along the particle axis of the K-fastest result.
// Kp > 0: TWO INDEPENDENT DRAWS of Kp particles each in one launch (K = 2 Kp): particles [Kp, 2 Kp) are the draw with Philox call id
// call + 1 and counters relative to their own first particle -- each half is bit for bit what a launch of its own would have
// written (the objectives draw every latent twice, stochastic_tensor.py:115-127 + elbo.py:122; at the config sizes a launch costs
// 4.7 us and the second half of this one 0.25).  A batch of particles never straddles Kp.
// In-kernel Philox only: with eps handed in (the parity path) the kernel is memory-bound and the row-per-lane-group
// kernel above, which needs no workgroup barrier, is the faster one (69 % against 65 % of the roofline at 4.2 M rows).
// ------------------------------------------------------------------------------------
template <int DIST, bool HAS_LP, bool NT>
__global__ __launch_bounds__(1024) void k_sample_tile(
    const float4* __restrict__ mu, const float4* __restrict__ sigma,
    uint64_t seed, uint64_t call, const uint64_t* __restrict__ rs, float4* __restrict__ z, float* __restrict__ lp,
    uint32_t K, int64_t R, uint32_t D4, uint32_t M4, uint32_t kchunk, uint32_t KB, uint32_t n_ptiles, uint32_t total,
    int64_t sk, int64_t sr, bool ls, uint64_t* __restrict__ rng_used, float4* __restrict__ z2, bool raw_draw, uint32_t Kp) {
  extern __shared__ float zs_k1_stage[];
  if (rs) { seed = rs[0]; call += rs[1]; }
  publish_rng(rng_used, seed, call);
  const uint32_t TB = blockDim.x, LDW = TB + 1, tid = threadIdx.x;
  const uint32_t rows_in_tile = TB / D4;
  for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {          // uniform: scalar registers throughout
    const uint32_t kt = t / n_ptiles, pt = t - kt * n_ptiles;
    const uint32_t m4 = pt * TB + tid;
    const bool on = m4 < M4;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s = make_float4(1.f, 1.f, 1.f, 1.f);
    if (on) {
      m = mu[m4];
      s = sigma_of(sigma[m4], ls);
    }
    if (DIST == D_UNIFORM) { s.x -= m.x; s.y -= m.y; s.z -= m.z; s.w -= m.w; }   // (low, high) -> (low, width)
    // per-lane constants, reused for every particle of the chunk:
    // Normal:   rowc = sum_j (c - log sigma_j)  (normal.py:121-124),  hp_j = 0.5 * exp(-2 log sigma_j)
    // Logistic: rowc = -sum_j log scale_j       (logistic.py:81-82)
    float rowc = 0.f, hp[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (DIST == D_UNIFORM) continue;                 // U1 has no density output
        const float l2 = log2_fast(sv[j]);
        if (DIST == D_NORMAL) {
          rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2;
          hp[j] = 0.5f * exp2_fast(-2.0f * l2);
        } else {
          rowc -= l2 * ZS_LN2;
        }
      }
    }
    const uint32_t k0 = kt * kchunk;
    const uint32_t k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    const int64_t rbase = (int64_t)pt * rows_in_tile;
    uint32_t kb = 0;
    for (uint32_t kb0 = k0; kb0 < k1; kb0 += kb) {
      kb = (k1 - kb0 < KB) ? (k1 - kb0) : KB;
      if (Kp && kb0 < Kp && kb0 + kb > Kp) kb = Kp - kb0;              // (two draws: a batch stays inside one of them)
      const bool second = Kp && kb0 >= Kp;
      const uint64_t callx = second ? call + 1 : call;
      const PhiloxCall pc = philox_call(callx, seed);                 // the uniform part of the generator, once per batch (zs_common.h)
      if (on) {
        const uint64_t base = (uint64_t)kb0 * M4;                      // uniform
        // uniform base pointer per particle (scalar registers) + this lane's 32-bit byte offset: the stores need no
        // per-lane 64-bit address arithmetic (M4 < 2^28 is guaranteed by the host)
        char* __restrict__ zk = reinterpret_cast<char*>(z + base);
        char* __restrict__ zk2 = reinterpret_cast<char*>(z2 + base);   // second output (Uniform: the cached draw)
        const uint32_t lane_off = m4 * 16u;
        const uint64_t step = (uint64_t)M4 * 16u;
        uint64_t g = (second ? base - (uint64_t)Kp * M4 : base) + m4;  // Philox group of (particle kb0 of its draw, this lane)
        float* __restrict__ stp = zs_k1_stage + tid;
        // one particle: sample, store, density partial.  `e` = the standard draw, `dens` = what the draw itself contributes
        // to the log-density (Logistic: sum_j log u_j + log(1 - u_j); unused for Normal)
        auto particle = [&](const float4& e, float dens) {
          float4 zz;
          zz.x = mul_add_2round(m.x, s.x, e.x);
          zz.y = mul_add_2round(m.y, s.y, e.y);
          zz.z = mul_add_2round(m.z, s.z, e.z);
          zz.w = mul_add_2round(m.w, s.w, e.w);
          {
            // scalar base + 32-bit lane offset addressing, spelled out: the compiler keeps a 64-bit pointer per lane
            // (one more VALU instruction per particle) for the C form of this store.
            // The s_nop is the wait a store of more than 8 bytes needs before a VALU instruction may overwrite its
            // data registers (the compiler's hazard recogniser does not look inside inline assembly; without it the
            // next particle's arithmetic clobbers the sample on its way out).
            const zs_f4v v = {zz.x, zz.y, zz.z, zz.w};
            if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(zk) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(zk) : "memory");
          }
          if (HAS_LP) {
            if (DIST == D_NORMAL) {
              const float d0 = zz.x - m.x, d1 = zz.y - m.y, d2 = zz.z - m.z, d3 = zz.w - m.w;
              // two packed chains (v_pk_mul_f32 + v_pk_fma_f32) and one add instead of four products and three adds: the loop is
              // bound by instruction issue
              const float t0 = fmaf(hp[2], d2 * d2, hp[0] * (d0 * d0)), t1 = fmaf(hp[3], d3 * d3, hp[1] * (d1 * d1));
              *stp = rowc - (t0 + t1);
            } else {
              *stp = rowc + dens;
            }
            stp += LDW;
          }
          zk += step;
        };
        // Uniform (uniform.py:63-70): cached value c = u (reparameterised) or low + u * width, sample = low + c * width
        auto particle_uniform = [&](const float4& u) {
          float4 c = u;
          if (!raw_draw) {
            c.x = mul_add_2round(m.x, u.x, s.x);
            c.y = mul_add_2round(m.y, u.y, s.y);
            c.z = mul_add_2round(m.z, u.z, s.z);
            c.w = mul_add_2round(m.w, u.w, s.w);
          }
          const zs_f4v o = {mul_add_2round(m.x, c.x, s.x), mul_add_2round(m.y, c.y, s.y), mul_add_2round(m.z, c.z, s.z),
                            mul_add_2round(m.w, c.w, s.w)};
          const zs_f4v cv = {c.x, c.y, c.z, c.w};
          if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(o), "s"(zk) : "memory");
          else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane_off), "v"(o), "s"(zk) : "memory");
          if (z2) {
            if (NT) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(cv), "s"(zk2) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane_off), "v"(cv), "s"(zk2) : "memory");
          }
          zk += step;
          zk2 += step;
        };
        for (uint32_t kk = 0; kk < kb; ++kk) {
          if (DIST == D_NORMAL) {
            particle(philox_normal4(g, pc), 0.f);
          } else if (DIST == D_UNIFORM) {
            const Philox4 r = philox4x32_10(g, pc);
            particle_uniform(make_float4(u01(r.x), u01(r.y), u01(r.z), u01(r.w)));
          } else {
            // Logistic draw (logistic.py:64-66): eps = log u - log(1 - u); its own log-density -eps - 2 softplus(-eps)
            // is log u + log(1 - u): the two logarithms serve both
            const Philox4 r = philox4x32_10(g, pc);
            float4 e;
            float d0, d1, d2, d3;
            logistic_draw(u01(r.x), e.x, d0);
            logistic_draw(u01(r.y), e.y, d1);
            logistic_draw(u01(r.z), e.z, d2);
            logistic_draw(u01(r.w), e.w, d3);
            particle(e, (d0 + d1) + (d2 + d3));
          }
          g += M4;
        }
      }
      if (HAS_LP) {
        __syncthreads();
        const uint32_t nout = rows_in_tile * kb;
        for (uint32_t o = tid; o < nout; o += TB) {
          uint32_t q, kk;
          if (kb == 16u) { q = o >> 4; kk = o & 15u; }       // the full batches: no integer division
          else { q = o / kb; kk = o - q * kb; }
          const float* __restrict__ src = zs_k1_stage + kk * LDW + q * D4;
          float sum = 0.f;
          for (uint32_t j = 0; j < D4; ++j) sum += src[j];
          if (rbase + q < R) lp[(int64_t)(kb0 + kk) * sk + (rbase + q) * sr] = sum;
        }
        __syncthreads();
      }
    }
  }
}

// Launch geometry of k_sample_tile.
struct K1Tile {
  bool ok;
  unsigned threads, grid, smem;
  uint32_t kchunk, KB, n_ptiles, total;
};
inline int64_t gcd64(int64_t a, int64_t b) { while (b) { const int64_t t = a % b; a = b; b = t; } return a; }
inline K1Tile k1_tile(int64_t K, int64_t R, int D4, bool want_lp) {   // (K1Tile: named after its first user)
  K1Tile g = {};
  // experiments: ZS_K1_TILE=0 disables the kernel, ZS_K1_KB / ZS_K1_ITEMS override the heuristics below
  static const int enable = env_knob("ZS_K1_TILE", 1), kb_env = env_knob("ZS_K1_KB", 0), items_env = env_knob("ZS_K1_ITEMS", 0);
  if (!enable || D4 < 1 || D4 > 64) return g;
  const int64_t M4 = R * (int64_t)D4;
  if (M4 >= (1ll << 28) || K >= (1ll << 31)) return g;                 // lane offsets are 32-bit byte offsets
  // workgroup = whole rows and whole waves: the smallest multiple of lcm(64, D4) lanes that is at least 256
  const int64_t l = 64 / gcd64(64, D4) * D4;
  int64_t TB = l * ((256 + l - 1) / l);
  if (TB > 1024) TB = l;
  if (TB > 1024) return g;                                             // e.g. D4 = 25: lcm = 1600 lanes
  const int64_t n_ptiles = (M4 + TB - 1) / TB;
  uint32_t KB = (uint32_t)(kb_env > 0 ? kb_env : 16);     // 16: the flush kernel's shift fast path; 20.5 KB of LDS at 320 lanes
  // every item pays the parameter loads and eight logarithms / exponentials once, so chunks of particles should be long, while
  // the hardware dispatcher needs spare items to even out the CUs (measured at the 1 M-row sweep point: 1 / 2 / 3 / 8 items per
  // slot = 37.5 / 35.8 / 35.2 / 42.8 us; at 131 k rows, where 3 per slot means one particle per item: 8.9 / 9.5 / 12.4 us)
  const int64_t resident = 256ll * (2048 / TB > 8 ? 8 : 2048 / TB);
  int64_t k_tiles, kchunk;
  if (items_env > 0 || !want_lp) {
    // fixed rule, about two items per slot: the experiments' override, and the store-only Uniform sampler (no density, hardly
    // any prologue: the scored split below cost it 7-12 points at 0.25 M and 1 M rows)
    const int64_t want = resident * (items_env > 0 ? items_env : 2);
    k_tiles = (want + n_ptiles - 1) / n_ptiles;
    if (k_tiles < 1) k_tiles = 1;
    if (k_tiles > K) k_tiles = K;
    kchunk = (K + k_tiles - 1) / k_tiles;
    if (kchunk < 3) kchunk = K < 3 ? K : 3;
  } else {
    // Split of the particle axis: the hardware hands items to workgroup slots as they free up, so a launch takes about
    // ceil(items / slots) item-times, and every item pays a prologue worth ~0.7 particles.  Pick the split with the best
    // (fill of the last round) x (particles per item / (particles per item + 0.7)); an item of one or two particles is
    // mostly prologue.  (Against "two items per slot", one process, Normal / Logistic: 0.25 M rows 38.7 -> 41.9 / 38 -> 40.5 %,
    // 0.5 M rows 49.2 -> 51.4 / 47.6 -> 50 %, 1 M rows 59.8 -> 61-62.5 %, 2 M rows and beyond unchanged.)
    double best = -1.0;
    k_tiles = 1;
    kchunk = K;
    for (int64_t kt = 1; kt <= K && kt <= 256; ++kt) {
      int64_t kc = (K + kt - 1) / kt;
      if (kc < 3) kc = K < 3 ? K : 3;
      const int64_t kt2 = (K + kc - 1) / kc;
      const int64_t items = n_ptiles * kt2;
      const int64_t rounds = (items + resident - 1) / resident;
      const double fill = (double)items / (double)(rounds * resident);
      const double score = fill * (double)kc / ((double)kc + 0.7);
      if (score > best + 1e-9) { best = score; k_tiles = kt2; kchunk = kc; }
      if (kc <= 3) break;
    }
  }
  k_tiles = (K + kchunk - 1) / kchunk;
  const int64_t total = n_ptiles * k_tiles;
  if (total >= (1ll << 31)) return g;
  if (kchunk < KB) KB = (uint32_t)kchunk;
  g.ok = true;
  g.threads = (unsigned)TB;
  g.smem = want_lp ? (unsigned)(KB * (TB + 1) * sizeof(float)) : 0u;
  g.kchunk = (uint32_t)kchunk;
  g.KB = KB;
  g.n_ptiles = (uint32_t)n_ptiles;
  g.total = (uint32_t)total;
  // one work item per workgroup: the hardware dispatcher hands out items as slots free up, which balances the CUs to
  // within one item (a fixed grid striding over the items leaves workgroups with floor/ceil(items / grid) of them --
  // 2 vs 3 at the 1 M-row sweep point: 71 % efficiency).  The stride loop in the kernel only serves grids beyond the cap.
  static const int grid_env = env_knob("ZS_K1_GRID", 0);
  const int64_t cap = grid_env > 0 ? grid_env : (1ll << 20);
  g.grid = (unsigned)(total < cap ? total : cap);
  return g;
}

// ------------------------------------------------------------------------------------
// Log-density of GIVEN values, parameters [R, D] repeated over the K particles (IWAE prior / q of a given sample,
// normal.py:112-116; logistic.py:81-82) and D4 <= 64: a wave owns `rpw` parameter rows, forms the per-lane constants
// once (Normal: log sigma and 0.5 sigma^-2; Logistic: 1/scale and log scale) and streams the K value rows past them,
// two rows in flight; no per-element index arithmetic.  DIST = D_NORMAL: K2, D_LOGISTIC: L2.
// ------------------------------------------------------------------------------------
struct RowMap {
  int G;      // lanes per row
  int rpw;    // rows per wave pass
  int p2;     // next pow2 >= G
};
inline RowMap row_map(int64_t D4) {
  RowMap m;
  m.G = D4 >= 64 ? 64 : (int)D4;
  m.rpw = 64 / m.G;
  m.p2 = next_pow2(m.G);
  return m;
}

// sum over the four elements of a lane of the value-dependent part of the (negated) log-density; c[j] per-lane constants
// (Normal: 0.5 sigma^-2; Logistic: 1/scale; Uniform: the upper bound, `m` the lower)
template <int DIST>
__device__ __forceinline__ float krep_terms(const float4 xv, const float4 m, const float c[4]) {
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ms[4] = {m.x, m.y, m.z, m.w};
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float d = xs[j] - ms[j];
    if (DIST == D_NORMAL) {
      acc += c[j] * (d * d);                             // 0.5 (x - mu)^2 / sigma^2
    } else if (DIST == D_LOGISTIC) {
      acc += logistic_neg_lp_term(d, c[j]);
    } else {
      acc += uniform_inside(xs[j], ms[j], c[j]) ? 0.f : INFINITY;
    }
  }
  return acc;
}

// value rows read with the non-temporal hint (the once-read stream of a problem beyond the Infinity Cache)
template <bool NTL>
__device__ __forceinline__ float4 ld_value4(const float4* __restrict__ q) {
  if (NTL) {
    const zs_f4v v = __builtin_nontemporal_load(reinterpret_cast<const zs_f4v*>(q));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *q;
}

// U value rows in flight per wave (2: the config sizes; 4: problems beyond the cache, where a wave's 2 x 960 bytes in flight
// left the CU with 60 KB outstanding), NTL: non-temporal loads of the value stream.
template <int DIST, int U, bool NTL>
__global__ __launch_bounds__(256) void k_logprob_krep(
    const float4* __restrict__ x, const float4* __restrict__ mu, const float4* __restrict__ sigma,
    float* __restrict__ lp, int64_t K, int64_t R, int D4, int G, int rpw, int p2, int64_t kchunk,
    int64_t sk, int64_t sr, bool ls) {
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
    const int64_t r = rt * rpw + rw;
    const bool on = lane_on && r < R;
    // every lane loads unconditionally (no exec-mask change between two loads); an idle lane re-reads the address of the wave's
    // LAST active row (same cache lines as its neighbours: one request -- clamping to element 0 made every wave of the launch
    // hit one line, 72 -> 64 % at 1 M rows), and its result is dropped
    const int64_t rc = r < R ? (lane_on ? r : rt * rpw + (rpw - 1 < R - 1 - rt * rpw ? rpw - 1 : R - 1 - rt * rpw)) : R - 1;
    const int64_t m4 = rc * D4 + (lane_on ? lig : 0);
    const float4 m = mu[m4];
    const float4 s = sigma_of(sigma[m4], ls);
    float rowc = 0.f, c[4];
    {
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (DIST == D_UNIFORM) {
          const float mv = j == 0 ? m.x : (j == 1 ? m.y : (j == 2 ? m.z : m.w));
          rowc -= logf(sv[j] - mv);                      // -log(high - low), once per item: precise
          c[j] = sv[j];
          continue;
        }
        const float l2 = log2_fast(sv[j]);
        if (DIST == D_NORMAL) {
          rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2;
          c[j] = 0.5f * exp2_fast(-2.0f * l2);
        } else {
          rowc -= l2 * ZS_LN2;
          c[j] = 1.0f / sv[j];
        }
      }
    }
    const int64_t k0 = kt * kchunk;
    const int64_t k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;
    float* __restrict__ lpp = lp + (k0 * sk + rc * sr);
    int64_t k = k0;
    for (; k + U - 1 < k1; k += U, g += U * M4, lpp += U * sk) {   // U value rows in flight
      float4 xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xv[u] = ld_value4<NTL>(x + g + u * M4);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float acc = rowc - krep_terms<DIST>(xv[u], m, c);
        acc = group_sum_down(acc, lig, G, p2);
        if (on && lig == 0) lpp[u * sk] = acc;
      }
    }
    for (; k < k1; ++k, g += M4, lpp += sk) {
      const float4 xa = ld_value4<NTL>(x + g);
      float acca = rowc - krep_terms<DIST>(xa, m, c);
      acca = group_sum_down(acca, lig, G, p2);
      if (on && lig == 0) lpp[0] = acca;
    }
  }
}

// The same log-density on the flat-plane tiling of k_sample_tile (a workgroup owns TB consecutive float4 groups of the parameter plane =
// whole rows; every lane busy; every load of a wave one contiguous 1 KB segment; four value rows requested before the first is
// used; row sums through LDS, written coalesced along the particle axis).  For streams beyond the Infinity Cache.
// SAMPLE: the given stream is the standard draw (Normal: eps; Logistic: u in (0, 1)) -- the parity route of K1 / L1 --: the sample
// z = mu + sigma * e (two roundings, normal.py:105) is written as well and the density is that of the fresh sample.
template <int DIST, bool NTL, bool SAMPLE = false>
__global__ __launch_bounds__(1024) void k_logprob_tile(
    const float4* __restrict__ x, const float4* __restrict__ mu, const float4* __restrict__ sigma, float* __restrict__ lp,
    uint32_t K, int64_t R, uint32_t D4, uint32_t M4, uint32_t kchunk, uint32_t KB, uint32_t n_ptiles, uint32_t total,
    int64_t sk, int64_t sr, bool ls, float4* __restrict__ z) {
  extern __shared__ float zs_k2_stage[];
  const uint32_t TB = blockDim.x, LDW = TB + 1, tid = threadIdx.x;
  const uint32_t rows_in_tile = TB / D4;
  // Items of one tile x kchunk particles, one per workgroup: the dispatcher evens the CUs out.  Every item re-reads its tile's
  // parameters: 1.14 x the algorithmic traffic at 4.2 M rows with four chunks per tile (profiles/r04_pmc_k2_4M.json).  (Equal
  // contiguous shares of the particle-tile plane for the resident workgroups remove those re-reads and were SLOWER, 71 - 73 % ->
  // 59 - 63 % of 8 TB/s at 4.2 M rows: profiles/r05_k2_balanced.txt; that variant lives in tools/lab/.)
  for (uint32_t t = blockIdx.x; t < total; t += gridDim.x) {
    const uint32_t kt = t / n_ptiles;
    const uint32_t pt = t - kt * n_ptiles;
    const uint32_t k0 = kt * kchunk;
    const uint32_t k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    const uint32_t m4 = pt * TB + tid;
    const bool on = m4 < M4;
    const uint32_t m4c = on ? m4 : M4 - 1;                  // (idle lanes of the last tile re-read its last piece: loads stay unconditional)
    const float4 m = mu[m4c];
    const float4 s = sigma_of(sigma[m4c], ls);
    float rowc = 0.f, c[4];
    {
      const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (DIST == D_UNIFORM) {
          const float mv = j == 0 ? m.x : (j == 1 ? m.y : (j == 2 ? m.z : m.w));
          rowc -= logf(sv[j] - mv);
          c[j] = sv[j];
          continue;
        }
        const float l2 = log2_fast(sv[j]);
        if (DIST == D_NORMAL) {
          rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2;
          c[j] = 0.5f * exp2_fast(-2.0f * l2);
        } else {
          rowc -= l2 * ZS_LN2;
          c[j] = 1.0f / sv[j];
        }
      }
    }
    const int64_t rbase = (int64_t)pt * rows_in_tile;
    for (uint32_t kb0 = k0; kb0 < k1; kb0 += KB) {
      const uint32_t kb = (k1 - kb0 < KB) ? (k1 - kb0) : KB;
      const float4* __restrict__ xp = x + ((uint64_t)kb0 * M4 + m4c);
      float* __restrict__ stp = zs_k2_stage + tid;
      for (uint32_t kk = 0; kk < kb; kk += 4) {
        float4 xv[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) xv[u] = ld_value4<NTL>(xp + (uint64_t)(kk + u < kb ? kk + u : kb - 1) * M4);
#pragma unroll
        for (uint32_t u = 0; u < 4; ++u) {
          if (kk + u >= kb) continue;
          if (!SAMPLE) {
            stp[(kk + u) * LDW] = rowc - krep_terms<DIST>(xv[u], m, c);
          } else {
            float4 e = xv[u];
            float dens = 0.f;
            if (DIST == D_LOGISTIC) {
              float d0, d1, d2, d3;
              logistic_draw(xv[u].x, e.x, d0);
              logistic_draw(xv[u].y, e.y, d1);
              logistic_draw(xv[u].z, e.z, d2);
              logistic_draw(xv[u].w, e.w, d3);
              dens = (d0 + d1) + (d2 + d3);
            }
            float4 zz;
            zz.x = mul_add_2round(m.x, s.x, e.x);
            zz.y = mul_add_2round(m.y, s.y, e.y);
            zz.z = mul_add_2round(m.z, s.z, e.z);
            zz.w = mul_add_2round(m.w, s.w, e.w);
            if (on) {
              float4* __restrict__ zp = z + ((uint64_t)(kb0 + kk + u) * M4 + m4);
              if (NTL) {
                const zs_f4v v = {zz.x, zz.y, zz.z, zz.w};
                __builtin_nontemporal_store(v, reinterpret_cast<zs_f4v*>(zp));
              } else {
                *zp = zz;
              }
            }
            if (DIST == D_NORMAL) {
              const float d0 = zz.x - m.x, d1 = zz.y - m.y, d2 = zz.z - m.z, d3 = zz.w - m.w;
              stp[(kk + u) * LDW] = rowc - (fmaf(c[2], d2 * d2, c[0] * (d0 * d0)) + fmaf(c[3], d3 * d3, c[1] * (d1 * d1)));
            } else {
              stp[(kk + u) * LDW] = rowc + dens;
            }
          }
        }
      }
      __syncthreads();
      const uint32_t nout = rows_in_tile * kb;
      for (uint32_t o = tid; o < nout; o += TB) {
        uint32_t q, kk;
        if (kb == 16u) { q = o >> 4; kk = o & 15u; }
        else { q = o / kb; kk = o - q * kb; }
        const float* __restrict__ src = zs_k2_stage + kk * LDW + q * D4;
        float sum = 0.f;
        for (uint32_t j = 0; j < D4; ++j) sum += src[j];
        if (lp && rbase + q < R) lp[(int64_t)(kb0 + kk) * sk + (rbase + q) * sr] = sum;
      }
      __syncthreads();
    }
  }
}

template <int DIST>
inline void launch_logprob_krep(int kid, const float* x, const float* mu, const float* sigma, float* lp, int64_t K, int64_t R,
                                int D4, int64_t sk, int64_t sr, bool ls, hipStream_t st) {
  const RowMap rm = row_map(D4);
  const int64_t row_tiles = (R + rm.rpw - 1) / rm.rpw;
  // One work item (row tile x chunk of particles) per wave, the grid sized to the items.  When every item fits on the
  // chip at once (256 CUs x 32 wave slots) the particles are split only as far as needed to give every SIMD two
  // waves -- one round, long chunks (1 M rows: 74 % with 50-particle chunks, 67 % with 25, 60 % with 5: every item pays
  // the parameter loads and eight logarithms / exponentials); with more row tiles than slots, four items per tile so
  // that the dispatcher can even out the rounds (4.2 M rows: 57 -> 60-62 %).
  const int64_t slots = 256 * 32;
  int64_t kt = row_tiles > slots ? 4 : (row_tiles >= 2048 ? 1 : slots / row_tiles);   // >= 2 waves per SIMD: do not split
  static const int kt_env = env_knob("ZS_K2_KT", 0);         // experiments only
  if (kt_env > 0) kt = kt_env;
  if (kt < 1) kt = 1;
  if (kt > K) kt = K;
  int64_t kchunk = (K + kt - 1) / kt;
  // chunks of at least 4 particles (two loop rounds of two rows in flight) -- 2 when even that leaves most wave slots empty
  // (the config shapes: the second round of cold-cache loads is what the kernel then waits for)
  static const int kmin_env = env_knob("ZS_K2_KMIN", 0);     // experiments only
  const int64_t kmin = kmin_env > 0 ? kmin_env : (row_tiles * ((K + 3) / 4) * 4 < slots ? 2 : 4);
  if (kchunk < kmin) kchunk = K < kmin ? K : kmin;
  const int64_t total = row_tiles * ((K + kchunk - 1) / kchunk);
  // Rows in flight per wave and the load policy, measured at K = 50, D = 40 (profiles/r04_k2_variants.txt; fraction of 8 TB/s for
  // U = 2 / 4 rows in flight, plain / non-temporal loads of the value stream):
  //      rows      MB     U2 plain   U2 nt   U4 plain   U4 nt
  //     131 k      22       42.1      39.2     44.8      42.1
  //       1 M     179       63.3      49.7     70.5      55.2        <- inside the 256 MB Infinity Cache: the hint throws the reuse away
  //       2 M     357       61.6      66.2     59.0      60.2
  //     4.2 M     715       62.5      66.7     64.7      66.9        <- read once from HBM: the hint pays, the depth does not matter
  // so: four rows in flight while the stream fits the cache (and the chunk has four rows), two rows + non-temporal loads beyond.
  // The flat-plane kernel where its tiling exists (rows of whole 16-byte pieces that fit a workgroup of whole waves): against the
  // row-tile kernel below, one box, K = 50, D = 40 (profiles/r04_k2_variants.txt): 131 k rows 6.4 -> 5.2 us, 1 M rows K2 70 -> 76 %,
  // L2 57 -> 73 %, U2 70 -> 75 %, 2 M rows 58 -> 64 %, 4.2 M rows 61 -> 64-69 % (non-temporal loads beyond the Infinity Cache).
  static const int tile_env = env_knob("ZS_K2_TILE", -1);      // experiments only: 0 = the row-tile kernel, 1 / 2 = plain / non-temporal loads
  if (tile_env != 0) {
    K1Tile g = k1_tile(K, R, D4, true);
    if (g.ok) {
      const bool stream_once = tile_env > 0 ? tile_env == 2 : (double)K * (double)R * (double)D4 * 16.0 > 268435456.0;
      if (stream_once)
        ZS_LAUNCH_SMEM(kid, (k_logprob_tile<DIST, true>), dim3(g.grid), dim3(g.threads), g.smem, st, (const float4*)x, (const float4*)mu,
                       (const float4*)sigma, lp, (uint32_t)K, R, (uint32_t)D4, (uint32_t)(R * D4), g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, ls,
                       (float4*)nullptr);
      else
        ZS_LAUNCH_SMEM(kid, (k_logprob_tile<DIST, false>), dim3(g.grid), dim3(g.threads), g.smem, st, (const float4*)x, (const float4*)mu,
                       (const float4*)sigma, lp, (uint32_t)K, R, (uint32_t)D4, (uint32_t)(R * D4), g.kchunk, g.KB, g.n_ptiles, g.total, sk, sr, ls,
                       (float4*)nullptr);
      return;
    }
  }
  static const int u_env = env_knob("ZS_K2_U", 0), ntl_env = env_knob("ZS_K2_NTL", -1);      // experiments only
  const bool big = (double)K * (double)R * (double)D4 * 16.0 > 268435456.0;
  const int U = u_env > 0 ? u_env : ((!big && kchunk >= 4) ? 4 : 2);
  const bool ntl = ntl_env >= 0 ? ntl_env != 0 : big;
#define ZS_LAUNCH_KREP(UU, NN)                                                                                          \
  ZS_LAUNCH(kid, (k_logprob_krep<DIST, UU, NN>), dim3(grid_for(total, 4, 1u << 22)), dim3(256), st, (const float4*)x, \
            (const float4*)mu, (const float4*)sigma, lp, K, R, D4, rm.G, rm.rpw, rm.p2, kchunk, sk, sr, ls)
  if (U >= 4) { if (ntl) ZS_LAUNCH_KREP(4, true); else ZS_LAUNCH_KREP(4, false); }
  else        { if (ntl) ZS_LAUNCH_KREP(2, true); else ZS_LAUNCH_KREP(2, false); }
#undef ZS_LAUNCH_KREP
}

// ------------------------------------------------------------------------------------
// Backward of the given-value log-density reduced over the K particles, parameters [R, D] repeated over them (K2 bwd-ksum,
// normal.py:102,112-116; L2 bwd-ksum, logistic.py:81-82): tile = 64 parameter float4 groups x 4 K-slices, the slices'
// partial sums combined through LDS.  Per element, with g = the row's incoming gradient:
//   Normal:   t = g (x - mu) / sigma^2:          gx = -t,  gmu += t,   gsigma += g ((x - mu)^2 / sigma^2 - 1) / sigma
//   Logistic: u = (x - loc) / s, h = tanh(u/2):  gx = -g h / s,  gloc += g h / s,  gscale += g (h u - 1) / s
// ------------------------------------------------------------------------------------
template <int DIST>
__device__ __forceinline__ void ksum_elem(float g, float xv, float mv, float c, float inv, float& gx, float& a, float& b) {
  const float diff = xv - mv;
  if (DIST == D_NORMAL) {            // c = sigma^-2, inv = 1/sigma (or 1: the parameter is log sigma)
    const float t = g * c * diff;
    gx = -t;
    a += t;
    b += g * (c * diff * diff - 1.0f) * inv;
  } else {                           // inv = 1/scale
    logistic_ksum_elem(g, diff, inv, gx, a, b);
  }
}

// (the body as a device function of the workgroup's index: also one ROLE of IW1's merged backward launch, zs_bernoulli.hip)
template <int DIST, int NS>
__device__ __forceinline__ void logprob_bwd_ksum_body(
    int64_t block, const float4* __restrict__ x, const float4* __restrict__ mu, const float4* __restrict__ sigma,
    const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float4* __restrict__ gx, float4* __restrict__ gmu, float4* __restrict__ gsigma,
    int64_t K, int64_t M4, int D4, bool ls, const float* __restrict__ gscale, int64_t gss) {
  __shared__ float4 red[2][NS][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int64_t m4 = block * 64 + lane;
  const bool on = m4 < M4;
  float4 am = make_float4(0.f, 0.f, 0.f, 0.f), as = am;
  if (on) {
    const int64_t r = (int64_t)((uint64_t)m4 >> 31 ? m4 / D4 : (int64_t)((uint32_t)m4 / (uint32_t)D4));
    const float4 m = mu[m4], s = sigma_of(sigma[m4], ls);
    const float sv[4] = {s.x, s.y, s.z, s.w};
    const float mv[4] = {m.x, m.y, m.z, m.w};
    float pr[4], inv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      pr[j] = DIST == D_NORMAL ? exp2_fast(-2.0f * log2_fast(sv[j])) : 0.f;
      inv[j] = (DIST == D_NORMAL && ls) ? 1.0f : 1.0f / sv[j];     // d/d logstd = sigma * d/d sigma
    }
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    // the slice index is wave-uniform: scalar loop counter, and the three streams advance by one 64-bit add each per particle
    // (instead of k * M4 + m4 and k * gsk + r * gsr with 64-bit multiplies per lane)
    const int64_t ks = __builtin_amdgcn_readfirstlane(slice);
    const float4* __restrict__ xp = x + (ks * M4 + m4);
    float4* __restrict__ gxp = gx ? gx + (ks * M4 + m4) : nullptr;
    const float* __restrict__ glpp = glp + (ks * gsk + r * gsr);
    // optional scale of the row gradients (the objective's incoming gradient: one device scalar, gss = 0, or one per row r)
    const float gs = gscale ? gscale[r * gss] : 1.0f;
    const int64_t zstep = (int64_t)NS * M4, lstep = (int64_t)NS * gsk;
    for (int64_t k = ks; k < K; k += NS) {         // (unrolling by 4 was measured: 76 -> 67 % at 1 M rows, registers)
      const float g = gscale ? *glpp * gs : *glpp;
      const float4 xv = *xp;
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) ksum_elem<DIST>(g, xs[j], mv[j], pr[j], inv[j], t[j], a[j], b[j]);
      if (gx) { *gxp = make_float4(t[0], t[1], t[2], t[3]); gxp += zstep; }
      xp += zstep;
      glpp += lstep;
    }
    am = make_float4(a[0], a[1], a[2], a[3]);
    as = make_float4(b[0], b[1], b[2], b[3]);
  }
  red[0][slice][lane] = am;
  red[1][slice][lane] = as;
  __syncthreads();
  if (slice == 0 && on) {
    float4 a = red[0][0][lane], b = red[1][0][lane];
#pragma unroll
    for (int s = 1; s < NS; ++s) {
      const float4 a2 = red[0][s][lane], b2 = red[1][s][lane];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    if (gmu) gmu[m4] = a;
    if (gsigma) gsigma[m4] = b;
  }
}

template <int DIST, int NS>
__global__ __launch_bounds__(64 * NS) void k_logprob_bwd_ksum(
    const float4* __restrict__ x, const float4* __restrict__ mu, const float4* __restrict__ sigma,
    const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float4* __restrict__ gx, float4* __restrict__ gmu, float4* __restrict__ gsigma,
    int64_t K, int64_t M4, int D4, bool ls, const float* __restrict__ gscale, int64_t gss) {
  logprob_bwd_ksum_body<DIST, NS>(blockIdx.x, x, mu, sigma, glp, gsk, gsr, gx, gmu, gsigma, K, M4, D4, ls, gscale, gss);
}

// rows that are not a multiple of four elements, or unaligned operands: a thread per parameter element
template <int DIST>
__global__ __launch_bounds__(256) void k_logprob_bwd_ksum_serial(
    const float* __restrict__ x, const float* __restrict__ mu, const float* __restrict__ sigma,
    const float* __restrict__ glp, int64_t gsk, int64_t gsr,
    float* __restrict__ gx, float* __restrict__ gmu, float* __restrict__ gsigma,
    int64_t K, int64_t M, int64_t D, bool ls, const float* __restrict__ gscale, int64_t gss) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = m / D;
    const float gs = gscale ? gscale[r * gss] : 1.0f;
    const float s = sigma_of(sigma[m], ls), mm = mu[m];
    const float inv = (DIST == D_NORMAL && ls) ? 1.0f : 1.0f / s;
    const float prec = DIST == D_NORMAL ? exp2_fast(-2.0f * log2_fast(s)) : 0.f;
    float a = 0.f, b = 0.f;
    for (int64_t k = 0; k < K; ++k) {
      const float g = gscale ? glp[k * gsk + r * gsr] * gs : glp[k * gsk + r * gsr];
      float t;
      ksum_elem<DIST>(g, x[k * M + m], mm, prec, inv, t, a, b);
      if (gx) gx[k * M + m] = t;
    }
    if (gmu) gmu[m] = a;
    if (gsigma) gsigma[m] = b;
  }
}

template <int DIST>
inline void launch_logprob_bwd_ksum(int kid, const float* x, const float* mu, const float* sigma, const float* glp, int64_t gsk,
                                    int64_t gsr, float* gx, float* gmu, float* gsigma, int64_t K, int64_t R, int64_t D, bool ls,
                                    hipStream_t st, const float* gscale = nullptr, int64_t gss = 0) {
  const int64_t M = R * D;
  const bool vec = (D % 4 == 0) && aligned16(x) && aligned16(mu) && aligned16(sigma) && (!gx || aligned16(gx)) &&
                   (!gmu || aligned16(gmu)) && (!gsigma || aligned16(gsigma));
  if (vec) {
    const int64_t M4 = M / 4;
    // K-slices per workgroup: 4, or 16 when the parameter plane gives fewer than 256 workgroups (the config shapes: 40
    // workgroups whose waves each walked 13 particles one after the other -- 7.5 us for 2.3 MB)
    if ((M4 + 63) / 64 < 256 && K >= 16)
      ZS_LAUNCH(kid, (k_logprob_bwd_ksum<DIST, 16>), dim3((unsigned)((M4 + 63) / 64)), dim3(1024), st, (const float4*)x,
                (const float4*)mu, (const float4*)sigma, glp, gsk, gsr, (float4*)gx, (float4*)gmu, (float4*)gsigma, K, M4,
                (int)(D / 4), ls, gscale, gss);
    else
      ZS_LAUNCH(kid, (k_logprob_bwd_ksum<DIST, 4>), dim3((unsigned)((M4 + 63) / 64)), dim3(256), st, (const float4*)x,
                (const float4*)mu, (const float4*)sigma, glp, gsk, gsr, (float4*)gx, (float4*)gmu, (float4*)gsigma, K, M4,
                (int)(D / 4), ls, gscale, gss);
  } else {
    ZS_LAUNCH(kid, (k_logprob_bwd_ksum_serial<DIST>), dim3(grid_for(M, 256)), dim3(256), st, x, mu, sigma, glp, gsk, gsr, gx, gmu,
              gsigma, K, M, D, ls, gscale, gss);
  }
}

}  // namespace zs
