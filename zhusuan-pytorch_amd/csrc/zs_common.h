// Device-side helpers shared by the gfx950 kernels of the VI hot path.
// Wavefront = 64 lanes everywhere (CDNA4); no MFMA on this path (no dense contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ZS_WAVE 64
#define ZS_LN2 0.69314718055994530942f
#define ZS_NEG_HALF_LOG_2PI (-0.91893853320467274178f)  // -0.5*log(2*pi), normal.py:122
#define ZS_BERN_EPS 1e-8f                                 // bernoulli.py:94

#define ZS_CHECK_LAUNCH()                           \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

namespace zs {

// ---------------------------------------------------------------- experiment knobs
// The SHIPPED library never reads the environment: dispatch depends on the arguments of a call and on nothing else.  The
// ZS_* knobs that kernel experiments use (tools/) exist only in a library built with
// `make EXTRA=-DZS_EXPERIMENTS`; zs_build_info() of a loaded library says which kind it is, and bench.py records it.
// ---------------------------------------------------------------- the hand-off between workgroups, two builds
// The one-launch kernels end with "every workgroup writes partials, the LAST to arrive combines them".  The shipped build orders that
// hand-off the cheap way MI355X_MICROARCH.md lists as MEASURED on gfx950 (write-through sc1 stores, every storing wave's
// s_waitcnt vmcnt(0), a workgroup barrier, ONE relaxed agent-scope atomic on the ticket, sc1 loads or one agent acquire on the
// consuming side: zs_onelaunch.h) -- not an architectural guarantee.  `make strict` (-DZS_STRICT_HANDOFF) builds the same kernels
// with the ticket taken ACQ_REL at agent scope -- buffer_wbl2 + wait before it, buffer_inv after it: the form the HIP memory
// model guarantees, 2 - 7 us slower per workgroup -- and tests/test_strict_handoff.py requires both builds to return the same
// bits over thousands of launches: a compiler or firmware change that breaks the cheap form shows there instead of as a rare
// wrong number.  Never shipped: zs_build_info() says which build a library is.
#ifdef ZS_STRICT_HANDOFF
#define ZS_TICKET_ORDER __ATOMIC_ACQ_REL
#define ZS_HANDOFF_KIND ", strict hand-off (agent-scope acq_rel tickets)"
#else
#define ZS_TICKET_ORDER __ATOMIC_RELAXED
#define ZS_HANDOFF_KIND ""
#endif

#ifdef ZS_EXPERIMENTS
inline int env_knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#define ZS_BUILD_KIND "experiments (reads ZS_* environment knobs)"
#else
constexpr int env_knob(const char*, int dflt) { return dflt; }
#define ZS_BUILD_KIND "release (no environment knobs)"
#endif

// ---------------------------------------------------------------- launch + optional per-kernel timing
// One id per C-ABI entry point.  When profiling is enabled (zs_prof_enable) a launch goes through
// hipExtLaunchKernelGGL with a start/stop event pair bound to the dispatch itself, so the elapsed time
// is the kernel's own duration on its stream (what rocprofv3 --kernel-trace reports), not the
// host-side gap between two hipEventRecord calls.
enum KernelId {
  KID_NORMAL_SAMPLE = 0, KID_NORMAL_SAMPLE_BWD, KID_NORMAL_LOGPROB, KID_NORMAL_LOGPROB_BWD,
  KID_NORMAL_LOGPROB_BWD_KSUM, KID_BERN_LOGPROB, KID_BERN_LOGPROB_BWD, KID_BERN_LOGITS_LOGPROB,
  KID_BERN_LOGITS_LOGPROB_BWD, KID_BERN_SAMPLE, KID_IW_REDUCE, KID_LME, KID_PHILOX,
  KID_LOGISTIC_SAMPLE, KID_LOGISTIC_SAMPLE_BWD, KID_LOGISTIC_LOGPROB, KID_LOGISTIC_LOGPROB_BWD,
  KID_UNIFORM_SAMPLE, KID_UNIFORM_LOGPROB, KID_PHILOX_UNIFORM, KID_REINFORCE, KID_IW_OBJECTIVE, KID_SCALAR_OBJECTIVE, KID_ADAM, KID_LOGISTIC_LOGPROB_BWD_KSUM,
  KID_LOGJOINT, KID_LOGJOINT_BWD, KID_NORMAL_SAMPLE_MULTI, KID_NORMAL_SAMPLE_MULTI_BWD, KID_PARTICLE_LINEAR, KID_PARTICLE_LINEAR_BWD, KID_COLUMN_SUM,
  KID_DENSE_ACT_BWD, KID_PARTICLE_RMSE, KID_PARTICLE_MLP, KID_PARTICLE_MLP_BWD, KID_BERN_IW_OBJECTIVE, KID_BERN_IW_OBJECTIVE_BWD,
  KID_NORMAL_SAMPLE_PAIR, KID_BERN_LOGPROB_BWD_X,
  KID_COUNT
};
bool prof_begin_launch(int kid, hipEvent_t* start, hipEvent_t* stop);  // defined in zs_iw.hip

#define ZS_LAUNCH(kid, kern, grid, block, st, ...)                                             \
  do {                                                                                         \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                 \
    if (zs::prof_begin_launch(kid, &e0__, &e1__))                                              \
      hipExtLaunchKernelGGL(kern, grid, block, 0, st, e0__, e1__, 0, __VA_ARGS__);             \
    else                                                                                       \
      hipLaunchKernelGGL(kern, grid, block, 0, st, __VA_ARGS__);                               \
  } while (0)

// same, with `smem` bytes of dynamic LDS
#define ZS_LAUNCH_SMEM(kid, kern, grid, block, smem, st, ...)                                  \
  do {                                                                                         \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                 \
    if (zs::prof_begin_launch(kid, &e0__, &e1__))                                              \
      hipExtLaunchKernelGGL(kern, grid, block, smem, st, e0__, e1__, 0, __VA_ARGS__);          \
    else                                                                                       \
      hipLaunchKernelGGL(kern, grid, block, smem, st, __VA_ARGS__);                            \
  } while (0)

// ---------------------------------------------------------------- math
// The per-element arithmetic of the kernels is __host__ __device__: tests/host_math/zs_host_math.hip compiles it for the
// HOST (hipcc --cuda-host-only) under AddressSanitizer + UndefinedBehaviorSanitizer and checks it against double-precision
// references (GPU sanitizers are not available on this pool).  ZS_ON_DEVICE selects the gfx950
// instruction; the host branch is the same function by definition (log2f for v_log_f32, ...), used by that test only.
#if defined(__HIP_DEVICE_COMPILE__)
#define ZS_ON_DEVICE 1
#else
#define ZS_ON_DEVICE 0
#include <math.h>
#endif
#define ZS_HD __host__ __device__ __forceinline__

// v_log_f32 / v_exp_f32 are base-2 and 1-ulp; natural log/exp are one multiply away.
ZS_HD float log2_fast(float x) {
#if ZS_ON_DEVICE
  return __builtin_amdgcn_logf(x);
#else
  return log2f(x);
#endif
}
ZS_HD float exp2_fast(float x) {
#if ZS_ON_DEVICE
  return __builtin_amdgcn_exp2f(x);
#else
  return exp2f(x);
#endif
}
ZS_HD float rcp_fast(float x) {
#if ZS_ON_DEVICE
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
ZS_HD float sqrt_fast(float x) {       // raw v_sqrt_f32 (1 ulp): the IEEE fix-up of sqrtf() costs ~20 instructions per root
#if ZS_ON_DEVICE
  return __builtin_amdgcn_sqrtf(x);
#else
  return sqrtf(x);
#endif
}
ZS_HD float sin_rev(float x) {         // v_sin_f32 / v_cos_f32 take their argument in revolutions
#if ZS_ON_DEVICE
  return __builtin_amdgcn_sinf(x);
#else
  return (float)sin(6.283185307179586476925 * (double)(x - floorf(x)));
#endif
}
ZS_HD float cos_rev(float x) {
#if ZS_ON_DEVICE
  return __builtin_amdgcn_cosf(x);
#else
  return (float)cos(6.283185307179586476925 * (double)(x - floorf(x)));
#endif
}
ZS_HD float ln_fast(float x) { return log2_fast(x) * ZS_LN2; }
ZS_HD float exp_fast(float x) { return exp2_fast(x * 1.44269504088896340736f); }

// Normal log-density term for one element given log2(sigma) and prec = sigma^-2
// (normal.py:121-124: c - logstd - 0.5 * precision * (x - mean)^2).
ZS_HD float normal_lp_term(float diff, float logstd, float prec) {
  return (ZS_NEG_HALF_LOG_2PI - logstd) - 0.5f * prec * (diff * diff);
}

// Bernoulli term in log2 units (bernoulli.py:94); caller multiplies the row sum by ln 2.
ZS_HD float bern_lp2_term(float p, float x) {
  float a = log2_fast(p + ZS_BERN_EPS);
  float b = log2_fast((1.0f - p) + ZS_BERN_EPS);
  return x * a + (1.0f - x) * b;
}
// The same term for one 16-byte piece (four elements) in PACKED fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma_f32: two elements
// per instruction), accumulated pairwise into `acc`: per element 2.5 full-rate instructions + the two logarithms instead of 7 + 2.
// The kernels that give a workgroup a fixed share of the rows (IW1) are bound by exactly this arithmetic.  `omx` = 1 - x.
typedef float zs_f2v __attribute__((ext_vector_type(2)));
typedef float zs_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bern_piece_acc(const float4& p, const float4& x, const float4& omx, zs_f2v& acc) {
  const zs_f2v eps = {ZS_BERN_EPS, ZS_BERN_EPS}, one = {1.0f, 1.0f};
  const zs_f2v p0 = {p.x, p.y}, p1 = {p.z, p.w};
  const zs_f2v a0 = p0 + eps, a1 = p1 + eps;
  const zs_f2v b0 = (one - p0) + eps, b1 = (one - p1) + eps;
  const zs_f2v la0 = {log2_fast(a0.x), log2_fast(a0.y)}, la1 = {log2_fast(a1.x), log2_fast(a1.y)};
  const zs_f2v lb0 = {log2_fast(b0.x), log2_fast(b0.y)}, lb1 = {log2_fast(b1.x), log2_fast(b1.y)};
  const zs_f2v x0 = {x.x, x.y}, x1 = {x.z, x.w}, o0 = {omx.x, omx.y}, o1 = {omx.z, omx.w};
  acc += __builtin_elementwise_fma(o0, lb0, x0 * la0);
  acc += __builtin_elementwise_fma(o1, lb1, x1 * la1);
}
// The same piece when every x of the row is exactly 0 or 1 (binarised observations: the examples' data): with s = 2x - 1 and
// c = 1 - x the selected argument is fma(p, s, c) -- p for x = 1, 1 - p (rounded as above) for x = 0 -- and the term is ONE
// logarithm: x * la + (1 - x) * lb has a factor exactly 1 and a factor exactly 0.  Bit-identical to bern_piece_acc for every p
// whose two arguments are positive (any p in [0, 1]); for an invalid p (outside [-1e-8, 1 + 1e-8]) the general form returns NaN
// through 0 * log(negative) where this one does not evaluate the other branch.
__device__ __forceinline__ void bern_piece_acc_bits(const float4& p, const float4& s, zs_f2v& acc) {
  const zs_f2v eps = {ZS_BERN_EPS, ZS_BERN_EPS}, half = {0.5f, 0.5f}, mhalf = {-0.5f, -0.5f};
  const zs_f2v p0 = {p.x, p.y}, p1 = {p.z, p.w}, s0 = {s.x, s.y}, s1 = {s.z, s.w};
  const zs_f2v c0 = __builtin_elementwise_fma(s0, mhalf, half), c1 = __builtin_elementwise_fma(s1, mhalf, half);     // 1 - x, exactly
  const zs_f2v a0 = __builtin_elementwise_fma(p0, s0, c0) + eps, a1 = __builtin_elementwise_fma(p1, s1, c1) + eps;
  const zs_f2v l0 = {log2_fast(a0.x), log2_fast(a0.y)}, l1 = {log2_fast(a1.x), log2_fast(a1.y)};
  acc += l0;
  acc += l1;
}
// ... with s = 2x - 1 AND c = 1 - x handed in (both parked in LDS once per datapoint by the persistent fused kernel): the same bits
// for rows of 0s and 1s, two packed instructions fewer per piece.  (s = 0, c = 1 makes the piece contribute log 1 = 0: padding.)
__device__ __forceinline__ void bern_piece_acc_bits2(const float4& p, const float4& s, const float4& c, zs_f2v& acc) {
  const zs_f2v eps = {ZS_BERN_EPS, ZS_BERN_EPS};
  const zs_f2v p0 = {p.x, p.y}, p1 = {p.z, p.w}, s0 = {s.x, s.y}, s1 = {s.z, s.w}, c0 = {c.x, c.y}, c1 = {c.z, c.w};
  const zs_f2v a0 = __builtin_elementwise_fma(p0, s0, c0) + eps, a1 = __builtin_elementwise_fma(p1, s1, c1) + eps;
  const zs_f2v l0 = {log2_fast(a0.x), log2_fast(a0.y)}, l1 = {log2_fast(a1.x), log2_fast(a1.y)};
  acc += l0;
  acc += l1;
}
// Gradient of that term w.r.t. p for one 16-byte piece, times the row gradient g, in packed fp32 arithmetic:
//   g * (x / (p + eps) - (1 - x) / ((1 - p) + eps))            [LOGITS: p = sigmoid(l), result times p (1 - p)]
// 3.5 packed instructions + two reciprocals per element instead of ~9 + 2: K3's backward stalls on instruction issue for half
// of its wave cycles (SQ_WAIT_INST_ANY 0.50 at 6.6 GB, 0.59 in IW1's merged backward; profiles/r04_pmc_*.json).
template <bool LOGITS>
__device__ __forceinline__ float4 bern_piece_grad(const float4& pl, const float4& x, float g) {
  const zs_f2v eps = {ZS_BERN_EPS, ZS_BERN_EPS}, one = {1.0f, 1.0f}, gg = {g, g};
  zs_f2v p0 = {pl.x, pl.y}, p1 = {pl.z, pl.w};
  if (LOGITS) {
    p0.x = rcp_fast(1.0f + exp_fast(-p0.x)); p0.y = rcp_fast(1.0f + exp_fast(-p0.y));
    p1.x = rcp_fast(1.0f + exp_fast(-p1.x)); p1.y = rcp_fast(1.0f + exp_fast(-p1.y));
  }
  const zs_f2v q0 = one - p0, q1 = one - p1;
  const zs_f2v a0 = p0 + eps, a1 = p1 + eps, b0 = q0 + eps, b1 = q1 + eps;
  const zs_f2v ra0 = {rcp_fast(a0.x), rcp_fast(a0.y)}, ra1 = {rcp_fast(a1.x), rcp_fast(a1.y)};
  const zs_f2v rb0 = {rcp_fast(b0.x), rcp_fast(b0.y)}, rb1 = {rcp_fast(b1.x), rcp_fast(b1.y)};
  const zs_f2v x0 = {x.x, x.y}, x1 = {x.z, x.w};
  zs_f2v t0 = __builtin_elementwise_fma(x0 - one, rb0, x0 * ra0);          // x ra - (1 - x) rb
  zs_f2v t1 = __builtin_elementwise_fma(x1 - one, rb1, x1 * ra1);
  t0 = gg * t0;
  t1 = gg * t1;
  if (LOGITS) {
    t0 = t0 * p0 * q0;
    t1 = t1 * p1 * q1;
  }
  return make_float4(t0.x, t0.y, t1.x, t1.y);
}
// torch.sigmoid: 1 / (1 + exp(-l)), bernoulli.py:50
ZS_HD float sigmoid_fast(float l) { return rcp_fast(1.0f + exp_fast(-l)); }
// d/dp of x*log(p+e) + (1-x)*log((1-p)+e)
ZS_HD float bern_dp(float p, float x) {
  return x * rcp_fast(p + ZS_BERN_EPS) - (1.0f - x) * rcp_fast((1.0f - p) + ZS_BERN_EPS);
}

// d/dx of x*log(p+e) + (1-x)*log((1-p)+e) = log(p+e) - log((1-p)+e): the gradient w.r.t. the observation (bernoulli.py:94)
ZS_HD float log_ratio_any(float p) { return (log2_fast(p + ZS_BERN_EPS) - log2_fast((1.0f - p) + ZS_BERN_EPS)) * ZS_LN2; }
ZS_HD double log_ratio_any(double p) { return log(p + 1e-8) - log((1.0 - p) + 1e-8); }
ZS_HD float sigmoid_any(float l) { return sigmoid_fast(l); }
ZS_HD double sigmoid_any(double l) { return 1.0 / (1.0 + exp(-l)); }

// ---------------------------------------------------------------- index arithmetic
// 64-bit integer division expands to ~100 VALU instructions on CDNA; row / tile indices almost always fit
// 31 bits, where the 32-bit expansion is 4-5x shorter.  (At the config sizes a wave handles 1-3 rows, so
// two 64-bit divisions per row were costing as much as the row's own arithmetic.)
ZS_HD void divmod(int64_t a, int64_t b, int64_t& q, int64_t& r) {
  if ((((uint64_t)a | (uint64_t)b) >> 31) == 0) {
    const uint32_t qq = (uint32_t)a / (uint32_t)b;
    q = (int64_t)qq;
    r = (int64_t)((uint32_t)a - qq * (uint32_t)b);
  } else {
    q = a / b;
    r = a - q * b;
  }
}
ZS_HD int64_t mod_fast(int64_t a, int64_t b) {
  int64_t q, r;
  divmod(a, b, q, r);
  return r;
}

// ---------------------------------------------------------------- wave reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, ZS_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, ZS_WAVE));
  return v;
}
// The same reductions by DPP (row-of-16 butterflies, then the row totals passed up by row_bcast:15 / :31): 6 VALU instructions
// with the result in LANE 63 -- the __shfl_xor butterflies above are 6 ds_bpermute round trips through the LDS crossbar, a
// dependent chain of ~6 x 100+ cycles for the one wave that runs a datapoint's K-particle reduction.  `_all` broadcasts lane 63
// (v_readlane).  Another order of summation than the butterfly: results differ from it in the last bits.
#if ZS_ON_DEVICE
#define ZS_DPP_F(v, ctrl, rmask, bc) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, bc))
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += ZS_DPP_F(v, 0xB1, 0xf, true);       // quad_perm [1,0,3,2]
  v += ZS_DPP_F(v, 0x4E, 0xf, true);       // quad_perm [2,3,0,1]
  v += ZS_DPP_F(v, 0x141, 0xf, true);      // row_half_mirror
  v += ZS_DPP_F(v, 0x140, 0xf, true);      // row_mirror: every lane = its row's total
  v += ZS_DPP_F(v, 0x142, 0xa, false);     // row_bcast:15 into rows 1 and 3
  v += ZS_DPP_F(v, 0x143, 0xc, false);     // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(v)), 63));
}
// the sum over lanes 0 .. 15 only (the first DPP row), in every lane
__device__ __forceinline__ float row0_sum_all(float v) {
  v += ZS_DPP_F(v, 0xB1, 0xf, true);
  v += ZS_DPP_F(v, 0x4E, 0xf, true);
  v += ZS_DPP_F(v, 0x141, 0xf, true);
  v += ZS_DPP_F(v, 0x140, 0xf, true);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
}
// max: the row_bcast steps must not see the 0 that `old` supplies to the rows they skip, so those rows keep their own value
__device__ __forceinline__ float wave_max_all(float v) {
  v = fmaxf(v, ZS_DPP_F(v, 0xB1, 0xf, true));
  v = fmaxf(v, ZS_DPP_F(v, 0x4E, 0xf, true));
  v = fmaxf(v, ZS_DPP_F(v, 0x141, 0xf, true));
  v = fmaxf(v, ZS_DPP_F(v, 0x140, 0xf, true));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x142, 0xa, 0xf, false)));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x143, 0xc, 0xf, false)));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#else
// (the host pass of a .hip file still type-checks __device__ callers)
__device__ __forceinline__ float wave_sum_to_lane63(float v) { return v; }
__device__ __forceinline__ float wave_sum_all(float v) { return v; }
__device__ __forceinline__ float row0_sum_all(float v) { return v; }
__device__ __forceinline__ float wave_max_all(float v) { return v; }
#endif

// Sum over groups of G consecutive lanes (G need not be a power of two, groups start at
// multiples of G).  The total lands in the first lane of each group; other lanes hold junk.
__device__ __forceinline__ float group_sum_down(float v, int lane_in_group, int G, int pow2_ge_G) {
  for (int o = pow2_ge_G >> 1; o > 0; o >>= 1) {
    float t = __shfl_down(v, o, ZS_WAVE);
    if (lane_in_group + o < G) v += t;
  }
  return v;
}
__host__ __device__ __forceinline__ int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ---------------------------------------------------------------- Philox4x32-10
struct Philox4 {
  uint32_t x, y, z, w;
};
// a ^ b ^ key in ONE VALU instruction: CDNA4's v_bitop3_b32 with the parity truth table 0x96 (gfx950 has no
// v_xor3_b32, and the compiler emits two v_xor_b32 for the C expression: 38 of the 104 instructions of K1's inner loop
// were xors).  `key` is a round key: uniform across the wavefront, so it stays in a scalar register.
ZS_HD uint32_t xor3_key(uint32_t a, uint32_t b, uint32_t key) {
#if ZS_ON_DEVICE
  uint32_t r;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "s"(key));
  return r;
#else
  return a ^ b ^ key;
#endif
}
// a uniform value pinned to the scalar unit (no-op on the host)
ZS_HD uint32_t uniform_u32(uint32_t v) {
#if ZS_ON_DEVICE
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

// Rounds of the generator.  10 = Philox4x32-10, the curand / torch / Random123 default and what the oracle implements;
// anything else is for experiments only (tools/k1_launch_distribution.py with ZS_HIP_LIBRARY) and breaks RNG parity.
#ifndef ZS_PHILOX_ROUNDS
#define ZS_PHILOX_ROUNDS 10
#endif
// Measured issue costs on gfx950 (tools/valu_rates.hip, cycles per wave64 instruction per SIMD with the chip full):
// v_fma / v_bitop3 / v_mul_hi 4.3, v_xor_b32 2.5, v_mad_u64_u32 ~6, v_log / v_sqrt / v_sin / v_cos 8.  The generator is
// what bounds the fused sampling kernel, so every instruction here is counted.
//
// (seed, call) are uniform across a launch: in the first round the product 0xCD9E8D57 * lo(call) and the key xors are
// scalar-unit work, and the second round still has one uniform counter word.  Those two rounds are therefore written as
// plain C (the compiler keeps uniform values in SGPRs: one v_mad_u64_u32 and three plain xors instead of two
// multiplies and two three-input xors fed by v_mov copies); rounds 3..10 use the one-instruction three-input xor.
//
// CONTRACT: `call` and `seed` must be WAVE-UNIFORM (the same value on all 64 lanes; every caller passes launch-wide
// constants or values read from the 2-word rng_state); only `group` may differ between lanes.  The readfirstlane pins
// below would otherwise take lane 0's value for the whole wave -- on the GPU only: the host build and the C oracle have
// no such instruction, so a per-lane `call` / `seed` would pass every CPU test and still draw wrong numbers on the
// device.  On the GPU every entry point that draws is compared with the oracle (tests/test_cabi.py::test_hip_rng,
// ::test_hip_normal_sample_and_backward, ::test_hip_device_rng_state; tests/test_locscale.py::test_hip_logistic_sample_and_backward,
// ::test_hip_uniform_sample).
// Everything of rounds 1 and 2 that depends on (seed, call) only -- uniform over a launch: formed ONCE per batch of particles by
// philox_call() and handed to the generator (round 5: left inside the generator, the three v_readfirstlane pins were re-executed for
// every group of four draws -- the compiler does not hoist a convergent operation out of the particle loop; K1 is bound by VALU
// issue, 77 instructions per four draws, so three of them are 4 %).
struct PhiloxCall {
  uint32_t u0, u1, u2, u3;      // hi(M1 * lo(call)) ^ k0;  lo(M1 * lo(call));  hi(call) ^ k1;  u1 ^ (k0 + W0)
  uint32_t k0, k1;
};
ZS_HD PhiloxCall philox_call(uint64_t call, uint64_t seed) {
  const uint32_t c2 = (uint32_t)call, c3 = (uint32_t)(call >> 32), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  PhiloxCall pc;
  pc.u0 = uniform_u32((uint32_t)(p1 >> 32) ^ k0);
  pc.u1 = uniform_u32((uint32_t)p1);
  pc.u2 = uniform_u32(c3 ^ k1);
  pc.u3 = uniform_u32(pc.u1 ^ (k0 + 0x9E3779B9u));
  pc.k0 = uniform_u32(k0);
  pc.k1 = uniform_u32(k1);
  return pc;
}
ZS_HD Philox4 philox4x32_10(uint64_t group, const PhiloxCall& pc) {
  uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2, c3;
  uint32_t k0 = pc.k0, k1 = pc.k1;
  {                                              // round 1: one multiply, two xors with uniform words
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint32_t n0 = pc.u0 ^ c1, n2 = (uint32_t)(p0 >> 32) ^ pc.u2;
    c0 = n0; c1 = pc.u1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  {                                              // round 2: c1 is still uniform
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ pc.u3, n2 = xor3_key((uint32_t)(p0 >> 32), c3, k1);
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int r = 2; r < ZS_PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = xor3_key((uint32_t)(p1 >> 32), c1, k0), n2 = xor3_key((uint32_t)(p0 >> 32), c3, k1);
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o = {c0, c1, c2, c3};
  return o;
}
ZS_HD Philox4 philox4x32_10(uint64_t group, uint64_t call, uint64_t seed) { return philox4x32_10(group, philox_call(call, seed)); }
// Uniform STRICTLY inside (0, 1): (m + 0.5) * 2^-23 with m = v >> 9, i.e. the 2^23 midpoints 2^-24 ... 1 - 2^-24.  m + 0.5
// needs 24 significant bits, so every value is exact in fp32 (shift, convert, one fma).  Round 1 used m = v >> 8 and
// 2^-24: there m + 0.5 needs 25 bits, the top 256 words rounded up to exactly 1.0 and a Logistic draw log(u) - log(1 - u)
// became +inf about once in 1.7e7 draws (found by the host-side sanitizer / reference test of this header,
// tests/host_math/zs_host_math.hip).
// On the device the same value in TWO instructions instead of three (the generators are bound by instruction issue): the upper 23
// bits become the mantissa of a float in [1, 2) (v_alignbit_b32, as in angle_rev below) and 1 - 2^-24 is subtracted --
// (1 + m * 2^-23) - (1 - 2^-24) = (2m + 1) * 2^-24 is representable, so the subtraction is exact and the result bit-identical to
// the fma form for all 2^23 mantissas (the device streams are compared with the oracle's bit for bit: tests/test_locscale.py
// test_hip_uniform_sample, tests/test_cabi.py Bernoulli sampler).
ZS_HD float u01(uint32_t v) {
#if ZS_ON_DEVICE
  return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, v, 9)) - 0.99999994039535522461f;
#else
  return __builtin_fmaf((float)(v >> 9), 1.1920928955078125e-07f, 5.9604644775390625e-08f);
#endif
}

// Angle of a Box-Muller pair, in revolutions, from one Philox word: the word's upper 23 bits become the mantissa of a
// float in [1, 2) -- ONE instruction, v_alignbit_b32 ({0x7F, w} >> 9 = 0x3F800000 | (w >> 9)) -- and v_sin_f32 /
// v_cos_f32 take revolutions, so the integer part drops out: cos(2*pi*(1 + t)) = cos(2*pi*t), t = (w >> 9) * 2^-23.
// (Shift, convert and scale would be three instructions per angle; the kernel is bound by VALU issue.)
ZS_HD float angle_rev(uint32_t w) {
#if ZS_ON_DEVICE
  return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, w, 9));
#else
  const uint32_t bits = 0x3F800000u | (w >> 9);
  float f;
  memcpy(&f, &bits, 4);
  return f;
#endif
}

// Four standard normals for Philox group `group`: Box-Muller on (x, y) and (z, w) --
//   radius from the uniform u01(x) strictly inside (0, 1) (at most 5.77 sigma), angle 2*pi*t from the upper 23 bits of y.
// (The logarithm is taken of the scaled uniform itself: log2(m + 0.5) - 24 would save nothing after the fma in u01 and
// cancels catastrophically for u close to 1, i.e. for radii close to 0.)
ZS_HD float4 philox_normal4(uint64_t group, const PhiloxCall& pc) {
  Philox4 r = philox4x32_10(group, pc);
  const float u0 = u01(r.x), u2 = u01(r.z);
  const float a1 = angle_rev(r.y), a3 = angle_rev(r.w);
  const float ra = sqrt_fast(-2.0f * ZS_LN2 * log2_fast(u0));
  const float rb = sqrt_fast(-2.0f * ZS_LN2 * log2_fast(u2));
  float4 n;
  n.x = ra * cos_rev(a1);
  n.y = ra * sin_rev(a1);
  n.z = rb * cos_rev(a3);
  n.w = rb * sin_rev(a3);
  return n;
}
ZS_HD float4 philox_normal4(uint64_t group, uint64_t call, uint64_t seed) { return philox_normal4(group, philox_call(call, seed)); }

ZS_HD float f4_get(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

__host__ __forceinline__ bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// grid size for a grid-stride kernel: enough blocks to fill 256 CUs several times over,
// capped so that launch overhead stays flat (cdna_hip_programming.md guideline 11).
__host__ __forceinline__ unsigned grid_for(int64_t work_items, int per_block, unsigned cap = 256u * 16u) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (int64_t)cap) b = cap;
  return (unsigned)b;
}

}  // namespace zs
