// Developer micro-benchmark (not part of the product): RNG choices for the fused Normal sample+log-prob kernel.
#include "../zhusuan-pytorch_amd/csrc/zs_common.h"
#include <stdio.h>
#include <vector>
#include <algorithm>
using namespace zs;
bool zs::prof_begin_launch(int, hipEvent_t*, hipEvent_t*) { return false; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int ROUNDS>
__device__ __forceinline__ Philox4 philox_r(uint64_t group, uint64_t call, uint64_t seed) {
  uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2 = (uint32_t)call, c3 = (uint32_t)(call >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = xor3_key((uint32_t)(p1 >> 32), c1, k0), n1 = (uint32_t)p1, n2 = xor3_key((uint32_t)(p0 >> 32), c3, k1), n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  Philox4 o = {c0, c1, c2, c3};
  return o;
}
__device__ __forceinline__ uint32_t rotl(uint32_t x, int k) { return __builtin_amdgcn_alignbit(x, x, 32 - k); }
struct Xo { uint32_t s0, s1, s2, s3; };
__device__ __forceinline__ uint32_t xo_next(Xo& s) {
  const uint32_t r = rotl(s.s0 + s.s3, 7) + s.s0;
  const uint32_t t = s.s1 << 9;
  s.s2 ^= s.s0; s.s3 ^= s.s1; s.s1 ^= s.s2; s.s0 ^= s.s3; s.s2 ^= t; s.s3 = rotl(s.s3, 11);
  return r;
}
__device__ __forceinline__ float4 bm4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  float u0 = u01(a), u1 = u01(b), u2 = u01(c), u3 = u01(d);
  float ra = __builtin_amdgcn_sqrtf(-2.0f * ZS_LN2 * log2_fast(u0)), rb = __builtin_amdgcn_sqrtf(-2.0f * ZS_LN2 * log2_fast(u2));
  return make_float4(ra * __builtin_amdgcn_cosf(u1), ra * __builtin_amdgcn_sinf(u1), rb * __builtin_amdgcn_cosf(u3), rb * __builtin_amdgcn_sinf(u3));
}

// MODE 0: philox10 per float4, 1: philox7, 2: xoshiro128++ seeded by one philox10 per (lane, k-chunk), 3: no rng
// RED 0: shfl_down loop (product), 1: no cross-lane reduction (cost probe; wrong sums)
template <int MODE, int RED, int NT = 0>
__global__ __launch_bounds__(256) void k1(const float4* __restrict__ mu, const float4* __restrict__ sigma, uint64_t seed, uint64_t call,
                                          float4* __restrict__ z, float* __restrict__ lp, int64_t K, int64_t R, int D4, int G, int rpw, int p2,
                                          int64_t kchunk, int64_t sk, int64_t sr) {
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw, k_tiles = (K + kchunk - 1) / kchunk, total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
    const int64_t rt = t % row_tiles, kt = t / row_tiles;
    const int64_t r = rt * rpw + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = r * D4 + lig;
    float4 m = make_float4(0, 0, 0, 0), s = make_float4(1, 1, 1, 1);
    if (on) { m = mu[m4]; s = sigma[m4]; }
    float rowc = 0.f, hp[4];
    { const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float l2 = log2_fast(sv[j]); rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2; hp[j] = 0.5f * exp2_fast(-2.0f * l2); } }
    const int64_t k0 = kt * kchunk, k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;
    float* __restrict__ lpp = lp + (k0 * sk + r * sr);
    Xo st;
    if (MODE == 2) { Philox4 p = philox4x32_10((uint64_t)m4, call + (uint64_t)kt, seed); st.s0 = p.x; st.s1 = p.y; st.s2 = p.z; st.s3 = p.w | 1u; }
    for (int64_t k = k0; k < k1; ++k, g += M4) {
      float4 e;
      if (MODE == 0) { Philox4 p = philox_r<10>((uint64_t)g, call, seed); e = bm4(p.x, p.y, p.z, p.w); }
      else if (MODE == 1) { Philox4 p = philox_r<7>((uint64_t)g, call, seed); e = bm4(p.x, p.y, p.z, p.w); }
      else if (MODE == 2) { uint32_t a = xo_next(st), b = xo_next(st), c = xo_next(st), d = xo_next(st); e = bm4(a, b, c, d); }
      else { e = make_float4(0.1f * lane, 0.2f, -0.3f, 0.4f + (float)k); }
      float4 zz;
      zz.x = m.x + s.x * e.x; zz.y = m.y + s.y * e.y; zz.z = m.z + s.z * e.z; zz.w = m.w + s.w * e.w;
      if (on) { if (NT) { typedef float f4v __attribute__((ext_vector_type(4))); f4v v = {zz.x, zz.y, zz.z, zz.w}; __builtin_nontemporal_store(v, (f4v*)&z[g]); } else z[g] = zz; }
      const float d0 = zz.x - m.x, d1 = zz.y - m.y, d2 = zz.z - m.z, d3 = zz.w - m.w;
      float acc = rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3));
      if (RED == 0) acc = group_sum_down(acc, lig, G, p2);
      if (on && lig == 0) *lpp = acc;
      lpp += sk;
    }
  }
}

// W2: each lane owns TWO adjacent float4 of a row (32 contiguous bytes): D = 40 -> 5 lanes per row, 12 rows per
// wave, 3-step reduction amortised over 8 elements.
template <int MODE>
__global__ __launch_bounds__(256) void k1w2(const float4* __restrict__ mu, const float4* __restrict__ sigma, uint64_t seed, uint64_t call,
                                            float4* __restrict__ z, float* __restrict__ lp, int64_t K, int64_t R, int D4, int G, int rpw, int p2,
                                            int64_t kchunk, int64_t sk, int64_t sr) {
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw, k_tiles = (K + kchunk - 1) / kchunk, total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
    const int64_t rt = t % row_tiles, kt = t / row_tiles;
    const int64_t r = rt * rpw + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = r * D4 + 2 * lig;
    float4 m[2], s[2];
    float rowc = 0.f, hp[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      m[h] = make_float4(0, 0, 0, 0); s[h] = make_float4(1, 1, 1, 1);
      if (on) { m[h] = mu[m4 + h]; s[h] = sigma[m4 + h]; }
      const float sv[4] = {s[h].x, s[h].y, s[h].z, s[h].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float l2 = log2_fast(sv[j]); rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2; hp[4 * h + j] = 0.5f * exp2_fast(-2.0f * l2); }
    }
    const int64_t k0 = kt * kchunk, k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;
    float* __restrict__ lpp = lp + (k0 * sk + r * sr);
    for (int64_t k = k0; k < k1; ++k, g += M4) {
      float acc = rowc;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 e;
        if (MODE == 0) { Philox4 p = philox_r<10>((uint64_t)(g + h), call, seed); e = bm4(p.x, p.y, p.z, p.w); }
        else if (MODE == 1) { Philox4 p = philox_r<7>((uint64_t)(g + h), call, seed); e = bm4(p.x, p.y, p.z, p.w); }
        else { e = make_float4(0.1f * lane, 0.2f, -0.3f, 0.4f + (float)k); }
        float4 zz;
        zz.x = m[h].x + s[h].x * e.x; zz.y = m[h].y + s[h].y * e.y; zz.z = m[h].z + s[h].z * e.z; zz.w = m[h].w + s[h].w * e.w;
        if (on) z[g + h] = zz;
        const float d0 = zz.x - m[h].x, d1 = zz.y - m[h].y, d2 = zz.z - m[h].z, d3 = zz.w - m[h].w;
        acc -= hp[4 * h] * (d0 * d0) + hp[4 * h + 1] * (d1 * d1) + hp[4 * h + 2] * (d2 * d2) + hp[4 * h + 3] * (d3 * d3);
      }
      acc = group_sum_down(acc, lig, G, p2);
      if (on && lig == 0) *lpp = acc;
      lpp += sk;
    }
  }
}

// LDS-staged row sums: every lane parks its partial in LDS each particle; after KB particles the wave reads them
// back transposed (lane = (row, particle)), adds the G partials of a row and writes log q coalesced along K.
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k1lds(const float4* __restrict__ mu, const float4* __restrict__ sigma, uint64_t seed, uint64_t call,
                                             float4* __restrict__ z, float* __restrict__ lp, int64_t K, int64_t R, int D4, int G, int rpw, int p2,
                                             int64_t kchunk, int64_t sk, int64_t sr) {
  constexpr int KB = 16, LDW = 65;
  __shared__ float stage[4][KB * LDW];
  float* __restrict__ st = stage[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw, k_tiles = (K + kchunk - 1) / kchunk, total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
    const int64_t rt = t % row_tiles, kt = t / row_tiles;
    const int64_t rbase = rt * rpw;
    const int64_t r = rbase + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = r * D4 + lig;
    float4 m = make_float4(0, 0, 0, 0), s = make_float4(1, 1, 1, 1);
    if (on) { m = mu[m4]; s = sigma[m4]; }
    float rowc = 0.f, hp[4];
    { const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float l2 = log2_fast(sv[j]); rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2; hp[j] = 0.5f * exp2_fast(-2.0f * l2); } }
    const int64_t k0 = kt * kchunk, k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;
    for (int64_t kb0 = k0; kb0 < k1; kb0 += KB) {
      const int kb = (int)((k1 - kb0 < KB) ? (k1 - kb0) : KB);
      for (int kk = 0; kk < kb; ++kk, g += M4) {
        float4 e;
        if (MODE == 0 || MODE == 4) { Philox4 p = philox_r<10>((uint64_t)g, call, seed); e = bm4(p.x, p.y, p.z, p.w); }
        else if (MODE == 5) { Philox4 p = philox_r<7>((uint64_t)g, call, seed); e = bm4(p.x, p.y, p.z, p.w); }
        else e = make_float4(0.1f * lane, 0.2f, -0.3f, 0.4f + (float)kk);
        float4 zz;
        zz.x = m.x + s.x * e.x; zz.y = m.y + s.y * e.y; zz.z = m.z + s.z * e.z; zz.w = m.w + s.w * e.w;
        if (on && (MODE != 4 || zz.x == 12345.678f)) { if (NT) { typedef float f4v __attribute__((ext_vector_type(4))); f4v v = {zz.x, zz.y, zz.z, zz.w}; __builtin_nontemporal_store(v, (f4v*)&z[g]); } else z[g] = zz; }
        const float d0 = zz.x - m.x, d1 = zz.y - m.y, d2 = zz.z - m.z, d3 = zz.w - m.w;
        st[kk * LDW + lane] = rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3));
      }
      const int nout = rpw * kb;
      for (int o = lane; o < nout; o += 64) {
        const int q = o / kb, kk = o - q * kb;
        float sum = 0.f;
        for (int j = 0; j < G; ++j) sum += st[kk * LDW + q * G + j];
        if (rbase + q < R) lp[(kb0 + kk) * sk + (rbase + q) * sr] = sum;
      }
    }
  }
}

// tuned: compile-time G, KB particles per LDS batch, two particles per loop iteration
template <int G, int KB, int NT>
__global__ __launch_bounds__(256) void k1tuned(const float4* __restrict__ mu, const float4* __restrict__ sigma, uint64_t seed, uint64_t call,
                                               float4* __restrict__ z, float* __restrict__ lp, int64_t K, int64_t R, int64_t kchunk, int64_t sk, int64_t sr) {
  constexpr int D4 = G, rpw = 64 / G, LDW = 65;
  __shared__ float stage[4][KB * LDW];
  float* __restrict__ st = stage[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw, k_tiles = (K + kchunk - 1) / kchunk, total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
    const int64_t rt = t % row_tiles, kt = t / row_tiles;
    const int64_t rbase = rt * rpw;
    const int64_t r = rbase + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = r * D4 + lig;
    float4 m = make_float4(0, 0, 0, 0), s = make_float4(1, 1, 1, 1);
    if (on) { m = mu[m4]; s = sigma[m4]; }
    float rowc = 0.f, hp[4];
    { const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float l2 = log2_fast(sv[j]); rowc += ZS_NEG_HALF_LOG_2PI - l2 * ZS_LN2; hp[j] = 0.5f * exp2_fast(-2.0f * l2); } }
    const int64_t k0 = kt * kchunk, k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    int64_t g = k0 * M4 + m4;
    for (int64_t kb0 = k0; kb0 < k1; kb0 += KB) {
      const int kb = (int)((k1 - kb0 < KB) ? (k1 - kb0) : KB);
      int kk = 0;
      for (; kk + 1 < kb; kk += 2, g += 2 * M4) {
        Philox4 pa = philox_r<10>((uint64_t)g, call, seed), pb = philox_r<10>((uint64_t)(g + M4), call, seed);
        const float4 ea = bm4(pa.x, pa.y, pa.z, pa.w), eb = bm4(pb.x, pb.y, pb.z, pb.w);
        float4 za, zb;
        za.x = m.x + s.x * ea.x; za.y = m.y + s.y * ea.y; za.z = m.z + s.z * ea.z; za.w = m.w + s.w * ea.w;
        zb.x = m.x + s.x * eb.x; zb.y = m.y + s.y * eb.y; zb.z = m.z + s.z * eb.z; zb.w = m.w + s.w * eb.w;
        if (on) {
          if (NT) { typedef float f4v __attribute__((ext_vector_type(4))); f4v va = {za.x, za.y, za.z, za.w}, vb = {zb.x, zb.y, zb.z, zb.w};
            __builtin_nontemporal_store(va, (f4v*)&z[g]); __builtin_nontemporal_store(vb, (f4v*)&z[g + M4]); }
          else { z[g] = za; z[g + M4] = zb; }
        }
        { const float d0 = za.x - m.x, d1 = za.y - m.y, d2 = za.z - m.z, d3 = za.w - m.w;
          st[kk * LDW + lane] = rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3)); }
        { const float d0 = zb.x - m.x, d1 = zb.y - m.y, d2 = zb.z - m.z, d3 = zb.w - m.w;
          st[(kk + 1) * LDW + lane] = rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3)); }
      }
      for (; kk < kb; ++kk, g += M4) {
        Philox4 pa = philox_r<10>((uint64_t)g, call, seed);
        const float4 ea = bm4(pa.x, pa.y, pa.z, pa.w);
        float4 za;
        za.x = m.x + s.x * ea.x; za.y = m.y + s.y * ea.y; za.z = m.z + s.z * ea.z; za.w = m.w + s.w * ea.w;
        if (on) z[g] = za;
        const float d0 = za.x - m.x, d1 = za.y - m.y, d2 = za.z - m.z, d3 = za.w - m.w;
        st[kk * LDW + lane] = rowc - (hp[0] * (d0 * d0) + hp[1] * (d1 * d1) + hp[2] * (d2 * d2) + hp[3] * (d3 * d3));
      }
      const int nout = rpw * kb;
      for (int o = lane; o < nout; o += 64) {
        const int q = o / kb, k2 = o - q * kb;
        const float* __restrict__ src = st + k2 * LDW + q * G;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < G; ++j) sum += src[j];
        if (rbase + q < R) lp[(kb0 + k2) * sk + (rbase + q) * sr] = sum;
      }
    }
  }
}

int main() {
  const int K = 50, D = 40, D4 = 10, G = 10, rpw = 6, p2 = 16;
  int64_t Bs[] = {20971, 83886};
  for (int bi = 0; bi < 2; ++bi) {
    const int64_t B = Bs[bi], M = B * D, N = K * B;
    float *mu, *sg, *z, *lp;
    CK(hipMalloc(&mu, M * 4)); CK(hipMalloc(&sg, M * 4)); CK(hipMalloc(&z, K * M * 4)); CK(hipMalloc(&lp, N * 4));
    std::vector<float> h(M, 0.5f);
    CK(hipMemcpy(mu, h.data(), M * 4, hipMemcpyHostToDevice));
    for (auto& v : h) v = 1.25f;
    CK(hipMemcpy(sg, h.data(), M * 4, hipMemcpyHostToDevice));
    const int64_t row_tiles = (B + rpw - 1) / rpw;
    int64_t kt = (2048 + row_tiles - 1) / row_tiles; if (kt < 1) kt = 1; if (kt > K) kt = K;
    const int64_t kchunk = (K + kt - 1) / kt, total = row_tiles * ((K + kchunk - 1) / kchunk);
    const unsigned grid = (unsigned)std::min<int64_t>((total + 3) / 4, 4096);
    const double bytes = 4.0 * N * D + 4.0 * N + 8.0 * M;
    printf("B=%ld N=%ld  %.1f MB  kchunk=%ld grid=%u\n", (long)B, (long)N, bytes / 1e6, (long)kchunk, grid);
#define RUN(name, MODE, RED) RUNX(name, MODE, RED, 0)
#define RUNX(name, MODE, RED, NT) { std::vector<float> ms; for (int it = 0; it < 12; ++it) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); \
      hipExtLaunchKernelGGL((k1<MODE, RED, NT>), dim3(grid), dim3(256), 0, 0, a, b, 0, (const float4*)mu, (const float4*)sg, (uint64_t)1, (uint64_t)it, (float4*)z, lp, (int64_t)K, B, D4, G, rpw, p2, kchunk, (int64_t)1, (int64_t)K); \
      CK(hipDeviceSynchronize()); float m_; CK(hipEventElapsedTime(&m_, a, b)); ms.push_back(m_); } std::sort(ms.begin(), ms.end()); \
      printf("  %-34s median %8.2f us  -> %7.1f GB/s (%4.1f%% of 8 TB/s)\n", name, ms[6] * 1e3, bytes / (ms[6] * 1e-3) / 1e9, bytes / (ms[6] * 1e-3) / 1e9 / 80.0); }
    RUN("philox10 + shfl reduce (product)", 0, 0);
    RUN("philox7  + shfl reduce", 1, 0);
    RUN("xoshiro128++ substreams + shfl", 2, 0);
    RUN("no rng + shfl reduce", 3, 0);
    RUN("philox10, no reduce (probe)", 0, 1);
    RUN("xoshiro, no reduce (probe)", 2, 1);
    RUN("no rng, no reduce (store floor)", 3, 1);
    RUNX("no rng, no reduce, nt stores", 3, 1, 1);
    RUNX("philox10 + shfl, nt stores", 0, 0, 1);
    RUNX("philox7 + shfl, nt stores", 1, 0, 1);
#define RUNL(name, MODE, NT) { std::vector<float> ms; for (int it = 0; it < 12; ++it) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); \
      hipExtLaunchKernelGGL((k1lds<MODE, NT>), dim3(grid), dim3(256), 0, 0, a, b, 0, (const float4*)mu, (const float4*)sg, (uint64_t)1, (uint64_t)it, (float4*)z, lp, (int64_t)K, B, D4, G, rpw, p2, kchunk, (int64_t)1, (int64_t)K); \
      CK(hipDeviceSynchronize()); float m_; CK(hipEventElapsedTime(&m_, a, b)); ms.push_back(m_); } std::sort(ms.begin(), ms.end()); \
      printf("  %-34s median %8.2f us  -> %7.1f GB/s (%4.1f%% of 8 TB/s)\n", name, ms[6] * 1e3, bytes / (ms[6] * 1e-3) / 1e9, bytes / (ms[6] * 1e-3) / 1e9 / 80.0); }
#define RUNT(name, G_, KB_, NT) { std::vector<float> ms; for (int it = 0; it < 12; ++it) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); \
      hipExtLaunchKernelGGL((k1tuned<G_, KB_, NT>), dim3(grid), dim3(256), 0, 0, a, b, 0, (const float4*)mu, (const float4*)sg, (uint64_t)1, (uint64_t)it, (float4*)z, lp, (int64_t)K, B, kchunk, (int64_t)1, (int64_t)K); \
      CK(hipDeviceSynchronize()); float m_; CK(hipEventElapsedTime(&m_, a, b)); ms.push_back(m_); } std::sort(ms.begin(), ms.end()); \
      printf("  %-34s median %8.2f us  -> %7.1f GB/s (%4.1f%% of 8 TB/s)\n", name, ms[6] * 1e3, bytes / (ms[6] * 1e-3) / 1e9, bytes / (ms[6] * 1e-3) / 1e9 / 80.0); }
    RUNT("tuned G=10 KB=16 2/iter", 10, 16, 0);
    RUNT("tuned G=10 KB=16 2/iter nt", 10, 16, 1);
    RUNT("tuned G=10 KB=32 2/iter nt", 10, 32, 1);
    RUNL("philox10 + LDS-staged sums", 0, 0);
    RUNL("philox10 + LDS-staged sums, nt", 0, 1);
    RUNL("no rng + LDS-staged sums, nt", 3, 1);
    RUNL("philox10 + LDS sums, NO z stores (VALU only)", 4, 1);
    RUNL("philox7 + LDS-staged sums, nt", 5, 1);
    RUNX("xoshiro + shfl, nt stores", 2, 0, 1);
    RUNX("xoshiro, no reduce, nt stores", 2, 1, 1);
    RUNX("philox10, no reduce, nt stores", 0, 1, 1);
    { std::vector<float> ms; for (int it = 0; it < 8; ++it) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventRecord(a, 0));
        CK(hipMemsetAsync(z, 0, (size_t)K * M * 4, 0)); CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize()); float m_; CK(hipEventElapsedTime(&m_, a, b)); ms.push_back(m_); }
      std::sort(ms.begin(), ms.end()); printf("  %-34s median %8.2f us  -> %7.1f GB/s\n", "hipMemsetAsync of z (event pair)", ms[4] * 1e3, 4.0 * K * M / (ms[4] * 1e-3) / 1e9); }
    { const int G2 = 5, rpw2 = 12, p22 = 8;
      const int64_t rt2 = (B + rpw2 - 1) / rpw2; int64_t kt2 = (2048 + rt2 - 1) / rt2; if (kt2 < 1) kt2 = 1; if (kt2 > K) kt2 = K;
      const int64_t kc2 = (K + kt2 - 1) / kt2, tot2 = rt2 * ((K + kc2 - 1) / kc2);
      const unsigned grid2 = (unsigned)std::min<int64_t>((tot2 + 3) / 4, 4096);
#define RUN2(name, MODE) { std::vector<float> ms; for (int it = 0; it < 12; ++it) { hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); \
      hipExtLaunchKernelGGL((k1w2<MODE>), dim3(grid2), dim3(256), 0, 0, a, b, 0, (const float4*)mu, (const float4*)sg, (uint64_t)1, (uint64_t)it, (float4*)z, lp, (int64_t)K, B, D4, G2, rpw2, p22, kc2, (int64_t)1, (int64_t)K); \
      CK(hipDeviceSynchronize()); float m_; CK(hipEventElapsedTime(&m_, a, b)); ms.push_back(m_); } std::sort(ms.begin(), ms.end()); \
      printf("  %-34s median %8.2f us  -> %7.1f GB/s (%4.1f%% of 8 TB/s)\n", name, ms[6] * 1e3, bytes / (ms[6] * 1e-3) / 1e9, bytes / (ms[6] * 1e-3) / 1e9 / 80.0); }
      RUN2("2xfloat4/lane philox10", 0);
      RUN2("2xfloat4/lane philox7", 1);
      RUN2("2xfloat4/lane no rng", 2);
    }
    CK(hipFree(mu)); CK(hipFree(sg)); CK(hipFree(z)); CK(hipFree(lp));
  }
  return 0;
}
