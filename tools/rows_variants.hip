// Microbenchmark: where does the generic row-sum ("wave tile") kernel of zs_locscale.hip lose time?
// Workload: U2-like, x [K, R, D] fp32 -> lp [R, K] (K-fastest), D = 40, K = 50.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rows_variants.hip -o tools/rows_variants && ./tools/rows_variants
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kTile = 1024, kTileRows = 256, kTileLds = kTile + kTile / 32 + kTileRows;
__device__ __forceinline__ int pad_idx(int e) { return e + (e >> 5); }

// STAGE 0: loads + arithmetic only; 1: + LDS term writes; 2: + phase-2 sums; 3: + global writes (full kernel)
// PART: park one partial per 4-element group instead of 4 terms (needs D % 4 == 0)
template <int STAGE, bool PART>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ x, const float* __restrict__ lo, const float* __restrict__ hi,
                                              float* __restrict__ lp, int64_t K, int64_t R, int D, int rr, int kk, int lgG,
                                              int64_t sk, int64_t sr) {
  __shared__ float lds[4][kTileLds];
  float* __restrict__ term = lds[threadIdx.x >> 6];
  float* __restrict__ res = term + (kTile + kTile / 32);
  const int lane = threadIdx.x & 63, G = 1 << lgG;
  const int seg_len = rr * D;
  const int64_t RD = R * (int64_t)D;
  const int64_t rtiles = (R + rr - 1) / rr, ktiles = (K + kk - 1) / kk, tiles = rtiles * ktiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  int segj[4], offj[4], ldsj[4];
  int64_t soj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = lane * 4 + 256 * j;
    segj[j] = e / seg_len; offj[j] = e - segj[j] * seg_len; soj[j] = segj[j] * RD + offj[j];
    ldsj[j] = PART ? (e >> 2) + (e >> 7) : pad_idx(e);
  }
  float sink = 0.f;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < tiles; t += nwaves) {
    const int64_t kt = t / rtiles, rt = t - kt * rtiles;
    const int64_t r0 = rt * rr, k0 = kt * kk;
    const int nr = (int)((R - r0 < rr) ? (R - r0) : rr), nk = (int)((K - k0 < kk) ? (K - k0) : kk);
    const int seg_valid = nr * D;
    const int64_t im0 = r0 * D, base = k0 * RD + im0;
    float4 xv[4], lv[4], hv[4];
    bool on[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      on[j] = segj[j] < nk && offj[j] < seg_valid;
      // unconditional loads from a clamped (always valid) address: a divergent guard around the loads makes the
      // compiler wait for each group before issuing the next one
      const int64_t so = on[j] ? soj[j] : 0;
      const int of = on[j] ? offj[j] : 0;
      xv[j] = *reinterpret_cast<const float4*>(x + base + so);
      lv[j] = *reinterpret_cast<const float4*>(lo + im0 + of);
      hv[j] = *reinterpret_cast<const float4*>(hi + im0 + of);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!on[j]) continue;
      float tv[4];
      const float xs[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, ls[4] = {lv[j].x, lv[j].y, lv[j].z, lv[j].w},
                  hs[4] = {hv[j].x, hv[j].y, hv[j].z, hv[j].w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        tv[q] = ((ls[q] <= xs[q] && hs[q] > xs[q]) ? 0.f : -INFINITY) - __builtin_amdgcn_logf(hs[q] - ls[q]) * 0.6931472f;
      if (STAGE == 0) { sink += (tv[0] + tv[1]) + (tv[2] + tv[3]); continue; }
      if (PART) term[ldsj[j]] = (tv[0] + tv[1]) + (tv[2] + tv[3]);
      else {
#pragma unroll
        for (int q = 0; q < 4; ++q) term[ldsj[j] + q] = tv[q];
      }
    }
    if (STAGE >= 2) {
      __builtin_amdgcn_wave_barrier();
      const int nrows = nk * nr, g = lane & (G - 1);
      const int DD = PART ? D / 4 : D, SL = PART ? seg_len / 4 : seg_len;
      for (int w = lane >> lgG; w < nrows; w += 64 >> lgG) {
        const int rrow = w / nk, seg = w - rrow * nk;
        const int b0 = seg * SL + rrow * DD;
        float acc = 0.f;
        for (int d = g; d < DD; d += G) acc += term[PART ? (b0 + d) + ((b0 + d) >> 5) : pad_idx(b0 + d)];
        for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (g == 0) res[w] = acc;
      }
      __builtin_amdgcn_wave_barrier();
      if (STAGE >= 3) {
        for (int o = lane; o < nrows; o += 64) {
          const int rrow = o / nk, seg = o - rrow * nk;
          lp[(k0 + seg) * sk + (r0 + rrow) * sr] = res[o];
        }
      } else {
        sink += res[lane];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (STAGE < 3 && sink == 12345.678f) lp[0] = sink;
}

// reference: K2-style mapping, G = D/4 lanes per row, 64/G rows per wave pass, shuffle reduction, direct write
__global__ __launch_bounds__(256) void k_lanes(const float4* __restrict__ x, const float4* __restrict__ lo, const float4* __restrict__ hi,
                                               float* __restrict__ lp, int64_t K, int64_t R, int D4, int64_t sk, int64_t sr) {
  const int lane = threadIdx.x & 63;
  const int G = D4, rpw = 64 / G;
  const int rw = lane / G, lig = lane - rw * G;
  int p2 = 1; while (p2 < G) p2 <<= 1;
  const int64_t rows = K * R, passes = (rows + rpw - 1) / rpw, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < passes; t += nwaves) {
    const int64_t row = t * rpw + rw;
    const bool on = rw < rpw && row < rows;
    float acc = 0.f;
    int64_t k = 0, r = 0;
    if (on) {
      k = row / R; r = row - k * R;
      const float4 xv = x[row * D4 + lig], lv = lo[r * D4 + lig], hv = hi[r * D4 + lig];
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ls[4] = {lv.x, lv.y, lv.z, lv.w}, hs[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        acc += ((ls[q] <= xs[q] && hs[q] > xs[q]) ? 0.f : -INFINITY) - __builtin_amdgcn_logf(hs[q] - ls[q]) * 0.6931472f;
    }
    for (int o = p2 >> 1; o > 0; o >>= 1) { float v = __shfl_down(acc, o, 64); if (lig + o < G) acc += v; }
    if (on && lig == 0) lp[k * sk + r * sr] = acc;
  }
}


// K2-krep style with U value rows in flight and LDS-staged row sums (the K1 tiling): a wave owns rpw = 64/G parameter
// rows (G = D/4 lanes each) and walks the K particles U at a time; each lane parks its 4-element partial per particle,
// every KB particles the wave reads them back transposed and writes the K-fastest result coalesced.
#define KB 32
#define LDW 65
template <int U>
__global__ __launch_bounds__(256) void k_krep(const float4* __restrict__ x, const float4* __restrict__ lo, const float4* __restrict__ hi,
                                              float* __restrict__ lp, int64_t K, int64_t R, int D4, int64_t kchunk, int64_t sk, int64_t sr) {
  __shared__ float stage[4][KB * LDW];
  float* __restrict__ st = stage[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const int G = D4, rpw = 64 / G;
  const int rw = lane / G, lig = lane - rw * G;
  const bool lane_on = rw < rpw;
  const int64_t M4 = R * (int64_t)D4;
  const int64_t row_tiles = (R + rpw - 1) / rpw, k_tiles = (K + kchunk - 1) / kchunk, total = row_tiles * k_tiles;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
    const int64_t kt = t / row_tiles, rt = t - kt * row_tiles;
    const int64_t rbase = rt * rpw, r = rbase + rw;
    const bool on = lane_on && r < R;
    const int64_t m4 = on ? r * D4 + lig : 0;
    const float4 lv = lo[m4], hv = hi[m4];
    const float ls[4] = {lv.x, lv.y, lv.z, lv.w}, hs[4] = {hv.x, hv.y, hv.z, hv.w};
    float c = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) c -= __builtin_amdgcn_logf(hs[q] - ls[q]) * 0.6931472f;
    const int64_t k0 = kt * kchunk, k1 = (k0 + kchunk < K) ? k0 + kchunk : K;
    for (int64_t kb0 = k0; kb0 < k1; kb0 += KB) {
      const int kb = (int)((k1 - kb0 < KB) ? (k1 - kb0) : KB);
      for (int kk = 0; kk < kb; kk += U) {
        float4 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int64_t k = kb0 + kk + (kk + u < kb ? u : 0);      // clamped: unconditional loads
          xv[u] = x[k * M4 + m4];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (kk + u < kb) {
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            float a = c;
#pragma unroll
            for (int q = 0; q < 4; ++q) a += (ls[q] <= xs[q] && hs[q] > xs[q]) ? 0.f : -INFINITY;
            st[(kk + u) * LDW + lane] = a;
          }
        }
      }
      const int nout = rpw * kb;
      for (int o = lane; o < nout; o += 64) {
        const int q = o / kb, kk = o - q * kb;
        const float* __restrict__ src = st + kk * LDW + q * G;
        float sum = 0.f;
        for (int j = 0; j < G; ++j) sum += src[j];
        if (rbase + q < R) lp[(kb0 + kk) * sk + (rbase + q) * sr] = sum;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ x, float* __restrict__ out, int64_t n4) {
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = x[i]; s += (v.x + v.y) + (v.z + v.w);
  }
  if (s == 12345.678f) out[0] = s;
}

template <typename L>
static void timeit(const char* name, double bytes, L launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int i = 0; i < 15; ++i) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m);
  }
  std::sort(ms.begin(), ms.end());
  printf("  %-44s median %9.2f us -> %7.1f GB/s (%4.1f%%)\n", name, ms[7] * 1e3, bytes / (ms[7] * 1e-3) / 1e9, bytes / (ms[7] * 1e-3) / 8e10);
}

int main() {
  const int64_t K = 50;
  const int D = 40;
  for (int64_t B : {2621, 20971, 83886}) {
    const int64_t R = B, N = K * R, n = N * D;
    float *x, *lo, *hi, *lp;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&lo, R * D * 4)); CK(hipMalloc(&hi, R * D * 4)); CK(hipMalloc(&lp, N * 4));
    CK(hipMemset(x, 0, n * 4)); CK(hipMemset(lo, 0, R * D * 4));
    std::vector<float> ones(R * D, 1.0f);
    CK(hipMemcpy(hi, ones.data(), R * D * 4, hipMemcpyHostToDevice));
    const double bytes = 4.0 * n + 4.0 * N + 8.0 * R * D;
    printf("B=%ld rows=%ld  x=%.1f MB\n", (long)B, (long)N, n * 4 / 1e6);
    timeit("read-only float4", 4.0 * n, [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const float4*)x, lp, n / 4); });
    timeit("K2-style lanes-per-row (no LDS), kfast out", bytes, [&] {
      hipLaunchKernelGGL(k_lanes, dim3(4096), dim3(256), 0, 0, (const float4*)x, (const float4*)lo, (const float4*)hi, lp, K, R, D / 4, (int64_t)1, K); });
    {
      const int rpw = 64 / (D / 4);
      const int64_t row_tiles = (R + rpw - 1) / rpw;
      for (int64_t kchunk : {50, 25}) {
        const int64_t total = row_tiles * ((K + kchunk - 1) / kchunk);
        const unsigned grid = (unsigned)std::min<int64_t>((total + 3) / 4, 4096);
        char nm[128];
        snprintf(nm, sizeof nm, "krep U=2 LDS-staged, kchunk=%ld", (long)kchunk);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_krep<2>), dim3(grid), dim3(256), 0, 0, (const float4*)x, (const float4*)lo, (const float4*)hi, lp, K, R, D / 4, kchunk, (int64_t)1, K); });
        snprintf(nm, sizeof nm, "krep U=4 LDS-staged, kchunk=%ld", (long)kchunk);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_krep<4>), dim3(grid), dim3(256), 0, 0, (const float4*)x, (const float4*)lo, (const float4*)hi, lp, K, R, D / 4, kchunk, (int64_t)1, K); });
        snprintf(nm, sizeof nm, "krep U=8 LDS-staged, kchunk=%ld", (long)kchunk);
        timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_krep<8>), dim3(grid), dim3(256), 0, 0, (const float4*)x, (const float4*)lo, (const float4*)hi, lp, K, R, D / 4, kchunk, (int64_t)1, K); });
      }
    }
    struct Geo { int rr, kk, lgG; const char* name; };
    const Geo geos[] = {{1, 25, 1, "tile rr=1 kk=25"}, {25, 1, 1, "tile rr=25 kk=1 (row-major reads)"}, {4, 6, 1, "tile rr=4 kk=6"}};
    for (const Geo& g : geos) {
      const int64_t tiles = ((R + g.rr - 1) / g.rr) * ((K + g.kk - 1) / g.kk);
      const unsigned grid = (unsigned)std::min<int64_t>((tiles + 3) / 4, 4096);
      char nm[128];
#define RUN(ST, PT)                                                                                                      \
  snprintf(nm, sizeof nm, "%s stage%d%s", g.name, ST, PT ? " partials" : "");                                              \
  timeit(nm, bytes, [&] { hipLaunchKernelGGL((k_rows<ST, PT>), dim3(grid), dim3(256), 0, 0, x, lo, hi, lp, K, R, D, g.rr, g.kk, g.lgG, (int64_t)1, K); });
      RUN(0, false) RUN(1, false) RUN(2, false) RUN(3, false) RUN(1, true) RUN(3, true)
    }
    CK(hipFree(x)); CK(hipFree(lo)); CK(hipFree(hi)); CK(hipFree(lp));
  }
  return 0;
}
