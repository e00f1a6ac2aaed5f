// Developer micro-benchmark (not part of the product): variants of the Bernoulli log-prob row-sum
// kernel (K3 forward) timed with kernel-bound HIP events on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/k3_variants.hip -o tools/k3_variants && ./tools/k3_variants
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define LN2 0.69314718055994530942f
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float term(float p, float x) { return x * lg2(p + 1e-8f) + (1.0f - x) * lg2((1.0f - p) + 1e-8f); }
__device__ __forceinline__ float term4(const float4& p, const float4& x) { return term(p.x, x.x) + term(p.y, x.y) + term(p.z, x.z) + term(p.w, x.w); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// V0: one wave per row, grid-stride over rows (the shipped kernel's structure), D4 = 196
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void v0(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                            int K, int B, int D4) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)K * B;
  const long nw = (long)gridDim.x * (BLOCK / 64);
  for (long row = (long)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); row < rows; row += nw) {
    const float4* pr = p + row * D4;
    const float4* xr = x + (row % B) * D4;
    float4 pv[4], xv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int c = lane + 64 * u; ok[u] = c < D4; if (ok[u]) { pv[u] = pr[c]; xv[u] = xr[c]; } }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) if (ok[u]) acc += term4(pv[u], xv[u]);
    acc = wsum(acc);
    const long k = row / B, b = row - k * B;
    if (lane == 0) lp[b * K + k] = acc * LN2;
  }
}

// V1: a wave owns one datapoint b and a chunk of KC particles: x[b,:] lives in registers, p rows stream.
template <int KC>
__global__ __launch_bounds__(256) void v1(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                          int K, int B, int D4) {
  const int lane = threadIdx.x & 63;
  const int kchunks = (K + KC - 1) / KC;
  const long items = (long)B * kchunks;
  const long nw = (long)gridDim.x * 4;
  for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += nw) {
    const int kc = (int)(it / B), b = (int)(it - (long)kc * B);   // adjacent waves -> adjacent b: contiguous p rows
    const float4* xr = x + (long)b * D4;
    float4 xv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int c = lane + 64 * u; ok[u] = c < D4; xv[u] = ok[u] ? xr[c] : make_float4(0, 0, 0, 0); }
    const int k0 = kc * KC, k1 = min(K, k0 + KC);
    for (int k = k0; k < k1; ++k) {
      const float4* pr = p + ((long)k * B + b) * D4;
      float4 pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) pv[u] = pr[lane + 64 * u];
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) acc += term4(pv[u], xv[u]);
      acc = wsum(acc);
      if (lane == 0) lp[(long)b * K + k] = acc * LN2;
    }
  }
}

// V2: like V1 but two particles in flight per iteration (8 independent 16-B loads per lane)
template <int KC>
__global__ __launch_bounds__(256) void v2(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                          int K, int B, int D4) {
  const int lane = threadIdx.x & 63;
  const int kchunks = (K + KC - 1) / KC;
  const long items = (long)B * kchunks;
  const long nw = (long)gridDim.x * 4;
  for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += nw) {
    const int kc = (int)(it / B), b = (int)(it - (long)kc * B);
    const float4* xr = x + (long)b * D4;
    float4 xv[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int c = lane + 64 * u; ok[u] = c < D4; xv[u] = ok[u] ? xr[c] : make_float4(0, 0, 0, 0); }
    const int k0 = kc * KC, k1 = min(K, k0 + KC);
    int k = k0;
    for (; k + 1 < k1; k += 2) {
      const float4* pa = p + ((long)k * B + b) * D4;
      const float4* pb = pa + (long)B * D4;
      float4 va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) { va[u] = pa[lane + 64 * u]; vb[u] = pb[lane + 64 * u]; }
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) { a += term4(va[u], xv[u]); c += term4(vb[u], xv[u]); }
      a = wsum(a); c = wsum(c);
      if (lane == 0) { lp[(long)b * K + k] = a * LN2; lp[(long)b * K + k + 1] = c * LN2; }
    }
    for (; k < k1; ++k) {
      const float4* pr = p + ((long)k * B + b) * D4;
      float4 pv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) pv[u] = pr[lane + 64 * u];
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) acc += term4(pv[u], xv[u]);
      acc = wsum(acc);
      if (lane == 0) lp[(long)b * K + k] = acc * LN2;
    }
  }
}

// V3: 49 lanes x 4 contiguous-stride loads: lane l reads chunks l, l+49, l+98, l+147 (784 B contiguous per instruction)
__global__ __launch_bounds__(256) void v3(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                          int K, int B, int D4) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)K * B;
  const long nw = (long)gridDim.x * 4;
  const int Q = D4 / 4;  // 49
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
    const float4* pr = p + row * D4;
    const float4* xr = x + (row % B) * D4;
    float acc = 0.f;
    if (lane < Q) {
      float4 pv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { pv[u] = pr[lane + Q * u]; xv[u] = xr[lane + Q * u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += term4(pv[u], xv[u]);
    }
    acc = wsum(acc);
    const long k = row / B, b = row - k * B;
    if (lane == 0) lp[b * K + k] = acc * LN2;
  }
}

// V4: a 256-thread block owns one datapoint b and 4 particles per pass: x[b,:] is staged once in LDS and shared by the
// 4 waves (halves the global-load instructions; p rows still stream 16 B per lane)
__global__ __launch_bounds__(256) void v4(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                          int K, int B, int D4) {
  __shared__ float4 xs[256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int kgroups = (K + 3) / 4;
  const long items = (long)B * kgroups;
  for (long it = blockIdx.x; it < items; it += gridDim.x) {
    const int kg = (int)(it / B), b = (int)(it - (long)kg * B);
    __syncthreads();
    if (threadIdx.x < D4) xs[threadIdx.x] = x[(long)b * D4 + threadIdx.x];
    __syncthreads();
    const int k = kg * 4 + w;
    if (k < K) {
      const float4* pr = p + ((long)k * B + b) * D4;
      float4 pv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { int c = lane + 64 * u; ok[u] = c < D4; if (ok[u]) pv[u] = pr[c]; }
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (ok[u]) acc += term4(pv[u], xs[lane + 64 * u]);
      acc = wsum(acc);
      if (lane == 0) lp[(long)b * K + k] = acc * LN2;
    }
  }
}

// V5 (round 3): one wave takes RPW consecutive rows per item, ALL their loads issued up front, clamped instead of
// predicated (lane + 64 * 3 >= D4 reads the row's last chunk again and its term is dropped): 16 (RPW = 2) or 32 (RPW = 4)
// independent 16-byte loads per lane, no branch between them
template <int RPW>
__global__ __launch_bounds__(256) void v5(const float4* __restrict__ p, const float4* __restrict__ x, float* __restrict__ lp,
                                          int K, int B, int D4) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)K * B;
  const long items = (rows + RPW - 1) / RPW;
  const long nw = (long)gridDim.x * 4;
  for (long it = (long)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += nw) {
    float4 pv[RPW][4], xv[RPW][4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ok[u] = lane + 64 * u < D4;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      long row = it * RPW + r;
      if (row >= rows) row = rows - 1;
      const float4* pr = p + row * D4;
      const float4* xr = x + (row % B) * D4;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int c = ok[u] ? lane + 64 * u : D4 - 1; pv[r][u] = pr[c]; xv[r][u] = xr[c]; }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const long row = it * RPW + r;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float t = term4(pv[r][u], xv[r][u]); acc += ok[u] ? t : 0.f; }
      acc = wsum(acc);
      if (lane == 0 && row < rows) { const long k = row / B, b = row - k * B; lp[b * K + k] = acc * LN2; }
    }
  }
}

__global__ __launch_bounds__(256) void vproduce(const float4* __restrict__ q, float4* __restrict__ p, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = q[i]; v.x = 0.5f * v.x + 0.25f; v.y = 0.5f * v.y + 0.25f; v.z = 0.5f * v.z + 0.25f; v.w = 0.5f * v.w + 0.25f; p[i] = v;
  }
}

// copy-rate reference: read p only (float4), trivial sum, same grid-stride structure -> the memory-side ceiling
__global__ __launch_bounds__(256) void vread(const float4* __restrict__ p, float* __restrict__ lp, long n4) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) lp[0] = acc;
}

struct Timer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  template <class F> void launch(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(a, b);
    ev.push_back({a, b});
  }
  void report(const char* name, double bytes) {
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (auto& pr : ev) { float m; CK(hipEventElapsedTime(&m, pr.first, pr.second)); ms.push_back(m); (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    ev.clear();
    std::sort(ms.begin(), ms.end());
    float med = ms[ms.size() / 2], mn = ms[0];
    printf("  %-26s median %8.2f us  min %8.2f us  -> %7.1f GB/s (median)  %7.1f GB/s (min)\n", name, med * 1e3, mn * 1e3,
           bytes / (med * 1e-3) / 1e9, bytes / (mn * 1e-3) / 1e9);
  }
};

int main(int argc, char** argv) {
  const int K = 50, D = 784, D4 = D / 4;
  int Bs[] = {256, 2048, 16384};
  if (argc > 1) { Bs[0] = Bs[1] = Bs[2] = atoi(argv[1]); }
  const int iters = 30;
  for (int bi = 0; bi < (argc > 1 ? 1 : 3); ++bi) {
    const int B = Bs[bi];
    const long rows = (long)K * B, n = rows * D;
    // rotate buffers for the small sizes: 3 (all stay in the 256 MB Infinity Cache) and 10 (400 MB: every launch reads HBM)
    const int NBUF = (n * 4 > (256l << 20)) ? 1 : 10;
    float* p[10]; float *x, *lp;
    for (int i = 0; i < NBUF; ++i) CK(hipMalloc(&p[i], n * 4));
    CK(hipMalloc(&x, (long)B * D * 4)); CK(hipMalloc(&lp, rows * 4));
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = 0.02f + 0.96f * ((i * 2654435761u) % 1000) / 1000.0f;
    for (int i = 0; i < NBUF; ++i) CK(hipMemcpy(p[i], h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> hx((long)B * D);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 40503u >> 3) & 1);
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    const double bytes = 4.0 * n + 4.0 * B * D + 4.0 * rows;
    printf("B=%d K=%d D=%d rows=%ld  p=%.1f MB  (NBUF=%d)\n", B, K, D, rows, n * 4 / 1e6, NBUF);
    Timer t;
    // 0: same buffer every launch (cache-warm), 1: rotate 3 buffers (cache-resident), 2: rotate 10 buffers (HBM),
    // 3: the buffer has just been (re)written by an element-wise producer kernel, as p is by the sigmoid in the step
    for (int mode = 0; mode < 4; ++mode) {
      if (mode >= 1 && NBUF == 1) break;
      const char* mnames[] = {"same-buffer", "rotate-3 (Infinity Cache)", "rotate-10 (HBM)", "freshly written by a producer kernel"};
      printf(" mode=%s\n", mnames[mode]);
#define P (const float4*)p[mode == 0 ? 0 : (mode == 1 ? it % 3 : (mode == 2 ? it % NBUF : 0))]
#define RUN(name, kern, grid, block, ...) \
      for (int it = 0; it < iters; ++it) { \
        if (mode == 3) hipLaunchKernelGGL(vproduce, dim3(2048), dim3(256), 0, 0, (const float4*)p[1], (float4*)p[0], n / 4); \
        t.launch([&](hipEvent_t a, hipEvent_t b) { hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, a, b, 0, __VA_ARGS__); }); } \
      t.report(name, bytes);
      unsigned g0 = (unsigned)std::min<long>((rows + 3) / 4, 4096);
      RUN("read-only float4 (ref)", vread, 2048, 256, P, lp, n / 4);
      RUN("v0 wave/row grid<=4096", (v0<256>), g0, 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v0 wave/row full grid", (v0<256>), (unsigned)((rows + 3) / 4), 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v0 wave/row grid 2048", (v0<256>), 2048, 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v0 block512 grid 1024", (v0<512>), 1024, 512, P, (const float4*)x, lp, K, B, D4);
      RUN("v3 49-lane rows", v3, g0, 256, P, (const float4*)x, lp, K, B, D4);
      { unsigned g = (unsigned)std::min<long>((long)B * ((K + 3) / 4), 4096);
        RUN("v4 x via LDS, block=(b,4k)", v4, g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)((long)B * ((K + 3) / 4));
        RUN("v4 x via LDS, full grid", v4, g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)std::min<long>(((long)B * ((K + 4) / 5) + 3) / 4, 4096);
        RUN("v1 x-in-regs KC=5", (v1<5>), g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)std::min<long>(((long)B * ((K + 9) / 10) + 3) / 4, 4096);
        RUN("v1 x-in-regs KC=10", (v1<10>), g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)std::min<long>(((long)B * ((K + 24) / 25) + 3) / 4, 4096);
        RUN("v1 x-in-regs KC=25", (v1<25>), g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)std::min<long>(((long)B * ((K + 9) / 10) + 3) / 4, 4096);
        RUN("v2 2-in-flight KC=10", (v2<10>), g, 256, P, (const float4*)x, lp, K, B, D4); }
      { unsigned g = (unsigned)std::min<long>(((long)B * ((K + 5) / 6) + 3) / 4, 4096);
        RUN("v2 2-in-flight KC=6", (v2<6>), g, 256, P, (const float4*)x, lp, K, B, D4); }
      RUN("v5 2 rows/wave clamped", (v5<2>), (unsigned)((rows / 2 + 3) / 4), 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v5 2 rows/wave grid 2048", (v5<2>), 2048, 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v5 1 row/wave clamped", (v5<1>), (unsigned)((rows + 3) / 4), 256, P, (const float4*)x, lp, K, B, D4);
      RUN("v5 4 rows/wave clamped", (v5<4>), (unsigned)((rows / 4 + 3) / 4), 256, P, (const float4*)x, lp, K, B, D4);
    }
    for (int i = 0; i < NBUF; ++i) CK(hipFree(p[i]));
    CK(hipFree(x)); CK(hipFree(lp));
  }
  return 0;
}
