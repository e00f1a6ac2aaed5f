// Issue cost of the VALU instructions the Philox4x32-10 + Box-Muller generator is made of, on gfx950:
// cycles per wave64 instruction per SIMD at 1, 2, 4 and 8 waves per SIMD (s_memtime around an unrolled stream of
// independent instructions).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_rates tools/valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP 64      // instructions per asm block (8 independent chains x 8)
#define ITERS 200

#define CHAIN8(INS)                                                                                     \
  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define BLOCK64(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS) CHAIN8(INS)

template <int KIND>
__global__ __launch_bounds__(512) void k(uint64_t* out, uint32_t seed) {
  uint32_t a[8], b[8];
  float f[8];
  uint64_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7 + i; b[i] = seed * 3 + i; f[i] = 1.0f + 0.001f * (threadIdx.x + i); w[i] = a[i]; }
  uint64_t t0 = 0, t1 = 0, r0 = 0, r1 = 0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITERS; ++it) {
    if (KIND == 0) {
#define I0(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(w[i]) : "v"(a[i]), "s"(0xD2511F53u) : "vcc"); a[i] = (uint32_t)(w[i] >> 32) ^ (uint32_t)w[i];
      // dependent through a cheap xor is unavoidable for a chain; count the xor too -> measured separately as KIND 6
      BLOCK64(I0)
    } else if (KIND == 1) {
#define I1(i) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "s"(0xD2511F53u));
      BLOCK64(I1)
    } else if (KIND == 2) {
#define I2(i) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "s"(0xD2511F53u));
      BLOCK64(I2)
    } else if (KIND == 3) {
#define I3(i) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]), "s"(seed));
      BLOCK64(I3)
    } else if (KIND == 4) {
#define I4(i) asm volatile("v_log_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
      BLOCK64(I4)
    } else if (KIND == 5) {
#define I5(i) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(f[i]) : "v"(f[i]), "s"(1.0001f));
      BLOCK64(I5)
    } else if (KIND == 6) {
#define I6(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
      BLOCK64(I6)
    } else if (KIND == 7) {
#define I7(i) asm volatile("v_sin_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
      BLOCK64(I7)
    } else if (KIND == 8) {
#define I8(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
      BLOCK64(I8)
    } else if (KIND == 9) {
#define I9(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(w[i]) : "v"(w[i]), "v"(w[(i + 1) & 7]));
      BLOCK64(I9)
    } else if (KIND == 10) {
#define I10(i) asm volatile("v_sqrt_f32 %0, %1" : "=v"(f[i]) : "v"(f[i]));
      BLOCK64(I10)
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  uint64_t acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i] + (uint64_t)f[i] + w[i];
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { out[0] = t1 - t0; out[2] = r1 - r0; }
  if (acc == 0x1234567) out[1] = acc;
}

template <int KIND>
void run(const char* name, uint64_t* d, int extra_per_ins) {
  (void)extra_per_ins;
  printf("%-28s", name);
  // (a) one wave's own view: s_memtime ticks per instruction with 1 and 2 waves on its SIMD (one block per CU)
  for (int waves_per_simd : {1, 2}) {
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 12345u);
    hipDeviceSynchronize();
    uint64_t h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("  w%d: %5.2f tick/ins", waves_per_simd, (double)h / ((double)ITERS * REP));
  }
  // (b) chip-wide throughput: 16 blocks of 512 threads per CU queued (8 waves per SIMD resident, two rounds), wall time
  // from HIP events -> ns per wave64 instruction per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * 16;
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, 12345u);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double wave_ins = (double)blocks * 8 * ITERS * REP;
  const double ns_per = (double)ms * 1e6 / (wave_ins / 1024.0);
  uint64_t h3[3] = {0, 0, 0};
  hipMemcpy(h3, d, 24, hipMemcpyDeviceToHost);
  // shader clock while the chip is full of this instruction: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz)
  const double ghz = h3[2] ? (double)h3[0] / (double)h3[2] * 0.1 : 0.0;
  printf("   chip-wide: %6.3f ns per wave-instruction per SIMD; shader clock %4.2f GHz -> %4.2f cycles\n", ns_per, ghz, ns_per * ghz);
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 64);
  run<5>("v_fma_f32", d, 0);
  run<6>("v_xor_b32", d, 0);
  run<3>("v_bitop3_b32", d, 0);
  run<9>("v_pk_fma_f32", d, 0);
  run<8>("v_cvt_f32_u32", d, 0);
  run<2>("v_mul_lo_u32", d, 0);
  run<1>("v_mul_hi_u32", d, 0);
  run<0>("v_mad_u64_u32 (+1 xor)", d, 0);
  run<4>("v_log_f32", d, 0);
  run<10>("v_sqrt_f32", d, 0);
  run<7>("v_sin_f32", d, 0);
  return 0;
}
