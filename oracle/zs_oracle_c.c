/*
 * zs_oracle_c.c -- plain-C CPU restatement of the hot path behind the SAME C ABI as the HIP
 * library (include/zs_hip.h), operating on HOST pointers.  TEST INFRASTRUCTURE ONLY:
 *
 *   - tests/ load it next to libzs_hip.so and call both through one ctypes wrapper, so the
 *     GPU parity tests compare the two libraries entry point by entry point;
 *   - the CPU test-suite injects it as the kernel library of the zhusuan package to exercise
 *     the host logic (Python -> C-ABI marshalling, autograd formulas) without a GPU;
 *   - the shipped package never loads it by itself.
 *
 * Parity is pinned: tests/test_oracle_c_golden.py checks these functions against the golden
 * fixtures generated from the real reference (tests/golden/gen_golden.py) and against
 * oracle/zs_oracle.py.
 *
 * Arithmetic is fp32 in the reference's op order (paths relative to /root/reference):
 *   Normal     zhusuan/distributions/normal.py:89-126
 *   Bernoulli  zhusuan/distributions/bernoulli.py:46-50,84-95
 *   group sum  zhusuan/distributions/base.py:175-176, framework/stochastic_tensor.py:160-181
 *   IW / VIMCO zhusuan/variational/importance_weighted_objective.py:16-25,123-132,152-191
 *   LME        zhusuan/utils.py:6-21
 *   Logistic   zhusuan/distributions/logistic.py:52-83;  Uniform  zhusuan/distributions/uniform.py:51-85
 * Serial loops, no threads, no SIMD intrinsics: clarity over speed.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../include/zs_hip.h"


/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox4(uint64_t group, uint64_t call, uint64_t seed, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)group, c1 = (uint32_t)(group >> 32), c2 = (uint32_t)call, c3 = (uint32_t)(call >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* uniform strictly inside (0, 1): the 2^23 midpoints (m + 0.5) * 2^-23, m = v >> 9, every one exact in fp32 */
static float u01(uint32_t v) { return ((float)(v >> 9) + 0.5f) * 1.1920928955078125e-07f; }
/* Box-Muller on the word pairs (0, 1) and (2, 3): radius from the uniform u01(word) in (0, 1), angle
 * 2*pi*t with t = (word >> 9) * 2^-23 in [0, 1) (the kernels' definition of the stream, csrc/zs_common.h) */
static void philox_normal4(uint64_t group, uint64_t call, uint64_t seed, float n[4]) {
  uint32_t r[4];
  philox4(group, call, seed, r);
  const double two_pi = 6.283185307179586476925;
  double u0 = u01(r[0]), u2 = u01(r[2]);
  double t1 = (double)(r[1] >> 9) * 1.1920928955078125e-07, t3 = (double)(r[3] >> 9) * 1.1920928955078125e-07;
  double ra = sqrt(-2.0 * log(u0)), rb = sqrt(-2.0 * log(u2));
  n[0] = (float)(ra * cos(two_pi * t1));
  n[1] = (float)(ra * sin(two_pi * t1));
  n[2] = (float)(rb * cos(two_pi * t3));
  n[3] = (float)(rb * sin(two_pi * t3));
}
int zs_abi_version(void) { return ZS_ABI_VERSION; }
/* The shapes the HIP library's flat-plane sampling kernel takes for 2 K particles (csrc/zs_sample_tile.h, k1_tile): rows of whole
 * 16-byte pieces, at most 64 of them, a workgroup of whole rows and whole waves within 1024 lanes.  Restated here so that the
 * package's host back-end pairs draws exactly where the GPU does. */
int zs_normal_sample_pair_one_launch(int64_t K, int64_t M, int64_t D, int want_lp) {
  (void)want_lp;
  if (K < 1 || M < 1 || D < 1 || (M % D) != 0 || (D % 4) != 0 || 2 * K > 0x7fffffff) return 0;
  const int64_t D4 = D / 4, M4 = M / 4;
  if (D4 > 64 || M4 >= ((int64_t)1 << 28)) return 0;
  int64_t a = 64, b = D4;
  while (b) { const int64_t t = a % b; a = b; b = t; }
  const int64_t l = 64 / a * D4;
  int64_t TB = l * ((256 + l - 1) / l);
  if (TB > 1024) TB = l;
  if (TB > 1024) return 0;
  return ((M4 + TB - 1) / TB) * (2 * K) < ((int64_t)1 << 31);       /* (an upper bound of the kernel's item count) */
}
const char* zs_build_info(void) { return "libzs_oracle: plain-C CPU restatement (test infrastructure, host pointers)"; }
const char* zs_error_string(int code) {
  if (code == 0) return "success";
  if (code == ZS_EINVAL) return "zs(oracle): invalid argument";
  if (code == ZS_ENOTSUP) return "zs(oracle): unsupported configuration";
  return "zs(oracle): unknown error";
}


/* ------------------------------------------------------------------ the numeric entry points, once per precision */
#define REAL float
#define SFX(name) name##_f32
#define SFXL(name) name##_f32_
#define R_LOG logf
#define R_EXP expf
#define R_LOG1P log1pf
#define R_C_NORM (-0.91893853320467274178f)
#define R_EPS 1e-8f
#include "zs_oracle_impl.inc"
#undef REAL
#undef SFX
#undef SFXL
#undef R_LOG
#undef R_EXP
#undef R_LOG1P
#undef R_C_NORM
#undef R_EPS

#define REAL double
#define SFX(name) name##_f64
#define SFXL(name) name##_f64_
#define R_LOG log
#define R_EXP exp
#define R_LOG1P log1p
#define R_C_NORM (-0.91893853320467274178)
#define R_EPS 1e-8
#include "zs_oracle_impl.inc"
#undef REAL
#undef SFX
#undef SFXL
#undef R_LOG
#undef R_EXP
#undef R_LOG1P
#undef R_C_NORM
#undef R_EPS

/* timing hooks exist only in the HIP library */
int zs_prof_enable(int on) { (void)on; return ZS_ENOTSUP; }
int zs_prof_kernel_id(const char* entry_point) { (void)entry_point; return ZS_ENOTSUP; }
int zs_prof_query(int kernel_id, double* total_ms, double* min_ms, double* max_ms, int64_t* count) {
  (void)kernel_id; (void)total_ms; (void)min_ms; (void)max_ms; (void)count;
  return ZS_ENOTSUP;
}

int64_t zs_prof_durations(int kernel_id, double* out_ms, int64_t capacity) {
  (void)kernel_id; (void)out_ms; (void)capacity;
  return ZS_ENOTSUP;
}

/* raw Philox words, for the known-answer test of the generator itself */
void zs_oracle_philox4x32_10(uint64_t group, uint64_t call, uint64_t seed, uint32_t out[4]) {
  philox4(group, call, seed, out);
}
