// IW1: the generator side of the importance-weighted objective in ONE launch (included by zs_bernoulli.hip).
//
// The reference evaluates it as a per-node loop (importance_weighted_objective.py:66-100): log p(z) by Normal._log_prob over
// the value it has just drawn (normal.py:109-126), log p(x|z) by Bernoulli._log_prob over [K, B, X] (bernoulli.py:84-95), their
// sum, minus log q, then compute_iw_term / vimco over the K particles and the batch mean (:16-25,123-132,152-191).  Rounds 1-3
// ran that as K2 + K3 + K4b: three launches, two of them launch-floor-sized, with the row sums handed from kernel to kernel
// through memory.  Here ONE WORKGROUP OWNS ONE DATAPOINT, so everything between the streaming pass and the objective stays on
// the CU:
//
//   the NW waves of workgroup r share out the K particle rows p[k, r, :] and issue ALL their loads up front (the observation
//   row x[r, :] and the prior's per-element constants are parked in LDS once per workgroup); the wave that streams row k also
//   evaluates the prior's log-density of the latent row z[k, r, :] (lanes < Dz / 4, 16 bytes each, loaded with the row); both
//   row sums go to LDS; after one barrier wave 0 -- lane = particle, its log q / extra-rows operands prefetched at kernel start -- forms
//   log w = ((rows_a + log p(z)) + log p(x|z)) - log q and runs the wave-level IW / VIMCO reduction of K4 (iw_wave_row).
//
// The batch mean is the only cross-workgroup step, and it is ONE atomic: every workgroup adds {1 << S | biased fixed-point
// cost} to a 64-bit word.  Integer addition is associative, so the sum does not depend on the order of arrival (deterministic,
// which a float atomic is not), and the workgroup whose add returns count R - 1 holds the total: it writes the mean and puts
// the word back to zero.  K4b's "partials + ticket, last arrival re-reads" is three dependent round trips through L2 (~4.5 us
// of its 5.3); this is one (~1.5 us).  Layout of the word for cb = ceil(log2 R):
//     bit 63            sticky flag "some cost was not representable" (non-finite, or |cost| >= 2^bound): the mean is then NaN
//     bits [S, 63)      number of workgroups counted so far, S = 62 - cb
//     bits [0, S)       sum of (round(cost * 2^scale) + BIAS), BIAS = 2^(S - 1 - cb): every addend is non-negative and the
//                       field cannot carry into the count (R * 2 * BIAS <= 2^S)
// with scale = S - 1 - cb - bound bits (R = 256: 2^-21 absolute resolution per datapoint; R = 32 768: 2^-11).
//
// A first attempt kept K3's one-wave-per-row grid and handed the row sums to a per-datapoint "last arrival" through memory
// (write-through stores, a ticket per datapoint): 29.5 us at B = 256, K = 50 against 16.5 us for the three launches it
// replaced -- every cross-CU hop costs about as much as a kernel boundary (profiles/r04_iw1_timing.txt, first table).
#pragma once
#include "zs_common.h"
#include "zs_iw_math.h"
#include "../../include/zs_hip.h"

namespace zs {

struct Iw1Args {
  // Bernoulli term: p (probabilities or logits) [K, R, D4 float4], x rows [R, D4] (shared by the particles) or [K * R, D4]
  const float4* p;
  const float4* x;
  int x_full;
  int64_t R;
  int D4;
  int K;
  // Normal term of a given value (optional): z [K, R, Dz4 float4], mean / scale [R, Dz4] or one scalar each.  Without the term
  // (has_z == 0) the three pointers are redirected to readable memory (p) with Dz4 = 1: the kernel loads unconditionally.
  const float4* z;
  const float* pmu;
  const float* psg;
  int has_z, pmu_scalar, psg_scalar, psg_is_logstd;
  int Dz4;
  // ready-made rows of further generator nodes (optional) and log q: K-fastest [R, ld]
  const float* rows_a;
  int64_t ld_a;
  const float* logq;
  int64_t ld_q;
  int estimator;
  // outputs
  float* lp_x;        // [R, K] row sums of the Bernoulli term
  float* lp_z;        // [R, K] row sums of the Normal term (optional)
  float* cost_b;
  float* bound_b;
  float* coef_p;      // [R, K]
  float* coef_q;      // [R, K]
  float scale;
  float* mean_cost;
  unsigned long long* acc;   // the batch mean's accumulator words (see above): [0] the total, [1 .. 16] shards; zero between launches
  int cb, bound_bits, sharded;
  float inv_B;
  int variant;        // timing experiments only (-DZS_EXPERIMENTS): 0 = the kernel as shipped
};

// layout of the accumulator for R datapoints
__host__ __device__ __forceinline__ int iw1_cb(int64_t R) {
  int cb = 0;
  while ((1ll << cb) < R) ++cb;
  return cb;
}

// ONE lane per workgroup: count this workgroup's cost; the lane that completes the sum writes the mean.
// Two levels when there are many workgroups: 256 same-address atomics arriving together serialise in L2 (~11 ns each: 2.8 us,
// MI355X_MICROARCH.md "dequeue"), so workgroup r adds to shard word 1 + (r mod 16) -- same layout, 16 arrivals each -- and the
// workgroup that completes a shard moves the shard's (still biased) sum and its count to word 0.
#define ZS_IW1_SHARDS 16
__device__ __forceinline__ void iw1_finish(const Iw1Args& a, unsigned long long tot, int S, int bias_bits, int scale_bits) {
  const long long sum = (long long)(tot & ((1ull << S) - 1ull)) - (long long)((unsigned long long)a.R << bias_bits);
  float m = (float)((double)sum / (double)(1ull << scale_bits) * (double)a.inv_B);
  if (tot >> 63) m = __uint_as_float(0x7fc00000u);
  a.mean_cost[0] = m;
  __hip_atomic_store(a.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // back to zero for the next launch
}
__device__ __forceinline__ void iw1_poison(unsigned long long* w) {
  // rare path: raise the flag, and let it land before the contribution is counted
  (void)__hip_atomic_fetch_or(w, 1ull << 63, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void iw1_add_cost(const Iw1Args& a, float cost, int64_t r) {
  const int S = 62 - a.cb, bias_bits = S - 1 - a.cb, scale_bits = bias_bits - a.bound_bits;
  const unsigned long long mask = (1ull << S) - 1ull;
  long long fx = 0;
  const bool ok = fabsf(cost) < __uint_as_float((unsigned)(127 + a.bound_bits) << 23);       // false for NaN / inf as well
  if (ok) fx = __double2ll_rn((double)cost * (double)(1ull << scale_bits));
  const unsigned long long add = (1ull << S) + (unsigned long long)((long long)(1ull << bias_bits) + fx);
  if (!a.sharded) {
    if (!ok) iw1_poison(a.acc);
    const unsigned long long tot = __hip_atomic_fetch_add(a.acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
    if (((tot << 1) >> (S + 1)) == (unsigned long long)a.R) iw1_finish(a, tot, S, bias_bits, scale_bits);
    return;
  }
  const int shard = (int)(r & (ZS_IW1_SHARDS - 1));
  const unsigned long long members = (unsigned long long)((a.R - shard + ZS_IW1_SHARDS - 1) / ZS_IW1_SHARDS);
  unsigned long long* w = a.acc + 1 + shard;
  if (!ok) iw1_poison(w);
  const unsigned long long tot = __hip_atomic_fetch_add(w, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add;
  if (((tot << 1) >> (S + 1)) != members) return;
  __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // the shard word back to zero
  if (tot >> 63) iw1_poison(a.acc);
  const unsigned long long add2 = (members << S) + (tot & mask);                       // the shard's count and its biased sum
  const unsigned long long tot2 = __hip_atomic_fetch_add(a.acc, add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + add2;
  if (((tot2 << 1) >> (S + 1)) == (unsigned long long)a.R) iw1_finish(a, tot2, S, bias_bits, scale_bits);
}

// Workgroup r = datapoint r; NW = blockDim.x / 64 waves share out its K rows, wave w taking rows w, w + NW, ... -- at most
// ROUNDS of them (K <= 64 and NW = min(K, 16), so ROUNDS <= 4).  Two rows per wave are in flight (96 KB per CU at 16 waves):
// the loads of row i + 2 go out before row i is reduced.  Measured at B = 256, K = 50 (profiles/r04_iw1_timing.txt): the
// arithmetic of a CU's 50 rows is 6.9 us when it starts only after every load has landed (all loads issued up front), the loads
// alone 6.6 us -- the two have to overlap, and the arithmetic has to be short (packed fp32: bern_piece_acc).
template <bool LOGITS, int ROUNDS>
__global__ __launch_bounds__(1024) void k_iw1_block(Iw1Args a) {
  __shared__ float s_lx[64], s_lz[64];
  __shared__ float4 s_x[256], s_omx[256];      // the observation row and 1 - x: shared by the datapoint's rows, read per round
  __shared__ float4 s_sgn[256];                // 2x - 1 (used when every x of the row is 0 or 1: one logarithm per element)
  __shared__ float4 s_zm[64], s_zl[64], s_zp[64];      // the prior's mean, log sigma, sigma^-2 per 16-byte piece of the latent row
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int64_t r = blockIdx.x;
  const int K = a.K, D4 = a.D4;
  // column of each of the lane's four 16-byte pieces, clamped into the row: loads are unconditional (no exec-mask change
  // between two loads), the arithmetic of a clamped piece is discarded
  int col[4];
  bool ok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + 64 * u;
    ok[u] = c < D4;
    col[u] = ok[u] ? c : D4 - 1;
  }
  const int zc = lane < a.Dz4 ? lane : (a.Dz4 > 0 ? a.Dz4 - 1 : 0);
  const bool has_z = a.has_z != 0;
  const bool zl = has_z && lane < a.Dz4;
  // ---- loads.  No branch between any two of them (a branch makes the compiler wait for the loads already issued), so every
  // lane loads unconditionally from a clamped / redirected address and the results that are not wanted are dropped.  The
  // small shared operands go FIRST: vector-memory results return in order, and what the barrier below waits for must not
  // queue behind the rows.
  const int xi = (int)threadIdx.x < D4 ? (int)threadIdx.x : D4 - 1, xi2 = (int)(threadIdx.x + blockDim.x) < D4 ? (int)(threadIdx.x + blockDim.x) : D4 - 1;
  const float4 x_stage = a.x[r * D4 + xi], x_stage2 = a.x[r * D4 + xi2];       // (two passes: rows of up to 2 * blockDim pieces;
                                                                                //  the staging loop below takes the rest)
  // the prior's parameters: element stride 0 for a scalar operand (its buffer has one element)
  const float* __restrict__ pmp = a.pmu + (a.pmu_scalar ? 0 : (r * a.Dz4 + zc) * 4);
  const float* __restrict__ psp = a.psg + (a.psg_scalar ? 0 : (r * a.Dz4 + zc) * 4);
  const int pms = a.pmu_scalar ? 0 : 1, pss = a.psg_scalar ? 0 : 1;
  const float4 pm4 = make_float4(pmp[0], pmp[pms], pmp[2 * pms], pmp[3 * pms]);
  const float4 ps4 = make_float4(psp[0], psp[pss], psp[2 * pss], psp[3 * pss]);
  const int tl = lane < K ? lane : K - 1;
  const float* __restrict__ arow = a.rows_a ? a.rows_a + r * a.ld_a : a.logq + r * a.ld_q;     // absent rows: log q read twice
  const float t_lq = a.logq[r * a.ld_q + tl], t_ra = arow[tl];                                  // (used by wave 0: lane = particle)
  float4 pv[ROUNDS][4], zv[ROUNDS];
  auto load_row = [&](int i) {
    // (a round beyond K re-reads the wave's first row -- at K = 50, 14 of a datapoint's 64 row slots: FETCH_SIZE shows 1.22 x the
    //  algorithmic bytes, served by the Infinity Cache.  Two ways of not issuing those requests were measured, one 16-byte piece for
    //  every lane and buffer loads with num_records = 0: both correct, both 14.4 -> 15.2-15.5 us at B = 256; profiles/r04_iw1_timing.txt)
    const int k = w + i * NW, kc = k < K ? k : w;
    const int64_t row = (int64_t)kc * a.R + r;
    const float4* __restrict__ prow = a.p + row * D4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pv[i][u] = prow[col[u]];
    }
    zv[i] = a.z[row * a.Dz4 + zc];
  };
  load_row(0);
  if (ROUNDS > 1) load_row(1);
  // ---- the shared operands into LDS
  const bool x_stager = !a.x_full && (int)threadIdx.x < D4, x_stager2 = !a.x_full && (int)(threadIdx.x + blockDim.x) < D4;
  const bool z_stager = has_z && w == NW - 1;
  auto one_minus = [](const float4& v) { return make_float4(1.0f - v.x, 1.0f - v.y, 1.0f - v.z, 1.0f - v.w); };
  auto is_bits = [](const float4& v) {
    return (v.x == 0.0f || v.x == 1.0f) && (v.y == 0.0f || v.y == 1.0f) && (v.z == 0.0f || v.z == 1.0f) && (v.w == 0.0f || v.w == 1.0f);
  };
  auto stage = [&](int c, const float4& v) {
    const float4 o = one_minus(v);
    s_x[c] = v;
    s_omx[c] = o;
    s_sgn[c] = make_float4(v.x - o.x, v.y - o.y, v.z - o.z, v.w - o.w);
    return is_bits(v);
  };
  bool bits = true;
  if (x_stager) bits = stage(threadIdx.x, x_stage);
  if (x_stager2) bits = stage(threadIdx.x + blockDim.x, x_stage2) && bits;
  if (!a.x_full)                       // (fewer than 4 waves and a row of more than 2 * blockDim pieces: the rest, plainly)
    for (int c = threadIdx.x + 2 * blockDim.x; c < D4; c += blockDim.x) bits = stage(c, a.x[r * D4 + c]) && bits;
  if (z_stager) {
    // the prior's parameters are the same for every row of the datapoint: one wave forms log sigma and sigma^-2 (normal.py:121-123)
    const float sv[4] = {ps4.x, ps4.y, ps4.z, ps4.w};
    float lg[4], pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float l2 = log2_fast(a.psg_is_logstd ? expf(sv[j]) : sv[j]);
      lg[j] = l2 * ZS_LN2;
      pr[j] = exp2_fast(-2.0f * l2);
    }
    s_zm[lane] = pm4;
    s_zl[lane] = make_float4(lg[0], lg[1], lg[2], lg[3]);
    s_zp[lane] = make_float4(pr[0], pr[1], pr[2], pr[3]);
  }
  // (the barrier of the staging; also: is every x of this datapoint's row 0 or 1?  Workgroup-uniform.)
  const bool xbits = __syncthreads_and((int)bits) != 0 && !a.x_full;
  float4 sg[4];                                // 2x - 1 of the lane's four pieces, in registers for every row of the wave
#pragma unroll
  for (int u = 0; u < 4; ++u) sg[u] = s_sgn[col[u]];
  // ---- the rows
#pragma unroll
  for (int i = 0; i < ROUNDS; ++i) {
    asm volatile("" ::: "memory");                                // keep the program order of the loads below (and re-read LDS per round)
    if (i + 2 < ROUNDS) load_row(i + 2);
    const int k = w + i * NW;
#ifdef ZS_EXPERIMENTS
    if (a.variant == 3 && pv[i][0].x != 123.456f) continue;       // (timing experiment: the loads without the arithmetic)
#endif
    if (k >= K) continue;                                         // wave-uniform
    zs_f2v acc2 = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 xv, ov;
      if (a.x_full) {
        xv = a.x[((int64_t)k * a.R + r) * D4 + col[u]];
        ov = one_minus(xv);
      } else {
        xv = s_x[col[u]];
        ov = s_omx[col[u]];
      }
      float4 q = pv[i][u];
      if (LOGITS) {
        q.x = sigmoid_fast(q.x);
        q.y = sigmoid_fast(q.y);
        q.z = sigmoid_fast(q.z);
        q.w = sigmoid_fast(q.w);
      }
      zs_f2v t = {0.f, 0.f};
      if (xbits) bern_piece_acc_bits(q, sg[u], t);
      else bern_piece_acc(q, xv, ov, t);
      if (ok[u]) acc2 += t;
    }
    const float acc = wave_sum(acc2.x + acc2.y) * ZS_LN2;
    float nz = 0.f;
    if (has_z) {
      const float4 m4 = s_zm[lane], l4 = s_zl[lane], p4 = s_zp[lane];
      const float t = (normal_lp_term(zv[i].x - m4.x, l4.x, p4.x) + normal_lp_term(zv[i].y - m4.y, l4.y, p4.y)) +
                      (normal_lp_term(zv[i].z - m4.z, l4.z, p4.z) + normal_lp_term(zv[i].w - m4.w, l4.w, p4.w));
      nz = wave_sum(zl ? t : 0.f);
    }
    if (lane == 0) {
      s_lx[k] = acc;
      s_lz[k] = nz;
    }
  }
  __syncthreads();
#ifdef ZS_EXPERIMENTS
  if (a.variant == 2) return;                                     // (timing experiment: no tail)
#endif
  if (w != 0) return;
  // ---- the tail: wave 0, lane = particle
  const bool on = lane < K;
  float l = -INFINITY;
  if (on) {
    const float lx = s_lx[lane], nz = s_lz[lane];
    // the reference adds the generator's nodes left to right, then subtracts log q (:66-77,97-98)
    float lp = lx;
    if (has_z) lp = (a.rows_a ? t_ra + nz : nz) + lx;
    else if (a.rows_a) lp = t_ra + lx;
    l = lp - t_lq;
    a.lp_x[r * K + lane] = lx;
    if (has_z && a.lp_z) a.lp_z[r * K + lane] = nz;
  }
  const float cost = iw_wave_row(l, t_lq, on, lane, K, a.estimator, a.scale, r, a.cost_b, a.bound_b, a.coef_p, a.coef_q);
  if (a.mean_cost && lane == 0) iw1_add_cost(a, cost, r);
}

}  // namespace zs
