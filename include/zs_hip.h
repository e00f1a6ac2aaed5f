/*
 * zs_hip.h -- C ABI of the MI355X (gfx950) variational-inference hot path.
 *
 * The reference (thuwzy/ZhuSuan-PyTorch) is pure Python and has no FFI of its
 * own; its boundary for this path is the Python class API.
 * This header is the private C ABI that sits underneath that API: each entry
 * point replaces the whole-tensor PyTorch op sequence of one reference method
 * (cited per function, paths relative to the reference root).  INTEGRATION.md
 * shows the ctypes stub a reference maintainer would add to call it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller;
 *     nothing is allocated, freed or synchronised inside a call;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     call only enqueues kernels on it and returns;
 *   - return value: 0 on success, a positive hipError_t if a launch failed,
 *     ZS_EINVAL / ZS_ENOTSUP (negative) for rejected arguments;
 *   - "periodic broadcast": an operand given with period P is read as
 *     a[i % P] for flat index i of the full [K, R, D] problem, which covers the
 *     reference's leading-axis `repeat` of parameters (normal.py:94-95,112-116)
 *     and scalars (P = 1).  P must divide K*R*D;
 *   - row results: element (k, r) of a [K, R] result is written at
 *     out[k * stride_k + r * stride_r] (strides in elements).  The objectives
 *     use the K-fastest layout stride_k = 1, stride_r = ld >= K so that the
 *     importance-weight reduction finds the K particles of one datapoint on
 *     the 64 lanes of one wavefront;
 *   - gfx950 ONLY, also where it is not spelled in C: the entry points that take a `ticket` / `tickets` / `acc` word combine
 *     partial results of many workgroups in the last one to arrive.  The hand-off is NOT a release/acquire pair of the HIP
 *     memory model (an agent-scope release per workgroup costs 1.7-6.5 us on this chip): partials are stored write-through
 *     (sc1), the storing waves drain (s_waitcnt vmcnt(0)), one lane per workgroup adds to the ticket with a relaxed agent-scope
 *     atomic and the last arrival reads with sc1 loads or behind one agent-scope acquire -- the form MI355X_MICROARCH.md lists as
 *     measured for gfx950 (csrc/zs_onelaunch.h).  A port to another target has to re-validate it; here it is held by the
 *     twice-on-one-workspace tests of every such kernel and by tools/soak.py (replayed step == eager twin after thousands of
 *     launches).  IW1's batch mean is exempt: its 64-bit atomics carry the data themselves.
 */
#ifndef ZS_HIP_H
#define ZS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZS_ABI_VERSION 15
#define ZS_EINVAL (-1)
#define ZS_ENOTSUP (-2)

#define ZS_IW_SGVB 0
#define ZS_IW_VIMCO 1

/* ABI version of the loaded library (== ZS_ABI_VERSION). */
int zs_abi_version(void);

/* One line describing the loaded library: target, ABI version and whether it was built with the experiment knobs
 * (`make EXTRA=-DZS_EXPERIMENTS`: dispatch overrides read from ZS_* environment variables).  The default build never
 * reads the environment; bench.py refuses an experiments build unless told otherwise and records this string. */
const char* zs_build_info(void);

/* Human-readable text for a return code of any function below. */
const char* zs_error_string(int code);

/* ---------------------------------------------------------------------------
 * K1  Normal: fused sample + log-prob.
 * Replaces Normal._sample (zhusuan/distributions/normal.py:89-107), the
 * log-density of the fresh sample Normal._log_prob (normal.py:109-126), the
 * group sum of Distribution.log_prob (zhusuan/distributions/base.py:175-176)
 * and the trailing reduce_sum of StochasticTensor.log_prob
 * (zhusuan/framework/stochastic_tensor.py:160-181).
 *
 *   z[k, m]   = mu[m] + sigma[m] * eps[k, m]              k < K, m < M
 *   lp[k, r]  = sum_{d < D} ( -0.5*log(2*pi) - log(sigma) - 0.5*exp(-2*log(sigma)) * (z - mu)^2 )
 *               over m = r*D + d,  r < R = M / D
 *
 * eps == NULL: eps is drawn in-kernel from Philox4x32-10 keyed by `seed`, with
 * counter (group = (k*M + m) / 4, call = offset); the same (seed, offset)
 * regenerates the same draw in the backward call.
 * rng_state (optional DEVICE pointer to two uint64 {seed, base}): when non-NULL the
 * kernel itself reads seed = rng_state[0] and uses call = rng_state[1] + offset, so a
 * launch captured in a hipGraph draws fresh numbers on every replay once the
 * caller bumps rng_state[1] between replays (`seed` is then ignored).
 * lp == NULL: sample only.
 * sigma_is_logstd != 0: the `sigma` operand holds log(sigma) -- the Normal(logstd=...) constructor,
 * normal.py:56 `std = exp(logstd)` -- and the kernel forms sigma = exp(.) itself (here and in every
 * Normal entry point below that takes the flag; their gsigma output is then d/d logstd = sigma * d/d sigma):
 * the caller launches no exp forward and no multiply backward.
 * rng_used (optional DEVICE pointer to two uint64): receives the resolved {seed, call} of this draw.  A backward
 * call that may run after the caller has advanced the live rng_state (two objectives per step, a delayed or
 * retained backward) passes it as ITS rng_state with offset 0 and regenerates exactly this draw.
 * -------------------------------------------------------------------------*/
int zs_normal_sample_logprob_f32(const float* mu, const float* sigma, const float* eps,
                                 uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                 float* z, float* lp,
                                 int64_t K, int64_t M, int64_t D,
                                 int64_t lp_stride_k, int64_t lp_stride_r,
                                 int sigma_is_logstd, uint64_t* rng_used, void* stream);

/* 1 when zs_normal_sample_logprob_pair_f32 (below) serves (K, M, D) with ONE launch (given 16-byte aligned operands), else 0. */
int zs_normal_sample_pair_one_launch(int64_t K, int64_t M, int64_t D, int want_lp);

/* K1 twice: two independent draws of K particles each with the Philox call ids `offset` and `offset + 1` -- what the
 * reference's objectives do with every latent: the node factory draws (stochastic_tensor.py:115-127 through bn.py:158),
 * the objective's re-read of node.tensor draws again (elbo.py:122, importance_weighted_objective.py:85) -- as ONE launch
 * where the flat-plane kernel takes the shape (else as the two launches it stands for).  z [2 K, M]: particles [0, K) are
 * the first draw; lp element (k, r) of draw j is written at lp[(j K + k) * lp_stride_k + r * lp_stride_r] (K-fastest
 * [R, 2 K]: lp_stride_k = 1, lp_stride_r = 2 K).  Each half is bit for bit what zs_normal_sample_logprob_f32 writes for its
 * call id (in-kernel Philox only); rng_used receives the first draw's ids (the second's are + 1). */
int zs_normal_sample_logprob_pair_f32(const float* mu, const float* sigma, uint64_t seed, uint64_t offset,
                                      const uint64_t* rng_state, float* z, float* lp, int64_t K, int64_t M, int64_t D,
                                      int64_t lp_stride_k, int64_t lp_stride_r, int sigma_is_logstd, uint64_t* rng_used,
                                      void* stream);

/* Backward of K1 for a reparameterised node (normal.py:104-105):
 *   gmu[m]    = sum_k gz[k, m]
 *   gsigma[m] = sum_k gz[k, m] * eps[k, m]  -  (sum_k glp[k, r(m)]) / sigma[m]
 * gz or glp may be NULL (treated as zero).  eps as in the forward call. */
int zs_normal_sample_logprob_bwd_f32(const float* sigma, const float* eps,
                                     uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                     const float* gz, const float* glp,
                                     int64_t glp_stride_k, int64_t glp_stride_r,
                                     float* gmu, float* gsigma,
                                     int64_t K, int64_t M, int64_t D, int sigma_is_logstd, void* stream);

/* ---------------------------------------------------------------------------
 * K2  Normal: log-prob of a given value (prior p(z), likelihood p(y|.)).
 * Replaces Normal._log_prob on an observed value with parameters repeated
 * along the sample axis (normal.py:109-126), plus the group / trailing sums
 * as for K1.  Problem [K, R, D]; x, mu, sigma periodic with Px, Pm, Ps.
 * -------------------------------------------------------------------------*/
int zs_normal_logprob_f32(const float* x, int64_t Px, const float* mu, int64_t Pm,
                          const float* sigma, int64_t Ps, float* lp,
                          int64_t K, int64_t R, int64_t D,
                          int64_t lp_stride_k, int64_t lp_stride_r, int sigma_is_logstd, void* stream);

/* Backward of K2, element-wise partials of size K*R*D each (any may be NULL):
 *   gx = -g*prec*(x-mu),  gmu = +g*prec*(x-mu),  gsigma = g*(prec*(x-mu)^2 - 1)/sigma,
 * with g = glp[k, r] and prec = exp(-2*log(sigma)). */
int zs_normal_logprob_bwd_f32(const float* x, int64_t Px, const float* mu, int64_t Pm,
                              const float* sigma, int64_t Ps,
                              const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                              float* gx, float* gmu, float* gsigma,
                              int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream);

/* Backward of K2 reduced over the K axis for parameters of period R*D
 * (mu, sigma of shape [R, D] repeated K times: the IWAE / non-reparameterised
 * case, normal.py:102,112-116).  x has full size [K, R, D].
 *   gmu[r, d] = sum_k ...,  gsigma[r, d] = sum_k ...;  gx (full size) optional. */
int zs_normal_logprob_bwd_ksum_f32(const float* x, const float* mu, const float* sigma,
                                   const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                   float* gx, float* gmu, float* gsigma,
                                   int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream);

/* ---------------------------------------------------------------------------
 * K3  Bernoulli: log-prob row sums.
 * Replaces Bernoulli._log_prob (zhusuan/distributions/bernoulli.py:84-95) and
 * the sums as above:
 *   lp[k, r] = sum_d  x*log(p + 1e-8) + (1 - x)*log((1 - p) + 1e-8)
 * p has full size [K, R, D]; x is periodic with Px (x [B, X] against
 * p [K, B, X], bernoulli.py:88-92).
 * -------------------------------------------------------------------------*/
int zs_bernoulli_logprob_f32(const float* p, const float* x, int64_t Px, float* lp,
                             int64_t K, int64_t R, int64_t D,
                             int64_t lp_stride_k, int64_t lp_stride_r, void* stream);

/* gp[k, r, d] = glp[k, r] * ( x/(p + 1e-8) - (1 - x)/((1 - p) + 1e-8) ) */
int zs_bernoulli_logprob_bwd_f32(const float* p, const float* x, int64_t Px,
                                 const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                 float* gp, int64_t K, int64_t R, int64_t D, void* stream);

/* Gradient w.r.t. the OBSERVATION: bernoulli.py:94 is differentiable in `sample`, and `given` keeps its graph through
 * base.py:161-178 (a model whose observed Bernoulli value comes out of a differentiable net).  The observation is periodic
 * with Px, so its gradient is the sum over the elements that read it:
 *   gx[j] = sum over i in [0, K*R*D), i % Px == j, of  glp[k, r] * s[r] * ( log(p_i + 1e-8) - log((1 - p_i) + 1e-8) ),
 * (k, r) = the row of element i, s[r] = gscale[r * gscale_stride] when gscale is given (the device-resident incoming
 * gradient of the objectives that keep their row gradients as coefficients), else 1; from_logits: p = sigmoid(streamed
 * operand).  gx has Px elements; the additions are made in a fixed order (deterministic: ascending i, or -- few observations,
 * many repetitions -- ascending within four equal shares of the repetitions that are then added in order). */
int zs_bernoulli_logprob_bwd_x_f32(const float* p, int from_logits, int64_t Px,
                                   const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                   const float* gscale, int64_t gscale_stride,
                                   float* gx, int64_t K, int64_t R, int64_t D, void* stream);

/* Same density evaluated from logits: p = sigmoid(logit) = 1/(1 + exp(-logit))
 * (bernoulli.py:46-50) followed by the formula above; removes the separate
 * sigmoid pass and the round trip of p.
 * probs_out (optional, full size) receives p. */
int zs_bernoulli_logits_logprob_f32(const float* logits, const float* x, int64_t Px,
                                    float* lp, float* probs_out,
                                    int64_t K, int64_t R, int64_t D,
                                    int64_t lp_stride_k, int64_t lp_stride_r, void* stream);

/* glogits = glp * ( x/(p+1e-8) - (1-x)/((1-p)+1e-8) ) * p * (1-p) */
int zs_bernoulli_logits_logprob_bwd_f32(const float* logits, const float* x, int64_t Px,
                                        const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                        float* glogits, int64_t K, int64_t R, int64_t D, void* stream);

/* K5  Bernoulli._sample (bernoulli.py:72-82): out[i] = (u_i < p[i % Pp]) ? 1 : 0,
 * u from Philox4x32-10 (seed, offset).  Generation path only. */
int zs_bernoulli_sample_f32(const float* p, int64_t Pp, float* out, int64_t N,
                            uint64_t seed, uint64_t offset, const uint64_t* rng_state, void* stream);

/* ---------------------------------------------------------------------------
 * K4  Importance-weighted reduction over the K particles of each datapoint.
 * Replaces compute_iw_term, ImportanceWeightedObjective.sgvb / .vimco
 * (zhusuan/variational/importance_weighted_objective.py:16-25, 123-132,
 * 152-191) and log_mean_exp (zhusuan/utils.py:6-21).
 * Inputs are K-fastest: logp[b*ld_p + k], logq[b*ld_q + k]; log_w = logp - logq.
 *   bound_b[b] = log_mean_exp_k(log_w)                      (the true IW bound)
 *   sgvb : cost_b[b] = -sum_k wt_k * log_w_k,               wt = softmax_k(log_w)
 *          coef_p = -wt,  coef_q = +wt
 *   vimco: cost_b[b] = -sum_k logq_k * signal_k - sum_k wt_k * log_w_k,
 *          signal_k = log_mean_exp(log_w) - log_mean_exp(log_w with entry k replaced by
 *                     the mean of the others);  coef_p = -wt,  coef_q = wt - signal
 * coef_p / coef_q are d cost_b / d logp, d cost_b / d logq, dense [B, K].
 * Any output pointer may be NULL.  vimco requires K >= 2.
 * -------------------------------------------------------------------------*/
int zs_iw_reduce_f32(const float* logp, int64_t ld_p, const float* logq, int64_t ld_q,
                     int64_t B, int64_t K, int estimator,
                     float* cost_b, float* bound_b, float* coef_p, float* coef_q, void* stream);

/* K4b  The whole importance-weighted objective in ONE launch: zs_iw_reduce plus what the callers of
 * ImportanceWeightedObjective.sgvb / .vimco otherwise do with separate whole-tensor ops
 * (importance_weighted_objective.py:97-98,131-132,191):
 *   - the log-joint of the generator given as the sum of two terms, logp = logp_a + logp_b (logp_b may be NULL),
 *     e.g. log p(x|z) + log p(z): rounded as (a + b) - logq, exactly like the reference's separate add;
 *   - the batch mean of the per-datapoint costs, mean_cost[0] = (1/B) sum_b cost_b, formed deterministically
 *     (per-workgroup partial sums in `workspace`, the last workgroup to finish adds them in index order);
 *   - both coefficient matrices in one [2, B, K] buffer (coef[0] = d cost / d logp, coef[1] = d cost / d logq),
 *     already scaled by 1/B when the mean is requested, so that backward is a single multiply by the incoming gradient.
 * want_mean != 0 requires mean_cost, workspace (workspace_len >= number of workgroups; 4096 is always enough) and
 * ticket (one uint32, zero before the first use; the kernel leaves it zero again).  cost_b / bound_b may be NULL. */
int zs_iw_objective_f32(const float* logp_a, int64_t ld_a, const float* logp_b, int64_t ld_b,
                        const float* logq, int64_t ld_q, int64_t B, int64_t K, int estimator, int want_mean,
                        float* cost_b, float* bound_b, float* coef, float* mean_cost,
                        float* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);


/* ---------------------------------------------------------------------------
 * IW1  The GENERATOR SIDE of the importance-weighted objective in one launch: the Bernoulli likelihood's row sums (K3), the
 * log-density of the latent value under a Normal prior (K2), the addition of the generator's log-joint terms, the
 * subtraction of log q and the whole of K4b.  Replaces the per-node loop of ImportanceWeightedObjective.log_joint /
 * .forward (zhusuan/variational/importance_weighted_objective.py:66-100) over Normal._log_prob (normal.py:109-126) and
 * Bernoulli._log_prob (bernoulli.py:84-95) plus .sgvb / .vimco (:102-191) for the IWAE caller
 * (examples/variational_autoencoder/iwae.py:49-81):
 *
 *   lp_x[r, k]  = sum_d  x*log(p + 1e-8) + (1 - x)*log((1 - p) + 1e-8)         p [K, R, D] (probabilities, or logits with
 *                                                                              from_logits != 0: p = sigmoid(logit) as K3);
 *                                                                              x periodic with Px = R*D (shared by the
 *                                                                              particles) or K*R*D
 *   lp_z[r, k]  = sum_d  log N(z[k, r, d] | pmu, psigma)                        optional (z == NULL: no such term); z [K, R, Dz];
 *                                                                              pmu / psigma with period 1 or R*Dz
 *   log_w[r, k] = ((rows_a[r, k] + lp_z[r, k]) + lp_x[r, k]) - logq[r, k]       rows_a optional: ready-made K-fastest rows of
 *                                                                              further generator nodes; absent terms are
 *                                                                              left out of the sum (the reference adds the
 *                                                                              nodes left to right)
 * and then exactly zs_iw_objective on log_w: cost_b, bound_b, coef [2, R, K] (scaled by 1/R when want_mean), mean_cost.
 * K-fastest [R, K] outputs: lp_x (required), lp_z (optional; the float64 twin requires it when z is given).  cost_b is
 * required; mean_cost (want_mean) is the deterministic batch mean: the grid is one workgroup per CU (workgroup g takes datapoints g,
 * g + G, ...: any R up to 2^20), a workgroup adds up the costs of its datapoints in fixed point and adds its share to the 64-bit words
 * `acc` (ZS_IW1_ACC_WORDS = 64 zero-initialised device words, handed back at zero; integer addition does not depend on the order of
 * arrival).  Two words per sum -- round(cost * 2^s1) and the rounding residual -- make the result the correctly rounded mean of the
 * fp32 costs whatever their magnitude (ABI 13: one word, 2^-21 absolute per datapoint at R = 256); a +inf / -inf / NaN cost gives
 * a +inf / -inf / NaN mean as the fp32 mean of importance_weighted_objective.py:191 would, a finite |cost| >= 2^24 gives NaN
 * (zs_iw_objective's float sum returns the finite mean there: the one documented divergence between the two entry points).
 * The mean is finished by a wave that WATCHES the words until every workgroup's share has arrived; should a launch lose workgroups
 * (it cannot in a healthy process) the watcher gives up after 2 s of the device clock, stores NaN, leaves the words as they are and
 * raises acc[ZS_IW1_POISON_WORD]: a NaN mean over finite costs means "accumulator poisoned" -- every later launch on it would
 * miscount -- and the owner re-zeroes all 64 words (zhusuan._ops.iw1_accumulators_ok() / reset_iw1_accumulators()).
 * A datapoint whose shared observation row holds only exact 0s and 1s (binarised data) is evaluated with ONE logarithm per
 * element -- log(fma(p, 2x - 1, 1 - x) + 1e-8), bit-identical to the two-term form for every p in [0, 1]; for an invalid p
 * (outside [-1e-8, 1 + 1e-8]) the two-term form's NaN from 0 * log(negative) is not reproduced.
 * Returns ZS_ENOTSUP outside the fused kernel's domain (K <= 64, R <= 2^20, D % 4 == 0, 256 <= D <= 1024, Dz % 4 == 0,
 * Dz <= 256, 16-byte aligned operands): the caller then composes K2 / K3 / K4b itself.
 * -------------------------------------------------------------------------*/
#define ZS_IW1_ACC_WORDS 64
#define ZS_IW1_POISON_WORD 63
int zs_bernoulli_iw_objective_f32(const float* p, int from_logits, const float* x, int64_t Px,
                                  int64_t K, int64_t R, int64_t D,
                                  const float* z, const float* pmu, int64_t Pm, const float* psigma, int64_t Ps,
                                  int64_t Dz, int psigma_is_logstd,
                                  const float* rows_a, int64_t ld_a, const float* logq, int64_t ld_q,
                                  int estimator, int want_mean,
                                  float* lp_x, float* lp_z, float* cost_b, float* bound_b, float* coef, float* mean_cost,
                                  uint64_t* acc, void* stream);

/* Backward of IW1.  The incoming gradient of the objective stays a DEVICE value: gout[r * gout_stride] (gout_stride = 0: the
 * 0-d gradient of the batch mean; 1: one value per datapoint) multiplies the coefficient rows inside the kernels, so no
 * pass over the [2, R, K] coefficient buffer is launched:
 *   gp[k, r, d]      = coef[0][r, k] * gout[r] * d lp_x / d p          (as zs_bernoulli_logprob_bwd; logits form with from_logits)
 * and, when the variational node is handed in (zq != NULL: a non-reparameterised Normal draw zq [K, R, Dq] with parameters
 * qmu, qsigma [R, Dq]; normal.py:102,112-116), the gradient of -sum log q terms w.r.t. its parameters
 *   gqmu[r, d], gqsigma[r, d] = sum_k coef[1][r, k] * gout[r] * d log N(zq[k, r, d] | qmu, qsigma) / d (qmu, qsigma)
 * (as zs_normal_logprob_bwd_ksum).  gp may be NULL (nothing flows into the decoder). */
int zs_bernoulli_iw_objective_bwd_f32(const float* p, int from_logits, const float* x, int64_t Px,
                                      int64_t K, int64_t R, int64_t D,
                                      const float* coef, const float* gout, int64_t gout_stride, float* gp,
                                      const float* zq, const float* qmu, const float* qsigma, int64_t Dq,
                                      int qsigma_is_logstd, float* gqmu, float* gqsigma, void* stream);

/* out[b] = log_mean_exp_k(x[b*ld + k])  (zhusuan/utils.py:6-21, K-fastest rows) */
int zs_log_mean_exp_f32(const float* x, int64_t ld, int64_t B, int64_t K, float* out, void* stream);

/* Standard normals from the same Philox4x32-10 + Box-Muller stream K1 uses:
 * out[i], i < N, group = i / 4.  For tests and for callers that need eps itself. */
int zs_philox_normal_f32(float* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                         void* stream);

/* ===========================================================================
 * Widening: the reference's two other hand-written
 * samplers, Logistic and Uniform, on the same Philox stream and row conventions.
 * ===========================================================================*/

/* ---------------------------------------------------------------------------
 * L1  Logistic: fused sample + log-prob.
 * Replaces Logistic._sample (zhusuan/distributions/logistic.py:52-67) and the
 * log-density of the fresh sample Logistic._log_prob (logistic.py:69-83) with the
 * group / trailing sums as for K1.
 *
 *   eps[k, m] = log(u) - log(1 - u),   u = u[k, m] uniform on (0, 1)        (logistic.py:64-65)
 *   z[k, m]   = loc[m] + scale[m] * eps[k, m]                               (logistic.py:66)
 *   lp[k, r]  = sum_d ( -t - 2*softplus(-t) - log(scale) ),  t = (z - loc) / scale   (logistic.py:81-82)
 *
 * u == NULL: u is drawn in-kernel, uniforms (w >> 9 + 0.5) * 2^-23 (strictly inside (0, 1)) of Philox4x32-10 words w with the
 * counter convention of K1 (group = (k*M + m) / 4, word = (k*M + m) % 4).
 * For the fresh sample t == eps, and -eps - 2*softplus(-eps) == log(u) + log(1 - u):
 * the kernel reuses the two logarithms of the draw.  rng_used: as for K1.
 * -------------------------------------------------------------------------*/
int zs_logistic_sample_logprob_f32(const float* loc, const float* scale, const float* u,
                                   uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                   float* z, float* lp,
                                   int64_t K, int64_t M, int64_t D,
                                   int64_t lp_stride_k, int64_t lp_stride_r, uint64_t* rng_used, void* stream);

/* Backward of L1 (Logistic is always reparameterised, logistic.py:36):
 *   gloc[m]   = sum_k gz[k, m]
 *   gscale[m] = sum_k gz[k, m] * eps[k, m]  -  (sum_k glp[k, r(m)]) / scale[m]
 * gz or glp may be NULL.  u as in the forward call. */
int zs_logistic_sample_logprob_bwd_f32(const float* scale, const float* u,
                                       uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                                       const float* gz, const float* glp,
                                       int64_t glp_stride_k, int64_t glp_stride_r,
                                       float* gloc, float* gscale,
                                       int64_t K, int64_t M, int64_t D, void* stream);

/* L2  Logistic log-prob of a given value (logistic.py:69-83); problem [K, R, D], periodic operands. */
int zs_logistic_logprob_f32(const float* x, int64_t Px, const float* loc, int64_t Pm,
                            const float* scale, int64_t Ps, float* lp,
                            int64_t K, int64_t R, int64_t D,
                            int64_t lp_stride_k, int64_t lp_stride_r, void* stream);

/* Backward of L2, element-wise partials of size K*R*D each (any may be NULL), with g = glp[k, r],
 * t = (x - loc)/scale and h = tanh(t/2) (= 1 - 2*sigmoid(-t)):
 *   gx = -g*h/scale,  gloc = +g*h/scale,  gscale = g*(h*t - 1)/scale. */
int zs_logistic_logprob_bwd_f32(const float* x, int64_t Px, const float* loc, int64_t Pm,
                                const float* scale, int64_t Ps,
                                const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                float* gx, float* gloc, float* gscale,
                                int64_t K, int64_t R, int64_t D, void* stream);

/* Backward of L2 reduced over the K axis for parameters of period R*D (loc, scale of shape [R, D] repeated K times: a
 * Logistic latent under a non-reparameterised / importance-weighted estimator, logistic.py:73-77).  x has full size
 * [K, R, D].   gloc[r, d] = sum_k ...,  gscale[r, d] = sum_k ...;  gx (full size) optional. */
int zs_logistic_logprob_bwd_ksum_f32(const float* x, const float* loc, const float* scale,
                                     const float* glp, int64_t glp_stride_k, int64_t glp_stride_r,
                                     float* gx, float* gloc, float* gscale,
                                     int64_t K, int64_t R, int64_t D, void* stream);

/* ---------------------------------------------------------------------------
 * U1  Uniform sample (zhusuan/distributions/uniform.py:51-70).  N elements, low / high periodic.
 *   reparam != 0:  cache = u,                       out = u * (high - low) + low      (uniform.py:66-70)
 *   reparam == 0:  cache = v = u*(high - low)+low,  out = v * (high - low) + low      (uniform.py:63-64,70:
 *                  the reference scales the already scaled draw a second time; kept as is)
 * `cache` is what the reference stores in sample_cache (uniform.py:69).  u == NULL: Philox draw as in L1.
 * -------------------------------------------------------------------------*/
int zs_uniform_sample_f32(const float* low, int64_t Pl, const float* high, int64_t Ph, const float* u,
                          uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                          float* out, float* cache, int64_t N, int reparam, void* stream);

/* U2  Uniform log-prob of a given value (uniform.py:72-85 -> torch.distributions.Uniform.log_prob):
 *   lp[k, r] = sum_d ( log( (low <= x) * (high > x) ) - log(high - low) )      (-inf outside the support)
 * Argument / support validation (ValueError in the reference) is the caller's job. */
int zs_uniform_logprob_f32(const float* x, int64_t Px, const float* low, int64_t Pl,
                           const float* high, int64_t Ph, float* lp,
                           int64_t K, int64_t R, int64_t D,
                           int64_t lp_stride_k, int64_t lp_stride_r, void* stream);

/* Uniform (0, 1) draws of the kernels' Philox stream: out[i] = u01(word i%4 of group i/4). */
int zs_philox_uniform_f32(float* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state,
                          void* stream);

/* ---------------------------------------------------------------------------
 * R1  Score-function (REINFORCE / NVIL) epilogue of the ELBO, one launch.
 * Replaces the ~20 whole-tensor scalar ops of ELBO.reinforce (zhusuan/variational/elbo.py:163-238).
 * n values of log p(x,z) and log q(z|x); baseline (optional) has period Pb (1 or n).
 *
 *   l0_i = logp_i - logq_i
 *   variance_reduction != 0:
 *     resid_i = l0_i - baseline_i,  l_i = resid_i                    (baseline != NULL, elbo.py:209-215)
 *     bc = mean_i(l_i) if do_mean else l_0 (n must be 1)              (elbo.py:217-220)
 *     moving_mean -= (moving_mean - bc) * (1 - decay);  local_step += 1
 *     moving_mean /= 1 - decay^local_step                             (in place, every call: elbo.py:221-224)
 *     l_i -= moving_mean
 *   signal_i = l_i                                                    (the detached learning signal)
 *   c_i = -(logp_i + l_i * logq_i) + 0.5 * resid_i^2                  (second term only with a baseline)
 *   cost[0] = mean_i(c_i) if do_mean, else cost[i] = c_i
 *
 * moving_mean (float, 1 element) and local_step (int32, 1 element) are DEVICE state, read and written by the kernel
 * (the module buffers of elbo.py:45-49), so the call is hipGraph-capturable.  resid may be NULL when baseline is.
 * Derivatives for the caller: d c_i / d logp_i = -1,  d c_i / d logq_i = -signal_i,  d c_i / d baseline_i = -resid_i.
 * workspace / ticket (optional; >= ZS_LJ_WORKSPACE doubles and one zero-initialised device word handed back at zero, as
 * for LJ1): with them, vectors of more than 16 384 elements are spread over many workgroups (one pass that also yields the
 * moving mean, then an element-wise launch that centres the learning signal: ~12 us at 10^6 elements); without them one
 * workgroup walks the vector (300 us at 10^6 elements).
 * -------------------------------------------------------------------------*/
int zs_reinforce_f32(const float* logp, const float* logq, const float* baseline, int64_t Pb, int64_t n,
                     int variance_reduction, int do_mean, double decay,
                     float* moving_mean, int32_t* local_step,
                     float* signal, float* cost, float* resid,
                     double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);

/* ---------------------------------------------------------------------------
 * S1  Scalar ELBO epilogue: out[0] = sum_t coef[t] * sum_i rows_t[i]  for up to ZS_MAX_TERMS dense vectors.
 * Replaces, for objectives whose nodes all reduce to scalars (the VAE and BNN callers: reduce_mean_dims /
 * reduce_sum_dims / multiplier of StochasticTensor.log_prob, zhusuan/framework/stochastic_tensor.py:160-181, then
 * ELBO.log_joint + ELBO.sgvb, zhusuan/variational/elbo.py:58-79,155-161), the per-node mean / sum / multiply launches
 * and the scalar adds, subtract and negation: one launch forward.  coef_out (n_terms values, optional) receives the
 * coefficients, so that backward is one multiply of that vector by the incoming gradient (d out / d rows_t[i] = coef[t]).
 * Unused slots: rows == NULL (n ignored).  One workgroup; sums are accumulated in double in a fixed order.
 * -------------------------------------------------------------------------*/
#define ZS_MAX_TERMS 6
int zs_scalar_objective_f32(const float* r0, int64_t n0, double c0, const float* r1, int64_t n1, double c1,
                            const float* r2, int64_t n2, double c2, const float* r3, int64_t n3, double c3,
                            const float* r4, int64_t n4, double c4, const float* r5, int64_t n5, double c5,
                            float* out, float* coef_out, void* stream);

/* ---------------------------------------------------------------------------
 * A1  Adam update of up to ZS_ADAM_MAX_TENSORS parameter tensors, one launch.  The callers' optimizer is
 * torch.optim.Adam(model.parameters(), lr) with its defaults (reference examples variational_autoencoder/vae_mnist.py:104,
 * iwae.py:141, bayesian_neural_nets/bnn_vi.py:135: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad); this is
 * that update.  The tensors form one flat index space: tensor s covers [starts[s], starts[s+1]) and is read / written at
 * param_ptrs[s], its gradient read at grad_ptrs[s]; both moments are flat [n].  A tensor whose gradient pointer is NULL
 * has no gradient this step and is LEFT ALONE, exactly like a parameter with `grad is None` in torch.optim.Adam: its
 * values, its moments and its step count do not change.  (The gradients of a data-parallel bucket are consecutive slices
 * of one buffer, zhusuan/dataparallel.py: never NULL.)
 * param_ptrs, grad_ptrs (n_tensors device pointers each) and starts (n_tensors + 1 values, starts[0] = 0,
 * starts[n_tensors] = n, strictly ascending) are HOST arrays, copied into the kernel arguments.  Per tensor s:
 *   t = steps[s] + 1
 *   g = grad_scale * grad[i]                      (grad_scale: the 1/world of the gradient mean, folded into the read)
 *   m[i] += (1 - beta1) * (g - m[i]);   v[i] = beta2 * v[i] + (1 - beta2) * g * g
 *   param[i] -= lr / (1 - beta1^t) * m[i] / (sqrt(v[i]) / sqrt(1 - beta2^t) + eps)
 *   steps[s] = t
 * steps (int64, n_tensors elements) is DEVICE state, so the call is hipGraph-capturable; hyper (optional) is a DEVICE
 * array of four doubles {lr, beta1, beta2, eps} that, when given, replaces the by-value arguments -- a captured launch then
 * follows learning-rate changes written into it between replays; ticket is a zero-initialised device word owned by the
 * caller (one per call that may run concurrently), handed back at zero.  More than ZS_ADAM_MAX_TENSORS tensors:
 * ZS_ENOTSUP (call once per 32). */
#define ZS_ADAM_MAX_TENSORS 32
int zs_adam_step_f32(float* const* param_ptrs, const float* const* grad_ptrs, const int64_t* starts, int n_tensors,
                     float* exp_avg, float* exp_avg_sq, int64_t* steps, uint32_t* ticket, int64_t n, double lr,
                     double beta1, double beta2, double eps, double grad_scale, const double* hyper, void* stream);




/* ---------------------------------------------------------------------------
 * LJ1  The scalar log-joint objective in ONE launch (forward) and ONE launch (backward).
 * For objectives whose nodes all reduce to scalars (the VAE and BNN callers) the reference evaluates one log-prob per
 * node in a Python loop -- ELBO.log_joint, zhusuan/variational/elbo.py:58-79, over StochasticTensor.log_prob,
 * zhusuan/framework/stochastic_tensor.py:160-181 (dist.log_prob, mean / sum over the reduce dims, multiplier) -- and
 * combines them in ELBO.sgvb (elbo.py:155-161).  Every such node contributes  coef_t * sum_i logprob_t(i)  with
 * coef_t = -+ multiplier / prod(sizes of the mean axes); this call evaluates ALL terms and their weighted sum:
 *     out[0] = sum_t coef_t * sum_{i < n_t} term_t(i)
 *   ZS_LJ_ROWS              term(i) = x[i]                      values that exist already (the log q rows K1 produced)
 *   ZS_LJ_NORMAL            term(i) = -0.5*log(2*pi) - log(b) - 0.5*exp(-2*log(b))*(x - a)^2      (normal.py:109-126)
 *   ZS_LJ_NORMAL_LOGSTD     the same with b = exp(given b)                                         (normal.py:56)
 *   ZS_LJ_BERNOULLI         term(i) = x*log(a + 1e-8) + (1 - x)*log((1 - a) + 1e-8)                (bernoulli.py:84-95)
 *   ZS_LJ_BERNOULLI_LOGITS  the same with a = sigmoid(given a)                                     (bernoulli.py:50)
 * Operands are periodic (x[i % px], a[i % pa], b[i % pb]; every period divides n).  `terms` is a HOST array, copied
 * into the kernel arguments.  Sums are accumulated per thread in the operands' precision over <= 16 elements, from there
 * on in double, and combined in a fixed order (per-workgroup partials in `workspace`, the last workgroup -- found with
 * `ticket`, a zero-initialised device word handed back at zero -- adds them by index): deterministic.
 * workspace: >= ZS_LJ_WORKSPACE doubles.
 *
 * Backward (zs_logjoint_scalar_bwd): with g = gout[0] (DEVICE scalar: the incoming gradient of out) every requested
 * gradient in one launch --  gx / ga / gb (px / pa / pb elements; NULL = not wanted) = g * coef * d term / d operand, summed
 * over the repeats of a periodic operand (in index order: deterministic).  For ZS_LJ_NORMAL_LOGSTD gb is d/d log std;
 * for the Bernoulli families gx (the observation) is not provided (ZS_ENOTSUP); ZS_LJ_ROWS terms have no outputs here:
 * their gradient is the scalar gcoef[t] = g * coef_t, written for every term (gcoef: n_terms values, optional).
 * -------------------------------------------------------------------------*/
#define ZS_LJ_MAX_TERMS 8
#define ZS_LJ_WORKSPACE 8192
#define ZS_LJ_ROWS 0
#define ZS_LJ_NORMAL 1
#define ZS_LJ_NORMAL_LOGSTD 2
#define ZS_LJ_BERNOULLI 3
#define ZS_LJ_BERNOULLI_LOGITS 4
typedef struct zs_lj_term {
  int32_t family;            /* ZS_LJ_* */
  int32_t reserved;
  int64_t n;                 /* elements of the term's full (broadcast) problem */
  const void* x; int64_t px; /* value; ZS_LJ_ROWS: the values to add up */
  const void* a; int64_t pa; /* mean / probs / logits (unused by ZS_LJ_ROWS) */
  const void* b; int64_t pb; /* std or log std (Normal families only) */
  double coef;
  void* gx; void* ga; void* gb;   /* backward only */
} zs_lj_term;
int zs_logjoint_scalar_f32(const zs_lj_term* terms, int n_terms, float* out, double* workspace, int64_t workspace_len,
                           uint32_t* ticket, void* stream);
int zs_logjoint_scalar_bwd_f32(const zs_lj_term* terms, int n_terms, const float* gout, float* gcoef, double* workspace,
                               int64_t workspace_len, uint32_t* ticket, void* stream);

/* ---------------------------------------------------------------------------
 * MS1  K1 for SEVERAL Normal nodes in one launch: the variational net of a model with more than one latent node (the
 * BNN's weight matrices, bnn_vi.py:83-93) draws them one after the other in the reference (ELBO.forward re-reads every
 * node's .tensor, elbo.py:122); here all of them are one launch forward and one backward.  Per term exactly
 * zs_normal_sample_logprob / _bwd (same formulas, same Philox stream: term t draws with call id  base + terms[t].offset,
 * group = flat index / 4), meant for the launch-bound small shapes (a wavefront per row; use K1 for a single large node).
 * `terms` is a HOST array.  rng_state / rng_used: as for K1, shared by all terms ({seed, base}).
 * Backward: gz / glp (either may be NULL) -> gmu, gsigma per term (reparameterised nodes: normal.py:104-105); gz2 (may be
 * NULL): a second gradient w.r.t. the same sample, added to gz element by element (a draw that feeds the model AND its own
 * prior term collects two gradients; autograd would add them with a launch of its own).
 * -------------------------------------------------------------------------*/
#define ZS_MS_MAX_TERMS 8
typedef struct zs_ms_term {
  const void* mu; const void* sigma; const void* eps;  /* [M], [M], [K, M] or NULL (in-kernel Philox) */
  void* z; void* lp;                                   /* [K, M]; row results (NULL: sample only) */
  int64_t K, M, D, lp_stride_k, lp_stride_r;
  uint64_t offset;                                     /* call id of this node's draw, relative to the base */
  int32_t sigma_is_logstd;
  int32_t reserved;
  const void* gz; const void* glp;                     /* backward only */
  int64_t glp_stride_k, glp_stride_r;
  void* gmu; void* gsigma;
  const void* gz2;                                     /* backward only, optional */
} zs_ms_term;
int zs_normal_sample_logprob_multi_f32(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                       uint64_t* rng_used, void* stream);
int zs_normal_sample_logprob_multi_bwd_f32(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state,
                                           void* stream);

/* ---------------------------------------------------------------------------
 * PL1  The particle-batched dense layer of the BNN caller (reference
 * examples/bayesian_neural_nets/bnn_vi.py:27-48): every particle k has its own weight matrix w[k] = [n_out, n_in + 1]
 * whose last column is the bias.  The reference repeats w over the batch (a 115 MB copy at config 5), appends a column
 * of ones to h, multiplies, divides by sqrt(n_in + 1) and applies ReLU -- five whole-tensor ops; this is one kernel:
 *     out[k, b, o] = act( ( sum_{i < n_in} h[k, b, i] * w[k, o, i]  +  w[k, o, n_in] ) / sqrt(n_in + 1) )
 * h is [K, B, n_in] (h_stride_k = B * n_in) or shared by the particles, [B, n_in] (h_stride_k = 0: the first layer's
 * input x, which the reference repeats K times, bnn_vi.py:27).  relu != 0: act = max(., 0), else identity.
 * n_in <= 255, n_out <= 256 (a particle's weights are staged in LDS); larger layers: ZS_ENOTSUP (use a batched GEMM).
 * Backward, one launch:  gpre = gout * (out > 0) [relu],
 *     gh[k, b, i] = sum_o gpre[k, b, o] * w[k, o, i] / sqrt(n_in + 1)          (optional; [K, B, n_in] also when h is shared:
 *                                                                                the caller sums over k if it needs d/dx)
 *     gw[k, o, i] = sum_b gpre[k, b, o] * h[k, b, i] / sqrt(n_in + 1),   gw[k, o, n_in] = sum_b gpre[k, b, o] / sqrt(n_in + 1)
 * The batch sum of gw is formed per tile of 16 to 64 rows (the kernel picks the tile so that the launch fills the chip) and
 * the tiles are added in tile order by the last workgroup of each particle (deterministic): workspace holds the tile
 * partials, >= K * ceil(B / 16) * n_out * (n_in + 1) elements always suffices; tickets is K zero-initialised device words
 * owned by the caller, handed back at zero.
 * -------------------------------------------------------------------------*/
int zs_particle_linear_f32(const float* h, int64_t h_stride_k, const float* w, float* out,
                           int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu, void* stream);
int zs_particle_linear_bwd_f32(const float* h, int64_t h_stride_k, const float* w, const float* out, const float* gout,
                               float* gh, float* gw, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu,
                               float* workspace, int64_t workspace_len, uint32_t* tickets, void* stream);

/* ---------------------------------------------------------------------------
 * PM1  The whole particle-batched network of the BNN caller (bnn_vi.py:27-48: the loop over layers, each PL1 above, ReLU after
 * every layer but the last) in one launch forward and one backward.  layers[l] describes layer l: w [K, n_out, n_in + 1],
 * out [K, B, n_out] (written by the forward call -- every layer's activation, the last one is the network's output -- and read
 * back by the backward call), gw [K, n_out, n_in + 1] (written by the backward call); layers[l].n_in == layers[l - 1].n_out.
 * x: [B, n_0] shared by the particles (x_stride_k = 0) or [K, B, n_0] (x_stride_k = B * n_0).  Backward: gout is the gradient
 * w.r.t. the last layer's output [K, B, n_L]; gx [K, B, n_0] optional (NULL: not formed).  Outputs equal a chain of PL1 calls bit
 * for bit.  1 <= n_layers <= ZS_PM_MAX_LAYERS, widths as PL1, and the LDS layouts of both directions must fit (ZS_ENOTSUP
 * otherwise: use PL1 per layer).  workspace: >= K * ceil(B / 16) * sum_l (n_out_l * (n_in_l + 1) + 3) elements always suffices;
 * tickets: K zero-initialised device words, handed back at zero.  The table is a HOST array (copied into the kernel arguments).
 * -------------------------------------------------------------------------*/
#define ZS_PM_MAX_LAYERS 4
typedef struct zs_pm_layer {
  const void* w;
  void* out;
  void* gw;      /* backward only */
  int64_t n_in, n_out;
} zs_pm_layer;
int zs_particle_mlp_f32(const float* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, int64_t K, int64_t B,
                        void* stream);
int zs_particle_mlp_bwd_f32(const float* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, const float* gout,
                            float* gx, int64_t K, int64_t B, float* workspace, int64_t workspace_len, uint32_t* tickets,
                            void* stream);

/* ---------------------------------------------------------------------------
 * CS1  Column sums of a row-major [rows, cols] matrix: out[c] = sum_r x[r, c].  The bias gradient of the callers' dense
 * layers (torch.nn.Linear in the reference's examples, variational_autoencoder/vae_mnist.py:22-28, iwae.py:40-47:
 * grad_bias = grad_output.sum(0)) -- the one reduction of their backward pass that is not a GEMM; caller-side glue like
 * PL1, not part of the distribution / objective path.  Deterministic (row chunks combined in chunk order).
 * workspace: >= 128 * (cols + 256) elements; tickets: n_tickets >= ceil(cols / 64) zero-initialised device words,
 * handed back at zero (both unused, and may be NULL, when the matrix has fewer than 64 rows).
 * -------------------------------------------------------------------------*/
int zs_column_sum_f32(const float* x, float* out, int64_t rows, int64_t cols, float* workspace, int64_t workspace_len,
                      uint32_t* tickets, int64_t n_tickets, void* stream);

/* ---------------------------------------------------------------------------
 * AB1  Backward of a dense layer's activation and its bias gradient in one pass over the gradient:
 *   gpre[r, c] = g[r, c] * act'(y[r, c]),   gbias[c] = sum_r gpre[r, c]
 * with y the layer's ACTIVATED output: ZS_ACT_RELU g * [y > 0] (torch's threshold_backward), ZS_ACT_SIGMOID
 * g * (1 - y) * y (torch's sigmoid_backward).  The callers' MLPs are Linear -> ReLU (-> ... -> Sigmoid) stacks
 * (variational_autoencoder/vae_mnist.py:22-28,44-48, iwae.py:40-47,68-75); torch runs the activation's backward and the
 * bias reduction as two passes.  gpre may be g itself (in place).  Workspace / tickets / determinism as CS1; caller-side
 * glue like CS1 and PL1.
 * -------------------------------------------------------------------------*/
#define ZS_ACT_NONE 0
#define ZS_ACT_RELU 1
#define ZS_ACT_SIGMOID 2
int zs_dense_act_bwd_f32(const float* g, const float* y, int act, float* gpre, float* gbias, int64_t rows, int64_t cols,
                         float* workspace, int64_t workspace_len, uint32_t* tickets, int64_t n_tickets, void* stream);

/* ---------------------------------------------------------------------------
 * PR1  RMSE of the particle-mean prediction, the diagnostic of the BNN caller's forward pass
 * (examples/bayesian_neural_nets/bnn_vi.py:84-87: y_pred = mean(y_mean, 0); rmse = sqrt(mean((y - y_pred) ** 2))):
 *   out[0] = sqrt( (1 / B) * sum_b (y[b] - (1 / K) * sum_k pred[k * B + b])^2 )
 * pred [K, B] contiguous, y [B].  NaN when K == 0 or B == 0 (torch's mean of nothing).  Caller-side glue like PL1.
 * workspace (DOUBLE, for both precisions): >= 1024 elements; ticket: one zero-initialised device word, handed back at
 * zero; both unused (may be NULL) when B <= 4096.  Deterministic.
 * -------------------------------------------------------------------------*/
int zs_particle_rmse_f32(const float* pred, const float* y, float* out, int64_t K, int64_t B, double* workspace,
                         int64_t workspace_len, uint32_t* ticket, void* stream);

/* ---------------------------------------------------------------------------
 * float64 twins.  The reference accepts float64 parameters for Normal / Bernoulli
 * (zhusuan/distributions/utils.py:5,57-64); every entry point above exists with the suffix _f64,
 * identical argument meaning, double* instead of float*.  They are plain (untuned) kernels: none of the
 * benchmark configurations uses float64.  Draws widen the same Philox / Box-Muller fp32 stream.
 * -------------------------------------------------------------------------*/
int zs_normal_sample_logprob_pair_f64(const double* mu, const double* sigma, uint64_t seed, uint64_t offset, const uint64_t* rng_state, double* z, double* lp, int64_t K, int64_t M, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, int sigma_is_logstd, uint64_t* rng_used, void* stream);
int zs_normal_sample_logprob_f64(const double* mu, const double* sigma, const double* eps, uint64_t seed, uint64_t offset, const uint64_t* rng_state, double* z, double* lp, int64_t K, int64_t M, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, int sigma_is_logstd, uint64_t* rng_used, void* stream);
int zs_normal_sample_logprob_bwd_f64(const double* sigma, const double* eps, uint64_t seed, uint64_t offset, const uint64_t* rng_state, const double* gz, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gmu, double* gsigma, int64_t K, int64_t M, int64_t D, int sigma_is_logstd, void* stream);
int zs_normal_logprob_f64(const double* x, int64_t Px, const double* mu, int64_t Pm, const double* sigma, int64_t Ps, double* lp, int64_t K, int64_t R, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, int sigma_is_logstd, void* stream);
int zs_normal_logprob_bwd_f64(const double* x, int64_t Px, const double* mu, int64_t Pm, const double* sigma, int64_t Ps, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gx, double* gmu, double* gsigma, int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream);
int zs_normal_logprob_bwd_ksum_f64(const double* x, const double* mu, const double* sigma, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gx, double* gmu, double* gsigma, int64_t K, int64_t R, int64_t D, int sigma_is_logstd, void* stream);
int zs_bernoulli_logprob_f64(const double* p, const double* x, int64_t Px, double* lp, int64_t K, int64_t R, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, void* stream);
int zs_bernoulli_logprob_bwd_f64(const double* p, const double* x, int64_t Px, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gp, int64_t K, int64_t R, int64_t D, void* stream);
int zs_bernoulli_logprob_bwd_x_f64(const double* p, int from_logits, int64_t Px, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, const double* gscale, int64_t gscale_stride, double* gx, int64_t K, int64_t R, int64_t D, void* stream);
int zs_bernoulli_logits_logprob_f64(const double* logits, const double* x, int64_t Px, double* lp, double* probs_out, int64_t K, int64_t R, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, void* stream);
int zs_bernoulli_logits_logprob_bwd_f64(const double* logits, const double* x, int64_t Px, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* glogits, int64_t K, int64_t R, int64_t D, void* stream);
int zs_bernoulli_sample_f64(const double* p, int64_t Pp, double* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state, void* stream);
int zs_iw_reduce_f64(const double* logp, int64_t ld_p, const double* logq, int64_t ld_q, int64_t B, int64_t K, int estimator, double* cost_b, double* bound_b, double* coef_p, double* coef_q, void* stream);
int zs_iw_objective_f64(const double* logp_a, int64_t ld_a, const double* logp_b, int64_t ld_b, const double* logq, int64_t ld_q, int64_t B, int64_t K, int estimator, int want_mean, double* cost_b, double* bound_b, double* coef, double* mean_cost, double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);
int zs_bernoulli_iw_objective_f64(const double* p, int from_logits, const double* x, int64_t Px, int64_t K, int64_t R, int64_t D, const double* z, const double* pmu, int64_t Pm, const double* psigma, int64_t Ps, int64_t Dz, int psigma_is_logstd, const double* rows_a, int64_t ld_a, const double* logq, int64_t ld_q, int estimator, int want_mean, double* lp_x, double* lp_z, double* cost_b, double* bound_b, double* coef, double* mean_cost, uint64_t* acc, void* stream);
int zs_bernoulli_iw_objective_bwd_f64(const double* p, int from_logits, const double* x, int64_t Px, int64_t K, int64_t R, int64_t D, const double* coef, const double* gout, int64_t gout_stride, double* gp, const double* zq, const double* qmu, const double* qsigma, int64_t Dq, int qsigma_is_logstd, double* gqmu, double* gqsigma, void* stream);
int zs_log_mean_exp_f64(const double* x, int64_t ld, int64_t B, int64_t K, double* out, void* stream);
int zs_philox_normal_f64(double* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state, void* stream);
int zs_logistic_sample_logprob_f64(const double* loc, const double* scale, const double* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, double* z, double* lp, int64_t K, int64_t M, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, uint64_t* rng_used, void* stream);
int zs_logistic_sample_logprob_bwd_f64(const double* scale, const double* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, const double* gz, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gloc, double* gscale, int64_t K, int64_t M, int64_t D, void* stream);
int zs_logistic_logprob_f64(const double* x, int64_t Px, const double* loc, int64_t Pm, const double* scale, int64_t Ps, double* lp, int64_t K, int64_t R, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, void* stream);
int zs_logistic_logprob_bwd_f64(const double* x, int64_t Px, const double* loc, int64_t Pm, const double* scale, int64_t Ps, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gx, double* gloc, double* gscale, int64_t K, int64_t R, int64_t D, void* stream);
int zs_logistic_logprob_bwd_ksum_f64(const double* x, const double* loc, const double* scale, const double* glp, int64_t glp_stride_k, int64_t glp_stride_r, double* gx, double* gloc, double* gscale, int64_t K, int64_t R, int64_t D, void* stream);
int zs_uniform_sample_f64(const double* low, int64_t Pl, const double* high, int64_t Ph, const double* u, uint64_t seed, uint64_t offset, const uint64_t* rng_state, double* out, double* cache, int64_t N, int reparam, void* stream);
int zs_uniform_logprob_f64(const double* x, int64_t Px, const double* low, int64_t Pl, const double* high, int64_t Ph, double* lp, int64_t K, int64_t R, int64_t D, int64_t lp_stride_k, int64_t lp_stride_r, void* stream);
int zs_philox_uniform_f64(double* out, int64_t N, uint64_t seed, uint64_t offset, const uint64_t* rng_state, void* stream);
int zs_reinforce_f64(const double* logp, const double* logq, const double* baseline, int64_t Pb, int64_t n, int variance_reduction, int do_mean, double decay, float* moving_mean, int32_t* local_step, double* signal, double* cost, double* resid, double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);
int zs_scalar_objective_f64(const double* r0, int64_t n0, double c0, const double* r1, int64_t n1, double c1, const double* r2, int64_t n2, double c2, const double* r3, int64_t n3, double c3, const double* r4, int64_t n4, double c4, const double* r5, int64_t n5, double c5, double* out, double* coef_out, void* stream);
int zs_logjoint_scalar_f64(const zs_lj_term* terms, int n_terms, double* out, double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);
int zs_logjoint_scalar_bwd_f64(const zs_lj_term* terms, int n_terms, const double* gout, double* gcoef, double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);
int zs_normal_sample_logprob_multi_f64(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state, uint64_t* rng_used, void* stream);
int zs_normal_sample_logprob_multi_bwd_f64(const zs_ms_term* terms, int n_terms, uint64_t seed, const uint64_t* rng_state, void* stream);
int zs_particle_linear_f64(const double* h, int64_t h_stride_k, const double* w, double* out, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu, void* stream);
int zs_particle_linear_bwd_f64(const double* h, int64_t h_stride_k, const double* w, const double* out, const double* gout, double* gh, double* gw, int64_t K, int64_t B, int64_t n_in, int64_t n_out, int relu, double* workspace, int64_t workspace_len, uint32_t* tickets, void* stream);
int zs_column_sum_f64(const double* x, double* out, int64_t rows, int64_t cols, double* workspace, int64_t workspace_len, uint32_t* tickets, int64_t n_tickets, void* stream);
int zs_dense_act_bwd_f64(const double* g, const double* y, int act, double* gpre, double* gbias, int64_t rows, int64_t cols, double* workspace, int64_t workspace_len, uint32_t* tickets, int64_t n_tickets, void* stream);
int zs_particle_rmse_f64(const double* pred, const double* y, double* out, int64_t K, int64_t B, double* workspace, int64_t workspace_len, uint32_t* ticket, void* stream);
int zs_particle_mlp_f64(const double* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, int64_t K, int64_t B, void* stream);
int zs_particle_mlp_bwd_f64(const double* x, int64_t x_stride_k, const zs_pm_layer* layers, int n_layers, const double* gout, double* gx, int64_t K, int64_t B, double* workspace, int64_t workspace_len, uint32_t* tickets, void* stream);
int zs_adam_step_f64(double* const* param_ptrs, const double* const* grad_ptrs, const int64_t* starts, int n_tensors, double* exp_avg, double* exp_avg_sq, int64_t* steps, uint32_t* ticket, int64_t n, double lr, double beta1, double beta2, double eps, double grad_scale, const double* hyper, void* stream);

/* ---------------------------------------------------------------------------
 * Per-kernel timing for the benchmark harness (no reference counterpart).
 * While enabled, every launch made by the entry points above is issued with a start/stop HIP event
 * pair bound to the dispatch on its stream (hipExtLaunchKernelGGL), i.e. the kernel's own duration.
 *   zs_prof_enable(1) clears earlier records and starts recording; zs_prof_enable(0) stops.
 *   zs_prof_kernel_id("zs_bernoulli_logprob_f32") -> id of that entry point's kernels (or ZS_EINVAL).
 *   zs_prof_query(id, &total_ms, &min_ms, &max_ms, &count) waits for the recorded launches of that
 *   entry point and returns their summed / extreme durations.
 * -------------------------------------------------------------------------*/
int zs_prof_enable(int on);
int zs_prof_kernel_id(const char* entry_point);
int zs_prof_query(int kernel_id, double* total_ms, double* min_ms, double* max_ms, int64_t* count);
/* The individual durations (ms, launch order) of the recorded launches of one entry point: writes the first `capacity`
 * of them to out_ms and returns how many were recorded (so the caller can take a median over back-to-back launches). */
int64_t zs_prof_durations(int kernel_id, double* out_ms, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* ZS_HIP_H */
