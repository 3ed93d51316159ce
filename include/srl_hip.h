/*
 * srl_hip.h — C ABI of libsrlhip.so: the MI355X (gfx950) kernels behind the rollout -> GAE -> PPO
 * hot path of openpsi-project/srl.
 *
 * The reference has no FFI: its hot path is Python calling torch ops.  Each entry point below
 * therefore replaces a *sequence of torch ops* inside one reference function; the reference
 * file:line it stands in for is cited per function (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes binding and the three-line plugin registration a maintainer adds.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is enqueued
 *     asynchronously on it, nothing synchronises the device;
 *   - tensors are dense, row-major; sample leaves are time-major [T, B, ...] exactly as the
 *     reference's SampleBatch stores them (api/trainer.py:14-82, base/buffer.py:120-121);
 *   - flags (done / truncated / on_reset / masks) are uint8 as on the wire
 *     (distributed/system/actor_worker.py:278-281) — they are NOT widened to float32 like the
 *     reference's prefetcher does (api/trainer.py:217);
 *   - return value: 0 on success, a negative errno-style code on failure (never throws);
 *     srl_last_error() returns a thread-local message for the last failure;
 *   - no global state, no allocation: callers own every buffer (workspaces are explicit).
 */
#ifndef SRL_HIP_H_
#define SRL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRL_HIP_ABI_VERSION 18

int srl_abi_version(void);
const char* srl_last_error(void);
/* Number of compute units / device name of the current device (host query helpers). */
int srl_device_info(int* num_cus, int* lds_bytes_per_cu, char* name, int name_len);
/* Host-side launch counters per kernel family since the last reset (process-wide; test / benchmark instrumentation:
 * "did this step run on the kernels the benchmark times?").  out[0] gemm3_kernel (float32 operands as exact bf16
 * pieces on the bf16 matrix cores), out[1] gemm_kernel (float32 MFMA), out[2] skinny kernels, out[3] / out[4] the
 * first layer's obs_fwd_bf16_kernel / obs_bwd_bf16_kernel, out[5] the two-plane f16 variant of gemm3_kernel;
 * entries beyond 8 are zero.  reset != 0 zeroes the counters after reading.  No reference counterpart. */
int srl_dispatch_counts(int64_t* out, int n, int reset);
/* The same per INSTANTIATION: keys[i] = family << 56 | p0 << 40 | p1 << 24 | p2 << 8 | flags, counts[i] launches; returns the
 * number of distinct keys seen (at most 256; the first `cap` are written).  gemm3 / gemm2h / gemm_f32: p0 x p1 = the tile
 * (rows x columns of the output), p2 = the split-K factor, flags = A mode << 5 | B mode << 2 | A k-major << 1 | B k-major (modes:
 * 0 plain, 1 convolution patches, 2 data-gradient patches, 3 / 4 first-layer frames); h2: p0 = 1 convolution / 2 weight gradient
 * / 3 dense product, p1 = the kind (srl_h2_conv / srl_h2_wgrad) or the 32-channel blocks per workgroup, p2 = the depth of the
 * LDS-DMA ring; first layer: p0 = the K depth, p1 = 1 with the h2 output, p2 = the position split.  Tests assert on these that
 * a benchmark-sized call ran the instantiation the benchmark times. */
int srl_dispatch_tiles(uint64_t* keys, int64_t* counts, int cap, int reset);

/* ------------------------------------------------------------------------------------------------
 * GAE / V-trace reverse scan, fused with value masking, the return, and the advantage statistics.
 * Replaces: MultiAgentPPO._compute_adv_and_value_target (legacy/algorithm/ppo/mappo.py:118-144)
 *           -> gae_trace (legacy/algorithm/modules/gae.py:8-97), and the three reductions of
 *           masked_normalization (legacy/algorithm/modules/utils.py:38-57).
 *
 *   v'[t]   = value[t] * (1 - done[t])                                    (mappo.py:124)
 *   delta_t = reward[t] + gamma * v'[t+1] * (1 - on_reset[t+1]) - v'[t]   (gae.py:63)
 *   m_t     = gamma * lambda * (1 - on_reset[t+1]) * (1 - truncated[t+1]) (gae.py:87)
 *   V-trace (imp_ratio != NULL): delta_t *= min(rho, ratio_t); m_t *= min(c, ratio_t)  (gae.py:64,88)
 *   adv[t]  = delta_t + m_t * adv[t+1], adv[T] = 0   — float64 arithmetic, float32 result (gae.py:46,97)
 *   ret[t]  = adv[t] + v'[t]                          — float32 (mappo.py:143)
 *   stats  += { sum mask, sum adv*mask, sum (adv*mask)^2 } in float64, mask[t] = 1 - on_reset[t+1]
 *             (the loss mask of mappo.py:260-261 for burn_in = 0, bootstrap = 1)
 *
 * reward [>=T, B, Nc] (rows 0..T-1 read), value [T+1, B, Nc], done/truncated/on_reset [T+1, B, 1],
 * imp_ratio [T, B, 1] or NULL; adv, ret [>=T, B, Nc] (rows 0..T-1 written; the caller provides the
 * zero row T that mappo.py:254-256 pads).
 * gamma_t / lambda_t: per-step discount / lambda tensors [T, B, 1] float32 (widened to float64 like
 * gae.py:51-60) or NULL, in which case the scalar `gamma` / `lambda` (a double, like the reference's
 * python float) is used.
 * stats: float64[3] or NULL.  With workspace == NULL it is zeroed by this call (one memset launch)
 * and accumulated with float64 atomics (order-dependent in the last bit).  With a workspace of
 * srl_gae_scan_workspace_bytes(B, Nc) bytes -- zeroed ONCE by the caller, then owned by this entry
 * point; one workspace per stream that may run it concurrently -- the workgroups leave partial sums
 * there and the last one to finish adds them in a fixed order and overwrites stats: no extra launch,
 * bitwise reproducible sums (batches below 98 304 columns; wider ones stream for hundreds of microseconds
 * and keep the memset + atomics, which is faster there).
 */
int srl_gae_scan(void* stream, const float* reward, const float* value, const uint8_t* done,
                 const uint8_t* truncated, const uint8_t* on_reset, const float* imp_ratio,
                 const float* gamma_t, const float* lambda_t, int T, int B, int Nc, double gamma,
                 double lambda, double rho, double c, float* adv, float* ret, double* stats,
                 void* workspace);
long srl_gae_scan_workspace_bytes(int B, int Nc);

/* Masked statistics {n, sum x*mask, sum (x*mask)^2} in float64 (utils.py:38-57).
 * mask: uint8[n] or NULL (no mask: n = count); mask_invert != 0 means mask = 1 - byte
 * (so on_reset[t+1] can be passed directly).  stats is zeroed then accumulated. */
int srl_masked_stats(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n,
                     double* stats);

/* out = float32( (x*mask - mean) / (sqrt(var) + eps) ) with mean = s/n, var = q/n - mean^2
 * (optionally * n/(n-1)) computed in float64 from stats = {n, s, q} (utils.py:62-67).  stats are
 * the *global* (all-reduced) sums when data-parallel (utils.py:58-61). */
int srl_masked_normalize(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n,
                         const double* stats, double eps, int unbiased, float* out);

/* ------------------------------------------------------------------------------------------------
 * PopArt value head: running statistics of the value targets and the two element-wise maps.
 * Replaces: RunningMeanStd.update / mean_std / normalize / denormalize (modules/utils.py:104-156) and
 *           PopArtValueHead.update (popart.py:42-51).
 * rms: float64[2*vd + 1] = {mean[vd], mean_sq[vd], debiasing_term}, the reference's three nn.Parameters.
 * ---------------------------------------------------------------------------------------------- */

/* Per value column c: stats[3c..3c+2] = {sum mask, sum x*mask, sum (x*mask)^2} in float64
 * (utils.py:113-120).  x float32[n, vd]; mask uint8[n] or NULL; mask_invert as in srl_masked_stats.
 * Data-parallel callers all-reduce stats (one message instead of utils.py:121-124's three). */
int srl_masked_stats_cols(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n,
                          int vd, double* stats);

/* col_stats [vd][3] from srl_masked_stats_cols -> stats[3] = { col_stats[0][0], sum_c col_stats[c][1], sum_c col_stats[c][2] }:
 * the sums masked_normalization takes over ALL channels of a [T, B, value_dim] advantage block under one [T, B, 1] mask
 * (modules/utils.py:38-57 with the mask broadcast). */
int srl_fold_col_stats(void* stream, const double* col_stats, int vd, double* stats);
/* ratio[i] = exp(new_lp[i] - old_lp[i]): the V-trace importance ratio of mappo.py:130-133. */
int srl_importance_ratio(void* stream, const float* new_lp, const float* old_lp, long n, float* ratio);
/* rms <- beta * rms + (1 - beta) * {s/n, q/n, 1} (utils.py:125-130).  rescale != 0 additionally rewrites
 * the head so that its de-normalised output is unchanged: w[r,:] *= old_std/new_std,
 * b = (old_std*b + old_mean - new_mean)/new_std (popart.py:49-51; the reference only does this after
 * burn_in_updates, infinite by default).  w float32[vd, in_features], b float32[vd]. */
int srl_popart_update(void* stream, const double* stats, double beta, double eps, int vd, double* rms,
                      float* w, float* b, int in_features, int rescale);

/* normalize != 0: out = float32(clip((x - mean)/std, -5, 5)) (utils.py:143-148); else out =
 * float32(x*std + mean) (:150-155); mean = rms_mean/max(debias, eps), std = sqrt(max(E[x^2] - mean^2,
 * 1e-2)) in float64 (:137-142).  x, out float32[n, vd]. */
int srl_popart_map(void* stream, const float* x, long n, int vd, const double* rms, double eps,
                   int normalize, float* out);

/* ------------------------------------------------------------------------------------------------
 * Continuous actions: diagonal Gaussian head, torch.distributions.Normal(mean, std) summed over the action
 * dimensions.  Replaces actor_critic_policy.py:128-133 (std), :318-321 (log_prob / entropy and their autograd
 * backward), :499-506 (rollout).  mean float32[n, A] with row pitch ld_mean; log_std float32[A] shared by all
 * rows (ld_log_std = 0: `fixed`, `separate_learnable`) or float32[n, A] with pitch ld_log_std
 * (`shared_learnable`: output of a second head); action float32[n, A] contiguous.
 * ---------------------------------------------------------------------------------------------- */
int srl_gaussian_fwd(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                     const float* action, long n, int A, float* logp, float* entropy);

/* d_mean, d_log_std float32[n, A] (per row; the caller column-sums d_log_std for the shared vector). */
int srl_gaussian_bwd(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                     const float* action, long n, int A, const float* d_logp, const float* d_entropy,
                     float* d_mean, float* d_log_std);

/* action = mean where is_eval (or sampled: Box-Muller on Philox4x32-10 counters (row0 + row, dim, offset) keyed by seed),
 * logp = its log-probability.  row0: the first row's number in the caller's whole batch -- a batch that goes through in
 * pieces (copy of piece i+1 under the compute of piece i) samples exactly what it would in one call. */
int srl_gaussian_sample(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                        const uint8_t* is_eval, long n, int A, uint64_t seed, uint64_t offset, float* action,
                        float* logp, int64_t row0);

/* ------------------------------------------------------------------------------------------------
 * Recurrent backbone: GRU cell between the GEMMs, one time step per launch.
 * Replaces: AutoResetRNN.forward around torch.nn.GRU (modules/autoreset_rnn.py:42-66,
 *           recurrent_backbone.py:61-66) and its autograd backward; chunking of modules/utils.py:164-182.
 * Per-step blocks: gates [N, 3H] in torch order r|z|n, states [N, H]; N = env columns of the chunked batch.
 * ---------------------------------------------------------------------------------------------- */

/* out = h * (1 - reset) row-wise (reset uint8[N] or NULL: copy); autoreset_rnn.py:59. */
int srl_gru_mask_state(void* stream, const float* h, const uint8_t* reset, long N, int H, float* out);

/* gi = W_ih x + b_ih, gh = W_hh h_in + b_hh (both from srl_gemm) -> y = h(t); gi is overwritten with the gates
 * (r, z, n) and gh keeps its n part, both for srl_gru_cell_bwd; hin_next (optional) = y * (1 - reset_next). */
int srl_gru_cell_fwd(void* stream, float* gi, float* gh, const float* hin, const uint8_t* reset_next,
                     long N, int H, float* y, float* hin_next);

/* dh = dy (or 0) + carry * (1 - reset_next) (carry = d loss / d h_in(t+1), or NULL at the last step).
 * gates <- d gi, gh <- d gh (in place), dh_direct = dh * z; the caller adds d gh . W_hh with srl_gemm
 * (accumulate) to obtain d loss / d h_in(t), the next call's carry. */
int srl_gru_cell_bwd(void* stream, const float* dy, const float* carry, const uint8_t* reset_next,
                     float* gates, float* gh, const float* hin, long N, int H, float* dh_direct);

/* LSTM cell (torch.nn.LSTM, gates i|f|g|o): pre [N, 4H] = W_ih x + b_ih + W_hh h_in + b_hh (two srl_gemm calls
 * accumulating into one block); cin = c(t-1) after the auto reset.  y = h(t), cnew = c(t); pre is overwritten with
 * the activated gates; hin_next / cin_next (optional, both or neither) = h(t), c(t) * (1 - reset_next). */
int srl_lstm_cell_fwd(void* stream, float* pre, const float* cin, const uint8_t* reset_next, long N, int H,
                      float* y, float* cnew, float* hin_next, float* cin_next);

/* carry_h / carry_c = d loss / d (h_in, c_in)(t+1) (both or neither), cut where reset_next.  gates <- d pre (in
 * place; d gi = d gh for an LSTM), dc_in = d loss / d c_in(t); d loss / d h_in(t) = d pre . W_hh via srl_gemm. */
int srl_lstm_cell_bwd(void* stream, const float* dy, const float* carry_h, const float* carry_c,
                      const uint8_t* reset_next, float* gates, const float* cin, const float* cnew, long N,
                      int H, float* dc_in);

/* The whole time loop of a chunk in one launch (csrc/rnn_seq.hip; H in {32, 64}: srl_rnn_seq_supported(kind, H), kind 0
 * GRU, 1 LSTM).  Same buffers, layouts and saved values as C calls of the per-step entry points above with the srl_gemm of
 * W_hh between them (autoreset_rnn.py:42-66: the rows are independent, only the steps are serial; float32 matrix cores, a
 * wavefront per 32 rows with the state in registers): every block is
 * [C][N][.], step c at offset c * N rows; reset [C][N] uint8 or NULL, reset[c] masks the state entering step c (c >= 1;
 * hin[0] / cin[0] are the caller's, already masked).
 * srl_lstm_seq_fwd: pre = W_ih x + b_ih on entry -> activated gates; y, cnew, hin[1..], cin[1..] written.
 * srl_lstm_seq_bwd: gates -> d pre in place (d h_in / d c_in are carried inside the kernel); dy [C][N][ld_dy] or NULL.
 * srl_gru_seq_fwd: gi = W_ih x + b_ih on entry -> gates (r, z, n); gh written (its n part kept); y, hin[1..] written.
 * srl_gru_seq_bwd: gates -> d gi, gh -> d gh in place. */
int srl_rnn_seq_supported(int kind, int H);
int srl_lstm_seq_fwd(void* stream, float* pre, const float* w_hh, const float* b_hh, float* hin, float* cin,
                     const uint8_t* reset, int64_t N, int H, int C, float* y, float* cnew);
int srl_lstm_seq_bwd(void* stream, const float* dy, int64_t ld_dy, float* gates, const float* w_hh, const float* cin,
                     const float* cnew, const uint8_t* reset, int64_t N, int H, int C);
int srl_gru_seq_fwd(void* stream, float* gi, float* gh, const float* w_hh, const float* b_hh, float* hin,
                    const uint8_t* reset, int64_t N, int H, int C, float* y);
int srl_gru_seq_bwd(void* stream, const float* dy, int64_t ld_dy, float* gates, float* gh, const float* w_hh,
                    const float* hin, const uint8_t* reset, int64_t N, int H, int C);

/* Rows of D floats between time-major [T*B] and chunk-major order: dst[(c, k*B + b)] = src[((k*C + c), b)]
 * (modules/utils.py:164-182 `to_chunk`: torch.cat(torch.split(x, C, dim=0), dim=1)); inverse != 0 undoes it. */
int srl_chunk_rows(void* stream, const float* src, float* dst, int T, int B, int C, int D, int inverse);

/* ------------------------------------------------------------------------------------------------
 * PPO loss, forward + backward in one pass over the batch.
 * Replaces: MultiAgentPPO._compute_loss (mappo.py:146-217) + value-loss factories
 *           (modules/utils.py:228-265) + the autograd backward through them (mappo.py:274).
 */
typedef struct srl_ppo_hparams {
  float eps_clip;             /* mappo.py:72  default 0.2 */
  float c_clip;               /* mappo.py:75  default 3   */
  float value_eps_clip;       /* mappo.py:90  default eps_clip */
  float value_loss_weight;    /* mappo.py:91  default 0.5 */
  float entropy_bonus_weight; /* mappo.py:93  default 0.01 */
  float huber_delta;          /* value_loss_config delta / beta */
  float norm_eps;             /* utils.py:17  default 1e-5 */
  int32_t dual_clip;          /* mappo.py:74  default 1 */
  int32_t clip_value;         /* mappo.py:73  default 0 */
  int32_t value_loss;         /* 0 mse, 1 huber, 2 smoothl1 (utils.py:232-247) */
  int32_t mask_invert;        /* 1: mask byte is on_reset[t+1] (mask = 1 - byte); 0: byte is the mask */
} srl_ppo_hparams;

enum { SRL_LT_POLICY = 0, SRL_LT_VALUE = 1, SRL_LT_ENTROPY = 2, SRL_LT_CLIP = 3, SRL_LT_RATIO = 4,
       SRL_LT_ADV = 5, SRL_LT_RET = 6, SRL_LT_MASK = 7, SRL_LT_DONE = 8, SRL_LT_TRUNC = 9, SRL_LT_COUNT = 10 };

/* new_lp, old_lp, entropy: float32[n]; mask uint8[n]; value, old_value, adv (raw, un-normalised), ret:
 * float32[n, value_dim] (the reference's loss is shape-agnostic: ratio, mask and entropy broadcast over the value
 * channels and every masked mean divides the sum over ALL channels by the mask count, mappo.py:184,197).
 * norm_stats: float64[3] global {n, s, q} for the advantage normalisation.
 * local_n: float64[1], this rank's sum(mask) — every masked mean divides by it (mappo.py:184,197,199).
 * Outputs: d_new_lp, d_entropy float32[n], d_value float32[n, value_dim] = d loss / d input;
 *          loss_terms float64[SRL_LT_COUNT] (zeroed, then accumulated): masked SUMS of policy loss,
 *          value loss, entropy, 1[s2<s1], ratio, adv, ret, and mask; the caller forms
 *          loss = (P + w_v*V - w_H*E) / M.  done / truncated (uint8[n], nullable) are only summed
 *          (unmasked) into SRL_LT_DONE / SRL_LT_TRUNC for the `done` / `truncated` statistics
 *          (mappo.py:40-41,293-296). */
int srl_ppo_loss_fwd_bwd(void* stream, const float* new_lp, const float* old_lp, const float* value,
                         const float* old_value, const float* adv, const float* ret, const float* entropy,
                         const uint8_t* mask, long n, int value_dim, const srl_ppo_hparams* hp,
                         const double* norm_stats, const double* local_n, const uint8_t* done, const uint8_t* truncated,
                         float* d_new_lp, float* d_value, float* d_entropy, double* loss_terms);

/* ------------------------------------------------------------------------------------------------
 * Categorical heads.  Replaces torch.distributions.Categorical log_prob / entropy / sample in
 * ActorCriticPolicy.__get_log_prob_and_entropy (actor_critic_policy.py:311-324) and rollout
 * (:484-497), including the -1e10 masking of unavailable actions (:135-136).
 * logits [n, sum(head_dims)] (row stride ld_logits); action int32 [n, n_heads]; avail uint8
 * [n, sum(head_dims)] or NULL.
 */
#define SRL_MAX_HEADS 8
int srl_categorical_fwd(void* stream, const float* logits, int ld_logits, const int32_t* action,
                        const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims,
                        float* logp, float* entropy);
/* d_logits [n, sum(head_dims)] (row stride ld_dlogits) = d_logp * dlogp/dlogits + d_entropy * dH/dlogits */
int srl_categorical_bwd(void* stream, const float* logits, int ld_logits, const int32_t* action,
                        const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims,
                        const float* d_logp, const float* d_entropy, float* d_logits, int ld_dlogits);
/* Rollout: action = argmax if is_eval[row] else inverse-CDF sample with Philox4x32-10
 * (key = seed, counter = {row0 + row, head, offset}); logp [n] = sum over heads of log p(action).
 * action_out int64 [n, n_heads] (dtype of actor_critic_policy.py:515).  row0: as for srl_gaussian_sample. */
int srl_categorical_sample(void* stream, const float* logits, int ld_logits, const uint8_t* avail,
                           const uint8_t* is_eval, long n, int n_heads, const int32_t* host_head_dims,
                           uint64_t seed, uint64_t offset, int64_t* action_out, float* logp, int64_t row0);

/* ------------------------------------------------------------------------------------------------
 * Phasic Policy Gradient, auxiliary phase (ABI 15).
 *
 * srl_categorical_log_softmax: out[i, head h] = masked logits - logsumexp over the head -- the `.logits` of the
 * torch.distributions.Categorical objects `analyze(target="ppg_aux_phase")` returns and the trainer's cache keeps
 * (actor_critic_policy.py:266-270, 417-435; phasic_policy_gradient.py:215-221), unavailable actions at -1e10 before the
 * normalisation (:135-136).  logits [n, sum(head_dims)] (row stride ld_logits), avail uint8 [n, sum(head_dims)] or NULL.
 *
 * srl_ppg_aux_loss_fwd_bwd: MultiAgentPPG._compute_aux_loss (phasic_policy_gradient.py:262-280) and its autograd backward with
 * respect to the network's outputs, one pass:
 *   L = aux_value_loss + beta_clone * policy_distance + value_head_weight * value_head_loss
 *   policy_distance = sum_heads sum_rows KL(old_h || new_h) (1 - done) / sum_rows (1 - done)      (_policy_distance, :140-144;
 *                     KL as torch.distributions.kl_divergence of two Categoricals: 0 where p == 0, +inf where q == 0 < p)
 *   *_value_loss    = 1/2 sum (v - target)^2 (1 - done) / sum_rows (1 - done)                     (_paper_value_loss, :146-147)
 * logq_old [n, A]: normalised log-probabilities kept by the cache entry; logits [n, A]: the current policy's raw logits;
 * aux_value / pred_value / target float32 [n, value_dim]; done uint8 [n] (the entry's info_mask, :264);
 * undone_count: device float64, sum_rows (1 - done) (srl_masked_stats with mask_invert).
 * Outputs: d_logits [n, A] (row stride ld_dlogits; zero at unavailable actions), d_aux, d_pred [n, value_dim] = d L / d input;
 * terms float64[3] (zeroed by the call) = {aux_value_loss, value_head_loss, policy_distance}. */
int srl_categorical_log_softmax(void* stream, const float* logits, int ld_logits, const uint8_t* avail, long n, int n_heads,
                                const int32_t* host_head_dims, float* out, int ld_out);
int srl_ppg_aux_loss_fwd_bwd(void* stream, const float* logq_old, int ld_old, const float* logits, int ld_logits,
                             const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims, const float* aux_value,
                             const float* pred_value, const float* target, int value_dim, const uint8_t* done,
                             const double* undone_count, float beta_clone, float value_head_weight, float* d_logits,
                             int ld_dlogits, float* d_aux, float* d_pred, double* terms);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction on the FP32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chains).
 * Replaces nn.Linear / Conv2d-as-GEMM forward and both backward products of
 * ActorCriticSeparate (actor_critic_policy.py:114-143; modules/utils.py:154-161; modules/cnn.py:93-135).
 *
 *   C[M,N] = epilogue( sum_k A(i,k) * B(k,j) )
 *   A(i,k) = a_kmajor ? A[k*lda + i] : A[i*lda + k]
 *   B(k,j) = b_kmajor ? B[k*ldb + j] : B[j*ldb + k]
 *
 *   forward   Y = act(X W^T + b):  A = X [M,K] (a_kmajor 0), B = W [N,K] (b_kmajor 0)
 *   dgrad     dX = dZ W:           A = dZ [M,N'] (a_kmajor 0), B = W [N',K'] (b_kmajor 1)
 *   wgrad     dW = dZ^T X:         A = dZ [rows,N'] (a_kmajor 1), B = X [rows,K'] (b_kmajor 1)
 */
typedef struct srl_gemm_desc {
  int64_t M, N, K;
  const float* A; int64_t lda; int32_t a_kmajor;
  const float* B; int64_t ldb; int32_t b_kmajor;
  float* C; int64_t ldc;
  const float* bias;        /* [N] added per output column, or NULL */
  int32_t act;              /* 0 none, 1 relu, 2 tanh — applied after bias */
  const float* dact_src;    /* [M, ld_dact] forward output Y; C *= act'(Y) (relu: Y>0, tanh: 1-Y^2), or NULL */
  int64_t ld_dact; int32_t dact;  /* activation kind for dact_src (1 relu, 2 tanh) */
  int32_t accumulate;       /* 1: C += result (after split reduction) */
  int32_t split_k;          /* >1: K is cut into split_k slices reduced through `workspace` */
  float* workspace;         /* >= split_k * M * N floats when split_k > 1 */
  float* a_colsum;          /* [M] += sum_k A(i, k), or NULL.  Needs a_kmajor and float4-stageable operands (16-byte
                             * aligned bases, pitches and contiguous extents multiples of 4): a weight-gradient
                             * product dZ^T X then also delivers the bias gradient sum_rows dZ (mappo.py:276's
                             * backward computes both), without a second pass over dZ. */
  /* Operand ranges (device floats, or NULL): an upper bound of max |A|, max |B|.  With BOTH given, a product of at least
   * 65 x 65 x 64 runs on two f16 pieces per operand and three piece products instead of three bf16 pieces and six: the
   * kernel scales each operand by the power of two that puts its bound into [2^13, 2^14) and undoes it on the
   * accumulators.  Every element is then carried to ~2^-22 relative if it lies within 2^16 of the operand's largest, and
   * to 2^-40 of that largest element below -- so hand over ranges for operands whose dot products are carried by their
   * large elements (weights, activations, and the gradients of a batch: what is summed over samples), not for operands
   * whose rows may consist of small elements only and matter individually.  SRL_F16X2=0 ignores the ranges. */
  const float* a_absmax;
  const float* b_absmax;
  float* out_absmax;        /* *out_absmax = max(*out_absmax, max |C| of the stored elements) (atomic; split_k == 1), or NULL:
                             * the range of the next layer's operand at no extra pass */
  /* Sign masks: the ReLU derivative needs one bit of the forward output, not the float.  mask_out (act == 1, split_k == 1,
   * N and ldc multiples of 32): bit (i * ldc + j) & 31 of word (i * ldc + j) >> 5 is set iff C[i, j] > 0.  dact_mask
   * (dact == 1, dact_src NULL, N and ld_dact multiples of 32): the same bits of the producer, its pitch in ld_dact -- C[i, j] is kept where bit
   * i * ld_dact + j is set and zeroed elsewhere, exactly what dact_src with the producer's floats gives at 1/32 of the
   * bytes read (modules/cnn.py:118's nn.ReLU backward). */
  uint32_t* mask_out;
  const uint32_t* dact_mask;
  /* B is srl_presplit's output instead of float32 (same shape, pitch and orientation): accepted only where the product takes
   * the two-piece kernel -- both ranges given, M > 64, N > 64, K >= 64, aligned operands, and not the small-product path (more
   * than 65 536 outputs or K > 512).  b_absmax must be the float srl_presplit was given. */
  int32_t b_presplit;
  /* B (b_kmajor 1, N a multiple of 32, ldb == N) is h2p rows (the pre-split format of the srl_h2_* kernels) under this scale
   * (device float) instead of float32: a weight-gradient product dZ^T X whose X was never written as float32.  Needs a_absmax
   * and takes the two-piece kernel; b_absmax is ignored. */
  const float* b_h2_scale;
} srl_gemm_desc;
/* dst[i .. i+3] (16 bytes) = the two f16 pieces of src[i .. i+3] under the scale of *absmax: what the two-piece kernels make of a
 * B operand every time they stage a tile of it, done once -- for weights, once per parameter update (srl_gemm_desc::b_presplit,
 * the w_presplit / wt_presplit arguments of the convolution entry points).  n a multiple of 4, 16-byte aligned pointers. */
int srl_presplit(void* stream, const float* src, const float* absmax, float* dst, int64_t n);
int srl_gemm(void* stream, const srl_gemm_desc* d);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (eps 1e-5, elementwise affine) over the last dimension D of [rows, D].
 * Replaces nn.LayerNorm in make_models_for_obs / mlp (policies/utils.py:47-49; modules/utils.py:159-160).
 * mean / rstd: float32[rows] saved for the backward.
 */
int srl_layernorm_fwd(void* stream, const float* x, int64_t ldx, const float* gamma, const float* beta,
                      int64_t rows, int D, float* y, int64_t ldy, float* mean, float* rstd);
/* dx = LN'(dy) (optionally * act'(dact_src) for the activation that FED the LayerNorm);
 * dgamma/dbeta += column sums (float32 atomics). dx may be NULL (input leaf). */
int srl_layernorm_bwd(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                      const float* gamma, const float* mean, const float* rstd, int64_t rows, int D,
                      float* dx, int64_t lddx, int dact, float* dgamma, float* dbeta, float* dx_absmax);
/* dx_absmax (or NULL; needs a dense dx, lddx == D): *dx_absmax = max(*dx_absmax, max |dx|) -- the range of the data gradient
 * for the two-piece products of the layer below, from the same pass (srl_gemm_desc's out_absmax). */

/* Whole-observation LayerNorm statistics for image observations [n, D] (D = C*H*W), uint8 or
 * float32 (policies/utils.py:53: nn.LayerNorm(v) over the full (C,H,W) shape). */
int srl_obs_ln_stats(void* stream, const void* obs, int is_u8, int64_t n, int D, float* mean, float* rstd);

/* im2col for the FIRST convolution: reads NCHW observations (uint8 or float32), applies the
 * whole-observation LayerNorm (mean/rstd per sample, affine gamma/beta [C,H,W]) on the fly and writes
 * the patch matrix P [n*OH*OW, C*KH*KW] with k = (c, kh, kw) (the reference's Conv2d weight order). */
int srl_im2col_obs_ln(void* stream, const void* obs, int is_u8, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, int64_t n, int C, int H, int W, int KH, int KW,
                      int stride, float* P);
/* im2col for later convolutions on NHWC float32 activations: P [n*OH*OW, KH*KW*C], k = (kh, kw, c). */
int srl_im2col_nhwc(void* stream, const float* x, int64_t n, int H, int W, int C, int KH, int KW, int stride,
                    float* P);
/* Gather form of col2im: dX[n,h,w,c] = sum of the dP entries that patch (h,w,c) went to;
 * optionally * act'(y) with y = forward activation of the same shape as dX (dact 1 relu, 2 tanh). */
int srl_col2im_nhwc(void* stream, const float* dP, int64_t n, int H, int W, int C, int KH, int KW, int stride,
                    const float* y, int dact, float* dX);
/* Backward of srl_im2col_obs_ln w.r.t. the LayerNorm affine parameters:
 * dgamma[c,h,w] += sum_n dXn * xhat, dbeta[c,h,w] += sum_n dXn, dXn = col2im(dP) (never materialised). */
int srl_obs_ln_affine_bwd(void* stream, const float* dP, const void* obs, int is_u8, const float* mean,
                          const float* rstd, int64_t n, int C, int H, int W, int KH, int KW, int stride,
                          float* dgamma, float* dbeta);

/* out[j] (+)= sum_i x[i*ld + j]  (bias gradients). */
int srl_colsum(void* stream, const float* x, int64_t ld, int64_t rows, int cols, float* out, int accumulate);
/* Strided 2-D copy dst[i*ldd + j] = src[i*lds + j] (observation concat / layout glue). */
int srl_copy2d(void* stream, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols);
/* dst = float32(src uint8), n elements (flags that feed float arithmetic on the host-facing API). */
int srl_u8_to_f32(void* stream, const uint8_t* src, float* dst, int64_t n);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolutions (padding 0): the patch matrix is never written to memory, the operand gather
 * happens inside the MFMA kernel's load path.  Replaces nn.Conv2d forward / backward of the `Convolution`
 * encoder (legacy/algorithm/modules/cnn.py:93-135) and, for the first layer, the whole-observation
 * nn.LayerNorm in front of it (policies/utils.py:53) including the float32 widening of uint8 frames
 * (api/trainer.py:217).  Activations between convolutions are NHWC; weights [Cout, KH, KW, Cin]; the first
 * layer reads the NCHW observation with the reference's weight layout [Cout, Cin, KH, KW].
 * The srl_im2col_* / srl_col2im_* entry points above remain as the general fallback for geometries
 * srl_conv2d_supported() rejects.
 */
typedef struct srl_conv_desc {
  int64_t n;                 /* images */
  int32_t H, W, Cin;         /* input */
  int32_t KH, KW, stride, Cout;
  int32_t act;               /* forward activation fused after the bias: 0 none, 1 relu, 2 tanh */
} srl_conv_desc;

/* 1 if the implicit path handles this geometry.  first_layer: 0 = NHWC activation layer, 1 = observation layer
 * with a planar (NCHW) observation, 2 = observation layer with a channels-last (NHWC) observation. */
int srl_conv2d_supported(const srl_conv_desc* d, int first_layer);
/* y[n,OH,OW,Cout] = act(conv(x[n,H,W,Cin], w) + bias) */
/* x_absmax / w_absmax / y_absmax (and the dz_absmax / dx_absmax of the gradient entry points below): as srl_gemm_desc's
 * a_absmax / b_absmax / out_absmax (NULL: three bf16 planes, no tracking). */
/* y_mask (ReLU layers, Cout a multiple of 32; or NULL): [ceil(n*OH*OW*Cout / 32)] words, bit e & 31 of word e >> 5 set iff
 * element e of y is > 0 -- the one bit the backward pass needs of y where it is only the ReLU derivative that is wanted
 * (srl_conv2d_nhwc_dgrad's x_mask, srl_gemm_desc's dact_mask): modules/cnn.py:118. */
/* w_presplit != 0: `w` is srl_presplit(w, w_absmax): accepted only where the layer takes the two-piece kernel (both ranges,
 * Cout in 33..., KH*KW*Cin >= 64). */
int srl_conv2d_nhwc_fwd(void* stream, const srl_conv_desc* d, const float* x, const float* w, const float* bias,
                        float* y, const float* x_absmax, const float* w_absmax, float* y_absmax, uint32_t* y_mask,
                        int w_presplit);
/* dw[Cout,KH,KW,Cin] += sum over (n,oh,ow) dz[.,Cout]^T patch(x); workspace: srl_conv2d_wgrad_workspace floats
 * (split over the n*OH*OW reduction) or NULL.  dbias (optional): [Cout] += sum over (n,oh,ow) dz, the bias
 * gradient, from the same pass over dz. */
int64_t srl_conv2d_wgrad_workspace(const srl_conv_desc* d);
int srl_conv2d_nhwc_wgrad(void* stream, const srl_conv_desc* d, const float* x, const float* dz, float* dw,
                          float* workspace, float* dbias, const float* x_absmax, const float* dz_absmax);
/* Data gradient.  Input pixels are split into stride*stride parity classes, each a dense stride-1 problem
 * over only the taps that reach it (no multiply-by-zero work).  wt = the weights regrouped per class
 * (srl_conv2d_dgrad_repack, srl_conv2d_dgrad_weight_elems floats; redo after every optimiser step).
 * dx[n,H,W,Cin] = (sum over taps dz * w) * act'(x_act)   (x_act = the forward activation that has dx's shape, or NULL) */
int64_t srl_conv2d_dgrad_weight_elems(const srl_conv_desc* d);
int srl_conv2d_dgrad_repack(void* stream, const srl_conv_desc* d, const float* w, float* wt);
/* x_mask (dact == 1, x_act NULL, Cin a multiple of 32): the sign bits of x_act as written by the producing layer's y_mask -- the same dx at 1/32
 * of the bytes read for the derivative. */
/* wt_presplit != 0: `wt` is srl_presplit(wt, w_absmax) of the regrouped weights: two-piece kernel only (both ranges, Cout a
 * multiple of 16, more than 32 packed columns). */
int srl_conv2d_nhwc_dgrad(void* stream, const srl_conv_desc* d, const float* dz, const float* wt, const float* x_act,
                          int dact, float* dx, const float* dz_absmax, const float* w_absmax, float* dx_absmax,
                          const uint32_t* x_mask, int wt_presplit);
/* First layer: y = act(conv(LayerNorm(obs), w) + bias) with the LayerNorm over the whole observation; obs uint8 or
 * float32, mean/rstd [n] from srl_obs_ln_stats / srl_obs_space_to_depth.  channels_last = 0: obs [n,Cin,H,W],
 * gamma/beta [Cin,H,W], w [Cout,Cin,KH,KW] (the reference's layouts); channels_last = 1: obs [n,H,W,Cin],
 * gamma/beta [H,W,Cin], w [Cout,KH,KW,Cin] (what srl_obs_space_to_depth produces). */
int64_t srl_conv2d_obs_fwd_workspace(const srl_conv_desc* d);
/* workspace (srl_conv2d_obs_fwd_workspace floats, or NULL): with it the layer runs position-batched, one GEMM per
 * output position over the samples with the LayerNorm affine folded into per-position weights (w*gamma) and biases
 * (bias + w.beta), so that the operand gather needs no table lookups; without it the affine is applied in the gather. */
/* row_index (int32 [n], device, or NULL): sample i of the batch is row row_index[i] of `obs`, and mean / rstd are indexed
 * by that same row -- the batch's frames are then read in place from the HBM observation ring (srl_gather_rows' comment)
 * with no gather pass at all.  Honoured by the byte kernels only: ask srl_conv2d_obs_row_index_supported first (1 = yes);
 * with 0 gather the rows (srl_gather_rows) and pass NULL. */
int srl_conv2d_obs_row_index_supported(const srl_conv_desc* d, int is_u8, int channels_last);
int srl_conv2d_obs_fwd(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w,
                       const float* bias, float* y, float* workspace, const int32_t* row_index, float* y_absmax,
                       uint32_t* y_mask, int reuse_folded);
/* reuse_folded != 0: `workspace` still holds the folded weights of the previous call with the SAME w / bias / gamma / beta
 * and geometry (the chunks of one update): the folding kernel is skipped. */
/* *out = max(*out, max_i |x[i]|) (atomic: several calls may fold into one slot; the caller zeroes it): the range of a
 * weight tensor for the two-plane f16 products, once per parameter update. */
int srl_absmax(void* stream, const float* x, int64_t n, float* out);
/* Direct convolutions for layers with 4 or 8 channels on both sides (csrc/conv_small.hip; ABI 17): NHWC float32, 3 x 3 or 5 x 5,
 * stride 1, no padding, Cin and Cout in {4, 8} (5 x 5: Cin 4) -- the reference's default convolution stack (modules/cnn.py:96-98: C -> C (5 x 5),
 * C -> 2C, 2C -> C (3 x 3)) as vector-unit kernels (a thread per pixel, weights as scalar operands; the weight gradient with up to
 * 288 sums of a pixel column in registers, slabs per wavefront + a second launch: no atomics).  w: [Cout, K, K, Cin].
 * fwd: y = act(conv(x, w) + bias) (bias may be NULL).  dgrad: dx = (dz conv^T w) * act'(x_act) with x_act = the layer's INPUT
 * activation (the output of the activation `dact`: 0 none / x_act NULL, 1 relu, 2 tanh).  wgrad: gw += dz^T patches(x), gb +=
 * column sums of dz (gb may be NULL); workspace: srl_conv2d_small_wgrad_workspace floats.  Replace nn.Conv2d forward / backward
 * (modules/cnn.py:57-72) for these geometries. */
int srl_conv2d_small_supported(const srl_conv_desc* d);
int srl_conv2d_small_fwd(void* stream, const srl_conv_desc* d, const float* x, const float* w, const float* bias, float* y);
int srl_conv2d_small_dgrad(void* stream, const srl_conv_desc* d, const float* dz, const float* w, const float* x_act, int32_t dact, float* dx);
int64_t srl_conv2d_small_wgrad_workspace(const srl_conv_desc* d);
int srl_conv2d_small_wgrad(void* stream, const srl_conv_desc* d, const float* x, const float* dz, float* workspace, float* gw, float* gb);
/* Sign words of a ReLU output (ABI 17): bit e & 31 of mask[e >> 5] = (x[e] > 0), n a multiple of 32 -- what the convolution
 * entry points write as y_mask when their channel count allows it, for an activation whose producer wrote none (a convolution with
 * 4 or 8 channels in front of a wide Linear whose data gradient reads the derivative from bits, srl_h2_gemm's mask_in). */
int srl_relu_mask(void* stream, const float* x, int64_t n, uint32_t* mask);
/* Space-to-depth of a planar observation for a strided first convolution (stride s | KH, KW, H, W):
 * out[n, H/s, W/s, (c, ph, pw)] = obs[n, c, a*s + ph, b*s + pw], same element type, plus the whole-observation
 * LayerNorm statistics in the same pass.  A KxK stride-s convolution on obs becomes a (K/s)x(K/s) stride-1
 * convolution on the channels-last result with Cin' = C*s*s, whose patch rows are contiguous 128-byte runs instead
 * of 8-byte ones (Atari: 8x8 stride 4 on 4x84x84 -> 2x2 stride 1 on 21x21x64). */
int srl_obs_space_to_depth(void* stream, const void* obs, int is_u8, int64_t n, int C, int H, int W, int s, void* out,
                           float* mean, float* rstd);
/* First layer backward without a data gradient: one batched GEMM over the OH*OW output positions forms
 * Q[pos] = dz[:,pos,:]^T xhat_patches[:,pos,:] (xhat = normalised, pre-affine observation); then
 * dw += sum_pos gamma*Q + beta*R, db += sum_pos R, dgamma += fold_o(w*Q), dbeta += fold_o(w*R), R = per-position
 * column sums of dz.  workspace: srl_conv2d_obs_bwd_workspace floats. */
int64_t srl_conv2d_obs_bwd_workspace(const srl_conv_desc* d);
/* phase: 3 = a call of its own (what every caller passes but the trainer's chunk loop); otherwise bit 0 = this call opens an
 * accumulation over calls (the sums Q, R, C in `workspace` start from zero), bit 1 = it closes one (the four gradients are formed
 * from the sums and ADDED into dw / db / dgamma / dbeta): the gradients are linear in the sums, so the chunks of one update share one
 * finalisation.  The workspace must not be touched between the calls of an accumulation; byte kernels with split slabs only.
 * dz_absmax (ABI 14, nullable): a device float >= max |dz| (the h2 data gradient that produced dz measures it).  With it, the
 * Atari geometry (64-channel space-to-depth'd frames, 2x2 taps, 32 channels) runs the block kernel of csrc/obs_h2.h, which needs
 * the bound for the power-of-two scale of its f16 pieces of dz; without it, the bf16 kernel of csrc/obs_bf16.h. */
int srl_conv2d_obs_bwd(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w,
                       const float* dz, float* dw, float* db, float* dgamma, float* dbeta, float* workspace,
                       const int32_t* row_index, int phase, const float* dz_absmax);

/* ------------------------------------------------------------------------------------------------
 * Pre-split ("h2") operands and the kernels over them (round 4: csrc/h2gemm.h, csrc/h2conv.h).
 * The contractions of the update run as three f16 piece products per multiply-add (gemm_bf16x3.h); the round-3 kernels
 * split every float32 operand again in every tile that staged it and were bound by that staging.  Here a tensor is split
 * ONCE, by the kernel that produces it, into two f16 pieces per element -- 4 bytes, what the float32 took -- and the
 * consumers move bytes (buffer_load ... lds, no register staging):
 *   value = (h0 + h1) / scale, scale a power of two in a device float beside the tensor, derived from an upper BOUND of the
 *   tensor's magnitude that is known before the tensor exists (max |input| x max row 1-norm of the weights + max |bias|),
 *   so a producer can split while it stores and an f16 overflow is impossible by construction;
 *   "h2p rows": a row of C elements (C a multiple of 32) is C / 32 blocks of 128 bytes = 4 groups x {8 first pieces,
 *   8 second pieces}; group g holds elements 4 (g >> 1) + 16 (g & 1) + {0,1,2,3, 8,9,10,11} of its block (what one lane of
 *   a 32x32 MFMA accumulator holds);  "planar image": the same 16-byte entries as [block][group][piece][pixel] per image.
 * Replaces, for the Atari network of modules/cnn.py:93-135 + actor_critic_policies/utils.py:33-63 (20x20x32 -> 9x9x64 ->
 * 7x7x64 -> Linear 3136 -> 512), the forward, data-gradient and weight-gradient contractions of mappo.py:243-284.
 */
/* float32 rows [rows, C] (pitch ld) <-> h2p rows (pitch C).  Scale: *scale_in if given, else derived from *absmax and
 * written to *scale_out. */
int srl_h2_pack_rows(void* stream, const float* src, int64_t ld, int64_t rows, int32_t C, const float* absmax,
                     const float* scale_in, float* scale_out, void* dst);
/* srl_h2_pack_rows with the column sums of src beside it (ABI 17): colsum[C] (+)= sum over the rows of src -- the bias gradient of
 * the layer whose output gradient is being packed, from the pass that reads it anyway (slabs per workgroup in `workspace`,
 * srl_h2_pack_rows_colsum_workspace floats, added in a fixed order: no atomics).  C <= 2048, ld a multiple of 4, src 16-byte aligned. */
int64_t srl_h2_pack_rows_colsum_workspace(int64_t rows, int32_t C);
int srl_h2_pack_rows_colsum(void* stream, const float* src, int64_t ld, int64_t rows, int32_t C, const float* absmax, const float* scale_in,
                            float* scale_out, void* dst, float* workspace, float* colsum, int32_t accumulate);
int srl_h2_unpack_rows(void* stream, const void* src, int64_t rows, int32_t C, const float* scale, float* dst, int64_t ld);
/* float32 NHWC images <-> h2 images.  layout 0: planar, raster pixel order; 1: h2p rows [n][H*W][C], raster order; 2: h2p
 * rows with the pixels of an image in parity-class-major order ((y & 1, x & 1), then y >> 1, x >> 1: what a stride-2
 * layer reads; pack only). */
int srl_h2_pack_image(void* stream, const float* src, int64_t n, int32_t H, int32_t W, int32_t C, int32_t layout,
                      const float* absmax, const float* scale_in, float* scale_out, void* dst);
int srl_h2_unpack_image(void* stream, const void* src, int64_t n, int32_t H, int32_t W, int32_t C, int32_t layout,
                        const float* scale, float* dst);
/* A weight matrix as h2p rows dst [rows][K] under the scale of *absmax (= max |w|, srl_absmax), with *rownorm_out = max over
 * rows of sum_k |.| (the factor of the output bound), once per parameter update.  mode 0: w is [rows][K]; 1: w is [K][rows]
 * (its transpose is packed: the data gradient of a Linear); 2: the data-gradient regrouping of a convolution weight
 * w [Cout][KH][KW][Cin] described by d: rows = stride^2 Cin as (parity class, cin), K = (KH/stride)(KW/stride) Cout as (tap, cout). */
int srl_h2_weights(void* stream, const float* w, int32_t rows, int32_t K, int32_t mode, const srl_conv_desc* d,
                   const float* absmax, float* scale_out, float* rownorm_out, void* dst);

/* Image-stationary convolutions of the Atari stack (csrc/h2conv.h): an image enters LDS once, every tap reads it there,
 * the weights of a wavefront's 16 output channels stay in registers.
 *   kind 0  conv2 forward   x: h2p rows [n][400][32] in parity-class order (srl_conv2d_obs_fwd_h2) -> out: planar [n] 9x9x64
 *   kind 1  conv3 forward   x: planar 9x9x64 -> out: h2p rows [n][49][64] (= [n][3136], the Linear's operand)
 *   kind 2  conv3 data gradient   x: dz as h2p rows [n][49][64], w: srl_h2_weights mode 2 -> out: planar 9x9x64
 *   kind 3  conv2 data gradient   x: dz planar 9x9x64, w: mode 2 (128 rows) -> out: float32 NHWC [n,20,20,32]
 * forward kinds: out = relu(conv + bias), mask_out = sign bytes in h2 order (one byte per group of 8 channels, bit j =
 * element j of the group; [image][pixel][block][group]); data gradients: out *= the ReLU derivative read from mask_in
 * (kind 2: h2 order, what kind 0 wrote; kind 3: natural order, 32-bit word per pixel, what srl_conv2d_obs_fwd writes).
 * *out_absmax = max(*out_absmax, max |out|) (the bound_in of the next layer). */
typedef struct srl_h2_conv_args {
  const void* x; const void* w;
  const float* sx; const float* sw;      /* scales of x and w (device floats) */
  int64_t n;
  const float* bias; int32_t act;
  void* out; float* out_scale;
  const float* bound_in; const float* bound_w; const float* bound_b;   /* |out| <= *bound_in * *bound_w + *bound_b (bound_b may be NULL) */
  float* out_absmax;
  void* mask_out; const void* mask_in;
} srl_h2_conv_args;
int srl_h2_conv(void* stream, int32_t kind, const srl_h2_conv_args* a);
/* Weight gradients, image-stationary: gw [64][KH][KW][Cin] += sum dz^T patches(x), gb [64] += sum dz (gb may be NULL).
 * kind 0: conv2 (x as kind 0's input, dz planar 9x9x64); kind 1: conv3 (x planar 9x9x64, dz h2p rows [n][49][64]).
 * workspace: srl_h2_wgrad_workspace(kind) floats (one slab per persistent workgroup, summed by a second launch). */
int64_t srl_h2_wgrad_workspace(int32_t kind);
int srl_h2_wgrad(void* stream, int32_t kind, const void* x, const void* dz, const float* sx, const float* sz, int64_t n,
                 float* workspace, float* gw, float* gb);
/* Dense product over h2p rows through an LDS-DMA ring (csrc/h2gemm.h): out[M, NC] = act(x[M, K] w[NC, K]^T + bias).
 * out_h2 0: float32 [M][NC]; 1: h2p rows (needs the bound's factors).  mask_out: ReLU sign bits of out, natural order
 * (bit c of the word of a row's 32-channel block); mask_in: out *= derivative bit at the output element, natural order or
 * (mask_in_h2order) the h2 order srl_h2_conv's forward kinds write.  Replaces nn.Linear forward / data gradient
 * (modules/utils.py:154-161) for the encoder's 3136 -> 512 layer. */
typedef struct srl_h2_gemm_desc {
  const void* x; const void* w; const float* sx; const float* sw;
  int64_t M; int32_t NC; int32_t K;
  const float* bias; int32_t act;
  int32_t out_h2; void* out; float* out_scale;
  const float* bound_in; const float* bound_w; const float* bound_b;
  float* out_absmax; uint32_t* mask_out; const uint32_t* mask_in; int32_t mask_in_h2order;
} srl_h2_gemm_desc;
int srl_h2_gemm(void* stream, const srl_h2_gemm_desc* d);
/* ... with the reduction split over `ksplits` workgroups per tile: out is [ksplits][M][NC] float32 slabs of raw partial sums (no
 * bias / activation / masks / h2 output: the consumer adds the slabs, e.g. srl_ln_heads_fwd's x_slabs).  For row counts that leave
 * most CUs without a tile -- the Linear forward (modules/cnn.py:128-133) of an inference batch -- or, `wide` != 0 (256 channels
 * per workgroup, NC a multiple of 256), for a training chunk: fewer operand bytes staged per multiply-add, the k-ranges restore
 * the workgroup count.  NC a multiple of 128. */
int srl_h2_gemm_splitk(void* stream, const srl_h2_gemm_desc* d, int32_t ksplits, int32_t wide);
/* Weight gradient of a dense layer over h2p rows (csrc/h2tn.h; ABI 17): gw[NA][NB] (+)= sum_m a[m, :]^T b[m, :] with a = the
 * gradient with respect to the layer's output as h2p rows [M][NA] (srl_h2_pack_rows) and b = the layer's input as h2p rows
 * [M][NB] (what srl_h2_conv kind 1 / srl_h2_gemm write), *sa, *sb their scales; row pitches in bytes (>= 4 NA, 4 NB; multiples of
 * 16).  Both operands reach LDS by DMA and the matrix cores by transposing reads (ds_read_b64_tr_b16); every workgroup writes a
 * float32 slab of its row range into `workspace` (srl_h2_wgrad_dense_workspace floats) and a second launch adds the slabs in a
 * fixed order (no atomics: bit-reproducible).  accumulate 0: gw = sum; 1: gw += sum.  NA, NB multiples of 32.  Replaces the
 * weight-gradient product of nn.Linear's backward (modules/utils.py:154-161, modules/cnn.py:128-133: the encoder's
 * 3136 -> 512 layer, football_rnn.py:34-55: the dense tower). */
int64_t srl_h2_wgrad_dense_workspace(int64_t M, int32_t NA, int32_t NB);
int srl_h2_wgrad_dense(void* stream, const void* a, const void* b, const float* sa, const float* sb, int64_t M, int32_t NA, int32_t NB,
                       int64_t a_row_bytes, int64_t b_row_bytes, float* workspace, float* gw, int32_t accumulate);
/* srl_conv2d_obs_fwd with the output as the h2p rows kind 0 above reads (ent_order 2) instead of float32: byte kernels
 * only (uint8 channels-last frames, Cout 32), y_mask / y_absmax / workspace required; *y_scale = the scale used.
 * records: NULL, or room for 16 (n + 32) bytes of per-sample records of THIS launch (16-byte aligned).  NULL keeps them inside
 * `workspace` behind the folded weights -- which then serves one stream at a time; with a room of its own per caller, several
 * streams may run this entry point on one set of folded weights (reuse_folded = 1) side by side. */
int srl_conv2d_obs_fwd_h2(void* stream, const srl_conv_desc* d, const void* obs, const float* mean, const float* rstd,
                          const float* gamma, const float* beta, const float* w, const float* bias, void* y_h2, float* y_scale,
                          float* workspace, const int32_t* row_index, float* y_absmax, uint32_t* y_mask, int reuse_folded,
                          int ent_order, void* records);
/* Only the folded weights that srl_conv2d_obs_fwd_h2 keeps in `workspace` (what reuse_folded = 1 then reads): they depend on
 * the parameters alone (the LayerNorm affine of modules/cnn.py:100 folded into the first convolution's weights, :108), so the
 * trainer enqueues them once per parameter version ahead of the update's first chunk.  Returns 0: written; 1: d is not the
 * geometry of the kernel that keeps folded weights in this format (nothing written; the forward call folds for itself). */
int srl_conv2d_obs_fold_h2(void* stream, const srl_conv_desc* d, const float* gamma, const float* beta, const float* w,
                           const float* bias, float* workspace);

/* Row gather behind the HBM observation ring: dst[i, :] = src[index[i], :], rows of row_bytes (a multiple of 4;
 * 16-byte pieces when row_bytes % 16 == 0 and both bases are 16-byte aligned).  The frames `rollout` uploaded
 * (actor_critic_policy.py:467-469) stay in a ring of fixed-size rows in HBM; a training sample names its rows by
 * ring slot, and this gather stands in for the SECOND host-to-device crossing of the same frames in
 * PyTorchGPUPrefetcher.push (api/trainer.py:211-228).  index: int32 [n] device, src rows addressed by slot. */
int srl_gather_rows(void* stream, const void* src, int64_t row_bytes, const int32_t* index, int64_t n, void* dst);
/* Ring stamps (the int64 values a sample carries, all alive: checked by the caller on the host) -> storage slots:
 * slots[i] = (refs[i] - base) % capacity, on the device copy of the stamps, so that binding a sample costs the host two
 * passes over them (min / max) and no upload of its own.  base = what a stamp carries besides the sequence number (the
 * ring's generation in its high bits, and the +1 that keeps 0 -- what a zero-filled sample field holds -- invalid). */
int srl_ring_slots(void* stream, const int64_t* refs, int64_t n, int64_t capacity, int64_t base, int32_t* slots);
/* Stack-aware staging: a frame-stacked observation [C, H, W] (atari_wrappers.py:211-242 `FrameStack`: the C latest frames, the
 * newest last; reset() = C copies of the first frame) differs from the same environment's previous one by one plane, and the
 * previous one is still in the ring.  Row slot0 + i of `store` (rows of C*H*W uint8 in srl_obs_space_to_depth's block-4 layout)
 * becomes [channels 1..C-1 of row prev[i], planes[i]]; prev[i] < 0: C copies of planes[i].  mean / rstd [slot]: the
 * whole-observation LayerNorm statistics of the assembled row, bit-identical to srl_obs_space_to_depth's on the full stack.
 * A rollout request then carries H*W bytes instead of C*H*W over the host link (actor_critic_policy.py:467-469 uploads the
 * whole stack every step).  planes: uint8 [n, H, W] device; prev: int32 [n] device (storage slots). */
int srl_ring_stack_push(void* stream, void* store, const void* planes, const int32_t* prev, int64_t slot0, int64_t n, int C, int H,
                        int W, float* mean, float* rstd);

/* LayerNorm over D features (eps 1e-5) and the 1-2 Linear heads that read its output, in one launch per direction
 * (csrc/ln_heads.hip): the tail of the reference's shared-backbone actor-critic -- features -> actor logits, critic value
 * (actor_critic_policy.py:117-140; the CNN / MLP base ends in a LayerNorm, modules/utils.py:154-161).  D in 256 | 512 | 1024; the
 * heads' outputs together at most SRL_LN_HEADS_MAX_OUT (srl_ln_heads_supported).
 * x [n, ldx], head h: W[h] [head_dims[h], D] row-major, b[h] (may be NULL), y[h] [n, ldy[h]].  The normalised features are not
 * stored: mean / rstd [n] are, and srl_ln_heads_bwd forms the features again from x.  Backward: dy[h] [n, lddy[h]] -> dx
 * [n, lddx] = d loss / d x times the derivative of the activation that produced x (in_act, from x's value), and dgamma, dbeta,
 * dW[h], db[h] are ADDED to (float atomics); dx_absmax (optional, zeroed by the caller) receives max |dx|.
 * Forward only (inference: nothing is kept for a backward pass): x may be the raw output of a split product -- x_slabs slabs,
 * x_slab_stride floats apart (srl_h2_gemm_splitk) -- finished while it is read: x = act(sum of the slabs + x_bias), x_act as
 * srl_mlp_layer::act; x_out (optional, [n, ldxo]): the finished x is written there too -- a training pass keeps it for
 * srl_ln_heads_bwd; (1, 0, NULL, 0, NULL, 0) for a plain x. */
#define SRL_LN_HEADS_MAX_OUT 8
int srl_ln_heads_supported(int D, int n_heads, const int32_t* head_dims);
int srl_ln_heads_fwd(void* stream, const float* x, int64_t ldx, int64_t n, int D, const float* gamma, const float* beta, int n_heads,
                     const float* const* W, const float* const* b, const int32_t* head_dims, float* const* y, const int64_t* ldy,
                     float* mean, float* rstd, int x_slabs, int64_t x_slab_stride, const float* x_bias, int x_act, float* x_out,
                     int64_t ldxo);
int srl_ln_heads_bwd(void* stream, const float* x, int64_t ldx, int64_t n, int D, const float* gamma, const float* beta,
                     const float* mean, const float* rstd, int n_heads, const float* const* W, const int32_t* head_dims,
                     const float* const* dy, const int64_t* lddy, int in_act, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                     float* const* dW, float* const* db, float* dx_absmax);

/* ------------------------------------------------------------------------------------------------
 * Optimiser on one flat parameter buffer.
 * Replaces clip_grad_norm_ / get_grad_norm + torch.optim.Adam/AdamW.step
 * (mappo.py:278-284, modules/utils.py:268-295).
 */
/* A whole small MLP chain in one launch per direction (csrc/mlp_small.hip): LayerNorm / Linear (+ ReLU / tanh) layers no
 * wider than 128 -- the nn.Sequential of modules/utils.py:154-161 and the heads of actor_critic_policy.py:92-107 for the
 * CartPole-sized configurations, whose update is otherwise a chain of ~40 kernels of 3-10 us.  Layer i's input width must be
 * layer i-1's output width.  srl_mlp_fwd: y = chain(x), and every layer's input (layers 1..n-1) is left in `tape`
 * ([rows, tape_ld], tape_ld >= srl_mlp_tape_floats) for srl_mlp_bwd, which adds the parameter gradients of every layer into
 * gw / gb (float atomics across 16-row workgroups) given dy = d loss / d y.  The gradient w.r.t. x is not formed.
 * From 512 rows, chains no wider than 64 with at most 8 layers run on the float32 matrix cores (csrc/mlp_mfma.h) and keep NO
 * tape -- srl_mlp_bwd walks the chain forward again from x, which must still hold the rows srl_mlp_fwd saw, under the same
 * parameters: srl_mlp_tape_floats_at returns 0 for such a (chain, rows) pair and `tape` may then be NULL in both calls. */
#define SRL_MLP_MAX_LAYERS 12
typedef struct srl_mlp_layer {
  int32_t kind;       /* 0: LayerNorm over `in` (w = gamma, b = beta, eps 1e-5); 1: Linear, w [out, in] row-major, b [out] */
  int32_t in, out;    /* widths, 1..128 (LayerNorm: out ignored) */
  int32_t act;        /* Linear: 0 none, 1 relu, 2 tanh, applied after the bias */
  const float* w;
  const float* b;
  float* gw;          /* gradients (srl_mlp_bwd): same shapes as w / b, accumulated into */
  float* gb;
} srl_mlp_layer;
int64_t srl_mlp_tape_floats(const srl_mlp_layer* layers, int n); /* floats per tape row; -1: chain not supported */
int64_t srl_mlp_tape_floats_at(const srl_mlp_layer* layers, int n, int64_t rows); /* ... at this row count (0: no tape kept) */
/* rows up to which the pair is worth taking over the layer-by-layer kernels (measured): 32768 when the chain's parameter
 * gradients (<= 12288 floats) are summed in LDS over a workgroup's row blocks, 8192 when every 16-row block has to add its
 * sums with float atomics; 0: not supported */
int64_t srl_mlp_bwd_max_rows(const srl_mlp_layer* layers, int n);
int srl_mlp_fwd(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows, float* tape,
                int64_t tape_ld, float* y, int64_t ldy);
int srl_mlp_bwd(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                const float* tape, int64_t tape_ld, const float* dy, int64_t lddy);
/* ... and d loss / d x into dx [rows, lddx] as well -- for a chain that sits BEHIND other layers (the LayerNorm + head after a
 * recurrent cell, actor_critic_policy.py:117-140 with modules/recurrent.py's cells in front): only for (chain, rows) pairs that
 * keep no tape (srl_mlp_tape_floats_at == 0); x carries no pending activation. */
int srl_mlp_bwd_dx(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows, const float* dy,
                   int64_t lddy, float* dx, int64_t lddx);
/* sumsq[0] = sum g^2 in float64 (zeroed first).  With data parallelism the caller all-reduces the
 * gradients before this call (DDP semantics), so no further reduction is needed. */
int srl_grad_sumsq(void* stream, const float* g, int64_t n, double* sumsq);
/* dst[i] += src[i]: the flat gradients of two row-chunk pipelines that ran side by side (each accumulates into its own
 * buffer: loss.backward() of mappo.py:274 over the chunks of one batch) become one before clip + optimiser. */
int srl_accumulate(void* stream, float* dst, const float* src, int64_t n);
/* dst[i] += srcs[0][i] + ... + srcs[k-1][i], added left to right (ABI 15): the slices of ALL the other pipelines into the first
 * buffer in ONE launch when a gradient bucket is folded (k launches before: with four pipelines three per bucket, each with its
 * host latency in the tail of a data-parallel update).  srcs: HOST array of k <= SRL_ACCUMULATE_MAX device pointers. */
#define SRL_ACCUMULATE_MAX 7
int srl_accumulate_n(void* stream, float* dst, const float* const* srcs, int32_t k, int64_t n);
/* Adam step with the clip coefficient min(1, max_norm / (sqrt(sumsq) + 1e-6)) applied to g on the
 * fly (max_norm < 0: no clipping; sumsq may then be NULL).  grad_scale multiplies g first
 * (1/world_size for the DDP mean).  step is the 1-based step count; weight_decay is decoupled
 * (AdamW) when adamw != 0, L2 (added to g) otherwise.  grad_norm_out float32[1] or NULL.
 * step_scalars: NULL, or device float32[2] = {lr / (1 - beta1^step), sqrt(1 - beta2^step)} read by the kernel
 * instead of deriving them from `step` -- the launch then carries no per-step scalar and can sit in a captured
 * hipGraph that is replayed every step (the caller refreshes the two floats before each replay). */
int srl_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int adamw, int64_t step, float grad_scale,
                  float max_norm, const double* sumsq, float* grad_norm_out,
                  const float* step_scalars);

/* `Convolution` encoder pieces beyond unpadded convolutions (legacy/algorithm/modules/cnn.py:99-118; policies/utils.py:53):
 * srl_obs_ln_nhwc(_bwd): the whole-observation LayerNorm as an explicit pass, planar uint8 / float32 observation
 *   [n,C,H,W] -> channels-last float32 [n,H,W,C] (gamma / beta [C,H,W]); backward accumulates dgamma, dbeta from
 *   dy [n,H,W,C].  srl_pad_nhwc / srl_crop_nhwc: zero padding of an NHWC activation and its adjoint (nn.Conv2d(padding=p,
 *   padding_mode='zeros')).  srl_maxpool2_nhwc_fwd / _bwd: MaxPool2d(2) (`use_maxpool`: in front of every layer but the
 *   last); the backward routes to the first maximum of a window (torch's rule) and multiplies by act'(x) of the
 *   activation that produced x (dact 0 none, 1 relu, 2 tanh). */
int srl_obs_ln_nhwc(void* stream, const void* obs, int is_u8, const float* mean, const float* rstd, const float* gamma,
                    const float* beta, int64_t n, int C, int H, int W, float* y);
int srl_obs_ln_nhwc_bwd(void* stream, const float* dy, const void* obs, int is_u8, const float* mean, const float* rstd,
                        int64_t n, int C, int H, int W, float* dgamma, float* dbeta);
int srl_pad_nhwc(void* stream, const float* x, int64_t n, int H, int W, int C, int pad, float* y);
int srl_crop_nhwc(void* stream, const float* yp, int64_t n, int H, int W, int C, int pad, float* x);
int srl_maxpool2_nhwc_fwd(void* stream, const float* x, int64_t n, int H, int W, int C, float* y);
int srl_maxpool2_nhwc_bwd(void* stream, const float* dy, const float* x, int64_t n, int H, int W, int C, int dact, float* dx);

/* The same encoder pieces for observations with one or three spatial dimensions (modules/cnn.py:60-71: nn.Conv1d /
 * nn.Conv3d with MaxPool1d / MaxPool3d by the observation's rank; modules_test.py:385-401), on channels-last volumes
 * [n, D, H, W, C] (Conv1d: D = H = 1).  srl_pad_ndhwc: border of (pd, ph, pw <= 8) voxels per side filled per
 * nn.ConvNd's `padding_mode` (mode 0 zeros, 1 reflect, 2 replicate, 3 circular; torch's preconditions: reflect pad <
 * extent, circular pad <= extent); srl_crop_ndhwc: its adjoint (mode 0: the crop; else every interior voxel also collects
 * the border voxels that were filled from it).  srl_maxpool_ndhwc_fwd / _bwd: windows of (wd, wh, ww) voxels, each 1 or 2, stride = window, floor; the
 * backward as srl_maxpool2_nhwc_bwd.  srl_im2col_ndhwc: patch matrix P[(s,od,oh,ow)][(kd,kh,kw,c)] for a dense GEMM
 * against weights laid out [Cout][KD][KH][KW][Cin]; srl_col2im_ndhwc: its adjoint (sums the taps that reach a voxel,
 * optionally times act'(y), y = the forward activation of that voxel, dact as above). */
int srl_pad_ndhwc(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int pd, int ph, int pw, int mode,
                  float* y);
int srl_crop_ndhwc(void* stream, const float* yp, int64_t n, int D, int H, int W, int C, int pd, int ph, int pw, int mode,
                   float* x);
int srl_maxpool_ndhwc_fwd(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int wd, int wh, int ww, float* y);
int srl_maxpool_ndhwc_bwd(void* stream, const float* dy, const float* x, int64_t n, int D, int H, int W, int C, int wd, int wh,
                          int ww, int dact, float* dx);
int srl_im2col_ndhwc(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int KD, int KH, int KW, int stride,
                     float* P);
int srl_col2im_ndhwc(void* stream, const float* dP, int64_t n, int D, int H, int W, int C, int KD, int KH, int KW, int stride,
                     const float* y, int dact, float* dX);

/* torch.optim.SGD / torch.optim.RMSprop on the flat buffer (modules/utils.py:268-286 accepts 'sgd' and 'rmsprop'
 * with their torch keyword configurations), same clip / grad_scale / grad_norm_out conventions as srl_adam_step.
 * SGD: momentum_buf may be NULL when momentum == 0; first_step != 0 initialises the buffer with the gradient
 * (torch's first step).  RMSprop: momentum_buf NULL when momentum == 0, grad_avg NULL unless centered. */
int srl_sgd_step(void* stream, float* p, const float* g, float* momentum_buf, int64_t n, float lr, float momentum,
                 float dampening, float weight_decay, int nesterov, int first_step, float grad_scale, float max_norm,
                 const double* sumsq, float* grad_norm_out);
int srl_rmsprop_step(void* stream, float* p, const float* g, float* square_avg, float* momentum_buf, float* grad_avg,
                     int64_t n, float lr, float alpha, float eps, float weight_decay, float momentum, int centered,
                     float grad_scale, float max_norm, const double* sumsq, float* grad_norm_out);

/* ------------------------------------------------------------------------------------------------
 * Native step driver (launch-bound configurations).  The device part of MultiAgentPPO.step (mappo.py:219-328) is
 * captured once into a hipGraph; a plan = that executable graph (hipGraphExec_t as void*) + its static input leaves +
 * its outputs.  srl_step_plan_run enqueues, on `stream`: one copy per input from srcs[i] (host or device memory; NULL
 * = leave the static leaf as it is) into the static leaf, the graph launch, one copy per output to its host
 * destination (fixed at registration or, if that was NULL, host_dsts[i]); sync != 0 waits for the stream.  One FFI
 * crossing per trainer step instead of one host operation per leaf.  add_input / add_output return the slot index. */
int srl_step_plan_create(void** plan_out, void* graph_exec);
int srl_step_plan_add_input(void* plan, void* dst_device, int64_t nbytes);
int srl_step_plan_add_output(void* plan, const void* src_device, int64_t nbytes, void* host_dst);
int srl_step_plan_run(void* plan, void* stream, const void* const* srcs, int n_srcs, void* const* host_dsts, int n_dsts,
                      int sync);
int srl_step_plan_destroy(void* plan);

/* ------------------------------------------------------------------------------------------------
 * Collectives (RCCL over xGMI), one process per GPU.  `comm` is an ncclComm_t behind void*; every
 * call is enqueued on `stream` and returns at once (stream-ordered like the kernels above).  librccl
 * is resolved with dlopen at first use (-ENOSYS if absent).
 *
 * Bootstrap: one rank calls srl_comm_unique_id and hands the SRL_COMM_ID_BYTES bytes to the others by
 * any side channel (the Python mirror broadcasts them through the process group the reference already
 * forms in `trainer.distributed`, api/trainer.py:113-128); every rank then calls srl_comm_init with its
 * rank, on the device it owns.
 *
 * srl_allreduce_stats_f64x3: in-place SUM of `count` float64 values.  Replaces the three all_reduce
 *   calls of masked_normalization (legacy/algorithm/modules/utils.py:58-61; count = 3: n, sum, sumsq)
 *   and of RunningMeanStd.update (utils.py:121-124; count = 3 * value_dim) with one message each.
 * srl_allreduce_grads: in-place SUM of n float32 gradients (a bucket of the flat gradient buffer).
 *   Replaces DistributedDataParallel's bucketed all-reduce (api/policy.py:219-238); the mean over
 *   ranks is the grad_scale of srl_adam_step.
 * srl_broadcast_params: `nbytes` bytes from rank `root` to all (the flat float32 parameter buffer,
 *   PopArt float64 statistics, the int64 version).  Replaces the DDP constructor's parameter
 *   broadcast (same lines) and the parameter-server push / pull between trainer and policy workers
 *   (distributed/system/parameter_db.py:250-324) for replicas on the node.
 */
#define SRL_COMM_ID_BYTES 128
/* 1 when librccl and every entry point the wrappers use resolve in this process, else 0.  Touches no device and forms no
 * communicator: ranks agree on this flag (a MIN all-reduce over their existing process group) BEFORE any of them enters
 * srl_comm_init, whose ncclCommInitRank blocks until every rank has joined. */
int srl_comm_available(void);
int srl_comm_unique_id(void* id_out);
int srl_comm_init(void** comm_out, const void* id, int rank, int world);
int srl_comm_world(void* comm, int* world_out);
int srl_comm_destroy(void* comm);
int srl_allreduce_stats_f64x3(void* stream, void* comm, double* stats, int count);
int srl_allreduce_grads(void* stream, void* comm, float* grad, int64_t n);
int srl_broadcast_params(void* stream, void* comm, void* buf, int64_t nbytes, int root);

#ifdef __cplusplus
}
#endif
#endif /* SRL_HIP_H_ */
