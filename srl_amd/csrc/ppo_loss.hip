// PPO loss forward+backward, categorical heads (log-prob / entropy / gradient / Philox sampling).
// All of it is HBM-bound streaming work: one thread per sample, coalesced float32 streams, float64
// block reductions for the loss terms.
#include "srl_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// value loss pieces (modules/utils.py:228-265): l(v, target) and dl/dv
__device__ __forceinline__ void vloss(int kind, float delta, float v, float t, float& l, float& dl) {
  const float d = v - t;
  if (kind == 0) {  // nn.MSELoss(reduction='none')
    l = d * d;
    dl = 2.0f * d;
    return;
  }
  const float a = fabsf(d);
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  if (kind == 1) {  // nn.HuberLoss(delta)
    if (a <= delta) { l = 0.5f * d * d; dl = d; }
    else { l = delta * (a - 0.5f * delta); dl = delta * sgn; }
  } else {  // nn.SmoothL1Loss(beta = delta)
    if (a < delta) { l = 0.5f * d * d / delta; dl = d / delta; }
    else { l = a - 0.5f * delta; dl = sgn; }
  }
}

struct LossArgs {
  const float *new_lp, *old_lp, *value, *old_value, *adv, *ret, *entropy;
  const uint8_t* mask;
  long n;
  int vd;  // value channels: value / old_value / adv / ret / d_value are [n, vd], the rest [n] (broadcast over channels)
  srl_ppo_hparams hp;
  const double* norm_stats;
  const double* local_n;
  const uint8_t *done, *truncated;
  float *d_new_lp, *d_value, *d_entropy;
  double* terms;
};

__global__ __launch_bounds__(256) void ppo_loss_kernel(LossArgs a) {
  __shared__ double red[4 * SRL_LT_COUNT];
  const srl_ppo_hparams hp = a.hp;
  // advantage normalisation constants in float64 (utils.py:62-67)
  const double cnt = a.norm_stats[0];
  const double mean = a.norm_stats[1] / cnt;
  const double var = a.norm_stats[2] / cnt - mean * mean;
  const double denom = sqrt(var) + (double)hp.norm_eps;
  const float inv_n = (float)(1.0 / a.local_n[0]);  // every masked mean divides by the LOCAL mask count
  const float lo = 1.0f - hp.eps_clip, hi = 1.0f + hp.eps_clip;

  double acc[SRL_LT_COUNT];
#pragma unroll
  for (int i = 0; i < SRL_LT_COUNT; ++i) acc[i] = 0.0;

  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    const float m = hp.mask_invert ? 1.0f - (float)a.mask[i] : (float)a.mask[i];
    const float ratio = expf(a.new_lp[i] - a.old_lp[i]);  // mappo.py:157
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const bool in_range = ratio >= lo && ratio <= hi;
    const float w = m * inv_n;
    const double md = (double)m;
    float g_lp = 0.f;
    // value channels: every term below is [T, B, vd] in the reference (ratio, mask and entropy broadcast over the
    // channels) and every masked mean divides the sum over ALL channels by the mask count (mappo.py:184,197)
    for (int c = 0; c < a.vd; ++c) {
      const long e = i * a.vd + c;
      const float adv = a.adv[e];
      const float nadv = (float)(((double)adv * (double)m - mean) / denom);  // mappo.py:187
      const float s1 = ratio * nadv, s2 = rc * nadv;  // mappo.py:188-190
      // d min(s1,s2) / d ratio with torch's tie rule (equal -> half through each branch)
      float gmin;
      if (s1 < s2) gmin = nadv;
      else if (s1 > s2) gmin = in_range ? nadv : 0.f;
      else gmin = 0.5f * nadv + (in_range ? 0.5f * nadv : 0.f);
      const float mn = fminf(s1, s2);
      float pl, gfac;
      if (hp.dual_clip) {
        const float sg = nadv > 0.f ? 1.f : (nadv < 0.f ? -1.f : 0.f);
        const float s3 = -sg * hp.c_clip * nadv;  // mappo.py:192
        pl = -fmaxf(mn, s3);
        gfac = mn > s3 ? 1.f : (mn == s3 ? 0.5f : 0.f);
      } else {
        pl = -mn;
        gfac = 1.f;
      }
      // value loss (utils.py:230-237 when clipped)
      const float v = a.value[e], tgt = a.ret[e];
      float l1, dl1;
      vloss(hp.value_loss, hp.huber_delta, v, tgt, l1, dl1);
      float vl = l1, dvl = dl1;
      if (hp.clip_value) {
        const float vo = a.old_value[e];
        const float dv = v - vo;
        const float vc = vo + fminf(fmaxf(dv, -hp.value_eps_clip), hp.value_eps_clip);
        float l2, dl2;
        vloss(hp.value_loss, hp.huber_delta, vc, tgt, l2, dl2);
        dl2 = (dv >= -hp.value_eps_clip && dv <= hp.value_eps_clip) ? dl2 : 0.f;
        if (l1 > l2) { vl = l1; dvl = dl1; }
        else if (l1 < l2) { vl = l2; dvl = dl2; }
        else { vl = l1; dvl = 0.5f * (dl1 + dl2); }
      }
      g_lp += -gfac * gmin * ratio * w;                    // d(policy_loss)/d new_lp, summed over the channels
      a.d_value[e] = hp.value_loss_weight * dvl * w;       // mappo.py:202
      acc[SRL_LT_POLICY] += md * (double)pl;
      acc[SRL_LT_VALUE] += md * (double)vl;
      acc[SRL_LT_CLIP] += md * (s2 < s1 ? 1.0 : 0.0);  // mappo.py:214
      acc[SRL_LT_ADV] += md * (double)adv;
      acc[SRL_LT_RET] += md * (double)tgt;
    }
    const float ent = a.entropy[i];
    a.d_new_lp[i] = g_lp;
    a.d_entropy[i] = -hp.entropy_bonus_weight * w;
    acc[SRL_LT_ENTROPY] += md * (double)ent;
    acc[SRL_LT_RATIO] += md * (double)ratio;
    acc[SRL_LT_MASK] += md;
    if (a.done) acc[SRL_LT_DONE] += (double)a.done[i];
    if (a.truncated) acc[SRL_LT_TRUNC] += (double)a.truncated[i];
  }
  block_sum<SRL_LT_COUNT, 256>(acc, red);
  // the block's SRL_LT_COUNT sums leave as ONE atomic instruction (lane i adds term i: one request for the line they share)
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < SRL_LT_COUNT; ++i) red[i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < SRL_LT_COUNT) atomicAdd(&a.terms[threadIdx.x], red[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
struct Heads {
  int n_heads;
  int dims[SRL_MAX_HEADS];
};

constexpr float kMaskedLogit = -1e10f;  // actor_critic_policy.py:136

__device__ __forceinline__ float masked_logit(const float* row, const uint8_t* av, int j) {
  return (av && av[j] == 0) ? kMaskedLogit : row[j];
}

__global__ __launch_bounds__(256) void categorical_fwd_kernel(const float* logits, int ld, const int32_t* action,
                                                              const uint8_t* avail, long n, Heads h, int atot,
                                                              float* logp, float* entropy) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* row = logits + i * ld;
  const uint8_t* av = avail ? avail + i * atot : nullptr;
  float lp = 0.f, ent = 0.f;
  int s = 0;
  for (int k = 0; k < h.n_heads; ++k) {
    const int d = h.dims[k];
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) mx = fmaxf(mx, masked_logit(row, av, s + j));
    float se = 0.f;
    for (int j = 0; j < d; ++j) se += expf(masked_logit(row, av, s + j) - mx);
    const float lse = mx + logf(se);
    float hk = 0.f;
    for (int j = 0; j < d; ++j) {
      const float nl = masked_logit(row, av, s + j) - lse;  // normalised logit = log p
      hk -= expf(nl) * nl;                                   // Categorical.entropy
    }
    lp += masked_logit(row, av, s + action[i * h.n_heads + k]) - lse;
    ent += hk;
    s += d;
  }
  logp[i] = lp;
  entropy[i] = ent;
}

__global__ __launch_bounds__(256) void categorical_bwd_kernel(const float* logits, int ld, const int32_t* action,
                                                              const uint8_t* avail, long n, Heads h, int atot,
                                                              const float* d_logp, const float* d_entropy,
                                                              float* d_logits, int ldd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* row = logits + i * ld;
  const uint8_t* av = avail ? avail + i * atot : nullptr;
  float* drow = d_logits + i * ldd;
  const float glp = d_logp[i], gent = d_entropy[i];
  int s = 0;
  for (int k = 0; k < h.n_heads; ++k) {
    const int d = h.dims[k];
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) mx = fmaxf(mx, masked_logit(row, av, s + j));
    float se = 0.f;
    for (int j = 0; j < d; ++j) se += expf(masked_logit(row, av, s + j) - mx);
    const float lse = mx + logf(se);
    float hk = 0.f;
    for (int j = 0; j < d; ++j) {
      const float nl = masked_logit(row, av, s + j) - lse;
      hk -= expf(nl) * nl;
    }
    const int act = action[i * h.n_heads + k];
    for (int j = 0; j < d; ++j) {
      const float nl = masked_logit(row, av, s + j) - lse;
      const float pj = expf(nl);
      // d logp / d z_j = 1[j==a] - p_j ;  d H / d z_j = -p_j (log p_j + H)
      float g = glp * ((j == act ? 1.f : 0.f) - pj) - gent * pj * (nl + hk);
      if (av && av[s + j] == 0) g = 0.f;  // overwritten logits receive no gradient (:135-136)
      drow[s + j] = g;
    }
    s += d;
  }
}

// ---- Philox4x32-10 (Salmon et al. 2011) -----------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}

__global__ __launch_bounds__(256) void categorical_sample_kernel(const float* logits, int ld, const uint8_t* avail,
                                                                 const uint8_t* is_eval, long n, Heads h, int atot,
                                                                 uint64_t seed, uint64_t offset, int64_t* action_out,
                                                                 float* logp, long row0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr_row = (uint64_t)(row0 + i);  // the row's number in the caller's whole batch: pieces of it sample alike
  const float* row = logits + i * ld;
  const uint8_t* av = avail ? avail + i * atot : nullptr;
  const bool greedy = is_eval && is_eval[i];
  float lp = 0.f;
  int s = 0;
  for (int k = 0; k < h.n_heads; ++k) {
    const int d = h.dims[k];
    float mx = -INFINITY;
    int amax = 0;
    for (int j = 0; j < d; ++j) {
      const float z = masked_logit(row, av, s + j);
      if (z > mx) { mx = z; amax = j; }  // first maximum, like argmax on probs
    }
    float se = 0.f;
    for (int j = 0; j < d; ++j) se += expf(masked_logit(row, av, s + j) - mx);
    int pick = amax;
    if (!greedy) {
      const uint4 rnd = philox4x32_10(make_uint4((uint32_t)ctr_row, (uint32_t)(ctr_row >> 32), (uint32_t)k,
                                                 (uint32_t)offset),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
      const float u = (float)(rnd.x >> 8) * (1.0f / 16777216.0f) * se;  // uniform in [0, se)
      float cdf = 0.f;
      pick = -1;
      int last_ok = amax;
      for (int j = 0; j < d; ++j) {
        const float e = expf(masked_logit(row, av, s + j) - mx);
        if (e > 0.f) last_ok = j;
        cdf += e;
        if (pick < 0 && u < cdf) pick = j;
      }
      if (pick < 0) pick = last_ok;  // rounding at the top of the CDF
    }
    action_out[i * h.n_heads + k] = pick;
    lp += masked_logit(row, av, s + pick) - (mx + logf(se));
    s += d;
  }
  logp[i] = lp;
}

int make_heads(int n_heads, const int32_t* dims, Heads& h, int& atot) {
  if (n_heads < 1 || n_heads > SRL_MAX_HEADS || !dims) return -1;
  h.n_heads = n_heads;
  atot = 0;
  for (int k = 0; k < n_heads; ++k) {
    if (dims[k] < 1) return -1;
    h.dims[k] = dims[k];
    atot += dims[k];
  }
  return 0;
}

}  // namespace

extern "C" int srl_ppo_loss_fwd_bwd(void* stream, const float* new_lp, const float* old_lp, const float* value,
                                    const float* old_value, const float* adv, const float* ret, const float* entropy,
                                    const uint8_t* mask, long n, int value_dim, const srl_ppo_hparams* hp,
                                    const double* norm_stats, const double* local_n, const uint8_t* done,
                                    const uint8_t* truncated, float* d_new_lp, float* d_value, float* d_entropy,
                                    double* loss_terms) {
  SRL_CHECK_ARG(value_dim >= 1 && n >= 0, "value_dim >= 1 and n >= 0 required");
  SRL_CHECK_ARG(new_lp && old_lp && value && adv && ret && entropy && mask && hp && norm_stats && local_n,
                "null input");
  SRL_CHECK_ARG(d_new_lp && d_value && d_entropy && loss_terms, "null output");
  SRL_CHECK_ARG(!hp->clip_value || old_value, "clip_value needs old_value");
  SRL_CHECK_ARG(hp->value_loss >= 0 && hp->value_loss <= 2, "value_loss must be 0|1|2");
  hipStream_t st = (hipStream_t)stream;
  SRL_HIP_TRY(hipMemsetAsync(loss_terms, 0, SRL_LT_COUNT * sizeof(double), st));
  if (n == 0) return 0;
  LossArgs a{new_lp, old_lp, value, old_value, adv, ret, entropy, mask, n, value_dim, *hp, norm_stats, local_n,
             done, truncated, d_new_lp, d_value, d_entropy, loss_terms};
  // (every block ends in SRL_LT_COUNT float64 atomics on ONE line, which the L2 serialises at ~10 ns each: 2048 blocks
  // spent 0.2 ms there at 524 288 rows; the grid is capped where the loop's loads still cover the latency)
  static const long cap = getenv("SRL_LOSS_GRID") ? atol(getenv("SRL_LOSS_GRID")) : 512;
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < cap ? srl_ceil_div(n, 256) : cap);
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(grid), dim3(256), 0, st, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_categorical_fwd(void* stream, const float* logits, int ld_logits, const int32_t* action,
                                   const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims,
                                   float* logp, float* entropy) {
  Heads h;
  int atot;
  SRL_CHECK_ARG(make_heads(n_heads, host_head_dims, h, atot) == 0, "bad head dims");
  SRL_CHECK_ARG(logits && action && logp && entropy && ld_logits >= atot, "null tensor / ld");
  if (n == 0) return 0;
  hipLaunchKernelGGL(categorical_fwd_kernel, dim3((unsigned)srl_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, ld_logits, action, avail, n, h, atot, logp, entropy);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_categorical_bwd(void* stream, const float* logits, int ld_logits, const int32_t* action,
                                   const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims,
                                   const float* d_logp, const float* d_entropy, float* d_logits, int ld_dlogits) {
  Heads h;
  int atot;
  SRL_CHECK_ARG(make_heads(n_heads, host_head_dims, h, atot) == 0, "bad head dims");
  SRL_CHECK_ARG(logits && action && d_logp && d_entropy && d_logits && ld_logits >= atot && ld_dlogits >= atot,
                "null tensor / ld");
  if (n == 0) return 0;
  hipLaunchKernelGGL(categorical_bwd_kernel, dim3((unsigned)srl_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, ld_logits, action, avail, n, h, atot, d_logp, d_entropy, d_logits, ld_dlogits);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_categorical_sample(void* stream, const float* logits, int ld_logits, const uint8_t* avail,
                                      const uint8_t* is_eval, long n, int n_heads, const int32_t* host_head_dims,
                                      uint64_t seed, uint64_t offset, int64_t* action_out, float* logp, int64_t row0) {
  Heads h;
  int atot;
  SRL_CHECK_ARG(make_heads(n_heads, host_head_dims, h, atot) == 0, "bad head dims");
  SRL_CHECK_ARG(logits && action_out && logp && ld_logits >= atot, "null tensor / ld");
  if (n == 0) return 0;
  hipLaunchKernelGGL(categorical_sample_kernel, dim3((unsigned)srl_ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, logits, ld_logits, avail, is_eval, n, h, atot, seed, offset, action_out,
                     logp, (long)row0);
  SRL_LAUNCH_CHECK();
  return 0;
}
