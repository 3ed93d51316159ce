// Phasic Policy Gradient, auxiliary phase (phasic_policy_gradient.py:140-153, 262-280): the action distributions a cache entry
// keeps, and the joint loss  L = aux_value_loss + beta_clone * policy_distance + value_head_weight * value_head_loss  with its
// gradient with respect to the network's outputs, in one pass over the rows.
//   policy_distance  = sum over heads of  sum_rows KL(old_h || new_h) (1 - done) / sum_rows (1 - done)      (:140-144)
//   *_value_loss     = 1/2 sum (v - target)^2 (1 - done) / sum_rows (1 - done)                               (:146-147)
// HBM-bound row work (a few dozen floats per row): one thread per row, no matrix cores.
#include "srl_common.h"

namespace {

struct Heads {
  int n_heads;
  int dims[SRL_MAX_HEADS];
};

constexpr float kMaskedLogit = -1e10f;  // actor_critic_policy.py:136

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

__device__ __forceinline__ float masked_logit(const float* row, const uint8_t* av, int j) {
  return (av && av[j] == 0) ? kMaskedLogit : row[j];
}

// out[i, head segment] = masked logits - logsumexp: what torch.distributions.Categorical(logits=...) keeps as `.logits`
__global__ __launch_bounds__(256) void categorical_log_softmax_kernel(const float* logits, int ld, const uint8_t* avail, long n, Heads h,
                                                                      int atot, float* out, int ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* row = logits + i * ld;
  const uint8_t* av = avail ? avail + i * atot : nullptr;
  float* o = out + i * ldo;
  int s = 0;
  for (int k = 0; k < h.n_heads; ++k) {
    const int d = h.dims[k];
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) mx = fmaxf(mx, masked_logit(row, av, s + j));
    float se = 0.f;
    for (int j = 0; j < d; ++j) se += expf(masked_logit(row, av, s + j) - mx);
    const float lse = mx + logf(se);
    for (int j = 0; j < d; ++j) o[s + j] = masked_logit(row, av, s + j) - lse;
    s += d;
  }
}

struct AuxArgs {
  const float* logq_old;  // [n, atot] normalised log-probabilities of the distributions kept by the cache entry
  const float* logits;    // [n, atot] raw logits of the current policy
  const uint8_t* avail;   // [n, atot] or null
  const float* aux;       // [n, vd]
  const float* pred;      // [n, vd]
  const float* target;    // [n, vd]
  const uint8_t* done;    // [n] (the cache entry's info_mask, :264)
  const double* count;    // sum over rows of (1 - done)
  long n;
  int ld_old, ld, vd, atot;
  float beta_clone, value_head_weight;
  float* d_logits;        // [n, atot] (pitch ldd)
  int ldd;
  float* d_aux;           // [n, vd]
  float* d_pred;          // [n, vd]
  double* terms;          // [3]: auxiliary value loss, value head loss, policy distance (each already divided by count)
  Heads h;
};

__global__ __launch_bounds__(256) void ppg_aux_loss_kernel(AuxArgs a) {
  __shared__ double red[4 * 3];
  double acc[3] = {0.0, 0.0, 0.0};
  const double cnt = *a.count;
  const float inv = cnt > 0.0 ? (float)(1.0 / cnt) : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    const float und = 1.f - (float)a.done[i];
    const float w = und * inv;
    const float* row = a.logits + i * a.ld;
    const float* old = a.logq_old + i * a.ld_old;
    const uint8_t* av = a.avail ? a.avail + i * a.atot : nullptr;
    float* drow = a.d_logits + i * a.ldd;
    int s = 0;
    float kl = 0.f;
    for (int k = 0; k < a.h.n_heads; ++k) {
      const int d = a.h.dims[k];
      float mx = -INFINITY;
      for (int j = 0; j < d; ++j) mx = fmaxf(mx, masked_logit(row, av, s + j));
      float se = 0.f;
      for (int j = 0; j < d; ++j) se += expf(masked_logit(row, av, s + j) - mx);
      const float lse = mx + logf(se);
      for (int j = 0; j < d; ++j) {
        const float lq = masked_logit(row, av, s + j) - lse, lp = old[s + j];
        const float p = expf(lp), q = expf(lq);
        // torch.distributions.kl._kl_categorical_categorical: p (log p - log q), +inf where q == 0, 0 where p == 0
        float t = p * (lp - lq);
        if (q == 0.f) t = INFINITY;
        if (p == 0.f) t = 0.f;
        kl += t;
        // d KL / d z_j = q_j - p_j (sum p = 1); logits overwritten with the mask constant receive no gradient (:135-136)
        drow[s + j] = (av && av[s + j] == 0) ? 0.f : a.beta_clone * w * (q - p);
      }
      s += d;
    }
    acc[2] += (double)(kl * und);
    for (int c = 0; c < a.vd; ++c) {
      const long e = i * a.vd + c;
      const float da = a.aux[e] - a.target[e], dp = a.pred[e] - a.target[e];
      a.d_aux[e] = da * w;
      a.d_pred[e] = a.value_head_weight * dp * w;
      acc[0] += 0.5 * (double)(da * da * und);
      acc[1] += 0.5 * (double)(dp * dp * und);
    }
  }
  block_sum<3, 256>(acc, red);
  if (threadIdx.x == 0 && cnt > 0.0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) atomicAdd(&a.terms[i], acc[i] / cnt);
  }
}

}  // namespace

extern "C" int srl_categorical_log_softmax(void* stream, const float* logits, int ld_logits, const uint8_t* avail, long n, int n_heads,
                                           const int32_t* host_head_dims, float* out, int ld_out) {
  Heads h;
  int atot;
  SRL_CHECK_ARG(make_heads(n_heads, host_head_dims, h, atot) == 0, "bad head dims");
  SRL_CHECK_ARG(logits && out && ld_logits >= atot && ld_out >= atot && n >= 0, "null tensor / ld");
  if (n == 0) return 0;
  hipLaunchKernelGGL(categorical_log_softmax_kernel, dim3((unsigned)srl_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     ld_logits, avail, n, h, atot, out, ld_out);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_ppg_aux_loss_fwd_bwd(void* stream, const float* logq_old, int ld_old, const float* logits, int ld_logits,
                                        const uint8_t* avail, long n, int n_heads, const int32_t* host_head_dims, const float* aux_value,
                                        const float* pred_value, const float* target, int value_dim, const uint8_t* done,
                                        const double* undone_count, float beta_clone, float value_head_weight, float* d_logits,
                                        int ld_dlogits, float* d_aux, float* d_pred, double* terms) {
  AuxArgs a{};
  SRL_CHECK_ARG(make_heads(n_heads, host_head_dims, a.h, a.atot) == 0, "bad head dims");
  SRL_CHECK_ARG(logq_old && logits && aux_value && pred_value && target && done && undone_count && d_logits && d_aux && d_pred && terms,
                "null tensor");
  SRL_CHECK_ARG(ld_old >= a.atot && ld_logits >= a.atot && ld_dlogits >= a.atot && value_dim >= 1 && n >= 0, "ld / value_dim");
  hipStream_t st = (hipStream_t)stream;
  SRL_HIP_TRY(hipMemsetAsync(terms, 0, 3 * sizeof(double), st));
  if (n == 0) return 0;
  a.logq_old = logq_old; a.logits = logits; a.avail = avail; a.aux = aux_value; a.pred = pred_value; a.target = target; a.done = done;
  a.count = undone_count; a.n = n; a.ld_old = ld_old; a.ld = ld_logits; a.vd = value_dim; a.beta_clone = beta_clone;
  a.value_head_weight = value_head_weight; a.d_logits = d_logits; a.ldd = ld_dlogits; a.d_aux = d_aux; a.d_pred = d_pred; a.terms = terms;
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(ppg_aux_loss_kernel, dim3(grid), dim3(256), 0, st, a);
  SRL_LAUNCH_CHECK();
  return 0;
}
