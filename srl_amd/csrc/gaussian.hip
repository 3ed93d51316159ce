// Diagonal Gaussian action head (continuous actions): torch.distributions.Normal(mean, std) as the reference uses
// it (legacy/algorithm/ppo/actor_critic_policies/actor_critic_policy.py:128-133, 318-321, 499-506).
//   log_prob(x) = -(x - mu)^2 / (2 sigma^2) - log sigma - log sqrt(2 pi),   entropy = 1/2 + log sqrt(2 pi) + log sigma
// summed over the action dimensions; log sigma is either one vector shared by all rows (`fixed`,
// `separate_learnable`: ld_ls = 0) or one row per sample from a second head (`shared_learnable`: ld_ls = its pitch).
#include "srl_common.h"

namespace {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // log sqrt(2 pi)

__global__ __launch_bounds__(256) void gaussian_fwd_kernel(const float* mean, int ld_mean, const float* log_std,
                                                           int ld_ls, const float* action, long n, int A, float* logp,
                                                           float* ent) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float lp = 0.f, en = 0.f;
  for (int j = 0; j < A; ++j) {
    const float ls = log_std[i * ld_ls + j];
    const float sd = expf(ls);
    const float d = action[i * A + j] - mean[i * ld_mean + j];
    lp += -(d * d) / (2.f * sd * sd) - logf(sd) - kHalfLog2Pi;  // torch/distributions/normal.py log_prob
    en += 0.5f + kHalfLog2Pi + logf(sd);                          // ... entropy
  }
  logp[i] = lp;
  ent[i] = en;
}

// d mean = d_logp * (x - mu) / sigma^2;  d log sigma = d_logp * ((x - mu)^2 / sigma^2 - 1) + d_ent   (per row)
__global__ __launch_bounds__(256) void gaussian_bwd_kernel(const float* mean, int ld_mean, const float* log_std,
                                                           int ld_ls, const float* action, long n, int A,
                                                           const float* d_logp, const float* d_ent, float* d_mean,
                                                           float* d_log_std) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n * A; e += (long)gridDim.x * 256) {
    const long i = e / A;
    const int j = (int)(e - i * A);
    const float sd = expf(log_std[i * ld_ls + j]);
    const float d = action[e] - mean[i * ld_mean + j];
    const float q = d / (sd * sd);
    d_mean[e] = d_logp[i] * q;
    d_log_std[e] = d_logp[i] * (d * q - 1.f) + d_ent[i];
  }
}

// Box-Muller on Philox4x32-10 counters (defined in ppo_loss.hip's translation unit as well; duplicated here as static)
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t lo0 = 0xD2511F53u * c.x, hi0 = mulhi32(0xD2511F53u, c.x);
    const uint32_t lo1 = 0xCD9E8D57u * c.z, hi1 = mulhi32(0xCD9E8D57u, c.z);
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* mean, int ld_mean, const float* log_std,
                                                              int ld_ls, const uint8_t* is_eval, long n, int A,
                                                              uint64_t seed, uint64_t offset, float* action,
                                                              float* logp, long row0) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t ctr_row = (uint64_t)(row0 + i);  // the row's number in the caller's whole batch
  const bool greedy = is_eval && is_eval[i];
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float mu = mean[i * ld_mean + j];
    const float sd = expf(log_std[i * ld_ls + j]);
    float x = mu;  // evaluation: the mean (actor_critic_policy.py:504)
    if (!greedy) {
      const uint4 rnd = philox(make_uint4((uint32_t)ctr_row, (uint32_t)(ctr_row >> 32), (uint32_t)j, (uint32_t)offset),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
      const float u1 = ((float)(rnd.x >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
      const float u2 = (float)(rnd.y >> 8) * (1.0f / 16777216.0f);
      x = mu + sd * sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
    const float d = x - mu;
    action[i * A + j] = x;
    lp += -(d * d) / (2.f * sd * sd) - logf(sd) - kHalfLog2Pi;
  }
  logp[i] = lp;
}

}  // namespace

extern "C" int srl_gaussian_fwd(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                                const float* action, long n, int A, float* logp, float* entropy) {
  SRL_CHECK_ARG(n >= 0 && A >= 1 && ld_mean >= A && (ld_log_std == 0 || ld_log_std >= A), "bad extents");
  if (n == 0) return 0;
  SRL_CHECK_ARG(mean && log_std && action && logp && entropy, "null tensor");
  hipLaunchKernelGGL(gaussian_fwd_kernel, dim3((unsigned)srl_ceil_div(n, 256L)), dim3(256), 0, (hipStream_t)stream, mean,
                     ld_mean, log_std, ld_log_std, action, n, A, logp, entropy);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gaussian_bwd(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                                const float* action, long n, int A, const float* d_logp, const float* d_entropy,
                                float* d_mean, float* d_log_std) {
  SRL_CHECK_ARG(n >= 0 && A >= 1 && ld_mean >= A && (ld_log_std == 0 || ld_log_std >= A), "bad extents");
  if (n == 0) return 0;
  SRL_CHECK_ARG(mean && log_std && action && d_logp && d_entropy && d_mean && d_log_std, "null tensor");
  const long tot = n * A;
  const unsigned grid = (unsigned)(srl_ceil_div(tot, 256L) < 4096 ? srl_ceil_div(tot, 256L) : 4096);
  hipLaunchKernelGGL(gaussian_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, mean, ld_mean, log_std,
                     ld_log_std, action, n, A, d_logp, d_entropy, d_mean, d_log_std);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gaussian_sample(void* stream, const float* mean, int ld_mean, const float* log_std, int ld_log_std,
                                   const uint8_t* is_eval, long n, int A, uint64_t seed, uint64_t offset, float* action,
                                   float* logp, int64_t row0) {
  SRL_CHECK_ARG(n >= 0 && A >= 1 && ld_mean >= A && (ld_log_std == 0 || ld_log_std >= A), "bad extents");
  if (n == 0) return 0;
  SRL_CHECK_ARG(mean && log_std && action && logp, "null tensor");
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3((unsigned)srl_ceil_div(n, 256L)), dim3(256), 0, (hipStream_t)stream,
                     mean, ld_mean, log_std, ld_log_std, is_eval, n, A, seed, offset, action, logp, (long)row0);
  SRL_LAUNCH_CHECK();
  return 0;
}
