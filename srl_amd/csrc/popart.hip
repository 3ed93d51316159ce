// PopArt value head support (reference legacy/algorithm/modules/popart.py:8-59 and the RunningMeanStd of
// modules/utils.py:70-151): float64 running statistics [mean(vd), mean_sq(vd), debiasing_term(1)] on the
// device, per-column masked sums of the value targets, the EMA update (+ optional rescale of the head so that
// its de-normalised output is preserved), and the two element-wise maps.  All tensors are tiny ([T, B, vd]).
#include "srl_common.h"

namespace {

__device__ __forceinline__ void mean_std(const double* rms, int vd, int c, double eps, double& mean, double& std) {
  const double deb = fmax(rms[2 * vd], eps);                         // utils.py:139-140
  mean = rms[c] / deb;
  const double var = fmax(rms[vd + c] / deb - mean * mean, 1e-2);    // utils.py:141
  std = sqrt(var);
}

__global__ __launch_bounds__(256) void stats_cols_kernel(const float* x, const uint8_t* mask, int invert, long n, int vd,
                                                         int c, double* stats) {
  __shared__ double red[12];
  double acc[3] = {0.0, 0.0, 0.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double m = 1.0;
    if (mask) m = invert ? 1.0 - (double)mask[i] : (double)mask[i];
    const double v = (double)x[i * vd + c] * m;  // utils.py:113-118
    acc[0] += m;
    acc[1] += v;
    acc[2] += v * v;
  }
  block_sum<3, 256>(acc, red);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[3 * c + 0], acc[0]);
    atomicAdd(&stats[3 * c + 1], acc[1]);
    atomicAdd(&stats[3 * c + 2], acc[2]);
  }
}

// one workgroup; thread c < vd updates column c, then (rescale) all threads walk the head's weight rows
__global__ __launch_bounds__(256) void popart_update_kernel(const double* stats, double beta, double eps, int vd,
                                                            double* rms, float* w, float* b, int in_features,
                                                            int rescale) {
  __shared__ double s_old_mean[64], s_old_std[64], s_new_mean[64], s_new_std[64];
  const int c = threadIdx.x;
  if (c < vd) mean_std(rms, vd, c, eps, s_old_mean[c], s_old_std[c]);
  __syncthreads();
  if (c < vd) {
    const double factor = stats[3 * c];
    const double bm = stats[3 * c + 1] / factor, bsq = stats[3 * c + 2] / factor;  // utils.py:125-126
    rms[c] = beta * rms[c] + bm * (1.0 - beta);                                     // :128
    rms[vd + c] = beta * rms[vd + c] + bsq * (1.0 - beta);                          // :129
  }
  __syncthreads();
  if (c == 0) rms[2 * vd] = beta * rms[2 * vd] + 1.0 - beta;                        // :130
  __threadfence_block();
  __syncthreads();
  if (!rescale) return;
  if (c < vd) mean_std(rms, vd, c, eps, s_new_mean[c], s_new_std[c]);
  __syncthreads();
  for (int e = threadIdx.x; e < vd * in_features; e += 256) {  // popart.py:50
    const int r = e / in_features;
    w[e] = (float)((double)w[e] * (s_old_std[r] / s_new_std[r]));
  }
  if (c < vd) b[c] = (float)((s_old_std[c] * (double)b[c] + s_old_mean[c] - s_new_mean[c]) / s_new_std[c]);  // :51
}

template <bool NORMALIZE>
__global__ __launch_bounds__(256) void popart_map_kernel(const float* x, long n, int vd, const double* rms, double eps,
                                                         float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n * vd; i += (long)gridDim.x * 256) {
    double mean, std;
    mean_std(rms, vd, (int)(i % vd), eps, mean, std);
    const double v = (double)x[i];
    if (NORMALIZE) out[i] = (float)fmin(fmax((v - mean) / std, -5.0), 5.0);  // utils.py:148
    else out[i] = (float)(v * std + mean);                                     // :155
  }
}

// per-channel sums -> the three sums of the advantage normalisation over every channel (the mask counted once)
__global__ void fold_col_stats_kernel(const double* cs, int vd, double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0, q = 0.0;
    for (int c = 0; c < vd; ++c) { s += cs[3 * c + 1]; q += cs[3 * c + 2]; }
    out[0] = cs[0]; out[1] = s; out[2] = q;
  }
}

__global__ __launch_bounds__(256) void importance_ratio_kernel(const float* new_lp, const float* old_lp, long n, float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = expf(new_lp[i] - old_lp[i]);
}

}  // namespace

extern "C" int srl_fold_col_stats(void* stream, const double* col_stats, int vd, double* stats) {
  SRL_CHECK_ARG(col_stats && stats && vd >= 1 && vd <= 64, "bad arguments (1 <= value_dim <= 64)");
  hipLaunchKernelGGL(fold_col_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, col_stats, vd, stats);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_importance_ratio(void* stream, const float* new_lp, const float* old_lp, long n, float* ratio) {
  SRL_CHECK_ARG(n >= 0, "negative count");
  if (n == 0) return 0;
  SRL_CHECK_ARG(new_lp && old_lp && ratio, "null tensor");
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256L) < 2048 ? srl_ceil_div(n, 256L) : 2048);
  hipLaunchKernelGGL(importance_ratio_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, new_lp, old_lp, n, ratio);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_masked_stats_cols(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n, int vd,
                                     double* stats) {
  SRL_CHECK_ARG(stats && n >= 0 && vd >= 1 && vd <= 64, "bad arguments (1 <= value_dim <= 64)");
  hipStream_t st = (hipStream_t)stream;
  SRL_HIP_TRY(hipMemsetAsync(stats, 0, 3 * vd * sizeof(double), st));
  if (n == 0) return 0;
  SRL_CHECK_ARG(x != nullptr, "null tensor");
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256L) < 512 ? srl_ceil_div(n, 256L) : 512);
  for (int c = 0; c < vd; ++c)
    hipLaunchKernelGGL(stats_cols_kernel, dim3(grid), dim3(256), 0, st, x, mask, mask_invert, n, vd, c, stats);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_popart_update(void* stream, const double* stats, double beta, double eps, int vd, double* rms,
                                 float* w, float* b, int in_features, int rescale) {
  SRL_CHECK_ARG(stats && rms && vd >= 1 && vd <= 64, "bad arguments (1 <= value_dim <= 64)");
  SRL_CHECK_ARG(!rescale || (w && b && in_features > 0), "rescale needs the head's weight and bias");
  hipLaunchKernelGGL(popart_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, beta, eps, vd, rms, w, b,
                     in_features, rescale);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_popart_map(void* stream, const float* x, long n, int vd, const double* rms, double eps, int normalize,
                              float* out) {
  SRL_CHECK_ARG(rms && n >= 0 && vd >= 1 && vd <= 64, "bad arguments (1 <= value_dim <= 64)");
  if (n == 0) return 0;
  SRL_CHECK_ARG(x && out, "null tensor");
  const unsigned grid = (unsigned)(srl_ceil_div(n * vd, 256L) < 2048 ? srl_ceil_div(n * vd, 256L) : 2048);
  if (normalize)
    hipLaunchKernelGGL(popart_map_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, vd, rms, eps, out);
  else
    hipLaunchKernelGGL(popart_map_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, vd, rms, eps, out);
  SRL_LAUNCH_CHECK();
  return 0;
}
