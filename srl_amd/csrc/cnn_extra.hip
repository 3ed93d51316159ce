// Pieces of the `Convolution` encoder (legacy/algorithm/modules/cnn.py:93-135) beyond unpadded convolutions: zero padding
// (nn.Conv2d(padding=p, padding_mode='zeros')), MaxPool2d(2) in front of a layer (`use_maxpool`, :100-106), and the
// whole-observation LayerNorm (policies/utils.py:53) as an explicit pass producing the channels-last float32 image those
// layers take.  All HBM-bound element-wise passes; the contractions stay on the implicit-GEMM kernels, which see an
// already padded / pooled activation.  No reference experiment sets padding or use_maxpool (only modules_test.py:385-401
// exercises the pooling), so this path is built for parity, not for speed.
#include "srl_common.h"

namespace {

inline unsigned grid_for(long n) {
  const long b = srl_ceil_div(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b));
}

// y[n,h,w,c] = (x[n,c,h,w] - mean_n) * rstd_n * gamma[c,h,w] + beta[c,h,w]
template <bool U8>
__global__ __launch_bounds__(256) void obs_ln_nhwc_kernel(const void* obs, const float* mean, const float* rstd,
                                                          const float* gamma, const float* beta, long n, int C, int H, int W,
                                                          float* y) {
  const long D = (long)C * H * W, total = n * D;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long s = e / D;
    const int r = (int)(e - s * D);  // (h, w, c) order of the output
    const int c = r % C, hw = r / C;
    const int p = c * H * W + hw;    // (c, h, w) order of the observation and of the LayerNorm tables
    const float x = U8 ? (float)static_cast<const uint8_t*>(obs)[s * D + p] : static_cast<const float*>(obs)[s * D + p];
    y[e] = (x - mean[s]) * rstd[s] * gamma[p] + beta[p];
  }
}

// dgamma[p] += sum_n dy[n, p'] * xhat[n, p];  dbeta[p] += sum_n dy[n, p']   (p' = the channels-last index of p)
template <bool U8>
__global__ __launch_bounds__(256) void obs_ln_nhwc_bwd_kernel(const float* dy, const void* obs, const float* mean,
                                                              const float* rstd, long n, int C, int H, int W, float* dgamma,
                                                              float* dbeta) {
  const int D = C * H * W;
  const int r = blockIdx.x * 256 + threadIdx.x;  // channels-last index: neighbouring threads read neighbouring dy
  if (r >= D) return;
  const int c = r % C, hw = r / C;
  const int p = c * H * W + hw;
  float ag = 0.f, ab = 0.f;
  for (long s = 0; s < n; ++s) {
    const float x = U8 ? (float)static_cast<const uint8_t*>(obs)[s * D + p] : static_cast<const float*>(obs)[s * D + p];
    const float g = dy[s * D + r];
    ag += g * ((x - mean[s]) * rstd[s]);
    ab += g;
  }
  dgamma[p] += ag;
  dbeta[p] += ab;
}

// zero padding of an NHWC activation; crop = its adjoint
__global__ __launch_bounds__(256) void pad_nhwc_kernel(const float* x, long n, int H, int W, int C, int pad, float* y) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const long total = n * Hp * Wp * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long t = e / C;
    const int xw = (int)(t % Wp), yh = (int)((t / Wp) % Hp);
    const long s = t / ((long)Wp * Hp);
    const int h = yh - pad, w = xw - pad;
    y[e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((s * H + h) * W + w) * C + c] : 0.f;
  }
}
__global__ __launch_bounds__(256) void crop_nhwc_kernel(const float* yp, long n, int H, int W, int C, int pad, float* x) {
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const long total = n * H * W * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long t = e / C;
    const int w = (int)(t % W), h = (int)((t / W) % H);
    const long s = t / ((long)W * H);
    x[e] = yp[((s * Hp + h + pad) * Wp + w + pad) * C + c];
  }
}

// MaxPool2d(2): y[n,a,b,c] = max over the 2x2 window (floor: a trailing odd row / column is dropped)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* x, long n, int H, int W, int C, float* y) {
  const int OH = H / 2, OW = W / 2;
  const long total = n * OH * OW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long t = e / C;
    const int b = (int)(t % OW), a = (int)((t / OW) % OH);
    const long s = t / ((long)OW * OH);
    const float* p = x + ((s * H + 2 * a) * W + 2 * b) * C + c;
    y[e] = fmaxf(fmaxf(p[0], p[C]), fmaxf(p[(long)W * C], p[(long)W * C + C]));
  }
}
// dx = route(dy) to the FIRST maximum of each window in scan order (torch's rule: a later element wins only if strictly
// greater), times act'(x) of the activation that produced x (dact 1 relu, 2 tanh, 0 none); elements no window covers get 0
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* dy, const float* x, long n, int H, int W, int C, int dact,
                                                           float* dx) {
  const long total = n * H * W * C;
  const int OH = H / 2, OW = W / 2;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % C);
    const long t = e / C;
    const int w = (int)(t % W), h = (int)((t / W) % H);
    const long s = t / ((long)W * H);
    const int a = h / 2, b = w / 2;
    float g = 0.f;
    if (a < OH && b < OW) {
      const float* p = x + ((s * H + 2 * a) * W + 2 * b) * C + c;
      const float v[4] = {p[0], p[C], p[(long)W * C], p[(long)W * C + C]};
      int arg = 0;
      float m = v[0];
#pragma unroll
      for (int i = 1; i < 4; ++i)
        if (v[i] > m) { m = v[i]; arg = i; }
      if (arg == (h & 1) * 2 + (w & 1)) g = dy[((s * OH + a) * OW + b) * C + c];
    }
    dx[e] = g * act_grad_from_output(x[e], dact);
  }
}

}  // namespace

extern "C" int srl_obs_ln_nhwc(void* stream, const void* obs, int is_u8, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, int64_t n, int C, int H, int W, float* y) {
  SRL_CHECK_ARG(obs && mean && rstd && gamma && beta && y && C >= 1 && H >= 1 && W >= 1, "null tensor / bad shape");
  if (n == 0) return 0;
  const long total = n * C * H * W;
  if (is_u8) hipLaunchKernelGGL(obs_ln_nhwc_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, obs, mean, rstd, gamma, beta, (long)n, C, H, W, y);
  else hipLaunchKernelGGL(obs_ln_nhwc_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, obs, mean, rstd, gamma, beta, (long)n, C, H, W, y);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_obs_ln_nhwc_bwd(void* stream, const float* dy, const void* obs, int is_u8, const float* mean,
                                   const float* rstd, int64_t n, int C, int H, int W, float* dgamma, float* dbeta) {
  SRL_CHECK_ARG(dy && obs && mean && rstd && dgamma && dbeta, "null tensor");
  if (n == 0) return 0;
  const int D = C * H * W;
  if (is_u8) hipLaunchKernelGGL(obs_ln_nhwc_bwd_kernel<true>, dim3((unsigned)srl_ceil_div(D, 256)), dim3(256), 0, (hipStream_t)stream, dy, obs, mean, rstd, (long)n, C, H, W, dgamma, dbeta);
  else hipLaunchKernelGGL(obs_ln_nhwc_bwd_kernel<false>, dim3((unsigned)srl_ceil_div(D, 256)), dim3(256), 0, (hipStream_t)stream, dy, obs, mean, rstd, (long)n, C, H, W, dgamma, dbeta);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_pad_nhwc(void* stream, const float* x, int64_t n, int H, int W, int C, int pad, float* y) {
  SRL_CHECK_ARG(x && y && pad >= 0 && H >= 1 && W >= 1 && C >= 1, "null tensor / bad shape");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pad_nhwc_kernel, dim3(grid_for(n * (H + 2 * pad) * (W + 2 * pad) * C)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)n, H, W, C, pad, y);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_crop_nhwc(void* stream, const float* yp, int64_t n, int H, int W, int C, int pad, float* x) {
  SRL_CHECK_ARG(x && yp && pad >= 0 && H >= 1 && W >= 1 && C >= 1, "null tensor / bad shape");
  if (n == 0) return 0;
  hipLaunchKernelGGL(crop_nhwc_kernel, dim3(grid_for(n * H * W * C)), dim3(256), 0, (hipStream_t)stream, yp, (long)n, H, W, C, pad, x);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_maxpool2_nhwc_fwd(void* stream, const float* x, int64_t n, int H, int W, int C, float* y) {
  SRL_CHECK_ARG(x && y && H >= 2 && W >= 2 && C >= 1, "null tensor / image smaller than the window");
  if (n == 0) return 0;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(n * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream, x, (long)n,
                     H, W, C, y);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_maxpool2_nhwc_bwd(void* stream, const float* dy, const float* x, int64_t n, int H, int W, int C, int dact,
                                     float* dx) {
  SRL_CHECK_ARG(dy && x && dx && H >= 2 && W >= 2 && C >= 1 && dact >= 0 && dact <= 2, "null tensor / bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(n * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, x, (long)n, H, W, C,
                     dact, dx);
  SRL_LAUNCH_CHECK();
  return 0;
}
