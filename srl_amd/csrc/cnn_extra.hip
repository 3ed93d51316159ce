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

// ---- the same pieces for any number of spatial dimensions: channels-last volumes [n, D, H, W, C] ---------------------
// Conv1d (D = H = 1) and Conv3d encoders (modules/cnn.py:60-71 picks nn.Conv1d / Conv2d / Conv3d and the matching MaxPool
// by the observation's rank).  Explicit patch matrix + the dense GEMM: built for parity (modules_test.py:385-401 is the
// only user in the reference), not for speed.
struct Vol {
  int D, H, W, C;
};

// border of (pd, ph, pw) voxels on both sides of each axis, filled per nn.ConvNd's padding_mode: 0 zeros, 1 reflect
// (without the edge), 2 replicate, 3 circular.  Source index of padded coordinate o (already shifted by -pad) along an
// axis of extent D, or -1 for a zero.
__device__ __forceinline__ int pad_src(int o, int D, int mode) {
  if (o >= 0 && o < D) return o;
  if (mode == 1) return o < 0 ? -o : 2 * (D - 1) - o;
  if (mode == 2) return o < 0 ? 0 : D - 1;
  if (mode == 3) return o < 0 ? o + D : o - D;
  return -1;
}
__global__ __launch_bounds__(256) void pad_nd_kernel(const float* x, long n, Vol v, int pd, int ph, int pw, int mode, float* y) {
  const int Dp = v.D + 2 * pd, Hp = v.H + 2 * ph, Wp = v.W + 2 * pw;
  const long total = n * Dp * Hp * Wp * v.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % v.C);
    long t = e / v.C;
    const int w = pad_src((int)(t % Wp) - pw, v.W, mode); t /= Wp;
    const int h = pad_src((int)(t % Hp) - ph, v.H, mode); t /= Hp;
    const int d = pad_src((int)(t % Dp) - pd, v.D, mode);
    const long s = t / Dp;
    y[e] = (d >= 0 && h >= 0 && w >= 0) ? x[(((s * v.D + d) * v.H + h) * v.W + w) * v.C + c] : 0.f;
  }
}
// The padded coordinates (0 .. D + 2p - 1) that read interior index i along one axis: the copy itself first, then the
// border voxels filled from it.  At most 1 + 2p of them (replicate at an edge); returns the count.
__device__ __forceinline__ int pad_preimage(int i, int D, int p, int mode, int* out) {
  int k = 0;
  out[k++] = i + p;
  if (p == 0 || mode == 0) return k;
  if (mode == 1) {
    if (i >= 1 && i <= p) out[k++] = p - i;
    if (i <= D - 2 && i >= D - 1 - p) out[k++] = p + 2 * (D - 1) - i;
  } else if (mode == 2) {
    if (i == 0) for (int o = 0; o < p; ++o) out[k++] = o;
    if (i == D - 1) for (int o = 0; o < p; ++o) out[k++] = p + D + o;
  } else {
    if (i >= D - p) out[k++] = i + p - D;
    if (i < p) out[k++] = i + p + D;
  }
  return k;
}
constexpr int kMaxPad = 8;  // per side (host-checked): bounds the preimage lists
// adjoint of pad_nd_kernel: x[i] = sum of the padded voxels that were filled from i (mode 0: the crop)
__global__ __launch_bounds__(256) void crop_nd_kernel(const float* yp, long n, Vol v, int pd, int ph, int pw, int mode, float* x) {
  const int Hp = v.H + 2 * ph, Wp = v.W + 2 * pw, Dp = v.D + 2 * pd;
  const long total = n * v.D * v.H * v.W * v.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % v.C);
    long t = e / v.C;
    const int w = (int)(t % v.W); t /= v.W;
    const int h = (int)(t % v.H); t /= v.H;
    const int d = (int)(t % v.D);
    const long s = t / v.D;
    int od[1 + 2 * kMaxPad], oh[1 + 2 * kMaxPad], ow[1 + 2 * kMaxPad];
    const int nd = pad_preimage(d, v.D, pd, mode, od), nh = pad_preimage(h, v.H, ph, mode, oh), nw = pad_preimage(w, v.W, pw, mode, ow);
    float acc = 0.f;
    for (int a = 0; a < nd; ++a)
      for (int b = 0; b < nh; ++b)
        for (int g = 0; g < nw; ++g) acc += yp[(((s * Dp + od[a]) * Hp + oh[b]) * Wp + ow[g]) * v.C + c];
    x[e] = acc;
  }
}

// max over windows of (wd, wh, ww) voxels (each 1 or 2), stride = window, floor
__global__ __launch_bounds__(256) void maxpool_nd_fwd_kernel(const float* x, long n, Vol v, int wd, int wh, int ww, float* y) {
  const int OD = v.D / wd, OH = v.H / wh, OW = v.W / ww;
  const long total = n * OD * OH * OW * v.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % v.C);
    long t = e / v.C;
    const int b = (int)(t % OW); t /= OW;
    const int a = (int)(t % OH); t /= OH;
    const int z = (int)(t % OD);
    const long s = t / OD;
    float m = -INFINITY;
    for (int i = 0; i < wd; ++i)
      for (int j = 0; j < wh; ++j)
        for (int k = 0; k < ww; ++k)
          m = fmaxf(m, x[(((s * v.D + z * wd + i) * v.H + a * wh + j) * v.W + b * ww + k) * v.C + c]);
    y[e] = m;
  }
}
// gradient to the first maximum of each window in (d, h, w) scan order, times act'(x); voxels no window covers get 0
__global__ __launch_bounds__(256) void maxpool_nd_bwd_kernel(const float* dy, const float* x, long n, Vol v, int wd, int wh, int ww,
                                                             int dact, float* dx) {
  const int OD = v.D / wd, OH = v.H / wh, OW = v.W / ww;
  const long total = n * v.D * v.H * v.W * v.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % v.C);
    long t = e / v.C;
    const int w = (int)(t % v.W); t /= v.W;
    const int h = (int)(t % v.H); t /= v.H;
    const int d = (int)(t % v.D);
    const long s = t / v.D;
    const int z = d / wd, a = h / wh, b = w / ww;
    float g = 0.f;
    if (z < OD && a < OH && b < OW) {
      float m = -INFINITY;
      int arg = -1, idx = 0;
      for (int i = 0; i < wd; ++i)
        for (int j = 0; j < wh; ++j)
          for (int k = 0; k < ww; ++k, ++idx) {
            const float val = x[(((s * v.D + z * wd + i) * v.H + a * wh + j) * v.W + b * ww + k) * v.C + c];
            if (arg < 0 || val > m) { m = val; arg = idx; }
          }
      const int mine = ((d - z * wd) * wh + (h - a * wh)) * ww + (w - b * ww);
      if (arg == mine) g = dy[(((s * OD + z) * OH + a) * OW + b) * v.C + c];
    }
    dx[e] = g * act_grad_from_output(x[e], dact);
  }
}

// P[(s, od, oh, ow)][(kd, kh, kw, c)] = x[s, od S + kd, oh S + kh, ow S + kw, c]
__global__ __launch_bounds__(256) void im2col_nd_kernel(const float* x, long n, Vol v, int KD, int KH, int KW, int S, int OD, int OH,
                                                        int OW, float* P) {
  const long Kp = (long)KD * KH * KW * v.C, total = n * OD * OH * OW * Kp;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    long k = e % Kp, m = e / Kp;
    const int c = (int)(k % v.C); k /= v.C;
    const int kw = (int)(k % KW); k /= KW;
    const int kh = (int)(k % KH);
    const int kd = (int)(k / KH);
    const int ow = (int)(m % OW); m /= OW;
    const int oh = (int)(m % OH); m /= OH;
    const int od = (int)(m % OD);
    const long s = m / OD;
    P[e] = x[(((s * v.D + od * S + kd) * v.H + oh * S + kh) * v.W + ow * S + kw) * v.C + c];
  }
}
// dX[s, d, h, w, c] = sum over the taps that reach the voxel of dP[(s, od, oh, ow)][(kd, kh, kw, c)]   (* act'(y))
__global__ __launch_bounds__(256) void col2im_nd_kernel(const float* dP, long n, Vol v, int KD, int KH, int KW, int S, int OD,
                                                        int OH, int OW, const float* y, int dact, float* dX) {
  const long Kp = (long)KD * KH * KW * v.C, total = n * v.D * v.H * v.W * v.C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int c = (int)(e % v.C);
    long t = e / v.C;
    const int w = (int)(t % v.W); t /= v.W;
    const int h = (int)(t % v.H); t /= v.H;
    const int d = (int)(t % v.D);
    const long s = t / v.D;
    float acc = 0.f;
    for (int kd = d % S; kd < KD && kd <= d; kd += S) {
      const int od = (d - kd) / S;
      if (od >= OD) continue;
      for (int kh = h % S; kh < KH && kh <= h; kh += S) {
        const int oh = (h - kh) / S;
        if (oh >= OH) continue;
        for (int kw = w % S; kw < KW && kw <= w; kw += S) {
          const int ow = (w - kw) / S;
          if (ow >= OW) continue;
          acc += dP[(((s * OD + od) * OH + oh) * OW + ow) * Kp + ((long)(kd * KH + kh) * KW + kw) * v.C + c];
        }
      }
    }
    if (y && dact) acc *= act_grad_from_output(y[e], dact);
    dX[e] = acc;
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

static bool vol_ok(int D, int H, int W, int C) { return D >= 1 && H >= 1 && W >= 1 && C >= 1; }

// nn.ConvNd's own preconditions: reflect needs pad < extent, circular pad <= extent (torch raises otherwise)
static bool pad_ok(int D, int H, int W, int pd, int ph, int pw, int mode) {
  if (pd < 0 || ph < 0 || pw < 0 || mode < 0 || mode > 3 || pd > kMaxPad || ph > kMaxPad || pw > kMaxPad) return false;
  if (mode == 1) return pd < D && ph < H && pw < W;
  if (mode == 3) return pd <= D && ph <= H && pw <= W;
  return true;
}

extern "C" int srl_pad_ndhwc(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int pd, int ph, int pw, int mode,
                             float* y) {
  SRL_CHECK_ARG(x && y && vol_ok(D, H, W, C) && pad_ok(D, H, W, pd, ph, pw, mode), "null tensor / bad shape / padding too wide");
  if (n == 0) return 0;
  const long total = n * (D + 2 * pd) * (H + 2 * ph) * (W + 2 * pw) * C;
  hipLaunchKernelGGL(pad_nd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long)n, Vol{D, H, W, C}, pd, ph, pw, mode, y);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_crop_ndhwc(void* stream, const float* yp, int64_t n, int D, int H, int W, int C, int pd, int ph, int pw, int mode,
                              float* x) {
  SRL_CHECK_ARG(x && yp && vol_ok(D, H, W, C) && pad_ok(D, H, W, pd, ph, pw, mode), "null tensor / bad shape / padding too wide");
  if (n == 0) return 0;
  hipLaunchKernelGGL(crop_nd_kernel, dim3(grid_for(n * D * H * W * C)), dim3(256), 0, (hipStream_t)stream, yp, (long)n, Vol{D, H, W, C}, pd, ph, pw, mode, x);
  SRL_LAUNCH_CHECK();
  return 0;
}

static bool win_ok(int D, int H, int W, int wd, int wh, int ww) {
  return (wd == 1 || wd == 2) && (wh == 1 || wh == 2) && (ww == 1 || ww == 2) && D >= wd && H >= wh && W >= ww;
}

extern "C" int srl_maxpool_ndhwc_fwd(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int wd, int wh, int ww,
                                     float* y) {
  SRL_CHECK_ARG(x && y && vol_ok(D, H, W, C) && win_ok(D, H, W, wd, wh, ww), "null tensor / volume smaller than the window");
  if (n == 0) return 0;
  const long total = n * (D / wd) * (H / wh) * (W / ww) * C;
  hipLaunchKernelGGL(maxpool_nd_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long)n, Vol{D, H, W, C}, wd, wh, ww, y);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_maxpool_ndhwc_bwd(void* stream, const float* dy, const float* x, int64_t n, int D, int H, int W, int C, int wd,
                                     int wh, int ww, int dact, float* dx) {
  SRL_CHECK_ARG(dy && x && dx && vol_ok(D, H, W, C) && win_ok(D, H, W, wd, wh, ww) && dact >= 0 && dact <= 2, "null tensor / bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(maxpool_nd_bwd_kernel, dim3(grid_for(n * D * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, x, (long)n,
                     Vol{D, H, W, C}, wd, wh, ww, dact, dx);
  SRL_LAUNCH_CHECK();
  return 0;
}

static int out_dim(int d, int k, int s) { return (d - k) / s + 1; }

extern "C" int srl_im2col_ndhwc(void* stream, const float* x, int64_t n, int D, int H, int W, int C, int KD, int KH, int KW, int stride,
                                float* P) {
  SRL_CHECK_ARG(x && P && vol_ok(D, H, W, C) && KD >= 1 && KH >= 1 && KW >= 1 && KD <= D && KH <= H && KW <= W && stride >= 1,
                "null tensor / invalid geometry");
  if (n == 0) return 0;
  const int OD = out_dim(D, KD, stride), OH = out_dim(H, KH, stride), OW = out_dim(W, KW, stride);
  const long total = n * OD * OH * OW * (long)KD * KH * KW * C;
  hipLaunchKernelGGL(im2col_nd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (long)n, Vol{D, H, W, C}, KD, KH, KW,
                     stride, OD, OH, OW, P);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_col2im_ndhwc(void* stream, const float* dP, int64_t n, int D, int H, int W, int C, int KD, int KH, int KW, int stride,
                                const float* y, int dact, float* dX) {
  SRL_CHECK_ARG(dP && dX && vol_ok(D, H, W, C) && KD >= 1 && KH >= 1 && KW >= 1 && KD <= D && KH <= H && KW <= W && stride >= 1 &&
                    dact >= 0 && dact <= 2, "null tensor / invalid geometry");
  if (n == 0) return 0;
  const int OD = out_dim(D, KD, stride), OH = out_dim(H, KH, stride), OW = out_dim(W, KW, stride);
  hipLaunchKernelGGL(col2im_nd_kernel, dim3(grid_for(n * D * H * W * C)), dim3(256), 0, (hipStream_t)stream, dP, (long)n, Vol{D, H, W, C},
                     KD, KH, KW, stride, OD, OH, OW, y, dact, dX);
  SRL_LAUNCH_CHECK();
  return 0;
}
