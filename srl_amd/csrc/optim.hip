// Optimiser on the flat parameter buffer: global gradient norm + fused clip / Adam(W) step.
// 28 algorithmic bytes per parameter (read p, g, m, v; write p, m, v): pure HBM streaming.
#include "srl_common.h"

namespace {

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* g, long n, double* out) {
  __shared__ double red[4];
  double acc[1] = {0.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double v = g[i];
    acc[0] += v * v;
  }
  block_sum<1, 256>(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, acc[0]);
}

// clip coefficient shared by the optimiser kernels: grad_scale (the DDP mean) times clip_grad_norm_'s
// clamp(max_norm / (total + 1e-6), max = 1); also reports the (scaled) total norm
__device__ __forceinline__ float clip_coef(const double* sumsq, float grad_scale, float max_norm, float* grad_norm_out) {
  float coef = grad_scale;
  if (sumsq) {
    const float norm = (float)sqrt(sumsq[0]) * grad_scale;
    if (max_norm >= 0.f) {
      const float c = max_norm / (norm + 1e-6f);
      coef *= c < 1.f ? c : 1.f;
    }
    if (grad_norm_out && blockIdx.x == 0 && threadIdx.x == 0) grad_norm_out[0] = norm;
  }
  return coef;
}

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
  float lr, b1, b2, eps, wd;
  int adamw;
  float step_size, bc2_sqrt, grad_scale, max_norm;
  const double* sumsq;
  float* grad_norm_out;
  const float* step_scalars;  // optional device {step_size, bc2_sqrt}: overrides the two fields above
};

// torch.optim.Adam single-tensor path: m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p.addcdiv_(m, denom, -lr/(1-b1^t)).
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  const float step_size = a.step_scalars ? a.step_scalars[0] : a.step_size;
  const float bc2_sqrt = a.step_scalars ? a.step_scalars[1] : a.bc2_sqrt;
  const float coef = clip_coef(a.sumsq, a.grad_scale, a.max_norm, a.grad_norm_out);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    float p = a.p[i];
    float g = a.g[i] * coef;
    if (a.wd != 0.f) {
      if (a.adamw) p *= 1.f - a.lr * a.wd;
      else g += a.wd * p;
    }
    float m = a.m[i];
    m = m + (1.f - a.b1) * (g - m);
    const float v = a.v[i] * a.b2 + (1.f - a.b2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p -= step_size * (m / denom);
    a.p[i] = p;
    a.m[i] = m;
    a.v[i] = v;
  }
}

struct SgdArgs {
  float* p;
  const float* g;
  float* buf;
  long n;
  float lr, momentum, dampening, wd;
  int nesterov, first;
  float grad_scale, max_norm;
  const double* sumsq;
  float* grad_norm_out;
};

// torch.optim.SGD single-tensor path (torch/optim/sgd.py): g += wd p; with momentum: buf = g on the first step, else
// buf = momentum buf + (1 - dampening) g; g = g + momentum buf (nesterov) or buf; p -= lr g
__global__ __launch_bounds__(256) void sgd_kernel(SgdArgs a) {
  const float coef = clip_coef(a.sumsq, a.grad_scale, a.max_norm, a.grad_norm_out);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    float p = a.p[i];
    float g = a.g[i] * coef;
    if (a.wd != 0.f) g += a.wd * p;
    if (a.momentum != 0.f) {
      const float b = a.first ? g : a.momentum * a.buf[i] + (1.f - a.dampening) * g;
      a.buf[i] = b;
      g = a.nesterov ? g + a.momentum * b : b;
    }
    a.p[i] = p - a.lr * g;
  }
}

struct RmsArgs {
  float* p;
  const float* g;
  float* sq;
  float* buf;
  float* gavg;
  long n;
  float lr, alpha, eps, wd, momentum;
  int centered;
  float grad_scale, max_norm;
  const double* sumsq;
  float* grad_norm_out;
};

// torch.optim.RMSprop single-tensor path (torch/optim/rmsprop.py): g += wd p; sq = alpha sq + (1 - alpha) g^2;
// centered: gavg = lerp(gavg, g, 1 - alpha), avg = sqrt(sq - gavg^2) + eps, else avg = sqrt(sq) + eps;
// momentum > 0: buf = momentum buf + g / avg, p -= lr buf; else p -= lr g / avg
__global__ __launch_bounds__(256) void rmsprop_kernel(RmsArgs a) {
  const float coef = clip_coef(a.sumsq, a.grad_scale, a.max_norm, a.grad_norm_out);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    float p = a.p[i];
    float g = a.g[i] * coef;
    if (a.wd != 0.f) g += a.wd * p;
    const float sq = a.alpha * a.sq[i] + (1.f - a.alpha) * g * g;
    a.sq[i] = sq;
    float avg;
    if (a.centered) {
      float ga = a.gavg[i];
      ga = ga + (1.f - a.alpha) * (g - ga);
      a.gavg[i] = ga;
      avg = sqrtf(sq - ga * ga) + a.eps;
    } else {
      avg = sqrtf(sq) + a.eps;
    }
    if (a.momentum > 0.f) {
      const float b = a.momentum * a.buf[i] + g / avg;
      a.buf[i] = b;
      p -= a.lr * b;
    } else {
      p -= a.lr * (g / avg);
    }
    a.p[i] = p;
  }
}

}  // namespace

extern "C" int srl_sgd_step(void* stream, float* p, const float* g, float* momentum_buf, int64_t n, float lr,
                            float momentum, float dampening, float weight_decay, int nesterov, int first_step,
                            float grad_scale, float max_norm, const double* sumsq, float* grad_norm_out) {
  SRL_CHECK_ARG(p && g && n >= 0 && (momentum == 0.f || momentum_buf), "null tensor");
  SRL_CHECK_ARG(max_norm < 0.f || sumsq, "clipping needs sumsq");
  SRL_CHECK_ARG(!nesterov || (momentum > 0.f && dampening == 0.f), "Nesterov momentum requires a momentum and zero dampening");
  if (n == 0) return 0;
  SgdArgs a{p, g, momentum_buf, (long)n, lr, momentum, dampening, weight_decay, nesterov, first_step, grad_scale, max_norm,
            sumsq, grad_norm_out};
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_rmsprop_step(void* stream, float* p, const float* g, float* square_avg, float* momentum_buf,
                                float* grad_avg, int64_t n, float lr, float alpha, float eps, float weight_decay,
                                float momentum, int centered, float grad_scale, float max_norm, const double* sumsq,
                                float* grad_norm_out) {
  SRL_CHECK_ARG(p && g && square_avg && n >= 0, "null tensor");
  SRL_CHECK_ARG((momentum <= 0.f || momentum_buf) && (!centered || grad_avg), "missing state buffer");
  SRL_CHECK_ARG(max_norm < 0.f || sumsq, "clipping needs sumsq");
  if (n == 0) return 0;
  RmsArgs a{p, g, square_avg, momentum_buf, grad_avg, (long)n, lr, alpha, eps, weight_decay, momentum, centered, grad_scale,
            max_norm, sumsq, grad_norm_out};
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ __launch_bounds__(256) void accumulate_kernel(float* dst, const float* src, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] += src[i];
}
struct AccSrcs {
  const float* p[SRL_ACCUMULATE_MAX];
  int k;
};
// dst += src_0 + src_1 + ... in that order (left to right: the same sums as k launches of accumulate_kernel)
__global__ __launch_bounds__(256) void accumulate_n_kernel(float* dst, AccSrcs s, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = dst[i];
    for (int j = 0; j < s.k; ++j) v += s.p[j][i];
    dst[i] = v;
  }
}
}  // namespace

extern "C" int srl_accumulate(void* stream, float* dst, const float* src, int64_t n) {
  SRL_CHECK_ARG(n >= 0, "negative count");
  if (n == 0) return 0;
  SRL_CHECK_ARG(dst && src, "null tensor");
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(accumulate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, src, (long)n);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_accumulate_n(void* stream, float* dst, const float* const* srcs, int32_t k, int64_t n) {
  SRL_CHECK_ARG(n >= 0 && k >= 0 && k <= SRL_ACCUMULATE_MAX, "negative count / more sources than SRL_ACCUMULATE_MAX");
  if (n == 0 || k == 0) return 0;
  SRL_CHECK_ARG(dst && srcs, "null tensor");
  AccSrcs s{};
  s.k = k;
  for (int j = 0; j < k; ++j) {
    SRL_CHECK_ARG(srcs[j], "null source");
    s.p[j] = srcs[j];
  }
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(accumulate_n_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, s, (long)n);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_grad_sumsq(void* stream, const float* g, int64_t n, double* sumsq) {
  SRL_CHECK_ARG(g && sumsq && n >= 0, "null tensor");
  hipStream_t st = (hipStream_t)stream;
  SRL_HIP_TRY(hipMemsetAsync(sumsq, 0, sizeof(double), st));
  if (n == 0) return 0;
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 1024 ? srl_ceil_div(n, 256) : 1024);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(grid), dim3(256), 0, st, g, (long)n, sumsq);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int adamw, int64_t step,
                             float grad_scale, float max_norm, const double* sumsq, float* grad_norm_out,
                             const float* step_scalars) {
  SRL_CHECK_ARG(p && g && m && v && n >= 0 && (step >= 1 || step_scalars), "null tensor or step < 1");
  SRL_CHECK_ARG(max_norm < 0.f || sumsq, "clipping needs sumsq");
  if (n == 0) return 0;
  AdamArgs a{p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, adamw, 0.f, 0.f, grad_scale, max_norm, sumsq,
             grad_norm_out, step_scalars};
  if (!step_scalars) {
    // bias corrections in double like python floats in torch/optim/adam.py
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
  }
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SRL_LAUNCH_CHECK();
  return 0;
}
