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
  float coef = a.grad_scale;
  if (a.sumsq) {
    const float norm = (float)sqrt(a.sumsq[0]) * a.grad_scale;
    if (a.max_norm >= 0.f) {
      const float c = a.max_norm / (norm + 1e-6f);  // clip_grad_norm_: clamp(max_norm/(total+1e-6), max=1)
      coef *= c < 1.f ? c : 1.f;
    }
    if (a.grad_norm_out && blockIdx.x == 0 && threadIdx.x == 0) a.grad_norm_out[0] = norm;
  }
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

}  // namespace

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
