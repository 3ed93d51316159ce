// Shared helpers for the gfx950 kernels of libsrlhip.so (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <errno.h>

#include "../../include/srl_hip.h"

#define SRL_WAVE 64  // CDNA4 wavefront width (hard-coded; warpSize folds to 64 on gfx950)

void srl_set_error(const char* fmt, ...);

// Which kernel family an entry point chose (host-side launch counters behind srl_dispatch_counts: the tests use them to
// assert that a step really ran on the kernels a benchmark times, include/srl_hip.h)
enum SrlDispatch {
  SRL_DISP_GEMM3 = 0,     // gemm3_kernel: float32 operands as exact bf16 pieces on the bf16 matrix cores
  SRL_DISP_GEMM_F32 = 1,  // gemm_kernel: float32 MFMA
  SRL_DISP_SKINNY = 2,    // skinny_{n,m,k}_kernel
  SRL_DISP_OBS_FWD_BF16 = 3,
  SRL_DISP_OBS_BWD_BF16 = 4,
  SRL_DISP_GEMM2H = 5,    // gemm3_kernel's two-plane f16 variant (three piece products)
  SRL_DISP_H2 = 6,        // h2gemm.h / h2conv.h: pre-split operands staged by LDS-DMA (round 4)
  SRL_DISP_FAMILIES = 7
};
void srl_count_dispatch(int family);
// ... and which instantiation: one counter per (family, three 16-bit template parameters, flags), srl_dispatch_tiles.
// gemm3 / gemm_f32: BM, BN, split-K factor, flags = AMODE << 5 | BMODE << 2 | AKM << 1 | BKM; h2: 1 conv / 2 wgrad / 3 gemm, the
// kind (or NCB), the ring depth; first layer: KP, h2 output?, position split.
void srl_count_tile(int family, int p0, int p1, int p2, int flags);
inline void srl_count_dispatch(int family, int p0, int p1, int p2, int flags = 0) {
  srl_count_dispatch(family);
  srl_count_tile(family, p0, p1, p2, flags);
}

#define SRL_CHECK_ARG(cond, msg)                                   \
  do {                                                             \
    if (!(cond)) {                                                 \
      srl_set_error("%s: invalid argument: %s", __func__, msg);    \
      return -EINVAL;                                              \
    }                                                              \
  } while (0)

#define SRL_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      srl_set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(_e));      \
      return -EIO;                                                                     \
    }                                                                                  \
  } while (0)

#define SRL_LAUNCH_CHECK()                                                             \
  do {                                                                                 \
    hipError_t _e = hipGetLastError();                                                 \
    if (_e != hipSuccess) {                                                            \
      srl_set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(_e));  \
      return -EIO;                                                                     \
    }                                                                                  \
  } while (0)

#ifdef __HIPCC__
#define SRL_HD __host__ __device__
#else
#define SRL_HD
#endif
SRL_HD static inline int64_t srl_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
// ---- wave / block reductions (64-lane wavefronts) -----------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_allmax(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Sum NV doubles per thread across a block of NT threads (NT multiple of 64, <= 1024).
// Result valid in thread 0.  `scratch` must hold (NT/64)*NV doubles of LDS.
template <int NV, int NT>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wid * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double acc = 0.0;
      for (int w = 0; w < NT / 64; ++w) acc += scratch[w * NV + i];
      v[i] = acc;
    }
  }
}

__device__ __forceinline__ float act_apply(float x, int act) {
  return act == 1 ? fmaxf(x, 0.0f) : (act == 2 ? tanhf(x) : x);
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  return act == 1 ? (y > 0.0f ? 1.0f : 0.0f) : (act == 2 ? 1.0f - y * y : 1.0f);
}
#endif
