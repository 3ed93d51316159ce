// Do packed float32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) keep their results when a kernel
// from ANOTHER queue runs matrix instructions on the same compute units at the same time?
//
// Found through the trainer's experimental two-pipeline mode: with two streams running the row chunks
// of one batch side by side, the narrow head products (skinny_n_kernel, whose inner loop the compiler SLP-vectorises into
// v_pk_fma_f32) returned different outputs from run to run for rows that sit in the LOW half of a packed pair, while every
// input was bit-identical; the library built with -fno-slp-vectorize (no packed float32 anywhere) was bit-reproducible.
// This probe isolates it: kernel A runs a chain of packed FMAs whose exact result is known (small integers), kernel B runs
// v_mfma_f32_32x32x16_bf16 back to back; A is checked alone and with B on a second stream.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/pk_hazard_probe scripts/pk_hazard_probe.hip && scripts/pk_hazard_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

// every thread: acc = (lo, hi); `iters` times acc = acc * (1, 1) + (1, 2) with v_pk_fma_f32: exact while below 2^24
__global__ __launch_bounds__(256) void packed_kernel(int iters, int* bad_lo, int* bad_hi) {
  f32x2 acc = {0.f, 0.f};
  const f32x2 one = {1.f, 1.f}, inc = {1.f, 2.f};
  for (int i = 0; i < iters; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(one), "v"(inc));
  if (acc[0] != (float)iters) atomicAdd(bad_lo, 1);
  if (acc[1] != 2.f * (float)iters) atomicAdd(bad_hi, 1);
}

// the same arithmetic with scalar FMAs (control)
__global__ __launch_bounds__(256) void scalar_kernel(int iters, int* bad_lo, int* bad_hi) {
  float lo = 0.f, hi = 0.f;
  for (int i = 0; i < iters; ++i) {
    asm volatile("v_fma_f32 %0, %0, 1.0, 1.0" : "+v"(lo));
    asm volatile("v_fma_f32 %0, %0, 1.0, 2.0" : "+v"(hi));
  }
  if (lo != (float)iters) atomicAdd(bad_lo, 1);
  if (hi != 2.f * (float)iters) atomicAdd(bad_hi, 1);
}

__global__ __launch_bounds__(256) void mfma_kernel(int iters, float* sink) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + (threadIdx.x & 3)), b[j] = (__bf16)(0.5f);
  for (int i = 0; i < iters; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.f) sink[0] = s;
}

int main() {
  int *bad, h[2];
  float* sink;
  CK(hipMalloc(&bad, 8));
  CK(hipMalloc(&sink, 4));
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  const int iters = 200000;  // 2 * iters < 2^24: exact
  for (int mode = 0; mode < 4; ++mode) {  // 0 packed alone, 1 packed beside MFMA, 2 scalar alone, 3 scalar beside MFMA
    int tot_lo = 0, tot_hi = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(bad, 0, 8));
      CK(hipDeviceSynchronize());
      if (mode & 1) hipLaunchKernelGGL(mfma_kernel, dim3(1024), dim3(256), 0, s1, 400000, sink);
      if (mode < 2) hipLaunchKernelGGL(packed_kernel, dim3(2048), dim3(256), 0, s0, iters, bad, bad + 1);
      else hipLaunchKernelGGL(scalar_kernel, dim3(2048), dim3(256), 0, s0, iters, bad, bad + 1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost));
      tot_lo += h[0];
      tot_hi += h[1];
    }
    const char* names[4] = {"packed alone", "packed beside MFMA kernel", "scalar alone", "scalar beside MFMA kernel"};
    printf("%-28s: threads with a wrong LOW half %d, a wrong HIGH half %d (of %d)\n", names[mode], tot_lo, tot_hi, 5 * 2048 * 256);
  }
  return 0;
}
