// Calibration probe: sustained rate of v_mfma_f32_32x32x2_f32 (a) with register-resident operands and (b) inside
// the skeleton of the GEMM k-step (barriers, LDS reads, LDS writes, global loads added one at a time), at 1..4
// workgroups per CU.  Gives the practical ceiling the GEMM core is measured against and prices each ingredient.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LEVEL 0: MFMA only; 1: + two barriers per 64 MFMAs; 2: + operand reads from LDS; 3: + LDS tile writes;
// 4: + 8 x 16-byte global loads per lane per k-step; 5: the global loads spread over the MFMA loop (one per two
// kk steps) instead of issued in a burst; 6: the LDS writes spread too (into a second buffer); 7: the tile written
// as transposed scalars; 8: the same 8 x 16 bytes per lane moved global -> LDS directly (buffer_load ... lds, no
// VGPR round trip and no ds_write), spread like level 6; 9: the same in one burst after the barrier
template <int LEVEL>
__global__ __launch_bounds__(256, 4) void probe(float* out, const float* src, int iters) {
  __shared__ float lds[4 * 32 * 132];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5, wid = tid >> 6;
  float a0 = tid * 1e-3f, a1 = a0 + 1.f, b0 = blockIdx.x * 1e-3f, b1 = b0 + 1.f;
  float4 r[8];
  for (int q = 0; q < 8; ++q) r[q] = make_float4(a0, a1, b0, b1);
  const float4* gp = reinterpret_cast<const float4*>(src) + (size_t)blockIdx.x * 2048 + tid;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float4*>(reinterpret_cast<const float4*>(src) + (size_t)blockIdx.x * 2048), 0, 2048 * 16, 0x00020000);
  if (LEVEL >= 2) {
    for (int e = tid; e < 4 * 32 * 132; e += 256) lds[e] = e * 1e-4f;
    __syncthreads();
  }
  const float* ap = lds + h * 132 + (wid >> 1) * 64 + l31;
  const float* bp = lds + 32 * 132 + h * 132 + (wid & 1) * 64 + l31;
  for (int it = 0; it < iters; ++it) {
    if (LEVEL >= 1) __syncthreads();
    if (LEVEL >= 3 && LEVEL < 6) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = tid + (q & 3) * 256;
        *reinterpret_cast<float4*>(lds + (q >> 2) * 32 * 132 + (u / 32) * 132 + (u % 32) * 4) = r[q];
      }
    }
    if (LEVEL == 7) {  // the same tile written as transposed scalars (k-contiguous operand): 32 ds_write_b32
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = tid + (q & 3) * 256, x = u / 8, k = (u % 8) * 4;
        float* d = lds + (q >> 2) * 32 * 132 + k * 132 + x;
        d[0] = r[q].x, d[132] = r[q].y, d[264] = r[q].z, d[396] = r[q].w;
      }
    }
    if (LEVEL >= 1) __syncthreads();
    if (LEVEL == 9) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 2 * 32 * 132 + (q * 4 + wid) * 256), 16,
                                             (int)(((size_t)((it * 8 + q) & 7) * 256 + tid) * 16), 0, 0, 0);
    }
    if (LEVEL == 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = gp[(size_t)((it * 8 + q) & 7) * 256];
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (LEVEL >= 2) {
        a0 = ap[kk * 2 * 132], a1 = ap[kk * 2 * 132 + 32];
        b0 = bp[kk * 2 * 132], b1 = bp[kk * 2 * 132 + 32];
      }
      if (LEVEL == 8 && (kk & 1) == 0) {
        const int q = kk >> 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 2 * 32 * 132 + (q * 4 + wid) * 256), 16,
                                             (int)(((size_t)((it * 8 + q) & 7) * 256 + tid) * 16), 0, 0, 0);
      }
      if (LEVEL >= 5 && LEVEL < 8 && (kk & 1) == 0) r[kk >> 1] = gp[(size_t)((it * 8 + (kk >> 1)) & 7) * 256];
      if (LEVEL >= 6 && (kk & 1) == 1) {
        const int q = kk >> 1, u = tid + (q & 3) * 256;
        *reinterpret_cast<float4*>(lds + (2 + (q >> 2)) * 32 * 132 + (u / 32) * 132 + (u % 32) * 4) = r[(q + 4) & 7];
      }
      if (LEVEL >= 5 && LEVEL < 9) __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = r[0].x + r[7].w;
  for (int i = 0; i < 4; ++i)
    for (int r2 = 0; r2 < 16; ++r2) s += acc[i][r2];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LEVEL>
void run(float* out, const float* src, hipEvent_t e0, hipEvent_t e1) {
  const int iters = 2000;
  for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {
    const int grid = 256 * wg_per_cu;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      probe<LEVEL><<<grid, 256>>>(out, src, iters);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = (double)grid * 4 * iters * 16 * 4 * 4096.0;
    printf("level %d  workgroups/CU=%d  %.3f ms  %.1f TFLOP/s\n", LEVEL, wg_per_cu, ms, flop / ms / 1e9);
  }
}

int main() {
  float *out, *src;
  (void)hipMalloc(&out, 256 * 4096 * sizeof(float));
  (void)hipMalloc(&src, (size_t)1024 * 2048 * 16);
  (void)hipMemset(src, 0, (size_t)1024 * 2048 * 16);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  run<0>(out, src, e0, e1);
  run<1>(out, src, e0, e1);
  run<2>(out, src, e0, e1);
  run<3>(out, src, e0, e1);
  run<4>(out, src, e0, e1);
  run<5>(out, src, e0, e1);
  run<6>(out, src, e0, e1);
  run<7>(out, src, e0, e1);
  run<8>(out, src, e0, e1);
  run<9>(out, src, e0, e1);
  return 0;
}
