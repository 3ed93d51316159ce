// GEMMs with one extent of at most 16 (the policy / value heads: Linear(H, A), Linear(H, 1) and their gradients).
// A 32-wide MFMA tile is mostly padding there and the work is a single pass over the wide operand, so these are
// plain bandwidth kernels: every element of the wide matrix is read (or written) once, coalesced as float4.
//
//   skinny_n :  C[M, N<=16]  = act(A[M, K] B[N, K]^T + bias)          forward of a head
//   skinny_m :  C[M<=16, N] += A[K, M]^T B[K, N]  (+ column sums of A)  its weight (and bias) gradient
//   skinny_k :  C[M, N]    (+)= (A[M, K<=16] B[K, N]) * act'(Y)        its data gradient
//
// try_skinny() returns 1 when it took the call, 0 when the shapes / alignments want the MFMA path.
#pragma once
#include "srl_common.h"

namespace srlskinny {

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// one wavefront per row; lanes stride K in float4 steps; B (N x K) lives in LDS
template <int NB>
__global__ __launch_bounds__(256) void skinny_n_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                       long ldb, float* __restrict__ C, long ldc,
                                                       const float* __restrict__ bias, long M, int N, int K, int act,
                                                       int accumulate) {
  extern __shared__ float4 bs[];  // [N][K / 4]
  const int K4 = K >> 2;
  for (int e = threadIdx.x; e < N * K4; e += 256) {
    const int n = e / K4, k4 = e - n * K4;
    bs[e] = *reinterpret_cast<const float4*>(B + (long)n * ldb + 4 * k4);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  constexpr int RPW = 4;  // rows in flight per wavefront: their loads are issued together (latency, not bandwidth, bounds
                          // a one-row-at-a-time loop)
  for (long row0 = wave * RPW; row0 < M; row0 += nwaves * RPW) {
    float acc[RPW][NB];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[r][n] = 0.f;
    for (int k4 = lane; k4 < K4; k4 += 64) {
      float4 a[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const long row = row0 + r < M ? row0 + r : M - 1;
        a[r] = reinterpret_cast<const float4*>(A + row * lda)[k4];
      }
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (n < N) {
          const float4 b = bs[n * K4 + k4];
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            acc[r][n] += dot4(a[r], b);
            // Keeps the compiler from pairing the accumulators of two rows into packed float32 instructions
            // (v_pk_fma_f32).  With the packed code this kernel returned run-to-run different sums for the rows in the low
            // half of a pair whenever a kernel of ANOTHER queue shared the compute units (the two-pipeline mode of the
            // trainer), with bit-identical inputs; the scalar code is bit-reproducible there (scripts/pipe_probe.py;
            // an isolated chain of v_pk_fma_f32 beside an MFMA kernel is exact -- scripts/pk_hazard_probe.hip -- so the
            // cause is in the packed variant of THIS kernel's code, not in the instruction).
            asm volatile("" : "+v"(acc[r][n]));
          }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (n < N) acc[r][n] = wave_allsum(acc[r][n]);
      float v = 0.f;  // lane n keeps output n
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (lane == n) v = acc[r][n];
      if (lane < N && row0 + r < M) {
        if (bias) v += bias[lane];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        float* c = C + (row0 + r) * ldc + lane;
        *c = accumulate ? *c + v : v;
      }
    }
  }
}

// Short rows (K / 4 = KL lanes < 64; the 64-wide hidden layers of the multi-agent nets): with one wavefront per row 64 - KL lanes
// idle and every output costs six shuffle steps -- 455 us for a 307 200 x 64 -> 9 head (SMAC, full size) whose bytes take 16.
// Here a wavefront carries 64 / KL rows side by side (lane = row slot x KL + float4 column), RPW such groups in flight, and the
// sums close inside aligned groups of KL lanes (log2 KL shuffle steps).  Needs N <= KL (lane kk of a row's group keeps output kk).
template <int NB, int KL>
__global__ __launch_bounds__(256) void skinny_n_short_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                             long ldb, float* __restrict__ C, long ldc,
                                                             const float* __restrict__ bias, long M, int N, int act, int accumulate) {
  extern __shared__ float4 bs[];  // [N][KL]
  for (int e = threadIdx.x; e < N * KL; e += 256) {
    const int n = e / KL, k4 = e - n * KL;
    bs[e] = *reinterpret_cast<const float4*>(B + (long)n * ldb + 4 * k4);
  }
  __syncthreads();
  constexpr int G = 64 / KL, RPW = 4;
  const int lane = threadIdx.x & 63, g = lane / KL, kk = lane % KL;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long row0 = wave * (RPW * G); row0 < M; row0 += nwaves * (RPW * G)) {
    float4 a[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const long row = row0 + r * G + g < M ? row0 + r * G + g : M - 1;
      a[r] = reinterpret_cast<const float4*>(A + row * lda)[kk];
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float v = 0.f;
#pragma unroll
      for (int n = 0; n < NB; ++n)
        if (n < N) {
          float s = dot4(a[r], bs[n * KL + kk]);
          asm volatile("" : "+v"(s));  // (scalar accumulators: see skinny_n_kernel)
#pragma unroll
          for (int off = KL / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
          if (kk == n) v = s;
        }
      const long row = row0 + r * G + g;
      if (kk < N && row < M) {
        if (bias) v += bias[kk];
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = tanhf(v);
        float* c = C + row * ldc + kk;
        *c = accumulate ? *c + v : v;
      }
    }
  }
}

// A workgroup owns 16 consecutive columns of B / C (four float4 column-threads x 64 row-lanes) and a slab of k rows:
// a wavefront reads 64-byte row segments of B, the 64 row-lanes combine through LDS and four threads send the
// workgroup's M x 16 partial sums as atomics (a split over rows only would end in M x N atomics per workgroup, which is
// what bounded the first version of this kernel)
template <int MB>
__global__ __launch_bounds__(256) void skinny_m_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                       long ldb, float* __restrict__ C, long ldc, int M, long N, long K,
                                                       long rows_per_block, float* __restrict__ a_colsum) {
  const int c4 = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const long j = (long)blockIdx.x * 16 + 4 * c4;
  const long k0 = (long)blockIdx.y * rows_per_block;
  const long k1 = k0 + rows_per_block < K ? k0 + rows_per_block : K;
  float4 acc[MB];
  float cs[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f), cs[m] = 0.f;
  const bool do_cs = a_colsum != nullptr && blockIdx.x == 0 && c4 == 0;
  const bool live = j < N;
  constexpr int U = 4;  // rows in flight per thread
  for (long kb = k0 + rl; kb < k1; kb += 64 * U) {
    float4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long k = kb + 64 * u;
      b[u] = (live && k < k1) ? *reinterpret_cast<const float4*>(B + k * ldb + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long k = kb + 64 * u;
      if (k < k1) {
        const float* ar = A + k * lda;
#pragma unroll
        for (int m = 0; m < MB; ++m)
          if (m < M) {
            const float a = ar[m];
            acc[m].x = fmaf(a, b[u].x, acc[m].x); acc[m].y = fmaf(a, b[u].y, acc[m].y);
            acc[m].z = fmaf(a, b[u].z, acc[m].z); acc[m].w = fmaf(a, b[u].w, acc[m].w);
            if (do_cs) cs[m] += a;
          }
      }
    }
  }
  __shared__ float4 red[256];
  __shared__ float redc[64];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    if (m < M) {  // workgroup-uniform
      __syncthreads();
      red[threadIdx.x] = acc[m];
      if (c4 == 0) redc[rl] = cs[m];
      __syncthreads();
      if (threadIdx.x < 4 && live) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        float tc = 0.f;
        for (int r = 0; r < 64; ++r) {
          const float4 o = red[4 * r + c4];
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
          tc += redc[r];
        }
        float* c = C + (long)m * ldc + j;
        atomicAdd(c, t.x); atomicAdd(c + 1, t.y); atomicAdd(c + 2, t.z); atomicAdd(c + 3, t.w);
        if (do_cs) atomicAdd(a_colsum + m, tc);
      }
    }
  }
}

// threads own 4 consecutive columns and keep their K x 4 block of B in registers; rows are walked grid-stride
template <int KB_>
__global__ __launch_bounds__(256) void skinny_k_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                       long ldb, float* __restrict__ C, long ldc, long M, long N, int K,
                                                       const float* __restrict__ dact_src, long ld_dact, int dact,
                                                       int accumulate, int ct) {
  const int cx = threadIdx.x % ct, rt = threadIdx.x / ct, nrt = 256 / ct;
  const long j = ((long)blockIdx.x * ct + cx) * 4;
  if (j >= N) return;
  float4 b[KB_];
#pragma unroll
  for (int k = 0; k < KB_; ++k)
    b[k] = k < K ? *reinterpret_cast<const float4*>(B + (long)k * ldb + j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (long row = (long)blockIdx.y * nrt + rt; row < M; row += (long)gridDim.y * nrt) {
    const float* ar = A + row * lda;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KB_; ++k)
      if (k < K) {
        const float a = ar[k];
        v.x = fmaf(a, b[k].x, v.x); v.y = fmaf(a, b[k].y, v.y); v.z = fmaf(a, b[k].z, v.z); v.w = fmaf(a, b[k].w, v.w);
      }
    if (dact_src) {
      const float4 y = *reinterpret_cast<const float4*>(dact_src + row * ld_dact + j);
      if (dact == 1) {
        v.x = y.x > 0.f ? v.x : 0.f; v.y = y.y > 0.f ? v.y : 0.f; v.z = y.z > 0.f ? v.z : 0.f; v.w = y.w > 0.f ? v.w : 0.f;
      } else if (dact == 2) {
        v.x *= 1.f - y.x * y.x; v.y *= 1.f - y.y * y.y; v.z *= 1.f - y.z * y.z; v.w *= 1.f - y.w * y.w;
      }
    }
    float4* c = reinterpret_cast<float4*>(C + row * ldc + j);
    if (accumulate) {
      const float4 o = *c;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *c = v;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int col_threads(long N) {  // threads across the columns: a power of two <= 256 covering N / 4 if possible
  int ct = 1;
  while (ct < 256 && ct * 4 < N) ct <<= 1;
  return ct;
}

inline int try_skinny(hipStream_t st, const srl_gemm_desc* d) {
  const long M = d->M, N = d->N, K = d->K;
  if (M == 0 || N == 0 || d->mask_out || d->dact_mask) return 0;  // sign masks: the MFMA epilogue's business
  // ---- forward of a narrow head
  if (!d->a_kmajor && !d->b_kmajor && N <= 16 && M >= 64 && K >= 4 && K % 4 == 0 && d->lda % 4 == 0 && d->ldb % 4 == 0 &&
      al16(d->A) && al16(d->B) && !d->dact_src && d->split_k <= 1 && !d->a_colsum && N * K * 4 <= 48 * 1024) {
    long blocks = srl_ceil_div(M, 32L);  // every workgroup stages B once: few, fat workgroups
    if (blocks > 512) blocks = 512;
    const size_t lds = (size_t)N * K * 4;
    auto short_rows = [&](auto kern) {
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, d->A, d->lda, d->B, d->ldb, d->C, d->ldc, d->bias, M, (int)N,
                         d->act, d->accumulate);
    };
    if (K == 64 && N <= 8) short_rows(skinny_n_short_kernel<8, 16>);
    else if (K == 64) short_rows(skinny_n_short_kernel<16, 16>);
    else if (K == 128 && N <= 8) short_rows(skinny_n_short_kernel<8, 32>);
    else if (K == 128) short_rows(skinny_n_short_kernel<16, 32>);
    else if (K == 32 && N <= 8) short_rows(skinny_n_short_kernel<8, 8>);
    else if (N <= 8)
      hipLaunchKernelGGL(skinny_n_kernel<8>, dim3((unsigned)blocks), dim3(256), lds, st, d->A, d->lda, d->B, d->ldb, d->C,
                         d->ldc, d->bias, M, (int)N, (int)K, d->act, d->accumulate);
    else
      hipLaunchKernelGGL(skinny_n_kernel<16>, dim3((unsigned)blocks), dim3(256), lds, st, d->A, d->lda, d->B, d->ldb, d->C,
                         d->ldc, d->bias, M, (int)N, (int)K, d->act, d->accumulate);
    return 1;
  }
  // ---- weight (and bias) gradient of a narrow head: C += A^T B over many rows
  if (d->a_kmajor && d->b_kmajor && M <= 16 && K >= 256 && N % 4 == 0 && d->ldb % 4 == 0 && d->ldc % 4 == 0 && al16(d->B) &&
      al16(d->C) && d->accumulate && !d->bias && !d->act && !d->dact_src) {
    const long xblocks = srl_ceil_div(N, 16L);
    long slabs = srl_ceil_div(1024L, xblocks);
    if (slabs * 64 * 8 > K) slabs = srl_ceil_div(K, 64L * 8);  // at least 8 rows per row-lane
    if (slabs < 1) slabs = 1;
    const long rpb = srl_ceil_div(K, slabs);
    const dim3 grid((unsigned)xblocks, (unsigned)srl_ceil_div(K, rpb));
    if (M <= 8)
      hipLaunchKernelGGL(skinny_m_kernel<8>, grid, dim3(256), 0, st, d->A, d->lda, d->B, d->ldb, d->C, d->ldc, (int)M, N, K,
                         rpb, d->a_colsum);
    else
      hipLaunchKernelGGL(skinny_m_kernel<16>, grid, dim3(256), 0, st, d->A, d->lda, d->B, d->ldb, d->C, d->ldc, (int)M, N, K,
                         rpb, d->a_colsum);
    return 1;
  }
  // ---- data gradient through a narrow head: C (+)= A B with a short K
  if (!d->a_kmajor && d->b_kmajor && K <= 16 && K >= 1 && M >= 64 && N % 4 == 0 && d->ldb % 4 == 0 && d->ldc % 4 == 0 &&
      al16(d->B) && al16(d->C) && !d->bias && !d->act && d->split_k <= 1 && !d->a_colsum &&
      (!d->dact_src || (d->ld_dact % 4 == 0 && al16(d->dact_src)))) {
    const int ct = col_threads(N);
    const long xblocks = srl_ceil_div(N, 4L * ct);
    const long nrt = 256 / ct;
    long yblocks = srl_ceil_div(M, 8 * nrt);  // >= 8 rows per thread: B's registers are loaded once per thread
    if (yblocks * xblocks > 2048) yblocks = srl_ceil_div(2048L, xblocks);
    if (yblocks < 1) yblocks = 1;
    const dim3 grid((unsigned)xblocks, (unsigned)yblocks);
    if (K <= 8)
      hipLaunchKernelGGL(skinny_k_kernel<8>, grid, dim3(256), 0, st, d->A, d->lda, d->B, d->ldb, d->C, d->ldc, M, N, (int)K,
                         d->dact_src, d->ld_dact, d->dact, d->accumulate, ct);
    else
      hipLaunchKernelGGL(skinny_k_kernel<16>, grid, dim3(256), 0, st, d->A, d->lda, d->B, d->ldb, d->C, d->ldc, M, N, (int)K,
                         d->dact_src, d->ld_dact, d->dact, d->accumulate, ct);
    return 1;
  }
  return 0;
}

}  // namespace srlskinny
