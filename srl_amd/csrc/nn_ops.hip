// Bandwidth-bound pieces of the actor/critic network: LayerNorm fwd/bwd, whole-observation LayerNorm
// statistics, patch gathers (im2col) with the observation LayerNorm fused into the uint8 load path,
// the gather form of col2im, column sums, strided copies.
#include "srl_common.h"
#include "gemm_bf16x3.h"

namespace {

constexpr float kLnEps = 1e-5f;  // nn.LayerNorm default, used everywhere in the reference

// ---- LayerNorm over the last dim: one wavefront per row ------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* x, long ldx, const float* gamma,
                                                            const float* beta, long rows, int D, float* y, long ldy,
                                                            float* mean_out, float* rstd_out) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ldx;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += xr[j];
  const float mean = wave_allsum(s) / (float)D;
  float q = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float d = xr[j] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(wave_allsum(q) / (float)D + kLnEps);
  float* yr = y + row * ldy;
  for (int j = lane; j < D; j += 64) yr[j] = (xr[j] - mean) * rstd * gamma[j] + beta[j];
  if (lane == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

// Narrow LayerNorms (D <= 128, the hidden sizes of the MLP / recurrent policies): LPR = 4..32 lanes own one row
// (one float4 each), so a wavefront normalises 64/LPR rows at once and its reductions are LPR-wide butterflies
// instead of a 64-wide one per 256-byte row.  Needs D % 4 == 0, pitches % 4 == 0 and 16-byte aligned bases.
template <int LPR>
__device__ __forceinline__ float sub_allsum(float v) {
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_small_kernel(const float* x, long ldx, const float* gamma,
                                                                  const float* beta, long rows, int D, float* y,
                                                                  long ldy, float* mean_out, float* rstd_out) {
  constexpr int RW = 64 / LPR;
  const int lane = threadIdx.x & 63, sl = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW + lane / LPR;
  const bool ok = row < rows && sl * 4 < D;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) v = *reinterpret_cast<const float4*>(x + row * ldx + sl * 4);
  const float mean = sub_allsum<LPR>(v.x + v.y + v.z + v.w) / (float)D;
  const float4 d = ok ? make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float rstd = rsqrtf(sub_allsum<LPR>(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) / (float)D + kLnEps);
  if (!ok) return;
  const float4 g = *reinterpret_cast<const float4*>(gamma + sl * 4), b = *reinterpret_cast<const float4*>(beta + sl * 4);
  *reinterpret_cast<float4*>(y + row * ldy + sl * 4) =
      make_float4(d.x * rstd * g.x + b.x, d.y * rstd * g.y + b.y, d.z * rstd * g.z + b.z, d.w * rstd * g.w + b.w);
  if (sl == 0) {
    mean_out[row] = mean;
    rstd_out[row] = rstd;
  }
}

template <int LPR>
__global__ __launch_bounds__(1024) void layernorm_bwd_small_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                                  const float* gamma, const float* mean,
                                                                  const float* rstd, long rows, int D, float* dx,
                                                                  long lddx, int dact, float* dgamma, float* dbeta,
                                                                  long rows_per_block) {
  constexpr int RW = 64 / LPR;
  const int nwv = blockDim.x >> 6;  // 4 or 16 wavefronts: every workgroup ends with 2 D atomics on the same addresses as
                                    // all the others, so big inputs use few, fat workgroups
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, sl = lane % LPR, sub = lane / LPR;
  const bool col_ok = sl * 4 < D;
  float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col_ok) gm = *reinterpret_cast<const float4*>(gamma + sl * 4);
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long base = r0 + (long)wid * RW; base < r1; base += (long)nwv * RW) {
    const long row = base + sub;
    const bool ok = row < r1 && col_ok;
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
    float mu = 0.f, rs = 0.f;
    if (ok) {
      const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + sl * 4);
      const float4 c = *reinterpret_cast<const float4*>(dy + row * lddy + sl * 4);
      xv[0] = a.x, xv[1] = a.y, xv[2] = a.z, xv[3] = a.w;
      dv[0] = c.x, dv[1] = c.y, dv[2] = c.z, dv[3] = c.w;
      mu = mean[row], rs = rstd[row];
    }
    const float gv[4] = {gm.x, gm.y, gm.z, gm.w};
    float s1 = 0.f, s2 = 0.f, xh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[e] = ok ? (xv[e] - mu) * rs : 0.f;
      const float g = dv[e] * gv[e];
      s1 += g;
      s2 += g * xh[e];
      pg[e] += dv[e] * xh[e];
      pb[e] += dv[e];
    }
    if (dx) {
      const float m1 = sub_allsum<LPR>(s1) / (float)D, m2 = sub_allsum<LPR>(s2) / (float)D;
      if (ok) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = rs * (dv[e] * gv[e] - m1 - xh[e] * m2);
          if (dact) o[e] *= act_grad_from_output(xv[e], dact);  // x is the activation output that fed this LayerNorm
        }
        *reinterpret_cast<float4*>(dx + row * lddx + sl * 4) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  // fold the 64/LPR row groups of the wavefront, then the 4 wavefronts through LDS, then one atomic per column
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int m = LPR; m < 64; m <<= 1) {
      pg[e] += __shfl_xor(pg[e], m, 64);
      pb[e] += __shfl_xor(pb[e], m, 64);
    }
  }
  __shared__ float sg[16][LPR * 8];
  if (lane < LPR) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sg[wid][lane * 8 + e] = pg[e];
      sg[wid][lane * 8 + 4 + e] = pb[e];
    }
  }
  __syncthreads();
  if (wid == 0 && lane < LPR && col_ok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float tg = 0.f, tb = 0.f;
      for (int w = 0; w < nwv; ++w) tg += sg[w][lane * 8 + e], tb += sg[w][lane * 8 + 4 + e];
      atomicAdd(dgamma + sl * 4 + e, tg);
      atomicAdd(dbeta + sl * 4 + e, tb);
    }
  }
}

inline bool ln_small_ok(const void* a, const void* b, const void* c, long lda, long ldb, int D) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return D <= 128 && D % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && al(a) && al(b) && al(c);
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += sum dy*xhat; dbeta += sum dy.
// Each workgroup walks `rows_per_block` rows with 4 wavefronts, keeps per-lane partial dgamma/dbeta in
// registers (D <= 64*LN_MAXV) and flushes them once with float atomics.
constexpr int LN_MAXV = 16;  // supports D <= 1024 in the register path
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                            const float* gamma, const float* mean, const float* rstd,
                                                            long rows, int D, float* dx, long lddx, int dact,
                                                            float* dgamma, float* dbeta, long rows_per_block,
                                                            float* dx_absmax) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float amx = 0.f;  // max |dx| of this wavefront's rows (dx_absmax: the range of the consumer's operand at no extra pass)
  float pg[LN_MAXV], pb[LN_MAXV];
#pragma unroll
  for (int v = 0; v < LN_MAXV; ++v) pg[v] = pb[v] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long row = r0 + wid; row < r1; row += 4) {
    const float* xr = x + row * ldx;
    const float* dyr = dy + row * lddy;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < LN_MAXV; ++v) {
      const int j = lane + v * 64;
      if (j < D) {
        const float xh = (xr[j] - mu) * rs, d = dyr[j];
        const float g = d * gamma[j];
        s1 += g;
        s2 += g * xh;
        pg[v] += d * xh;
        pb[v] += d;
      }
    }
    if (dx) {
      const float m1 = wave_allsum(s1) / (float)D, m2 = wave_allsum(s2) / (float)D;
      float* dxr = dx + row * lddx;
#pragma unroll
      for (int v = 0; v < LN_MAXV; ++v) {
        const int j = lane + v * 64;
        if (j < D) {
          const float xv = xr[j];
          const float xh = (xv - mu) * rs;
          float o = rs * (dyr[j] * gamma[j] - m1 - xh * m2);
          if (dact) o *= act_grad_from_output(xv, dact);  // x is the activation output that fed this LayerNorm
          dxr[j] = o;
          amx = fmaxf(amx, fabsf(o));
        }
      }
    }
  }
  if (dx_absmax) {  // the slot only grows: load first, atomic only when this wavefront has something to add (gemm_core.h)
    amx = wave_allmax(amx);
    if (lane == 0 && amx > 0.f) {
      unsigned int* d = reinterpret_cast<unsigned int*>(dx_absmax);
      if (__float_as_uint(amx) > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, __float_as_uint(amx));
    }
  }
  // combine the 4 wavefronts through LDS, then one atomic per column per workgroup
  __shared__ float sg[4][64 * LN_MAXV / 4 + 1];  // processed in 4 passes of LN_MAXV/4 values to bound LDS
  for (int pass = 0; pass < 4; ++pass) {
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
#pragma unroll
      for (int v = 0; v < LN_MAXV / 4; ++v) sg[wid][v * 64 + lane] = which ? pb[pass * (LN_MAXV / 4) + v] : pg[pass * (LN_MAXV / 4) + v];
      __syncthreads();
      if (wid == 0) {
#pragma unroll
        for (int v = 0; v < LN_MAXV / 4; ++v) {
          const int j = lane + (pass * (LN_MAXV / 4) + v) * 64;
          if (j < D) {
            const float t = sg[0][v * 64 + lane] + sg[1][v * 64 + lane] + sg[2][v * 64 + lane] + sg[3][v * 64 + lane];
            atomicAdd((which ? dbeta : dgamma) + j, t);
          }
        }
      }
    }
  }
}

// D a multiple of 256, at most 1024 (the 512-wide feature LayerNorm of the Atari nets), 16-byte aligned rows: a lane owns float4
// columns 4 lane + 256 v, a row's x and dy are read ONCE into registers and serve both the row sums and dx (the kernel above
// reads them twice in 4-byte pieces: 50 us on 16 384 x 512, this one ~half).  Same sums, same flush.
template <int NV>
__global__ __launch_bounds__(512) void layernorm_bwd_vec_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                                const float* gamma, const float* mean, const float* rstd,
                                                                long rows, float* dx, long lddx, int dact, float* dgamma,
                                                                float* dbeta, long rows_per_block, float* dx_absmax) {
  constexpr int D = 256 * NV;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float amx = 0.f;
  float4 g4[NV], pg[NV], pb[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    g4[v] = *reinterpret_cast<const float4*>(gamma + 4 * lane + 256 * v);
    pg[v] = pb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (long row = r0 + wid; row < r1; row += 8) {
    const float mu = mean[row], rs = rstd[row];
    float4 xv[NV], dv[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      xv[v] = *reinterpret_cast<const float4*>(x + row * ldx + 4 * lane + 256 * v);
      dv[v] = *reinterpret_cast<const float4*>(dy + row * lddy + 4 * lane + 256 * v);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float xa[4] = {xv[v].x, xv[v].y, xv[v].z, xv[v].w}, da[4] = {dv[v].x, dv[v].y, dv[v].z, dv[v].w};
      const float ga[4] = {g4[v].x, g4[v].y, g4[v].z, g4[v].w};
      float* pga = reinterpret_cast<float*>(&pg[v]);
      float* pba = reinterpret_cast<float*>(&pb[v]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = (xa[q] - mu) * rs, g = da[q] * ga[q];
        s1 += g;
        s2 += g * xh;
        pga[q] += da[q] * xh;
        pba[q] += da[q];
      }
    }
    if (dx) {
      const float m1 = wave_allsum(s1) / (float)D, m2 = wave_allsum(s2) / (float)D;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float xa[4] = {xv[v].x, xv[v].y, xv[v].z, xv[v].w}, da[4] = {dv[v].x, dv[v].y, dv[v].z, dv[v].w};
        const float ga[4] = {g4[v].x, g4[v].y, g4[v].z, g4[v].w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xh = (xa[q] - mu) * rs;
          o[q] = rs * (da[q] * ga[q] - m1 - xh * m2);
          if (dact) o[q] *= act_grad_from_output(xa[q], dact);
          amx = fmaxf(amx, fabsf(o[q]));
        }
        *reinterpret_cast<float4*>(dx + row * lddx + 4 * lane + 256 * v) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  if (dx_absmax) {
    amx = wave_allmax(amx);
    if (lane == 0 && amx > 0.f) {
      unsigned int* d = reinterpret_cast<unsigned int*>(dx_absmax);
      if (__float_as_uint(amx) > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, __float_as_uint(amx));
    }
  }
  // the 8 wavefronts' column sums fold in LDS, then one atomic per column and workgroup (at most 256 workgroups: the 2 D float
  // atomics of each were the kernel's floor at 1024 workgroups -- 1 M atomics, ~36 us)
  __shared__ float sg[2][D];
  for (int e = threadIdx.x; e < 2 * D; e += 512) sg[0][e] = 0.f;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float* pga = reinterpret_cast<const float*>(&pg[v]);
    const float* pba = reinterpret_cast<const float*>(&pb[v]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      atomicAdd(&sg[0][4 * lane + 256 * v + q], pga[q]);
      atomicAdd(&sg[1][4 * lane + 256 * v + q], pba[q]);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < D; e += 512) {
    atomicAdd(dgamma + e, sg[0][e]);
    atomicAdd(dbeta + e, sg[1][e]);
  }
}

// Wide rows (D > 1024: the halving Linear towers behind the default convolution stack, modules/cnn.py:86-91): one
// workgroup per row for dx, and a separate column pass for dgamma / dbeta in which a thread owns one column and walks a
// slab of rows (coalesced across the workgroup), one atomic per column per slab.
__global__ __launch_bounds__(256) void layernorm_bwd_wide_dx_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                                    const float* gamma, const float* mean,
                                                                    const float* rstd, int D, float* dx, long lddx,
                                                                    int dact) {
  __shared__ float red[2][4];
  const long row = blockIdx.x;
  const float* xr = x + row * ldx;
  const float* dyr = dy + row * lddy;
  const float mu = mean[row], rs = rstd[row];
  float s1 = 0.f, s2 = 0.f;
  for (int j = threadIdx.x; j < D; j += 256) {
    const float g = dyr[j] * gamma[j];
    s1 += g;
    s2 += g * (xr[j] - mu) * rs;
  }
  s1 = wave_allsum(s1);
  s2 = wave_allsum(s2);
  if ((threadIdx.x & 63) == 0) red[0][threadIdx.x >> 6] = s1, red[1][threadIdx.x >> 6] = s2;
  __syncthreads();
  const float m1 = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)D;
  const float m2 = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)D;
  float* dxr = dx + row * lddx;
  for (int j = threadIdx.x; j < D; j += 256) {
    const float xv = xr[j];
    const float xh = (xv - mu) * rs;
    float o = rs * (dyr[j] * gamma[j] - m1 - xh * m2);
    if (dact) o *= act_grad_from_output(xv, dact);
    dxr[j] = o;
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_wide_param_kernel(const float* dy, long lddy, const float* x, long ldx,
                                                                       const float* mean, const float* rstd, long rows,
                                                                       int D, float* dgamma, float* dbeta,
                                                                       long rows_per_block) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  if (j >= D) return;
  float pg = 0.f, pb = 0.f;
  for (long row = r0; row < r1; ++row) {
    const float d = dy[row * lddy + j];
    pg += d * (x[row * ldx + j] - mean[row]) * rstd[row];
    pb += d;
  }
  atomicAdd(dgamma + j, pg);
  atomicAdd(dbeta + j, pb);
}

// ---- whole-observation LayerNorm statistics: one workgroup per sample ------------------------------------
template <bool U8>
__global__ __launch_bounds__(256) void obs_ln_stats_kernel(const void* obs, long n, int D, float* mean, float* rstd) {
  __shared__ double red[8];
  const long s = blockIdx.x;
  double acc[2] = {0.0, 0.0};
  if (U8) {
    // exact integer sums: D*255 and D*255^2 stay far below 2^63
    const uint8_t* p = static_cast<const uint8_t*>(obs) + s * D;
    unsigned long long a = 0, b = 0;
    for (int j = threadIdx.x; j < D; j += 256) {
      const unsigned v = p[j];
      a += v;
      b += v * v;
    }
    acc[0] = (double)a;
    acc[1] = (double)b;
  } else {
    const float* p = static_cast<const float*>(obs) + s * D;
    for (int j = threadIdx.x; j < D; j += 256) {
      const double v = p[j];
      acc[0] += v;
      acc[1] += v * v;
    }
  }
  block_sum<2, 256>(acc, red);
  if (threadIdx.x == 0) {
    const double mu = acc[0] / D;
    double var = acc[1] / D - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[s] = (float)mu;
    rstd[s] = (float)(1.0 / sqrt(var + (double)kLnEps));
  }
}

// ---- im2col of the first convolution with the observation LayerNorm fused in ---------------------------------
// P[(s*OH + oh)*OW + ow][(c*KH + kh)*KW + kw] = LN(obs)[s, c, oh*S + kh, ow*S + kw]
template <bool U8>
__global__ __launch_bounds__(256) void im2col_obs_ln_kernel(const void* obs, const float* mean, const float* rstd,
                                                            const float* gamma, const float* beta, long n, int C, int H,
                                                            int W, int KH, int KW, int S, int OH, int OW, float* P) {
  const int Kp = C * KH * KW;
  const long total = n * OH * OW * (long)Kp;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int k = (int)(e % Kp);
    const long m = e / Kp;
    const int ow = (int)(m % OW), oh = (int)((m / OW) % OH);
    const long s = m / ((long)OW * OH);
    const int kw = k % KW, kh = (k / KW) % KH, c = k / (KW * KH);
    const int pos = (c * H + oh * S + kh) * W + ow * S + kw;
    const long src = s * (long)C * H * W + pos;
    const float x = U8 ? (float)static_cast<const uint8_t*>(obs)[src] : static_cast<const float*>(obs)[src];
    P[e] = (x - mean[s]) * rstd[s] * gamma[pos] + beta[pos];
  }
}

// P[(s*OH+oh)*OW+ow][(kh*KW + kw)*C + c] = x[s, oh*S+kh, ow*S+kw, c]   (NHWC; V = 4: float4 along c, V = 1: any C)
template <int V>
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const float* x, long n, int H, int W, int C, int KH, int KW,
                                                          int S, int OH, int OW, float* P) {
  const int CV = C / V;
  const int Kq = KH * KW * CV;
  const long total = n * OH * OW * (long)Kq;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int kq = (int)(e % Kq);
    const long m = e / Kq;
    const int ow = (int)(m % OW), oh = (int)((m / OW) % OH);
    const long s = m / ((long)OW * OH);
    const int cv = kq % CV, kw = (kq / CV) % KW, kh = kq / (CV * KW);
    const float* src = x + ((s * H + oh * S + kh) * W + ow * S + kw) * C + cv * V;
    if (V == 4) *reinterpret_cast<float4*>(P + e * 4) = *reinterpret_cast<const float4*>(src);
    else P[e] = src[0];
  }
}

// dX[s,h,w,c] = sum_{kh,kw : (h-kh)%S==0, (w-kw)%S==0, in range} dP[(s,oh,ow)][(kh,kw,c)]  (* act'(y))
template <int V>
__global__ __launch_bounds__(256) void col2im_nhwc_kernel(const float* dP, long n, int H, int W, int C, int KH, int KW,
                                                          int S, int OH, int OW, const float* y, int dact, float* dX) {
  const int CV = C / V;
  const long total = n * H * W * (long)CV;
  const long Kp = (long)KH * KW * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int cv = (int)(e % CV);
    const long pix = e / CV;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const long s = pix / ((long)W * H);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int kh = h % S; kh < KH && kh <= h; kh += S) {
      const int oh = (h - kh) / S;
      if (oh >= OH) continue;
      for (int kw = w % S; kw < KW && kw <= w; kw += S) {
        const int ow = (w - kw) / S;
        if (ow >= OW) continue;
        const float* src = dP + ((s * OH + oh) * OW + ow) * Kp + (kh * KW + kw) * C + cv * V;
        if (V == 4) {
          const float4 v = *reinterpret_cast<const float4*>(src);
          acc[0] += v.x; acc[1 % V] += v.y; acc[2 % V] += v.z; acc[3 % V] += v.w;
        } else {
          acc[0] += src[0];
        }
      }
    }
    if (y && dact) {
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] *= act_grad_from_output(y[e * V + j], dact);
    }
    if (V == 4) *reinterpret_cast<float4*>(dX + e * 4) = make_float4(acc[0], acc[1 % V], acc[2 % V], acc[3 % V]);
    else dX[e] = acc[0];
  }
}

// LayerNorm-affine gradients of the observation LayerNorm in front of the first convolution.
// thread <-> observation position (c,h,w); grid.y splits the samples; float atomics at the end.
template <bool U8>
__global__ __launch_bounds__(256) void obs_ln_affine_bwd_kernel(const float* dP, const void* obs, const float* mean,
                                                                const float* rstd, long n, int C, int H, int W, int KH,
                                                                int KW, int S, int OH, int OW, float* dgamma,
                                                                float* dbeta) {
  const int D = C * H * W;
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= D) return;
  const int w = pos % W, h = (pos / W) % H, c = pos / (W * H);
  const long Kp = (long)C * KH * KW;
  const long per = srl_ceil_div(n, gridDim.y);
  const long s0 = (long)blockIdx.y * per, s1 = s0 + per < n ? s0 + per : n;
  float ag = 0.f, ab = 0.f;
  for (long s = s0; s < s1; ++s) {
    float d = 0.f;
    for (int kh = h % S; kh < KH && kh <= h; kh += S) {
      const int oh = (h - kh) / S;
      if (oh >= OH) continue;
      for (int kw = w % S; kw < KW && kw <= w; kw += S) {
        const int ow = (w - kw) / S;
        if (ow >= OW) continue;
        d += dP[((s * OH + oh) * OW + ow) * Kp + (c * KH + kh) * KW + kw];
      }
    }
    const long src = s * (long)D + pos;
    const float x = U8 ? (float)static_cast<const uint8_t*>(obs)[src] : static_cast<const float*>(obs)[src];
    ag += d * ((x - mean[s]) * rstd[s]);
    ab += d;
  }
  atomicAdd(dgamma + pos, ag);
  atomicAdd(dbeta + pos, ab);
}

// out[j] (+)= sum_i x[i*ld + j]; 64 columns x 4 row-lanes per workgroup, grid.y splits rows.
__global__ __launch_bounds__(256) void colsum_kernel(const float* x, long ld, long rows, int cols, float* out) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  const long per = srl_ceil_div(rows, gridDim.y);
  const long r0 = (long)blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s = 0.f;
  if (col < cols)
    for (long r = r0 + ry; r < r1; r += 4) s += x[r * ld + col];
  part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && col < cols) atomicAdd(out + col, part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, long lds_, float* dst, long ldd, long rows,
                                                     int cols) {
  const long total = rows * cols;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
    dst[(e / cols) * ldd + e % cols] = src[(e / cols) * lds_ + e % cols];
}

__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* src, float* dst, long n) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) dst[e] = (float)src[e];
}

unsigned stream_grid(long work_items) {
  const long b = srl_ceil_div(work_items, 256);
  return (unsigned)(b < 1 ? 1 : (b < 16384 ? b : 16384));
}

}  // namespace

extern "C" int srl_layernorm_fwd(void* stream, const float* x, int64_t ldx, const float* gamma, const float* beta,
                                 int64_t rows, int D, float* y, int64_t ldy, float* mean, float* rstd) {
  SRL_CHECK_ARG(x && gamma && beta && y && mean && rstd && D >= 1 && rows >= 0, "null tensor");
  if (rows == 0) return 0;
  if (ln_small_ok(x, y, gamma, ldx, ldy, D) && (reinterpret_cast<uintptr_t>(beta) & 15) == 0) {
    hipStream_t st = (hipStream_t)stream;
#define SRL_LN_FWD(LPR)                                                                                               \
  hipLaunchKernelGGL(layernorm_fwd_small_kernel<LPR>, dim3((unsigned)srl_ceil_div(rows, 4L * (64 / LPR))), dim3(256), 0, \
                     st, x, ldx, gamma, beta, rows, D, y, ldy, mean, rstd)
    if (D <= 16) SRL_LN_FWD(4);
    else if (D <= 32) SRL_LN_FWD(8);
    else if (D <= 64) SRL_LN_FWD(16);
    else SRL_LN_FWD(32);
#undef SRL_LN_FWD
    SRL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)srl_ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     x, ldx, gamma, beta, rows, D, y, ldy, mean, rstd);
  SRL_LAUNCH_CHECK();
  return 0;
}

static int layernorm_bwd_impl(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                              const float* mean, const float* rstd, int64_t rows, int D, float* dx, int64_t lddx, int dact,
                              float* dgamma, float* dbeta, float* dx_absmax, int* tracked);

extern "C" int srl_layernorm_bwd(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx,
                                 const float* gamma, const float* mean, const float* rstd, int64_t rows, int D,
                                 float* dx, int64_t lddx, int dact, float* dgamma, float* dbeta, float* dx_absmax) {
  SRL_CHECK_ARG(!dx_absmax || (dx && lddx == D), "dx_absmax: a dense dx");
  int tracked = 0;
  const int rc = layernorm_bwd_impl(stream, dy, lddy, x, ldx, gamma, mean, rstd, rows, D, dx, lddx, dact, dgamma, dbeta,
                                    dx_absmax, &tracked);
  if (rc != 0 || !dx_absmax || tracked || rows == 0) return rc;
  return srl_absmax(stream, dx, rows * (int64_t)D, dx_absmax);  // the narrow / wide kernels do not track: one pass over dx
}

static int layernorm_bwd_impl(void* stream, const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                              const float* mean, const float* rstd, int64_t rows, int D, float* dx, int64_t lddx, int dact,
                              float* dgamma, float* dbeta, float* dx_absmax, int* tracked) {
  SRL_CHECK_ARG(dy && x && gamma && mean && rstd && dgamma && dbeta, "null tensor");
  SRL_CHECK_ARG(D >= 1, "LayerNorm width must be positive");
  if (rows == 0) return 0;
  if (D > 64 * LN_MAXV) {
    hipStream_t st = (hipStream_t)stream;
    if (dx)
      hipLaunchKernelGGL(layernorm_bwd_wide_dx_kernel, dim3((unsigned)rows), dim3(256), 0, st, dy, lddy, x, ldx, gamma, mean,
                         rstd, D, dx, lddx, dact);
    const long col_blocks = srl_ceil_div((long)D, 256L);
    long slabs = srl_ceil_div(2048L, col_blocks);  // ~2048 workgroups in all
    if (slabs > rows) slabs = rows;
    const long rpb_w = srl_ceil_div(rows, slabs);
    hipLaunchKernelGGL(layernorm_bwd_wide_param_kernel, dim3((unsigned)col_blocks, (unsigned)srl_ceil_div(rows, rpb_w)),
                       dim3(256), 0, st, dy, lddy, x, ldx, mean, rstd, rows, D, dgamma, dbeta, rpb_w);
    SRL_LAUNCH_CHECK();
    return 0;
  }
  // ~4 workgroups per CU; each flushes D*2 atomics, so keep the row share large
  long rpb = srl_ceil_div(rows, 1024);
  if (rpb < 16) rpb = 16;
  if (ln_small_ok(x, dy, gamma, ldx, lddy, D) && (!dx || ((reinterpret_cast<uintptr_t>(dx) & 15) == 0 && lddx % 4 == 0))) {
    hipStream_t st = (hipStream_t)stream;
    if (rpb < 64) rpb = 64;
    int threads = 256;
    if (rows >= 65536) {  // 16-wavefront workgroups, at most 256 of them
      threads = 1024;
      rpb = srl_ceil_div(rows, 256);
      if (rpb < 256) rpb = 256;
    }
#define SRL_LN_BWD(LPR)                                                                                              \
  hipLaunchKernelGGL(layernorm_bwd_small_kernel<LPR>, dim3((unsigned)srl_ceil_div(rows, rpb)), dim3(threads), 0, st, dy, lddy, \
                     x, ldx, gamma, mean, rstd, rows, D, dx, lddx, dact, dgamma, dbeta, rpb)
    if (D <= 16) SRL_LN_BWD(4);
    else if (D <= 32) SRL_LN_BWD(8);
    else if (D <= 64) SRL_LN_BWD(16);
    else SRL_LN_BWD(32);
#undef SRL_LN_BWD
    SRL_LAUNCH_CHECK();
    return 0;
  }
  const bool vec_ok = D % 256 == 0 && D <= 1024 && ldx % 4 == 0 && lddy % 4 == 0 && (!dx || lddx % 4 == 0) &&
                      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(gamma) |
                        reinterpret_cast<uintptr_t>(dx)) & 15) == 0;
  if (vec_ok) {
    long rpv = srl_ceil_div(rows, 256L);  // at most 256 workgroups of 8 wavefronts
    if (rpv < 8) rpv = 8;
#define SRL_LN_VEC(NV)                                                                                                         \
  hipLaunchKernelGGL(layernorm_bwd_vec_kernel<NV>, dim3((unsigned)srl_ceil_div(rows, rpv)), dim3(512), 0, (hipStream_t)stream, dy, \
                     lddy, x, ldx, gamma, mean, rstd, rows, dx, lddx, dact, dgamma, dbeta, rpv, dx_absmax)
    switch (D / 256) {
      case 1: SRL_LN_VEC(1); break;
      case 2: SRL_LN_VEC(2); break;
      case 3: SRL_LN_VEC(3); break;
      default: SRL_LN_VEC(4); break;
    }
#undef SRL_LN_VEC
    *tracked = 1;
    SRL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)srl_ceil_div(rows, rpb)), dim3(256), 0, (hipStream_t)stream,
                     dy, lddy, x, ldx, gamma, mean, rstd, rows, D, dx, lddx, dact, dgamma, dbeta, rpb, dx_absmax);
  *tracked = 1;
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_obs_ln_stats(void* stream, const void* obs, int is_u8, int64_t n, int D, float* mean, float* rstd) {
  SRL_CHECK_ARG(obs && mean && rstd && D >= 1, "null tensor");
  if (n == 0) return 0;
  if (is_u8) hipLaunchKernelGGL(obs_ln_stats_kernel<true>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, obs, n, D, mean, rstd);
  else hipLaunchKernelGGL(obs_ln_stats_kernel<false>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, obs, n, D, mean, rstd);
  SRL_LAUNCH_CHECK();
  return 0;
}

static int conv_out(int in, int k, int s) { return (in - k) / s + 1; }

extern "C" int srl_im2col_obs_ln(void* stream, const void* obs, int is_u8, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int64_t n, int C, int H, int W, int KH, int KW,
                                 int stride, float* P) {
  SRL_CHECK_ARG(obs && mean && rstd && gamma && beta && P, "null tensor");
  SRL_CHECK_ARG(KH <= H && KW <= W && stride >= 1, "bad conv geometry");
  if (n == 0) return 0;
  const int OH = conv_out(H, KH, stride), OW = conv_out(W, KW, stride);
  const long total = n * OH * OW * (long)C * KH * KW;
  if (is_u8) hipLaunchKernelGGL(im2col_obs_ln_kernel<true>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, obs, mean, rstd, gamma, beta, n, C, H, W, KH, KW, stride, OH, OW, P);
  else hipLaunchKernelGGL(im2col_obs_ln_kernel<false>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, obs, mean, rstd, gamma, beta, n, C, H, W, KH, KW, stride, OH, OW, P);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_im2col_nhwc(void* stream, const float* x, int64_t n, int H, int W, int C, int KH, int KW, int stride,
                               float* P) {
  SRL_CHECK_ARG(x && P, "null tensor");
  SRL_CHECK_ARG(C >= 1 && KH <= H && KW <= W && stride >= 1, "invalid geometry");
  if (n == 0) return 0;
  const int OH = conv_out(H, KH, stride), OW = conv_out(W, KW, stride);
  const int V = C % 4 == 0 ? 4 : 1;
  const long total = n * OH * OW * (long)KH * KW * (C / V);
  if (V == 4) hipLaunchKernelGGL(im2col_nhwc_kernel<4>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, n, H, W, C, KH, KW, stride, OH, OW, P);
  else hipLaunchKernelGGL(im2col_nhwc_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, x, n, H, W, C, KH, KW, stride, OH, OW, P);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_col2im_nhwc(void* stream, const float* dP, int64_t n, int H, int W, int C, int KH, int KW,
                               int stride, const float* y, int dact, float* dX) {
  SRL_CHECK_ARG(dP && dX, "null tensor");
  SRL_CHECK_ARG(C >= 1 && KH <= H && KW <= W && stride >= 1, "invalid geometry");
  if (n == 0) return 0;
  const int OH = conv_out(H, KH, stride), OW = conv_out(W, KW, stride);
  const int V = C % 4 == 0 ? 4 : 1;
  const long total = n * H * W * (long)(C / V);
  if (V == 4) hipLaunchKernelGGL(col2im_nhwc_kernel<4>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, dP, n, H, W, C, KH, KW, stride, OH, OW, y, dact, dX);
  else hipLaunchKernelGGL(col2im_nhwc_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, (hipStream_t)stream, dP, n, H, W, C, KH, KW, stride, OH, OW, y, dact, dX);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_obs_ln_affine_bwd(void* stream, const float* dP, const void* obs, int is_u8, const float* mean,
                                     const float* rstd, int64_t n, int C, int H, int W, int KH, int KW, int stride,
                                     float* dgamma, float* dbeta) {
  SRL_CHECK_ARG(dP && obs && mean && rstd && dgamma && dbeta, "null tensor");
  if (n == 0) return 0;
  const int OH = conv_out(H, KH, stride), OW = conv_out(W, KW, stride);
  const int D = C * H * W;
  const unsigned gx = (unsigned)srl_ceil_div(D, 256);
  long gy = srl_ceil_div(2048, gx);  // ~2048 workgroups in flight
  if (gy > n) gy = n;
  if (gy < 1) gy = 1;
  if (is_u8) hipLaunchKernelGGL(obs_ln_affine_bwd_kernel<true>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dP, obs, mean, rstd, n, C, H, W, KH, KW, stride, OH, OW, dgamma, dbeta);
  else hipLaunchKernelGGL(obs_ln_affine_bwd_kernel<false>, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dP, obs, mean, rstd, n, C, H, W, KH, KW, stride, OH, OW, dgamma, dbeta);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_colsum(void* stream, const float* x, int64_t ld, int64_t rows, int cols, float* out, int accumulate) {
  SRL_CHECK_ARG(x && out && cols >= 1, "null tensor");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) SRL_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * cols, st));
  if (rows == 0) return 0;
  const unsigned gx = (unsigned)srl_ceil_div(cols, 64);
  long gy = srl_ceil_div(rows, 256);  // >= 256 rows per workgroup
  if (gy > 2048) gy = 2048;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(colsum_kernel, dim3(gx, (unsigned)gy), dim3(256), 0, st, x, ld, rows, cols, out);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_copy2d(void* stream, const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols) {
  SRL_CHECK_ARG(src && dst && cols >= 1, "null tensor");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(copy2d_kernel, dim3(stream_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd,
                     rows, cols);
  SRL_LAUNCH_CHECK();
  return 0;
}

namespace {
// 16-byte loads over the aligned body, the ragged head / tail element-wise by the first workgroup
__global__ __launch_bounds__(256) void absmax_kernel(const float* x, long n, float* out) {
  const long head = (4 - (((uintptr_t)x >> 2) & 3)) & 3;  // elements before the first 16-byte boundary
  const long h = head < n ? head : n;
  const long nv = (n - h) / 4;
  const float4* xv = reinterpret_cast<const float4*>(x + h);
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const float4 v = xv[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0) {
    for (long i = threadIdx.x; i < h; i += 256) m = fmaxf(m, fabsf(x[i]));
    for (long i = h + 4 * nv + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
  }
  m = wave_allmax(m);
  // Thousands of wavefronts folding into ONE address serialise at ~10 ns per atomic (8 192 of them were 85 of this kernel's
  // 98 us on a 33 MB tensor; reading the slot first does not help -- they all start together and all read the old value).  So:
  // at most 512 workgroups (the launcher), one atomic each.
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > 0.f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}
}  // namespace

// four consecutive float32 -> the 16 bytes gemm3_kernel's BPRE variant expects in their place: the four first pieces h0 =
// f16(x s), then the four second pieces h1 = f16(x s - h0), s = range_scale(*absmax) -- the arithmetic of split2h_quad
namespace {
__global__ __launch_bounds__(256) void presplit_kernel(const float4* src, const float* absmax, float4* dst, long quads) {
  const float sc = srlgemm::range_scale(absmax);
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
    const float4 v = src[q];
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint2 pl[2];
    srlgemm::split2h_quad(f, sc, pl);
    dst[q] = make_float4(__uint_as_float(pl[0].x), __uint_as_float(pl[0].y), __uint_as_float(pl[1].x), __uint_as_float(pl[1].y));
  }
}
}  // namespace

extern "C" int srl_presplit(void* stream, const float* src, const float* absmax, float* dst, int64_t n) {
  SRL_CHECK_ARG(src && absmax && dst && n >= 0 && n % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
                "null / unaligned tensor, or a length that is not a multiple of 4");
  if (n == 0) return 0;
  const long quads = n / 4;
  long blocks = srl_ceil_div(quads, 256L);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), absmax, reinterpret_cast<float4*>(dst), quads);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_absmax(void* stream, const float* x, int64_t n, float* out) {
  SRL_CHECK_ARG(out && n >= 0, "null output");
  if (n == 0) return 0;
  SRL_CHECK_ARG(x != nullptr && ((uintptr_t)x & 3) == 0, "null / unaligned tensor");
  long blocks = srl_ceil_div(n, 4096);  // at least sixteen floats per thread
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
  SRL_LAUNCH_CHECK();
  return 0;
}

namespace {
// one thread per word: 32 floats as eight 16-byte loads
__global__ __launch_bounds__(256) void relu_mask_kernel(const float4* __restrict__ x, long nwords, uint32_t* __restrict__ mask) {
  for (long w = (long)blockIdx.x * 256 + threadIdx.x; w < nwords; w += (long)gridDim.x * 256) {
    uint32_t bits = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = x[w * 8 + q];
      bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * q);
    }
    mask[w] = bits;
  }
}
}  // namespace

extern "C" int srl_relu_mask(void* stream, const float* x, int64_t n, uint32_t* mask) {
  SRL_CHECK_ARG(x && mask && n >= 0 && n % 32 == 0 && ((uintptr_t)x & 15) == 0, "null / unaligned tensor, or a length that is not a multiple of 32");
  if (n == 0) return 0;
  const long nwords = n / 32;
  long blocks = srl_ceil_div(nwords, 256L);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x), nwords, mask);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_u8_to_f32(void* stream, const uint8_t* src, float* dst, int64_t n) {
  SRL_CHECK_ARG(src && dst, "null tensor");
  if (n == 0) return 0;
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  SRL_LAUNCH_CHECK();
  return 0;
}

// ---- space-to-depth of a planar observation + whole-observation LayerNorm statistics, one workgroup per sample -----
namespace {
template <bool U8>
__global__ __launch_bounds__(256) void obs_s2d_kernel(const void* obs, long n, int C, int H, int W, int S, void* out,
                                                      float* mean, float* rstd) {
  __shared__ double red[8];
  const long smp = blockIdx.x;
  const int D = C * H * W, Wb = W / S, SS = S * S;
  double acc[2] = {0.0, 0.0};
  if (U8 && S == 4 && W % 4 == 0 && H % 4 == 0) {
    // dword path: the 4 bytes (pw = 0..3) of one (c, h, b) stay together.  The sample is staged in LDS so that both
    // the read and the re-tiled write are fully coalesced.
    extern __shared__ __attribute__((aligned(16))) uint32_t stage[];
    const uint32_t* src = reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(obs) + smp * D);
    uint32_t* dst = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(out) + smp * D);
    // 32-bit partial sums are exact: even the whole sample's sum of squares fits (D * 255^2 < 2^32 for D <= 66051,
    // and this path is taken for D <= 61440, the LDS staging limit)
    uint32_t a = 0, b = 0;
    auto tally = [&](uint32_t w) {
      const unsigned b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
      a += b0 + b1 + b2 + b3;
      b += b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
    };
    // H and W are multiples of 4 here, so D is a multiple of 16 and every sample starts 16-byte aligned
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    uint4* st4 = reinterpret_cast<uint4*>(stage);
    for (int e = threadIdx.x; e < D / 16; e += 256) {
      const uint4 q = src4[e];
      st4[e] = q;
      tally(q.x); tally(q.y); tally(q.z); tally(q.w);
    }
    __syncthreads();
    // one thread writes the 16 bytes of a (block, channel): ph = 0..3 are rows ab*4 + ph of the source plane
    uint4* dst4 = reinterpret_cast<uint4*>(dst);
    for (int o4 = threadIdx.x; o4 < D / 16; o4 += 256) {  // o4 = (ab*Wb + bq)*C + c
      const int c = o4 % C, blk = o4 / C;
      const int bq = blk % Wb, ab = blk / Wb;
      const uint32_t* sp = stage + (c * H + ab * 4) * Wb + bq;
      dst4[o4] = make_uint4(sp[0], sp[Wb], sp[2 * Wb], sp[3 * Wb]);
    }
    acc[0] = (double)a;
    acc[1] = (double)b;
  } else {
    for (int e = threadIdx.x; e < D; e += 256) {
      const int x = e % W, y = (e / W) % H, c = e / (W * H);
      const int o = (((y / S) * Wb + x / S) * C + c) * SS + (y % S) * S + x % S;
      double v;
      if (U8) {
        const uint8_t t = static_cast<const uint8_t*>(obs)[smp * D + e];
        static_cast<uint8_t*>(out)[smp * D + o] = t;
        v = (double)t;
      } else {
        const float t = static_cast<const float*>(obs)[smp * D + e];
        static_cast<float*>(out)[smp * D + o] = t;
        v = (double)t;
      }
      acc[0] += v;
      acc[1] += v * v;
    }
  }
  block_sum<2, 256>(acc, red);
  if (threadIdx.x == 0) {
    const double mu = acc[0] / D;
    double var = acc[1] / D - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[smp] = (float)mu;
    rstd[smp] = (float)(1.0 / sqrt(var + (double)kLnEps));
  }
}
}  // namespace

extern "C" int srl_obs_space_to_depth(void* stream, const void* obs, int is_u8, int64_t n, int C, int H, int W, int s,
                                      void* out, float* mean, float* rstd) {
  SRL_CHECK_ARG(obs && out && mean && rstd, "null tensor");
  SRL_CHECK_ARG(s >= 1 && H % s == 0 && W % s == 0, "stride must divide H and W");
  if (n == 0) return 0;
  const size_t lds = (is_u8 && s == 4 && W % 4 == 0 && (size_t)C * H * W <= 60 * 1024) ? (size_t)C * H * W : 0;
  SRL_CHECK_ARG(!(is_u8 && s == 4 && W % 4 == 0) || lds, "observation too large for the LDS-staged re-tiling");
  if (is_u8) hipLaunchKernelGGL(obs_s2d_kernel<true>, dim3((unsigned)n), dim3(256), lds, (hipStream_t)stream, obs, (long)n, C, H, W, s, out, mean, rstd);
  else hipLaunchKernelGGL(obs_s2d_kernel<false>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, obs, (long)n, C, H, W, s, out, mean, rstd);
  SRL_LAUNCH_CHECK();
  return 0;
}
