// The fused MLP chain of mlp_small.hip on the float32 matrix cores, for row counts that fill the chip (round 4).
//
// mlp_small.hip walks 16 rows per workgroup through the layers with scalar FMAs: right for 256 rows (latency-bound), 4 % of the
// float32-MFMA roof at 65 536.  Here a WAVEFRONT owns 32 rows and keeps them in registers through the whole chain:
//   * a layer is D[out channel][row] = W . X^T on v_mfma_f32_32x32x2_f32 (exact float32 multiply-adds, the same arithmetic as
//     the FMA chain): weights are the A operand -- staged ONCE per workgroup into LDS in fragment order, every layer of the
//     chain resident -- and the activations the B operand;
//   * the accumulator layout IS the next layer's operand layout: lane (row r, half hb) holds channels (e & 3) + 8 (e >> 2)
//     + 4 hb of a 32-channel block in registers e = 0..15, and an MFMA of the next layer takes register e of both half-lanes as
//     its two k-values -- so the weights' k order is permuted once at staging and NO data moves between layers;
//   * bias = the accumulator's initial value, activation / LayerNorm on the accumulator registers (a row's features are 16
//     registers in each of two lanes: sums are in-lane adds and one cross-lane exchange);
//   * the tape (every layer's input, for the backward pass) is the only traffic besides x and y.
// Backward: the data gradient is the same chain with W^T fragments; the weight gradient dW[o][k] = sum_rows dz[r][o] x[r][k]
// reduces over ROWS, so both operands go through a 32-row transposition tile in LDS (rows become the k dimension) and the
// products accumulate in persistent accumulator blocks -- one 32 x 32 block per (out block, in block) of every Linear, for all
// the rows a wavefront walks -- that are folded across the workgroup in LDS and reach the gradient buffer once per workgroup.
// Bias / LayerNorm-affine gradients are column sums of the same tiles, kept per lane in LDS.
// Eligible chains: widths <= 64 (two 32-channel blocks), <= 12 accumulator blocks; the others keep mlp_small.hip's kernels.
// Reference: modules/utils.py:154-161 (mlp), actor_critic_policy.py:92-107 (heads).
#pragma once

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMB = 2;          // 32-channel blocks per side of a layer at most
constexpr int kTld = 68;        // floats per row of a transposition tile (64 + 4: float4 stores stay aligned)
constexpr int kMaxAcc = 12;     // persistent dW accumulator blocks of a chain at most (192 accumulator registers)

struct MArgs {
  Args a;
  int wf[SRL_MLP_MAX_LAYERS];    // LDS float offset: Linear forward fragments [nbo][nbi][16][64]; LayerNorm: gamma | beta tables
  int tb[SRL_MLP_MAX_LAYERS];    // Linear: bias table [nbo * 32]
  int wt[SRL_MLP_MAX_LAYERS];    // backward: Linear transposed fragments [nbi][nbo][16][64] (layers > 0); LayerNorm: gamma table
  int accb[SRL_MLP_MAX_LAYERS];  // backward: first accumulator block of a Linear
  int pg[SRL_MLP_MAX_LAYERS];    // backward: offset of the layer's per-lane sums (Linear: bias gradient; LayerNorm: dgamma | dbeta)
  int fwd_floats, bwd_floats, nacc, npg;
};

__device__ __forceinline__ int mm_ch(int e, int hb) { return (e & 3) + 8 * (e >> 2) + 4 * hb; }

// rows [row] of a row-major [rows][ld] matrix, columns 0 .. dim - 1, into the accumulator layout (zeros beyond dim / the rows)
__device__ __forceinline__ void mm_load(const float* base, long ld, long row, bool rok, int dim, int hb, float (&v)[kMB][16]) {
  const bool vec = (ld & 3) == 0 && ((uintptr_t)base & 15) == 0;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 32 * ib + 8 * j + 4 * hb;
      if (rok && vec && col + 3 < dim) {
        const float4 q = *reinterpret_cast<const float4*>(base + row * ld + col);
        v[ib][4 * j] = q.x; v[ib][4 * j + 1] = q.y; v[ib][4 * j + 2] = q.z; v[ib][4 * j + 3] = q.w;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[ib][4 * j + q] = (rok && col + q < dim) ? base[row * ld + col + q] : 0.f;
      }
    }
}
__device__ __forceinline__ void mm_store(float* base, long ld, long row, bool rok, int dim, int hb, const float (&v)[kMB][16]) {
  if (!rok) return;
  const bool vec = (ld & 3) == 0 && ((uintptr_t)base & 15) == 0;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 32 * ib + 8 * j + 4 * hb;
      if (vec && col + 3 < dim) {
        *reinterpret_cast<float4*>(base + row * ld + col) = make_float4(v[ib][4 * j], v[ib][4 * j + 1], v[ib][4 * j + 2], v[ib][4 * j + 3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (col + q < dim) base[row * ld + col + q] = v[ib][4 * j + q];
      }
    }
}

// LayerNorm statistics of the rows held in the accumulator layout (padding channels hold zeros)
__device__ __forceinline__ void mm_ln_stats(const float (&v)[kMB][16], int dim, int hb, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[ib][e];
  s += __shfl_xor(s, 32);
  mean = s / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float d = v[ib][e] - mean;
      q += (32 * ib + mm_ch(e, hb) < dim) ? d * d : 0.f;
    }
  q += __shfl_xor(q, 32);
  rstd = rsqrtf(q / (float)dim + kLnEps);
}

// stage the chain's parameters into LDS in operand order (every thread of the workgroup)
__device__ __forceinline__ void mm_stage(const MArgs& m, float* sm, int tid, bool bwd) {
  for (int i = 0; i < m.a.n; ++i) {
    const Layer L = m.a.L[i];
    if (L.kind == 1) {
      const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
      if (!bwd) {
        for (int e = tid; e < nbo * nbi * 1024; e += 256) {
          const int l = e & 63, e16 = (e >> 6) & 15, blk = e >> 10, ib = blk % nbi, ob = blk / nbi;
          const int o = 32 * ob + (l & 31), k = 32 * ib + mm_ch(e16, l >> 5);
          sm[m.wf[i] + e] = (o < L.out && k < L.in) ? L.w[o * L.in + k] : 0.f;
        }
        for (int c = tid; c < nbo * 32; c += 256) sm[m.tb[i] + c] = (c < L.out && L.b) ? L.b[c] : 0.f;
      } else if (i > 0) {
        for (int e = tid; e < nbo * nbi * 1024; e += 256) {
          const int l = e & 63, e16 = (e >> 6) & 15, blk = e >> 10, ob = blk % nbo, ib = blk / nbo;
          const int k = 32 * ib + (l & 31), o = 32 * ob + mm_ch(e16, l >> 5);
          sm[m.wt[i] + e] = (o < L.out && k < L.in) ? L.w[o * L.in + k] : 0.f;
        }
      }
    } else {
      const int nb = (L.in + 31) >> 5;
      const int off = bwd ? m.wt[i] : m.wf[i];
      for (int c = tid; c < nb * 32; c += 256) {
        sm[off + c] = c < L.in ? L.w[c] : 0.f;
        if (!bwd) sm[off + nb * 32 + c] = c < L.in ? L.b[c] : 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(256) void mlp_fwd_mfma_kernel(MArgs m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const Args& a = m.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  mm_stage(m, sm, tid, false);
  __syncthreads();
  const long ntiles = (a.rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool rok = row < a.rows;
    float cur[kMB][16];
    mm_load(a.x, a.ldx, row, rok, a.L[0].in, hb, cur);
    int dim = a.L[0].in;
    for (int i = 0; i < a.n; ++i) {
      const Layer L = a.L[i];
      if (i > 0) mm_store(a.tape + L.toff, a.tld, row, rok, L.in, hb, cur);  // this layer's input: what the backward pass reads back
      if (L.kind == 0) {
        float mean, rstd;
        mm_ln_stats(cur, L.in, hb, mean, rstd);
        const float* gt = sm + m.wf[i];
        const int nb = (L.in + 31) >> 5;
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4;
            if (ib < nb) {
              g4 = *reinterpret_cast<const float4*>(gt + 32 * ib + 8 * j + 4 * hb);
              b4 = *reinterpret_cast<const float4*>(gt + nb * 32 + 32 * ib + 8 * j + 4 * hb);
            }
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[ib][4 * j + q] = fmaf((cur[ib][4 * j + q] - mean) * rstd, gv[q], bv[q]);
          }
        dim = L.in;
      } else {
        const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
        f32x16 acc[kMB];
#pragma unroll
        for (int ob = 0; ob < kMB; ++ob) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[ob][e] = 0.f;
          if (ob < nbo) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 b4 = *reinterpret_cast<const float4*>(sm + m.tb[i] + 32 * ob + 8 * j + 4 * hb);
              acc[ob][4 * j] = b4.x; acc[ob][4 * j + 1] = b4.y; acc[ob][4 * j + 2] = b4.z; acc[ob][4 * j + 3] = b4.w;
            }
#pragma unroll
            for (int ib = 0; ib < kMB; ++ib)
              if (ib < nbi) {
                const float* wfr = sm + m.wf[i] + (ob * nbi + ib) * 1024 + lane;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfr[64 * e], cur[ib][e], acc[ob], 0, 0, 0);
              }
          }
        }
        // (blocks beyond the layer: zeros.  The activation is wave-uniform: branches, not a select over an evaluated tanh)
        if (L.act == 1) {
#pragma unroll
          for (int ob = 0; ob < kMB; ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[ob][e] = fmaxf(acc[ob][e], 0.f);
        } else if (L.act == 2) {
#pragma unroll
          for (int ob = 0; ob < kMB; ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[ob][e] = tanhf(acc[ob][e]);
        } else {
#pragma unroll
          for (int ob = 0; ob < kMB; ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[ob][e] = acc[ob][e];
        }
        dim = L.out;
      }
    }
    mm_store(a.y, a.ldy, row, rok, dim, hb, cur);
  }
}

// one 32 x 32 block of a weight gradient: rows of the two transposition tiles are the k dimension
__device__ __forceinline__ void mm_wgrad_block(f32x16& acc, const float* dT, const float* xT, int ob, int ib, int lane) {
  const float* ap = dT + (lane >> 5) * kTld + 32 * ob + (lane & 31);
  const float* bp = xT + (lane >> 5) * kTld + 32 * ib + (lane & 31);
#pragma unroll
  for (int mm = 0; mm < 16; ++mm) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * mm * kTld], bp[2 * mm * kTld], acc, 0, 0, 0);
}

__device__ __forceinline__ void mm_tile_write(float* T, int r, int hb, const float (&v)[kMB][16]) {
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(T + r * kTld + 32 * ib + 8 * j + 4 * hb) = make_float4(v[ib][4 * j], v[ib][4 * j + 1], v[ib][4 * j + 2], v[ib][4 * j + 3]);
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_mfma_kernel(MArgs m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const Args& a = m.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  // LDS: parameters (m.bwd_floats) | per wavefront: dT, xT tiles, per-lane sums [npg][64]
  float* const dT = sm + m.bwd_floats + wave * (2 * 32 * kTld + m.npg * 64);
  float* const xT = dT + 32 * kTld;
  float* const pgs = xT + 32 * kTld;
  mm_stage(m, sm, tid, true);
  for (int e = lane; e < m.npg * 64; e += 64) pgs[e] = 0.f;
  __syncthreads();
  f32x16 A0, A1, A2, A3, A4, A5, A6, A7, A8, A9, A10, A11;
#pragma unroll
  for (int e = 0; e < 16; ++e) A0[e] = A1[e] = A2[e] = A3[e] = A4[e] = A5[e] = A6[e] = A7[e] = A8[e] = A9[e] = A10[e] = A11[e] = 0.f;
#define SRL_MM_BLOCKS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
  const long ntiles = (a.rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool rok = row < a.rows;
    float d[kMB][16], xin[kMB][16];
    {
      const Layer& last = a.L[a.n - 1];
      mm_load(a.dy, a.lddy, row, rok, last.kind == 1 ? last.out : last.in, hb, d);
    }
    for (int i = a.n - 1; i >= 0; --i) {
      const Layer L = a.L[i];
      if (i == 0) mm_load(a.x, a.ldx, row, rok, L.in, hb, xin);
      else mm_load(a.tape + L.toff, a.tld, row, rok, L.in, hb, xin);
      // the activation that produced this input: its derivative (from the input's value) closes the data gradient
      const int pact = (i > 0 && a.L[i - 1].kind == 1) ? a.L[i - 1].act : 0;
      if (L.kind == 1) {
        const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
        mm_tile_write(dT, r, hb, d);
        mm_tile_write(xT, r, hb, xin);
        // (a wavefront's LDS operations execute in order: its own tiles need no barrier)
        if (L.gb) {  // bias gradient: column sums of dz, channel = lane
          float s = 0.f;
#pragma unroll 8
          for (int rr = 0; rr < 32; ++rr) s += dT[rr * kTld + lane];
          pgs[m.pg[i] * 64 + lane] += s;
        }
        for (int ob = 0; ob < nbo; ++ob)
          for (int ib = 0; ib < nbi; ++ib) {
            switch (m.accb[i] + ob * nbi + ib) {
#define SRL_MM_CASE(k) case k: mm_wgrad_block(A##k, dT, xT, ob, ib, lane); break;
              SRL_MM_BLOCKS(SRL_MM_CASE)
#undef SRL_MM_CASE
            }
          }
        if (i > 0) {
          f32x16 acc[kMB];
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ib][e] = 0.f;
            if (ib < nbi) {
#pragma unroll
              for (int ob = 0; ob < kMB; ++ob)
                if (ob < nbo) {
                  const float* wfr = sm + m.wt[i] + (ib * nbo + ob) * 1024 + lane;
#pragma unroll
                  for (int e = 0; e < 16; ++e) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfr[64 * e], d[ob][e], acc[ib], 0, 0, 0);
                }
            }
          }
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) d[ib][e] = acc[ib][e] * act_der(xin[ib][e], pact);
        }
      } else {
        // LayerNorm: statistics recomputed from the input; dgamma / dbeta = column sums of gy * xhat / gy
        float mean, rstd;
        mm_ln_stats(xin, L.in, hb, mean, rstd);
        const float* gt = sm + m.wt[i];
        const int nb = (L.in + 31) >> 5;
        float gg[kMB][16];
        float m1 = 0.f, m2 = 0.f;
        mm_tile_write(xT, r, hb, d);  // gy: its column sums are dbeta
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ib < nb) g4 = *reinterpret_cast<const float4*>(gt + 32 * ib + 8 * j + 4 * hb);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
            float gyx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int e = 4 * j + q;
              const bool in = 32 * ib + mm_ch(e, hb) < L.in;
              const float xh = in ? (xin[ib][e] - mean) * rstd : 0.f;
              xin[ib][e] = in ? xin[ib][e] : 0.f;
              gyx[q] = d[ib][e] * xh;
              gg[ib][e] = d[ib][e] * gv[q];
              m1 += gg[ib][e];
              m2 = fmaf(gg[ib][e], xh, m2);
            }
            *reinterpret_cast<float4*>(dT + r * kTld + 32 * ib + 8 * j + 4 * hb) = make_float4(gyx[0], gyx[1], gyx[2], gyx[3]);  // dgamma's terms
          }
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        m1 /= (float)L.in;
        m2 /= (float)L.in;
        {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
          for (int rr = 0; rr < 32; ++rr) {
            s1 += dT[rr * kTld + lane];
            s2 += xT[rr * kTld + lane];
          }
          pgs[m.pg[i] * 64 + lane] += s1;
          pgs[(m.pg[i] + 1) * 64 + lane] += s2;
        }
        if (i > 0) {
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const bool in = 32 * ib + mm_ch(e, hb) < L.in;
              const float xh = (xin[ib][e] - mean) * rstd;
              d[ib][e] = in ? rstd * (gg[ib][e] - m1 - xh * m2) * act_der(xin[ib][e], pact) : 0.f;
            }
        }
      }
    }
  }
  // ---- fold the wavefronts' sums in LDS, then one atomic per parameter and workgroup --------------------------------------------
  // (the per-lane sums sit inside the region that becomes the fold buffer: each wavefront takes its own out first)
  float pgv[2 * SRL_MLP_MAX_LAYERS];
#pragma unroll
  for (int q = 0; q < 2 * SRL_MLP_MAX_LAYERS; ++q) pgv[q] = q < m.npg ? pgs[q * 64 + lane] : 0.f;
  __syncthreads();  // every wavefront is past its tiles: the tile region is free
  float* const red = sm + m.bwd_floats;  // [kMaxP]: fits the tile region (mm_plan)
  for (int e = tid; e < kMaxP; e += 256) red[e] = 0.f;
  __syncthreads();
  for (int i = 0; i < a.n; ++i) {
    const Layer L = a.L[i];
    if (L.kind == 1) {
      const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
      for (int ob = 0; ob < nbo; ++ob)
        for (int ib = 0; ib < nbi; ++ib) {
          // accumulator of dz^T x: rows = out channel (register e, half hb), columns = in channel (lane & 31)
          const int k = 32 * ib + (lane & 31);
          float* const dst = red + L.pw + k;
          const bool kok = k < L.in;
          switch (m.accb[i] + ob * nbi + ib) {
#define SRL_MM_CASE(kk)                                                                  \
  case kk:                                                                               \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                     \
      const int o = 32 * ob + mm_ch(e, hb);                                              \
      if (o < L.out && kok) atomicAdd(dst + o * L.in, A##kk[e]);                         \
    }                                                                                    \
    break;
            SRL_MM_BLOCKS(SRL_MM_CASE)
#undef SRL_MM_CASE
          }
        }
    }
  }
#pragma unroll
  for (int q = 0; q < 2 * SRL_MLP_MAX_LAYERS; ++q) {
    // which layer's sums sit in slot q: found by walking the layers (uniform, a handful of iterations)
    for (int i = 0; i < a.n; ++i) {
      const Layer L = a.L[i];
      if (L.kind == 1 && L.gb && m.pg[i] == q && lane < L.out) atomicAdd(red + L.pb + lane, pgv[q]);
      if (L.kind == 0 && m.pg[i] == q && lane < L.in) atomicAdd(red + L.pw + lane, pgv[q]);
      if (L.kind == 0 && m.pg[i] + 1 == q && lane < L.in) atomicAdd(red + L.pb + lane, pgv[q]);
    }
  }
  __syncthreads();
  for (int i = 0; i < a.n; ++i) {
    const Layer L = a.L[i];
    const int nw = L.kind == 1 ? L.out * L.in : L.in, nbv = L.kind == 1 ? L.out : L.in;
    for (int e = tid; e < nw; e += 256) atomicAdd(L.gw + e, red[L.pw + e]);
    if (L.gb)
      for (int e = tid; e < nbv; e += 256) atomicAdd(L.gb + e, red[L.pb + e]);
  }
#undef SRL_MM_BLOCKS
}

// LDS plan of a chain for the MFMA kernels; false: not eligible
inline bool mm_plan(MArgs& m) {
  const Args& a = m.a;
  int f = 0, b = 0, nacc = 0, npg = 0;
  for (int i = 0; i < a.n; ++i) {
    const Layer& L = a.L[i];
    if (L.in > 32 * kMB || L.out > 32 * kMB) return false;
    const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
    if (L.kind == 1) {
      m.wf[i] = f; f += nbo * nbi * 1024;
      m.tb[i] = f; f += nbo * 32;
      m.wt[i] = b; if (i > 0) b += nbo * nbi * 1024;
      m.accb[i] = nacc; nacc += nbo * nbi;
      m.pg[i] = npg; npg += 1;
    } else {
      m.wf[i] = f; f += 2 * nbi * 32;
      m.wt[i] = b; b += nbi * 32;
      m.accb[i] = nacc;
      m.pg[i] = npg; npg += 2;
    }
  }
  m.fwd_floats = f; m.bwd_floats = b; m.nacc = nacc; m.npg = npg;
  if (nacc > kMaxAcc || !a.lds_acc) return false;
  const long fwd_bytes = 4L * f, tiles = 4L * (2 * 32 * kTld + npg * 64);
  const long bwd_bytes = 4L * b + 4L * (tiles > kMaxP ? tiles : kMaxP);
  return fwd_bytes <= 150 * 1024 && bwd_bytes <= 150 * 1024;
}

}  // namespace
