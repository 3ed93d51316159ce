// The matrix-core MLP chain of mlp_mfma.h with the chain's SHAPE as a template argument (round 5).
//
// mlp_mfma.h walks any eligible chain: layer kinds, widths and activations are run-time values, so every layer of every 32-row
// tile pays for descriptor loads, four-way switches over the MFMA counts, masks for widths the compiler cannot see and register
// arrays sized for the widest case -- 2 400 vector instructions per tile of the forward pass beside 168 MFMAs (the counters: matrix
// pipe 41 % busy, vector pipe 37 %, and they do not overlap), and 115 spilled registers in the backward pass.  Here the same
// algorithm -- same fragments in LDS, same MFMA sequences, same sums in the same order: the results equal mlp_mfma.h's to the
// compiler's choice of fused multiply-adds in the LayerNorm arithmetic -- is instantiated per signature, a list of (kind, in, out, activation): offsets, trip counts and masks are
// constants, a 4-wide input costs 4 registers instead of 32, and nothing is left of a layer but its MFMAs, its LDS reads and the
// arithmetic of its activation / LayerNorm.  srl_mlp_fwd / srl_mlp_bwd look a chain up in the list below (the towers of the
// reference's vector-observation policies at BASELINE's sizes) and fall back to mlp_mfma.h's kernels for every other chain.
// Reference: modules/utils.py:154-161 (mlp), actor_critic_policy.py:92-107 (heads).
#pragma once

namespace {

#define SRL_SIG_LN(d) (0 | ((d) << 3) | ((d) << 10))
#define SRL_SIG_LIN(i, o, act) (1 | ((act) << 1) | ((i) << 3) | ((o) << 10))

template <int... C>
struct MSig {
  static constexpr int n = sizeof...(C);
  static constexpr int code[sizeof...(C)] = {C...};
  static constexpr int kind(int i) { return code[i] & 1; }
  static constexpr int act(int i) { return (code[i] >> 1) & 3; }
  static constexpr int in(int i) { return (code[i] >> 3) & 127; }
  static constexpr int out(int i) { return (code[i] >> 10) & 127; }
  static constexpr int nbi(int i) { return (in(i) + 31) >> 5; }
  static constexpr int nbo(int i) { return (out(i) + 31) >> 5; }
  // LDS plan (the same layout as mm_plan's)
  static constexpr int wf(int i) {
    int f = 0;
    for (int k = 0; k < i; ++k) f += kind(k) == 1 ? nbo(k) * nbi(k) * kFB + nbo(k) * 32 : 2 * nbi(k) * 32;
    return f;
  }
  static constexpr int tb(int i) { return wf(i) + nbo(i) * nbi(i) * kFB; }
  static constexpr int par_floats = wf(n);
  static constexpr int pg(int i) {
    int p = 0;
    for (int k = 0; k < i; ++k) p += kind(k) == 1 ? 1 : 2;
    return p;
  }
  static constexpr int npg = pg(n);
  static constexpr int lin(int i) {   // index of Linear layer i among the chain's Linear layers
    int p = 0;
    for (int k = 0; k < i; ++k) p += kind(k);
    return p;
  }
  static constexpr int nlin = lin(n);
  static constexpr int out_dim = kind(n - 1) == 1 ? out(n - 1) : in(n - 1);
  static bool matches(const Args& a) {
    if (a.n != n) return false;
    for (int i = 0; i < n; ++i)
      if (a.L[i].kind != kind(i) || a.L[i].in != in(i) || (kind(i) == 1 && (a.L[i].out != out(i) || a.L[i].act != act(i)))) return false;
    return true;
  }
};

constexpr int sx_ne(int dim, int blk) {   // = mm_ne
  const int left = dim - 32 * blk;
  return left >= 32 ? 16 : (left <= 0 ? 0 : 4 * ((left + 7) >> 3));
}

struct XArgs {
  const float* x;
  long ldx, rows;
  float* y;
  long ldy;
  const float* dy;
  long lddy;
  float* dx;
  long lddx;
  const float* w[kMaxNL];
  const float* b[kMaxNL];
  float* gw[kMaxNL];
  float* gb[kMaxNL];
  int dbg;
};

// channel 32 ib + mm_ch(e, hb) < DIM (a constant wherever DIM is a multiple of 8)
template <int DIM>
__device__ __forceinline__ bool sx_valid(int ib, int e, int hb) {
  if constexpr (DIM % 8 == 0) return 32 * ib + (e & 3) + 8 * (e >> 2) < DIM;
  else return 32 * ib + mm_ch(e, hb) < DIM;
}

// rows of a row-major matrix, columns 0 .. DIM - 1, into the accumulator layout: only the registers below sx_ne are touched
template <int DIM>
__device__ __forceinline__ void sx_load(const float* base, long ld, long row, bool rok, int hb, float (&v)[kMB][16]) {
  const bool vec = (ld & 3) == 0 && ((uintptr_t)base & 15) == 0;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * j < sx_ne(DIM, ib)) {
        const int col = 32 * ib + 8 * j + 4 * hb;
        if (rok && vec && col + 3 < DIM) {
          const float4 q = *reinterpret_cast<const float4*>(base + row * ld + col);
          v[ib][4 * j] = q.x; v[ib][4 * j + 1] = q.y; v[ib][4 * j + 2] = q.z; v[ib][4 * j + 3] = q.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) v[ib][4 * j + q] = (rok && col + q < DIM) ? base[row * ld + col + q] : 0.f;
        }
      }
}
template <int DIM>
__device__ __forceinline__ void sx_store(float* base, long ld, long row, bool rok, int hb, const float (&v)[kMB][16]) {
  if (!rok) return;
  const bool vec = (ld & 3) == 0 && ((uintptr_t)base & 15) == 0;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * j < sx_ne(DIM, ib)) {
        const int col = 32 * ib + 8 * j + 4 * hb;
        if (vec && col + 3 < DIM) {
          *reinterpret_cast<float4*>(base + row * ld + col) = make_float4(v[ib][4 * j], v[ib][4 * j + 1], v[ib][4 * j + 2], v[ib][4 * j + 3]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (col + q < DIM) base[row * ld + col + q] = v[ib][4 * j + q];
        }
      }
}

template <class S>
__device__ __forceinline__ void sx_stage(const XArgs& a, float* sm, int tid, int nthr) {
  for (int e = tid; e < S::par_floats; e += nthr) sm[e] = 0.f;   // the padding of every block and table
  __syncthreads();
  mm_static_for<S::n>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr int in = S::in(i), out = S::out(i);
    const float* w = a.w[i];
    const float* b = a.b[i];
    if constexpr (S::kind(i) == 1) {
      constexpr int nbi = S::nbi(i);
      for (int idx = tid; idx < out * in; idx += nthr) {   // coalesced over the in channel; the LDS rows of 65 take any order
        const int o = idx / in, k = idx - o * in;
        const int kl = k & 31, blk = (o >> 5) * nbi + (k >> 5);
        sm[S::wf(i) + blk * kFB + ((kl & 3) + 4 * (kl >> 3)) * kFP + ((kl >> 2) & 1) * 32 + (o & 31)] = w[idx];
      }
      if (b)
        for (int c = tid; c < out; c += nthr) sm[S::tb(i) + c] = b[c];
    } else {
      constexpr int nb = S::nbi(i);
      for (int c = tid; c < in; c += nthr) {
        sm[S::wf(i) + c] = w[c];
        sm[S::wf(i) + nb * 32 + c] = b[c];
      }
    }
  });
}

// LayerNorm statistics of the rows held in the accumulator layout (= mm_ln_stats: the same sums in the same order)
template <int DIM>
__device__ __forceinline__ void sx_ln_stats(const float (&v)[kMB][16], int hb, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < sx_ne(DIM, ib)) s += v[ib][e];
  s += __shfl_xor(s, 32);
  mean = s / (float)DIM;
  float q = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < sx_ne(DIM, ib)) {
        const float d = v[ib][e] - mean;
        q += sx_valid<DIM>(ib, e, hb) ? d * d : 0.f;
      }
  q += __shfl_xor(q, 32);
  rstd = rsqrtf(q / (float)DIM + kLnEps);
}

template <class S, int I>
__device__ __forceinline__ void sx_layer_fwd(const float* sm, float (&cur)[kMB][16], int lane, int hb) {
  constexpr int in = S::in(I), out = S::out(I);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (S::kind(I) == 0) {
    float mean, rstd;
    sx_ln_stats<in>(cur, hb, mean, rstd);
    const float* gt = sm + S::wf(I);
    constexpr int nb = S::nbi(I);
#pragma unroll
    for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * j < sx_ne(in, ib)) {
          const float4 g4 = *reinterpret_cast<const float4*>(gt + 32 * ib + 8 * j + 4 * hb);
          const float4 b4 = *reinterpret_cast<const float4*>(gt + nb * 32 + 32 * ib + 8 * j + 4 * hb);
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) cur[ib][4 * j + q] = fmaf((cur[ib][4 * j + q] - mean) * rstd, gv[q], bv[q]);
        }
  } else {
    constexpr int nbi = S::nbi(I), nbo = S::nbo(I), act = S::act(I);
    f32x16 acc[kMB];
    mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
      constexpr int ob = decltype(OB)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b4 = *reinterpret_cast<const float4*>(sm + S::tb(I) + 32 * ob + 8 * j + 4 * hb);
        acc[ob][4 * j] = b4.x; acc[ob][4 * j + 1] = b4.y; acc[ob][4 * j + 2] = b4.z; acc[ob][4 * j + 3] = b4.w;
      }
      mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
        constexpr int ib = decltype(IB)::value;
        mm_chain_n<sx_ne(in, ib), false>(acc[ob], sm + S::wf(I) + (ob * nbi + ib) * kFB + lane, cur[ib]);
      });
      // (one basic block per tile: without fences the scheduler lifts every fragment read of the chain to the top of it and the
      // register allocator spills what it lifted)
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int ob = 0; ob < kMB; ++ob)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (e < sx_ne(out, ob)) cur[ob][e] = act == 1 ? fmaxf(acc[ob][e], 0.f) : (act == 2 ? tanhf(acc[ob][e]) : acc[ob][e]);
  }
}

template <class S>
__global__ __launch_bounds__(256, 2) void mlp_fwd_sig_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  sx_stage<S>(a, sm, tid, 256);
  __syncthreads();
  const long ntiles = (a.rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool rok = row < a.rows;
    float cur[kMB][16];
    sx_load<S::in(0)>(a.x, a.ldx, row, rok, hb, cur);
    mm_static_for<S::n>([&](auto IC) __attribute__((always_inline)) { sx_layer_fwd<S, decltype(IC)::value>(sm, cur, lane, hb); });
    sx_store<S::out_dim>(a.y, a.ldy, row, rok, hb, cur);
  }
}

// ---- backward: mlp_bwd_mfma_kernel with the shape known ---------------------------------------------------------------------------
template <int NE>
__device__ __forceinline__ void sx_half_write(float* T, int r, int hb, const float (&v)[16]) {   // (columns beyond the width: whatever
  // the tile held -- they only ever meet accumulator rows / columns that are not exported)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (4 * j < NE)
      *reinterpret_cast<float4*>(T + r * kTh + 8 * j + 4 * hb) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

// DX: d loss / d x is formed and stored as well (chains behind a recurrent layer)
template <class S, bool DX>
__global__ __launch_bounds__(64 * kBwdWaves, 1) void mlp_bwd_sig_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  constexpr int NL = S::n;
  // LDS: parameters | per wavefront: per-channel sums [npg][64] | per wavefront: tile area (dz halves | x halves)
  float* const pgs = sm + S::par_floats + wave * (S::npg * 64);
  float* const tiles = sm + S::par_floats + kBwdWaves * (S::npg * 64);
  float* const myT = tiles + wave * kTileF;
  sx_stage<S>(a, sm, tid, 64 * kBwdWaves);
  for (int e = lane; e < S::npg * 64; e += 64) pgs[e] = 0.f;
  // (garbage in the tile area must at least be finite where a column sum of exported channels could meet it: it cannot -- every
  // exported column is rewritten before it is read -- but zeros cost nothing here)
  for (int e = lane; e < kTileF; e += 64) myT[e] = 0.f;
  __syncthreads();
  f32x16 Wb[S::nlin > 0 ? S::nlin : 1];   // this wavefront's block of every Linear layer
#pragma unroll
  for (int k = 0; k < (S::nlin > 0 ? S::nlin : 1); ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) Wb[k][e] = 0.f;
  const int trow = ((r & 3) + 4 * (r >> 3)) * kFP + ((r >> 2) & 1) * 32 + 4 * hb;   // this lane's row of a block read transposed
  const long ntiles = (a.rows + 31) / 32;
  for (long tile0 = (long)blockIdx.x * kBwdWaves; tile0 < ntiles; tile0 += (long)gridDim.x * kBwdWaves) {
    const long row = (tile0 + wave) * 32 + r;   // (a tile beyond the rows: zeros all the way, its wavefront keeps the barriers' count)
    const bool rok = row < a.rows && !(a.dbg & 64);
    float d[kMB][16], xs[NL][kMB][16];
    sx_load<S::out_dim>(a.dy, a.lddy, row, rok, hb, d);   // (first: it arrives under the forward walk)
    {
      float cur[kMB][16];
      sx_load<S::in(0)>(a.x, a.ldx, row, rok, hb, cur);
      // the chain forward again, every layer's input kept (the last layer's output is not needed)
      mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
        constexpr int i = decltype(IC)::value;
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (e < sx_ne(S::in(i), ib)) xs[i][ib][e] = cur[ib][e];
        if constexpr (i + 1 < NL) {
          if (!(a.dbg & 128)) sx_layer_fwd<S, i>(sm, cur, lane, hb);
        }
      });
    }
    mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
      constexpr int i = NL - 1 - decltype(IC)::value;
      constexpr int in = S::in(i), out = S::out(i);
      float (&xin)[kMB][16] = xs[i];
      // the activation that produced this input: its derivative (from the input's value) closes the data gradient
      constexpr int pact = (i > 0 && S::kind(i > 0 ? i - 1 : 0) == 1) ? S::act(i > 0 ? i - 1 : 0) : 0;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (S::kind(i) == 1) {
        constexpr int nbi = S::nbi(i), nbo = S::nbo(i), lin = S::lin(i);
        mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
          constexpr int ob = decltype(OB)::value;
          sx_half_write<sx_ne(out, ob)>(myT + ob * 32 * kTh, r, hb, d[ob]);
          if (a.gb[i] && !(a.dbg & 16)) mm_half_colsum(myT + ob * 32 * kTh, pgs + S::pg(i) * 64 + 32 * ob, lane);  // bias gradient: column sums of dz
        });
        mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
          constexpr int ib = decltype(IB)::value;
          sx_half_write<sx_ne(in, ib)>(myT + (kMB + ib) * 32 * kTh, r, hb, xin[ib]);
        });
        __syncthreads();
        if (!(a.dbg & 2)) {
          // this wavefront's block of the layer, over its share of the four tiles: 4 blocks -> every tile; 2 -> two tiles; 1 -> its own
          constexpr int nblk = nbo * nbi, per = nblk >= kBwdWaves ? kBwdWaves : nblk;
          const int b = wave % nblk, ob = b / nbi, ib = b - ob * nbi, t0 = (wave / nblk) * per;
          mm_wgrad_tiles<per>(Wb[lin], tiles, t0, ob, ib, lane);
        }
        __syncthreads();   // the tile areas are rewritten by the next layer below
        if constexpr (i > 0 || DX) {
          f32x16 acc[kMB];
          mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
            constexpr int ib = decltype(IB)::value;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ib][e] = 0.f;
            if (!(a.dbg & 4))
              mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
                constexpr int ob = decltype(OB)::value;
                mm_chain_n<sx_ne(out, ob), true>(acc[ib], sm + S::wf(i) + (ob * nbi + ib) * kFB + trow, d[ob]);
              });
            __builtin_amdgcn_sched_barrier(0);
          });
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (e < sx_ne(in, ib)) d[ib][e] = acc[ib][e] * act_der(xin[ib][e], pact);
        }
      } else {
        // LayerNorm: statistics recomputed from the input; dgamma / dbeta = column sums of gy * xhat / gy (own tile area: no barrier)
        float mean, rstd;
        sx_ln_stats<in>(xin, hb, mean, rstd);
        const float* gt = sm + S::wf(i);
        constexpr int nb = S::nbi(i);
        float gg[kMB][16];
        float m1 = 0.f, m2 = 0.f;
        mm_static_for<nb>([&](auto IB) __attribute__((always_inline)) {
            constexpr int ib = decltype(IB)::value;
            float gyx[16];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (4 * j < sx_ne(in, ib)) {
                const float4 g4 = *reinterpret_cast<const float4*>(gt + 32 * ib + 8 * j + 4 * hb);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int e = 4 * j + q;
                  const bool ok = sx_valid<in>(ib, e, hb);
                  const float xh = ok ? (xin[ib][e] - mean) * rstd : 0.f;
                  xin[ib][e] = ok ? xin[ib][e] : 0.f;
                  gyx[e] = d[ib][e] * xh;
                  gg[ib][e] = d[ib][e] * gv[q];
                  m1 += gg[ib][e];
                  m2 = fmaf(gg[ib][e], xh, m2);
                }
              }
            sx_half_write<sx_ne(in, ib)>(myT, r, hb, gyx);                 // dgamma's terms
            sx_half_write<sx_ne(in, ib)>(myT + 32 * kTh, r, hb, d[ib]);   // gy: its column sums are dbeta
            mm_half_colsum(myT, pgs + S::pg(i) * 64 + 32 * ib, lane);
            mm_half_colsum(myT + 32 * kTh, pgs + (S::pg(i) + 1) * 64 + 32 * ib, lane);
        });
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        m1 /= (float)in;
        m2 /= (float)in;
        if constexpr (i > 0 || DX) {
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (e < sx_ne(in, ib)) {
                const float xh = (xin[ib][e] - mean) * rstd;
                d[ib][e] = sx_valid<in>(ib, e, hb) ? rstd * (gg[ib][e] - m1 - xh * m2) * act_der(xin[ib][e], pact) : 0.f;
              }
        }
      }
    });
    if constexpr (DX) sx_store<S::in(0)>(a.dx, a.lddx, row, row < a.rows, hb, d);   // (the chain's input carries no activation)
  }
  // ---- the workgroup's sums meet in LDS (the tile region is free), then one atomic per parameter and workgroup ---------------
  __syncthreads();
  float* const accs = tiles;
#pragma unroll
  for (int k = 0; k < S::nlin; ++k) {
    float* slot = accs + (wave * kMaxLin + k) * 1024 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) slot[e * 64] = Wb[k][e];
  }
  __syncthreads();
  if (a.dbg & 8) return;
  mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr int in = S::in(i), out = S::out(i);
    auto psum = [&](int slot, int c) {   // the four wavefronts' per-channel sums
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kBwdWaves; ++w) s += sm[S::par_floats + w * (S::npg * 64) + slot * 64 + c];
      return s;
    };
    if constexpr (S::kind(i) == 1) {
      constexpr int nbi = S::nbi(i), nbo = S::nbo(i), nblk = nbo * nbi;
      // accumulator of dz^T x: rows = out channel (register e, half hb), columns = in channel (lane & 31)
      for (int idx = tid; idx < nblk * 1024; idx += 64 * kBwdWaves) {
        const int l = idx & 63, e = (idx >> 6) & 15, blk = idx >> 10, ob = blk / nbi, ib = blk - ob * nbi;
        const int o = 32 * ob + mm_ch(e, l >> 5), k = 32 * ib + (l & 31);
        if (o < out && k < in) {
          float v = 0.f;   // the wavefronts that formed this block: blk, blk + nblk, ...
          for (int w = blk; w < kBwdWaves; w += nblk) v += accs[(w * kMaxLin + S::lin(i)) * 1024 + (idx & 1023)];
          atomicAdd(a.gw[i] + o * in + k, v);
        }
      }
      if (a.gb[i])
        for (int c = tid; c < out; c += 64 * kBwdWaves) atomicAdd(a.gb[i] + c, psum(S::pg(i), c));
    } else {
      for (int c = tid; c < in; c += 64 * kBwdWaves) {
        atomicAdd(a.gw[i] + c, psum(S::pg(i), c));
        atomicAdd(a.gb[i] + c, psum(S::pg(i) + 1, c));
      }
    }
  });
}

template <class S>
constexpr long sx_bwd_lds_bytes() {
  const long tiles = (long)kBwdWaves * kTileF, accs = (long)kBwdWaves * kMaxLin * 1024;
  return 4L * (S::par_floats + kBwdWaves * S::npg * 64 + (tiles > accs ? tiles : accs));
}

inline void sx_args(const Args& a, XArgs& x) {
  x = XArgs{};
  x.x = a.x; x.ldx = a.ldx; x.rows = a.rows; x.y = a.y; x.ldy = a.ldy; x.dy = a.dy; x.lddy = a.lddy; x.dx = a.dx; x.lddx = a.lddx;
  for (int i = 0; i < a.n && i < kMaxNL; ++i) {
    x.w[i] = a.L[i].w; x.b[i] = a.L[i].b; x.gw[i] = a.L[i].gw; x.gb[i] = a.L[i].gb;
  }
}

template <class S>
bool sx_try_fwd(const Args& a, hipStream_t st) {
  if (!S::matches(a)) return false;
  static_assert(4L * S::par_floats <= 150 * 1024 && S::n <= kMaxNL && S::nlin <= kMaxLin, "chain does not fit");
  XArgs x;
  sx_args(a, x);
  const long tiles4 = srl_ceil_div(a.rows, 128L);
  constexpr int lds = 4 * S::par_floats;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_sig_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL(mlp_fwd_sig_kernel<S>, dim3((unsigned)(tiles4 < 768 ? tiles4 : 768)), dim3(256), lds, st, x);
  return true;
}
template <class S, bool DX>
void sx_launch_bwd(const XArgs& x, long rows, hipStream_t st) {
  const long groups = srl_ceil_div(rows, 32L * kBwdWaves);
  constexpr int lds = (int)sx_bwd_lds_bytes<S>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_sig_kernel<S, DX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((mlp_bwd_sig_kernel<S, DX>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kBwdWaves), lds, st, x);
}
// WITH_DX: whether the shape is instantiated with the input gradient too (the chains behind a recurrent layer; the others would
// only add compile time)
template <class S, bool WITH_DX>
bool sx_try_bwd(const Args& a, int dbg, hipStream_t st) {
  if (!S::matches(a) || (a.dx && !WITH_DX)) return false;
  static_assert(sx_bwd_lds_bytes<S>() <= 158 * 1024, "chain does not fit");
  XArgs x;
  sx_args(a, x);
  x.dbg = dbg;
  if constexpr (WITH_DX) {
    if (a.dx) {
      sx_launch_bwd<S, true>(x, a.rows, st);
      return true;
    }
  }
  sx_launch_bwd<S, false>(x, a.rows, st);
  return true;
}

// The instantiated shapes (SRL_MLP_SIG=0 keeps every chain on the generic kernels: A/B, parity tests):
//  * the separate actor / critic towers of the reference's vector-observation policy at BASELINE configs[0] (CartPole: 4
//    observations, 2 x 64 hidden, 2 actions / 1 value; actor_critic_policy.py:60-107 with modules/utils.py:154-161's LayerNorm ->
//    Linear -> ReLU -> LayerNorm base);
//  * the observation / state encoders in front of the recurrent cells of the multi-agent policy at BASELINE configs[3] (SMAC 3m:
//    30 local observations, 48 state features, 64 hidden).
using SigC1Actor = MSig<SRL_SIG_LN(4), SRL_SIG_LIN(4, 64, 1), SRL_SIG_LN(64), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LIN(64, 2, 0)>;
using SigC1Critic = MSig<SRL_SIG_LN(4), SRL_SIG_LIN(4, 64, 1), SRL_SIG_LN(64), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LIN(64, 1, 0)>;
using SigSmacObs = MSig<SRL_SIG_LN(30), SRL_SIG_LIN(30, 64, 1), SRL_SIG_LN(64), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LN(64)>;
using SigSmacState = MSig<SRL_SIG_LN(48), SRL_SIG_LIN(48, 64, 1), SRL_SIG_LN(64), SRL_SIG_LIN(64, 64, 1), SRL_SIG_LN(64)>;
//  * what follows that policy's recurrent cells: LayerNorm 64 and the head (9 actions on 3m / 1 value), with the gradient w.r.t. the
//    cell's output.
using SigSmacActorTail = MSig<SRL_SIG_LN(64), SRL_SIG_LIN(64, 9, 0)>;
using SigSmacCriticTail = MSig<SRL_SIG_LN(64), SRL_SIG_LIN(64, 1, 0)>;

template <class S, bool WITH_DX = false>
struct SigE {
  using sig = S;
  static constexpr bool dx = WITH_DX;
};
template <class... Es>
struct SigList {
  static bool fwd(const Args& a, hipStream_t st) { return (sx_try_fwd<typename Es::sig>(a, st) || ...); }
  static bool bwd(const Args& a, int dbg, hipStream_t st) { return (sx_try_bwd<typename Es::sig, Es::dx>(a, dbg, st) || ...); }
};
using Sigs = SigList<SigE<SigC1Actor>, SigE<SigC1Critic>, SigE<SigSmacObs>, SigE<SigSmacState>, SigE<SigSmacActorTail, true>,
                     SigE<SigSmacCriticTail, true>>;

inline bool sx_enabled() {
  static const bool v = [] { const char* e = getenv("SRL_MLP_SIG"); return !(e && e[0] == '0'); }();
  return v;
}
inline bool sx_fwd(const Args& a, hipStream_t st) { return sx_enabled() && Sigs::fwd(a, st); }
inline bool sx_bwd(const Args& a, int dbg, hipStream_t st) { return sx_enabled() && Sigs::bwd(a, dbg, st); }

}  // namespace
