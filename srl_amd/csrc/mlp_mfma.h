// The fused MLP chain of mlp_small.hip on the float32 matrix cores, for row counts that fill the chip (rounds 4-5).
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
//   * NO TAPE (round 5): x and y are the only traffic.  The tape -- every layer's input, 1 KB per row of a 2 x 64 chain, written
//     in 16-byte pieces -- was what the forward launch took its time for (545 MB, 257 us at 524 288 rows against 72 us of MFMAs)
//     and the backward launch waited on; the backward pass walks the chain forward again from x (16 bytes per row), every layer's
//     input staying in registers, for 136 more MFMAs per 32 rows.
// Backward: the data gradient is the same chain with the SAME fragments read transposed (rows of 65 floats: conflict-free both
// ways); the weight gradient dW[o][k] = sum_rows dz[r][o] x[r][k] reduces over ROWS, so both operands go through a 32-row
// transposition tile in LDS (rows become the k dimension) and the products accumulate in persistent accumulator blocks -- one
// 32 x 32 block per (out block, in block) of every Linear, for all the rows a workgroup walks -- that meet in LDS once, at the end.
// Bias / LayerNorm-affine gradients are column sums of the same tiles, kept per lane in LDS.
// Eligible chains: widths <= 64 (two 32-channel blocks), <= 8 layers of which <= 4 Linear; the others keep mlp_small.hip's kernels.
// Reference: modules/utils.py:154-161 (mlp), actor_critic_policy.py:92-107 (heads).
#pragma once

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMB = 2;          // 32-channel blocks per side of a layer at most
constexpr int kMaxAcc = 12;     // dW accumulator blocks of a chain at most
constexpr int kMaxNL = 8;       // layers of an eligible chain at most (their inputs stay in registers through the backward pass)
constexpr int kFP = 65;         // floats per fragment row: 64 lanes + 1, so that the data gradient's transposed reads hit 32 banks
constexpr int kFB = 16 * kFP;   // floats per 32 x 32 weight block

struct MArgs {
  Args a;
  int wf[SRL_MLP_MAX_LAYERS];    // LDS float offset: Linear fragments [nbo][nbi][16][kFP]; LayerNorm: gamma | beta tables
  int tb[SRL_MLP_MAX_LAYERS];    // Linear: bias table [nbo * 32]
  int pg[SRL_MLP_MAX_LAYERS];    // backward: offset of the layer's per-lane sums (Linear: bias gradient; LayerNorm: dgamma | dbeta)
  int par_floats, nacc, npg, nlin;
  int dbg;   // timing experiments (wrong results; SRL_MLP_DBG): 2 no weight-gradient blocks, 4 no data gradient, 8 no final global adds, 16 no column sums, 32 no parameter staging, 64 no global loads, 128 no forward walk
};

__device__ __forceinline__ int mm_ch(int e, int hb) { return (e & 3) + 8 * (e >> 2) + 4 * hb; }
// how many of a block's 16 k-pairs (register e of both half-lanes) hold channels below `dim`: registers 4g .. 4g + 3 carry channels
// 8g .. 8g + 7, so a 4-wide input needs 4 of the 16 MFMAs of its block (the others multiply padding zeros)
__device__ __forceinline__ int mm_ne(int dim, int blk) {
  const int left = dim - 32 * blk;
  return left >= 32 ? 16 : (left <= 0 ? 0 : 4 * ((left + 7) >> 3));
}

// NE MFMAs of one (out block, in block) product as straight-line code (a run-time bound inside the unrolled loop put every MFMA into
// a basic block of its own: the operand reads could no longer run ahead, fwd 83 -> 170 us).  TR: the block read transposed (the
// data gradient): `wfr` then points at this lane's row of the block and register e sits at column mm_ch(e, 0).
template <int NE, bool TR>
__device__ __forceinline__ void mm_chain_n(f32x16& acc, const float* wfr, const float (&v)[16]) {
#pragma unroll
  for (int e = 0; e < NE; ++e)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(TR ? wfr[(e & 3) + 8 * (e >> 2)] : wfr[kFP * e], v[e], acc, 0, 0, 0);
}
template <bool TR>
__device__ __forceinline__ void mm_chain(f32x16& acc, const float* wfr, const float (&v)[16], int ne) {
  if (ne == 16) mm_chain_n<16, TR>(acc, wfr, v);
  else if (ne == 4) mm_chain_n<4, TR>(acc, wfr, v);
  else if (ne == 8) mm_chain_n<8, TR>(acc, wfr, v);
  else if (ne == 12) mm_chain_n<12, TR>(acc, wfr, v);
}

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

// stage the chain's parameters into LDS in operand order (every thread of the workgroup); ONE copy serves both directions
__device__ __forceinline__ void mm_stage(const MArgs& m, float* sm, int tid, int nthr) {
  for (int i = 0; i < m.a.n; ++i) {
    const Layer L = m.a.L[i];
    if (L.kind == 1) {
      const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
      for (int e = tid; e < nbo * nbi * 1024; e += nthr) {
        const int l = e & 63, e16 = (e >> 6) & 15, blk = e >> 10, ib = blk % nbi, ob = blk / nbi;
        const int o = 32 * ob + (l & 31), k = 32 * ib + mm_ch(e16, l >> 5);
        sm[m.wf[i] + blk * kFB + e16 * kFP + l] = (o < L.out && k < L.in) ? L.w[o * L.in + k] : 0.f;
      }
      for (int c = tid; c < nbo * 32; c += nthr) sm[m.tb[i] + c] = (c < L.out && L.b) ? L.b[c] : 0.f;
    } else {
      const int nb = (L.in + 31) >> 5;
      for (int c = tid; c < nb * 32; c += nthr) {
        sm[m.wf[i] + c] = c < L.in ? L.w[c] : 0.f;
        sm[m.wf[i] + nb * 32 + c] = c < L.in ? L.b[c] : 0.f;
      }
    }
  }
}

// layer i applied to the rows held in `cur` (accumulator layout in, accumulator layout out)
__device__ __forceinline__ void mm_layer_fwd(const MArgs& m, const float* sm, int i, float (&cur)[kMB][16], int lane, int hb) {
  const Layer L = m.a.L[i];
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
    return;
  }
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
        if (ib < nbi) mm_chain<false>(acc[ob], sm + m.wf[i] + (ob * nbi + ib) * kFB + lane, cur[ib], mm_ne(L.in, ib));
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
}

__global__ __launch_bounds__(256) void mlp_fwd_mfma_kernel(MArgs m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const Args& a = m.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  mm_stage(m, sm, tid, 256);
  __syncthreads();
  const long ntiles = (a.rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool rok = row < a.rows;
    float cur[kMB][16];
    mm_load(a.x, a.ldx, row, rok, a.L[0].in, hb, cur);
    for (int i = 0; i < a.n; ++i) mm_layer_fwd(m, sm, i, cur, lane, hb);
    const Layer& last = a.L[a.n - 1];
    mm_store(a.y, a.ldy, row, rok, last.kind == 1 ? last.out : last.in, hb, cur);
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------------------
// Round 4 kept one persistent 32 x 32 accumulator block per (out block, in block) of every Linear in REGISTERS, per wavefront: 12
// blocks = 192 accumulator registers, 512 registers per wavefront, 184 bytes of scratch per lane -- 358 us per 131 072 rows of a
// chain whose MFMAs take ~40.  First attempt of round 5: the blocks in LDS, shared by the workgroup, every wavefront adding its
// tile's 32 x 32 contribution with ds_add_f32 -- no scratch, two wavefronts per SIMD, and SLOWER where it mattered: a ds_add_f32
// wave-instruction takes ~160 cycles (leave-out: 229 us with, 90 without the adds on a 4-64-64-2 chain).  Now the blocks are OWNED:
// the four wavefronts of a workgroup walk four tiles side by side; at a Linear layer each writes its tile's dz and x halves to
// LDS, and after a barrier wavefront w forms block (w mod blocks) of that layer over the tiles of ALL four (or, with 2 / 1 blocks,
// over its share of them) -- so a wavefront accumulates ONE block per Linear layer, 16 registers each, for the whole launch, and
// the sums meet in LDS once, at the end.
constexpr int kTh = 36;         // floats per row of a half tile (32 + 4: conflict-free writes and transposed reads)
constexpr int kBwdWaves = 4;
constexpr int kMaxLin = 4;      // Linear layers of an eligible chain at most (one accumulator block per wavefront and layer)
constexpr int kTileF = 2 * kMB * 32 * kTh;   // floats of a wavefront's tile area: dz halves | x halves

__device__ __forceinline__ void mm_half_write(float* T, int r, int hb, const float (&v)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(T + r * kTh + 8 * j + 4 * hb) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

// column sums of a half tile into this wavefront's per-channel sums: lane = (channel c, row half)
__device__ __forceinline__ void mm_half_colsum(const float* T, float* dst, int lane) {
  const int c = lane & 31, r0 = (lane >> 5) * 16;
  float s = 0.f;
#pragma unroll 8
  for (int rr = 0; rr < 16; ++rr) s += T[(r0 + rr) * kTh + c];
  s += __shfl_xor(s, 32);
  if (lane < 32) dst[c] += s;
}

// NT tiles' contributions to a 32 x 32 block of a weight gradient, as one straight line of 16 NT MFMAs (the operand reads of the
// next tile run under the MFMAs of this one): the rows of the two half tiles are the k dimension
template <int NT>
__device__ __forceinline__ void mm_wgrad_tiles(f32x16& acc, const float* tiles, int t0, int ob, int ib, int lane) {
  const float* ap = tiles + t0 * kTileF + ob * 32 * kTh + (lane >> 5) * kTh + (lane & 31);
  const float* bp = tiles + t0 * kTileF + (kMB + ib) * 32 * kTh + (lane >> 5) * kTh + (lane & 31);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mm = 0; mm < 16; ++mm)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[t * kTileF + 2 * mm * kTh], bp[t * kTileF + 2 * mm * kTh], acc, 0, 0, 0);
}
__device__ __forceinline__ void mm_wgrad_block(f32x16& acc, const float* tiles, int per, int t0, int ob, int ib, int lane) {
  if (per == 4) mm_wgrad_tiles<4>(acc, tiles, t0, ob, ib, lane);
  else if (per == 2) mm_wgrad_tiles<2>(acc, tiles, t0, ob, ib, lane);
  else mm_wgrad_tiles<1>(acc, tiles, t0, ob, ib, lane);
}

#ifdef SRL_MLP_PROF   // variant builds only (scripts/mlp_prof.sh): shader-clock stamps of workgroup phases, read back through mlp_prof_dump
__device__ long long g_mlp_prof[256 * 40];
#define SRL_MLP_STAMP(k) do { if (tid == 0 && blockIdx.x < 256) g_mlp_prof[blockIdx.x * 40 + (k)] = clock64(); } while (0)
#define SRL_MLP_LSTAMP(k) do { if (it_ == 2) SRL_MLP_STAMP(20 + 5 * lin + (k)); } while (0)
#else
#define SRL_MLP_STAMP(k) do { } while (0)
#define SRL_MLP_LSTAMP(k) do { } while (0)
#endif

template <int I> struct mm_ic { static constexpr int value = I; };
// f(mm_ic<0>), f(mm_ic<1>), ... as separate inlined calls: the layer index is a constant in every copy of the body from the
// front end on (an unrolled loop left the per-layer register arrays in scratch)
template <int N, int I = 0, class F>
__device__ __forceinline__ void mm_static_for(F&& f) {
  if constexpr (I < N) {
    f(mm_ic<I>{});
    mm_static_for<N, I + 1>(f);
  }
}

// NL: the chain's layer count rounded up (4 / 6 / 8) -- the layer loops are unrolled so that every layer's input has registers of
// its own (`xs`), 32 per layer
template <int NL>
__global__ __launch_bounds__(64 * kBwdWaves, 1) void mlp_bwd_mfma_kernel(MArgs m) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const Args& a = m.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  // LDS: parameters (m.par_floats) | per wavefront: per-channel sums [npg][64] | per wavefront: tile area (dz halves | x halves)
  float* const pgs = sm + m.par_floats + wave * (m.npg * 64);
  float* const tiles = sm + m.par_floats + kBwdWaves * (m.npg * 64);
  float* const myT = tiles + wave * kTileF;
  SRL_MLP_STAMP(0);
  if (!(m.dbg & 32)) mm_stage(m, sm, tid, 64 * kBwdWaves);
  for (int e = lane; e < m.npg * 64; e += 64) pgs[e] = 0.f;
  __syncthreads();
  f32x16 W0, W1, W2, W3;   // this wavefront's block of the chain's 1st .. 4th Linear layer (counted from the input side)
#pragma unroll
  for (int e = 0; e < 16; ++e) W0[e] = W1[e] = W2[e] = W3[e] = 0.f;
  // this lane's row of a weight block read transposed: lane = in channel k, its row of the fragment is e16(k), half hb(k)
  const int trow = ((r & 3) + 4 * (r >> 3)) * kFP + ((r >> 2) & 1) * 32 + 4 * hb;
  const long ntiles = (a.rows + 31) / 32;
  SRL_MLP_STAMP(1);
  int it_ = 0;
  for (long tile0 = (long)blockIdx.x * kBwdWaves; tile0 < ntiles; tile0 += (long)gridDim.x * kBwdWaves) {
    if (it_ < 16) SRL_MLP_STAMP(8 + it_);
    ++it_;
    const long row = (tile0 + wave) * 32 + r;   // (a tile beyond the rows: zeros all the way, its wavefront keeps the barriers' count)
    const bool rok = row < a.rows && !(m.dbg & 64);
    float d[kMB][16], xs[NL][kMB][16];
    {
      const Layer& last = a.L[a.n - 1];
      mm_load(a.dy, a.lddy, row, rok, last.kind == 1 ? last.out : last.in, hb, d);   // (first: it arrives under the forward walk)
      float cur[kMB][16];
      mm_load(a.x, a.ldx, row, rok, a.L[0].in, hb, cur);
      // the chain forward again, every layer's input kept (the last layer's output is not needed)
      mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
        constexpr int i = decltype(IC)::value;
        if (i < a.n) {
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) xs[i][ib][e] = cur[ib][e];
          if (i + 1 < a.n && !(m.dbg & 128)) mm_layer_fwd(m, sm, i, cur, lane, hb);
        }
      });
    }
    int lin = m.nlin;
    mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
      constexpr int i = NL - 1 - decltype(IC)::value;
      if (i >= a.n) return;
      const Layer L = a.L[i];
      float (&xin)[kMB][16] = xs[i];
      // the activation that produced this input: its derivative (from the input's value) closes the data gradient
      const int pact = (i > 0 && a.L[i > 0 ? i - 1 : 0].kind == 1) ? a.L[i > 0 ? i - 1 : 0].act : 0;
      if (L.kind == 1) {
        --lin;
        const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
        SRL_MLP_LSTAMP(0);
#pragma unroll
        for (int ob = 0; ob < kMB; ++ob)
          if (ob < nbo) {
            mm_half_write(myT + ob * 32 * kTh, r, hb, d[ob]);
            if (L.gb && !(m.dbg & 16)) mm_half_colsum(myT + ob * 32 * kTh, pgs + m.pg[i] * 64 + 32 * ob, lane);  // bias gradient: column sums of dz
          }
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib)
          if (ib < nbi) mm_half_write(myT + (kMB + ib) * 32 * kTh, r, hb, xin[ib]);
        SRL_MLP_LSTAMP(1);
        __syncthreads();
        SRL_MLP_LSTAMP(2);
        if (!(m.dbg & 2)) {
          // this wavefront's block of the layer, over its share of the four tiles: 4 blocks -> every tile; 2 -> two tiles; 1 -> its own
          const int nblk = nbo * nbi, b = wave % nblk, ob = b / nbi, ib = b - ob * nbi;
          const int per = nblk >= kBwdWaves ? kBwdWaves : nblk, t0 = (wave / nblk) * per;
          switch (lin) {
            case 0: mm_wgrad_block(W0, tiles, per, t0, ob, ib, lane); break;
            case 1: mm_wgrad_block(W1, tiles, per, t0, ob, ib, lane); break;
            case 2: mm_wgrad_block(W2, tiles, per, t0, ob, ib, lane); break;
            default: mm_wgrad_block(W3, tiles, per, t0, ob, ib, lane); break;
          }
        }
        SRL_MLP_LSTAMP(3);
        __syncthreads();   // the tile areas are rewritten by the next layer below
        SRL_MLP_LSTAMP(4);
        if (i > 0 || a.dx) {
          f32x16 acc[kMB];
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ib][e] = 0.f;
            if (ib < nbi && !(m.dbg & 4)) {
#pragma unroll
              for (int ob = 0; ob < kMB; ++ob)
                if (ob < nbo) mm_chain<true>(acc[ib], sm + m.wf[i] + (ob * nbi + ib) * kFB + trow, d[ob], mm_ne(L.out, ob));
            }
          }
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) d[ib][e] = acc[ib][e] * act_der(xin[ib][e], pact);
        }
      } else {
        // LayerNorm: statistics recomputed from the input; dgamma / dbeta = column sums of gy * xhat / gy (own tile area: no barrier)
        float mean, rstd;
        mm_ln_stats(xin, L.in, hb, mean, rstd);
        const float* gt = sm + m.wf[i];
        const int nb = (L.in + 31) >> 5;
        float gg[kMB][16];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib) {
          float gyx[16];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ib < nb) g4 = *reinterpret_cast<const float4*>(gt + 32 * ib + 8 * j + 4 * hb);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int e = 4 * j + q;
              const bool in = 32 * ib + mm_ch(e, hb) < L.in;
              const float xh = in ? (xin[ib][e] - mean) * rstd : 0.f;
              xin[ib][e] = in ? xin[ib][e] : 0.f;
              gyx[e] = d[ib][e] * xh;
              gg[ib][e] = d[ib][e] * gv[q];
              m1 += gg[ib][e];
              m2 = fmaf(gg[ib][e], xh, m2);
            }
          }
          if (ib < nb) {
            mm_half_write(myT, r, hb, gyx);                     // dgamma's terms
            mm_half_write(myT + 32 * kTh, r, hb, d[ib]);       // gy: its column sums are dbeta
            mm_half_colsum(myT, pgs + m.pg[i] * 64 + 32 * ib, lane);
            mm_half_colsum(myT + 32 * kTh, pgs + (m.pg[i] + 1) * 64 + 32 * ib, lane);
          }
        }
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        m1 /= (float)L.in;
        m2 /= (float)L.in;
        if (i > 0 || a.dx) {
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
    });
    if (a.dx) mm_store(a.dx, a.lddx, row, row < a.rows, a.L[0].in, hb, d);   // d loss / d x (the chain's input carries no activation)
  }
  // ---- the workgroup's sums meet in LDS (the tile region is free), then one atomic per parameter and workgroup ---------------
  __syncthreads();
  SRL_MLP_STAMP(2);
  int lin_of[SRL_MLP_MAX_LAYERS];
  {
    int k = 0;
    for (int i = 0; i < a.n; ++i) lin_of[i] = a.L[i].kind == 1 ? k++ : -1;
  }
  // every wavefront parks its blocks in a slot of its own [wave][Linear layer][16][64] (plain stores: a ds_add_f32 wave-instruction
  // took ~800 cycles here, 19 us of a 142 us launch for 48 of them per wavefront); the sums over the wavefronts that shared a block
  // are formed by the threads that send them to the gradient buffer
  float* const accs = tiles;
  {
#define SRL_MM_PARK(Wk, k)                                                                    \
  if ((k) < m.nlin) {                                                                          \
    float* slot = accs + (wave * kMaxLin + (k)) * 1024 + lane;                                \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) slot[e * 64] = Wk[e];                       \
  }
    SRL_MM_PARK(W0, 0) SRL_MM_PARK(W1, 1) SRL_MM_PARK(W2, 2) SRL_MM_PARK(W3, 3)
#undef SRL_MM_PARK
  }
  __syncthreads();
  SRL_MLP_STAMP(3);
  for (int i = 0; i < ((m.dbg & 8) ? 0 : a.n); ++i) {
    const Layer L = a.L[i];
    auto psum = [&](int slot, int c) {   // the four wavefronts' per-channel sums
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kBwdWaves; ++w) s += sm[m.par_floats + w * (m.npg * 64) + slot * 64 + c];
      return s;
    };
    if (L.kind == 1) {
      const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
      // accumulator of dz^T x: rows = out channel (register e, half hb), columns = in channel (lane & 31)
      const int nblk = nbo * nbi;
      for (int idx = tid; idx < nblk * 1024; idx += 64 * kBwdWaves) {
        const int l = idx & 63, e = (idx >> 6) & 15, blk = idx >> 10, ob = blk / nbi, ib = blk - ob * nbi;
        const int o = 32 * ob + mm_ch(e, l >> 5), k = 32 * ib + (l & 31);
        if (o < L.out && k < L.in) {
          float v = 0.f;   // the wavefronts that formed this block: blk, blk + nblk, ...
          for (int w = blk; w < kBwdWaves; w += nblk) v += accs[(w * kMaxLin + lin_of[i]) * 1024 + (idx & 1023)];
          atomicAdd(L.gw + o * L.in + k, v);
        }
      }
      if (L.gb)
        for (int c = tid; c < L.out; c += 64 * kBwdWaves) atomicAdd(L.gb + c, psum(m.pg[i], c));
    } else {
      for (int c = tid; c < L.in; c += 64 * kBwdWaves) {
        atomicAdd(L.gw + c, psum(m.pg[i], c));
        atomicAdd(L.gb + c, psum(m.pg[i] + 1, c));
      }
    }
  }
  SRL_MLP_STAMP(4);
}

// LDS of the backward kernel: parameters | per-channel sums and a tile area per wavefront (the accumulator blocks reuse the tile
// region at the end)
inline long mm_bwd_lds_bytes(const MArgs& m) {
  const long tiles = (long)kBwdWaves * kTileF, accs = (long)kBwdWaves * kMaxLin * 1024;
  return 4L * (m.par_floats + kBwdWaves * m.npg * 64 + (tiles > accs ? tiles : accs));
}

// LDS plan of a chain for the MFMA kernels; false: not eligible
inline bool mm_plan(MArgs& m) {
  const Args& a = m.a;
  if (a.n > kMaxNL) return false;
  int f = 0, nacc = 0, npg = 0, nlin = 0;
  for (int i = 0; i < a.n; ++i) {
    const Layer& L = a.L[i];
    if (L.in > 32 * kMB || L.out > 32 * kMB) return false;
    const int nbi = (L.in + 31) >> 5, nbo = (L.out + 31) >> 5;
    if (L.kind == 1) {
      m.wf[i] = f; f += nbo * nbi * kFB;
      f = (f + 3) & ~3;   // (the tables are read as float4)
      m.tb[i] = f; f += nbo * 32;
      nacc += nbo * nbi;
      m.pg[i] = npg; npg += 1;
      ++nlin;
    } else {
      m.wf[i] = f; f += 2 * nbi * 32;
      m.pg[i] = npg; npg += 2;
    }
  }
  m.par_floats = f; m.nacc = nacc; m.npg = npg; m.nlin = nlin;
  if (nacc > kMaxAcc) return false;
  return 4L * f <= 150 * 1024 && nlin <= kMaxLin && mm_bwd_lds_bytes(m) <= 158 * 1024;
}

}  // namespace
