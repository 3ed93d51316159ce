// The shape kernels of mlp_sig.h on the 16-bit matrix pipe (round 6): every Linear layer of the chain as THREE products of f16
// pieces (h1 h0' + h0 h1' + h0 h0', v_mfma_f32_32x32x16_f16) instead of sixteen float32 MFMAs per 32 x 32 x 32 block -- 96 matrix
// cycles where the float32 instructions take 512 (of the backward launch's 434 us, 212 were its float32 MFMAs).
//
// Same walk as mlp_sig.h -- a wavefront owns 32 rows and keeps them in registers through the chain, the accumulator layout is the
// next layer's operand layout -- with these differences:
//   * the weights sit in LDS as two f16 pieces under ONE power-of-two scale per layer (max |w| -> [2^13, 2^14)): per 32 x 32 block
//     and piece 32 rows (out channel) of 64 bytes = the block's 32 k-values as four 16-byte chunks (chunk 2 s + g holds what lane
//     half g supplies to k-step s: channels 16 s + 4 g + {0..3, 8..11}), chunk index XOR ((row >> 2) & 3).  The forward product reads
//     a fragment with one conflict-free ds_read_b128; the data gradient reads the SAME copy transposed (ds_read_b64_tr_b16: four
//     rows x 64 bytes per 32 lanes at a 64-byte pitch is every bank once), so one copy serves both directions as before;
//   * a layer's input rows are split in registers, under a power-of-two scale PER ROW (its largest |value| -> [2^13, 2^14): the
//     contraction runs over channels, so a row's scale leaves with the epilogue): two mixed-precision instructions per element;
//   * bias is added in the epilogue (the accumulator is in scaled units): y = acc / (scale_w scale_row) + b, one fused multiply-add.
// Error of a product against float32: the dropped h1 h1' term and the pieces' rounding, <= 2^-21 of |w| |x| per term (the same
// arithmetic as the Atari encoder's h2 kernels); parity tests as for mlp_sig.h.  SRL_MLP_F16=0: the float32 kernels (A/B).
// Reference: modules/utils.py:154-161 (mlp), actor_critic_policy.py:92-107 (heads).
#pragma once

namespace {

typedef _Float16 hx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hx_f16x2 __attribute__((ext_vector_type(2)));

constexpr int kHB = 4096;   // bytes of a 32 x 32 weight block: piece 0 | piece 1, 32 rows of 64 bytes each

template <class S>
struct HP {   // LDS plan in bytes
  static constexpr int wb(int i) {
    int f = 0;
    for (int k = 0; k < i; ++k) f += S::kind(k) == 1 ? S::nbo(k) * S::nbi(k) * kHB + S::nbo(k) * 32 * 4 : 2 * S::nbi(k) * 32 * 4;
    return f;
  }
  static constexpr int tb(int i) { return wb(i) + S::nbo(i) * S::nbi(i) * kHB; }   // Linear: bias table; LayerNorm: wb = gamma | beta
  static constexpr int par_bytes = wb(S::n);
  static constexpr int inv = par_bytes;              // [n] floats: 1 / the layer's weight scale
  static constexpr int mx = par_bytes + 4 * S::n;    // [n] ints: max |w| of the layer (staging)
  static constexpr int total = (par_bytes + 8 * S::n + 15) / 16 * 16;
};

constexpr int hx_steps(int dim, int blk) { return (sx_ne(dim, blk) + 7) / 8; }   // k-steps of 16 channels of a 32-channel block

// byte offset of element (row r, k-value kl) of a block's piece plane
__device__ __forceinline__ int hx_welem(int r, int kl) {
  const int s = kl >> 4, kk = kl & 15, run = kk >> 3, g = (kk >> 2) & 1, jj = kk & 3;
  return r * 64 + (((2 * s + g) ^ ((r >> 2) & 3)) << 4) + ((4 * run + jj) << 1);
}

__device__ __forceinline__ float hx_pow2_scale(float m) {   // m -> the power of two that takes it into [2^13, 2^14); 0 -> 1
  int e;
  (void)frexpf(m, &e);
  return m > 0.f ? ldexpf(1.f, 14 - e) : 1.f;
}

template <class S>
__device__ __forceinline__ void hx_stage(const XArgs& a, uint8_t* smb, int tid, int nthr) {
  for (int e = tid; e < HP<S>::total / 4; e += nthr) reinterpret_cast<uint32_t*>(smb)[e] = 0u;   // padding of every block and table
  __syncthreads();
  mm_static_for<S::n>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    if constexpr (S::kind(i) == 1) {
      float m = 0.f;
      for (int idx = tid; idx < S::out(i) * S::in(i); idx += nthr) m = fmaxf(m, fabsf(a.w[i][idx]));
      atomicMax(reinterpret_cast<int*>(smb + HP<S>::mx) + i, __float_as_int(m));
    }
  });
  __syncthreads();
  mm_static_for<S::n>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr int in = S::in(i), out = S::out(i);
    const float* w = a.w[i];
    const float* b = a.b[i];
    if constexpr (S::kind(i) == 1) {
      constexpr int nbi = S::nbi(i);
      const float sc = hx_pow2_scale(__int_as_float(reinterpret_cast<const int*>(smb + HP<S>::mx)[i]));
      for (int idx = tid; idx < out * in; idx += nthr) {
        const int o = idx / in, k = idx - o * in;
        const float v = w[idx] * sc;
        const _Float16 h0 = (_Float16)v;
        const _Float16 h1 = (_Float16)(v - (float)h0);
        uint8_t* blk = smb + HP<S>::wb(i) + ((o >> 5) * nbi + (k >> 5)) * kHB + hx_welem(o & 31, k & 31);
        *reinterpret_cast<_Float16*>(blk) = h0;
        *reinterpret_cast<_Float16*>(blk + 2048) = h1;
      }
      if (b)
        for (int c = tid; c < out; c += nthr) reinterpret_cast<float*>(smb + HP<S>::tb(i))[c] = b[c];
      if (tid == 0) reinterpret_cast<float*>(smb + HP<S>::inv)[i] = 1.f / sc;
    } else {
      constexpr int nb = S::nbi(i);
      float* t = reinterpret_cast<float*>(smb + HP<S>::wb(i));
      for (int c = tid; c < in; c += nthr) {
        t[c] = w[c];
        t[nb * 32 + c] = b[c];
      }
    }
  });
}

// this lane's row: max |v| over its channels of the layer's input (both half-lanes), as the power-of-two scale and its inverse
template <int DIM>
__device__ __forceinline__ void hx_row_scale(const float (&v)[kMB][16], float& sc, float& inv) {
  float m = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < sx_ne(DIM, ib)) m = fmaxf(m, fabsf(v[ib][e]));
  m = fmaxf(m, __shfl_xor(m, 32));
  int ex;
  (void)frexpf(m, &ex);
  ex = m > 0.f ? ex : 14;
  sc = ldexpf(1.f, 14 - ex);
  inv = ldexpf(1.f, ex - 14);
}

// registers 8 s .. 8 s + 7 of a block (this lane's 8 k-values of step s) as two f16 pieces of v * sc; registers at or beyond NE: zeros
template <int NE>
__device__ __forceinline__ void hx_split8(const float (&v)[16], int s, float sc, hx_f16x8& p0, hx_f16x8& p1) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int e = 8 * s + j;
    if (e < NE) {
      const _Float16 a0 = (_Float16)__builtin_fmaf(v[e], sc, 0.f);
      p0[j] = a0;
      p1[j] = (_Float16)__builtin_fmaf(v[e], sc, -(float)a0);
    } else {
      p0[j] = (_Float16)0.f;
      p1[j] = (_Float16)0.f;
    }
  }
}

__device__ __forceinline__ hx_f16x8 hx_frag(const uint8_t* p) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  return __builtin_bit_cast(hx_f16x8, q);
}

template <class S, int I>
__device__ __forceinline__ void hx_layer_fwd(const uint8_t* smb, float (&cur)[kMB][16], int lane, int hb) {
  constexpr int in = S::in(I), out = S::out(I);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (S::kind(I) == 0) {
    float mean, rstd;
    sx_ln_stats<in>(cur, hb, mean, rstd);
    const float* gt = reinterpret_cast<const float*>(smb + HP<S>::wb(I));
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
    float sc, inv;
    hx_row_scale<in>(cur, sc, inv);
    inv *= reinterpret_cast<const float*>(smb + HP<S>::inv)[I];
    hx_f16x8 b0[kMB][2], b1[kMB][2];
    mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
      constexpr int ib = decltype(IB)::value;
      mm_static_for<hx_steps(in, ib)>([&](auto SS) __attribute__((always_inline)) {
        constexpr int s = decltype(SS)::value;
        hx_split8<sx_ne(in, ib)>(cur[ib], s, sc, b0[ib][s], b1[ib][s]);
      });
    });
    const int r = lane & 31;
    const int frow = r * 64, fx = (r >> 2) & 3;
    f32x16 acc[kMB];
    mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
      constexpr int ob = decltype(OB)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ob][e] = 0.f;
      mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
        constexpr int ib = decltype(IB)::value;
        const uint8_t* blk = smb + HP<S>::wb(I) + (ob * nbi + ib) * kHB + frow;
        mm_static_for<hx_steps(in, ib)>([&](auto SS) __attribute__((always_inline)) {
          constexpr int s = decltype(SS)::value;
          const int ch = ((2 * s + hb) ^ fx) << 4;
          const hx_f16x8 a0 = hx_frag(blk + ch), a1 = hx_frag(blk + 2048 + ch);
          acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0[ib][s], acc[ob], 0, 0, 0);
          acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1[ib][s], acc[ob], 0, 0, 0);
          acc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0[ib][s], acc[ob], 0, 0, 0);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    const float* bt = reinterpret_cast<const float*>(smb + HP<S>::tb(I));
#pragma unroll
    for (int ob = 0; ob < kMB; ++ob)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * j < sx_ne(out, ob)) {
          const float4 b4 = *reinterpret_cast<const float4*>(bt + 32 * ob + 8 * j + 4 * hb);
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float y = fmaf(acc[ob][4 * j + q], inv, bv[q]);
            cur[ob][4 * j + q] = act == 1 ? fmaxf(y, 0.f) : (act == 2 ? tanhf(y) : y);
          }
        }
  }
}

template <class S>
__global__ __launch_bounds__(256, 2) void mlp_fwd_h_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  hx_stage<S>(a, smb, tid, 256);
  __syncthreads();
  const long ntiles = (a.rows + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool rok = row < a.rows;
    float cur[kMB][16];
    sx_load<S::in(0)>(a.x, a.ldx, row, rok, hb, cur);
    mm_static_for<S::n>([&](auto IC) __attribute__((always_inline)) { hx_layer_fwd<S, decltype(IC)::value>(smb, cur, lane, hb); });
    sx_store<S::out_dim>(a.y, a.ldy, row, rok, hb, cur);
  }
}

template <class S>
bool hx_try_fwd(const Args& a, hipStream_t st) {
  if (!S::matches(a)) return false;
  static_assert(HP<S>::total <= 76 * 1024 && S::n <= kMaxNL, "chain does not fit (two workgroups per CU)");
  XArgs x;
  sx_args(a, x);
  const long tiles4 = srl_ceil_div(a.rows, 128L);
  constexpr int lds = HP<S>::total;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_h_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL(mlp_fwd_h_kernel<S>, dim3((unsigned)(tiles4 < 768 ? tiles4 : 768)), dim3(256), lds, st, x);
  return true;
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
// mlp_bwd_sig_kernel's walk (four wavefronts, four tiles side by side, no tape: the chain forward again from x) with all three
// products of a Linear layer on f16 pieces:
//   * forward walk: hx_layer_fwd;
//   * dz (the gradient at the layer's output) is split ONCE per layer and tile under a power-of-two scale of the TILE (the largest
//     |dz| of the wavefront's 32 rows: the weight gradient sums over rows, so a row's own scale would not leave the sum), into
//     packed registers -- the B operand of the data gradient -- and into the wavefront's dz tile in LDS;
//   * data gradient: A = the layer's weights read transposed from the one copy in LDS (two ds_read_b64_tr_b16 per fragment);
//   * weight gradient: the dz and x tiles are rows of 320 bytes (64 channels x piece 0 | piece 1, + 64: four consecutive rows fall
//     into different 64-byte bank windows) with the 8-byte slot index XOR ((row >> 2) & 7) (the 32 rows of a write instruction hit
//     32 different bank pairs); both operands of dW[o][k] = sum_rows dz[row][o] x[row][k] come out through transposing reads.  A
//     tile's product is formed in a FRESH accumulator and folded into the wavefront's persistent block with the tile's scales (one
//     fused multiply-add per accumulator register and tile);
//   * bias gradient: column sums of dz from the weight gradient's own A fragments (v_dot2_f32_f16 against ones), by the wavefronts
//     whose block has in-block 0, kept in a register per layer.
constexpr int kHRow = 320;                 // bytes per row of a piece tile
constexpr int kHTile = 32 * kHRow;         // one operand's tile
constexpr int kHArea = 2 * kHTile;         // a wavefront's tile area: dz | x  (LayerNorm layers use it as two float32 half tiles)
static_assert(kHArea >= 2 * 32 * kTh * 4, "LayerNorm's half tiles fit");

typedef short hx_s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ hx_s16x4 hx_tr(const uint8_t* p) {
  typedef __attribute__((address_space(3))) hx_s16x4 lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}
__device__ __forceinline__ hx_f16x8 hx_tr2(const uint8_t* p0, const uint8_t* p1) {
  union { hx_s16x4 s[2]; hx_f16x8 v; } f;
  f.s[0] = hx_tr(p0);
  f.s[1] = hx_tr(p1);
  return f.v;
}

// largest value of v (>= 0) over the wavefront, in every lane's copy (DPP row shifts and broadcasts, then lane 63)
__device__ __forceinline__ float hx_wave_max(float v) {
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true)));   // row_shr:1
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true)));   // row_shr:2
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true)));   // row_shr:4
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true)));   // row_shr:8
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false)));  // row_bcast:15
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false)));  // row_bcast:31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int DIM>
__device__ __forceinline__ float hx_lane_max(const float (&v)[kMB][16]) {
  float m = 0.f;
#pragma unroll
  for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (e < sx_ne(DIM, ib)) m = fmaxf(m, fabsf(v[ib][e]));
  return m;
}

__device__ __forceinline__ void hx_pow2(float m, float& sc, float& inv) {   // wave-uniform m >= 0
  int ex;
  (void)frexpf(m, &ex);
  ex = m > 0.f ? ex : 14;
  sc = ldexpf(1.f, 14 - ex);
  inv = ldexpf(1.f, ex - 14);
}

// this lane's registers 8 s .. 8 s + 7 of a block, as pieces, into row r of a piece tile: two runs of four channels per piece
__device__ __forceinline__ void hx_tile_write(uint8_t* T, int r, int hb, int blk, int s, const hx_f16x8& p0, const hx_f16x8& p1) {
  const uint4 q0 = __builtin_bit_cast(uint4, p0), q1 = __builtin_bit_cast(uint4, p1);
  const int x = (r >> 2) & 7;
  uint8_t* row = T + r * kHRow;
  const int sa = 8 * blk + 4 * s + hb, sb = sa + 2;     // 8-byte slots of the two runs inside a piece's 128 bytes
  *reinterpret_cast<uint2*>(row + ((sa ^ x) << 3)) = make_uint2(q0.x, q0.y);
  *reinterpret_cast<uint2*>(row + ((sb ^ x) << 3)) = make_uint2(q0.z, q0.w);
  *reinterpret_cast<uint2*>(row + (((16 + sa) ^ x) << 3)) = make_uint2(q1.x, q1.y);
  *reinterpret_cast<uint2*>(row + (((16 + sb) ^ x) << 3)) = make_uint2(q1.z, q1.w);
}

__device__ __forceinline__ void hx_tile_read(const uint8_t* T, int r, int hb, int blk, int s, hx_f16x8& p0, hx_f16x8& p1) {
  const int x = (r >> 2) & 7;
  const uint8_t* row = T + r * kHRow;
  const int sa = 8 * blk + 4 * s + hb, sb = sa + 2;
  const uint2 a0 = *reinterpret_cast<const uint2*>(row + ((sa ^ x) << 3)), a1 = *reinterpret_cast<const uint2*>(row + ((sb ^ x) << 3));
  const uint2 b0 = *reinterpret_cast<const uint2*>(row + (((16 + sa) ^ x) << 3)), b1 = *reinterpret_cast<const uint2*>(row + (((16 + sb) ^ x) << 3));
  p0 = __builtin_bit_cast(hx_f16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
  p1 = __builtin_bit_cast(hx_f16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
}

// DX: d loss / d x is formed and stored as well (chains behind a recurrent layer)
template <class S, bool DX>
__global__ __launch_bounds__(64 * kBwdWaves, 1) void mlp_bwd_h_kernel(XArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smb[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  constexpr int NL = S::n;
  constexpr int NLIN = S::nlin > 0 ? S::nlin : 1;
  // LDS: parameters | per wavefront: per-channel sums [npg][64] | tile scales [waves][2] | per wavefront: tile area
  constexpr int PG0 = HP<S>::total, TS0 = PG0 + kBwdWaves * S::npg * 64 * 4, T0 = (TS0 + kBwdWaves * 8 + 15) / 16 * 16;
  float* const pgs = reinterpret_cast<float*>(smb + PG0) + wave * (S::npg * 64);
  float* const tsc = reinterpret_cast<float*>(smb + TS0);
  uint8_t* const tiles = smb + T0;
  uint8_t* const myT = tiles + wave * kHArea;
  hx_stage<S>(a, smb, tid, 64 * kBwdWaves);
  for (int e = lane; e < S::npg * 64; e += 64) pgs[e] = 0.f;
  for (int e = lane; e < kHArea / 4; e += 64) reinterpret_cast<uint32_t*>(myT)[e] = 0u;
  __syncthreads();
  f32x16 Wb[NLIN];      // this wavefront's block of every Linear layer
  float bsum[NLIN];     // ... and, for blocks with in-block 0, its rows' share of the bias gradient (lane = out channel, row half)
#pragma unroll
  for (int k = 0; k < NLIN; ++k) {
    bsum[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) Wb[k][e] = 0.f;
  }
  const hx_f16x2 ones2 = {(_Float16)1.f, (_Float16)1.f};
  const long ntiles = (a.rows + 31) / 32;
  for (long tile0 = (long)blockIdx.x * kBwdWaves; tile0 < ntiles; tile0 += (long)gridDim.x * kBwdWaves) {
    const long row = (tile0 + wave) * 32 + r;   // (a tile beyond the rows: zeros all the way, its wavefront keeps the barriers' count)
    const bool rok = row < a.rows && !(a.dbg & 64);
    // lane roles of the transposing reads: 16-lane group g = 2 kg + h, lane i of the group = 4 q + p.  From an OPAQUE copy of the
    // lane index, made per tile: the few dozen fragment addresses below are loop-invariant, and the compiler otherwise keeps every
    // one of them in a register across the tile loop (47 spilled registers; the same trap as h2gemmp.h's epilogue)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int tq = (lv >> 2) & 3, tp = lv & 3, th = (lv >> 4) & 1, tkg = lv >> 5;
    float d[kMB][16], xs[NL][kMB][16];
    sx_load<S::out_dim>(a.dy, a.lddy, row, rok, hb, d);
    {
      float cur[kMB][16];
      sx_load<S::in(0)>(a.x, a.ldx, row, rok, hb, cur);
      mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
        constexpr int i = decltype(IC)::value;
#pragma unroll
        for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (e < sx_ne(S::in(i), ib)) xs[i][ib][e] = cur[ib][e];
        if constexpr (i + 1 < NL) {
          if (!(a.dbg & 128)) hx_layer_fwd<S, i>(smb, cur, lv, lv >> 5);
        }
      });
    }
    mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
      constexpr int i = NL - 1 - decltype(IC)::value;
      constexpr int in = S::in(i), out = S::out(i);
      float (&xin)[kMB][16] = xs[i];
      constexpr int pact = (i > 0 && S::kind(i > 0 ? i - 1 : 0) == 1) ? S::act(i > 0 ? i - 1 : 0) : 0;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (S::kind(i) == 1) {
        constexpr int nbi = S::nbi(i), nbo = S::nbo(i), lin = S::lin(i);
        // ---- scales of the tile, dz and x as pieces: dz into registers (the data gradient's B operand) and both into the tiles
        float sdz, idz, sx, ix;
        hx_pow2(hx_wave_max(hx_lane_max<out>(d)), sdz, idz);
        hx_pow2(hx_wave_max(hx_lane_max<in>(xin)), sx, ix);
        mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
          constexpr int ob = decltype(OB)::value;
          mm_static_for<hx_steps(out, ob)>([&](auto SS) __attribute__((always_inline)) {
            constexpr int s = decltype(SS)::value;
            hx_f16x8 p0, p1;
            hx_split8<sx_ne(out, ob)>(d[ob], s, sdz, p0, p1);
            hx_tile_write(myT, lv & 31, lv >> 5, ob, s, p0, p1);
          });
        });
        mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
          constexpr int ib = decltype(IB)::value;
          mm_static_for<hx_steps(in, ib)>([&](auto SS) __attribute__((always_inline)) {
            constexpr int s = decltype(SS)::value;
            hx_f16x8 x0, x1;
            hx_split8<sx_ne(in, ib)>(xin[ib], s, sx, x0, x1);
            hx_tile_write(myT + kHTile, lv & 31, lv >> 5, ib, s, x0, x1);
          });
        });
        if (lane == 0) {
          tsc[2 * wave] = idz;
          tsc[2 * wave + 1] = idz * ix;
        }
        // ---- data gradient (before the barrier: it needs registers and the weights only)
        if constexpr (i > 0 || DX) {
          f32x16 acc[kMB];
          const float dinv = idz * reinterpret_cast<const float*>(smb + HP<S>::inv)[i];
          mm_static_for<nbi>([&](auto IB) __attribute__((always_inline)) {
            constexpr int ib = decltype(IB)::value;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ib][e] = 0.f;
            if (!(a.dbg & 4))
              mm_static_for<nbo>([&](auto OB) __attribute__((always_inline)) {
                constexpr int ob = decltype(OB)::value;
                const uint8_t* blk = smb + HP<S>::wb(i) + (ob * nbi + ib) * kHB;
                mm_static_for<hx_steps(out, ob)>([&](auto SS) __attribute__((always_inline)) {
                  constexpr int s = decltype(SS)::value;
                  // rows (out channels) 16 s + 4 kg + q and + 8; this lane's run: k-values 16 h + 4 p .. + 3 of the block
                  const int o0 = 16 * s + 4 * tkg + tq, o1 = o0 + 8;
                  const int c0 = (((2 * th + (tp & 1)) ^ ((o0 >> 2) & 3)) << 4) + ((tp >> 1) << 3);
                  const int c1 = (((2 * th + (tp & 1)) ^ ((o1 >> 2) & 3)) << 4) + ((tp >> 1) << 3);
                  const hx_f16x8 a0 = hx_tr2(blk + o0 * 64 + c0, blk + o1 * 64 + c1);
                  const hx_f16x8 a1 = hx_tr2(blk + 2048 + o0 * 64 + c0, blk + 2048 + o1 * 64 + c1);
                  // B: this lane's own 8 k-values of dz, back from its row of the tile (32 registers less than keeping them)
                  hx_f16x8 pd0, pd1;
                  hx_tile_read(myT, lv & 31, lv >> 5, ob, s, pd0, pd1);
                  acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, pd0, acc[ib], 0, 0, 0);
                  acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pd1, acc[ib], 0, 0, 0);
                  acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, pd0, acc[ib], 0, 0, 0);
                });
              });
            __builtin_amdgcn_sched_barrier(0);
          });
#pragma unroll
          for (int ib = 0; ib < kMB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (e < sx_ne(in, ib)) d[ib][e] = acc[ib][e] * dinv * act_der(xin[ib][e], pact);
        }
        __syncthreads();
        if (!(a.dbg & 2)) {
          // this wavefront's block of the layer, over its share of the four tiles: 4 blocks -> every tile; 2 -> two tiles; 1 -> its own
          constexpr int nblk = nbo * nbi, per = nblk >= kBwdWaves ? kBwdWaves : nblk;
          const int b = wave % nblk, ob = b / nbi, ib = b - ob * nbi, t0 = (wave / nblk) * per;
          const bool bias = a.gb[i] != nullptr && ib == 0 && !(a.dbg & 16);
          // fragment addresses inside a tile: rows 16 ts + 8 kg + q (+ 4), 8-byte slot 8 blk + 4 h + p of piece 0 (+ 16: piece 1)
#pragma unroll
          for (int t = 0; t < per; ++t) {
            const uint8_t* TA = tiles + (t0 + t) * kHArea;
            const uint8_t* TB = TA + kHTile;
            f32x16 D;
#pragma unroll
            for (int e = 0; e < 16; ++e) D[e] = 0.f;
            float bs = 0.f;
#pragma unroll
            for (int ts = 0; ts < 2; ++ts) {
              const int r0 = 16 * ts + 8 * tkg, x0 = (r0 >> 2) & 7, x1 = ((r0 + 4) >> 2) & 7;
              const int ra = (r0 + tq) * kHRow, rb = (r0 + 4 + tq) * kHRow;
              const int sa = 8 * ob + 4 * th + tp, sb = 8 * ib + 4 * th + tp;
              const hx_f16x8 a0 = hx_tr2(TA + ra + ((sa ^ x0) << 3), TA + rb + ((sa ^ x1) << 3));
              const hx_f16x8 a1 = hx_tr2(TA + ra + (((16 + sa) ^ x0) << 3), TA + rb + (((16 + sa) ^ x1) << 3));
              const hx_f16x8 b0 = hx_tr2(TB + ra + ((sb ^ x0) << 3), TB + rb + ((sb ^ x1) << 3));
              const hx_f16x8 b1 = hx_tr2(TB + ra + (((16 + sb) ^ x0) << 3), TB + rb + (((16 + sb) ^ x1) << 3));
              D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, D, 0, 0, 0);
              D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, D, 0, 0, 0);
              D = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, D, 0, 0, 0);
              if (bias) {
                const uint4 q0 = __builtin_bit_cast(uint4, a0), q1 = __builtin_bit_cast(uint4, a1);
                const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) bs = __builtin_amdgcn_fdot2(__builtin_bit_cast(hx_f16x2, w[k]), ones2, bs, false);
              }
            }
            const float it = tsc[2 * (t0 + t) + 1];
#pragma unroll
            for (int e = 0; e < 16; ++e) Wb[lin][e] = fmaf(D[e], it, Wb[lin][e]);
            if (bias) bsum[lin] = fmaf(bs, tsc[2 * (t0 + t)], bsum[lin]);
          }
        }
        __syncthreads();   // the tile areas are rewritten by the next layer below
      } else {
        // LayerNorm: statistics recomputed from the input; dgamma / dbeta = column sums of gy * xhat / gy (own tile area: no barrier)
        float* const fT = reinterpret_cast<float*>(myT);
        float mean, rstd;
        sx_ln_stats<in>(xin, hb, mean, rstd);
        const float* gt = reinterpret_cast<const float*>(smb + HP<S>::wb(i));
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
            sx_half_write<sx_ne(in, ib)>(fT, r, hb, gyx);                 // dgamma's terms
            sx_half_write<sx_ne(in, ib)>(fT + 32 * kTh, r, hb, d[ib]);   // gy: its column sums are dbeta
            mm_half_colsum(fT, pgs + S::pg(i) * 64 + 32 * ib, lane);
            mm_half_colsum(fT + 32 * kTh, pgs + (S::pg(i) + 1) * 64 + 32 * ib, lane);
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
  float* const accs = reinterpret_cast<float*>(tiles);
#pragma unroll
  for (int k = 0; k < S::nlin; ++k) {
    float* slot = accs + (wave * kMaxLin + k) * 1024 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) slot[e * 64] = Wb[k][e];
  }
  // bias gradients: lane (out channel, row half) of the wavefronts that formed a block with in-block 0
  mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    if constexpr (S::kind(i) == 1) {
      constexpr int nbi = S::nbi(i), nblk = S::nbo(i) * nbi, lin = S::lin(i);
      const int b = wave % nblk, ob = b / nbi, ib = b - ob * nbi;
      const float v = bsum[lin] + __shfl_xor(bsum[lin], 32);
      if (ib == 0 && lane < 32) pgs[S::pg(i) * 64 + 32 * ob + lane] = v;
    }
  });
  __syncthreads();
  if (a.dbg & 8) return;
  mm_static_for<NL>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value;
    constexpr int in = S::in(i), out = S::out(i);
    auto psum = [&](int slot, int c) {   // the four wavefronts' per-channel sums
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kBwdWaves; ++w) s += reinterpret_cast<const float*>(smb + PG0)[w * (S::npg * 64) + slot * 64 + c];
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
constexpr long hx_bwd_lds_bytes() {
  const long tiles = (long)kBwdWaves * kHArea, accs = 4L * kBwdWaves * kMaxLin * 1024;
  return (HP<S>::total + kBwdWaves * S::npg * 64 * 4 + kBwdWaves * 8 + 15) / 16 * 16 + (tiles > accs ? tiles : accs);
}

template <class S, bool DX>
void hx_launch_bwd(const XArgs& x, long rows, hipStream_t st) {
  const long groups = srl_ceil_div(rows, 32L * kBwdWaves);
  constexpr int lds = (int)hx_bwd_lds_bytes<S>();
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_h_kernel<S, DX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((mlp_bwd_h_kernel<S, DX>), dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kBwdWaves), lds, st, x);
}
template <class S, bool WITH_DX>
bool hx_try_bwd(const Args& a, int dbg, hipStream_t st) {
  if (!S::matches(a) || (a.dx && !WITH_DX)) return false;
  static_assert(hx_bwd_lds_bytes<S>() <= 158 * 1024, "chain does not fit");
  XArgs x;
  sx_args(a, x);
  x.dbg = dbg;
  if constexpr (WITH_DX) {
    if (a.dx) {
      hx_launch_bwd<S, true>(x, a.rows, st);
      return true;
    }
  }
  hx_launch_bwd<S, false>(x, a.rows, st);
  return true;
}

template <class... Es>
struct HSigList {
  static bool fwd(const Args& a, hipStream_t st) { return (hx_try_fwd<typename Es::sig>(a, st) || ...); }
  static bool bwd(const Args& a, int dbg, hipStream_t st) { return (hx_try_bwd<typename Es::sig, Es::dx>(a, dbg, st) || ...); }
};
using HSigs = HSigList<SigE<SigC1Actor>, SigE<SigC1Critic>, SigE<SigSmacObs>, SigE<SigSmacState>, SigE<SigSmacActorTail, true>,
                       SigE<SigSmacCriticTail, true>>;

inline int hx_mode() {   // SRL_MLP_F16: 0 float32 kernels; 1 (default) the f16-piece kernels
  static const int v = [] { const char* e = getenv("SRL_MLP_F16"); return e ? atoi(e) : 1; }();
  return v;
}
inline bool hx_fwd(const Args& a, hipStream_t st) { return sx_enabled() && hx_mode() != 0 && HSigs::fwd(a, st); }
inline bool hx_bwd(const Args& a, int dbg, hipStream_t st) { return sx_enabled() && (hx_mode() & 1) && !(hx_mode() & 2) && HSigs::bwd(a, dbg, st); }

}  // namespace
