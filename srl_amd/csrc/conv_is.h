// Image-stationary convolutions for small feature maps (the 20x20x32 and 9x9x64 layers of the Atari network).
//
// The implicit-GEMM kernels (gemm_bf16x3.h) gather every k-step of a tile's patch rows from global memory: an input pixel
// is requested once per tap that touches it (4x for 4x4 stride 2, 9x for 3x3 stride 1), in 64-byte half lines a k-step
// apart -- leave-out experiments showed those kernels held by the memory system below L1, not by the matrix pipe (22 % busy)
// or by latency.  Here a workgroup stages WHOLE images once:
//   * G consecutive images [G][H][W][C] float32 are read from HBM with plain contiguous float4 loads (each byte exactly once),
//     split on the way into two f16 planes (h0 = f16(x s), h1 = f16(x s - h0): gemm_bf16x3.h's NP == 2 arithmetic, the same
//     three products h0 h0' + h0 h1' + h1 h0') and kept in LDS, 4 bytes per element like the float32 they came from;
//   * every tap's A fragments are ds_read_b128 from those planes (rows = the G * OH * OW output pixels, 8 consecutive
//     channels per lane) -- no gather address arithmetic per k-step, no re-fetch;
//   * the weights are split ONCE per launch by a small kernel into the fragment order of the B operand and stream through
//     a double-buffered 8-16 KB LDS slice per tap (they stay in L2: every workgroup reads the same 128-147 KB);
//   * the next pass's images are loaded into REGISTERS while this pass computes (one workgroup per CU: 100 registers per
//     lane are there), so HBM streams continuously; one workgroup per CU, 4 wavefronts, each owning one 32-column half of
//     the 64 output channels and TMW 32-row blocks of the pass's rows.
// Per image the kernel moves H W C + OH OW Cout floats through HBM and nothing else: the layer becomes a streaming kernel.
#pragma once
#include "gemm_bf16x3.h"

namespace srlis {

using srlgemm::f16x8;
using srlgemm::f32x16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16 bytes in registers (a native vector: arrays of it stay in
                                                             // registers where arrays of HIP's uint4 struct went to scratch)

struct FwdArgs {
  const float* x;      // [n][H][W][C]
  const uint4* wq;     // [KH*KW][C/16][2 planes][64][16] f16: srl_is_prep_kernel
  const float* bias;   // [64] or NULL
  float* y;            // [n][OH][OW][64]
  uint32_t* y_mask;    // sign words of y (gemm_core.h GemmArgs::mask_out layout) or NULL
  float* y_absmax;     // folded max |y| or NULL
  const float* x_absmax;
  const float* w_absmax;
  long n;
  int act;
};

// timing experiments (scripts/build_variant.sh -DSRL_IS_DBG=<bits>; wrong results): 1 no in-loop image loads, 2 no MFMAs,
// 4 no fragment reads, 8 no output stores, 16 no in-loop weight loads, 32 no LDS writes in the loop
#ifndef SRL_IS_DBG
#define SRL_IS_DBG 0
#endif

#ifdef __HIPCC__
template <int... K, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, K...>, F&& f) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// weights [64][KH][KW][C] float32 -> wq[tap][cb][plane][co][16] f16 pieces of w * scale(w_absmax)
__global__ __launch_bounds__(256) void is_prep_kernel(const float* w, const float* w_absmax, _Float16* wq, int taps, int C) {
  const float sc = srlgemm::range_scale(w_absmax);
  const int total = 64 * taps * C;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int c = e % C, t = (e / C) % taps, co = e / (C * taps);
    const float v = w[e] * sc;
    const _Float16 h0 = (_Float16)v;
    const _Float16 h1 = (_Float16)(v - (float)h0);
    const long base = ((long)(t * (C / 16) + (c >> 4)) * 2) * 64 * 16 + co * 16 + (c & 15);
    wq[base] = h0;
    wq[base + 64 * 16] = h1;
  }
}

template <int H, int W, int C, int KH, int KW, int S>
struct IsGeom {
  static constexpr int OH = (H - KH) / S + 1, OW = (W - KW) / S + 1, P = OH * OW;
  static constexpr int CB = C / 16;            // 16-deep k-steps per tap
  static constexpr int CH = C / 8;             // 16-byte chunks (8 f16) per pixel and plane
  static constexpr int PIXB = C * 2;           // bytes per pixel and plane
  static constexpr int IMGB = H * W * PIXB;    // bytes per image and plane
  static_assert(C % 16 == 0 && (CH == 4 || CH == 8), "32 or 64 input channels");
  static_assert(S == 1 || (S == 2 && W % 2 == 0), "stride 1, or stride 2 on an even width");
  // staged pixel index: stride 2 keeps the even-x and the odd-x pixels of a row apart, so that the rows of an MFMA block
  // (consecutive ox) read consecutive staged pixels for every tap
  __device__ static __forceinline__ int pidx(int iy, int ix) {
    return S == 2 ? iy * W + (ix & 1) * (W / 2) + (ix >> 1) : iy * W + ix;
  }
  // 16-byte chunks of a pixel are XOR-swizzled by the pixel index: 8 consecutive staged pixels read the same chunk from 8
  // different 16-byte slots of the 128-byte bank line
  __device__ static __forceinline__ int swz(int p) { return CH == 8 ? (p & 7) : ((p >> 1) & 3); }
  __device__ static __forceinline__ int tapoff(int kh, int kw) {
    return S == 2 ? kh * W + (kw & 1) * (W / 2) + (kw >> 1) : kh * W + kw;
  }
};

// G images per pass, TMW 32-row blocks per wavefront (2 * TMW * 32 >= G * P).  Everything that comes from memory arrives
// through register rings D steps deep (a step = one weight slice of TPS taps, ~0.3-0.6 us of MFMAs; an L2 hit under load is
// 1-2 us, and with one wavefront per SIMD nothing else hides it -- with one step of lead this kernel was 2.4x SLOWER than
// the implicit GEMM):
//   * weight slice k waits in register slot k % D and is written to the other LDS slice buffer one step before its use;
//   * the NEXT pass's images arrive in pieces of 512 float4 (two per lane and step), split into the two f16 planes and written
//     into the OTHER image buffer D steps after their loads were issued -- so the images are double-buffered in LDS, the loads
//     of a pass are spread over the whole previous pass, and there is one barrier per step and nothing else.
// NT threads: NT / 64 wavefronts = (NT / 128 wavefront rows of TMW blocks) x (2 column halves).  With 4 wavefronts -- one per
// SIMD, which issues in order -- the MFMAs and everything else of a step simply added up (leaving the MFMAs out halved the
// kernel's time, leaving anything else out took off its own share): two wavefronts per SIMD fill each other's gaps.
template <int H, int W, int C, int KH, int KW, int S, int G, int TMW, int TPS, int D, int NT, bool MASK>
__global__ __launch_bounds__(NT, 1) void is_fwd_kernel(FwdArgs a) {
  using GE = IsGeom<H, W, C, KH, KW, S>;
  constexpr int P = GE::P, ROWS = G * P, CB = GE::CB, TAPS = KH * KW, NSTEP = TAPS / TPS, NS = TPS * CB;
  static_assert((NT / 128) * TMW * 32 >= ROWS, "the wavefront rows must cover the pass");
  static_assert(TAPS % TPS == 0 && NSTEP % D == 0, "slices divide the taps, the register rings divide the slices");
  constexpr int PQ = 2 * 256 / NT;                    // float4 of an image piece (512 of them) per thread
  static_assert(PQ * NT == 512, "256 or 512 threads");
  constexpr int PLANE = G * GE::IMGB;                 // bytes of one f16 plane of a pass's images
  constexpr int IBUF = 2 * PLANE;                     // one image buffer: both planes
  constexpr int BSL = TPS * CB * 2 * 64 * 32;         // bytes of one weight slice
  constexpr int QUADS = G * H * W * C / 4, NPIECE = (QUADS + 511) / 512, SHIFT = NSTEP - NPIECE;
  static_assert(NPIECE <= NSTEP, "a pass's images arrive within one pass");
  static_assert(2 * IBUF + 2 * BSL <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * IBUF + 2 * BSL];
  uint8_t* const ldb = lds + 2 * IBUF;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int nb = wave & 1, mrow = wave >> 1;
  const float sx = srlgemm::range_scale(a.x_absmax), sw = srlgemm::range_scale(a.w_absmax);
  const float inv = 1.f / (sx * sw);

  const long npass = (a.n + G - 1) / G;
  const long per = (npass + gridDim.x - 1) / gridDim.x;
  const long p0 = (long)blockIdx.x * per, p1 = p0 + per < npass ? p0 + per : npass;
  if (p0 >= p1) return;

  // this lane's A rows: block i covers pass rows (mrow * TMW + i) * 32 + l31 -> (g, oy, ox) -> staged pixel of tap (0, 0)
  int aoff[TMW], apix[TMW];
#pragma unroll
  for (int i = 0; i < TMW; ++i) {
    int r = (mrow * TMW + i) * 32 + l31;
    if (r >= ROWS) r = 0;  // padding rows: any valid address, the results are not stored
    const int g = r / P, p = r - g * P, oy = p / GE::OW, ox = p - oy * GE::OW;
    apix[i] = GE::pidx(S * oy, S * ox);
    aoff[i] = g * GE::IMGB;
  }

  // ---- image pieces: quads tid and tid + 256 of piece * 512 + (0 .. 511) of the pass's G H W C floats ---------------------
  float4 ist[D][PQ];
  // (every load is issued unconditionally, from a clamped address, and masked afterwards: with loads under branches the
  // compiler's wait-count bookkeeping falls back to "wait for everything", i.e. one step of lead instead of D)
  auto piece_load = [&](auto slot_c, long pass, int piece) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const bool exists = pass < p1;
    const long img0 = (exists ? pass : p0) * G;
    const float4* src = reinterpret_cast<const float4*>(a.x + img0 * (long)(H * W * C));
    const int lim = exists ? (int)(a.n - img0 < G ? a.n - img0 : (long)G) * (H * W * C / 4) : 0;  // quads that exist
#pragma unroll
    for (int j = 0; j < PQ; ++j) {
      const int q = piece * 512 + tid + NT * j;
      const bool ok = q < QUADS && q < lim;
      const float4 v = src[ok ? q : 0];
      ist[SL][j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto piece_store = [&](auto slot_c, int ibuf, int piece) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    uint8_t* dst = lds + ibuf * IBUF;
#pragma unroll
    for (int j = 0; j < PQ; ++j) {
      const int q = piece * 512 + tid + NT * j;
      if (q >= QUADS) continue;
      const int e = 4 * q;
      const int g = e / (H * W * C), rem = e - g * (H * W * C), pix = rem / C, c = rem - pix * C;
      const int iy = pix / W, ix = pix - iy * W;
      const int pp = GE::pidx(iy, ix);
      const int off = g * GE::IMGB + pp * GE::PIXB + 16 * ((c >> 3) ^ GE::swz(pp)) + (c & 7) * 2;
      uint2 pl[2];
      const float v[4] = {ist[SL][j].x, ist[SL][j].y, ist[SL][j].z, ist[SL][j].w};
      srlgemm::split2h_quad(v, sx, pl);
      *reinterpret_cast<uint2*>(dst + off) = pl[0];
      *reinterpret_cast<uint2*>(dst + PLANE + off) = pl[1];
    }
  };
  // ---- weight slices: BSL contiguous bytes of wq each; slice k waits in register slot k % D --------------------------------
  constexpr int BQ = BSL / 16 / NT;  // 16-byte quads per thread
  static_assert(BSL % (16 * NT) == 0, "weight slice");
  u32x4 bst[D][BQ];
  auto bload = [&](auto slot_c, int slice) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wq) + (long)slice * (BSL / 16);
#pragma unroll
    for (int j = 0; j < BQ; ++j) bst[SL][j] = src[tid + NT * j];
  };
  auto bstore = [&](auto slot_c, int buf) __attribute__((always_inline)) {
    constexpr int SL = decltype(slot_c)::value;
    u32x4* dst = reinterpret_cast<u32x4*>(ldb + buf * BSL);
#pragma unroll
    for (int j = 0; j < BQ; ++j) dst[tid + NT * j] = bst[SL][j];
  };

  // ---- prologue: the first pass's images and slice 0 straight into LDS; the rings primed as the (virtual) previous pass
  // would have left them
  bload(std::integral_constant<int, 0>{}, 0);
#pragma unroll 1
  for (int pc = 0; pc < NPIECE; ++pc) {
    piece_load(std::integral_constant<int, 0>{}, p0, pc);
    piece_store(std::integral_constant<int, 0>{}, 0, pc);
  }
  bstore(std::integral_constant<int, 0>{}, 0);
  static_for<D>([&](auto k) __attribute__((always_inline)) {
    constexpr int K = decltype(k)::value;
    bload(std::integral_constant<int, (K + 1) % D>{}, (K + 1) % NSTEP);  // slices 1 .. D into slots 1 .. D-1, 0
    if constexpr (K - SHIFT >= 0) piece_load(k, p0 + 1, K - SHIFT);                // what steps NSTEP-D+K of pass p0-1 would have loaded
  });
  __syncthreads();

  const float bv = a.bias ? a.bias[nb * 32 + l31] : 0.f;
  float amx = 0.f;
  int bbuf = 0, ibuf = 0;
  // fragments of one 16-deep k-step: B (2 planes) and the TMW row blocks of A (2 planes each)
  struct Frag { u32x4 b[2]; u32x4 a[TMW][2]; };
  auto fload = [&](Frag& f, int sub, int tap0) __attribute__((always_inline)) {  // sub = (tap in slice) * CB + cb (constants)
    const int ts = sub / CB, cb = sub % CB;
    const int tap = tap0 + ts, kh = tap / KW, kw = tap - kh * KW;
    const int toff = GE::tapoff(kh, kw);
    const uint8_t* bb = ldb + bbuf * BSL + (nb * 32 + l31) * 32 + 16 * h + sub * (2 * 64 * 32);
    const uint8_t* ib = lds + ibuf * IBUF;
    f.b[0] = *reinterpret_cast<const u32x4*>(bb);
    f.b[1] = *reinterpret_cast<const u32x4*>(bb + 64 * 32);
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int pp = apix[i] + toff;
      const int off = aoff[i] + pp * GE::PIXB + 16 * ((cb * 2 + h) ^ GE::swz(pp));
      f.a[i][0] = *reinterpret_cast<const u32x4*>(ib + off);
      f.a[i][1] = *reinterpret_cast<const u32x4*>(ib + PLANE + off);
    }
  };
#pragma unroll 1
  for (long pass = p0; pass < p1; ++pass) {
    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto mfmas = [&](const Frag& f) __attribute__((always_inline)) {
      union U { u32x4 u; f16x8 v; };
      U b0, b1;
      b0.u = f.b[0]; b1.u = f.b[1];
#pragma unroll
      for (int i = 0; i < TMW; ++i) {
        U a0, a1;
        a0.u = f.a[i][0]; a1.u = f.a[i][1];
        // small terms first (gemm_bf16x3.h)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.v, b0.v, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.v, b1.v, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.v, b0.v, acc[i], 0, 0, 0);
      }
    };
    // one step: the slice's NS k-steps (the fragments of k-step j + 1 are read before the MFMAs of j), then the rings
    auto step = [&](auto st_c) __attribute__((always_inline)) {
      constexpr int ST = decltype(st_c)::value, K = ST % D;  // step of the pass, its ring slot
      Frag f[2];
      fload(f[0], 0, ST * TPS);
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        // (scheduling fences: left alone the compiler re-used ONE register quad for all A fragments, i.e. read -> wait ->
        // MFMA sixteen times per step with the LDS latency exposed each time -- 3x the step's MFMA time)
        if (j + 1 < NS && !(SRL_IS_DBG & 4)) fload(f[(j + 1) & 1], j + 1, ST * TPS);
        __builtin_amdgcn_sched_barrier(0);
        if (!(SRL_IS_DBG & 2)) mfmas(f[j & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // weights: slice ST + 1 (slot (K + 1) % D) -> the other slice buffer; the slot takes slice ST + 1 + D
      if (!(SRL_IS_DBG & 32)) bstore(std::integral_constant<int, (K + 1) % D>{}, bbuf ^ 1);
      if (!(SRL_IS_DBG & 16)) bload(std::integral_constant<int, (K + 1) % D>{}, (ST + 1 + D) % NSTEP);
      // images: piece ST - SHIFT of the next pass (slot K, loaded D steps ago) -> the other image buffer; the slot takes
      // the piece that is due D steps from now
      if constexpr (ST - SHIFT >= 0 && !(SRL_IS_DBG & 32)) piece_store(std::integral_constant<int, K>{}, ibuf ^ 1, ST - SHIFT);
      constexpr int T = ST + D, PO = T >= NSTEP ? 1 : 0, PL = T - PO * NSTEP - SHIFT;
      if constexpr (PL >= 0 && !(SRL_IS_DBG & 1)) piece_load(std::integral_constant<int, K>{}, pass + 1 + PO, PL);
      __syncthreads();
      bbuf ^= 1;
    };
    static_for<NSTEP>(step);
    // ---- epilogue: rows of the pass are consecutive rows of y ([n][P][64]) ------------------------------------------
    const long img0 = pass * G;
    const long vrows = (a.n - img0 < G ? a.n - img0 : (long)G) * P;  // rows of this pass that exist
    float* yb = a.y + img0 * (long)(P * 64) + nb * 32 + l31;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
      const int rb = (mrow * TMW + i) * 32;
      if (rb >= ROWS) continue;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = fmaf(acc[i][r], inv, bv);
        if (a.act == 1) v[r] = fmaxf(v[r], 0.f);
        else if (a.act == 2) v[r] = tanhf(v[r]);
      }
      if (MASK) {
        const uint32_t wv = srlgemm::sign_words(v, l31, std::make_integer_sequence<int, 16>{});
        const int row = rb + l31;
        if (h == 0 && row < vrows) a.y_mask[(img0 * P + row) * 2 + nb] = wv;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < vrows && (!(SRL_IS_DBG & 8) || v[r] == 12345.f)) {
          yb[(long)row * 64] = v[r];
          amx = fmaxf(amx, fabsf(v[r]));
        }
      }
    }
    ibuf ^= 1;
  }
  if (a.y_absmax) srlgemm::absmax_commit(a.y_absmax, amx);
}
#endif  // __HIPCC__

}  // namespace srlis
