// First convolution on uint8 frames through the BF16 matrix cores, exactly.
//
// The float32 MFMA of gfx950 runs at the vector rate (157 TFLOP/s); the bf16 MFMA at 16x that.  For the first layer the
// 16x is available WITHOUT giving up float32 results, because one operand is bytes:
//   * a uint8 value 0..255 is exactly representable in bf16 (8 significand bits);
//   * a float32 value splits exactly into three bf16 pieces, v = v0 + v1 + v2 (truncation splits: v0 = top 8
//     significand bits, v1 = the next 8, v2 = the last 8; every subtraction below is exact in float32);
//   * a product (bf16 piece) x (byte) has at most 16 significand bits: exact in the MFMA's float32 accumulate.
// So  sum_k x_k * v_k  =  sum over the three planes of a bf16 MFMA, with NO dropped term: the only roundings are the
// float32 accumulations (3 per 16 k-values instead of 16 for the float32 FMA chain).  3 bf16 MFMAs (32 cycles each)
// replace 8 float32 MFMAs (64 cycles each) per 32x32x16 block: 5.3x fewer matrix-pipe cycles.
//
// The whole-observation LayerNorm (x - mean_n) * rstd_n is taken out of the contraction, which sees the bytes minus an
// INTEGER centre c_n = round(mean_n): x - c_n is an integer of magnitude <= 255, still exact in bf16, and what is left
// outside, (mean_n - c_n) in [-0.5, 0.5], is too small to cancel against anything (with raw bytes a flat Atari
// background -- large mean, small variance -- cost several digits: sum x w and mean * sum w nearly cancel):
//   forward   y[n,pos,o] = act( rstd_n * ( sum_k (x[n,pos,k] - c_n) wg[pos,o,k]  -  (mean_n - c_n) S[pos,o] ) + b2[pos,o] )
//             wg = w * gamma (LayerNorm affine folded per output position), S = sum_k wg, b2 = bias + w . beta
//   backward  Q[pos,o,k] = sum_n dz'[n,pos,o] (x[n,pos,k] - c_n)  -  C[pos,o],  dz' = dz rstd_n,  C = sum_n dz' (mean_n - c_n)
//             (Q is what conv.hip's finalisation kernels turn into dW, dgamma, dbeta; R = sum_n dz as before)
// Layout requirements (srl_conv2d_obs_* fall back to the float32 kernels otherwise): uint8 channels-last frames (the
// space-to-depth'd stack), Cout == 32, patch length Kp == 256 bytes made of runs (KW * Cin bytes) that are a power of
// two >= 16 bytes, 16-byte aligned samples.
#pragma once
#include "gemm_core.h"
#include "h2gemm.h"

namespace srlobs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCout = 32;

struct ObsGeom {
  const uint8_t* frames;   // [n][H][W][Cin] uint8
  long img_stride;         // bytes per sample
  int W, Cin, OW, stride;  // position (oy, ox) starts at ((oy * W + ox) * stride) * Cin
  int run_shift;           // log2(KW * Cin): bytes of one contiguous run of the patch
  int run_stride;          // W * Cin: bytes between the runs (kernel rows)
  const float* mean;
  const float* rstd;
  long n;
  // rows addressed through a slot index (observation ring): sample i is frames[row_index[i]], with mean / rstd indexed by
  // the same slot; NULL: sample i is row i
  const int32_t* row_index;
};

#ifdef __HIPCC__
__device__ __forceinline__ long obs_slot(const ObsGeom& g, long n) { return g.row_index ? (long)g.row_index[n] : n; }

// exact three-way truncation split of a float32 into bf16 pieces (returned as the high halves of float32 words)
__device__ __forceinline__ void split3(float v, uint32_t& b0, uint32_t& b1, uint32_t& b2) {
  b0 = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(b0);
  b1 = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(b1);
  b2 = __float_as_uint(r2) & 0xffff0000u;  // r2 has at most 8 significant bits: nothing is cut here
}
// [lo.hi16, hi.hi16]: two bf16 in one dword from the high halves of two float32 words
__device__ __forceinline__ uint32_t pack_hi(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// 8 bytes (two dwords) -> 8 bf16, element j = byte j minus the (integer-valued) centre c
__device__ __forceinline__ bf16x8 bytes_to_bf16x8(uint32_t d0, uint32_t d1, float c) {
  union { uint32_t u[4]; bf16x8 v; } r;
#define SRL_B2F(d, i) __float_as_uint((float)(((d) >> (8 * (i))) & 255u) - c)
  r.u[0] = pack_hi(SRL_B2F(d0, 0), SRL_B2F(d0, 1));
  r.u[1] = pack_hi(SRL_B2F(d0, 2), SRL_B2F(d0, 3));
  r.u[2] = pack_hi(SRL_B2F(d1, 0), SRL_B2F(d1, 1));
  r.u[3] = pack_hi(SRL_B2F(d1, 2), SRL_B2F(d1, 3));
#undef SRL_B2F
  return r.v;
}

// byte offset of patch byte b (a multiple of 16) from the patch origin
__device__ __forceinline__ int patch_off(const ObsGeom& g, int b) {
  return (b >> g.run_shift) * g.run_stride + (b & ((1 << g.run_shift) - 1));
}

// (position, sample range) of a workgroup.  Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its
// own 4 MiB L2.  Every XCD gets its own group of ceil(P / 8) consecutive output positions -- neighbouring positions
// read the same frame rows, so the group's workgroups share them through that L2 -- and ALL XCDs walk the sample
// ranges in the same order, so that the frames in flight chip-wide are one or two ranges (tens of MB: they stay in the
// 256 MB Infinity Cache for the other XCDs' overlapping rows) instead of a different range per XCD.
// grid = 8 * ceil(P / 8) * nsplit; returns false for the padding ids of a last, shorter group.
__device__ __forceinline__ bool xcd_position_split(int P, int& pos, int& split) {
  const int per = (P + 7) >> 3;
  const int xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  pos = xcd * per + slot % per;
  split = slot / per;
  return pos < P && slot % per < per;
}
inline unsigned xcd_position_grid(int P, int nsplit) { return 8u * (unsigned)((P + 7) / 8) * (unsigned)nsplit; }

// ---- fold + split: wq[pos][plane][kb][lane][8] bf16 (MFMA B-operand fragments, columns = output channels) -------------
// MFMA k-block kb = 2c + e of a lane with half h covers patch bytes 32c + 16h + 8e + (0..7): a lane's 16-byte load of
// the patch feeds two consecutive k-blocks.  Any pairing of k-values works as long as both operands use the same one.
template <class Index>
__global__ __launch_bounds__(256) void obs_fold_split_kernel(const float* w, const float* bias, const float* gamma,
                                                             const float* beta, int P, int Kp, Index ix, uint16_t* wq,
                                                             float* S, float* b2, float* bound = nullptr, float sqrt_n = 0.f) {
  __shared__ double red[12];
  const int pos = blockIdx.x / kCout, o = blockIdx.x % kCout;
  const int oh = pos / ix.OW, ow = pos % ix.OW;
  const int nkb = Kp / 16;
  double acc[3] = {0.0, 0.0, 0.0};  // sum_k w gamma, sum_k w beta, sum_k (w gamma)^2
  for (int k = threadIdx.x; k < Kp; k += 256) {
    int ci, kh, kw;
    ix.split_k(k, ci, kh, kw);
    const int p = ix.p_of(ci, oh * ix.S + kh, ow * ix.S + kw);
    const float wv = w[o * Kp + k];
    const float wg = wv * gamma[p];  // the float32 product the float32 path folds as well
    acc[0] += (double)wg;
    acc[1] += (double)wv * (double)beta[p];
    acc[2] += (double)wg * (double)wg;
    uint32_t b[3];
    split3(wg, b[0], b[1], b[2]);
    const int c = k >> 5, h = (k >> 4) & 1, e = (k >> 3) & 1, j = k & 7;
    const int kb = 2 * c + e, lane = o + 32 * h;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) wq[((((long)pos * 3 + pl) * nkb + kb) * 64 + lane) * 8 + j] = (uint16_t)(b[pl] >> 16);
  }
  block_sum<3, 256>(acc, red);
  if (threadIdx.x == 0) {
    S[pos * kCout + o] = (float)acc[0];
    const double bb = (bias ? (double)bias[o] : 0.0) + acc[1];
    b2[pos * kCout + o] = (float)bb;
    // |y| = |sum_k xhat_k wg_k + b2| <= |xhat_patch|_2 |wg|_2 + |b2| <= sqrt(N) |wg|_2 + |b2|: the normalised observation has
    // sum of squares <= N over ALL its N elements (Cauchy-Schwarz; holds for any frame)
    if (bound) atomicMax(reinterpret_cast<int*>(bound), __float_as_int((float)(sqrt(acc[2]) * (double)sqrt_n + fabs(bb)) * 1.0001f));
  }
}

// ---- forward ------------------------------------------------------------------------------------------------------------
struct FwdArgs {
  ObsGeom g;
  const uint4* wq;  // [P][3][KP/16][64] fragments
  const float* S;   // [P][32]
  const float* b2;  // [P][32]
  float* y;         // [n][P][32]
  int P, nsplit, act;
  float* y_absmax;  // max |y| folded in with an atomic max (the range of the next layer's operand), or NULL
  uint32_t* y_mask;  // [n][P] words: bit o of word (n, pos) = (y[n][pos][o] > 0), the ReLU derivative for the data gradient
                     // of the next layer (GemmArgs::dact_mask), or NULL
  // H2OUT: instead of y, the pre-split activation the image-stationary convolution behind this layer reads (h2conv.h):
  // rows of h2p, [n][P][32 channels -> 128 bytes], the P positions of a sample in `ent_order` (h2_entry: 2 = parity-class
  // major for a stride-2 consumer), under the power-of-two scale derived from *bound (obs_fold_split_kernel: an upper bound
  // of |y| from the folded weights, known before a single sample is seen); *y_scale = that scale, for the consumers
  uint8_t* y_h2;
  const float* bound;
  float* y_scale;
  int ent_order, OH;
};

// One workgroup = one output position x one range of samples.  The position's folded weights (3 planes, 48 KB) sit in
// LDS in fragment order; every wavefront walks its own 32-sample tiles: 16-byte loads of the raw patch bytes straight
// into registers (no staging: a tile's bytes are used by this wavefront only), bytes -> bf16 in registers, 3 MFMAs per
// 16 k-values (A = the samples' bytes, rows = samples; B = weights from LDS, columns = output channels), LayerNorm +
// bias + activation on the accumulators, whole-line stores.  The next tile's loads are issued before the current
// tile's arithmetic.
// DBG (timing experiments only, wrong results; SRL_OBS_DBG): 1 = bytes reinterpreted instead of converted, 2 = no MFMAs,
// 4 = no stores, 8 = the patch loads of every tile go to the first tile's rows (cache-hot)
// MASK: also leave the sign bits of the ReLU output (FwdArgs::y_mask).  A template parameter because the kernel sits at the
// register limit of three wavefronts per SIMD: the variant without it must stay exactly what it was.
template <int KP, int DBG = 0, bool MASK = false, bool H2OUT = false>
__global__ __launch_bounds__(256, 3) void obs_fwd_bf16_kernel(FwdArgs a) {
  constexpr int NKB = KP / 16, NC = KP / 32;
  __shared__ uint4 Wl[3 * NKB * 64];
  __shared__ __attribute__((aligned(16))) float rowl[4][2][2][32];  // [wave][register set][rstd | -(mean - c) rstd][row of the tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int pos, split;
  if (!xcd_position_split(a.P, pos, split)) return;
  {
    const uint4* src = a.wq + (long)pos * 3 * NKB * 64;
#pragma unroll
    for (int i = 0; i < 3 * NKB * 64 / 256; ++i) Wl[tid + 256 * i] = src[tid + 256 * i];
  }
  const float S_o = a.S[pos * kCout + l31], b2_o = a.b2[pos * kCout + l31];  // this lane's output channel is l31
  __shared__ __attribute__((aligned(16))) float sbl[2][32];  // H2OUT: S and b2 of the position's 32 channels (lanes hold samples there)
  float oscale = 1.f;
  int ent = 0;
  if (H2OUT) {
    if (tid < 32) { sbl[0][tid] = a.S[pos * kCout + tid]; sbl[1][tid] = a.b2[pos * kCout + tid]; }
    oscale = srlh2::h2_scale_for(*a.bound);
    if (blockIdx.x == 0 && tid == 0) *a.y_scale = oscale;
    const int py = pos / a.g.OW, px = pos % a.g.OW;
    ent = a.ent_order == 2 ? ((py & 1) * 2 + (px & 1)) * ((a.OH / 2) * (a.g.OW / 2)) + (py >> 1) * (a.g.OW / 2) + (px >> 1) : pos;
  }
  __syncthreads();
  const long ntiles = (a.g.n + 31) / 32;
  const long t0 = ntiles * split / a.nsplit, t1 = ntiles * (split + 1) / a.nsplit;
  const int oy = pos / a.g.OW, ox = pos % a.g.OW;
  const long posoff = ((long)(oy * a.g.W + ox) * a.g.stride) * a.g.Cin;
  int poff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) poff[c] = patch_off(a.g, 32 * c + 16 * h);
  const long ldy = (long)a.P * kCout;

  // The patch bytes of a tile sit in ONE register set: as soon as a 32-byte chunk has been converted its registers take
  // the same chunk of the wavefront's NEXT tile, so the loads run a whole tile ahead at half the registers of two sets
  // (which, with the fragment registers, would leave two wavefronts per SIMD instead of three).
  uint4 raw[NC];
  auto rowptr = [&](long tile) {
    long n = tile * 32 + l31;
    if (n >= a.g.n) n = a.g.n - 1;  // clamped: the loads stay in bounds, the stores of such rows are masked
    return a.g.frames + ((DBG & 8) ? (long)l31 : obs_slot(a.g, n)) * a.g.img_stride + posoff;
  };
  // (rstd, -(mean - c) rstd) of the accumulator ROWS go through LDS for the wavefront's own later reads (a wavefront's
  // LDS instructions execute in order; no other wavefront touches this slice); returns the integer centre c in [0, 255]
  auto rowstats = [&](int set, long tile) {
    long n = tile * 32 + l31;
    if (n >= a.g.n) n = a.g.n - 1;
    n = obs_slot(a.g, n);
    const float r = a.g.rstd[n], m = a.g.mean[n];
    const float c = rintf(m);
    if (h == 0) {
      rowl[wave][set][0][l31] = r;
      rowl[wave][set][1][l31] = -(m - c) * r;
    }
    return c;
  };
  float amx = 0.f;
  auto compute = [&](int set, long tile, float cen, const uint8_t* next_row) {
    asm volatile("" ::: "memory");  // keeps the compiler from hoisting the (tile-invariant) fragment reads out of the loop
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      uint4 wfr[6];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wfr[3 * e + pl] = Wl[(pl * NKB + 2 * c + e) * 64 + lane];
      const uint4 q = raw[c];
      bf16x8 x0, x1;
      if (DBG & 1) {
        union { uint4 u; bf16x8 v; } t0, t1;
        t0.u = make_uint4(q.x, q.y, q.x, q.y);
        t1.u = make_uint4(q.z, q.w, q.z, q.w);
        x0 = t0.v, x1 = t1.v;
      } else {
        x0 = bytes_to_bf16x8(q.x, q.y, cen), x1 = bytes_to_bf16x8(q.z, q.w, cen);
      }
      // the next tile's loads go out one 128-byte run at a time (its four 32-byte chunks back to back, as soon as the
      // last of their registers is free): issued chunk by chunk, between the MFMAs, the four requests for a line were
      // far enough apart for the line to leave the L1 in between
      if (next_row && (c & 3) == 3) {
#pragma unroll
        for (int cc = c - 3; cc <= c; ++cc) raw[cc] = *reinterpret_cast<const uint4*>(next_row + poff[cc]);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          union { uint4 u; bf16x8 v; } wf;
          wf.u = wfr[3 * e + pl];
          if (DBG & 2) {
            union { bf16x8 v; float f[4]; } xx;
            xx.v = e ? x1 : x0;
            acc[3 * e + pl] += __uint_as_float(wf.u.x) * xx.f[0];
          } else {
            // H2OUT: operands swapped -- D[o][n], lanes = samples: a lane then holds 16 channels of ONE sample, which is what
            // a 16-byte chunk of h2p is made of
            if (H2OUT) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf.v, e ? x1 : x0, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e ? x1 : x0, wf.v, acc, 0, 0, 0);
          }
        }
    }
    // D[n][o]: this lane holds output channel o = l31 of the samples tile * 32 + 8g + 4h + (0..3) in registers 4g .. 4g+3,
    // so one store instruction writes two whole 128-byte rows of y (the channels of a sample at this position).  With the
    // accumulator the other way round (rows = channels) every instruction wrote 32-byte pieces of 32 different lines:
    // the stores alone then cost 0.25 ms of the kernel's 0.66.
    __builtin_amdgcn_wave_barrier();
    const long n0 = tile * 32;
    const bool full = n0 + 32 <= a.g.n;  // wave-uniform: only the batch's last tile is ragged
    if (H2OUT) {
      // lane = sample n0 + l31 (both halves h), register r = channel (r & 3) + 8 (r >> 2) + 4 h: registers 0-7 are group 2 h of
      // h2p, registers 8-15 group 2 h + 1 -- 64 contiguous bytes per lane, the two halves complete the sample's 128-byte row
      const float rv = rowl[wave][set][0][l31], mv = rowl[wave][set][1][l31];
      const bool ok = full || n0 + l31 < a.g.n;
      float v[16];
      uint32_t bits = 0;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 s4 = *reinterpret_cast<const float4*>(&sbl[0][8 * g4 + 4 * h]);
        const float4 b4 = *reinterpret_cast<const float4*>(&sbl[1][8 * g4 + 4 * h]);
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float t = fmaf(rv, acc[4 * g4 + i], fmaf(mv, sv[i], bv[i]));
          if (a.act == 1) t = fmaxf(t, 0.f);
          else if (a.act == 2) t = tanhf(t);
          v[4 * g4 + i] = t;
          if (ok) amx = fmaxf(amx, fabsf(t));
          bits |= (t > 0.f ? 1u : 0u) << (8 * g4 + 4 * h + i);
        }
      }
      uint4 c[4];
      srlh2::h2_split_pair(v[0], v[1], oscale, c[0].x, c[1].x);
      srlh2::h2_split_pair(v[2], v[3], oscale, c[0].y, c[1].y);
      srlh2::h2_split_pair(v[4], v[5], oscale, c[0].z, c[1].z);
      srlh2::h2_split_pair(v[6], v[7], oscale, c[0].w, c[1].w);
      srlh2::h2_split_pair(v[8], v[9], oscale, c[2].x, c[3].x);
      srlh2::h2_split_pair(v[10], v[11], oscale, c[2].y, c[3].y);
      srlh2::h2_split_pair(v[12], v[13], oscale, c[2].z, c[3].z);
      srlh2::h2_split_pair(v[14], v[15], oscale, c[2].w, c[3].w);
      if (ok) {
        uint4* d = reinterpret_cast<uint4*>(a.y_h2 + ((n0 + l31) * (long)a.P + ent) * 128 + h * 64);
        d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = c[3];
      }
      if (MASK) {
        bits |= (uint32_t)__shfl_xor((int)bits, 32);
        if (ok && h == 0) a.y_mask[(n0 + l31) * a.P + pos] = bits;
      }
      return;
    }
    float* yp = a.y + n0 * ldy + (long)pos * kCout + l31;
    const int ldy32 = (int)ldy;
    float vv[16];  // MASK: the stored values in accumulator order
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 r4 = *reinterpret_cast<const float4*>(&rowl[wave][set][0][8 * g4 + 4 * h]);
      const float4 m4 = *reinterpret_cast<const float4*>(&rowl[wave][set][1][8 * g4 + 4 * h]);
      const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = fmaf(rv[i], acc[4 * g4 + i], fmaf(mv[i], S_o, b2_o));
        if (a.act == 1) v = fmaxf(v, 0.f);
        else if (a.act == 2) v = tanhf(v);
        const int row = 8 * g4 + 4 * h + i;
        vv[4 * g4 + i] = v;
        if ((full || n0 + row < a.g.n) && (!(DBG & 4) || v == 12345.f)) {
          yp[row * ldy32] = v;
          amx = fmaxf(amx, fabsf(v));
        }
      }
    }
    if (MASK) {  // lanes = channels: a ballot per accumulator register is the sign words of two rows, dropped into the lanes of
      // those rows (srlgemm::sign_words); then one store per tile
      const uint32_t mword = srlgemm::sign_words(vv, l31, std::make_integer_sequence<int, 16>{});
      if (h == 0 && (full || n0 + l31 < a.g.n)) a.y_mask[(n0 + l31) * a.P + pos] = mword;
    }
  };
  long t = t0 + wave;
  if (t < t1) {
    const uint8_t* rp = rowptr(t);
#pragma unroll
    for (int c = 0; c < NC; ++c) raw[c] = *reinterpret_cast<const uint4*>(rp + poff[c]);
  }
  float cen = t < t1 ? rowstats(0, t) : 0.f;
  int set = 0;
  for (; t < t1; t += 4) {
    const bool more = t + 4 < t1;
    const float cen_next = more ? rowstats(set ^ 1, t + 4) : 0.f;
    compute(set, t, cen, more ? rowptr(t + 4) : nullptr);
    cen = cen_next;
    set ^= 1;
  }
  if (a.y_absmax) srlgemm::absmax_commit(a.y_absmax, amx);
}

// ---- backward: Q[pos][o][k] (+ split-K slabs), R[pos][o] += sum dz, C[pos][o] += sum dz' mean -----------------------------
struct BwdArgs {
  ObsGeom g;
  const float* dz;  // [n][P][32]
  float* Q;         // [nsplit][P][32][KP] (slabs) or [P][32][KP]
  long slab;
  float* R;         // [P][32], atomically accumulated
  float* C;         // [P][32], atomically accumulated
  int P, nsplit;
};

// One workgroup = one output position x one range of samples; K-steps of 32 samples.  Both operands are contiguous
// along the OUTPUT index in memory (dz rows of 32 channels, patch rows of KP bytes) while the MFMA wants 8 consecutive
// k (= samples) per lane: the tiles go to LDS as they come -- [sample][channel] bf16 (three planes of dz') and
// [sample][patch byte] bf16 -- and the fragments are fetched with ds_read_b64_tr_b16, the transposing LDS read of
// gfx950 (4 rows x 16 columns per 16 lanes, delivered column-major): no register transposes, no scalar LDS stores.
// Wavefront w owns patch columns 64 w .. 64 w + 63 (two 32 x 32 accumulator tiles); all wavefronts share the dz' tile.
template <int KP>
__global__ __launch_bounds__(256, 2) void obs_bwd_bf16_kernel(BwdArgs a) {
  constexpr int KS = 32;                       // samples per K-step
  constexpr int A_BYTES = KS * kCout * 2;      // one plane of dz': [32 samples][32 channels] bf16, 64-byte rows
  constexpr int B_BYTES = KS * KP * 2;         // [32 samples][KP] bf16, KP * 2-byte rows, 16-byte chunks XOR-swizzled
  constexpr int BUF = 3 * A_BYTES + B_BYTES;
  constexpr int NBQ = KP / 128;                // 16-byte chunks of patch bytes per thread and K-step (8 threads per sample)
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  int pos, split;
  if (!xcd_position_split(a.P, pos, split)) return;
  const long nsteps_all = (a.g.n + KS - 1) / KS;
  const long s0 = nsteps_all * split / a.nsplit, s1 = nsteps_all * (split + 1) / a.nsplit;
  const int oy = pos / a.g.OW, ox = pos % a.g.OW;
  const long posoff = ((long)(oy * a.g.W + ox) * a.g.stride) * a.g.Cin;
  const long ldz = (long)a.P * kCout;

  // staging roles: thread -> sample row sr = tid / 8 of the K-step, 4 channels o4 of dz, NBQ 16-byte patch chunks
  const int sr = tid >> 3, sub = tid & 7;
  int boff[NBQ];
#pragma unroll
  for (int i = 0; i < NBQ; ++i) boff[i] = patch_off(a.g, 16 * (sub + 8 * i));
  // two register sets: the global loads of a K-step are issued TWO steps before its tile is written to LDS (one step
  // of lead left the wavefronts parked on s_waitcnt: half of their cycles by the counters)
  float4 dzr[2];
  uint4 xb[2][NBQ];
  float s_rs[2] = {0.f, 0.f}, s_mean[2] = {0.f, 0.f}, s_cen[2] = {0.f, 0.f};
  float rsum[4] = {0.f, 0.f, 0.f, 0.f}, csum[4] = {0.f, 0.f, 0.f, 0.f};

#ifndef SRL_OBSB_DBG
#define SRL_OBSB_DBG 0  // timing experiments (wrong results): 1 no in-loop global loads, 2 no conversions, 4 no MFMAs, 8 no LDS stores
#endif
  auto gload = [&](int set, long step) {
    if ((SRL_OBSB_DBG & 1) && step > s0 + 2) return;
    const long n = step * KS + sr;
    const bool ok = n < a.g.n;
    const long nn = ok ? n : a.g.n - 1;
    dzr[set] = ok ? *reinterpret_cast<const float4*>(a.dz + nn * ldz + (long)pos * kCout + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    const long slot = obs_slot(a.g, nn);
    const uint8_t* rowp = a.g.frames + slot * a.g.img_stride + posoff;
#pragma unroll
    for (int i = 0; i < NBQ; ++i) xb[set][i] = *reinterpret_cast<const uint4*>(rowp + boff[i]);
    s_rs[set] = ok ? a.g.rstd[slot] : 0.f;  // rows past the end contribute zeros (dz' = 0)
    s_mean[set] = a.g.mean[slot];
  };
  auto lstore = [&](int set, uint8_t* buf) {
    const float d[4] = {dzr[set].x, dzr[set].y, dzr[set].z, dzr[set].w};
    const float cen = rintf(s_mean[set]);  // integer centre in [0, 255]
    uint32_t pl[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ds = d[i] * s_rs[set];
      rsum[i] += d[i];
      csum[i] = fmaf(ds, s_mean[set] - cen, csum[i]);
      split3(ds, pl[0][i], pl[1][i], pl[2][i]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint2*>(buf + p * A_BYTES + sr * (kCout * 2) + sub * 8) =
          make_uint2(pack_hi(pl[p][0], pl[p][1]), pack_hi(pl[p][2], pl[p][3]));
    uint8_t* bb = buf + 3 * A_BYTES + sr * (KP * 2);
#pragma unroll
    for (int i = 0; i < NBQ; ++i) {
      // patch bytes 16 j .. 16 j + 15 (j = sub + 8 i) -> bf16 columns: 32 bytes = chunks 2 j, 2 j + 1 of the row
      const int j = sub + 8 * i;
      union { bf16x8 v; uint4 u; } lo, hi;
      if (SRL_OBSB_DBG & 2) {
        lo.u = make_uint4(xb[set][i].x, xb[set][i].y, xb[set][i].x, xb[set][i].y);
        hi.u = make_uint4(xb[set][i].z, xb[set][i].w, xb[set][i].z, xb[set][i].w);
      } else {
        lo.v = bytes_to_bf16x8(xb[set][i].x, xb[set][i].y, cen);
        hi.v = bytes_to_bf16x8(xb[set][i].z, xb[set][i].w, cen);
      }
      if (SRL_OBSB_DBG & 8) continue;
      const int sw = (sr & 3) << 2;  // rows 4 apart in time share banks otherwise: see the transposed reads below
      *reinterpret_cast<uint4*>(bb + 16 * ((2 * j) ^ sw)) = lo.u;
      *reinterpret_cast<uint4*>(bb + 16 * ((2 * j + 1) ^ sw)) = hi.u;
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // transposed-read addresses.  ds_read_b64_tr_b16: in each group of 16 lanes, lane 4 q + p supplies the address of
  // row q, columns 4 p .. 4 p + 3 of a 4 x 16 block; lane i of the group receives column i (4 rows).  For the
  // 32 x 32 x 16 operand a lane with half h needs k-rows 8 h .. 8 h + 7 of its column: two reads (rows +0..3, +4..7).
  const int grp = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int col0 = 16 * (grp & 1) + 4 * p;  // first of the 4 columns this lane addresses, within a 32-column tile
  auto a_addr = [&](const uint8_t* buf, int plane, int kb, int r) {
    const int row = 16 * kb + 8 * h + 4 * r + q;
    return buf + plane * A_BYTES + row * (kCout * 2) + col0 * 2;
  };
  auto b_addr = [&](const uint8_t* buf, int jt, int kb, int r) {
    const int row = 16 * kb + 8 * h + 4 * r + q;
    const int colb = (64 * wave + 32 * jt + col0) * 2;  // byte offset of the column inside the row
    const int ch = colb >> 4;
    return buf + 3 * A_BYTES + row * (KP * 2) + 16 * (ch ^ ((row & 3) << 2)) + (colb & 15);
  };
  auto trread = [&](const uint8_t* p0, const uint8_t* p1) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
    return u.v;
  };

  // one K-step on LDS buffer CUR (= parity of s - s0): MFMAs, with the staging of the following tiles in the middle
  auto kstep = [&](auto cur_c, long s) {
    constexpr int CUR = decltype(cur_c)::value;
    const uint8_t* buf = lds + CUR * BUF;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 bf[2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) bf[jt] = trread(b_addr(buf, jt, kb, 0), b_addr(buf, jt, kb, 1));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const bf16x8 af = trread(a_addr(buf, pl, kb, 0), a_addr(buf, pl, kb, 1));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          if (SRL_OBSB_DBG & 4) { union { bf16x8 v; float f[4]; } xa, xb2; xa.v = af; xb2.v = bf[jt]; acc[jt][pl] += xa.f[0] * xb2.f[0]; }
          else acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[jt], acc[jt], 0, 0, 0);
        }
      }
      if (kb == 0 && s + 1 < s1) {  // tile s+1: registers (set CUR^1) -> the other buffer (its readers left at the last barrier)
        lstore(CUR ^ 1, lds + (CUR ^ 1) * BUF);
        if (s + 3 < s1) gload(CUR ^ 1, s + 3);  // ... and that register set takes the loads of tile s+3
      }
    }
    __syncthreads();
  };
  if (s0 < s1) {
    gload(0, s0);
    lstore(0, lds);
    if (s0 + 1 < s1) gload(1, s0 + 1);
    if (s0 + 2 < s1) gload(0, s0 + 2);
  }
  __syncthreads();
  for (long s = s0; s < s1; s += 2) {
    kstep(std::integral_constant<int, 0>{}, s);
    if (s + 1 < s1) kstep(std::integral_constant<int, 1>{}, s + 1);
  }

  // column sums of dz (R) and the mean correction (C): the 32 threads that staged the same 4 channels combine in LDS
  {
    float* red = reinterpret_cast<float*>(lds);  // the tiles are dead after the loop's last barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[tid * 8 + i] = rsum[i];
      red[tid * 8 + 4 + i] = csum[i];
    }
    __syncthreads();
    if (tid < 64) {  // tid = (which << 5) | channel
      const int ch = tid & 31, which = tid >> 5;
      float t = 0.f;
      for (int r = 0; r < 32; ++r) t += red[(r * 8 + (ch >> 2)) * 8 + which * 4 + (ch & 3)];
      atomicAdd((which ? a.C : a.R) + pos * kCout + ch, t);
    }
  }
  // D[o][k]: registers 4g .. 4g+3 hold channels 8g + 4h + (0..3) of patch column 64 wave + 32 jt + l31
  float* qo = a.Q + (long)split * a.slab + (long)pos * kCout * KP;
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 8 * (r >> 2) + (r & 3) + 4 * h;
      qo[(long)o * KP + 64 * wave + 32 * jt + l31] = acc[jt][r];
    }
}
#endif  // __HIPCC__

}  // namespace srlobs
