// First convolution on uint8 frame stacks, block-of-positions form (round 4): the forward pass that writes the pre-split
// activation (h2p rows, h2gemm.h) for the image-stationary convolution behind it.
//
// obs_bf16.h gives every output position a workgroup of its own: the 256 patch bytes of a sample are fetched once per position,
// i.e. every byte of a frame four times (2 x 2 overlapping patches of the space-to-depth grid), by different workgroups at
// different times -- 1.08 GB of frame reads per 16 384-frame chunk where the frames are 0.46 GB -- and the position's weights
// are re-read from LDS for every tile.  Here a workgroup owns a 2 x 4 BLOCK of positions (8 wavefronts, one position each, two
// wavefronts per SIMD) and walks the samples in tiles of 32:
//   * the block's 3 x 5 pixels of a sample (960 bytes: 3 runs of 320) come in ONCE, by LDS-DMA, one wave-instruction per sample,
//     into a 3-stage ring of 32-sample tiles; 0.79 GB of requested bytes per chunk (1.15 KB of 128-byte lines per sample and
//     block: 0.94 GB fetched) instead of 1.08;
//   * a position's folded weights w * gamma do not change with the sample: they stay in REGISTERS for the whole walk -- as two
//     f16 pieces under a power-of-two scale per output channel (22 significand bits; the bytes, minus the integer centre of
//     the sample's mean, are exact in f16): 2 MFMAs per 16 k-values instead of obs_bf16.h's 3 bf16 pieces.  128 VGPRs -- the
//     A/B operands of an MFMA must be architectural registers, so one position per wavefront is what fits beside the working
//     set in the 256 registers of a wavefront that shares its SIMD with another;
//   * per-sample records {slot, rstd, mean} are written by a tiny pre-pass and prefetched by DMA four tiles ahead: no dependent
//     scalar loads in the loop (row_index -> mean / rstd is two memory latencies deep), no 64-bit divisions (cursors);
//   * D[channel][sample] (weights as the A operand): a lane holds 16 channels of one sample = 64 contiguous bytes of the
//     sample's h2p row at this position, two lanes complete the 128-byte line;
//   * the only LDS traffic besides the DMA is 8 conflict-free ds_read_b128 per lane and tile (61 chunks between samples).
// Measured on 16 384 frames (scripts/obs_fwd_bench.py; consecutive / permuted slots): obs_bf16.h 642 / 733 us; 2 x 2 blocks, 4
// wavefronts, one workgroup per CU 600 / 693; the same, two workgroups per CU 512 / 583; 2 x 4 blocks, 8 wavefronts 462 / 542.
// Same arithmetic contract as obs_bf16.h (its header): y = act(rstd_n (sum_k (x - c_n) wg - (mean_n - c_n) S) + b2).
#pragma once
#include "obs_bf16.h"

namespace srlobs {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kBlkH = 2, kBlkW = 4;        // positions per workgroup: 2 rows x 4 columns, one per wavefront (8 wavefronts)
constexpr int kWaves = kBlkH * kBlkW;
constexpr int kTile = 32;                   // samples per tile
constexpr int kRowChunks = (kBlkW + 1) * 4; // 16-byte chunks of one row of the block's pixels ((kBlkW + 1) pixels of 64 bytes)
constexpr int kChunks = 3 * kRowChunks;     // ... of the block's 3 rows of one sample
constexpr int kSampleBytes = (kChunks + 1) * 16;  // an ODD number of chunks between samples: the 16 lanes of a ds_read_b128 group
                                            // (16 samples, one chunk each) then fall on 16 different 16-byte bank groups
static_assert(kChunks % 2 == 0 && kChunks <= 64, "one LDS-DMA instruction per sample");
constexpr int kStageBytes = kTile * kSampleBytes;
constexpr int kStages = 3;
constexpr int kMeta = 5;                     // tiles of per-sample records in flight: the one computed .. four ahead

struct FwdH2Args {
  const uint8_t* frames;     // [slot][GH][GW][64] uint8 (space-to-depth'd stacks)
  long img_stride;           // bytes per sample
  const uint4* meta;         // [ceil(n / 32) * 32] x {slot, rstd, mean, 0}: obs_meta_kernel (rows past n repeat row n - 1)
  long n;
  const uint4* wq;           // [P][2 pieces][16 k-blocks][64 lanes] x 16 bytes: A-operand fragments (rows = channels)
  const float* winv;         // [P][32]: 1 / the power-of-two scale of the channel's weights
  const float* S;            // [P][32]
  const float* b2;           // [P][32]
  uint8_t* y_h2;             // [n][P][128 bytes] h2p rows, positions in parity-class-major order
  const float* bound;
  float* y_scale;
  uint32_t* y_mask;          // [n][P]
  float* y_absmax;
  int GW, OW, OH, P, act;
};

#ifdef __HIPCC__
// ---- fold: wg = w * gamma per position, as two f16 pieces under a per-(position, channel) power-of-two scale ---------------
template <class Index>
__global__ __launch_bounds__(256) void obs_fold_h2_kernel(const float* w, const float* bias, const float* gamma, const float* beta,
                                                          int P, Index ix, uint16_t* wq, float* winv, float* S, float* b2,
                                                          float* bound, float sqrt_n) {
  __shared__ double red[12];
  __shared__ float mx[4];
  constexpr int Kp = 256;
  const int pos = blockIdx.x / kCout, o = blockIdx.x % kCout;
  const int oh = pos / ix.OW, ow = pos % ix.OW;
  const int k = threadIdx.x;
  int ci, kh, kw;
  ix.split_k(k, ci, kh, kw);
  const int p = ix.p_of(ci, oh * ix.S + kh, ow * ix.S + kw);
  const float wv = w[o * Kp + k];
  const float wg = wv * gamma[p];
  double acc[3] = {(double)wg, (double)wv * (double)beta[p], (double)wg * (double)wg};
  float m = fabsf(wg);
#pragma unroll
  for (int off = 32; off; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) mx[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
  int e = 0;
  if (m > 0.f) (void)frexpf(m, &e);        // m = f 2^e, f in [0.5, 1): m 2^(14 - e) in [2^13, 2^14)
  const float s = ldexpf(1.f, 14 - e);
  const float v = wg * s;
  const _Float16 h0 = (_Float16)v;
  const _Float16 h1 = (_Float16)(v - (float)h0);
  const int c = k >> 5, h = (k >> 4) & 1, eb = (k >> 3) & 1, j = k & 7;
  const int kb = 2 * c + eb, lane = o + 32 * h;
  union { _Float16 f; uint16_t u; } c0, c1;
  c0.f = h0; c1.f = h1;
  wq[((((long)pos * 2 + 0) * 16 + kb) * 64 + lane) * 8 + j] = c0.u;
  wq[((((long)pos * 2 + 1) * 16 + kb) * 64 + lane) * 8 + j] = c1.u;
  block_sum<3, 256>(acc, red);
  if (threadIdx.x == 0) {
    S[pos * kCout + o] = (float)acc[0];
    const double bb = (bias ? (double)bias[o] : 0.0) + acc[1];
    b2[pos * kCout + o] = (float)bb;
    winv[pos * kCout + o] = 1.f / s;
    if (bound) atomicMax(reinterpret_cast<int*>(bound), __float_as_int((float)(sqrt(acc[2]) * (double)sqrt_n + fabs(bb)) * 1.0001f));
  }
}

// per-sample record of a launch: the slot of the sample's frame and its LayerNorm statistics side by side, so that the main kernel
// finds both with ONE LDS-DMA per tile and no dependent scalar loads (row_index -> mean / rstd is two memory latencies deep)
__global__ __launch_bounds__(256) void obs_meta_kernel(const int32_t* row_index, const float* mean, const float* rstd, long n,
                                                       long n_pad, uint4* meta) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  const long smp = i < n ? i : n - 1;
  const long slot = row_index ? (long)row_index[smp] : smp;
  meta[i] = make_uint4((uint32_t)slot, __float_as_uint(rstd[slot]), __float_as_uint(mean[slot]), 0u);
}

__device__ __forceinline__ void obs_wait_vm_dyn(int n) {
  switch (n) {
#define SRL_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    SRL_W(0) SRL_W(1) SRL_W(2) SRL_W(3) SRL_W(4) SRL_W(5) SRL_W(6) SRL_W(7) SRL_W(8) SRL_W(9) SRL_W(10) SRL_W(11) SRL_W(12)
    SRL_W(13) SRL_W(14) SRL_W(15) SRL_W(16) SRL_W(17) SRL_W(18) SRL_W(19) SRL_W(20) SRL_W(21) SRL_W(22) SRL_W(23) SRL_W(24)
#undef SRL_W
    default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;  // (more outstanding than ever issued between two uses)
  }
}

// 16 bytes per active lane into LDS at `lds` + 16 lane, from the lane's own address
__device__ __forceinline__ void obs_dma(uint32_t lds, const void* src) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds)), "v"(src)
               : "memory");
}

// bytes (b0 b1 b2 b3) of a dword -> two dwords of f16 pairs holding b - cen, exact: 0x6400 | b is the f16 1024 + b
__device__ __forceinline__ void obs_bytes_to_f16(uint32_t d, uint32_t negc, uint32_t& lo, uint32_t& hi) {
  typedef _Float16 h2t __attribute__((ext_vector_type(2)));
  union { uint32_t u; h2t v; } a, b, c;
  a.u = __builtin_amdgcn_perm(0x64646464u, d, 0x05010400u);  // (b0, 0x64, b1, 0x64)
  b.u = __builtin_amdgcn_perm(0x64646464u, d, 0x05030402u);  // (b2, 0x64, b3, 0x64)
  c.u = negc;                                                // (-(1024 + cen)) twice
  a.v = a.v + c.v;
  b.v = b.v + c.v;
  lo = a.u;
  hi = b.u;
}

// ACT: the activation (0 none, 1 ReLU, 2 tanh).  DBG (timing experiments, wrong results; SRL_OBS_DBG): 1 = no output stores, 2 = no LDS
// reads / conversions / MFMAs, 4 = no DMA
template <int ACT, int DBG = 0>
__global__ __launch_bounds__(64 * kWaves, kWaves == 4 ? 2 : 1) void obs_fwd_h2_kernel(FwdH2Args a) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  // [kStages][32 samples][kChunks + 1 chunks][16 B] | meta ring [kMeta][32] x 16 B | tables [kWaves][3][32] float
  uint4* const metal = reinterpret_cast<uint4*>(lds + kStages * kStageBytes);
  float* const tabs = reinterpret_cast<float*>(metal + kMeta * kTile);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds, ldsm = lds0 + kStages * kStageBytes;

  const int nbx = a.OW / kBlkW, nblk = (a.OH / kBlkH) * nbx;
  const long ntiles = (a.n + kTile - 1) / kTile;
  const long units = (long)nblk * ntiles;
  const long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  if (u0 >= u1) return;
  const float oscale = srlh2::h2_scale_for(*a.bound);
  if (blockIdx.x == 0 && tid == 0) *a.y_scale = oscale;

  // DMA: one instruction = one sample's 36 chunks (3 rows x 12), lane = chunk
  const bool dma_lane = lane < kChunks;
  const uint32_t dvoff = (uint32_t)((lane / kRowChunks) * (a.GW * 64) + (lane % kRowChunks) * 16);
  // fragment reads: patch bytes 32 c + 16 h of this wave's position (py, px) of the block: row py + (c >> 2), chunk 4 px + 2 (c & 3) + h
  const int py = wave / kBlkW, px = wave % kBlkW;
  const uint32_t rdbase = (uint32_t)(l31 * kSampleBytes + (py * kRowChunks + px * 4 + h) * 16);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.y_h2, 0, (int)(a.n * (long)a.P * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(a.y_mask, 0, (int)(a.n * (long)a.P * 4), 0x00020000);

  uint4 wf[32];  // [k-block][piece]: this position's folded weights, A-operand fragments
  int cur_blk = -1, pos = 0, ent = 0;
  float* const tb = tabs + wave * 96;
  int issued = 0;                        // VMEM operations this wavefront has issued through asm / buffer builtins
  int mark0 = 0, mark1 = 0, mark2 = 0;   // `issued` right after the DMA of the tile in that stage
  const uint4* const wq = a.wq;
  const float* const g_winv = a.winv;
  const float* const g_S = a.S;
  const float* const g_b2 = a.b2;
  const int OW = a.OW, OH = a.OH, P = a.P;
  const uint8_t* const frames = a.frames;
  const uint4* const meta = a.meta;
  const long img_stride = a.img_stride, nsamp = a.n;
  const int GW = a.GW;
  // Cursors instead of divisions (a 64-bit division is hundreds of instructions): (block, tile) of the tile being computed, of
  // the tile whose frames are fetched (two ahead) and the tile whose records are fetched (four ahead), and their ring entries
  struct Cur { int blk, tile, ent; };
  auto advance = [&](Cur& c) __attribute__((always_inline)) {
    if (++c.tile == (int)ntiles) { c.tile = 0; ++c.blk; }
    if (++c.ent == kMeta) c.ent = 0;
  };
  Cur cc{(int)(u0 / ntiles), (int)(u0 % ntiles), 0}, cd = cc, cm = cc;
  // the tile's 32 records -> ring entry: wavefront 0 only, lanes 0..31
#define SRL_OBS_META()                                                                                \
  do {                                                                                                \
    if (wave == 0) {                                                                                  \
      if (lane < kTile) obs_dma(ldsm + (uint32_t)cm.ent * (kTile * 16), meta + (long)cm.tile * kTile + lane); \
      issued += 1;                                                                                    \
    }                                                                                                 \
    advance(cm);                                                                                      \
  } while (0)
  // the tile's frames: 8 samples per wavefront, the slot read from the ring (the same address in every lane: a broadcast)
#define SRL_OBS_DMA(STAGE)                                                                            \
  do {                                                                                                \
    const long blkoff_ = ((long)((cd.blk / nbx) * kBlkH) * GW + (cd.blk % nbx) * kBlkW) * 64 + dvoff; \
    constexpr int per_ = kTile / kWaves;                                                              \
    const uint4* m_ = metal + cd.ent * kTile + per_ * wave;                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < per_; ++i_) {                                                \
      const uint32_t slot_ = m_[i_].x;                    \
      if (dma_lane && !(DBG & 4))                                                                     \
        obs_dma(lds0 + (STAGE) * kStageBytes + (per_ * wave + i_) * kSampleBytes, frames + (long)slot_ * img_stride + blkoff_); \
    }                                                                                                 \
    issued += per_;                                                                                   \
    if ((STAGE) == 0) mark0 = issued;                                                                 \
    else if ((STAGE) == 1) mark1 = issued;                                                            \
    else mark2 = issued;                                                                              \
    advance(cd);                                                                                      \
  } while (0)

  // epilogue of a tile: lane = sample 32 tilep + l31 (both halves), register r = channel (r & 3) + 8 (r >> 2) + 4 h
#define SRL_OBS_FINISH(accp, rvp, mvp, tilep)                                                                                                \
  do {                                                                                                                    \
    const long n0_ = (long)tilep * kTile;                                                                                 \
    const bool ok_ = n0_ + l31 < nsamp;                                                                                   \
    float v_[16];                                                                                                         \
    uint32_t bits_ = 0;                                                                                                   \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                                    \
      const float4 i4 = *reinterpret_cast<const float4*>(tb + 8 * g4 + 4 * h);                                            \
      const float4 s4 = *reinterpret_cast<const float4*>(tb + 32 + 8 * g4 + 4 * h);                                       \
      const float4 b4 = *reinterpret_cast<const float4*>(tb + 64 + 8 * g4 + 4 * h);                                       \
      const float iv[4] = {i4.x, i4.y, i4.z, i4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};    \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                     \
        float t = fmaf(rvp * iv[i], accp[4 * g4 + i], fmaf(mvp, sv[i], bv[i]));                                           \
        if (ACT == 1) t = fmaxf(t, 0.f);                                                                                  \
        else if (ACT == 2) t = tanhf(t);                                                                                  \
        v_[4 * g4 + i] = t;                                                                                               \
        if (ok_) amx = fmaxf(amx, fabsf(t));                                                                              \
        bits_ |= (t > 0.f ? 1u : 0u) << (8 * g4 + 4 * h + i);                                                             \
      }                                                                                                                   \
    }                                                                                                                     \
    uint4 c4[4];                                                                                                          \
    srlh2::h2_split_pair(v_[0], v_[1], oscale, c4[0].x, c4[1].x);                                                         \
    srlh2::h2_split_pair(v_[2], v_[3], oscale, c4[0].y, c4[1].y);                                                         \
    srlh2::h2_split_pair(v_[4], v_[5], oscale, c4[0].z, c4[1].z);                                                         \
    srlh2::h2_split_pair(v_[6], v_[7], oscale, c4[0].w, c4[1].w);                                                         \
    srlh2::h2_split_pair(v_[8], v_[9], oscale, c4[2].x, c4[3].x);                                                         \
    srlh2::h2_split_pair(v_[10], v_[11], oscale, c4[2].y, c4[3].y);                                                       \
    srlh2::h2_split_pair(v_[12], v_[13], oscale, c4[2].z, c4[3].z);                                                       \
    srlh2::h2_split_pair(v_[14], v_[15], oscale, c4[2].w, c4[3].w);                                                       \
    /* rows past the batch: an offset beyond the buffer drops the store -- the same number of stores whatever the tile.     \
       (Whole 128-byte lines per instruction, through an LDS transposition, were timed equal: the L2 merges the pieces;      \
       non-temporal stores, which do not merge, 15-65 % slower.) */                                                        \
    const uint32_t ooff = (ok_ && !(DBG & 1)) ? (uint32_t)(((n0_ + l31) * (long)P + ent) * 128 + h * 64) : 0x80000000u;    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                                         \
      u32x4 d = {c4[i].x, c4[i].y, c4[i].z, c4[i].w};                                                                     \
      __builtin_amdgcn_raw_buffer_store_b128(d, r_out, ooff + 16 * i, 0, 0);                                              \
    }                                                                                                                     \
    bits_ |= (uint32_t)__shfl_xor((int)bits_, 32);                                                                        \
    const uint32_t moff = (ok_ && h == 0) ? (uint32_t)(((n0_ + l31) * (long)P + pos) * 4) : 0x80000000u;                 \
    __builtin_amdgcn_raw_buffer_store_b32(bits_, r_msk, moff, 0, 0);                                                      \
    issued += 5;                                                                                                          \
  } while (0)

  float amx = 0.f;
  const int nu = (int)(u1 - u0);
  for (int m = 0; m < 4 && m < nu; ++m) SRL_OBS_META();
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  SRL_OBS_DMA(0);
  if (nu > 1) SRL_OBS_DMA(1);
  int stage = 0;
  for (int it = 0; it < nu; ++it) {
    obs_wait_vm_dyn(issued - (stage == 0 ? mark0 : stage == 1 ? mark1 : mark2));
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (it + 4 < nu) SRL_OBS_META();               // (ahead of the frames below: their mark covers it two tiles on)
    if (it + 2 < nu) {
      if (stage == 0) SRL_OBS_DMA(2);              // into the stage the previous tile has just left
      else if (stage == 1) SRL_OBS_DMA(0);
      else SRL_OBS_DMA(1);
    }
    const int blk = cc.blk;
    if (blk != cur_blk) {
      const int oy = (blk / nbx) * kBlkH + py, ox = (blk % nbx) * kBlkW + px;
      pos = oy * OW + ox;
      ent = ((oy & 1) * 2 + (ox & 1)) * ((OH / 2) * (OW / 2)) + (oy >> 1) * (OW / 2) + (ox >> 1);
      const uint4* src = wq + (long)pos * 2 * 16 * 64 + lane;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wf[2 * kb + pl] = src[(pl * 16 + kb) * 64];
      if (lane < 32) {
        tb[lane] = g_winv[pos * kCout + lane];
        tb[32 + lane] = g_S[pos * kCout + lane];
        tb[64 + lane] = g_b2[pos * kCout + lane];
      }
      cur_blk = blk;
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wf[i].x), "+v"(wf[i].y), "+v"(wf[i].z), "+v"(wf[i].w));
    }
    const int tile_now = cc.tile;
    const uint8_t* sb = lds + stage * kStageBytes + rdbase;
    const uint4 mrec = metal[cc.ent * kTile + l31];
    const float rv = __uint_as_float(mrec.y), mean_n = __uint_as_float(mrec.z);
    const float cen = rintf(mean_n);
    const float mv = -(mean_n - cen) * rv;
    union { _Float16 f[2]; uint32_t u; } nc;
    nc.f[0] = nc.f[1] = (_Float16)(-(1024.f + cen));
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < ((DBG & 2) ? 0 : 8); ++c) {
      const uint4 q = *reinterpret_cast<const uint4*>(sb + (c >> 2) * (kRowChunks * 16) + (c & 3) * 32);
      union { uint32_t u[4]; f16x8 v; } x0, x1;
      obs_bytes_to_f16(q.x, nc.u, x0.u[0], x0.u[1]);
      obs_bytes_to_f16(q.y, nc.u, x0.u[2], x0.u[3]);
      obs_bytes_to_f16(q.z, nc.u, x1.u[0], x1.u[1]);
      obs_bytes_to_f16(q.w, nc.u, x1.u[2], x1.u[3]);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          union { uint4 u; f16x8 v; } w;
          w.u = wf[2 * (2 * c + e) + pl];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.v, e ? x1.v : x0.v, acc, 0, 0, 0);
        }
    }
    SRL_OBS_FINISH(acc, rv, mv, tile_now);
    advance(cc);
    stage = stage == kStages - 1 ? 0 : stage + 1;
  }
  if (a.y_absmax) srlgemm::absmax_commit(a.y_absmax, amx);
#undef SRL_OBS_META
#undef SRL_OBS_DMA
#undef SRL_OBS_FINISH
}
#endif  // __HIPCC__

}  // namespace srlobs
