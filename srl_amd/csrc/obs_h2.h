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
constexpr int kMeta = 8;                     // ring of per-sample record tiles: fetched five tiles ahead of the one computed

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
  int GW, OW, OH, P, act, nsplit;
  int xcd;   // 1: XCD-contiguous (block, split) numbering (SRL_OBS_XCD=1; default: by blockIdx)
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
// rstd_max (optional, zeroed by the caller): the largest rstd of the launch, for the weight gradient's scale
__global__ __launch_bounds__(256) void obs_meta_kernel(const int32_t* row_index, const float* mean, const float* rstd, long n,
                                                       long n_pad, uint4* meta, float* rstd_max = nullptr) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float rs = 0.f;
  if (i < n_pad) {
    const long smp = i < n ? i : n - 1;
    const long slot = row_index ? (long)row_index[smp] : smp;
    rs = rstd[slot];
    const float mu = mean[slot];
    union { _Float16 f[2]; uint32_t u; } nc;   // -(1024 + round(mean)) twice: what the byte -> f16 conversions add
    nc.f[0] = nc.f[1] = (_Float16)(-(1024.f + rintf(mu)));
    meta[i] = make_uint4((uint32_t)slot, __float_as_uint(rs), __float_as_uint(mu), nc.u);
  }
  if (rstd_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rs = fmaxf(rs, __shfl_xor(rs, o));
    if ((threadIdx.x & 63) == 0 && rs > 0.f) atomicMax(reinterpret_cast<int*>(rstd_max), __float_as_int(rs));
  }
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

// the same from a wavefront-uniform base (scalar registers) + the lane's 32-bit byte offset: no 64-bit vector arithmetic
__device__ __forceinline__ void obs_dma_s(uint32_t lds, const void* sbase, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds)), "v"(voff), "s"(sbase)
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

#ifndef SRL_OBS_LINE_STORES
#define SRL_OBS_LINE_STORES 1
#endif
// ACT: the activation (0 none, 1 ReLU, 2 tanh).  DBG (timing experiments, wrong results; SRL_OBS_DBG): 1 = no output stores, 2 = no LDS
// reads / conversions / MFMAs, 4 = no DMA
template <int ACT, int DBG = 0>
__global__ __launch_bounds__(64 * kWaves, kWaves == 4 ? 2 : 1) void obs_fwd_h2_kernel(FwdH2Args a) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  // [kStages][32 samples][kChunks + 1 chunks][16 B] | meta ring [kMeta][32] x 16 B | tables [kWaves][3][32] float
  uint4* const metal = reinterpret_cast<uint4*>(lds + kStages * kStageBytes);
  float* const tabs = reinterpret_cast<float*>(metal + kMeta * kTile);
  uint8_t* const trbuf = reinterpret_cast<uint8_t*>(tabs + kWaves * 96);   // SRL_OBS_LINE_STORES: [kWaves][32 rows][144 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds, ldsm = lds0 + kStages * kStageBytes;

  // a workgroup = one block of positions x one of nsplit ranges of the launch's 32-sample tiles
  const int nbx = a.OW / kBlkW;
  // a.xcd (opt-in, SRL_OBS_XCD=1): every XCD owns one contiguous run of (block, split) pairs -- the ~6 blocks an XCD works on are
  // neighbours in the frame, so that the window rows and columns they share come out of ITS L2 (workgroup ids go round-robin over
  // the eight XCDs: numbered by blockIdx alone, the 50 blocks of a sample range are spread over all eight L2s and every frame is
  // fetched 2.2 times).  A third fewer bytes fetched, no time gained (conv.hip: obs_xcd_order)
  unsigned lid;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    lid = a.xcd ? xcd * q + (xcd < r ? xcd : r) + slot : blockIdx.x;
  }
  const int blk = (int)(lid / (unsigned)a.nsplit), split = (int)(lid % (unsigned)a.nsplit);
  const long ntiles = (a.n + kTile - 1) / kTile;
  const long t0 = ntiles * split / a.nsplit, t1 = ntiles * (split + 1) / a.nsplit;
  const int nu = (int)(t1 - t0);
  if (nu <= 0) return;
  const float oscale = srlh2::h2_scale_for(*a.bound);
  if (blockIdx.x == 0 && tid == 0) *a.y_scale = oscale;

  // DMA: one instruction = one sample's 60 chunks (3 rows x 20) of the block's frame window, lane = chunk
  const bool dma_lane = lane < kChunks;
  const uint32_t blkoff = (uint32_t)(((blk / nbx) * kBlkH * a.GW + (blk % nbx) * kBlkW) * 64 +
                                     (lane / kRowChunks) * (a.GW * 64) + (lane % kRowChunks) * 16);
  // fragment reads: patch bytes 32 c + 16 h of this wave's position (py, px) of the block: row py + (c >> 2), chunk 4 px + 2 (c & 3) + h
  const int py = wave / kBlkW, px = wave % kBlkW;
  const uint32_t rdbase = (uint32_t)(l31 * kSampleBytes + (py * kRowChunks + px * 4 + h) * 16);
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.y_h2, 0, (int)(a.n * (long)a.P * 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_msk = __builtin_amdgcn_make_buffer_rsrc(a.y_mask, 0, (int)(a.n * (long)a.P * 4), 0x00020000);

  float* const tb = tabs + wave * 96;
  const int P = a.P;
  const uint8_t* const frames = a.frames;
  const uint4* const meta = a.meta + t0 * kTile;
  const long img_stride = a.img_stride, nsamp = a.n;
  const int oy = (blk / nbx) * kBlkH + py, ox = (blk % nbx) * kBlkW + px;
  const int pos = oy * a.OW + ox;
  const int ent = ((oy & 1) * 2 + (ox & 1)) * ((a.OH / 2) * (a.OW / 2)) + (oy >> 1) * (a.OW / 2) + (ox >> 1);
  // this position's folded weights, A-operand fragments [k-block][piece]: 128 registers for the whole launch
  uint4 wf[32];
  {
    const uint4* src = a.wq + (long)pos * 2 * 16 * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) wf[2 * kb + pl] = src[(pl * 16 + kb) * 64];
    if (lane < 32) {
      tb[lane] = a.winv[pos * kCout + lane];
      tb[32 + lane] = a.S[pos * kCout + lane];
      tb[64 + lane] = a.b2[pos * kCout + lane];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wf[i].x), "+v"(wf[i].y), "+v"(wf[i].z), "+v"(wf[i].w));
  }
  // Every tile issues the same memory operations in the same order -- [records of tile j + 5 (wavefront 0)] [frames of tile
  // j + 2: four samples per wavefront] [five stores of tile j] -- past the end of the range the last tile's again, so that the
  // s_waitcnt in front of a tile's frames is a constant.  (It was a count kept in registers and a 25-way branch tree per wait,
  // with cursors over (block, tile) units: 147 us of the kernel's 447 ran without any load, store or MFMA in it.)
  auto issue_meta = [&](int j) __attribute__((always_inline)) {   // records of tile j -> ring entry j & 7
    if (wave == 0) {
      const int jc = j < nu ? j : nu - 1;
      if (lane < kTile) obs_dma(ldsm + (uint32_t)(j & 7) * (kTile * 16), meta + jc * kTile + lane);
    }
  };
  // slots of the wavefront's four samples of tile j, from the ring (uniform addresses: broadcasts) -> scalar registers
  uint32_t slot[4] = {0, 0, 0, 0};
  auto read_slots = [&](int j) __attribute__((always_inline)) {
    const uint4* m_ = metal + (j & 7) * kTile + 4 * wave;
#pragma unroll
    for (int i = 0; i < 4; ++i) slot[i] = __builtin_amdgcn_readfirstlane(m_[i].x);
  };
  auto issue = [&](int j, int stage_j) __attribute__((always_inline)) {
    issue_meta(j + 3);
    if (dma_lane && !(DBG & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        obs_dma_s(lds0 + stage_j * kStageBytes + (4 * wave + i) * kSampleBytes, frames + (long)slot[i] * img_stride, blkoff);
    }
  };
#define SRL_OBS_WAIT(N0, N)                                                                 \
  do {                                                                                      \
    if (wave == 0) asm volatile("s_waitcnt vmcnt(" #N0 ")" ::: "memory");                   \
    else asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                              \
  } while (0)

  // epilogue of a tile: lane = sample 32 tilep + l31 (both halves), register r = channel (r & 3) + 8 (r >> 2) + 4 h
#define SRL_OBS_FINISH(accp, rvp, mvp, tilep)                                                                                                \
  do {                                                                                                                    \
    typedef float f2_ __attribute__((ext_vector_type(2)));                                                                \
    typedef _Float16 hh2_ __attribute__((ext_vector_type(2)));                                                            \
    const long n0_ = (long)tilep * kTile;                                                                                 \
    const bool ok_ = n0_ + l31 < nsamp;                                                                                   \
    uint32_t bits_ = 0;                                                                                                   \
    uint32_t c4[4][4];                                                                                                    \
    /* pairs of channels on the packed float32 instructions: 8 vector instructions per value (14 one value at a time; the  \
       vector and the matrix instructions of a SIMD do not overlap here, see the weight gradient below) */                 \
    _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4) {                                                                    \
      const float4 i4 = *reinterpret_cast<const float4*>(tb + 8 * g4 + 4 * h);                                            \
      const float4 s4 = *reinterpret_cast<const float4*>(tb + 32 + 8 * g4 + 4 * h);                                       \
      const float4 b4 = *reinterpret_cast<const float4*>(tb + 64 + 8 * g4 + 4 * h);                                       \
      const f2_ iv[2] = {{i4.x, i4.y}, {i4.z, i4.w}}, sv[2] = {{s4.x, s4.y}, {s4.z, s4.w}}, bv[2] = {{b4.x, b4.y}, {b4.z, b4.w}}; \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                     \
        const f2_ a2 = {accp[4 * g4 + 2 * i], accp[4 * g4 + 2 * i + 1]};                                                  \
        f2_ t = (iv[i] * rvp) * a2 + (sv[i] * mvp + bv[i]);   /* contracted to fused multiply-adds, as one value at a time */ \
        const int k_ = 8 * g4 + 4 * h + 2 * i;                                                                            \
        if (ACT == 1) {                                                                                                   \
          t[0] = fmaxf(t[0], 0.f); t[1] = fmaxf(t[1], 0.f);                                                               \
          /* rows past the batch repeat its last sample (obs_meta_kernel): their values change no maximum */              \
          amx = __builtin_fmaxf(__builtin_fmaxf(amx, t[0]), t[1]);                                                        \
          /* t >= +0: positive <=> a non-zero word */                                                                      \
          const uint32_t p0_ = __float_as_uint(t[0]) < 1u ? __float_as_uint(t[0]) : 1u, p1_ = __float_as_uint(t[1]) < 1u ? __float_as_uint(t[1]) : 1u; \
          bits_ |= p0_ << k_;                                                                                             \
          bits_ |= p1_ << (k_ + 1);                                                                                       \
        } else {                                                                                                          \
          if (ACT == 2) { t[0] = tanhf(t[0]); t[1] = tanhf(t[1]); }                                                       \
          amx = fmaxf(amx, fmaxf(fabsf(t[0]), fabsf(t[1])));                                                              \
          bits_ |= (t[0] > 0.f ? 1u : 0u) << k_;                                                                          \
          bits_ |= (t[1] > 0.f ? 1u : 0u) << (k_ + 1);                                                                    \
        }                                                                                                                 \
        /* the two f16 pieces of the pair (srlh2::h2_split_pair, packed): t s exact (a power of two), h0 rounded, the     \
           residual exact, h1 rounded */                                                                                  \
        const f2_ vs = t * oscale;                                                                                        \
        union { hh2_ v; uint32_t u; } h0_, h1_;                                                                           \
        h0_.v = __builtin_convertvector(vs, hh2_);                                                                        \
        const f2_ r_ = vs - f2_{(float)h0_.v[0], (float)h0_.v[1]};                                                        \
        h1_.v = __builtin_convertvector(r_, hh2_);                                                                        \
        /* value 4 g4 + 2 i (+ 1) of the lane: pair j = 2 g4 + i -> word (j & 3) of chunk 2 (j >> 2) (h0) / + 1 (h1) */     \
        c4[2 * ((2 * g4 + i) >> 2)][(2 * g4 + i) & 3] = h0_.u;                                                            \
        c4[2 * ((2 * g4 + i) >> 2) + 1][(2 * g4 + i) & 3] = h1_.u;                                                        \
      }                                                                                                                   \
    }                                                                                                                     \
    /* rows past the batch: an offset beyond the buffer drops the store -- the same number of stores whatever the tile.     \
       (Non-temporal stores, which do not merge in L2: 15-65 % slower.  Whole 128-byte rows per instruction were timed equal   \
       on the first block kernel; on this one, whose loads, stores and compute add up, 449 -> 387 us.) */                     \
    if (SRL_OBS_LINE_STORES) {                                                                                            \
      /* through the wavefront's own LDS rows (144-byte pitch: conflict-free 16-byte slots): an instruction then stores 8     \
         whole 128-byte rows instead of two 16-byte pieces of 32 rows */                                                     \
      uint8_t* tr_ = trbuf + wave * (kTile * 144);                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                       \
        *reinterpret_cast<uint4*>(tr_ + l31 * 144 + h * 64 + 16 * i) = make_uint4(c4[i][0], c4[i][1], c4[i][2], c4[i][3]); \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                     \
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                                       \
        const int smp_ = 8 * i + (lane >> 3);                                                                             \
        const uint4 q_ = *reinterpret_cast<const uint4*>(tr_ + smp_ * 144 + (lane & 7) * 16);                             \
        const uint32_t oo_ = (n0_ + smp_ < nsamp && !(DBG & 1)) ? (uint32_t)(((n0_ + smp_) * (long)P + ent) * 128 + (lane & 7) * 16) : 0x80000000u; \
        u32x4 d = {q_.x, q_.y, q_.z, q_.w};                                                                               \
        __builtin_amdgcn_raw_buffer_store_b128(d, r_out, oo_, 0, 0);                                                      \
      }                                                                                                                   \
    } else {                                                                                                              \
    const uint32_t ooff = (ok_ && !(DBG & 1)) ? (uint32_t)(((n0_ + l31) * (long)P + ent) * 128 + h * 64) : 0x80000000u;    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));                                                         \
      u32x4 d = {c4[i][0], c4[i][1], c4[i][2], c4[i][3]};                                                                 \
      __builtin_amdgcn_raw_buffer_store_b128(d, r_out, ooff + 16 * i, 0, 0);                                              \
    }                                                                                                                     \
    }                                                                                                                     \
    bits_ |= (uint32_t)__shfl_xor((int)bits_, 32);                                                                        \
    const uint32_t moff = (ok_ && h == 0) ? (uint32_t)(((n0_ + l31) * (long)P + pos) * 4) : 0x80000000u;                 \
    __builtin_amdgcn_raw_buffer_store_b32(bits_, r_msk, moff, 0, 0);                                                      \
  } while (0)

  float amx = 0.f;
  for (int m = 0; m < 5; ++m) issue_meta(m);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int j = 0; j < 2; ++j) {   // the two tiles ahead of the loop, with the five (dropped) stores of a tile each: the same counts
    read_slots(j);
    issue(j, j);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, r_out, 0x80000000u + 16 * i, 0, 0);   // (distinct: not merged)
    }
    __builtin_amdgcn_raw_buffer_store_b32(0u, r_msk, 0x80000000u, 0, 0);
  }
  read_slots(2);
  int stage = 0;
  for (int it = 0; it < nu; ++it) {
    SRL_OBS_WAIT(15, 14);  // behind the frames of tile it: the stores of it - 2 and one tile's [records] frames stores
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(it + 2, stage == 0 ? 2 : stage - 1);   // into the stage the previous tile has just left
    const int tile_now = (int)t0 + it;
    const uint8_t* sb = lds + stage * kStageBytes + rdbase;
    const uint4 mrec = metal[(it & 7) * kTile + l31];
    const float rv = __uint_as_float(mrec.y), mean_n = __uint_as_float(mrec.z);
    const float mv = -(mean_n - rintf(mean_n)) * rv;
    const uint32_t ncu = mrec.w;   // -(1024 + round(mean)) as two f16
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < ((DBG & 2) ? 0 : 8); ++c) {
      const uint4 q = *reinterpret_cast<const uint4*>(sb + (c >> 2) * (kRowChunks * 16) + (c & 3) * 32);
      union { uint32_t u[4]; f16x8 v; } x0, x1;
      obs_bytes_to_f16(q.x, ncu, x0.u[0], x0.u[1]);
      obs_bytes_to_f16(q.y, ncu, x0.u[2], x0.u[3]);
      obs_bytes_to_f16(q.z, ncu, x1.u[0], x1.u[1]);
      obs_bytes_to_f16(q.w, ncu, x1.u[2], x1.u[3]);
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
    read_slots(it + 3);  // for the next tile's fetch (its records came in five tiles ahead): no LDS round trip behind the barrier
    stage = stage == kStages - 1 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.y_absmax) srlgemm::absmax_commit(a.y_absmax, amx);
#undef SRL_OBS_WAIT
#undef SRL_OBS_FINISH
}
#endif  // __HIPCC__

// ---- weight gradient in the same block structure -------------------------------------------------------------------------------
// Q[pos][o][k] = sum_n dz'[n,pos,o] (x[n,pos,k] - c_n), R = sum dz, C = sum dz' (mean - c) as obs_bwd_bf16_kernel (obs_bf16.h), for
// the finalisation kernels of conv.hip -- but a workgroup owns a 2 x 4 block of positions and ONE nsplit-th of the samples of the
// launch: the frames come in once per block by LDS-DMA as BYTES (the forward's tiles, 16 samples each) and go to the matrix cores
// through ds_read_b64_tr_b8, the transposing byte read of gfx950 -- of 16 row addresses x 8 bytes a lane receives one byte column
// of the even (lanes 0-7) or odd (lanes 8-15) addresses: with even / odd addresses pointing at the same 8 samples, 8 bytes apart,
// ONE read hands a wavefront the 32-column x 16-sample B operand of a 32x32x16 MFMA (8 bytes per lane = 8 consecutive samples of
// its column), converted in registers (a byte minus the sample's integer centre is exact in f16).  obs_bf16.h converts while
// STAGING and keeps a 16-bit image of every patch in LDS (4 x the bytes; 63 % of its LDS cycles were bank conflicts).  dz' =
// dz rstd_n is the A operand: the float32 rows come in by LDS-DMA too (three tiles ahead), are split into kPiecesB = 2 f16 pieces
// under the power-of-two scale of the bound max |dz| max rstd -- what every h2 operand of the update carries (h2gemm.h: 22
// significand bits; the bytes are exact, so a product drops nothing else) -- into a per-wavefront image, and fetched with
// ds_read_b64_tr_b16.  A wavefront = one position, all 256 patch columns: 8 accumulator tiles, 16 MFMAs per 16 samples.
// (Three pieces: + 51 us of MFMA time per 16 384-frame launch for 2^-33 instead of 2^-22 of an element; the vector and the matrix
// instructions of a SIMD's two wavefronts were measured NOT to overlap here -- MFMA-only 153 us + vector-only 98 us = 324 us of
// the launch without memory traffic, whether interleaved inside each wavefront's stream, alternated between the two wavefronts
// of a SIMD, or with the accumulators in AGPRs -- so work removed is time removed.)
constexpr int kTileB = 16;                                  // samples per tile of the weight gradient (= the MFMA's k)
constexpr int kSampleBytesB = (kChunks + 2) * 16;           // 62 chunks = 248 dwords = 8 x 31 banks: 8 samples x 32 bytes of a read
constexpr int kStageBytesB = kTileB * kSampleBytesB;        //   of 32 lanes fall on 64 different banks
constexpr int kPiecesB = 2;                                 // f16 pieces of dz' (as every h2 operand: 22 significand bits)
constexpr int kStagesB = 4;                                 // frames and dz rows are fetched three tiles ahead
constexpr int kMetaB = 8;                                   // records seven tiles ahead: in LDS a tile before the frames' fetch reads the slots
constexpr int kLdsB = kStagesB * kStageBytesB + kMetaB * kTileB * 16 + kWaves * (kPiecesB * 1024 + kStagesB * 2048 + 64);

struct BwdH2Args {
  const uint8_t* frames;
  long img_stride;
  const uint4* meta;        // [ceil(n / 32) * 32] x {slot, rstd, mean, -(1024 + round(mean)) as two f16}
  long n;
  const float* dz;          // [n][P][32]
  const float* dz_bound;    // device float: max |dz|
  const float* rstd_bound;  // device float: max rstd over the launch's samples (obs_meta_kernel)
  float* Q;                 // [nsplit][P][32][256] slabs
  long slab;
  float* R;                 // [P][32], atomically accumulated
  float* C;                 // [P][32], atomically accumulated
  int GW, OW, OH, P, nsplit;
  int xcd;
};

#ifdef __HIPCC__
// The schedule of a workgroup, tile it of its sample range: [wait: all but the last tile's fetch landed | barrier | fetch of tile
// it + 3 (records of it + 7, frames, dz rows) | the MFMAs of tile it, whose A fragments and first B operands are in registers when
// the barrier opens; in their gaps the conversions of the next B operands, dz(it + 1): float32 rows -> f16 planes -> the next
// tile's A fragments, and the LDS reads of what comes after].  Every tile issues the SAME number of memory operations in the same
// order (past the end: the last tile again), which makes the s_waitcnt of a tile a compile-time constant.
// DBG (timing experiments, wrong results): 1 no frame DMA, 2 no dz DMA, 4 no MFMAs, 8 dz staged as zeros, 16 no vector work between the MFMAs
template <int DBG = 0>
__global__ __launch_bounds__(512, 1) void obs_bwd_h2_kernel(BwdH2Args a) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  // [4][16 samples][62 chunks][16 B] | record ring [8][16] x 16 B | per wavefront: 3 planes [16 samples][32 channels] f16 | per
  // wavefront: 4 x [16 samples][32 channels] float32 (dz as it comes)
  uint4* const metal = reinterpret_cast<uint4*>(lds + kStagesB * kStageBytesB);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds, ldsm = lds0 + kStagesB * kStageBytesB;
  constexpr int kPlanes0 = kStagesB * kStageBytesB + kMetaB * kTileB * 16;
  uint8_t* const myp = lds + kPlanes0 + wave * (kPiecesB * 1024);
  constexpr int kRaw0 = kPlanes0 + kWaves * (kPiecesB * 1024);
  const uint8_t* const myraw = lds + kRaw0 + wave * (kStagesB * 2048);
  const uint32_t ldsraw = lds0 + kRaw0 + wave * (kStagesB * 2048);

  const int nbx = a.OW / kBlkW;
  unsigned lid;   // (XCD-contiguous numbering: see the forward kernel)
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    lid = a.xcd ? xcd * q + (xcd < r ? xcd : r) + slot : blockIdx.x;
  }
  const int blk = (int)(lid / (unsigned)a.nsplit), split = (int)(lid % (unsigned)a.nsplit);
  const long ntiles = (a.n + kTileB - 1) / kTileB;
  const long t0 = ntiles * split / a.nsplit, t1 = ntiles * (split + 1) / a.nsplit;
  const int nu = (int)(t1 - t0);
  if (nu <= 0) return;
  const int py = wave / kBlkW, px = wave % kBlkW;
  const int oy = (blk / nbx) * kBlkH + py, ox = (blk % nbx) * kBlkW + px;
  const int pos = oy * a.OW + ox;
  const float scale = srlh2::h2_scale_for(*a.dz_bound * *a.rstd_bound), inv_scale = 1.0f / scale;

  const bool dma_lane = lane < kChunks;
  const uint32_t blkoff = (uint32_t)(((blk / nbx) * kBlkH * a.GW + (blk % nbx) * kBlkW) * 64 +
                                     (lane / kRowChunks) * (a.GW * 64) + (lane % kRowChunks) * 16);
  const uint8_t* const frames = a.frames;
  const uint4* const meta = a.meta + t0 * kTileB;
  const long img_stride = a.img_stride, nsamp = a.n;
  const uint32_t ldzb = (uint32_t)a.P * kCout * 4;          // bytes per sample of dz (the whole tensor: < 4 GiB, checked by the host)
  const float* const dzpos = a.dz + (long)pos * kCout;      // wavefront-uniform: the position's first row
  const uint32_t dzlane = (uint32_t)(lane & 7) * 16;

  // B operand (bytes): lane L of group g = L >> 4 supplies the address of sample 8 (g >> 1) + ((L & 15) >> 1), byte column
  // 16 (g & 1) + 8 (L & 1) of the 32-column tile; patch tile t: row t >> 2 of the patch, bytes 32 (t & 3) .. + 31 of its 128
  const int g16 = lane >> 4, j16 = lane & 15;
  const uint32_t bbase = (uint32_t)((8 * (g16 >> 1) + (j16 >> 1)) * kSampleBytesB + (py * kRowChunks + px * 4) * 16 + 16 * (g16 & 1) + 8 * (j16 & 1));
  // A operand (16-bit planes, 64-byte rows): ds_read_b64_tr_b16, lane 4 q + p of a group supplies row q, columns 4 p .. 4 p + 3
  const int q4 = (lane >> 2) & 3, p4 = lane & 3;
  const uint8_t* const pa = myp + (8 * h + q4) * 64 + (16 * (g16 & 1) + 4 * p4) * 2;
  // dz staging: lane -> sample lane >> 2 of the tile, channels 8 (lane & 3) .. + 7
  const int ds_s = lane >> 2, ds_q = lane & 3;
  uint8_t* const pdst = myp + ds_s * 64 + ds_q * 16;
  const uint8_t* const praw = myraw + ds_s * 128 + ds_q * 32;

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x2 rsum[4], csum[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rsum[i] = csum[i] = f32x2{0.f, 0.f};

  // records of tile j -> ring entry j & 7 (wavefront 0, lanes 0..15)
  auto issue_meta = [&](int j) __attribute__((always_inline)) {
    if (wave == 0) {
      const int jc = j < nu ? j : nu - 1;
      if (lane < kTileB) obs_dma(ldsm + (uint32_t)(j & 7) * (kTileB * 16), meta + jc * kTileB + lane);
    }
  };
  // slots of the wavefront's two samples of tile j, from the ring (uniform address: a broadcast) -> scalar registers
  uint32_t slot0 = 0, slot1 = 0;
  auto read_slots = [&](int j) __attribute__((always_inline)) {
    const uint4* m_ = metal + (j & 7) * kTileB + 2 * wave;
    slot0 = __builtin_amdgcn_readfirstlane(m_[0].x);
    slot1 = __builtin_amdgcn_readfirstlane(m_[1].x);
  };
  // tile j: records of j + 4, frames (2 samples per wavefront, slots read before), dz rows (2 x 8 samples x 128 bytes)
  auto issue = [&](int j) __attribute__((always_inline)) {
    issue_meta(j + 4);
    if (dma_lane && !((DBG & 1) && j > 3)) {
      obs_dma_s(lds0 + (j & 3) * kStageBytesB + (2 * wave) * kSampleBytesB, frames + (long)slot0 * img_stride, blkoff);
      obs_dma_s(lds0 + (j & 3) * kStageBytesB + (2 * wave + 1) * kSampleBytesB, frames + (long)slot1 * img_stride, blkoff);
    }
    const int jc = j < nu ? j : nu - 1;
    const uint32_t smp0 = (uint32_t)((t0 + jc) * kTileB) + (lane >> 3), last = (uint32_t)(nsamp - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t smp = smp0 + 8 * i < last ? smp0 + 8 * i : last;
      if (!((DBG & 2) && j > 3)) obs_dma_s(ldsraw + (j & 3) * 2048 + i * 1024, dzpos, smp * ldzb + dzlane);
    }
  };
  // dz rows of tile j (raw image j & 3) -> three f16 planes; R / C sums.  In three steps (loads | 4 x a pair of channels | stores)
  // for the main loop to place between its MFMAs.
  float st_rss, st_mcs, st_okf;
  f32x2 st_d[4];
  uint32_t st_pl[kPiecesB][4];
  auto stage_load = [&](int j) __attribute__((always_inline)) {
    const uint4 mrec = metal[(j & 7) * kTileB + ds_s];
    const bool ok = (t0 + j) * kTileB + ds_s < nsamp && j < nu && !((DBG & 8) && j > 3);
    const float mean_n = __uint_as_float(mrec.z);
    st_rss = ok ? __uint_as_float(mrec.y) * scale : 0.f;   // dz -> scaled dz'; 0 past the end
    st_mcs = (mean_n - rintf(mean_n)) * inv_scale;         // scaled dz' -> its share of C
    st_okf = ok ? 1.f : 0.f;
    const float4* rp = reinterpret_cast<const float4*>(praw + (j & 3) * 2048);
    const float4 d0 = rp[0], d1 = rp[1];
    st_d[0] = f32x2{d0.x, d0.y}; st_d[1] = f32x2{d0.z, d0.w}; st_d[2] = f32x2{d1.x, d1.y}; st_d[3] = f32x2{d1.z, d1.w};
  };
  // a pair of channels in kPiecesB steps of four vector instructions
  typedef _Float16 st_h2t __attribute__((ext_vector_type(2)));
  f32x2 st_r;
  auto stage_unit = [&](int i, int u) __attribute__((always_inline)) {
    if (u >= kPiecesB) return;
    union { st_h2t hv; uint32_t w; } c;
    if (u == 0) {
      rsum[i] += st_d[i] * st_okf;
      st_r = st_d[i] * st_rss;
      csum[i] += st_r * st_mcs;
    } else {
      // the residual of a rounded piece is exact in float32 (v - h0 has fewer bits than v): 11 + 11 bits from the leading one down
      c.w = st_pl[u - 1][i];
      st_r = st_r - f32x2{(float)c.hv[0], (float)c.hv[1]};
    }
    c.hv = __builtin_convertvector(st_r, st_h2t);
    st_pl[u][i] = c.w;
  };
  auto stage_pair = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kPiecesB; ++u) stage_unit(i, u);
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < kPiecesB; ++p)
      *reinterpret_cast<uint4*>(pdst + p * 1024) = make_uint4(st_pl[p][0], st_pl[p][1], st_pl[p][2], st_pl[p][3]);
  };
  auto stage_dz = [&](int j) __attribute__((always_inline)) {
    stage_load(j);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_pair(i);
    stage_store();
  };
  // wave 0 issues one operation more per tile (the records)
#define SRL_OBSB_WAIT(N0, N)                                                                 \
  do {                                                                                       \
    if (wave == 0) asm volatile("s_waitcnt vmcnt(" #N0 ")" ::: "memory");                    \
    else asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");                               \
  } while (0)

  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef int v2i __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) v2i lds_v2i;
  typedef _Float16 h2t __attribute__((ext_vector_type(2)));
  // centres of a tile's samples 8 h .. 8 h + 7, as f16 pairs -(1024 + c): the records' fourth words, gathered by the wavefront
  // into 32 contiguous bytes of its own (two tiles ahead), so that a lane fetches its four pairs with one ds_read_b128
  uint32_t negc[4];
  uint8_t* const mycen = lds + kRaw0 + kWaves * (kStagesB * 2048) + wave * 64;
  auto cen_write = [&](int j) __attribute__((always_inline)) {
    if (lane < kTileB) *reinterpret_cast<uint16_t*>(mycen + (j & 1) * 32 + lane * 2) = (uint16_t)metal[(j & 7) * kTileB + lane].w;
  };
  auto load_negc = [&](int j) __attribute__((always_inline)) {
    const uint4 c = *reinterpret_cast<const uint4*>(mycen + (j & 1) * 32 + 16 * h);
    negc[0] = c.x; negc[1] = c.y; negc[2] = c.z; negc[3] = c.w;
  };
  // bytes (b0 .. b7) of a transposed read = samples 8 h .. 8 h + 7 of this lane's column -> b - c_sample, exact in f16
  // (0x6400 | b = 1024 + b); half a read (4 samples) per call: four vector instructions
  auto conv_half = [&](uint32_t raw, uint32_t na, uint32_t nb, uint32_t& o0, uint32_t& o1) __attribute__((always_inline)) {
    union { uint32_t u; h2t v; } w0, w1, c0, c1;
    w0.u = __builtin_amdgcn_perm(0x64646464u, raw, 0x05010400u);
    w1.u = __builtin_amdgcn_perm(0x64646464u, raw, 0x05030402u);
    c0.u = na; c1.u = nb;
    w0.v = w0.v + c0.v; w1.v = w1.v + c1.v;
    o0 = w0.u; o1 = w1.u;
  };
  union XB { uint32_t u[4]; f16x8 v; };
  // B operands, two column tiles at a time.  A tile's 24 MFMAs come in four groups of six (patch row 0 tiles 0-1, 2-3, row 1
  // tiles 0-1, 2-3); while a group runs from one pair, the next group's bytes are converted into the other.
  XB xbA[2], xbB[2];
  v2i raw[2];
  auto read_raw = [&](int j, int q) __attribute__((always_inline)) {   // group q (0..3) of tile j
    const uint8_t* sb = lds + (j & 3) * kStageBytesB + bbase + (q >> 1) * (kRowChunks * 16) + (q & 1) * 64;
    raw[0] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(sb));
    raw[1] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i*)(sb + 32));
  };
  auto conv_unit = [&](XB* xb, int u) __attribute__((always_inline)) {   // u = 2 * tile + half
    if (u & 1) conv_half((uint32_t)raw[u >> 1].y, negc[2], negc[3], xb[u >> 1].u[2], xb[u >> 1].u[3]);
    else conv_half((uint32_t)raw[u >> 1].x, negc[0], negc[1], xb[u >> 1].u[0], xb[u >> 1].u[1]);
  };
  struct AF { f16x8 p[kPiecesB]; };   // A fragments: the pieces of dz'
  auto read_af = [&](AF& af) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < kPiecesB; ++p) {
      union { s16x4 s[2]; f16x8 v; } u;
      u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + p * 1024));
      u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(pa + p * 1024 + 4 * 64));
      af.p[p] = u.v;
    }
  };

  for (int m = 0; m < 4; ++m) issue_meta(m);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int j = 0; j < 3; ++j) {
    read_slots(j);
    issue(j);
  }
  read_slots(3);
  SRL_OBSB_WAIT(10, 8);  // behind dz(0): two tiles of 5 / 4 operations
  asm volatile("s_barrier" ::: "memory");  // (the frames of tile 0 came in through all eight wavefronts)
  AF afA, afB;
  stage_dz(0);
  read_af(afA);
  cen_write(0);
  cen_write(1);
  load_negc(0);
  read_raw(0, 0);
#pragma unroll
  for (int u = 0; u < 4; ++u) conv_unit(xbA, u);
  read_raw(0, 1);

  // Tile it.  Everything its first MFMAs need is in registers when the barrier opens (the A fragments and the first pair of B
  // operands were fetched under the previous tile's MFMAs); between the MFMAs, in the order written (sched_barrier: nothing moves
  // across) and at most two four-instruction units per gap: the conversions of the next group, the staging of dz(it + 1), the
  // LDS reads of what comes after.  An MFMA occupies the matrix pipeline for 32 cycles and the vector issue for 8 of them; left
  // to the compiler, the vector work came as one block ahead of 24 back-to-back MFMAs and the two wavefronts of a SIMD, in step
  // through the barrier, queued for the vector unit and then for the matrix unit (both 41-45 % busy, one after the other).
  auto tile = [&](int it, const AF& af, AF& afn) __attribute__((always_inline)) {
    SRL_OBSB_WAIT(5, 4);  // all but the last tile's fetch: frames and dz of tile it + 1 are in
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(it + 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 24; ++g) {
      const int k = g / 6, j = g % 6;
      XB* const cur = (k & 1) ? xbB : xbA;
      XB* const nxt = (k & 1) ? xbA : xbB;
      if (!(DBG & 4) && (j >> 1) < kPiecesB) acc[2 * k + (j & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af.p[j >> 1], cur[j & 1].v, acc[2 * k + (j & 1)], 0, 0, 0);
      if (DBG & 16) {
      } else if (j < 4) conv_unit(nxt, j);                                     // group k + 1 (k = 3: group 0 of tile it + 1)
      else if (j == 4) read_raw(k < 2 ? it : it + 1, (k + 2) & 3);             // the bytes of group k + 2
      // staging of dz(it + 1), three units per pair of channels; the centres of tile it + 1 before its first conversions
      if (DBG & 16) {
      } else if (g == 0) stage_load(it + 1);
      else if (g == 1) stage_unit(0, 0);
      else if (g == 3) stage_unit(0, 1);
      else if (g == 4) stage_unit(0, 2);
      else if (g == 5) { stage_unit(1, 0); cen_write(it + 2); }
      else if (g == 7) stage_unit(1, 1);
      else if (g == 9) stage_unit(1, 2);
      else if (g == 10) stage_unit(2, 0);
      else if (g == 11) stage_unit(2, 1);
      else if (g == 13) stage_unit(2, 2);
      else if (g == 15) stage_unit(3, 0);
      else if (g == 16) { stage_unit(3, 1); load_negc(it + 1); }
      else if (g == 17) stage_unit(3, 2);
      else if (g == 18) stage_store();
      else if (g == 20) read_af(afn);
      else if (g == 23) read_slots(it + 4);  // for the next tile's fetch (its records came in seven tiles ahead)
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int it = 0; it < nu; it += 2) {
    tile(it, afA, afB);
    if (it + 1 < nu) tile(it + 1, afB, afA);
  }
#undef SRL_OBSB_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // R / C: lanes with the same channel group (lane & 3) hold different samples: fold through LDS, one atomic per channel and wavefront
  {
    float* red = reinterpret_cast<float*>(lds) + wave * (64 * 16);
    const float cs = 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[lane * 16 + 2 * i] = rsum[i][0];
      red[lane * 16 + 2 * i + 1] = rsum[i][1];
      red[lane * 16 + 8 + 2 * i] = csum[i][0] * cs;
      red[lane * 16 + 8 + 2 * i + 1] = csum[i][1] * cs;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a wavefront's own LDS traffic is ordered)
    const int ch = lane & 31, which = lane >> 5;  // channel ch = 8 q + i: lanes with lane & 3 == q, value which * 8 + i
    const int qq = ch >> 3, ii = ch & 7;
    float tsum = 0.f;
    for (int s16 = 0; s16 < 16; ++s16) tsum += red[(4 * s16 + qq) * 16 + which * 8 + ii];
    atomicAdd((which ? a.C : a.R) + pos * kCout + ch, tsum);
  }
  // D[o][k]: registers 4 g .. 4 g + 3 hold channels 8 g + 4 h + (0..3) of patch column 32 t + l31
  const float inv = inv_scale;
  float* qo = a.Q + (long)split * a.slab + (long)pos * kCout * 256;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 8 * (r >> 2) + (r & 3) + 4 * h;
      qo[(long)o * 256 + 32 * t + l31] = acc[t][r] * inv;
    }
}
#endif  // __HIPCC__

}  // namespace srlobs
