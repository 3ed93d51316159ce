// Contractions over PRE-SPLIT operands (round 4): OUT^T[ch, pos] = sum_k W[ch, k] X[pos, k], both operands stored as two f16
// pieces per element ("h2p" format, below), staged into LDS by the DMA path (buffer_load ... lds: no register, no vector
// operation and no LDS store instruction is spent on staging), consumed by 32x32x16 f16 MFMAs as the three piece products
// h0 h0' + h0 h1' + h1 h0' (gemm_bf16x3.h: exact in the float32 accumulate).
//
// Why: the round-3 kernels (gemm3_kernel) were staging-bound -- every tile loaded float32 operands into registers, split
// them (2.5 vector operations per element) and wrote the planes to LDS with ds_write_b64 (85 B/clk/CU); leaving the in-loop
// loads out gained 30-42 %, the split 7-28 %.  Here an activation is split ONCE, by the epilogue of the kernel that produces
// it (same 4 bytes per element in HBM as the float32 it replaces), weights once per update, and the consumers move bytes.
//
// h2p format of a tensor whose innermost extent C is a multiple of 32: every aligned block of 32 elements is 128 bytes =
// 4 groups x { 8 first pieces (16 B), 8 second pieces (16 B) }; group g holds elements base(g) + {0,1,2,3, 8,9,10,11} of the
// block, base(g) = 4 (g >> 1) + 16 (g & 1).  That order is what a 32x32 MFMA accumulator holds per lane when the block index
// is the accumulator's ROW index (lane half h, registers 0-7 / 8-15: groups 2h / 2h+1), so a producer stores 64 contiguous
// bytes per lane with no cross-lane movement; a contraction does not care about the order of its summation index as long as
// both operands agree, and both are h2p along k.  value = (h0 + h1) / scale, scale a power of two kept in a device float
// beside the tensor (h2_scale_for: chosen from an upper BOUND of the tensor's magnitude that is known before the tensor is
// produced -- max |x| max_row |w|_1 + max |b| -- so that a producer can split while it stores; f16 overflow is impossible by
// construction; what a loose bound costs is absolute resolution of the small elements, 2^-25 / scale).
//
// Pipeline: a workgroup of 8 wavefronts owns NCB x 32 channels x 256 positions.  A k-step is 32 k-values: the X tile is 256
// rows x 128 B, the W tile NCB*32 rows x 128 B, rows XOR-swizzled in 16-byte slots ((row >> 1) & 7: conflict-free
// ds_read_b128 fragments) by permuting the SOURCE addresses of the DMA, whose LDS side is linear.  S stages form a ring;
// step t+S-1 is issued right after the barrier of step t, completion is counted (s_waitcnt vmcnt(N), never 0 in the loop),
// one raw s_barrier per step.  Wavefront w: position block w & 3 (64 positions), k-half w >> 2 (16 of the step's 32
// k-values): each wavefront reads 4 + 4 fragments and issues 12 (NCB = 2) MFMAs per step; the two k-halves are added
// through LDS once, after the loop.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srlh2 {

typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef int h2_i32x4 __attribute__((ext_vector_type(4)));

enum { H2X_DENSE = 0, H2X_CONV = 1, H2X_GROUPED = 2 };
enum { H2O_F32 = 0, H2O_H2P = 1 };

// element j (0..7) of group g (0..3) of a 32-block -> element index inside the block
__host__ __device__ inline int h2p_elem(int g, int j) { return 4 * (g >> 1) + 16 * (g & 1) + (j & 3) + 8 * (j >> 2); }

// power-of-two scale for a tensor bounded by `bound`: bound * scale in [2^14, 2^15)
__host__ __device__ inline float h2_scale_for(float bound) {
  if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
  int e;
  frexpf(bound, &e);  // bound = m 2^e, m in [0.5, 1)
  int p = 15 - e;
  if (p > 100) p = 100;
  if (p < -100) p = -100;
  return ldexpf(1.f, p);
}

struct H2Args {
  const void* x;  // positions operand, h2p rows
  const void* w;  // channels operand, h2p [NC][K]
  const float* sx;  // device floats: scales of x and w
  const float* sw;
  int64_t M;        // positions (DENSE / CONV: rows; GROUPED: images)
  int32_t NC;       // channels (rows of w), a multiple of 32
  int32_t nk;       // k-steps of 32 (DENSE / CONV); GROUPED: taps * cbk
  uint32_t w_row_bytes;  // K * 4
  uint32_t x_row_bytes;  // DENSE: row pitch; CONV / GROUPED: image pitch (H * W * Cin * 4 / OH * OW * Cout * 4)
  // CONV (forward): input H x W x Cin, output OH x OW, stride; taps KH x KW.  GROUPED (data gradient): "input" is dz
  // [OH, OW, Cout], rows of a tile are images at ONE pixel (a, b) of the per-class grid GH x GW
  int32_t H, W, C, OH, OW, stride, KH, KW;
  int32_t GH, GW;       // GROUPED: pixel grid per parity class
  int32_t tiles_c;      // channel tiles (NC / (NCB * 32))
  uint32_t koff_x[64];  // CONV: byte offset of k-step t inside the image relative to the row's first pixel
  uint32_t koff_w[64];  // CONV: byte offset of k-step t inside a row of w
  // epilogue
  const float* bias;     // [NC] or null
  int32_t act;           // 1: relu
  int32_t out_fmt;       // H2O_*
  void* out;
  uint32_t out_row_bytes;   // DENSE / CONV: pitch of an output row (position); GROUPED: image pitch
  int32_t out_C;            // GROUPED: channels per pixel of the output (Cin of the layer); classes = NC / out_C
  int32_t out_H, out_W;     // GROUPED: output image
  float* out_scale;         // H2O_H2P: the scale this launch uses is written here (for the consumers)
  const float* bound_in;    // H2O_H2P: device float, max |x| (measured by x's producer) ...
  const float* bound_w;     // ... times this device float (max row 1-norm of w, in x's units) ...
  const float* bound_b;     // ... plus this one (max |bias|, or null) bounds |out|
  float* out_absmax;        // max |out| folded in (atomic max), or null
  uint32_t* mask_out;       // relu sign bits of out, natural element order, or null
  const uint32_t* mask_in;  // out *= bit of this mask at the output element (relu derivative), or null
  int32_t mask_in_h2;       // mask_in's bits are in h2 order (bit 8 g + j of a 32-block's word: what h2conv.h's forward kernels write)
  int32_t dbg;              // timing experiments (wrong results; SRL_H2G_DBG): 1 no DMA, 2 no fragment reads / MFMAs, 4 no epilogue stores
  // DENSE only: the reduction split over `ksplits` workgroups per tile, split s writing its raw partial sums (no bias, activation or
  // mask: the consumer adds the slabs) to out + s * slab_bytes.  A few hundred rows times a long reduction (the Linear forward of
  // an inference batch: 2048 x 512 over K = 3136 is 32 tiles on 256 CUs, each filling 4.8 MB at a CU's ~33 GB/s: 104 us).
  int32_t ksplits;
  int64_t slab_bytes;
};

#ifdef __HIPCC__

__device__ __forceinline__ h2_i32x4 h2_rsrc(const void* p) {
  const uint64_t a = (uint64_t)p;
  h2_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
  r[2] = -1;
  r[3] = 0x00020000;
  return r;
}

// NI DMA instructions of one wavefront into LDS at m0 = lds, lds + 8 KB, ...; instruction i reads 64 x 16 bytes at
// rsrc + voff[i] + soff.  One statement: the compiler schedules nothing between the M0 writes and their readers.
template <int NI, int STRIDE = 0x2000>
__device__ __forceinline__ void h2_dma(uint32_t lds, const uint32_t (&voff)[4], h2_i32x4 rsrc, uint32_t soff) {
  static_assert(NI >= 1 && NI <= 4, "");
  if (NI == 4)
    asm volatile(
        "s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
        "s_add_u32 m0, m0, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds"
        ::"s"(lds), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(rsrc), "s"(soff), "n"(STRIDE)
        : "memory", "scc");
  else if (NI == 2)
    asm volatile(
        "s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
        "s_add_u32 m0, m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
        ::"s"(lds), "v"(voff[0]), "v"(voff[1]), "s"(rsrc), "s"(soff), "n"(STRIDE)
        : "memory", "scc");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff[0]), "s"(rsrc),
                 "s"(soff)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void h2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// two f16 pieces of v * scale, packed: lo = pieces of a, hi = pieces of b
__device__ __forceinline__ void h2_split_pair(float a, float b, float scale, uint32_t& p0, uint32_t& p1) {
  const _Float16 a0 = (_Float16)__builtin_fmaf(a, scale, 0.f), b0 = (_Float16)__builtin_fmaf(b, scale, 0.f);
  const _Float16 a1 = (_Float16)__builtin_fmaf(a, scale, -(float)a0), b1 = (_Float16)__builtin_fmaf(b, scale, -(float)b0);
  union { h2_f16x2 h; uint32_t u; } x, y;
  x.h = h2_f16x2{a0, b0};
  y.h = h2_f16x2{a1, b1};
  p0 = x.u;
  p1 = y.u;
}

// ---------------------------------------------------------------------------------------------------------------------
// float32 [rows, C] (pitch ld floats) -> h2p rows (pitch C * 4 bytes) under *scale_out = h2_scale_for(*absmax * (headroom))
// One thread per group of 8 elements.  scale: if scale_in != null use *scale_in, else derive from *absmax and write *scale_out.
static __global__ void h2_pack_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int C, const float* absmax,
                               const float* scale_in, float* scale_out, uint8_t* __restrict__ dst) {
  const float scale = scale_in ? *scale_in : h2_scale_for(*absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  const int64_t ngrp = rows * (C / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngrp; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / (C / 8);
    const int gi = (int)(t - row * (C / 8));
    const int blk = gi >> 2, g = gi & 3;
    const float* s = src + row * ld + blk * 32;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[h2p_elem(g, j)];
    uint4 h0, h1;
    h2_split_pair(v[0], v[1], scale, h0.x, h1.x);
    h2_split_pair(v[2], v[3], scale, h0.y, h1.y);
    h2_split_pair(v[4], v[5], scale, h0.z, h1.z);
    h2_split_pair(v[6], v[7], scale, h0.w, h1.w);
    uint4* d = reinterpret_cast<uint4*>(dst + (row * C + blk * 32) * 4 + g * 32);
    d[0] = h0;
    d[1] = h1;
  }
}

// ... with the column sums of src beside it (round 6: the bias gradient of the layer whose output gradient is being packed -- a
// pass of its own over the 33 MB of dy took 32 us per 16 384 rows, with float atomics).  A workgroup takes a contiguous range of
// rows: thread t = (row lane t / (C / 8), group of 8 elements t % (C / 8)) keeps the 8 sums of its columns in registers, the row
// lanes meet in LDS, the workgroup writes slab[blockIdx][C]; h2_colsum_finish_kernel adds the slabs in a fixed order.
// C / 8 <= 256 threads (C <= 2048), blockDim = 256.
static __global__ __launch_bounds__(256) void h2_pack_colsum_kernel(const float* __restrict__ src, int64_t ld, int64_t rows, int C,
                                                                    const float* absmax, const float* scale_in, float* scale_out,
                                                                    uint8_t* __restrict__ dst, float* __restrict__ slabs) {
  extern __shared__ float h2pc_part[];   // [row lanes][C]
  const float scale = scale_in ? *scale_in : h2_scale_for(*absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  const int ng = C / 8, lanes = 256 / ng;            // groups per row; row lanes of this workgroup
  const int gi = threadIdx.x % ng, rl = threadIdx.x / ng;
  const int blk = gi >> 2, g = gi & 3;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < lanes) {
    for (int64_t row = r0 + rl; row < r1; row += lanes) {
      const float* s = src + row * ld + blk * 32 + 4 * (g >> 1) + 16 * (g & 1);
      const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 8);   // h2p_elem(g, 0..3), (g, 4..7)
      cs[0] += a.x; cs[1] += a.y; cs[2] += a.z; cs[3] += a.w; cs[4] += b.x; cs[5] += b.y; cs[6] += b.z; cs[7] += b.w;
      uint4 h0, h1;
      h2_split_pair(a.x, a.y, scale, h0.x, h1.x);
      h2_split_pair(a.z, a.w, scale, h0.y, h1.y);
      h2_split_pair(b.x, b.y, scale, h0.z, h1.z);
      h2_split_pair(b.z, b.w, scale, h0.w, h1.w);
      uint4* d = reinterpret_cast<uint4*>(dst + (row * C + blk * 32) * 4 + g * 32);
      d[0] = h0;
      d[1] = h1;
    }
    float* p = h2pc_part + rl * C + blk * 32 + 4 * (g >> 1) + 16 * (g & 1);
    p[0] = cs[0]; p[1] = cs[1]; p[2] = cs[2]; p[3] = cs[3]; p[8] = cs[4]; p[9] = cs[5]; p[10] = cs[6]; p[11] = cs[7];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += h2pc_part[l * C + c];
    slabs[(int64_t)blockIdx.x * C + c] = t;
  }
}
// out[c] (+)= sum over the slabs, in a fixed order: a workgroup takes 32 columns, its 8 slab lanes every eighth slab each (four
// independent sums per lane), and the lanes meet in LDS.  (First version: one thread per column walked all 512 slabs, two
// workgroups per launch -- 35 us of dependent loads for 1 MB, in front of the layer's weight-gradient launch.)
static __global__ __launch_bounds__(256) void h2_colsum_finish_kernel(const float* __restrict__ slabs, int nslab, int C, float* __restrict__ out,
                                                                      int accumulate) {
  __shared__ float part[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  if (c < C) {
    int k = sl;
    for (; k + 24 < nslab; k += 32) {
      t0 += slabs[(int64_t)k * C + c]; t1 += slabs[(int64_t)(k + 8) * C + c];
      t2 += slabs[(int64_t)(k + 16) * C + c]; t3 += slabs[(int64_t)(k + 24) * C + c];
    }
    for (; k < nslab; k += 8) t0 += slabs[(int64_t)k * C + c];
  }
  part[sl][cl] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (sl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) t += part[j][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// h2p rows -> float32 (tests, fallbacks)
static __global__ void h2_unpack_kernel(const uint8_t* __restrict__ src, int64_t rows, int C, const float* scale, float* __restrict__ dst,
                                 int64_t ld) {
  const float inv = 1.f / *scale;
  const int64_t ngrp = rows * (C / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngrp; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / (C / 8);
    const int gi = (int)(t - row * (C / 8));
    const int blk = gi >> 2, g = gi & 3;
    const _Float16* s = reinterpret_cast<const _Float16*>(src + (row * C + blk * 32) * 4 + g * 32);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[row * ld + blk * 32 + h2p_elem(g, j)] = ((float)s[j] + (float)s[8 + j]) * inv;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// KSPLIT (the default geometry): wavefront w = (position block w & 3, k-half w >> 2), every wavefront all NCB channel blocks.
// !KSPLIT (NCB = 8, the data gradient of a wide Linear: 256 channels per workgroup, a third less staged per flop): wavefront
// w = (position block w & 3, channel half w >> 2), both k-halves of every step; no exchange at the end.
// HALF (round 5; NCB = 8, dense, unsplit): 128 positions x 256 channels per workgroup of FOUR wavefronts, k-steps of 16 k-values
// (64-byte rows: one k-half of the h2p block), S stages of 24 KB -- so that TWO workgroups share a CU.  With one 8-wavefront
// workgroup per CU a tile's ring fill, k-loop and 256 KB of stores run one after the other (leave-outs, DESIGN section 4: 89 + 45 +
// 84 us of the Linear's data gradient); a second resident workgroup is what the hardware can overlap them with.
template <int NCB, int XMODE, int S, bool KSPLIT = true, bool HALF = false>
__global__ __launch_bounds__(HALF ? 256 : 512, 2) void h2gemm_kernel(H2Args g) {
  static_assert(NCB == 2 || NCB == 4 || (NCB == 8 && !KSPLIT), "64, 128 or (unsplit) 256 channels per workgroup");
  static_assert(S == 2 || S == 3 || S == 4, "ring depth");
  static_assert(!HALF || (NCB == 8 && !KSPLIT && XMODE == H2X_DENSE), "half steps: the dense 256-channel geometry");
  constexpr int NCW = KSPLIT ? NCB : NCB / 2;   // channel blocks per wavefront
  constexpr int BP = HALF ? 128 : 256;          // positions per workgroup
  constexpr int PITCH = HALF ? 64 : 128;        // bytes of a row per k-step in LDS
  constexpr int NWAVES = HALF ? 4 : 8;
  constexpr int XT = BP * PITCH, WT = NCB * 32 * PITCH, STAGE = XT + WT;
  constexpr int NXI = XT / 1024 / NWAVES;       // X-tile DMA instructions per wavefront and step (1 KB each)
  constexpr int NWI = WT / 1024 / NWAVES;       // W-tile DMA instructions per wavefront and step
  constexpr int DSTRIDE = NWAVES * 1024;        // LDS distance between a wavefront's consecutive DMA instructions
  constexpr int L = NXI + NWI;                  // DMA instructions per wavefront and step
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = HALF ? wid & 1 : wid & 3, kh = KSPLIT ? wid >> 2 : 0, chalf = KSPLIT ? 0 : (HALF ? wid >> 1 : wid >> 2);
  // logical tile id: every XCD owns one contiguous run
  unsigned lid;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    lid = xcd * q + (xcd < r ? xcd : r) + slot;
  }
  unsigned ks = 0;   // this workgroup's slice of the reduction
  if (XMODE == H2X_DENSE && g.ksplits > 1) {
    ks = lid % (unsigned)g.ksplits;
    lid /= (unsigned)g.ksplits;
  }
  const unsigned tile_c = lid % (unsigned)g.tiles_c, tile_p = lid / (unsigned)g.tiles_c;
  const int c0 = tile_c * (NCB * 32);

  // ---- X rows of this wavefront's DMA instructions: instruction q covers tile rows 8 (wid + 8 q) .. + 7, lane i -> row + (i >> 3),
  // 16-byte slot (i & 7) ^ f(row)
  uint32_t xv[4];
  long m0 = 0;            // first position of the tile (DENSE / CONV) or first image (GROUPED)
  uint32_t gsoff = 0;     // GROUPED: byte offset of the tile's pixel inside an image, before the tap
  int ga = 0, gb = 0;     // GROUPED: the tile's pixel on the class grid
  if (XMODE == H2X_GROUPED) {
    const unsigned P = (unsigned)(g.GH * g.GW);
    const unsigned ig = tile_p / P, p = tile_p - ig * P;
    m0 = (long)ig * BP;
    ga = p / g.GW;
    gb = p - ga * g.GW;
  } else {
    m0 = (long)tile_p * BP;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) xv[q] = 0;
#pragma unroll
  for (int q = 0; q < NXI; ++q) {
    // an instruction = 1 KB: 8 rows of 128 bytes (slot = 16-byte piece, XOR (row >> 1) & 7) or, HALF, 16 rows of 64 (XOR (row >> 2) & 3)
    const int row = HALF ? 16 * (wid + NWAVES * q) + (lane >> 2) : 8 * (wid + NWAVES * q) + (lane >> 3);
    const int slot = HALF ? (lane & 3) ^ ((row >> 2) & 3) : (lane & 7) ^ ((row >> 1) & 7);
    long m = m0 + row;
    if (m >= g.M) m = g.M - 1;
    uint32_t base;
    if (XMODE == H2X_DENSE) base = (uint32_t)m * g.x_row_bytes;
    else if (XMODE == H2X_GROUPED) base = (uint32_t)m * g.x_row_bytes;
    else {
      const uint32_t ohw = (uint32_t)(g.OH * g.OW);
      const uint32_t img = (uint32_t)m / ohw, rem = (uint32_t)m - img * ohw;
      const uint32_t oy = rem / (uint32_t)g.OW, ox = rem - oy * (uint32_t)g.OW;
      base = img * g.x_row_bytes + ((oy * g.stride) * g.W + ox * g.stride) * (uint32_t)g.C * 4u;
    }
    xv[q] = base + 16u * slot;
  }
  uint32_t wv[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < NWI; ++q) {
    const int row = HALF ? 16 * (wid + NWAVES * q) + (lane >> 2) : 8 * (wid + NWAVES * q) + (lane >> 3);
    const int slot = HALF ? (lane & 3) ^ ((row >> 2) & 3) : (lane & 7) ^ ((row >> 1) & 7);
    int ch = c0 + row;
    if (ch >= g.NC) ch = g.NC - 1;
    wv[q] = (uint32_t)ch * g.w_row_bytes + 16u * slot;
  }
  const h2_i32x4 rx = h2_rsrc(g.x), rw = h2_rsrc(g.w);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const uint32_t lds_x = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);        // + stage * STAGE
  const uint32_t lds_w = __builtin_amdgcn_readfirstlane(lds0 + XT + wid * 1024);

  // ---- k-step sequence
  int nsteps = HALF ? 2 * g.nk : g.nk;
  int step0 = 0;
  if (XMODE == H2X_DENSE && g.ksplits > 1) {
    step0 = (int)((long)ks * nsteps / g.ksplits);
    nsteps = (int)((long)(ks + 1) * nsteps / g.ksplits) - step0;
  }
  uint64_t vmask = 0;  // GROUPED: valid taps
  const int cbk = XMODE == H2X_GROUPED ? g.C / 32 : 1;
  if (XMODE == H2X_GROUPED) {
    // taps (dy, dx) of the per-class problem: dz position (ga - dy, gb - dx); class-uniform geometry (host checks)
    const int TH = g.KH / g.stride, TW = g.KW / g.stride;
    nsteps = 0;
    for (int t = 0; t < TH * TW; ++t) {
      const int dy = t / TW, dx = t - dy * TW;
      if ((unsigned)(ga - dy) < (unsigned)g.OH && (unsigned)(gb - dx) < (unsigned)g.OW) { vmask |= 1ull << t; ++nsteps; }
    }
    nsteps *= cbk;
    (void)gsoff;
  }
  int gt = -1, gcb = cbk - 1;  // GROUPED iterator state: current tap, channel block
  auto step_offsets = [&](int t, uint32_t& sx, uint32_t& sw) {
    if (XMODE == H2X_DENSE) {
      sx = sw = (uint32_t)(t + step0) * (uint32_t)PITCH;
    } else if (XMODE == H2X_CONV) {
      sx = g.koff_x[t];
      sw = g.koff_w[t];
    } else {
      if (++gcb == cbk) {
        gcb = 0;
        gt = __builtin_ctzll(vmask);
        vmask &= vmask - 1;
      }
      const int TW = g.KW / g.stride;
      const int dy = gt / TW, dx = gt - dy * TW;
      sx = (uint32_t)(((ga - dy) * g.OW + (gb - dx)) * g.C * 4 + gcb * 128);
      sw = (uint32_t)((gt * cbk + gcb) * 128);
    }
  };
  // the offsets of the step issued NEXT are fetched one issue early (CONV: scalar loads from the kernel arguments, whose
  // latency would otherwise sit between the barrier and the DMA instructions of every step)
  uint32_t nsx = 0, nsw = 0;
  step_offsets(0, nsx, nsw);
  auto issue_off = [&](uint32_t stage_off, int t) {
    const uint32_t sx = __builtin_amdgcn_readfirstlane(nsx), sw = __builtin_amdgcn_readfirstlane(nsw);
    if (!(g.dbg & 1)) {
      h2_dma<NXI, DSTRIDE>(lds_x + stage_off, xv, rx, sx);
      h2_dma<NWI, DSTRIDE>(lds_w + stage_off, wv, rw, sw);
    }
    if (t + 1 < nsteps) step_offsets(t + 1, nsx, nsw);
  };
  auto issue = [&](int stage, int t) { issue_off((uint32_t)stage * STAGE, t); };

  h2_f32x16 acc[NCW][2];
#pragma unroll
  for (int i = 0; i < NCW; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row r = lane & 31 of a 32-row block, group 2 kh + (lane >> 5), plane p: slot (2 g + p) ^ f(r)
  // (HALF: a staged row is one k-half, 4 slots: group (lane >> 5), plane p at slot (2 group + p) ^ ((r >> 2) & 3))
  const int fr_r = lane & 31, fr_f = HALF ? (fr_r >> 2) & 3 : (fr_r >> 1) & 7;
  auto fr_off = [&](int khh, int pl) {
    return HALF ? fr_r * 64 + 16 * ((2 * (lane >> 5) + pl) ^ fr_f) : fr_r * 128 + 16 * ((4 * khh + 2 * (lane >> 5) + pl) ^ fr_f);
  };
  constexpr int BLK = 32 * PITCH;   // bytes of a 32-row block of a staged tile

  auto compute = [&](uint32_t stage_off) {
    const uint8_t* sx = lds + stage_off + wp * (2 * BLK);
    const uint8_t* sw = lds + stage_off + XT + chalf * (NCW * BLK);
#pragma unroll
    for (int k2 = 0; k2 < ((KSPLIT || HALF) ? 1 : 2); ++k2) {
      const int khh = KSPLIT ? kh : k2;
      const int o0 = fr_off(khh, 0), o1 = fr_off(khh, 1);
      h2_f16x8 xf[2][2], wf[NCW][2];
#pragma unroll
      for (int i = 0; i < NCW; ++i) {
        wf[i][0] = *reinterpret_cast<const h2_f16x8*>(sw + i * BLK + o0);
        wf[i][1] = *reinterpret_cast<const h2_f16x8*>(sw + i * BLK + o1);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        xf[j][0] = *reinterpret_cast<const h2_f16x8*>(sx + j * BLK + o0);
        xf[j][1] = *reinterpret_cast<const h2_f16x8*>(sx + j * BLK + o1);
      }
      // small terms first
#pragma unroll
      for (int i = 0; i < NCW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][1], xf[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NCW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][0], xf[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NCW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][0], xf[j][0], acc[i][j], 0, 0, 0);
    }
  };

  // ---- ring: prologue S-1 steps, then per step { wait own DMA of step t; barrier; issue step t+S-1; compute step t }.
  // At the wait of step t the steps up to t+S-2 have been issued: with (S-2) L instructions allowed in flight step t has
  // landed (in-order completion); the last S-1 steps, with nothing left to issue, drain completely.
  const int nk = nsteps;
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) issue(s, s);
  uint32_t st_cur = 0, st_nxt = (S - 1) * STAGE;  // byte offsets of the stage being computed / being filled
  for (int t = 0; t < nk; ++t) {
    const bool more = t + (S - 1) < nk;
    if (more) h2_wait_vm<L*(S - 2)>();
    else h2_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (more) issue_off(st_nxt, t + S - 1);
    if (!(g.dbg & 2)) compute(st_cur);
    st_cur = st_cur + STAGE == S * STAGE ? 0 : st_cur + STAGE;
    st_nxt = st_nxt + STAGE == S * STAGE ? 0 : st_nxt + STAGE;
  }

  // ---- KSPLIT: add the two k-halves: wavefront kh keeps channel blocks [kh * NCB/2, (kh+1) * NCB/2) and receives its partner's sums
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();  // every wavefront is past its last fragment read: the ring is free
  constexpr int NH = KSPLIT ? NCB / 2 : NCW;
  h2_f32x16 fin[NH][2];
  // (static register indices only: an accumulator array indexed by the runtime k-half would live in scratch memory)
  auto exchange = [&](auto kh_c) {
    constexpr int KH_ = decltype(kh_c)::value;
    if constexpr (KSPLIT) {
      float4* ex = reinterpret_cast<float4*>(lds) + (size_t)wid * (NH * 2 * 4 * 64);
#pragma unroll
      for (int i = 0; i < NH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const h2_f32x16& a = acc[(1 - KH_) * NH + i][j];
            ex[((i * 2 + j) * 4 + q) * 64 + lane] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
          }
      __syncthreads();
      const float4* ey = reinterpret_cast<const float4*>(lds) + (size_t)(wid ^ 4) * (NH * 2 * 4 * 64);
#pragma unroll
      for (int i = 0; i < NH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = ey[((i * 2 + j) * 4 + q) * 64 + lane];
            const h2_f32x16& a = acc[KH_ * NH + i][j];
            fin[i][j][4 * q] = a[4 * q] + v.x;
            fin[i][j][4 * q + 1] = a[4 * q + 1] + v.y;
            fin[i][j][4 * q + 2] = a[4 * q + 2] + v.z;
            fin[i][j][4 * q + 3] = a[4 * q + 3] + v.w;
          }
      __syncthreads();  // the epilogue's transposition blocks (lds + wid * 4096) lie inside other wavefronts' exchange regions
    }
  };
  if constexpr (KSPLIT) {
    if (kh == 0) exchange(std::integral_constant<int, 0>{});
    else exchange(std::integral_constant<int, 1>{});
  } else {
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) fin[i][j] = acc[i][j];
  }

  // ---- epilogue: lane = position (column), registers = channels (rows): r -> channel (r & 3) + 8 (r >> 2) + 4 h
  const float inv = 1.f / (*g.sx * *g.sw);
  float oscale = 1.f;
  if (g.out_fmt == H2O_H2P) {
    float bound = *g.bound_in * *g.bound_w;
    if (g.bound_b) bound += *g.bound_b;
    oscale = h2_scale_for(bound);
    if (lid == 0 && tid == 0) *g.out_scale = oscale;
  }
  const int hl = lane >> 5, pl = lane & 31;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int cb = c0 / 32 + (KSPLIT ? kh : chalf) * NH + i;  // global channel block
    if (cb * 32 >= g.NC) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long row = m0 + wp * 64 + j * 32 + pl;  // position or image
      const bool ok = row < g.M;
      // output address of this (row, channel block)
      uint8_t* optr;
      long elem0;  // element index of channel 0 of the block (mask addressing)
      if (XMODE == H2X_GROUPED) {
        const int cpb = g.out_C / 32;  // channel blocks per class
        const int cls = cb / cpb, cbl = cb - cls * cpb;
        const int py = cls / g.stride, px = cls - py * g.stride;
        const long pix = (long)(ga * g.stride + py) * g.out_W + (gb * g.stride + px);
        elem0 = ((long)row * g.out_H * g.out_W + pix) * g.out_C + cbl * 32;
        optr = static_cast<uint8_t*>(g.out) + elem0 * 4;
      } else {
        optr = static_cast<uint8_t*>(g.out) + (long)row * g.out_row_bytes + (long)cb * 128;
        elem0 = ((long)row * g.out_row_bytes) / 4 + (long)cb * 32;
      }
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = fin[i][j][r] * inv;
      if (g.bias) {  // registers 4q .. 4q+3: channels 8q + 4 hl + {0..3} of the block
        const float4* bq = reinterpret_cast<const float4*>(g.bias + cb * 32 + 4 * hl);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = bq[2 * q];
          v[4 * q] += b4.x; v[4 * q + 1] += b4.y; v[4 * q + 2] += b4.z; v[4 * q + 3] += b4.w;
        }
      }
      if (g.act == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      }
      if (g.mask_in) {
        const uint32_t mw = ok ? g.mask_in[elem0 >> 5] : 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r)   // h2 order: registers 0-7 are group 2 hl, 8-15 group 2 hl + 1: bit 16 hl + r
          if (!((mw >> (g.mask_in_h2 ? 16 * hl + r : (r & 3) + 8 * (r >> 2) + 4 * hl)) & 1u)) v[r] = 0.f;
      }
      if (g.mask_out) {
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= (v[r] > 0.f ? 1u : 0u) << ((r & 3) + 8 * (r >> 2) + 4 * hl);
        bits |= (uint32_t)__shfl_xor((int)bits, 32);
        if (ok && hl == 0) g.mask_out[elem0 >> 5] = bits;
      }
      if (g.out_absmax) {
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, ok ? fabsf(v[r]) : 0.f);
      }
      if (g.dbg & 4) continue;
      if (XMODE != H2X_GROUPED) {
        // Whole lines per store instruction.  A lane holds 64 bytes of ITS row (position): stored from there, an instruction
        // wrote 64 pieces of 16 bytes to 64 different lines (the Linear's data gradient: 72 of its 230 us were these stores).
        // Through the wavefront's own 4 KB of LDS (32 rows x 128 bytes, 16-byte slots XOR-swizzled by row & 7: conflict-free both
        // ways) eight lanes carry one row's 128 bytes and an instruction writes eight complete lines.  The ring is free here
        // (barrier above); a wavefront's LDS operations execute in order, so no barrier between the blocks either.
        uint8_t* tb = lds + wid * 4096;
        uint4* trow = reinterpret_cast<uint4*>(tb + pl * 128);
        const int sw = pl & 7;
        if (g.out_fmt == H2O_H2P) {
          // registers 0-7: group 2 hl, registers 8-15: group 2 hl + 1 -> slots 4 hl .. 4 hl + 3 of the row's 128 bytes
          uint4 h0a, h1a, h0b, h1b;
          h2_split_pair(v[0], v[1], oscale, h0a.x, h1a.x);
          h2_split_pair(v[2], v[3], oscale, h0a.y, h1a.y);
          h2_split_pair(v[4], v[5], oscale, h0a.z, h1a.z);
          h2_split_pair(v[6], v[7], oscale, h0a.w, h1a.w);
          h2_split_pair(v[8], v[9], oscale, h0b.x, h1b.x);
          h2_split_pair(v[10], v[11], oscale, h0b.y, h1b.y);
          h2_split_pair(v[12], v[13], oscale, h0b.z, h1b.z);
          h2_split_pair(v[14], v[15], oscale, h0b.w, h1b.w);
          trow[(4 * hl + 0) ^ sw] = h0a;
          trow[(4 * hl + 1) ^ sw] = h1a;
          trow[(4 * hl + 2) ^ sw] = h0b;
          trow[(4 * hl + 3) ^ sw] = h1b;
        } else {
          // float32: registers 4q .. 4q+3 = channels 8q + 4 hl + {0..3} -> slot 2 q + hl
#pragma unroll
          for (int q = 0; q < 4; ++q)
            trow[(2 * q + hl) ^ sw] = make_uint4(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]),
                                                 __float_as_uint(v[4 * q + 3]));
        }
        const int rr = lane >> 3, pp = lane & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const uint4 val = *reinterpret_cast<const uint4*>(tb + (8 * it + rr) * 128 + 16 * (pp ^ rr));
          const long rowg = m0 + wp * 64 + j * 32 + 8 * it + rr;
          if (rowg < g.M)
            *reinterpret_cast<uint4*>(static_cast<uint8_t*>(g.out) + (long)ks * g.slab_bytes + rowg * (long)g.out_row_bytes + (long)cb * 128 +
                                      16 * pp) = val;
        }
        continue;
      }
      if (!ok) continue;
      if (g.out_fmt == H2O_H2P) {
        // registers 0-7: group 2 hl, registers 8-15: group 2 hl + 1; 64 contiguous bytes per lane
        uint4 h0a, h1a, h0b, h1b;
        h2_split_pair(v[0], v[1], oscale, h0a.x, h1a.x);
        h2_split_pair(v[2], v[3], oscale, h0a.y, h1a.y);
        h2_split_pair(v[4], v[5], oscale, h0a.z, h1a.z);
        h2_split_pair(v[6], v[7], oscale, h0a.w, h1a.w);
        h2_split_pair(v[8], v[9], oscale, h0b.x, h1b.x);
        h2_split_pair(v[10], v[11], oscale, h0b.y, h1b.y);
        h2_split_pair(v[12], v[13], oscale, h0b.z, h1b.z);
        h2_split_pair(v[14], v[15], oscale, h0b.w, h1b.w);
        uint4* d = reinterpret_cast<uint4*>(optr + hl * 64);
        d[0] = h0a;
        d[1] = h1a;
        d[2] = h0b;
        d[3] = h1b;
      } else {
        float* d = reinterpret_cast<float*>(optr);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(d + 8 * q + 4 * hl) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
      }
    }
  }
  if (g.out_absmax) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) {
      const float cur = __hip_atomic_load(g.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (amax > cur) atomicMax(reinterpret_cast<int*>(g.out_absmax), __float_as_int(amax));  // non-negative floats order as ints
    }
  }
}

template <int NCB, int XMODE, int S, bool KSPLIT = true, bool HALF = false>
inline int h2gemm_launch(hipStream_t st, H2Args a) {
  constexpr int BP = HALF ? 128 : 256, PITCH = HALF ? 64 : 128;
  constexpr int STAGE = BP * PITCH + NCB * 32 * PITCH;
  a.tiles_c = (a.NC + NCB * 32 - 1) / (NCB * 32);
  long tiles_p;
  if (XMODE == H2X_GROUPED) tiles_p = ((a.M + BP - 1) / BP) * (long)(a.GH * a.GW);
  else tiles_p = (a.M + BP - 1) / BP;
  if (a.ksplits < 1 || XMODE != H2X_DENSE) a.ksplits = 1;
  const long nblk = tiles_p * a.tiles_c * a.ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffL) return -22;
  static bool attr_set = false;
  auto kern = h2gemm_kernel<NCB, XMODE, S, KSPLIT, HALF>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S * STAGE);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(HALF ? 256 : 512), S * STAGE, st, a);
  return 0;
}

#endif  // __HIPCC__

}  // namespace srlh2
