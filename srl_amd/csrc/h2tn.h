// Weight gradients of dense layers over PRE-SPLIT operands (round 6):  dW[o, c] = sum_m A[m, o] B[m, c], both operands h2p rows
// (h2gemm.h) whose SLOW index m is the summation index -- a "TN" product.  Replaces round 3's gemm3_kernel on this product (the
// Linear 3136 -> 512 of the Atari encoder: 196 us per 16 384 rows against a 63 us matrix floor; it loaded float32 / h2p operands
// into registers, re-ordered them with vector instructions and wrote the planes to LDS with ds_write_b64: lds_busy 0.61).
//
// Here nothing but DMA touches the operands on their way into LDS and nothing but the matrix cores' transposing reads on the
// way out:
//   * a workgroup owns 256 channels of A x 256 channels of B and a range of rows; a k-step is 16 rows: 16 DMA instructions per
//     operand, each ONE row's 1 KB (256 channels x 4 bytes, contiguous in HBM and in LDS -- whole 128-byte lines);
//   * both MFMA operands (v_mfma_f32_32x32x16_f16: 32 channels x 16 rows) come out of LDS through ds_read_b64_tr_b16: a 16-lane
//     group hands in 4 row addresses x 4 runs of 8 bytes and every lane receives ONE channel's four rows.  In h2p a run of 8 bytes
//     is 4 consecutive channels of one piece, so the 16 "columns" of a group are channels {0-3, 8-11} of groups g = 2 y, 2 y + 1
//     of the 32-block (y = the lane group's parity), and lane index i of the MFMA block <-> channel
//         ch(i) = (i & 3) | ((i >> 4) & 1) << 2 | ((i >> 2) & 1) << 3 | ((i >> 3) & 1) << 4
//     -- the same map on both operands, undone in the epilogue's addresses;
//   * rows are stored [r & 3][r >> 2] with the four r & 3 classes skewed by {0, 16, 128, 144} bytes: the 32 lanes of a transposed
//     read (4 rows x 2 lane groups x 4 runs) then fall on 32 distinct 8-byte bank pairs (conflict-free), and every fragment
//     address is one lane register plus an immediate;
//   * 8 wavefronts = 2 (A halves) x 4 (B quarters), 4 x 2 accumulator blocks each: 24 transposed reads (12 KB) and 24 MFMAs per
//     wavefront and k-step; ring of four 33 KB slots, three k-steps in flight, ONE barrier per k-step, counted vmcnt;
//   (template argument DBG: timing experiments with wrong results -- 1 no DMA, 2 no MFMAs, 4 no stores)
//   * each workgroup writes its partial products to a float32 slab of its row range (full-matrix layout: a 32 x 32 block's
//     register is two complete 128-byte lines per store instruction); gemm_core.h's reduce_slabs adds them (no atomics:
//     bit-reproducible).
#pragma once
#include "h2gemm.h"
#include <type_traits>

namespace srlh2 {

struct H2TnArgs {
  const void* a;   // h2p rows [M][NA] (the layer's output gradient)
  const void* b;   // h2p rows [M][NB] (the layer's input)
  const float* sa; // device floats: the operands' scales
  const float* sb;
  int64_t M;
  int32_t NA, NB;          // multiples of 32
  int64_t a_row_bytes, b_row_bytes;
  int32_t splits;          // row ranges (slabs)
  int64_t rows_per_split;  // a multiple of 16
  float* slabs;            // [splits][NA][NB] float32
  int32_t tiles_a, tiles_b;
};

#ifdef __HIPCC__

typedef short h2tn_s16x4 __attribute__((ext_vector_type(4)));

// lane index of a 32-wide MFMA block -> channel of the 32-block (see the header)
__host__ __device__ constexpr int h2tn_ch(int i) { return (i & 3) | (((i >> 4) & 1) << 2) | (((i >> 2) & 1) << 3) | (((i >> 3) & 1) << 4); }

constexpr int H2TN_OPB = 16 * 1024 + 160;      // bytes of one operand's 16 rows in a slot (skews up to 144 bytes)
constexpr int H2TN_SLOT = 2 * H2TN_OPB;        // 33 088
constexpr int H2TN_NSLOT = 4;
__host__ __device__ constexpr int h2tn_skew(int q) { return (q & 1) * 16 + (q >> 1) * 128; }
// LDS byte offset of row r (0..15) inside an operand's region
__host__ __device__ constexpr int h2tn_row_off(int r) { return ((r & 3) * 4 + (r >> 2)) * 1024 + h2tn_skew(r & 3); }

template <int DBG>
__global__ __launch_bounds__(512, 2) void h2tn_kernel(H2TnArgs g) {
  constexpr int TA = 4, TB = 2;     // accumulator blocks per wavefront: 128 channels of A x 64 channels of B
  constexpr int DPW = 4;            // DMA instructions per wavefront and k-step (32 rows of 1 KB over 8 wavefronts)
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wid & 1, wb = wid >> 1;
  // logical work item: every XCD owns one contiguous run; a row range's tiles are neighbours (they read the same rows)
  unsigned lid;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    lid = xcd * q + (xcd < r ? xcd : r) + slot;
  }
  const unsigned ntile = (unsigned)(g.tiles_a * g.tiles_b);
  const unsigned sp = lid / ntile, tile = lid - sp * ntile;
  const unsigned ta = tile % (unsigned)g.tiles_a, tb = tile / (unsigned)g.tiles_a;
  const int64_t m_lo = (int64_t)sp * g.rows_per_split;
  int64_t m_hi = m_lo + g.rows_per_split;
  if (m_hi > g.M) m_hi = g.M;
  const int nrows = (int)(m_hi - m_lo);            // > 0 (host)
  const int nu = (nrows + 15) >> 4;

  // ---- DMA: wavefront w carries rows w and w + 8 of both operands.  Source = a per-step base pointer (64-bit, scalar) + the
  // row's offset (scalar) + this lane's 16 bytes of the tile's 1 KB (a lane beyond the matrix's width re-reads the tile's first
  // 16 bytes: its channels are never stored)
  const int a_col0 = (int)ta * 1024, b_col0 = (int)tb * 1024;   // byte offsets of the tiles inside a row
  uint32_t va = (uint32_t)(lane * 16), vb = (uint32_t)(lane * 16);
  if (a_col0 + (int)va >= g.NA * 4) va = 0;
  if (b_col0 + (int)vb >= g.NB * 4) vb = 0;
  const uint64_t pa0 = (uint64_t)g.a + (uint64_t)m_lo * (uint64_t)g.a_row_bytes + (uint64_t)a_col0;
  const uint64_t pb0 = (uint64_t)g.b + (uint64_t)m_lo * (uint64_t)g.b_row_bytes + (uint64_t)b_col0;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const uint32_t lrow0 = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)h2tn_row_off(0) + (uint32_t)(((wid & 3) * 4 + (wid >> 2)) * 1024 + h2tn_skew(wid & 3)));
  // (row w: class w & 3, index w >> 2; row w + 8: index + 2 -> + 2 KB)
  // one DMA instruction: LDS base m (scalar), 64 x 16 bytes from rsrc + voff + soff
  auto dma1 = [&](uint32_t m, uint32_t voff, h2_i32x4 rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  // the four pieces of k-step u this wavefront carries: piece 0 / 1 = rows w / w + 8 of A, 2 / 3 = of B
  struct Pieces { h2_i32x4 ra, rb; uint32_t sa1, sb1, l0; };
  auto pieces = [&](int u) {
    Pieces p;
    const uint32_t slot = (uint32_t)(u & (H2TN_NSLOT - 1)) * H2TN_SLOT;
    const int r0 = 16 * u + wid, r1 = r0 + 8;                       // rows of the range
    // (rows past the range re-read its last row -- a valid address; they are zeroed in LDS before the step's fragments are read)
    const int e0 = r0 < nrows ? r0 : nrows - 1, e1 = r1 < nrows ? r1 : nrows - 1;   // e0 <= e1
    p.ra = h2_rsrc((const void*)(pa0 + (uint64_t)e0 * (uint64_t)g.a_row_bytes));
    p.rb = h2_rsrc((const void*)(pb0 + (uint64_t)e0 * (uint64_t)g.b_row_bytes));
    p.sa1 = __builtin_amdgcn_readfirstlane((uint32_t)((int64_t)(e1 - e0) * g.a_row_bytes));
    p.sb1 = __builtin_amdgcn_readfirstlane((uint32_t)((int64_t)(e1 - e0) * g.b_row_bytes));
    p.l0 = __builtin_amdgcn_readfirstlane(lrow0 + slot);
    return p;
  };
  auto piece = [&](const Pieces& p, int i) {
    if (DBG & 1) return;
    if (i == 0) dma1(p.l0, va, p.ra, 0u);
    else if (i == 1) dma1(p.l0 + 2048u, va, p.ra, p.sa1);
    else if (i == 2) dma1(p.l0 + (uint32_t)H2TN_OPB, vb, p.rb, 0u);
    else dma1(p.l0 + (uint32_t)H2TN_OPB + 2048u, vb, p.rb, p.sb1);
  };
  auto issue = [&](int u) {
    const Pieces p = pieces(u);
#pragma unroll
    for (int i = 0; i < DPW; ++i) piece(p, i);
  };

  // ---- fragment addresses (bytes inside a slot): lane = (k-half h, lane group y, row class q, run p4)
  const int h = lane >> 5, y = (lane >> 4) & 1, q = (lane >> 2) & 3, p4 = lane & 3;
  const uint32_t fbase = (uint32_t)((4 * q + 2 * h) * 1024 + h2tn_skew(q) + (4 * y + 2 * (p4 >> 1)) * 16 + 8 * (p4 & 1));
  const uint32_t fa = fbase + (uint32_t)(wa * TA * 128);                         // + block * 128 + piece * 16 + read * 1024
  const uint32_t fb = fbase + (uint32_t)(H2TN_OPB + wb * TB * 128);

  h2_f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto tr = [&](const uint8_t* p) {
    typedef __attribute__((address_space(3))) h2tn_s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  };
  auto frag = [&](const uint8_t* p) {
    union { h2tn_s16x4 s[2]; h2_f16x8 v; } f;
    f.s[0] = tr(p);
    f.s[1] = tr(p + 1024);
    return f.v;
  };
  auto slot_of = [&](int u) { return lds + (size_t)(u & (H2TN_NSLOT - 1)) * H2TN_SLOT; };
  // the last k-step of a range that is not a multiple of 16 rows: the missing rows contribute zeros (their DMA re-read the
  // range's last row).  Called by every wavefront behind the barrier that made the slot visible.
  auto zero_tail = [&](int u) {
    uint8_t* sl = slot_of(u);
    const int nv = nrows & 15;
    for (int idx = tid; idx < (16 - nv) * 128; idx += 512) {
      const int r = nv + (idx >> 7), op = (idx >> 6) & 1, c = idx & 63;
      *reinterpret_cast<uint4*>(sl + op * H2TN_OPB + h2tn_row_off(r) + c * 16) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
  };
  const bool ragged = (nrows & 15) != 0;
  const int dsh = wid >> 2;   // wavefronts w and w + 4 share a SIMD

  // ---- software pipeline.  Ring of four slots: while step u is multiplied, step u + 1 has landed (its fragments are fetched
  // BETWEEN step u's MFMAs, each into the registers of the fragment it replaces, right behind that fragment's last use), step
  // u + 2 is in flight and step u + 3's four DMA instructions are issued one at a time between the MFMAs of the first product.
  // (Issued as a block behind the barrier, the DMA instructions held every wavefront for the whole fill -- a CU's DMA path takes
  // one 1 KB instruction per ~38 cycles, 32 of them per k-step -- and the MFMAs behind them started when the fill was done:
  // 174 us per 16 384 rows, 141 without the DMA, 82 without the MFMAs.  With the reads of a step in one block before / between
  // the products the two wavefronts of a SIMD, in lockstep behind the barrier, read at the same time and left the matrix pipe
  // idle: 197 us.)
  // Order of the three piece products (small terms first): a0 b1, then a1 b0, then a0 b0.  b1 is free behind the first product,
  // a1[i] behind its two MFMAs of the second, a0[i] behind its two of the third, b0 at the step's end -- and the next step does
  // not touch b0 before its second product.  Source order is pinned (sched_barrier): the interleave below is the schedule.
#pragma unroll
  for (int s = 0; s < H2TN_NSLOT - 1; ++s)
    if (s < nu) issue(s);
  if (nu >= 3) h2_wait_vm<2 * DPW>();
  else if (nu == 2) h2_wait_vm<DPW>();
  else h2_wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (nu == 1 && ragged) zero_tail(0);
  h2_f16x8 a0[TA], a1[TA], b0[TB], b1[TB];
  {
    const uint8_t* sl = slot_of(0);
#pragma unroll
    for (int j = 0; j < TB; ++j) { b0[j] = frag(sl + fb + j * 128); b1[j] = frag(sl + fb + j * 128 + 16); }
#pragma unroll
    for (int i = 0; i < TA; ++i) { a0[i] = frag(sl + fa + i * 128); a1[i] = frag(sl + fa + i * 128 + 16); }
  }
  auto step = [&](int u, auto nxt_c, auto more_c) {
    constexpr bool NXT = decltype(nxt_c)::value, MORE = decltype(more_c)::value;
    if (NXT) {
      // step u + 1 landed (this wavefront's part; behind the barrier: everybody's).  Every wavefront's reads of slot u - 1 --
      // which step u + 3 overwrites -- were issued during step u - 2 and have been consumed by step u - 1's MFMAs.
      if (MORE || u + 2 < nu) h2_wait_vm<DPW>();
      else h2_wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      if (!MORE && u + 1 == nu - 1 && ragged) zero_tail(u + 1);
    }
    const uint8_t* sn = slot_of(u + 1);
    Pieces pc = {};
    if (MORE) pc = pieces(u + H2TN_NSLOT - 1);
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool mm = !(DBG & 2);
    // twelve MFMA pairs; behind pair k: the fragment reads that pair has freed, and -- one per three pairs, the two wavefronts
    // of a SIMD (w, w + 4) a pair apart -- a DMA instruction of step u + 3 (all four in the first product, the 32 instructions of
    // a CU's step met its DMA path, one 1 KB instruction per ~38 cycles, within ~500 cycles and the issuing wavefronts stalled)
    auto dma_at = [&](int k) {
      if (!MORE) return;
      if (k % 3 == 1) { if (dsh == 0) piece(pc, k / 3); }
      else if (k % 3 == 2) { if (dsh == 1) piece(pc, k / 3); }
    };
    // product 1: a0 b1
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b1[j], acc[i][j], 0, 0, 0);
      }
      dma_at(i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // product 2: a1 b0; behind a1[i]'s MFMAs its successor, and b1's in the first two gaps
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b0[j], acc[i][j], 0, 0, 0);
      }
      if (NXT) {
        if (i < TB) b1[i] = frag(sn + fb + i * 128 + 16);
        a1[i] = frag(sn + fa + i * 128 + 16);
      }
      dma_at(TA + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // product 3: a0 b0
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
      }
      if (NXT) a0[i] = frag(sn + fa + i * 128);
      dma_at(2 * TA + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NXT) {
#pragma unroll
      for (int j = 0; j < TB; ++j) b0[j] = frag(sn + fb + j * 128);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    int u = 0;
    for (; u + H2TN_NSLOT - 1 < nu; ++u) step(u, std::true_type{}, std::true_type{});
    for (; u + 1 < nu; ++u) step(u, std::true_type{}, std::false_type{});
    step(u, std::false_type{}, std::false_type{});
  }

  // ---- this range's slab: block (i, j), register r, lane l -> dW[o][c], o = A channel of row index (r & 3) + 8 (r >> 2) + 4 h,
  // c = B channel of column l & 31
  if (DBG & 4) return;
  const float inv = 1.f / (*g.sa * *g.sb);
  float* slab = g.slabs + (size_t)sp * (size_t)g.NA * (size_t)g.NB;
  const int cl = h2tn_ch(lane & 31);
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int c = (int)tb * 256 + (wb * TB + j) * 32 + cl;
      const int ob = (int)ta * 256 + (wa * TA + i) * 32;
      if (c >= g.NB) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ob + h2tn_ch((r & 3) + 8 * (r >> 2) + 4 * h);
        if (o < g.NA) slab[(size_t)o * (size_t)g.NB + c] = acc[i][j][r] * inv;
      }
    }
}

// rows per range and the number of ranges for `tiles` output tiles: one round of workgroups on the chip's 256 CUs
inline void h2tn_plan(int64_t M, int tiles, int* splits, int64_t* rows_per_split) {
  int s = 256 / (tiles > 0 ? tiles : 1);
  if (s < 1) s = 1;
  int64_t rp = ((M + s - 1) / s + 31) / 32 * 32;
  if (rp < 32) rp = 32;
  s = (int)((M + rp - 1) / rp);
  *splits = s;
  *rows_per_split = rp;
}

template <int DBG = 0>
inline int h2tn_launch(hipStream_t st, H2TnArgs a) {
  a.tiles_a = (a.NA + 255) / 256;
  a.tiles_b = (a.NB + 255) / 256;
  const long nblk = (long)a.tiles_a * a.tiles_b * a.splits;
  if (nblk <= 0 || nblk > 0x7fffffffL) return -22;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h2tn_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, H2TN_NSLOT * H2TN_SLOT);
    attr_set = true;
  }
  hipLaunchKernelGGL(h2tn_kernel<DBG>, dim3((unsigned)nblk), dim3(512), H2TN_NSLOT * H2TN_SLOT, st, a);
  return 0;
}

#endif  // __HIPCC__

}  // namespace srlh2
