// Dense products over pre-split operands, PERSISTENT form (round 6): out[M, NC] = epilogue(X[M, K] W[NC, K]^T), both operands h2p
// rows (h2gemm.h), for a WIDE output over a SHORT reduction -- the data gradient of the encoder's Linear (3136 channels, K = 512):
// 832 tiles of 256 x 256 whose k-loop is 16 k-blocks long.  With one workgroup per tile (h2gemm_kernel<8, DENSE, 2, false>) a
// tile's ring fill latency, its k-loop and its 256 KB of stores ran one after the other on the CU (round 5's leave-outs: 89 + 45 +
// 84 us of 207 per 16 384 rows), 3.25 rounds of them.
//
// Here a workgroup stays on its CU and walks its tiles, and the (tile, k-step) pairs form ONE sequence:
//   * ring of four 32 KB slots, a slot = one k-step of 16 k-values: 256 rows of X and 256 rows of W, 64 bytes each (one half of
//     the 128-byte h2p block).  The DMA fills slots in PAIRS -- both halves of the same 128-byte lines by neighbouring
//     instructions (a half fetched a k-step later has left the 32 KB L1 and costs a second L2 read: the fill IS the L2's rate) --
//     and the pair of the NEXT tile's first k-block is issued while the last k-block of this tile is multiplied: no tile starts
//     with an empty ring;
//   * two barriers per pair of k-steps: one in front of it (the pair has landed), one between its steps (every wavefront holds the
//     pair's fragments in registers -- the odd step's are fetched between the MFMAs of the even one, each into the registers of the
//     fragment it replaces, behind that fragment's last use: h2tn.h's schedule -- so the pair's slots are free and the pair after
//     the next is issued between the MFMAs of the odd step);
//   * a tile's stores are not waited for.  The VM counter completes in order, so the waits in front of the next tile's first TWO
//     pairs (issued before the stores) are vmcnt(number of stores + ...) -- exact, because every wavefront issues every store
//     instruction of every tile, rows and channels beyond the matrix going to an out-of-range buffer offset; the stores have the
//     two pairs up to the wait for the next tile's third pair to drain into L2;
//   * epilogue as h2gemm.h's (lane = position; bias, activation, ReLU masks in / out, measured range, float32 or h2p rows,
//     whole 128-byte lines per store instruction through the wavefront's own 4 KB of LDS), outside the ring: 4 x 32 KB + 8 x 4 KB
//     = the CU's 160 KB.
// Wavefront w: position half w & 1 (128 rows = 4 blocks), channel quarter w >> 1 (64 channels = 2 blocks): 12 fragment reads
// (ds_read_b128) and 24 MFMAs per k-step.
#pragma once
#include "h2gemm.h"
#include <type_traits>

namespace srlh2 {

#ifdef __HIPCC__

constexpr int H2P_SLOT = 32768, H2P_NSLOT = 4, H2P_WOFF = 16384, H2P_STG = H2P_NSLOT * H2P_SLOT, H2P_LDS = H2P_STG + 8 * 4096;

// DBG (timing experiments, wrong results): 1 no DMA, 2 no MFMAs, 4 no stores
template <int DBG>
__global__ __launch_bounds__(512, 2) void h2gemmp_kernel(H2Args g) {
  constexpr int NP = 4, NQ = 2;      // accumulator blocks per wavefront: positions x channels
  constexpr int DPP = 8;             // DMA instructions per wavefront and PAIR of k-steps (2 operands x 2 row groups x 2 halves)
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wid & 1, wc = wid >> 1;

  // ---- this workgroup's tiles: every XCD owns one contiguous run of the tile sequence (channel tile fastest: neighbours share
  // their rows of X), its workgroups take the run's tiles round-robin
  const unsigned tiles_c = (unsigned)g.tiles_c;
  const unsigned nt_all = (unsigned)((g.M + 255) / 256) * tiles_c;
  unsigned run0, runlen;
  {
    const unsigned q = nt_all >> 3, r = nt_all & 7u, xcd = blockIdx.x & 7u;
    run0 = xcd * q + (xcd < r ? xcd : r);
    runlen = q + (xcd < r ? 1u : 0u);
  }
  const unsigned wslot = blockIdx.x >> 3, wstride = gridDim.x >> 3;   // (grid: a multiple of 8)
  const int ntl = wslot < runlen ? (int)((runlen - wslot + wstride - 1) / wstride) : 0;   // tiles of this workgroup
  float oscale = 1.f;
  if (g.out_fmt == H2O_H2P) {
    float bound = *g.bound_in * *g.bound_w;
    if (g.bound_b) bound += *g.bound_b;
    oscale = h2_scale_for(bound);
    if (blockIdx.x == 0 && tid == 0) *g.out_scale = oscale;
  }
  if (ntl == 0) return;
  const int npair = g.nk;                       // pairs of k-steps per tile (one 32-k block each)
  const int ptotal = ntl * npair;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const uint32_t ldma = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wid * 1024u);   // + slot * SLOT + operand * WOFF + q * 8192

  // ---- DMA stream: tile kd, pair pd of it
  // instruction (operand, q): rows 16 (wid + 8 q) .. + 15 of the tile, lane i -> row + (i >> 2), 16-byte piece (i & 3) ^ ((i >> 4) & 3)
  // (the row's swizzle, h2gemm.h HALF: (row >> 2) & 3) of the k-step's 64 bytes
  int kd = 0, pd = 0;
  uint32_t xv[2], wv[2];
  h2_i32x4 rx = h2_rsrc(g.x), rw = h2_rsrc(g.w);
  auto tile_of = [&](int k, unsigned& tp, unsigned& tc) {
    const unsigned t = run0 + wslot + (unsigned)k * wstride;
    tp = t / tiles_c;
    tc = t - tp * tiles_c;
  };
  auto dma_tile = [&](int k) {
    unsigned tp, tc;
    tile_of(k, tp, tc);
    const long m0 = (long)tp * 256;
    const int c0 = (int)tc * 256;
    const long mrem = g.M - m0;
    const int crem = g.NC - c0;
    const int piece = 16 * ((lane & 3) ^ ((lane >> 4) & 3));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 16 * (wid + 8 * q) + (lane >> 2);
      const long rx_ = row < mrem ? row : mrem - 1;
      const int rw_ = row < crem ? row : crem - 1;
      xv[q] = (uint32_t)(rx_ * (long)g.x_row_bytes) + (uint32_t)piece;
      wv[q] = (uint32_t)rw_ * g.w_row_bytes + (uint32_t)piece;
    }
    rx = h2_rsrc(static_cast<const uint8_t*>(g.x) + m0 * (long)g.x_row_bytes);
    rw = h2_rsrc(static_cast<const uint8_t*>(g.w) + (long)c0 * (long)g.w_row_bytes);
  };
  auto dma1 = [&](uint32_t m, uint32_t voff, h2_i32x4 rsrc, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
  };
  // piece i (0..7) of the pair the DMA stream stands at: (operand, q, half) = (i >> 2, (i >> 1) & 1, i & 1); the two halves of a
  // line are neighbours
  uint32_t dslot2 = 0;    // first slot of the pair being issued (0 or 2)
  auto piece = [&](int i) {
    if (DBG & 1) return;
    const int op = i >> 2, q = (i >> 1) & 1, half = i & 1;
    const uint32_t m = ldma + (dslot2 + (uint32_t)half) * H2P_SLOT + (uint32_t)op * H2P_WOFF + (uint32_t)q * 8192u;
    const uint32_t soff = (uint32_t)pd * 128u + (uint32_t)half * 64u;
    if (op == 0) dma1(m, xv[q], rx, soff);
    else dma1(m, wv[q], rw, soff);
  };
  auto advance_dma = [&]() {   // behind the last piece of a pair
    dslot2 ^= 2u;
    if (++pd == npair) {
      pd = 0;
      if (++kd < ntl) dma_tile(kd);
    }
  };

  // ---- fragment addresses inside a slot: row r = lane & 31 of a 32-row block, pieces (2 hl + plane) ^ ((r >> 2) & 3)
  const int hl = lane >> 5, pl = lane & 31;
  const uint32_t f0 = (uint32_t)(pl * 64 + 16 * ((2 * hl + 0) ^ ((pl >> 2) & 3)));
  const uint32_t f1 = (uint32_t)(pl * 64 + 16 * ((2 * hl + 1) ^ ((pl >> 2) & 3)));
  const uint32_t fx = (uint32_t)(wp * NP * 2048), fw = (uint32_t)(H2P_WOFF + wc * NQ * 2048);
  auto ldf = [&](const uint8_t* sl, uint32_t off) { return *reinterpret_cast<const h2_f16x8*>(sl + off); };

  h2_f32x16 acc[NP][NQ];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  h2_f16x8 p0[NP], p1[NP], q0[NQ], q1[NQ];   // first / second pieces of the X (positions) and W (channels) fragments

  // one k-step.  PRE: fetch the next k-step's fragments (slot sn) between the MFMAs; ISS (uniform): issue the DMA stream's pair.
  // Products (small terms first): p0 q1, then p1 q0, then p0 q0 -- q1 is free behind the first, p1[i] behind its MFMAs of the
  // second, p0[i] behind its MFMAs of the third, q0 at the end.
  auto mm_step = [&](const uint8_t* sn, auto pre_c, const bool ISS) {
    constexpr bool PRE = decltype(pre_c)::value;
    constexpr bool mm = !(DBG & 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q1[j], p0[i], acc[i][j], 0, 0, 0);
      }
      if (ISS) piece(i);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0[j], p1[i], acc[i][j], 0, 0, 0);
      }
      if (PRE) {
        if (i < NQ) q1[i] = ldf(sn, fw + i * 2048 + f1);
        p1[i] = ldf(sn, fx + i * 2048 + f1);
      }
      if (ISS) piece(NP + i);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (mm) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(q0[j], p0[i], acc[i][j], 0, 0, 0);
      }
      if (PRE) p0[i] = ldf(sn, fx + i * 2048 + f0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (PRE) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) q0[j] = ldf(sn, fw + j * 2048 + f0);
    }
    if (ISS) advance_dma();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- epilogue constants
  const float inv = 1.f / (*g.sx * *g.sw);
  float amax = 0.f, amax_raw = 0.f;
  const bool fast = g.out_fmt == H2O_H2P && !g.bias && g.act == 0 && !g.mask_out;
  constexpr uint32_t OOB = 0x80000000u;
  const bool has_mo = g.mask_out != nullptr;

  auto epilogue = [&](int k) {
    unsigned tp, tc;
    tile_of(k, tp, tc);
    const long m0 = (long)tp * 256;
    const int c0 = (int)tc * 256;
    const long mrem = g.M - m0;                          // rows of this tile that exist (> 0)
    const int nrow = mrem < 256 ? (int)mrem : 256;
    // stores through a descriptor of THIS tile's rows (row pitch x 256 < 2^31 for every layer here; host-checked)
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint8_t*>(g.out) + m0 * (long)g.out_row_bytes, 0,
                                                                           (int)((long)nrow * (long)g.out_row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_mo = __builtin_amdgcn_make_buffer_rsrc(has_mo ? (void*)(g.mask_out + (m0 * (long)g.out_row_bytes) / 128) : g.out, 0,
                                                                          (int)((long)nrow * (long)(g.out_row_bytes / 32)), 0x00020000);
    uint8_t* tb = lds + H2P_STG + wid * 4096;
    // (the lane's coordinates through an opaque copy: computed from `lane` itself, every address and shift of the epilogue is
    // loop-invariant, gets hoisted in front of the tile loop and is kept alive through the k-loop -- the accumulators were spilled
    // to make room: 486 registers)
    int lv = lane;
    asm volatile("" : "+v"(lv));
    const int hl = lv >> 5, pl = lv & 31;
    // ReLU-derivative words of this lane's (row, channel block)s, fetched together
    uint32_t mw[NP][NQ];
    if (g.mask_in) {
#pragma unroll
      for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          const int rowl = wp * 128 + i * 32 + pl;
          const int cb = c0 / 32 + wc * NQ + j;
          const bool ok = rowl < nrow && cb * 32 < g.NC;
          const long elem0 = ((m0 + rowl) * (long)g.out_row_bytes) / 4 + (long)cb * 32;
          mw[i][j] = ok ? g.mask_in[elem0 >> 5] : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const int rowl = wp * 128 + i * 32 + pl;         // row inside the tile
        const int cb = c0 / 32 + wc * NQ + j;            // global channel block
        const bool okc = cb * 32 < g.NC;
        const bool ok = rowl < nrow && okc;
        float v[16];
        if (fast) {
          // h2p output with nothing between the sums and the split (a data gradient): mask the raw sums, track their range, and let
          // the split's multiply carry both scales (powers of two) -- ~4.5 vector instructions per value instead of ~10, and the
          // epilogue is a third of this kernel's time that nothing overlaps (both wavefronts of a SIMD are in it together)
          if (g.mask_in) {
            if (g.mask_in_h2) {
              const uint32_t m = mw[i][j] >> (16 * hl);
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(__float_as_uint(acc[i][j][r]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, r, 1));
            } else {
              const uint32_t m = mw[i][j] >> (4 * hl);
#pragma unroll
              for (int r = 0; r < 16; ++r)
                v[r] = __uint_as_float(__float_as_uint(acc[i][j][r]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, (r & 3) + 8 * (r >> 2), 1));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
          }
          // (rows / channels beyond the matrix repeat its last row / channel or are masked to zero: they cannot raise the maximum)
#pragma unroll
          for (int r = 0; r < 16; r += 2) amax_raw = __builtin_fmaxf(__builtin_fmaxf(fabsf(v[r]), fabsf(v[r + 1])), amax_raw);
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] * inv;
        if (g.bias && okc) {   // registers 4q .. 4q+3: channels 8q + 4 hl + {0..3} of the block
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
          if (g.mask_in_h2) {   // h2 order: registers 0-7 are group 2 hl, 8-15 group 2 hl + 1: bit 16 hl + r
            const uint32_t m = mw[i][j] >> (16 * hl);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(__float_as_uint(v[r]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, r, 1));
          } else {              // natural order: channel (r & 3) + 8 (r >> 2) + 4 hl of the block
            const uint32_t m = mw[i][j] >> (4 * hl);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = __uint_as_float(__float_as_uint(v[r]) & (uint32_t)__builtin_amdgcn_sbfe((int)m, (r & 3) + 8 * (r >> 2), 1));
          }
        }
        if (has_mo) {
          uint32_t bits = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) bits |= (v[r] > 0.f ? 1u : 0u) << ((r & 3) + 8 * (r >> 2) + 4 * hl);
          bits |= (uint32_t)__shfl_xor((int)bits, 32);
          const uint32_t moff = (ok && hl == 0 && !(DBG & 4)) ? (uint32_t)rowl * (g.out_row_bytes / 32) + (uint32_t)cb * 4u : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(bits, r_mo, moff, 0, 0);
        }
        if (g.out_absmax) {
#pragma unroll
          for (int r = 0; r < 16; ++r) amax = fmaxf(amax, ok ? fabsf(v[r]) : 0.f);
        }
        }
        const float ssc = fast ? oscale * inv : oscale;
        // whole lines per store instruction, through the wavefront's own 4 KB (h2gemm.h)
        uint4* trow = reinterpret_cast<uint4*>(tb + pl * 128);
        const int sw = pl & 7;
        if (g.out_fmt == H2O_H2P) {
          uint4 h0a, h1a, h0b, h1b;
          h2_split_pair(v[0], v[1], ssc, h0a.x, h1a.x);
          h2_split_pair(v[2], v[3], ssc, h0a.y, h1a.y);
          h2_split_pair(v[4], v[5], ssc, h0a.z, h1a.z);
          h2_split_pair(v[6], v[7], ssc, h0a.w, h1a.w);
          h2_split_pair(v[8], v[9], ssc, h0b.x, h1b.x);
          h2_split_pair(v[10], v[11], ssc, h0b.y, h1b.y);
          h2_split_pair(v[12], v[13], ssc, h0b.z, h1b.z);
          h2_split_pair(v[14], v[15], ssc, h0b.w, h1b.w);
          trow[(4 * hl + 0) ^ sw] = h0a;
          trow[(4 * hl + 1) ^ sw] = h1a;
          trow[(4 * hl + 2) ^ sw] = h0b;
          trow[(4 * hl + 3) ^ sw] = h1b;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            trow[(2 * q + hl) ^ sw] = make_uint4(__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]),
                                                 __float_as_uint(v[4 * q + 3]));
        }
        const int rr = lv >> 3, pp = lv & 7;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const uint4 val = *reinterpret_cast<const uint4*>(tb + (8 * it + rr) * 128 + 16 * (pp ^ rr));
          const int rl = wp * 128 + i * 32 + 8 * it + rr;
          const uint32_t off = (rl < nrow && okc && !(DBG & 4)) ? (uint32_t)rl * g.out_row_bytes + (uint32_t)cb * 128u + 16u * (uint32_t)pp : OOB;
          const u32x4 d = {val.x, val.y, val.z, val.w};
          __builtin_amdgcn_raw_buffer_store_b128(d, r_out, off, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // one block at a time (scheduled across the blocks the epilogue spilled 486 registers)
      }
  };

  // ---- the sequence
  dma_tile(0);
  {
    // pairs 0 and 1 (a tile has at least one pair; a second tile's first may follow at once)
#pragma unroll
    for (int i = 0; i < DPP; ++i) piece(i);
    advance_dma();
    if (ptotal > 1) {
#pragma unroll
      for (int i = 0; i < DPP; ++i) piece(i);
      advance_dma();
    }
  }
  // Timeline of pair pc (slots A = its two, B = the other two):  barrier -- even step (A's even fragments read at its start, A's
  // odd fragments fetched between its MFMAs) -- lgkmcnt(0), barrier: every wavefront holds all of pair pc in registers, A is
  // free -- odd step, issuing pair pc + 2 into A.  A pair is issued one and a half pairs before it is waited for (issued during
  // the even step of the pair before it -- the first version -- it had half a pair to a pair: 193 us per 16 384 rows, 162 without
  // the DMA).  In front of pair pc's barrier the VM counter holds, younger than pair pc: pair pc + 1 (if it exists) and the stores
  // of every tile that ended behind pair pc - 2 or pc - 1.
  int pc = 0;   // global pair index of the compute stream
  const int nst = has_mo ? 40 : 32;   // store instructions per wavefront and tile
  for (int k = 0; k < ntl; ++k) {
    for (int p = 0; p < npair; ++p, ++pc) {
      {
        // tiles that ended at pair pc - 1 (p == 0) or pc - 2 (p == 1, or p == 0 with one pair per tile)
        int ends = 0;
        if (pc >= 1 && p == 0) ++ends;
        if (pc >= 2 && (p == 1 || (p == 0 && npair == 1))) ++ends;
        const bool nextp = pc + 1 < ptotal;
        // (an immediate smaller than the true count only waits for a few of the youngest stores as well)
        if (ends == 0) { if (nextp) h2_wait_vm<DPP>(); else h2_wait_vm<0>(); }
        else if (ends == 1) {
          if (nextp) { if (nst == 40) h2_wait_vm<DPP + 40>(); else h2_wait_vm<DPP + 32>(); }
          else { if (nst == 40) h2_wait_vm<40>(); else h2_wait_vm<32>(); }
        } else {
          if (nextp) h2_wait_vm<63>(); else if (nst == 40) h2_wait_vm<63>(); else h2_wait_vm<63>();
        }
      }
      __builtin_amdgcn_s_barrier();
      const uint8_t* se = lds + (size_t)((pc & 1) * 2) * H2P_SLOT;
      // the even step's fragments
#pragma unroll
      for (int j = 0; j < NQ; ++j) q1[j] = ldf(se, fw + j * 2048 + f1);
#pragma unroll
      for (int i = 0; i < NP; ++i) p0[i] = ldf(se, fx + i * 2048 + f0);
#pragma unroll
      for (int j = 0; j < NQ; ++j) q0[j] = ldf(se, fw + j * 2048 + f0);
#pragma unroll
      for (int i = 0; i < NP; ++i) p1[i] = ldf(se, fx + i * 2048 + f1);
      // (ONE copy of a step's MFMAs, the DMA instructions behind uniform branches: as two instantiations under an if / else the
      // accumulators met in phi nodes the register allocator did not coalesce -- MFMAs with D != C, 112 registers spilled per pair)
      mm_step(se + H2P_SLOT, std::true_type{}, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      mm_step(se, std::false_type{}, pc + 2 < ptotal);
    }
    epilogue(k);
    zero_acc();
  }
  if (g.out_absmax) {
    amax = fmaxf(amax, amax_raw * inv);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) {
      const float cur = __hip_atomic_load(g.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (amax > cur) atomicMax(reinterpret_cast<int*>(g.out_absmax), __float_as_int(amax));  // non-negative floats order as ints
    }
  }
}

template <int DBG = 0>
inline int h2gemmp_launch(hipStream_t st, H2Args a, int ncu = 256) {
  a.tiles_c = (a.NC + 255) / 256;
  const long nt = ((a.M + 255) / 256) * (long)a.tiles_c;
  if (nt <= 0 || nt > 0x7fffffffL || a.nk < 1) return -22;
  if ((long)a.out_row_bytes * 256 >= 0x7fffffffL || (long)a.x_row_bytes * 256 >= 0x7fffffffL || (long)a.w_row_bytes * 256 >= 0x7fffffffL) return -22;
  long grid = nt < ncu ? ((nt + 7) / 8) * 8 : ncu;   // a multiple of 8; workgroups beyond their XCD's run leave at once
  static bool attr_set = false;
  auto kern = h2gemmp_kernel<DBG>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, H2P_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), H2P_LDS, st, a);
  return 0;
}

#endif  // __HIPCC__

}  // namespace srlh2
