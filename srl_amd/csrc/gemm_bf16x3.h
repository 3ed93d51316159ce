// float32 contractions on the BF16 matrix cores: C = epilogue( sum_k A(i,k) B(k,j) ) with float32 operands and results.
//
// gfx950 multiplies float32 on the matrix cores at the vector rate (157 TFLOP/s, v_mfma_f32_32x32x2_f32) and bf16 at
// 16x that.  A float32 splits EXACTLY into three bf16 pieces, a = a0 + a1 + a2 (top / middle / low 8 significand bits;
// truncation splits, every subtraction exact), and a product of two pieces (8 x 8 bits) is exact in the MFMA's float32
// accumulate.  Of the nine piece products the six with i + j <= 2 are formed,
//     a b  =  a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0   + O(2^-23 |a b|)        (dropped: a1 b2 + a2 b1 + a2 b2)
// i.e. every product is good to 2^-23 relative -- a float32 multiply rounds at 2^-24 -- and the sums are float32 like the
// float32 MFMA's (6 accumulations per 16 k-values instead of 16).  6 bf16 MFMAs of 32 cycles replace 8 float32 MFMAs of
// 64: 2.67x fewer matrix-pipe cycles.  No scaling is needed (bf16 has float32's exponent range).
//
// Operands stay float32 in HBM.  The global -> register half of the staging (addresses, gathers of the implicit
// convolutions, zero padding) is gemm_core.h's Stage::prepare / Stage::load unchanged; what is new is everything behind
// it: the split when a tile is written to LDS (three planes of bf16), the fragment reads (ds_read_b128 for an operand
// that is contiguous along k; ds_read_b64_tr_b16, the transposing LDS read of gfx950, for one that is contiguous along
// the output index -- no transposing stores), and the MFMA loop.  The epilogue is the shared one.
//
// NP == 2 (every product whose operands both have a tracked range -- all contractions of the benchmarked update): two f16 pieces per operand instead of three
// bf16 ones, a = h0 + h1 + O(2^-23 |a|) (h0 = f16(a s), h1 = f16(a s - h0), s a power of two that brings the operand into
// f16's range; 11-bit significands: 22 bits in two planes), and the THREE products h0 h0' + h0 h1' + h1 h0' -- each exact
// in the float32 accumulate (22 bits) -- instead of six: half the matrix-pipe cycles, two thirds of the LDS traffic, a
// split of ~3 instead of 6.5 vector operations per element.  The price is f16's exponent range: an element more than 2^16
// below the operand's largest keeps less than 22 bits (its error is bounded by 2^-40 of that largest element instead).
#pragma once
#include "gemm_core.h"

namespace srlgemm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// timing experiments only (scripts/build_variant.sh -DSRL_GEMM3_DBG=<bits>; results are wrong): 1 no in-loop global loads,
// 2 no in-loop LDS writes, 4 no split arithmetic, 8 no barrier, 16 no epilogue, 128 one k-step only, 32 / 64 every gather
// redirected into a 16 KB / 1 MB window (gemm_core.h bload4).  DESIGN.md section 4 records what they showed.
#ifndef SRL_GEMM3_DBG
#define SRL_GEMM3_DBG 0
#endif
// 1: the global loads of two consecutive k-steps are issued together (see gemm3_kernel); 0: one tile per step
#ifndef SRL_GEMM3_PAIR
#define SRL_GEMM3_PAIR 1
#endif

#ifdef __HIPCC__
// one operand tile in LDS: three planes of BX x KB bf16
template <int BX, bool KMAJOR, int KB, int NP = 3>
struct Tile3 {
  static constexpr int PLANE = BX * KB * 2;  // bytes
  static constexpr int BYTES = NP * PLANE;
  // k-contiguous: rows of KB bf16 (PB bytes), their 16-byte chunks XOR-swizzled so that the 16 lanes of a ds_read_b128
  // group (rows 4 apart in the same chunk) land on distinct 16-byte slots of the 256-byte bank line
  static constexpr int PB = KB * 2, CPR = PB / 16, RPL = 256 / PB > 0 ? 256 / PB : 1;
  __device__ static __forceinline__ int off_kc(int x, int k) {  // byte offset of element (row x, k)
    const int c = (k * 2) >> 4;
    return x * PB + 16 * (c ^ ((x / RPL) % CPR)) + ((k * 2) & 15);
  }
  // k-major: rows of BX bf16 (RB bytes); 64-byte pieces XOR-swizzled so that the four k-rows a transposed read takes
  // (same columns, consecutive k) land on the four 64-byte quarters of the bank line
  static constexpr int RB = BX * 2, P64 = RB / 64;
  __device__ static __forceinline__ int off_km(int k, int x) {  // byte offset of element (k-row k, column x)
    const int b = x * 2;
    int pc = b >> 6;
    if (P64 >= 4) pc ^= (k & 3);
    else if (P64 == 2) pc ^= ((k >> 1) & 1);
    return k * RB + 64 * pc + (b & 63);
  }
};

__device__ __forceinline__ void split3_quad(const float* v, uint2 (&pl)[3]) {
  uint32_t b[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (SRL_GEMM3_DBG & 4) { b[0][i] = b[1][i] = b[2][i] = __float_as_uint(v[i]); continue; }
    b[0][i] = __float_as_uint(v[i]) & 0xffff0000u;
    const float r1 = v[i] - __uint_as_float(b[0][i]);
    b[1][i] = __float_as_uint(r1) & 0xffff0000u;
    b[2][i] = __float_as_uint(r1 - __uint_as_float(b[1][i]));  // at most 8 significant bits: its high half is all of it
  }
#pragma unroll
  for (int p = 0; p < 3; ++p)
    pl[p] = make_uint2(__builtin_amdgcn_perm(b[p][1], b[p][0], 0x07060302u), __builtin_amdgcn_perm(b[p][3], b[p][2], 0x07060302u));
}

// two f16 pieces of v * scale: h0 = f16(v * scale) (round to nearest), h1 = f16(v * scale - h0) (the difference is exact in
// float32; scale is a power of two).  Written as fused multiply-adds so that each piece is ONE mixed-precision instruction
// (v_fma_mixlo / mixhi_f16: float32 product and sum, f16 result): two vector operations per element.
__device__ __forceinline__ void split2h_quad(const float* v, float scale, uint2 (&pl)[2]) {
  f16x2 h0[2], h1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const _Float16 a0 = (_Float16)__builtin_fmaf(v[2 * i], scale, 0.f), a1 = (_Float16)__builtin_fmaf(v[2 * i + 1], scale, 0.f);
    h0[i] = f16x2{a0, a1};
    h1[i] = f16x2{(_Float16)__builtin_fmaf(v[2 * i], scale, -(float)a0), (_Float16)__builtin_fmaf(v[2 * i + 1], scale, -(float)a1)};
  }
  union { f16x2 h; uint32_t u; } a0, a1, b0, b1;
  a0.h = h0[0]; a1.h = h0[1]; b0.h = h1[0]; b1.h = h1[1];
  pl[0] = make_uint2(a0.u, a1.u);
  pl[1] = make_uint2(b0.u, b1.u);
}

// registers of a staged tile (Stage::r, float4 quads in Stage's thread -> (row, k) assignment) -> NP planes of 16-bit
// pieces in LDS (three bf16 planes, or two f16 planes of the scaled operand)
template <class ST, int BX, bool KMAJOR, int KB, int NT, int NP = 3, int RAW = 0>
__device__ __forceinline__ void store3(const float* regs, uint8_t* lds, float scale = 1.f) {
  using T3 = Tile3<BX, KMAJOR, KB, NP>;
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < ST::NV; ++q) {
    const int u = tid + q * NT;
    if (ST::PARTIAL && u >= ST::QUADS) continue;
    if (RAW == 2) {
      // the operand is h2p rows (h2gemm.h): the 16 bytes this thread loaded at columns x .. x + 3 of k-row k are 8 elements of ONE
      // piece -- slot (x / 4) % 8 of the 32-column block x / 32: group g = slot >> 1, piece slot & 1, columns 4 (g >> 1) + 16 (g & 1)
      // + {0..3} and + {8..11} of the block -- and go to that piece's plane as two 8-byte runs
      static_assert(RAW != 2 || (KMAJOR && NP == 2), "h2p rows: a k-major operand of the two-piece kernel");
      const int k = u / (BX / 4), x = (u % (BX / 4)) * 4;
      const int sl = (x >> 2) & 7, g = sl >> 1, c0 = (x & ~31) + 4 * (g >> 1) + 16 * (g & 1);
      uint8_t* pb = lds + (sl & 1) * T3::PLANE;
      *reinterpret_cast<uint2*>(pb + T3::off_km(k, c0)) = make_uint2(__float_as_uint(regs[4 * q]), __float_as_uint(regs[4 * q + 1]));
      *reinterpret_cast<uint2*>(pb + T3::off_km(k, c0 + 8)) = make_uint2(__float_as_uint(regs[4 * q + 2]), __float_as_uint(regs[4 * q + 3]));
      continue;
    }
    uint2 pl[3];
    if (RAW == 1) {  // the operand arrives already split (BPRE; SRL_GEMM3_DBG & 256 as a timing experiment): two pieces per 16 bytes
      pl[0] = make_uint2(__float_as_uint(regs[4 * q]), __float_as_uint(regs[4 * q + 1]));
      pl[1] = make_uint2(__float_as_uint(regs[4 * q + 2]), __float_as_uint(regs[4 * q + 3]));
      pl[2] = pl[0];
    } else if (NP == 3) split3_quad(regs + 4 * q, pl);
    else split2h_quad(regs + 4 * q, scale, reinterpret_cast<uint2(&)[2]>(pl));
    int off;
    if (!KMAJOR) off = T3::off_kc(u / ST::KQ, (u % ST::KQ) * 4);
    else off = T3::off_km(u / (BX / 4), (u % (BX / 4)) * 4);
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(lds + p * T3::PLANE + off) = pl[p];
  }
}

// the lane's fragment (8 consecutive k of one row / column) of 32-wide block `blk`, k-block kb, plane p
template <int BX, bool KMAJOR, int KB, int NP = 3>
__device__ __forceinline__ bf16x8 frag3(const uint8_t* lds, int p, int x0, int kb, int lane) {
  using T3 = Tile3<BX, KMAJOR, KB, NP>;
  const uint8_t* base = lds + p * T3::PLANE;
  if (!KMAJOR) {
    union { uint4 u; bf16x8 v; } f;
    f.u = *reinterpret_cast<const uint4*>(base + T3::off_kc(x0 + (lane & 31), 16 * kb + 8 * (lane >> 5)));
    return f.v;
  }
  // ds_read_b64_tr_b16: in each group of 16 lanes, lane 4q + p4 supplies the address of k-row q, columns 4 p4 .. + 3 of a
  // 4 x 16 block; lane i of the group receives column i (4 k-rows).  A lane with half h needs k-rows 8h .. 8h + 7.
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const int grp = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, h = lane >> 5;
  const int col = x0 + 16 * (grp & 1) + 4 * p4;
  union { s16x4 s[2]; bf16x8 v; } f;
  f.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + T3::off_km(16 * kb + 8 * h + q, col)));
  f.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + T3::off_km(16 * kb + 8 * h + 4 + q, col)));
  return f.v;
}

constexpr int min_waves3(int bm, int bn, int kb, int np = 3) {
  const int lds = 2 * np * (bm + bn) * kb * 2;  // two buffers of two np-plane tiles
  const int by_lds = 160 * 1024 / lds;
  return by_lds >= 3 ? 3 : (by_lds >= 2 ? 2 : 1);
}

// BPRE (NP == 2, dense B): B arrives ALREADY split -- every 16 bytes of it hold the two f16 pieces of four consecutive elements
// (srl_presplit: same addresses, same strides as the float32 matrix) -- so its tiles go from the registers to LDS as they
// are.  Weights are split once per update instead of once per tile that stages them: leaving B's split out of the kernels
// (timing experiment) gave 5-6.5 % on the convolutions and 13-15 % on the FC products.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AMODE, int BMODE, int KB, int NP = 3, int MK = 0, int BPRE = 0>
__global__ __launch_bounds__(WM * WN * 64, min_waves3(BM, BN, KB, NP)) void gemm3_kernel(GemmArgs g) {
  static_assert(NP == 3 || NP == 2, "three bf16 pieces or two f16 pieces");
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 wavefronts per workgroup");
  static_assert(!is_obs(AMODE) && !is_obs(BMODE), "observation sources have their own bf16 kernels (obs_bf16.h)");
  constexpr int NT = 256;
  // PAIR: two register sets per operand, and the global loads of two consecutive k-steps issued TOGETHER, every other
  // step.  A k-contiguous operand is read 64 bytes per row and k-step, i.e. half a cache line, the other half one k-step
  // (~2 us) later -- by then the line has left the 32 KB L1 (three workgroups stream 16 KB per step each) and is requested
  // from L2 a second time.  Issued together the second request hits: dense products +2.5...4 % (FC forward 0.271 -> 0.264
  // ms, data gradient 0.285 -> 0.275, with the mask 0.315 -> 0.301; same box).  Dense operands only: the convolutions'
  // gathers did not move (forward) or lost (conv3 data gradient 0.486 -> 0.506), and products whose operands are both
  // k-major read whole lines anyway.
  constexpr bool PAIR = SRL_GEMM3_PAIR && KB == 16 && !(AKM && BKM) && AMODE == SRC_PLAIN && BMODE == SRC_PLAIN;
  using SA = Stage<BM, AKM, AMODE, NT, KB, false, PAIR>;
  using SB = Stage<BN, BKM, BMODE, NT, KB, false, PAIR>;
  using TA = Tile3<BM, AKM, KB, NP>;
  using TB = Tile3<BN, BKM, KB, NP>;
  constexpr int BUF = TA::BYTES + TB::BYTES;
  const float sc_a = NP == 2 ? range_scale(g.range_a) : 1.f, sc_b = BPRE == 2 ? *g.b_h2_scale : NP == 2 ? range_scale(g.range_b) : 1.f;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  // One-dimensional grid of tiles x k-ranges.  Every XCD owns one contiguous run of logical workgroup ids (see
  // gemm_kernel), and a logical id is (k-range, tile) with the tile fastest: the tiles of one k-range of a split-K
  // product -- the 2-3 column tiles of a convolution's weight gradient read the SAME rows of both operands -- run on one
  // XCD at the same time and share its L2 (as grid.z they were dealt round-robin over the XCDs: every line fetched
  // from HBM once per column tile).
  unsigned lid, kz;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned logical = xcd * q + (xcd < r ? xcd : r) + slot;
    kz = logical / g.tiles_all;
    lid = logical - kz * g.tiles_all;
  }
  const unsigned bx = g.nbatch > 1 ? lid / g.nbatch : lid;
  long tile_m, tile_n;
  if (AMODE != SRC_PLAIN || BMODE != SRC_PLAIN || g.grp_n >= (unsigned)g.tiles_n) {
    tile_m = bx / g.tiles_n; tile_n = bx % g.tiles_n;
  } else {  // column groups (launch3): the B panels of one group stay in the XCD's L2 while the tile rows sweep past them
    const unsigned grp = bx / g.grp_sz, r = bx - grp * g.grp_sz, c0 = grp * g.grp_n;
    const unsigned left = (unsigned)g.tiles_n - c0, gw = left < g.grp_n ? left : g.grp_n;
    tile_m = r / gw; tile_n = c0 + r % gw;
  }
  const long m0 = tile_m * BM, n0 = tile_n * BN;
  const long kbeg = (long)kz * g.k_per_split;
  const long kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;
  const int by = g.nbatch > 1 ? (int)(lid % g.nbatch) : 0;
  if (by) {
    const long ao = (long)(by / g.a.brw) * g.a.by_stride + (long)(by % g.a.brw) * g.a.bx_stride;
    const long bo = (long)(by / g.b.brw) * g.b.by_stride + (long)(by % g.b.brw) * g.b.bx_stride;
    g.a.base = static_cast<const float*>(g.a.base) + ao;
    g.b.base = static_cast<const float*>(g.b.base) + bo;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  SA sa;
  SB sb;
  sa.prepare(g.a, m0, g.M, kbeg, true);
  sb.prepare(g.b, n0, g.N, kbeg, true);
  // column sums of a dense k-major A for free (the bias gradient of a weight-gradient product): see gemm_kernel
  constexpr bool CSUM = AKM && AMODE == SRC_PLAIN;
  const bool do_cs = CSUM && g.a_colsum != nullptr && n0 == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto cs_acc = [&](const float* r) {
    if (CSUM && do_cs) {
#pragma unroll
      for (int q = 0; q < SA::NV; ++q) {
        cs[0] += r[4 * q]; cs[1] += r[4 * q + 1]; cs[2] += r[4 * q + 2]; cs[3] += r[4 * q + 3];
      }
    }
  };
  // k-steps; a data-gradient tile of position-grouped rows only visits the steps whose tap reaches the gradient image
  constexpr bool SKIP = AMODE == SRC_DGRAD;
  uint64_t kmask = 0;
  bool use_mask = false;
  if (SKIP && g.a.grp_shift) {
    const uint32_t tpos = (uint32_t)tile_m - fdiv((uint32_t)tile_m, g.a.f_img) * g.a.f_img.d;
    const int py = (int)fdiv(tpos, g.a.f_line), px = (int)tpos - py * (int)g.a.f_line.d;
    const int nsteps = (int)((kend - kbeg + KB - 1) / KB);  // <= 64 (host)
    for (int t = 0; t < nsteps; ++t) {
      const ColInfo ci = col_info<SRC_DGRAD>(g.a, (uint32_t)(kbeg + (long)t * KB));
      if ((unsigned)(py - ci.jh) < (unsigned)g.a.OH && (unsigned)(px - ci.jw) < (unsigned)g.a.OW) kmask |= 1ull << t;
    }
    use_mask = true;
  }
  constexpr bool PERM = AMODE == SRC_CONV && !AKM && KB == 16;  // forward convolutions: host-chosen order of the k-steps
  KStepOrder ord;
  ord.reset();
  auto nextk = [&](long k) -> long {
    if (SKIP && use_mask) {
      if (!kmask) return -1;
      const int t = __builtin_ctzll(kmask);
      kmask &= kmask - 1;
      return kbeg + (long)t * KB;
    }
    if (PERM && g.kp_cb) return ord.next(g) ? ord.k(g) : -1;
    return k + KB < kend ? k + KB : -1;
  };
  long kcur = (SKIP && use_mask) ? nextk(0) : (PERM && g.kp_cb) ? ord.k(g) : kbeg;
  long kend_l = kend;
  if (kcur < 0) { kcur = kbeg; kend_l = kbeg; }  // no tap reaches this pixel: one step on an all-zero tile
  long knext = nextk(kcur);
  if (SRL_GEMM3_DBG & 128) knext = -1;

  // prologue: the first tile into LDS buffer 0, the second into registers (PAIR: both loaded together, set 0 and set 1)
  sa.load(g.a, m0, g.M, kcur, kend_l, true);
  sb.load(g.b, n0, g.N, kcur, kend_l, true);
  if (PAIR) {
    sa.template load<1>(g.a, m0, g.M, knext >= 0 ? knext : kend, kend, true);
    sb.template load<1>(g.b, n0, g.N, knext >= 0 ? knext : kend, kend, true);
  }
  cs_acc(sa.r);
  store3<SA, BM, AKM, KB, NT, NP>(sa.r, lds, sc_a);
  store3<SB, BN, BKM, KB, NT, NP, BPRE ? BPRE : ((SRL_GEMM3_DBG & 256) != 0 ? 1 : 0)>(sb.r, lds + TA::BYTES, sc_b);
  __syncthreads();
  if (!PAIR) {  // the second tile -- or, for a product of a single k-step, an out-of-range tile: zeros (the step's staging is
    // unconditional, and the fused column sums would count the first tile twice if it were still in the registers)
    const long kn = knext >= 0 ? knext : kend;
    sa.load(g.a, m0, g.M, kn, kend, true);
    sb.load(g.b, n0, g.N, kn, kend, true);
  }

  // one k-step on LDS buffer `cur`: per 16-deep k-block, fragments of the three planes of both operands and the six
  // piece products per 32x32 block, with the staging of the following tiles (tile t+1: registers -> the other LDS buffer
  // with the split; tile t+2: global -> registers) INTERLEAVED with the MFMAs, then the single barrier of the step.
  // An MFMA occupies the wavefront's issue for 8 of its 32 cycles: four or five independent VALU operations fit in its
  // shadow.  Left to itself the compiler issues the 24 MFMAs of a step back to back and the ~100 operations of the split
  // after them, with the matrix pipe idle (60 % busy by the counters); the staging is therefore unconditional -- a step
  // without a successor stages an out-of-range tile: zeros, written to a buffer nobody reads -- so that the whole step is
  // one basic block, and sched_group_barrier lays the instruction pipeline out: 1 MFMA, 4 VALU, the LDS writes and the
  // global loads spread between them.
  auto kstep = [&](auto cur_c, long k1, long k2, long k3) {  // k2 (and, PAIR, k3): the tiles whose loads this step issues
    constexpr int cur = decltype(cur_c)::value;
    const uint8_t* la = lds + cur * BUF;
    const uint8_t* lb = la + TA::BYTES;
    uint8_t* nxt = lds + (cur ^ 1) * BUF;
    const long k2x = k2 >= 0 ? k2 : kend;  // out of range: every load returns zeros
    const long k3x = k3 >= 0 ? k3 : kend;
    // PAIR: step A (cur == 0) writes set 1 (tile t+1) to LDS and then loads tiles t+2 -> set 0, t+3 -> set 1;
    //       step B (cur == 1) writes set 0 (tile t+1 of its own count) and loads nothing
    const float* ra = (PAIR && cur == 0) ? sa.r2 : sa.r;
    const float* rb = (PAIR && cur == 0) ? sb.r2 : sb.r;
#pragma unroll
    for (int kb = 0; kb < KB / 16; ++kb) {
      bf16x8 a[TM][NP], b[TN][NP];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) a[i][p] = frag3<BM, AKM, KB, NP>(la, p, wm * (TM * 32) + i * 32, kb, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) b[j][p] = frag3<BN, BKM, KB, NP>(lb, p, wn * (TN * 32) + j * 32, kb, lane);
      // small terms first: they meet an accumulator that has not grown by this block's leading term yet
      constexpr int NPROD = NP == 3 ? 6 : 3;
      constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if (NP == 3) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i][j], 0, 0, 0);
            } else {
              union { bf16x8 b; f16x8 h; } ua, ub;
              ua.b = a[i][PA[t]]; ub.b = b[j][PB[t]];
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc[i][j], 0, 0, 0);
            }
          }
      if (kb == 0) {
        (void)k1;
        cs_acc(ra);
        if (!(SRL_GEMM3_DBG & 2)) {
          store3<SA, BM, AKM, KB, NT, NP>(ra, nxt, sc_a);           // tile t+1 (or zeros): registers -> the other LDS buffer
          store3<SB, BN, BKM, KB, NT, NP, BPRE ? BPRE : ((SRL_GEMM3_DBG & 256) != 0 ? 1 : 0)>(rb, nxt + TA::BYTES, sc_b);
        }
        if (!(SRL_GEMM3_DBG & 1) && (!PAIR || cur == 0)) {
          sa.load(g.a, m0, g.M, k2x, kend, true);         // tile t+2 (or nothing): global -> registers
          sb.load(g.b, n0, g.N, k2x, kend, true);
          if (PAIR) {
            sa.template load<1>(g.a, m0, g.M, k3x, kend, true);
            sb.template load<1>(g.b, n0, g.N, k3x, kend, true);
          }
        }
      }
      if (KB == 16) {  // the issue pipeline of the step: see above
        constexpr int NM = NPROD * TM * TN;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 4 : 5, 0);     // vector operations in its shadow
          if (NP == 2 || m % 2 == 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // LDS writes between them
          if (m % (NP == 3 ? 6 : 3) == (NP == 3 ? 5 : 2)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // global loads
        }
      }
    }
    if (!(SRL_GEMM3_DBG & 8)) __syncthreads();
  };
  if (!PAIR) {
    for (;;) {
      long k2 = knext >= 0 ? nextk(knext) : -1;
      kstep(std::integral_constant<int, 0>{}, knext, k2, -1);
      if (knext < 0) break;
      knext = k2;
      k2 = knext >= 0 ? nextk(knext) : -1;
      kstep(std::integral_constant<int, 1>{}, knext, k2, -1);
      if (knext < 0) break;
      knext = k2;
    }
  } else {
    for (;;) {  // knext: tile t+1, already in set 1
      const long k2 = knext >= 0 ? nextk(knext) : -1, k3 = k2 >= 0 ? nextk(k2) : -1;
      kstep(std::integral_constant<int, 0>{}, knext, k2, k3);  // step A on tile t
      if (knext < 0) break;
      kstep(std::integral_constant<int, 1>{}, k2, -1, -1);     // step B on tile t+1: set 0 (tile t+2 or zeros) -> LDS 0
      if (k2 < 0) break;
      knext = k3;
    }
  }
  if (CSUM && do_cs) {  // workgroup-uniform; the tiles in LDS are dead after the loop's last barrier
    constexpr int G = BM / 4;
    float* lf = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int c = 0; c < 4; ++c) lf[tid * 4 + c] = cs[c];
    __syncthreads();
    if (tid < G) {
      float t4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < NT / G; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) t4[c] += lf[(tid + j * G) * 4 + c];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (m0 + tid * 4 + c < g.M) atomicAdd(g.a_colsum + (long)by * g.a_colsum_batch + m0 + tid * 4 + c, t4[c]);
    }
  }
  if (NP == 2 && (g.range_a || g.range_b || BPRE == 2)) {  // powers of two: exact
    const float inv = 1.f / (sc_a * sc_b);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv;
  }
  constexpr int EPI = AMODE == SRC_DGRAD ? 1 : 0;
  if ((SRL_GEMM3_DBG & 16) && acc[0][0][0] != 12345.f) return;
  bool interior = m0 + BM <= g.M && n0 + BN <= g.N;
  if (g.o.rowmap && g.o.grp_shift)  // position-grouped rows: the tile's BM images must exist as well
    interior = interior && (((fdiv((uint32_t)(m0 >> g.o.grp_shift), g.o.f_img) + 1u) << g.o.grp_shift) <= (uint32_t)g.o.n_img);
  gemm_epilogue<TM, TN, EPI, MK>(g, acc, m0, n0, wm, wn, l31, h, by, kz, interior);
}

inline bool tile_groups() {  // SRL_TILE_GROUP=0: row-major tile numbering everywhere (A/B switch)
  static const bool on = [] { const char* e = getenv("SRL_TILE_GROUP"); return !(e && e[0] == '0'); }();
  return on;
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AMODE, int BMODE, int KB = 32, int NP = 3>
inline int launch3(hipStream_t st, GemmArgs a, int batch, int nsplit) {
  if (!(a.vec_a && a.vec_b)) return -EINVAL;
  const long tiles_m = srl_ceil_div(a.M, BM);
  a.tiles_n = (int)srl_ceil_div(a.N, BN);
  a.nbatch = batch > 1 ? batch : 1;
  const long nblk = tiles_m * a.tiles_n * a.nbatch;
  if (nblk * nsplit > 0x7fffffffL) return -EINVAL;
  a.tiles_all = (unsigned)nblk;
  a.grp_n = (unsigned)a.tiles_n;
  if (AMODE == SRC_PLAIN && BMODE == SRC_PLAIN && a.nbatch == 1 && tile_groups()) {
    // A dense product whose XCD share of tiles (one contiguous run of the numbering) exceeds the XCD's workgroup slots is
    // worked off row after row; in row-major numbering every new tile row streams ALL of B again, from HBM once B is
    // larger than the L2 (FC data gradient, 16384 x 3136 x 512: B = 6.4 MB re-read by each of the 16 tile rows of every
    // XCD, 827 MB fetched per launch for 245 MB of operands).  Column groups whose B panels fit in half the 4 MB L2 are
    // swept by all tile rows instead; A is then re-read once per group.
    const long kr = nsplit > 1 ? a.k_per_split : a.K;
    const double panel_b = (double)BN * (double)kr * 4.0, l2_half = 2.0 * 1024 * 1024;
    const long run = srl_ceil_div(nblk * (long)nsplit, 8), slots = 32L * min_waves3(BM, BN, KB, NP);
    if (run > slots && panel_b * a.tiles_n > l2_half) {
      const long gn = (long)(l2_half / panel_b);
      a.grp_n = (unsigned)(gn < 1 ? 1 : gn);
    } else if (tiles_m < a.tiles_n && tiles_m * tiles_m <= run) {
      // few tile rows (a weight gradient: FC, 512 x 3136 over 5 k-ranges = 4 x 25 tiles each): an XCD's run of row-major
      // ids is a couple of whole tile rows, i.e. ALL of B's panels; numbered down the columns it is every tile row of ~16
      // columns -- 738 -> 347 MB fetched per launch (239 MB of operands), and the tiles that share a panel start together
      a.grp_n = 1;
    }
  }
  a.grp_sz = (unsigned)tiles_m * a.grp_n;
  dim3 grid((unsigned)(nblk * nsplit), 1, 1);
  srl_count_dispatch(NP == 3 ? SRL_DISP_GEMM3 : SRL_DISP_GEMM2H, BM, BN, nsplit, AMODE << 5 | BMODE << 2 | (int)AKM << 1 | (int)BKM);
  constexpr bool CAN_W = !AKM && !BKM && BMODE == SRC_PLAIN;  // sign masks: see launch() in gemm_core.h
  constexpr bool CAN_R = !AKM && BKM && BMODE == SRC_PLAIN && (AMODE == SRC_PLAIN || AMODE == SRC_DGRAD);
  constexpr bool CAN_PRE = NP == 2 && BMODE == SRC_PLAIN && KB == 16;  // pre-split B: the two-piece kernels, dense B
  auto go = [&](auto mk_c, auto pre_c) {
    constexpr int MK = decltype(mk_c)::value;
    constexpr int PRE = decltype(pre_c)::value;
    hipLaunchKernelGGL((gemm3_kernel<BM, BN, WM, WN, AKM, BKM, AMODE, BMODE, KB, NP, MK, PRE>), grid, dim3(256), 0, st, a);
  };
  using std::integral_constant;
  constexpr bool CAN_H2 = NP == 2 && BKM && BMODE == SRC_PLAIN && KB == 16;  // B as h2p rows: the two-piece kernels, k-major dense B
  if (a.b_h2_scale) {
    if constexpr (!CAN_H2) return -ENOTSUP;
    else if (a.mask_out || a.dact_mask || a.b_presplit) return -ENOTSUP;
    else go(integral_constant<int, 0>{}, integral_constant<int, 2>{});
  } else if (a.b_presplit) {
    if constexpr (!CAN_PRE) return -ENOTSUP;
    else if (a.mask_out) {
      if constexpr (CAN_W) go(integral_constant<int, 2>{}, integral_constant<int, 1>{});
      else return -ENOTSUP;
    } else if (a.dact_mask && !a.dact_src) {
      if constexpr (CAN_R) go(integral_constant<int, 1>{}, integral_constant<int, 1>{});
      else return -ENOTSUP;
    } else go(integral_constant<int, 0>{}, integral_constant<int, 1>{});
  } else if (a.mask_out) {
    if constexpr (CAN_W) go(integral_constant<int, 2>{}, integral_constant<int, 0>{});
    else return -ENOTSUP;
  } else if (a.dact_mask && !a.dact_src) {
    if constexpr (CAN_R) go(integral_constant<int, 1>{}, integral_constant<int, 0>{});
    else return -ENOTSUP;
  } else {
    go(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  }
  return 0;
}
#endif  // __HIPCC__

// Products whose caller hands over the range of both operands (srl_gemm_desc::a_absmax / b_absmax, the ranges of the
// srl_conv2d_nhwc_* entry points) run on the two-plane f16 variant; SRL_F16X2=0 keeps them on the three bf16 planes (A/B)
inline bool use_f16x2() {
  const char* e = getenv("SRL_F16X2");
  return !(e && e[0] == '0');
}

// SRL_MFMA=f32 forces the float32 MFMA kernels everywhere (A/B timing, cross-checks in the tests)
inline bool use_bf16x3() {
  const char* e = getenv("SRL_MFMA");
  return !(e && e[0] == 'f');
}

}  // namespace srlgemm
