// FP32 contraction core on the CDNA4 matrix cores, shared by the dense GEMM entry point (gemm.hip) and the
// implicit-GEMM convolution entry points (conv.hip):   C[M,N] = epilogue( sum_k A(i,k) * B(k,j) ).
//
// v_mfma_f32_32x32x2_f32 (exact float32 FMA chains, 64 FLOP/clk/SIMD = the FP32 peak of gfx950): lane l of a
// wavefront supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the 32x32 result sits in 16 accumulator
// registers per lane with col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// A workgroup of 4 (or 8) wavefronts (WM x WN) owns a BM x BN output tile and walks K in steps of KB (16; 32 for the
// byte-staged first-layer kernels).  An operand's *orientation* decides how it is staged and fed:
//   k-major      (rows = reduction index): LDS [KB][BX + 4], float4 along the output index (ds_write_b128), one
//                ds_read_b32 per MFMA step;
//   k-contiguous (rows = output index): LDS [BX][KB] unpadded, 16-byte chunks XOR-swizzled by the row (ds_write_b128
//                along k as the data comes -- no transposing scalar stores), each lane fetches its KB/2 k-values with
//                ds_read_b128; MFMA step kk consumes k = (KB/2) * (lane >> 5) + kk, a pairing both operands share.
//   uint8 frames (SRC_OBSN, flag OBS8) stay bytes in LDS and are widened + normalised after the fragment read.
// What else varies is the operand's *source*:
//   source       SRC_PLAIN    a dense row-major matrix
//                SRC_CONV     rows are convolution patches gathered on the fly from an NHWC activation
//                SRC_DGRAD    rows are the output-gradient taps that reach one input pixel (zero where none)
//                SRC_OBS      rows are patches of the NCHW uint8/float32 observation with the whole-observation
//                             LayerNorm (and optionally its affine) applied in the load path
// so a convolution never materialises its patch matrix (no im2col / col2im traffic).  Row / column indices are
// split with precomputed multiply-shift divisors.
//
// Pipeline (measured with scripts/mfma_peak.hip: a wavefront that leaves the MFMA stream to stage a tile idles
// the matrix pipe, and co-resident workgroups run in phase, so they do not fill the gap for each other): LDS holds
// two tiles; inside k-step t every wavefront issues a quarter of its MFMAs, writes tile t+1 (already in registers) to
// the other LDS buffer, issues the global loads of tile t+2 (t+3 for operands staged two deep), issues the rest of its
// MFMAs, and meets the single barrier of the step.  No wavefront ever waits on "write LDS -> barrier -> read LDS" with
// nothing to do.  A data-gradient tile of position-grouped rows skips the k-steps of taps that leave the image.
#pragma once
#include "srl_common.h"
#include <type_traits>
#include <utility>

namespace srlgemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
enum { SRC_PLAIN = 0, SRC_CONV = 1, SRC_DGRAD = 2, SRC_OBS = 3, SRC_OBSN = 4 };  // OBSN: OBS without the affine
constexpr bool is_obs(int m) { return m == SRC_OBS || m == SRC_OBSN; }

// ---- division by a runtime-invariant 32-bit divisor (valid for dividends < 2^31) ----------------------------------
struct FastDiv {
  uint32_t d, mul, shr, one;  // one: all ones when d == 1 (the 32-bit magic cannot express it), else 0
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{d ? d : 1u, 0u, 0u, 0u};
  f.one = f.d == 1 ? 0xffffffffu : 0u;
  if (f.d != 1) {
    uint32_t lg = 31 - __builtin_clz(f.d);
    if (f.d & (f.d - 1)) lg += 1;  // ceil(log2 d)
    const uint32_t p = 31 + lg;
    f.mul = (uint32_t)(((1ull << p) + f.d - 1) / f.d);
    f.shr = p - 32;
  }
  return f;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  // branch-free on purpose: written as a select on a wave-uniform divisor, the compiler branched around the multiply and
  // cut the k-step of the k-major gathers into a dozen basic blocks (their loads then issued after the MFMAs, not among them)
  return (__umulhi(n, f.mul) >> f.shr) | (n & f.one);
}
#endif

// One operand.  For the gather sources a "row" r splits as r = (n * rows_per_img + y * rows_per_line + x) and a
// "column" c as described per mode; element address = row_off(r) + col_off(c) (32-bit element offsets).
struct SrcDesc {
  const void* base;
  long ld;  // SRC_PLAIN row pitch
  FastDiv f_img, f_line;  // rows per image, rows per image line
  int img_stride, y_stride, x_stride;
  FastDiv f_inner, f_tap;  // column split (CONV: kw*C run; DGRAD: Co then taps-per-line; OBS: KH*KW then KW)
  int k1_stride, k2_stride;  // CONV: W*C per kh.  OBS: H*W per channel, W per kh.
  int OH, OW;                // DGRAD: bounds of the gradient image
  const float *mean, *rstd, *gamma, *beta;  // OBS LayerNorm
  int is_u8, affine;
  // batch (grid.y) offset of the base: (by / brw) * by_stride + (by % brw) * bx_stride, in elements
  int brw, by_stride, bx_stride;
  // DGRAD, position-grouped rows (grp_shift > 0): row r = ((g * P + pos) << grp_shift) + nl stands for pixel pos of
  // image n = (g << grp_shift) + nl (valid while n < n_img), P = f_img.d positions per image.  A tile of 1 << grp_shift
  // rows then holds ONE pixel position of that many images, so the taps that fall outside the gradient image are the
  // same for all its rows and their k-steps are skipped instead of multiplied by zeros.
  int grp_shift, n_img;
};

struct OutDesc {
  float* out;
  long ldo;
  int rowmap;  // 0: out[row*ldo + col];  1: row -> (n,y,x) -> n*img + y*ys + x*xs + col
  FastDiv f_img, f_line;
  long img_stride, y_stride, x_stride;
  long batch_stride;  // batch b: out += (b / brw) * batch_stride + (b % brw) * bx_stride  (brw 0: b * batch_stride)
  int brw;
  long bx_stride;
  // column groups (cg_width > 0): column c = (group, ci), ci = c % cg_width; the group's columns land at
  // + (group / cg_brw) * cg_ystride + (group % cg_brw) * cg_xstride + ci  (parity classes of a strided data gradient
  // packed along N: they share the A operand and differ only in where their rows land)
  int cg_width, cg_brw;
  FastDiv f_cg;
  long cg_ystride, cg_xstride;
  int grp_shift, n_img;  // rowmap with position-grouped rows (see SrcDesc)
};

struct GemmArgs {
  long M, N, K;
  SrcDesc a, b;
  OutDesc o;
  long slab;  // split-K: out + z*slab
  const float* bias;
  long bias_batch;  // per-batch (grid.y) stride of the bias vector
  const float* dact_src;  // same addressing as the output (own pitch ld_dact when !rowmap)
  long ld_dact;
  // The ReLU derivative needs one BIT of the producer's output, not the float: a forward launch with act == 1 and mask_out
  // sets bit e of mask_out (word e >> 5, bit e & 31) iff output element e is > 0, e = the element's offset from the start
  // of the output tensor; a data-gradient launch with dact == 1 reads dact_mask the same way instead of dact_src (mask_off
  // = offset of o.out inside the tensor) -- 1/32 of the bytes.  Both sides move whole words: one 32-column block of a row is
  // one word (N, the pitches and every base offset are multiples of 32: checked by the callers), lane r of a wavefront
  // stores / loads the word of row r of a 32 x 32 block, and the bits travel between that lane and the accumulator layout
  // through v_writelane / ds_bpermute -- one memory instruction per block instead of one per accumulator register (which is
  // what made an earlier version of this SLOWER than reading the floats: the read was bound by issue, not bytes).
  // mask_out: dense outputs only (!rowmap).
  uint32_t* mask_out;
  const uint32_t* dact_mask;
  uint32_t mask_off;
  int b_presplit;  // gemm3_kernel NP == 2: B holds srl_presplit's output (see BPRE)
  const float* b_h2_scale;  // gemm3_kernel NP == 2, k-major dense B: B is h2p rows under this scale (BPRE == 2)
  int act, dact, accumulate;
  long k_per_split;
  int vec_a, vec_b;
  int tiles_n;
  unsigned tiles_all;  // gemm3_kernel: tiles (x batches) per k-range; its grid is tiles_all x k-ranges in one dimension
  // gemm3_kernel, dense operands: tiles are numbered column group by column group (grp_n tile columns wide, every tile row
  // inside a group before the next group; grp_sz = tile rows x grp_n).  grp_n >= tiles_n: plain row-major numbering.
  unsigned grp_n, grp_sz;
  // bf16x3 forward convolutions: the 16-deep k-steps (k = (kh, kw, c), c fastest) are visited in the order of KStepOrder
  // (kp_cb == 0: ascending).  A sum's terms may come in any order; the order decides which input lines are re-read soon.
  int kp_s, kp_kh, kp_kw, kp_cb;  // stride, kernel height / width, 16-channel blocks per pixel
  int nbatch;
  float* a_colsum;  // [M] += sum_k A(i, k) (k-major dense A only): the bias gradient of a weight-gradient product
  long a_colsum_batch;  // per-batch (grid.y) stride of a_colsum
  // gemm3_kernel's two-plane f16 variant (NP == 2): device floats holding (an upper bound of) max |a|, max |b|; the kernel
  // derives the power-of-two scale that puts that maximum into [2^13, 2^14) (range_scale), undone on the accumulators
  const float* range_a;
  const float* range_b;
  // max over the tile's stored outputs of |value| is folded into *out_absmax with an atomic max (the range of the next
  // layer's operand); NULL: not tracked
  float* out_absmax;
};

#ifdef __HIPCC__
// power-of-two scale that brings an operand whose largest magnitude is *amax into [2^13, 2^14): f16's top, with headroom
__device__ __forceinline__ float range_scale(const float* amax) {
  if (!amax) return 1.f;
  // a plain (scalar) load: the float was finished by the previous kernels of this stream.  (An agent-scope atomic load here
  // -- tried while chasing the two-pipeline irreproducibility, which it did not cure -- turns the scale into a vector
  // register value and cost the conv kernels a factor two to three.)
  const float a = *amax;
  const int e = (int)((__float_as_uint(a) >> 23) & 0xffu);  // biased exponent: 2^(e-127) <= amax < 2^(e-126)
  if (e == 0 || e == 255) return 1.f;                           // zero / subnormal / inf / nan: nothing to scale by
  int se = 127 + 13 - (e - 127);
  se = se < 87 ? 87 : (se > 167 ? 167 : se);                    // scales within 2^-40 .. 2^40
  return __uint_as_float((uint32_t)se << 23);
}
// the sign words of a 32 x 32 accumulator block (layout of gemm_epilogue) into the lanes that own the rows: register R holds
// rows cr and cr + 4 (cr = (R & 3) + 8 (R >> 2)) in the two halves of the wavefront, so one ballot is both rows' words.
// (Compare + select per row rather than v_writelane_b32 in inline assembly: the ballot is a VALU-written SGPR, and the
// wait states v_writelane needs after that are only inserted for instructions the compiler knows -- without them the
// lane received the PREVIOUS ballot.)
template <int R>
__device__ __forceinline__ void sign_rows(const float (&v)[16], int lane31, uint32_t& wv) {
  const unsigned long long bal = __ballot(v[R] > 0.f);
  constexpr int cr = (R & 3) + 8 * (R >> 2);
  wv = lane31 == cr ? (uint32_t)bal : wv;
  wv = lane31 == cr + 4 ? (uint32_t)(bal >> 32) : wv;
}
template <int... R>
__device__ __forceinline__ uint32_t sign_words(const float (&v)[16], int lane31, std::integer_sequence<int, R...>) {
  uint32_t wv = 0;
  (sign_rows<R>(v, lane31, wv), ...);
  return wv;
}
__device__ __forceinline__ void absmax_commit(float* dst, float mx) {  // mx >= 0: unsigned order = float order
  // One atomic per wavefront to ONE address is a serial chain through one L2 channel (~10 ns each): 51 000 wavefronts of a
  // strided data gradient spent 0.2 ms of its 0.67 there.  The slot only grows, so a wavefront whose maximum does not exceed
  // what a (relaxed, possibly stale -- staleness only means a superfluous atomic) load returns has nothing to add: after the
  // first few tiles almost every wavefront leaves with the load alone.
  mx = wave_allmax(mx);
  if ((threadIdx.x & 63) == 0 && mx > 0.f) {
    unsigned int* d = reinterpret_cast<unsigned int*>(dst);
    if (__float_as_uint(mx) > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, __float_as_uint(mx));
  }
}

// Odometer over the k-steps of a forward convolution, slowest digit first: 128-byte line of the pixel (two 16-channel
// blocks), parity class of the tap (kh % S, kw % S), the taps of the class, the blocks of the line.  Taps that read the
// same input pixels become neighbours in time (see conv.hip fwd_kstep_order); a few scalar operations per step.
struct KStepOrder {
  int line, cy, cx, qh, qw, c2;
  __device__ __forceinline__ void reset() { line = cy = cx = qh = qw = c2 = 0; }
  __device__ __forceinline__ long k(const GemmArgs& g) const {
    return (long)(((cy + g.kp_s * qh) * g.kp_kw + cx + g.kp_s * qw) * g.kp_cb + 2 * line + c2) * 16;
  }
  __device__ __forceinline__ bool next(const GemmArgs& g) {  // false: past the last step
    if (++c2 < 2 && 2 * line + c2 < g.kp_cb) return true;
    c2 = 0;
    if (cx + g.kp_s * ++qw < g.kp_kw) return true;
    qw = 0;
    if (cy + g.kp_s * ++qh < g.kp_kh) return true;
    qh = 0;
    if (++cx < g.kp_s && cx < g.kp_kw) return true;
    cx = 0;
    if (++cy < g.kp_s && cy < g.kp_kh) return true;
    cy = 0;
    return 2 * ++line < g.kp_cb;
  }
};

// ---- source policies ------------------------------------------------------------------------------------------------
struct RowInfo {
  int off;  // element offset of the row's origin
  int y, x;
  float rs, mr;  // OBS: rstd and mean of the row's sample
  int pos;       // OBS: offset inside the image (index into gamma/beta)
  bool ok;       // DGRAD with grouped rows: the row's image exists
};

template <int MODE>
__device__ __forceinline__ RowInfo row_info(const SrcDesc& s, uint32_t r) {
  RowInfo ri;
  uint32_t n = fdiv(r, s.f_img);
  uint32_t rem = r - n * s.f_img.d;
  ri.ok = true;
  if (MODE == SRC_DGRAD && s.grp_shift) {
    const uint32_t tile = r >> s.grp_shift, nl = r & ((1u << s.grp_shift) - 1u);
    const uint32_t g = fdiv(tile, s.f_img);
    rem = tile - g * s.f_img.d;
    n = (g << s.grp_shift) + nl;
    ri.ok = n < (uint32_t)s.n_img;
    if (!ri.ok) n = 0;
  }
  const uint32_t y = fdiv(rem, s.f_line);
  const uint32_t x = rem - y * s.f_line.d;
  ri.y = (int)y;
  ri.x = (int)x;
  ri.pos = (int)(y * s.y_stride + x * s.x_stride);
  ri.off = (int)n * s.img_stride + ri.pos;
  if (is_obs(MODE)) {  // raw statistics: arithmetic on loaded values is deferred to the LDS-store phase
    ri.rs = s.rstd[n];
    ri.mr = s.mean[n];
  } else {
    ri.rs = 1.f;
    ri.mr = 0.f;
  }
  return ri;
}

struct ColInfo {
  int off;
  int jh, jw;
};

template <int MODE>
__device__ __forceinline__ ColInfo col_info(const SrcDesc& s, uint32_t c) {
  ColInfo ci;
  ci.jh = ci.jw = 0;
  const uint32_t t = fdiv(c, s.f_inner);
  const uint32_t rem = c - t * s.f_inner.d;
  if (MODE == SRC_CONV) {
    ci.off = (int)(t * s.k1_stride + rem);
  } else if (MODE == SRC_DGRAD) {
    const uint32_t jh = fdiv(t, s.f_tap);
    const uint32_t jw = t - jh * s.f_tap.d;
    ci.jh = (int)jh;
    ci.jw = (int)jw;
    ci.off = -(int)((jh * s.OW + jw) * s.f_inner.d) + (int)rem;
  } else {  // SRC_OBS
    const uint32_t kh = fdiv(rem, s.f_tap);
    const uint32_t kw = rem - kh * s.f_tap.d;
    ci.off = (int)(t * s.k1_stride + kh * s.k2_stride + kw);
  }
  return ci;
}

template <int MODE>
__device__ __forceinline__ float4 gather4(const SrcDesc& s, const RowInfo& ri, const ColInfo& ci) {
  if (MODE == SRC_DGRAD) {
    if ((unsigned)(ri.y - ci.jh) >= (unsigned)s.OH || (unsigned)(ri.x - ci.jw) >= (unsigned)s.OW)
      return make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int off = ri.off + ci.off;
  if (!is_obs(MODE)) return *reinterpret_cast<const float4*>(static_cast<const float*>(s.base) + off);
  float4 v;
  if (s.is_u8) {  // raw bytes travel as one dword; decoded in obs_finish()
    v.x = __uint_as_float(*reinterpret_cast<const uint32_t*>(static_cast<const uint8_t*>(s.base) + off));
    v.y = v.z = v.w = 0.f;
  } else {
    v = *reinterpret_cast<const float4*>(static_cast<const float*>(s.base) + off);
  }
  return v;
}

// LayerNorm of the sample applied to one gathered quad: (x - mean) * rstd [* gamma + beta]
__device__ __forceinline__ float4 obs_finish(const SrcDesc& s, float4 raw, float rs, float mean, const float4& g,
                                             const float4& b, bool affine) {
  float4 v = raw;
  if (s.is_u8) {
    const uint32_t w = __float_as_uint(raw.x);
    v = make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
  }
  const float mr = -mean * rs;
  v.x = fmaf(v.x, rs, mr); v.y = fmaf(v.y, rs, mr); v.z = fmaf(v.z, rs, mr); v.w = fmaf(v.w, rs, mr);
  if (affine) {
    v.x = fmaf(v.x, g.x, b.x); v.y = fmaf(v.y, g.y, b.y); v.z = fmaf(v.z, g.z, b.z); v.w = fmaf(v.w, g.w, b.w);
  }
  return v;
}

// ---- buffer loads: 32-bit per-lane byte offsets against a wave-uniform descriptor; an offset >= num_records makes
// the hardware return zeros, which is how every out-of-bounds / padding element is produced (no branches)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNumRecords = 0xFFFFFF00u;
constexpr uint32_t kInvalidOff = 0xFFFFFF00u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)kNumRecords, 0x00020000);
}
#ifndef SRL_GEMM3_DBG
#define SRL_GEMM3_DBG 0  // timing experiments only, see gemm_bf16x3.h
#endif
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
  if ((SRL_GEMM3_DBG & 32) && voff != kInvalidOff) voff &= 0x3ff0u;    // every gather lands in one 16 KB window (L1 hits)
  if ((SRL_GEMM3_DBG & 64) && voff != kInvalidOff) voff &= 0xffff0u;   // ... in one 1 MB window (L2 hits)
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}

// ---- staging of one operand tile: BX output indices x BK reduction indices ------------------------------------------
// Everything that does not change along k is computed once in prepare(): per-lane byte offsets (rows of the tile for a
// k-contiguous operand, columns for a k-major one), validity, LayerNorm statistics.  Per k-tile a dense operand costs
// no vector ALU work at all (the descriptor base advances on the scalar unit); a gathered one costs one column (or
// row) decomposition per lane plus an add per quad.
// U8 (SRC_OBSN over uint8 frames only): the tile stays BYTES in LDS -- [BX][KB + 8] for a k-contiguous operand,
// [KB][BX + 16] (+ the KB rows' LayerNorm statistics) for a k-major one -- and is widened and normalised after the
// fragment read, in registers.  Every element of these operands is consumed by exactly one wavefront, so the arithmetic
// is the same as normalising at the store, but the LDS traffic of the operand drops fourfold: with 32-wide tiles the
// float staging of the frames alone took ~2/3 of the LDS bandwidth of a CU.
template <int BX, bool KMAJOR, int MODE, int NT = 256, int KB = BK, bool U8 = false, bool TWO_ = false>
struct Stage {
  static constexpr int PK8 = KB;       // U8, k-contiguous: row pitch in bytes; 16-byte chunks XOR-swizzled like the float
                                       // tiles (swz8): dword writes and b64 / b128 fragment reads without bank conflicts
  static constexpr int CPR8 = KB / 16 > 0 ? KB / 16 : 1, RPL8 = 256 / KB;
  __device__ static __forceinline__ int swz8(int x, int c) { return c ^ ((x / RPL8) % CPR8); }
  // U8, k-major (BX == 256): dwords D[KB / 4][BX], D[g][col] = the bytes of rows 4g .. 4g+3 of column col.  A thread
  // stages four consecutive k-rows of its four columns, transposes the 4 x 4 bytes in registers and writes ONE
  // ds_write_b128; a lane reads its column's KB / 8 dwords per k-step with ds_read_b32 over consecutive columns.
  static constexpr int ST8 = KMAJOR ? KB * BX / 4 : BX * PK8 / 4;  // U8: floats of tile bytes (k-major: + 2 KB stats)
  // LDS tile: a k-major operand is stored [KB][BX + 4] (ds_write_b128 along the output index); a k-contiguous one
  // is stored as it comes, row by row (ds_write_b128 along k: no transposing scalar stores, which cost 14 % of the
  // MFMA rate in scripts/mfma_peak.hip), in the swizzled layout described below.
  static constexpr int LD = BX + 4;
  // k-contiguous tile: [BX][KB] floats, NO padding; the 16-byte chunk c of row x sits at chunk position
  // c ^ ((x / RPL) % CPR) (CPR chunks per row, RPL rows per 256-byte bank line).  A 16-lane pass of ds_write_b128 covers
  // RPL whole rows = every bank once whatever the permutation; a 16-lane pass of ds_read_b128 reads the same chunk c
  // of 16 consecutive rows, and the rows that share a bank line get distinct positions from the XOR: neither side
  // conflicts (the padded layout [BX][KB + 4] had clean reads and two-way write conflicts: SQ_LDS_BANK_CONFLICT was
  // 17-36 % of the LDS-active cycles of the k-contiguous kernels, zero in the k-major ones).
  static constexpr int PK = KB;
  static constexpr int CPR = KB / 4, RPL = 16 / CPR > 0 ? 16 / CPR : 1;
  __device__ static __forceinline__ int swz(int x, int c) { return c ^ ((x / RPL) % CPR); }
  static constexpr int TILE = U8 ? (KMAJOR ? ST8 + 2 * KB : ST8) : (KMAJOR ? KB * LD : BX * PK);
  static constexpr int QUADS = BX * KB / 4;            // float4 per tile
  static constexpr int NV = (QUADS + NT - 1) / NT;       // float4 per thread
  static constexpr int NF = NV * 4;                    // floats per thread
  static constexpr bool PARTIAL = QUADS < NV * NT;    // narrow tiles: only threads with u < QUADS stage
  static constexpr int KQ = KB / 4;                    // quads per k-contiguous row
  static constexpr bool GATHER = MODE != SRC_PLAIN;
  float r[NF];
  // A second register set lets an operand be staged TWO tiles ahead (load<SET> / store<SET>, SET = tile parity), as the
  // byte tiles below are.  Measured for the k-major patch gathers of the weight gradients (GATHER && KMAJOR && MODE ==
  // SRC_CONV; skipping their loads altogether gains 15 %): 16 more registers, same occupancy, and the step's weight
  // gradients got SLOWER, 6.03 -> 6.22 ms, A/B on one box -- so no float operand uses it.
  static constexpr bool TWO = TWO_;
  float r2[TWO ? NF : 1];
  uint32_t voff[NV];                                 // k-invariant byte offset of quad q (kInvalidOff: always zero)
  int yx[(MODE == SRC_DGRAD) ? NV : 1];              // DGRAD: (y << 16) | x of the row
  int pos[(MODE == SRC_OBS && !KMAJOR) ? NV : 1];       // OBS: offset of the row inside its image
  // SRC_OBS: the LayerNorm is applied when the tile is written to LDS (after the MFMAs of the previous tile), so
  // that nothing between the global loads and the MFMA loop depends on loaded data
  float d_rs[is_obs(MODE) ? NV : 1], d_mean[is_obs(MODE) ? NV : 1];
  int d_gp[MODE == SRC_OBS ? NV : 1];
  float st_rs, st_mr;  // U8, k-major: lane q < NV holds (rstd, mean) of the wavefront's q-th k-row (one vector load each,
                       // in flight like the tile itself; as scalar loads they stalled the wavefront row after row)
  uint32_t vmask;  // OBSN, k-contiguous: bit q = quad q of the current tile is in bounds  // gamma/beta offset of the quad, or -1 (out of bounds: the quad is zero)
  const char* cur;                     // dense operands: base of the first k-tile (wave-uniform)
  long step;                           // dense operands: bytes per k-tile
  long kb0;                            // dense operands: k of that first tile
  uint32_t esz;                        // element size of the source (1 for uint8 observations)

  // x0: first output index of the tile; xn: extent of that dimension; kbeg: first k of this workgroup
  __device__ __forceinline__ void prepare(const SrcDesc& s, long x0, long xn, long kbeg, bool vec) {
    const int tid = threadIdx.x;
    esz = (is_obs(MODE) && s.is_u8) ? 1u : 4u;
    kb0 = kbeg;
    if (!GATHER) {
      if (!vec) return;  // scalar fallback addresses directly
      const long ld = s.ld;
      if (!KMAJOR) { cur = static_cast<const char*>(s.base) + (x0 * ld + kbeg) * 4; step = KB * 4; }
      else { cur = static_cast<const char*>(s.base) + (kbeg * ld + x0) * 4; step = (long)KB * ld * 4; }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * NT;
        if (!KMAJOR) voff[q] = x0 + u / KQ < xn ? (uint32_t)(((u / KQ) * ld + (u % KQ) * 4) * 4) : kInvalidOff;
        else voff[q] = x0 + (u % (BX / 4)) * 4 < xn ? (uint32_t)(((u / (BX / 4)) * ld + (u % (BX / 4)) * 4) * 4) : kInvalidOff;
        if (PARTIAL && u >= QUADS) voff[q] = kInvalidOff;
      }
      return;
    }
    cur = static_cast<const char*>(s.base);
    step = 0;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int u = tid + q * NT;
      if (!KMAJOR) {
        const long x = x0 + u / KQ;
        const RowInfo ri = row_info<MODE>(s, (uint32_t)(x < xn ? x : xn - 1));
        voff[q] = (x < xn && ri.ok) ? (uint32_t)ri.off * esz : kInvalidOff;
        if (MODE == SRC_DGRAD) yx[q] = (ri.y << 16) | ri.x;
        if (is_obs(MODE)) { d_rs[q] = ri.rs; d_mean[q] = ri.mr; }
        if (MODE == SRC_OBS) pos[q] = ri.pos;
      } else {
        const long x = x0 + (u % (BX / 4)) * 4;
        const ColInfo ci = col_info<MODE>(s, (uint32_t)(x < xn ? x : 0));
        voff[q] = x < xn ? (uint32_t)ci.off * esz : kInvalidOff;
      }
      if (PARTIAL && u >= QUADS) voff[q] = kInvalidOff;
    }
  }

  // ---- U8, k-major: two register sets (SET = parity of the tile), so that a tile's loads are issued TWO k-steps before
  // its LDS store.  Every k-step of this operand touches KB samples it has never seen (rows = samples), i.e. cold
  // lines with full memory latency: with one k-step of lead the wavefronts stalled on them (skipping these loads
  // altogether took the first-layer weight gradient from 81 to 107 TFLOP/s); a set costs NV + 2 registers.
  uint32_t w8[(U8 && KMAJOR) ? 2 : 1][(U8 && KMAJOR) ? NV : 1];
  float s8_rs[2], s8_mr[2];
  template <int SET>
  __device__ __forceinline__ void load8(const SrcDesc& s, long k0, long kend) {
    const int tid = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(cur);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const long k = k0 + NV * wave + q;  // NV consecutive k-rows per wavefront (transposed 4 x 4 at the store)
      const bool ok = k < kend && voff[q] != kInvalidOff;
      const RowInfo ri = row_info<SRC_CONV>(s, (uint32_t)(k < kend ? k : kend - 1));  // address only (no statistics)
      w8[SET][q] = __float_as_uint(bload1(rs, ok ? voff[q] + (uint32_t)ri.off : kInvalidOff));
    }
    const int lq = tid & 63;  // statistics of the wavefront's NV rows: lane q fetches row q's
    const long k = k0 + NV * wave + (lq < NV ? lq : 0);
    const bool okk = lq < NV && k < kend;
    const uint32_t n = fdiv((uint32_t)(okk ? k : 0), s.f_img);
    s8_rs[SET] = okk ? s.rstd[n] : 0.f;
    s8_mr[SET] = okk ? s.mean[n] : 0.f;
  }
  template <int SET>
  __device__ __forceinline__ void store8(float* __restrict__ lds) {
    static_assert(!(U8 && KMAJOR) || (BX == 256 && NT == 256 && NV % 4 == 0), "byte k-major staging: 256 columns, 4 wavefronts");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
#pragma unroll
    for (int g4 = 0; g4 < NV / 4; ++g4) {
      // rows 4g .. 4g+3 (g = (NV/4) wave + g4) x columns 4 lane .. 4 lane + 3: transpose the 4 x 4 bytes
      const uint32_t r0 = w8[SET][4 * g4], r1 = w8[SET][4 * g4 + 1], r2 = w8[SET][4 * g4 + 2], r3 = w8[SET][4 * g4 + 3];
      const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);  // r0.b0 r1.b0 r0.b1 r1.b1
      const uint32_t t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);  // r0.b2 r1.b2 r0.b3 r1.b3
      const uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
      const uint32_t t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
      uint4 c;
      c.x = __builtin_amdgcn_perm(t2, t0, 0x05040100u);  // column 0: r0.b0 r1.b0 r2.b0 r3.b0
      c.y = __builtin_amdgcn_perm(t2, t0, 0x07060302u);  // column 1
      c.z = __builtin_amdgcn_perm(t3, t1, 0x05040100u);  // column 2
      c.w = __builtin_amdgcn_perm(t3, t1, 0x07060302u);  // column 3
      *reinterpret_cast<uint4*>(l32 + ((NV / 4) * wave + g4) * BX + 4 * lane) = c;
    }
    if (lane < NV) {  // (rstd, -mean rstd) of this wavefront's k-rows, behind the tile (zeros for padding rows)
      const int k = NV * wave + lane;
      lds[ST8 + 2 * k] = s8_rs[SET];
      lds[ST8 + 2 * k + 1] = -s8_mr[SET] * s8_rs[SET];
    }
  }

  // loads k-tile [k0, k0 + KB) clipped to kend; k0 - kbeg is a multiple of KB (tiles may be skipped)
  template <int SET = 0>
  __device__ __forceinline__ void load(const SrcDesc& s, long x0, long xn, long k0, long kend, bool vec) {
    const int tid = threadIdx.x;
    float* rr = (TWO && SET) ? r2 : r;
    if (GATHER) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(cur);
      if (!KMAJOR) {
        const long k = k0 + (tid % KQ) * 4;  // u % KQ == tid % KQ for every q
        const bool kok = k < kend;
        const ColInfo ci = col_info<MODE>(s, (uint32_t)(kok ? k : 0));
        const uint32_t cb = (uint32_t)ci.off * esz;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          bool ok = kok && voff[q] != kInvalidOff;
          if (MODE == SRC_DGRAD)
            ok = ok && (unsigned)((yx[q] >> 16) - ci.jh) < (unsigned)s.OH && (unsigned)((yx[q] & 0xffff) - ci.jw) < (unsigned)s.OW;
          const uint32_t o = ok ? voff[q] + cb : kInvalidOff;
          if (is_obs(MODE)) {
            if (MODE == SRC_OBS) d_gp[q] = ok ? pos[q] + ci.off : -1;
            if (MODE == SRC_OBSN) vmask = (q == 0 ? 0u : vmask) | ((ok ? 1u : 0u) << q);
            if (s.is_u8) rr[4 * q] = bload1(rs, o);
            else { const float4 v = bload4(rs, o); rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w; }
          } else {
            const float4 v = bload4(rs, o);
            rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
          }
        }
      } else {
        // with 64 quads per k-row (BX == 256) a wavefront stages exactly one row per q: the row index, its
        // decomposition and its LayerNorm statistics are wave-uniform and live on the scalar unit
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int u = tid + q * NT;
          // U8: a wavefront takes NV consecutive k-rows (it transposes 4 x 4 bytes at the store); else rows wave + 4q
          const long k = k0 + (U8 ? NV * wave + q : BX == 256 ? wave + (NT / 64) * q : u / (BX / 4));
          const bool ok = k < kend && voff[q] != kInvalidOff;
          const RowInfo ri = row_info<MODE>(s, (uint32_t)(k < kend ? k : kend - 1));
          // OR, not a select of the sum: the compiler sank the row decomposition under `ok` (an exec-mask branch per load),
          // which cut the k-step into blocks; anything >= kInvalidOff is out of the descriptor's range and reads as zero
          const uint32_t o = (voff[q] + (uint32_t)ri.off * esz) | (ok ? 0u : kInvalidOff);
          if (is_obs(MODE)) {
            if (!U8) { d_rs[q] = ok ? ri.rs : 0.f; d_mean[q] = ok ? ri.mr : 0.f; }
            if (MODE == SRC_OBS) d_gp[q] = !ok ? -1 : ri.pos + (int)(voff[q] / esz);
            if (s.is_u8) rr[4 * q] = bload1(rs, o);
            else { const float4 v = bload4(rs, o); rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w; }
          } else {
            const float4 v = bload4(rs, o);
            rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
          }
        }
        if (U8) {  // statistics of this wavefront's NV k-rows: lane q fetches row q's
          const int lq = tid & 63;
          const long k = k0 + NV * wave + (lq < NV ? lq : 0);
          const bool okk = lq < NV && k < kend;
          const uint32_t n = fdiv((uint32_t)(okk ? k : 0), s.f_img);
          st_rs = okk ? s.rstd[n] : 0.f;
          st_mr = okk ? s.mean[n] : 0.f;
        }
      }
      return;
    }
    if (vec) {
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(cur + ((k0 - kb0) / KB) * step);
      const long kleft = kend - k0;  // > 0
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * NT;
        const bool kok = !KMAJOR ? (tid % KQ) * 4 < kleft : u / (BX / 4) < kleft;
        const float4 v = bload4(rs, kok ? voff[q] : kInvalidOff);
        rr[4 * q] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
      }
      return;
    }
    const float* __restrict__ src = static_cast<const float*>(s.base);
    const long ld = s.ld;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
      const int e = tid + q * NT;
      float v = 0.f;
      if (PARTIAL && e >= BX * KB) {
      } else if (!KMAJOR) {
        const long x = x0 + e / KB, k = k0 + e % KB;
        if (x < xn && k < kend) v = src[x * ld + k];
      } else {
        const long k = k0 + e / BX, x = x0 + e % BX;
        if (x < xn && k < kend) v = src[k * ld + x];
      }
      rr[q] = v;
    }
  }

  template <int SET = 0>
  __device__ __forceinline__ void store(float* __restrict__ lds, bool vec, const SrcDesc& s) {
    const int tid = threadIdx.x;
    float* rr = (TWO && SET) ? r2 : r;
    if (MODE == SRC_OBS) {
      float4 g[NV], b[NV];
      const __amdgpu_buffer_rsrc_t rg = make_rsrc(s.gamma), rb = make_rsrc(s.beta);
#pragma unroll
      for (int q = 0; q < NV; ++q) {  // all table loads first (L2-resident), then the arithmetic
        const uint32_t o = d_gp[q] < 0 ? kInvalidOff : (uint32_t)d_gp[q] * 4u;
        g[q] = bload4(rg, o);
        b[q] = bload4(rb, o);
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d_gp[q] >= 0)
          v = obs_finish(s, make_float4(rr[4 * q], rr[4 * q + 1], rr[4 * q + 2], rr[4 * q + 3]), d_rs[q], d_mean[q], g[q], b[q],
                         true);
        rr[4 * q + 0] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
      }
    } else if (MODE == SRC_OBSN && U8) {
      // raw bytes: dword rr[4q] goes to LDS as it is.  k-major: the thread that stages the first quad of a k-row also
      // leaves that row's (rstd, -mean * rstd) behind the tile (both zero for padding rows, whose bytes are zero too)
      uint32_t* l32 = reinterpret_cast<uint32_t*>(lds);
      if (KMAJOR) {
        static_assert(!KMAJOR || !U8 || (BX == 256 && NT == 256 && NV % 4 == 0), "byte k-major staging: 256 columns, 4 wavefronts");
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int g4 = 0; g4 < NV / 4; ++g4) {
          // rows 4g .. 4g+3 (g = (NV/4) wave + g4) x columns 4 lane .. 4 lane + 3: transpose the 4 x 4 bytes
          const uint32_t r0 = __float_as_uint(rr[16 * g4]), r1 = __float_as_uint(rr[16 * g4 + 4]);
          const uint32_t r2 = __float_as_uint(rr[16 * g4 + 8]), r3 = __float_as_uint(rr[16 * g4 + 12]);
          const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u);  // r0.b0 r1.b0 r0.b1 r1.b1
          const uint32_t t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);  // r0.b2 r1.b2 r0.b3 r1.b3
          const uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u);
          const uint32_t t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
          uint4 c;
          c.x = __builtin_amdgcn_perm(t2, t0, 0x05040100u);  // column 0: r0.b0 r1.b0 r2.b0 r3.b0
          c.y = __builtin_amdgcn_perm(t2, t0, 0x07060302u);  // column 1
          c.z = __builtin_amdgcn_perm(t3, t1, 0x05040100u);  // column 2
          c.w = __builtin_amdgcn_perm(t3, t1, 0x07060302u);  // column 3
          const int g = (NV / 4) * wave + g4;
          *reinterpret_cast<uint4*>(l32 + g * BX + 4 * lane) = c;
        }
        if (lane < NV) {  // (rstd, -mean rstd) of this wavefront's k-rows, behind the tile (zeros for padding rows)
          const int k = NV * wave + lane;
          lds[ST8 + 2 * k] = st_rs;
          lds[ST8 + 2 * k + 1] = -st_mr * st_rs;
        }
        return;
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * NT;
        if (PARTIAL && u >= QUADS) continue;
        const int x = u / KQ, k = (u % KQ) * 4;
        l32[(x * PK8 + 16 * swz8(x, k >> 4) + (k & 15)) >> 2] = __float_as_uint(rr[4 * q]);
      }
      return;
    } else if (MODE == SRC_OBSN) {
      // no tables: padding quads (loaded as zeros) must stay zero, so their statistics were zeroed in load()
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = obs_finish(s, make_float4(rr[4 * q], rr[4 * q + 1], rr[4 * q + 2], rr[4 * q + 3]), d_rs[q], d_mean[q], z, z,
                              false);
        if (!KMAJOR && !((vmask >> q) & 1u)) v = z;
        rr[4 * q + 0] = v.x; rr[4 * q + 1] = v.y; rr[4 * q + 2] = v.z; rr[4 * q + 3] = v.w;
      }
    }
    if (vec || GATHER) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * NT;
        if (PARTIAL && u >= QUADS) continue;
        if (!KMAJOR) {
          const int x = u / KQ, k = (u % KQ) * 4;
          *reinterpret_cast<float4*>(lds + x * PK + 4 * swz(x, k >> 2)) =
              make_float4(rr[4 * q], rr[4 * q + 1], rr[4 * q + 2], rr[4 * q + 3]);
        } else {
          const int k = u / (BX / 4), x = (u % (BX / 4)) * 4;
          *reinterpret_cast<float4*>(lds + k * LD + x) = make_float4(rr[4 * q], rr[4 * q + 1], rr[4 * q + 2], rr[4 * q + 3]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int e = tid + q * NT;
        if (PARTIAL && e >= BX * KB) continue;
        if (!KMAJOR) lds[(e / KB) * PK + 4 * swz(e / KB, (e % KB) >> 2) + (e & 3)] = rr[q];
        else lds[(e / BX) * LD + e % BX] = rr[q];
      }
    }
  }
};

// ---- epilogue shared by the float32 and the bf16x3 kernels: bias, activation, activation-derivative mask of the
// producer, accumulate, split-K slabs, row / column-group maps.  acc[i][j] is the 32x32 block (i, j) of the wavefront's
// tile in the MFMA accumulator layout (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
// EPI == 1 (data gradients of the implicit convolutions): no bias, no activation, no accumulate -- known at compile time,
// so the run-time switches and their branches are gone; `interior` (workgroup-uniform: the whole tile lies inside the
// matrix and, for position-grouped rows, inside the batch) takes a copy of the block loop whose loads and stores are
// unconditional (64 conditional stores per lane otherwise: an exec-mask branch each).
// MK (sign masks, GemmArgs::mask_out / dact_mask): 0 none, 1 the ReLU derivative is read from dact_mask (and the float path
// is compiled out), 2 mask_out is written.  A template parameter, not a run-time switch: the kernels sit at the register
// limit of their occupancy, and code that is never executed still raised the pressure into spills.
template <int TM, int TN, int EPI = 0, int MK = 0>
__device__ __forceinline__ void gemm_epilogue(GemmArgs& g, f32x16 (&acc)[TM][TN], long m0, long n0, int wm, int wn, int l31,
                                              int h, int by, unsigned kz, bool interior = false) {  // kz: k-range (split-K slab)
  // ---- epilogue: per 32x32 accumulator block, all loads batched ahead of the arithmetic and the stores ----------------
  const long obatch = g.o.brw ? (long)(by / g.o.brw) * g.o.batch_stride + (long)(by % g.o.brw) * g.o.bx_stride
                              : (long)by * g.o.batch_stride;
  float* out = g.o.out + (long)kz * g.slab + obatch;
  if (g.o.rowmap && g.dact_src) g.dact_src += obatch;  // the activation shares the output's map
  const uint32_t* mk = MK == 1 ? g.dact_mask : nullptr;  // sign bits of the producer's output instead of its floats
  uint32_t* mko = (EPI == 0 && MK == 2) ? g.mask_out : nullptr;
  if (MK == 1) g.dact_src = nullptr;
  const float* bias = g.bias ? g.bias + (long)by * g.bias_batch : nullptr;
  // Row addressing stays 32-bit: a 64-bit base per 32-row block plus element offsets (dense output), or offsets
  // from the tensor base through the (image, line, pixel) map (callers keep mapped outputs below 2^32 elements).
  const uint32_t ldo = (uint32_t)g.o.ldo, ldd = (uint32_t)g.ld_dact;
  float amx = 0.f;
  // Position-grouped rows (SrcDesc::grp_shift): a tile's rows are consecutive images at ONE pixel position, so the
  // (group, pixel) decomposition belongs to the tile, not to the row -- two divisions per workgroup on the scalar unit
  // instead of three per row on the vector unit (with K = 256 the epilogue of a stride-2 data gradient was half of the
  // tile's vector work)
  uint32_t grp_pix = 0, grp_n0 = 0;
  const bool grouped = g.o.rowmap && g.o.grp_shift;
  if (grouped) {
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)(m0 >> g.o.grp_shift));
    const uint32_t gq = fdiv(tile, g.o.f_img);
    const uint32_t rem = tile - gq * g.o.f_img.d;
    const uint32_t y = fdiv(rem, g.o.f_line), x = rem - y * g.o.f_line.d;
    grp_pix = (uint32_t)__builtin_amdgcn_readfirstlane((int)(y * (uint32_t)g.o.y_stride + x * (uint32_t)g.o.x_stride));
    grp_n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(gq << g.o.grp_shift));
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long row0 = m0 + wm * (TM * 32) + i * 32 + 4 * h;  // accumulator register r holds row0 + (r&3) + 8*(r>>2)
    float* ob = out;
    const float* db = MK == 1 ? nullptr : g.dact_src;
    uint32_t ro[16];
    uint32_t okm = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cr = (r & 3) + 8 * (r >> 2);
      const bool ok = row0 + cr < g.M;
      okm |= (ok ? 1u : 0u) << r;
      if (grouped) {
        const uint32_t n = grp_n0 + ((uint32_t)(row0 + cr) & ((1u << g.o.grp_shift) - 1u));
        const bool in = n < (uint32_t)g.o.n_img;
        if (!in) okm &= ~(1u << r);
        ro[r] = (in ? n : 0u) * (uint32_t)g.o.img_stride + grp_pix;
      } else if (g.o.rowmap) {
        const uint32_t rr = ok ? (uint32_t)(row0 + cr) : 0u;
        const uint32_t n = fdiv(rr, g.o.f_img);
        const uint32_t rem = rr - n * g.o.f_img.d;
        const uint32_t y = fdiv(rem, g.o.f_line);
        const uint32_t x = rem - y * g.o.f_line.d;
        ro[r] = n * (uint32_t)g.o.img_stride + y * (uint32_t)g.o.y_stride + x * (uint32_t)g.o.x_stride;
      } else {
        ro[r] = (uint32_t)cr * ldo;
      }
    }
    if (!g.o.rowmap) {
      ob += row0 * g.o.ldo;
      if (db) db += row0 * g.ld_dact;
    }
    // sign masks: the element offset (= bit number) of ROW l31 of this block row, column 0 of the output -- lane l31 moves
    // that row's words
    const long rowl = row0 - 4 * h + l31;  // row0 - 4 h: first row of the block row
    const bool rowl_ok = rowl < g.M;
    uint32_t e_rowl = 0;
    if (MK == 1) {
      if (grouped) {
        const uint32_t n = grp_n0 + ((uint32_t)rowl & ((1u << g.o.grp_shift) - 1u));
        e_rowl = g.mask_off + (uint32_t)obatch + (n < (uint32_t)g.o.n_img ? n : 0u) * (uint32_t)g.o.img_stride + grp_pix;
      } else if (g.o.rowmap) {
        const uint32_t rr = rowl_ok ? (uint32_t)rowl : 0u;
        const uint32_t n = fdiv(rr, g.o.f_img);
        const uint32_t rem = rr - n * g.o.f_img.d;
        const uint32_t y = fdiv(rem, g.o.f_line);
        e_rowl = g.mask_off + (uint32_t)obatch + n * (uint32_t)g.o.img_stride + y * (uint32_t)g.o.y_stride +
                 (rem - y * g.o.f_line.d) * (uint32_t)g.o.x_stride;
      } else {
        e_rowl = (uint32_t)((rowl_ok ? rowl : 0L) * g.ld_dact);
      }
    } else if (MK == 2) {
      e_rowl = (uint32_t)obatch + (uint32_t)((rowl_ok ? rowl : 0L) * g.o.ldo);
    }
    // column offsets of the TN blocks, then -- before any arithmetic or store of this block row -- the loads of the
    // producer's activation for ALL of them (the derivative mask of a data gradient): their latency is paid once per block
    // row instead of once per 32x32 block (the stores of block j would otherwise sit between the loads of j and j + 1;
    // that read costs 11-17 % of the data gradients when issued block by block)
    uint32_t cbj[TN], ccj[TN], okj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long col = n0 + wn * (TN * 32) + j * 32 + l31;
      const bool cok = col < g.N;
      cbj[j] = cok ? (uint32_t)col : 0u;  // bias index
      ccj[j] = cbj[j];                     // offset of the column inside an output row
      if (g.o.cg_width) {
        const uint32_t grp = fdiv(cbj[j], g.o.f_cg);
        ccj[j] = (uint32_t)((grp / g.o.cg_brw) * g.o.cg_ystride + (grp % g.o.cg_brw) * g.o.cg_xstride) + (cbj[j] - grp * g.o.cg_width);
      }
      okj[j] = cok ? okm : 0u;
    }
    auto blocks = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      float yv[MK == 1 ? 1 : TN][16];
      if (MK != 1 && db) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t o = g.o.rowmap ? ro[r] : (uint32_t)((r & 3) + 8 * (r >> 2)) * ldd;
            yv[MK == 1 ? 0 : j][r] = (FULL || ((okj[j] >> r) & 1u)) ? db[o + ccj[j]] : 1.f;
          }
      }
      uint32_t mw[TN];  // lane l31: the sign word of row l31 of block j (column offsets: ccj[j] - l31 is this block's first)
      if (MK == 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) mw[j] = mk[(e_rowl + ccj[j]) >> 5];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint32_t cb = cbj[j], cc = ccj[j];
        float v[16];
        const float bv = (EPI == 0 && bias) ? bias[cb] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
        if (EPI == 0 && g.act == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
        } else if (EPI == 0 && g.act == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = tanhf(v[r]);
        }
        if (EPI == 0 && MK == 2) {  // sign bits of the ReLU output: a ballot per accumulator register = the words of two rows,
          // dropped into the lanes that own those rows; then one store for the 32 x 32 block
          const uint32_t wv = sign_words(v, l31, std::make_integer_sequence<int, 16>{});
          if (h == 0 && rowl_ok && (FULL || n0 + wn * (TN * 32) + j * 32 < g.N)) mko[(e_rowl + cc) >> 5] = wv;
        }
        if (MK == 1) {  // the ReLU derivative from the sign words: row cr + 4 h of the block sits in lane cr + 4 h
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cr = (r & 3) + 8 * (r >> 2);
            const uint32_t wd = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * cr + 16 * h, (int)mw[j]);
            v[r] = ((wd >> l31) & 1u) ? v[r] : 0.f;
          }
        } else if (db) {
          if (g.dact == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = yv[MK == 1 ? 0 : j][r] > 0.f ? v[r] : 0.f;
          } else if (g.dact == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] *= 1.f - yv[MK == 1 ? 0 : j][r] * yv[MK == 1 ? 0 : j][r];
          }
        }
        if (EPI == 0 && g.accumulate) {
          float ov[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) ov[r] = (FULL || ((okj[j] >> r) & 1u)) ? ob[ro[r] + cc] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += ov[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (FULL || ((okj[j] >> r) & 1u)) ob[ro[r] + cc] = v[r];
        if (g.out_absmax) {  // an upper bound is all that is asked for: elements beyond the matrix edge (computed from
          // zero-filled operands: at most |bias|) are not masked out, which keeps this at one v_max per element
#pragma unroll
          for (int r = 0; r < 16; ++r) amx = fmaxf(amx, fabsf(v[r]));
        }
      }
    };
    if (interior) blocks(std::true_type{});
    else blocks(std::false_type{});
  }
  if (g.out_absmax) absmax_commit(g.out_absmax, amx);
}

// GEN: the dense operands may need the element-wise (unaligned / ragged leading dimension) staging path.  The
// float4-only instantiation (GEN = false) carries no per-element pointers and needs far fewer registers.
// Wavefronts per SIMD the register allocator is asked to fit (second __launch_bounds__ argument on AMD): three
// workgroups per CU for the 64-accumulator tiles, four for the 32-accumulator ones; fewer where the gather state
// of the observation modes (or the element-wise staging of GEN) would otherwise spill.
constexpr int min_waves(int bm, int bn, int amode, int bmode, bool gen, int nwaves = 4) {
  if (nwaves == 8) return 2;  // one 8-wavefront workgroup per CU (LDS-bound): 2 wavefronts per SIMD
  const bool big = bm * bn >= 16384;
  if (gen || amode == SRC_OBS || bmode == SRC_OBS) return 1;
  if (amode == SRC_OBSN || bmode == SRC_OBSN) return big ? 2 : 3;
  return big ? 3 : 4;
}

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AMODE, int BMODE, bool GEN, int KB, bool OBS8, int MK = 0>
__global__ __launch_bounds__(WM * WN * 64, min_waves(BM, BN, AMODE, BMODE, GEN, WM * WN)) void gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static_assert((WM * WN == 4 || WM * WN == 8) && TM >= 1 && TN >= 1, "4 or 8 wavefronts per workgroup");
  constexpr int NT = WM * WN * 64;
  constexpr bool A8 = OBS8 && AMODE == SRC_OBSN && !AKM;  // uint8 frames as the k-contiguous A (first-layer forward)
  constexpr bool B8 = OBS8 && BMODE == SRC_OBSN && BKM;   // ... as the k-major B (first-layer weight gradient)
  static_assert(!OBS8 || A8 || B8, "OBS8 without a byte-staged operand");
  // two-deep staging of A (Stage TWO_): measured on the forward convolutions' patch rows (AMODE == SRC_CONV && !AKM),
  // 5.19 -> 5.27 ms per step -- off, like the k-major patch gathers; only the byte tiles of the frames profit
  constexpr bool A2 = false;
  using SA = Stage<BM, AKM, AMODE, NT, KB, A8, A2>;
  using SB = Stage<BN, BKM, BMODE, NT, KB, B8>;
  constexpr int A_FLOATS = SA::TILE, TILE_FLOATS = SA::TILE + SB::TILE;  // multiples of 4: 16-byte aligned
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  // grid.x carries (tile, batch) with the batch index fastest when nbatch > 1 (workgroups that run together then
  // share the same rows of the gathered tensor through L2); grid.y is unused in that case
  // Workgroup ids are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  Renumber them so that every
  // XCD owns one contiguous run of logical ids: neighbours in (tile, batch) order -- which read the same rows of the
  // gathered tensor or the same operand panel -- then share an L2 instead of each fetching their own copy.
  unsigned lid;
  {
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    lid = xcd * q + (xcd < r ? xcd : r) + slot;
  }
  const unsigned bx = g.nbatch > 1 ? lid / g.nbatch : lid;
  const long tile_m = bx / g.tiles_n, tile_n = bx % g.tiles_n;
  const long m0 = tile_m * BM, n0 = tile_n * BN;
  const long kbeg = (long)blockIdx.z * g.k_per_split;
  const long kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

  // batch (grid.y): shift the operand bases
  const int by = g.nbatch > 1 ? (int)(lid % g.nbatch) : 0;
  if (by) {
    const long ao = (long)(by / g.a.brw) * g.a.by_stride + (long)(by % g.a.brw) * g.a.bx_stride;
    const long bo = (long)(by / g.b.brw) * g.b.by_stride + (long)(by % g.b.brw) * g.b.bx_stride;
    g.a.base = (is_obs(AMODE) && g.a.is_u8) ? (const void*)(static_cast<const uint8_t*>(g.a.base) + ao)
                                               : (const void*)(static_cast<const float*>(g.a.base) + ao);
    g.b.base = (is_obs(BMODE) && g.b.is_u8) ? (const void*)(static_cast<const uint8_t*>(g.b.base) + bo)
                                               : (const void*)(static_cast<const float*>(g.b.base) + bo);
    if (BMODE == SRC_OBS && g.b.affine) { g.b.gamma += bo; g.b.beta += bo; }
    if (AMODE == SRC_OBS && g.a.affine) { g.a.gamma += ao; g.a.beta += ao; }
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
  const bool va = GEN ? g.vec_a != 0 : true, vb = GEN ? g.vec_b != 0 : true;
  sa.prepare(g.a, m0, g.M, kbeg, va);
  sb.prepare(g.b, n0, g.N, kbeg, vb);
  // A8: (rstd, -mean * rstd) of the rows whose fragments this lane feeds to the MFMAs -- fixed for the whole tile
  float a8_rs[A8 ? TM : 1], a8_mr[A8 ? TM : 1];
  if (A8) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const long row = m0 + wm * (TM * 32) + i * 32 + l31;
      const RowInfo ri = row_info<SRC_OBSN>(g.a, (uint32_t)(row < g.M ? row : g.M - 1));
      a8_rs[i] = ri.rs;
      a8_mr[i] = -ri.mr * ri.rs;
    }
  }
  // Column sums of a dense k-major A, for free: a thread stages the same four output indices of every k-tile
  // (NT is a multiple of BM/4), so it keeps four running sums of what it writes to LDS; the workgroups of the first
  // column of tiles combine theirs at the end.  (A weight-gradient product dZ^T X thereby also yields the bias
  // gradient sum_rows dZ, which used to be a separate pass over dZ.)
  constexpr bool CSUM = AKM && AMODE == SRC_PLAIN && !GEN;
  const bool do_cs = CSUM && g.a_colsum != nullptr && n0 == 0;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  auto cs_acc = [&]() {
    if (CSUM && do_cs) {
#pragma unroll
      for (int q = 0; q < SA::NV; ++q) {
        cs[0] += sa.r[4 * q]; cs[1] += sa.r[4 * q + 1]; cs[2] += sa.r[4 * q + 2]; cs[3] += sa.r[4 * q + 3];
      }
    }
  };
  // k-steps.  Normally kbeg, kbeg + KB, ... < kend.  A data-gradient tile of position-grouped rows (SrcDesc::grp_shift)
  // only visits the k-steps whose tap reaches the gradient image from the tile's pixel: bit t of kmask = step
  // kbeg + t * KB is to be done (everything here is workgroup-uniform and lives on the scalar unit).
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
  auto nextk = [&](long k) -> long {  // the k-step after k, or -1
    if (SKIP && use_mask) {
      if (!kmask) return -1;
      const int t = __builtin_ctzll(kmask);
      kmask &= kmask - 1;
      return kbeg + (long)t * KB;
    }
    return k + KB < kend ? k + KB : -1;
  };
  long kcur = (SKIP && use_mask) ? nextk(0) : kbeg;
  long kend_l = kend;
  if (kcur < 0) { kcur = kbeg; kend_l = kbeg; }  // no tap reaches this pixel: one step on an all-zero tile
  long knext = nextk(kcur);
  // B2: the B operand is staged TWO tiles ahead, in two register sets chosen by tile parity -- the k-major gathers
  // whose every k-step touches rows it has never seen (frames of new samples, patches of new pixels): cold lines with
  // full memory latency, on which the wavefronts stalled with one k-step of lead (skipping those loads altogether
  // gains 24 % in the first-layer weight gradient and 15 % in the others)
  constexpr bool B2 = B8 || SB::TWO;
  auto loadB = [&](auto set_c, long k, long kendx) {
    constexpr int S = decltype(set_c)::value;
    if constexpr (B8) sb.template load8<S>(g.b, k, kendx);
    else sb.template load<S>(g.b, n0, g.N, k, kendx, vb);
  };
  auto storeB = [&](auto set_c, float* dst) {
    constexpr int S = decltype(set_c)::value;
    if constexpr (B8) sb.template store8<S>(dst);
    else sb.template store<S>(dst, vb, g.b);
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  auto loadA = [&](auto set_c, long k, long kendx) { sa.template load<decltype(set_c)::value>(g.a, m0, g.M, k, kendx, va); };
  auto storeA = [&](auto set_c, float* dst) { sa.template store<decltype(set_c)::value>(dst, va, g.a); };
  long knext2 = (B2 || A2) && knext >= 0 ? nextk(knext) : -1;
  // prologue: the first tile into LDS buffer 0, the second (two-deep operands: and third) into registers
  loadA(Set0{}, kcur, kend_l);
  loadB(Set0{}, kcur, kend_l);
  cs_acc();
  storeA(Set0{}, lds);
  storeB(Set0{}, lds + A_FLOATS);
  __syncthreads();
  if (knext >= 0) {
    if (A2) loadA(Set1{}, knext, kend);
    else loadA(Set0{}, knext, kend);
    if (B2) loadB(Set1{}, knext, kend);
    else loadB(Set0{}, knext, kend);
  }
  if (B2 && knext2 >= 0) loadB(Set0{}, knext2, kend);
  if (A2 && knext2 >= 0) loadA(Set0{}, knext2, kend);

  // one k-step on LDS buffer `cur` (compile-time): MFMAs, with the staging of the following tiles in the middle
  // k1 / k2 / k3: the k of the next steps (-1: none); tile t+1 sits in registers (B8: t+1 and t+2, sets by tile parity)
  auto kstep = [&](auto cur_c, long k1, long k2, long k3) {
    constexpr int cur = decltype(cur_c)::value;
    // MFMA step kk consumes k = (KB/2)*h + kk of the tile (h = lane >> 5): any pairing of the KB k values works as long
    // as both operands use the same one, and this one lets a k-contiguous operand fetch its 8 values with two
    // ds_read_b128.  A k-major operand reads row 8*h + kk, one step ahead of its MFMAs.
    const float* ap = lds + cur * TILE_FLOATS +
                      (AKM ? (KB / 2) * h * SA::LD + wm * (TM * 32) + l31 : (wm * (TM * 32) + l31) * SA::PK);
    const float* bp = lds + cur * TILE_FLOATS + A_FLOATS +
                      (BKM ? (KB / 2) * h * SB::LD + wn * (TN * 32) + l31 : (wn * (TN * 32) + l31) * SB::PK);
    float* nxt = lds + (cur ^ 1) * TILE_FLOATS;
    float a[AKM ? 2 : KB / 2][TM], b[BKM ? 2 : KB / 2][TN];
    // byte-staged operands (see Stage): widen + normalise here, after the fragment read
    const uint8_t* ap8 = reinterpret_cast<const uint8_t*>(lds + cur * TILE_FLOATS) + (wm * (TM * 32) + l31) * SA::PK8;
    // B8: the lane's columns' dwords for this half-wave's k-rows (KB / 8 per column; byte e of dword g = row 4g + e)
    uint32_t bw[B8 ? TN : 1][B8 ? KB / 8 : 1];
    if (B8) {
      const uint32_t* dp = reinterpret_cast<const uint32_t*>(lds + cur * TILE_FLOATS + A_FLOATS) + (KB / 8) * h * BN +
                           wn * (TN * 32) + l31;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g2 = 0; g2 < KB / 8; ++g2) bw[j][g2] = dp[g2 * BN + j * 32];
    }
    // (rstd, -mean rstd) of this half-wave's KB/2 k-rows: fetched once per k-step (KB/4 ds_read_b128)
    float bst[B8 ? KB : 1];
    if (B8) {
      const float4* sp = reinterpret_cast<const float4*>(lds + cur * TILE_FLOATS + A_FLOATS + SB::ST8 + KB * h);
#pragma unroll
      for (int c4 = 0; c4 < KB / 4; ++c4) {
        const float4 q = sp[c4];
        bst[4 * c4] = q.x, bst[4 * c4 + 1] = q.y, bst[4 * c4 + 2] = q.z, bst[4 * c4 + 3] = q.w;
      }
    }
    auto read_b8 = [&](int kk, int j) {
      const float raw = (float)((bw[j][kk >> 2] >> (8 * (kk & 3))) & 255u);
      return fmaf(raw, bst[2 * kk], bst[2 * kk + 1]);
    };
    if (AKM) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[0][i] = ap[i * 32];
    } else if (A8) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // the lane's KB/2 bytes: k = (KB/2) h + 0 .. KB/2 - 1, i.e. 8-byte pieces p = (KB/16) h + c8 of the row
        const int row = wm * (TM * 32) + i * 32 + l31;
#pragma unroll
        for (int c8 = 0; c8 < KB / 16; ++c8) {
          const int piece = (KB / 16) * h + c8;  // 8-byte piece index within the row
          const uint2 w = *reinterpret_cast<const uint2*>(ap8 + i * 32 * SA::PK8 + 16 * SA::swz8(row, piece >> 1) +
                                                          8 * (piece & 1));
          const uint32_t ww[2] = {w.x, w.y};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            a[8 * c8 + e][i] = fmaf((float)((ww[e >> 2] >> (8 * (e & 3))) & 255u), a8_rs[i], a8_mr[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int c4 = 0; c4 < KB / 8; ++c4) {
          const float4 q = *reinterpret_cast<const float4*>(
              ap + i * 32 * SA::PK + 4 * SA::swz(wm * (TM * 32) + i * 32 + l31, (KB / 8) * h + c4));
          a[4 * c4][i] = q.x, a[4 * c4 + 1][i] = q.y, a[4 * c4 + 2][i] = q.z, a[4 * c4 + 3][i] = q.w;
        }
      }
    }
    if (B8) {
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = read_b8(0, j);
    } else if (BKM) {
#pragma unroll
      for (int j = 0; j < TN; ++j) b[0][j] = bp[j * 32];
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int c4 = 0; c4 < KB / 8; ++c4) {
          const float4 q = *reinterpret_cast<const float4*>(
              bp + j * 32 * SB::PK + 4 * SB::swz(wn * (TN * 32) + j * 32 + l31, (KB / 8) * h + c4));
          b[4 * c4][j] = q.x, b[4 * c4 + 1][j] = q.y, b[4 * c4 + 2][j] = q.z, b[4 * c4 + 3][j] = q.w;
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB / 2; ++kk) {
      if (kk + 1 < KB / 2) {
        if (AKM) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[(kk + 1) & 1][i] = ap[(kk + 1) * SA::LD + i * 32];
        }
        if (B8) {
#pragma unroll
          for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = read_b8(kk + 1, j);
        } else if (BKM) {
#pragma unroll
          for (int j = 0; j < TN; ++j) b[(kk + 1) & 1][j] = bp[(kk + 1) * SB::LD + j * 32];
        }
      }
      if (kk == KB / 4) {
        if (k1 >= 0) {  // tile t+1: registers -> the other LDS buffer (its readers left at the last barrier)
          cs_acc();
          storeA(std::integral_constant<int, A2 ? (cur ^ 1) : 0>{}, nxt);
          storeB(std::integral_constant<int, B2 ? (cur ^ 1) : 0>{}, nxt + A_FLOATS);
        }
        if (k2 >= 0) {  // tile t+2: global -> registers
          if (!A2) loadA(Set0{}, k2, kend);
          if (!B2) loadB(Set0{}, k2, kend);
        }
        if (B2 && k3 >= 0) loadB(std::integral_constant<int, cur ^ 1>{}, k3, kend);  // tile t+3 into the set t+1 left
        if (A2 && k3 >= 0) loadA(std::integral_constant<int, cur ^ 1>{}, k3, kend);
      }
      // (no sched_barrier: order-pinning was measured)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[AKM ? (kk & 1) : kk][i], b[BKM ? (kk & 1) : kk][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // tile t+1 is visible; everyone is done reading tile t
  };
  for (;;) {  // k1 = knext; B2 keeps k2 = knext2 as well (its loads were issued a step earlier)
    long k2 = (B2 || A2) ? knext2 : (knext >= 0 ? nextk(knext) : -1);
    long k3 = (B2 || A2) && k2 >= 0 ? nextk(k2) : -1;
    kstep(std::integral_constant<int, 0>{}, knext, k2, k3);
    if (knext < 0) break;
    knext = k2;
    knext2 = k3;
    k2 = (B2 || A2) ? knext2 : (knext >= 0 ? nextk(knext) : -1);
    k3 = (B2 || A2) && k2 >= 0 ? nextk(k2) : -1;
    kstep(std::integral_constant<int, 1>{}, knext, k2, k3);
    if (knext < 0) break;
    knext = k2;
    knext2 = k3;
  }
  if (CSUM && do_cs) {  // workgroup-uniform; the tiles in LDS are dead after the loop's last barrier
    constexpr int G = BM / 4;  // threads that share a k-row of the A tile; thread t owns columns 4 * (t % G) ..+3
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[tid * 4 + c] = cs[c];
    __syncthreads();
    if (tid < G) {
      float t4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < NT / G; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) t4[c] += lds[(tid + j * G) * 4 + c];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (m0 + tid * 4 + c < g.M) atomicAdd(g.a_colsum + (long)by * g.a_colsum_batch + m0 + tid * 4 + c, t4[c]);
    }
  }

  gemm_epilogue<TM, TN, 0, MK>(g, acc, m0, n0, wm, wn, l31, h, by, blockIdx.z);
}

static __global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* ws, int nslab, long batch, long M, long N,
                                                           float* C, long ldc, long c_batch, int accumulate) {
  const long per = M * N, total = batch * per;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += ws[(long)z * total + e];
    const long b = e / per, w = e % per;
    float* dst = C + b * c_batch + (w / N) * ldc + (w % N);
    *dst = accumulate ? *dst + s : s;
  }
}
// The common case: the output is one dense block (ldc == N, batches back to back), so slab element e lands at C[e] and
// the reduction is a pure stream -- no index arithmetic (the general kernel spends its time in 64-bit divisions).
static __global__ __launch_bounds__(256) void reduce_slabs_dense_kernel(const float4* ws, int nslab, long total4, float4* C,
                                                                 int accumulate) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
    float4 s = ws[e];
    for (int z = 1; z < nslab; ++z) {
      const float4 v = ws[(long)z * total4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
      const float4 o = C[e];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    C[e] = s;
  }
}

// Few outputs under many slabs (a convolution's weight gradient: 64 x 512 outputs, ~100 slabs): one thread per output
// would walk its slabs as a serial chain of dependent-latency loads on a handful of workgroups.  ZL threads share an
// output, each taking every ZL-th slab with several loads in flight, and combine through LDS.
template <int ZL>
static __global__ __launch_bounds__(256) void reduce_slabs_dense_z_kernel(const float4* ws, int nslab, long total4, float4* C,
                                                                   int accumulate) {
  constexpr int EL = 256 / ZL;
  __shared__ float4 red[256];
  const int ex = threadIdx.x % EL, zy = threadIdx.x / EL;
  const long e = (long)blockIdx.x * EL + ex;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < total4) {
#pragma unroll 4
    for (int z = zy; z < nslab; z += ZL) {
      const float4 v = ws[(long)z * total4 + e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (zy == 0 && e < total4) {
    for (int g2 = 1; g2 < ZL; ++g2) {
      const float4 v = red[g2 * EL + ex];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (accumulate) {
      const float4 o = C[e];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    C[e] = s;
  }
}

// sum of the split-K slabs into C[batch][M][N] (pitch ldc, batch stride c_batch)
inline void reduce_slabs(hipStream_t st, const float* ws, int nslab, long batch, long M, long N, float* C, long ldc,
                         long c_batch, int accumulate) {
  const long total = batch * M * N;
  const bool dense = ldc == N && (batch == 1 || c_batch == M * N) && total % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
  if (dense) {
    const long total4 = total / 4;
    if (nslab >= 16 && total4 <= 256 * 1024) {
      hipLaunchKernelGGL(reduce_slabs_dense_z_kernel<16>, dim3((unsigned)srl_ceil_div(total4, 16)), dim3(256), 0, st,
                         reinterpret_cast<const float4*>(ws), nslab, total4, reinterpret_cast<float4*>(C), accumulate);
      return;
    }
    if (nslab >= 4 && total4 <= 1024 * 1024) {
      hipLaunchKernelGGL(reduce_slabs_dense_z_kernel<4>, dim3((unsigned)srl_ceil_div(total4, 64)), dim3(256), 0, st,
                         reinterpret_cast<const float4*>(ws), nslab, total4, reinterpret_cast<float4*>(C), accumulate);
      return;
    }
    const unsigned grid = (unsigned)(srl_ceil_div(total4, 256) < 4096 ? srl_ceil_div(total4, 256) : 4096);
    hipLaunchKernelGGL(reduce_slabs_dense_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(ws), nslab,
                       total4, reinterpret_cast<float4*>(C), accumulate);
    return;
  }
  const unsigned grid = (unsigned)(srl_ceil_div(total, 256) < 8192 ? srl_ceil_div(total, 256) : 8192);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid), dim3(256), 0, st, ws, nslab, batch, M, N, C, ldc, c_batch, accumulate);
}
#endif  // __HIPCC__

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline SrcDesc plain_src(const float* base, long ld) {
  SrcDesc s{};
  s.base = base;
  s.ld = ld;
  s.f_img = s.f_line = s.f_inner = s.f_tap = make_fastdiv(1);
  s.brw = 1;
  return s;
}

// k range per split: a multiple of BK so that float4 loads never straddle a split boundary
inline int plan_split(long K, int want, long* k_per_split) {
  if (want < 1) want = 1;
  long kps = srl_ceil_div(srl_ceil_div(K, want), 2 * BK) * 2 * BK;  // multiple of every k-step depth in use
  if (kps == 0) kps = 2 * BK;
  *k_per_split = kps;
  const long n = srl_ceil_div(K, kps);
  return (int)(n > 0 ? n : 1);
}

#ifdef __HIPCC__
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int AMODE, int BMODE, bool GEN = false, int KB = BK,
          bool OBS8 = false>
inline int launch(hipStream_t st, GemmArgs a, int batch, int nsplit) {
  if (!GEN && !(a.vec_a && a.vec_b)) return -EINVAL;  // float4-only instantiation
  const long tiles_m = srl_ceil_div(a.M, BM);
  a.tiles_n = (int)srl_ceil_div(a.N, BN);
  a.nbatch = batch > 1 ? batch : 1;
  const long nblk = tiles_m * a.tiles_n * a.nbatch;
  if (nblk > 0x7fffffffL || nsplit > 65535) return -EINVAL;
  dim3 grid((unsigned)nblk, 1, (unsigned)nsplit);
  srl_count_dispatch(SRL_DISP_GEMM_F32, BM, BN, nsplit, AMODE << 5 | BMODE << 2 | (int)AKM << 1 | (int)BKM);
  // sign masks (gemm_epilogue's MK): written by products in the forward orientation, read by those in the data-gradient one
  constexpr bool CAN_W = !AKM && !BKM && !GEN && BMODE == SRC_PLAIN;
  constexpr bool CAN_R = !AKM && BKM && !GEN && BMODE == SRC_PLAIN && (AMODE == SRC_PLAIN || AMODE == SRC_DGRAD);
  if (a.mask_out) {
    if constexpr (CAN_W)
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKM, BKM, AMODE, BMODE, GEN, KB, OBS8, 2>), grid, dim3(WM * WN * 64), 0, st, a);
    else return -ENOTSUP;
  } else if (a.dact_mask && !a.dact_src) {
    if constexpr (CAN_R)
      hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKM, BKM, AMODE, BMODE, GEN, KB, OBS8, 1>), grid, dim3(WM * WN * 64), 0, st, a);
    else return -ENOTSUP;
  } else {
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKM, BKM, AMODE, BMODE, GEN, KB, OBS8>), grid, dim3(WM * WN * 64), 0, st, a);
  }
  return 0;
}
#endif

}  // namespace srlgemm
