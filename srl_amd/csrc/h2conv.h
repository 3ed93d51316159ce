// Image-stationary convolutions over pre-split ("h2") activations (round 4).
//
// What round 3's implicit GEMMs and this round's first attempt (h2gemm.h's CONV mode: measured level with them, 476 / 315 us
// for conv2 / conv3 forward) have in common: every tap of a convolution re-stages its patch rows into LDS, so the bytes
// pushed into LDS are KH*KW/stride^2 times the input (4x conv2, 9x conv3) and a CU takes in only ~30 GB/s through that
// path whatever feeds it (the same 1.45 us per 40-48 KB k-step with 3 or 4 stages in flight, with 12 or 24 MFMAs per
// wavefront in it).  Here an image enters LDS ONCE (linear DMA, 1 KB per instruction, no address arithmetic) and every tap
// reads its operand from there; the weights never enter LDS at all:
//   * a wavefront keeps the weights of ITS 16 output channels in registers for the whole launch (16 x K f16 pairs: 128-144
//     registers; persistent workgroups, one per CU, walk batches of images);
//   * the positions operand of v_mfma_f32_16x16x32_f16 (16 positions x 32 channels) is two ds_read_b128 per piece product
//     triple, at an address that is ONE lane register plus an immediate: images are stored PLANAR -- per 32-channel block,
//     per group g of 8 channels (h2gemm.h's grouping) and per piece, an array of 16-byte entries indexed by pixel -- so a tap
//     is a constant entry offset, 16 consecutive entries are 16 distinct bank groups whatever the tap (the 16-lane banking
//     groups of ds_read_b128 mix two g: plane pitches are multiples of 256 B), and blocks of 16 output positions are cut
//     along the INPUT grid's numbering (a block never wraps a row: positions beyond the output's width are idle lanes)
//     which keeps every read conflict-free (brute-forced over all bases and taps: 4 LDS cycles per read);
//   * a stride-2 layer's input is stored parity-class-major by its producer (class (y & 1, x & 1), then y >> 1, x >> 1):
//     each tap then walks consecutive entries of one class, like a stride-1 layer;
//   * the data gradients are the same kernel on a zero-bordered grid (the border is zeroed once, DMA lanes that would land
//     on it are masked off) with the regrouped weights: conv3's as a 3x3 stride-1 problem on an 11x11 grid, conv2's as four
//     2x2 stride-1 problems (one per parity class of the input pixel) sharing their positions operand, 128 "channels".
//   No barrier inside a batch; one per batch (the next batch's DMA was issued a whole batch earlier).
#pragma once
#include "h2gemm.h"

namespace srlh2 {

typedef float h2_f32x4 __attribute__((ext_vector_type(4)));

enum { H2C_F2 = 0, H2C_F3 = 1, H2C_D3 = 2, H2C_D2 = 3 };
enum { H2S_LINEAR = 0, H2S_PLANAR = 1, H2S_ROWS = 2, H2S_ROWSWZ = 3 };  // where an LDS entry's 16 bytes come from
enum { H2D_PLANAR = 0, H2D_ROWS = 1, H2D_F32_ROUTED = 2 };    // where a result goes

// Geometry of one instantiation.  Output positions are enumerated as entries e = oy * GW + ox of the LDS grid (ox < OW,
// oy < OH valid); the LDS entry of tap t for output entry e is e + tap_u(t); W's k-block of (tap t, channel block c) is
// tap_k(t) * CB + c.
template <int ID> struct H2Geo;

// conv2 forward: 20x20x32 -> 9x9x64, 4x4 stride 2.  The input comes from the first layer, whose workgroups own ONE output
// position over many samples: it can only write whole pixels (128-byte rows of h2p), not per-image planes.  So this layer's
// image is h2p ROWS in parity-class-major pixel order, copied to LDS as it lies (pixel-major, linear DMA) with the 16-byte
// slots of pixel P XOR-ed by sigma(P) = bit 1 of P -> slot bit 0, bit 2 of P -> slot bit 2 (one of the 192 bit-linear maps
// under which 16 consecutive pixels read by the mixed-group lanes of a ds_read_b128 fall into 16 different bank groups); a
// tap's address is then (lane base + tap offset) ^ a byte of a per-lane table: two vector operations per tap.
template <> struct H2Geo<H2C_F2> {
  static constexpr int CGW = 1;   // channel groups of 16 per wavefront (they share the positions' fragments)
  static constexpr int CB = 1, NCH = 64, NTAP = 16, GW = 10, OH = 9, OW = 9, NPIX_L = 400, G = 1, PLANE_E = 400;
  static constexpr int SRC = H2S_ROWSWZ, PAD = 0, SH = 20, SW = 20, SPIX = 400;
  static constexpr int DST = H2D_PLANAR, OPIX = 81, OCH = 64;
  static constexpr bool ZERO_BORDER = false;
  static constexpr bool DENSE = false;
  __host__ __device__ static constexpr int tap_u(int t) { return (((t >> 2) & 1) * 2 + (t & 1)) * 100 + (t >> 3) * 10 + ((t & 3) >> 1); }
  __host__ __device__ static constexpr int tap_k(int t) { return t; }
};
// conv3 forward: 9x9x64 (planar, 81 entries) -> 7x7x64, 3x3 stride 1; result as h2p rows for the Linear behind it
template <> struct H2Geo<H2C_F3> {
  static constexpr int CGW = 1;   // channel groups of 16 per wavefront (they share the positions' fragments)
  static constexpr int CB = 2, NCH = 64, NTAP = 9, GW = 9, OH = 7, OW = 7, NPIX_L = 81, G = 2, PLANE_E = 176;
  static constexpr int SRC = H2S_PLANAR, PAD = 0, SH = 9, SW = 9, SPIX = 81;
  static constexpr int DST = H2D_ROWS, OPIX = 49, OCH = 64;
  static constexpr bool ZERO_BORDER = false;
  static constexpr bool DENSE = false;
  __host__ __device__ static constexpr int tap_u(int t) { return (t / 3) * 9 + t % 3; }
  __host__ __device__ static constexpr int tap_k(int t) { return t; }
};
// conv3 data gradient: dz 7x7x64 (h2p rows) on an 11x11 zero-bordered grid -> dx 9x9x64 (planar); out (y, x) reads
// grid (y + 2 - ky, x + 2 - kx)
template <> struct H2Geo<H2C_D3> {
  static constexpr int CGW = 1;   // channel groups of 16 per wavefront (they share the positions' fragments)
  static constexpr int CB = 2, NCH = 64, NTAP = 9, GW = 11, OH = 9, OW = 9, NPIX_L = 121, G = 2, PLANE_E = 256;
  static constexpr int SRC = H2S_ROWS, PAD = 2, SH = 7, SW = 7, SPIX = 49;
  static constexpr int DST = H2D_PLANAR, OPIX = 81, OCH = 64;
  static constexpr bool ZERO_BORDER = true;
  // outputs enumerated densely (16 consecutive OUTPUT pixels per block, 81 = 5 blocks + 1 pixel) instead of along the 11-wide
  // input grid (97 entries in 7 blocks): a lane's grid entry is its output index + 2 rows' worth of border, a per-lane
  // address again -- 42 instead of 51 tap-blocks per image after the border's tap rows are skipped
  static constexpr bool DENSE = true;
  __host__ __device__ static constexpr int tap_u(int t) { return (2 - t / 3) * 11 + (2 - t % 3); }
  __host__ __device__ static constexpr int tap_k(int t) { return t; }
};
// conv2 data gradient: dz 9x9x64 (planar) on an 11x11 zero-bordered grid -> the four parity classes of dx 20x20x32, float32
// NHWC; class-grid pixel (a, b) reads grid (a + 1 - dy, b + 1 - dx); 128 channels = (class, cin)
template <> struct H2Geo<H2C_D2> {
  static constexpr int CGW = 2;   // channel groups of 16 per wavefront (they share the positions' fragments)
  static constexpr int CB = 2, NCH = 128, NTAP = 4, GW = 11, OH = 10, OW = 10, NPIX_L = 121, G = 2, PLANE_E = 256;
  static constexpr int SRC = H2S_PLANAR, PAD = 1, SH = 9, SW = 9, SPIX = 81;
  static constexpr int DST = H2D_F32_ROUTED, OPIX = 400, OCH = 32;
  static constexpr bool ZERO_BORDER = true;
  static constexpr bool DENSE = true;   // (as conv3's: 100 outputs in 7 blocks either way, but the last block is row 9 alone)
  __host__ __device__ static constexpr int tap_u(int t) { return (1 - t / 2) * 11 + (1 - t % 2); }
  __host__ __device__ static constexpr int tap_k(int t) { return t; }
};

struct H2ConvArgs {
  const void* x;   // images (format per geometry)
  const void* w;   // h2p rows [NCH][K]
  const float* sx;
  const float* sw;
  int64_t n;       // images
  const float* bias;  // [NCH] or null
  int32_t act;        // 1 relu
  void* out;
  float* out_scale;        // h2 outputs: the scale used, written for the consumers
  const float* bound_in;   // h2 outputs: max |x| ...
  const float* bound_w;    // ... * max row 1-norm of w ...
  const float* bound_b;    // ... + max |bias| (or null)
  float* out_absmax;       // required
  // ReLU sign bits.  "h2 order": one byte per group of 8 channels, bit j = element j of the group (h2p_elem), bytes in
  // [image][pixel][32-block][group] order -- what a lane of these kernels holds.  "natural": bit c of the 32-bit word of a
  // pixel's 32-channel block (what round 3's kernels write: the first layer's y_mask).
  uint8_t* mask_out;       // forward kernels (F2, F3): h2 order, required
  const void* mask_in;     // data gradients: the relu derivative at the OUTPUT elements, required: D3 reads h2 order, D2 natural order
};

#ifdef __HIPCC__

// planar image bytes
template <int ID> constexpr int h2c_src_img_bytes() { return H2Geo<ID>::SPIX * H2Geo<ID>::CB * 128; }
template <int ID> constexpr int h2c_slot_bytes() { return H2Geo<ID>::CB * 8 * H2Geo<ID>::PLANE_E * 16; }  // (pixel-major: the same bytes)

// x of lane ^ 32 (v_permlane32_swap: one vector operation, no LDS)
__device__ __forceinline__ uint32_t h2_xor32(uint32_t x, bool upper) {
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
  typedef unsigned h2_u32x2 __attribute__((ext_vector_type(2)));
  const h2_u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return upper ? r[0] : r[1];
#else
  (void)upper;
  return (uint32_t)__shfl_xor((int)x, 32);
#endif
}

#ifndef SRL_H2C_SCHED
#define SRL_H2C_SCHED 1
#endif
// timing experiments (wrong results): 1 no DMA after the first batches, 2 no MFMA chain, 4 no stores
#ifndef SRL_H2C_DBG
#define SRL_H2C_DBG 0
#endif

// k-blocks (tap, channel block) of a block's MFMA chain whose tap row is in the row mask RM (bit r = taps r * TPR .. + TPR - 1)
template <int NKB> struct H2ActiveKb { int n; int kb[NKB]; };
template <int RM, int NTAP, int CB, int TPR> constexpr H2ActiveKb<NTAP * CB> h2_active_kbs() {
  H2ActiveKb<NTAP * CB> a{};
  for (int kb = 0; kb < NTAP * CB; ++kb)
    if ((RM >> ((kb / CB) / TPR)) & 1) a.kb[a.n++] = kb;
  return a;
}

template <int ID, int NSLOT>
__global__ __launch_bounds__(512, 2) void h2conv_kernel(H2ConvArgs a) {
  using GE = H2Geo<ID>;
  constexpr int CB = GE::CB, NKB = GE::NTAP * CB, NPL = CB * 8, PLANE_B = GE::PLANE_E * 16;
  constexpr int SLOT = NPL * PLANE_B;
  constexpr int NCG = GE::NCH / 16;              // channel groups of 16
  constexpr int CGW = GE::CGW, NWG = NCG / CGW;  // ... CGW of them per wavefront: NWG wavefronts across the channels
  constexpr int NPH = 8 / NWG;                   // position halves (8 wavefronts)
  static_assert(NWG == 4 || NWG == 8, "");
  constexpr int NE = GE::DENSE ? GE::OH * GE::OW : (GE::OH - 1) * GE::GW + GE::OW;   // entries to walk per image
  constexpr int NB_IMG = (NE + 15) / 16;               // blocks of 16 entries per image
  constexpr int NBT = NB_IMG * GE::G;                  // blocks per batch
  static_assert(NBT % NPH == 0 && NPH <= 2, "blocks split evenly over the (at most two) position halves");
  constexpr int NDMA = SLOT / 1024;                    // DMA instructions per batch
  constexpr int NDW = (NDMA + 7) / 8;                  // ... per wavefront
  static_assert(SLOT % 1024 == 0, "");
  constexpr int KROW = NKB * 128;                      // bytes of one row of w
  constexpr int IMGB = h2c_src_img_bytes<ID>();
  constexpr bool ROUTED = GE::DST == H2D_F32_ROUTED;
  constexpr int OIMGB = GE::OPIX * GE::OCH * 4;        // output image bytes (every format: 4 bytes per element)
  constexpr int MIMGB = GE::OPIX * GE::OCH / 8;        // mask bytes per output image
  constexpr bool MASK_IN = ID == H2C_D3 || ID == H2C_D2, MASK_OUT = !MASK_IN;
  // data gradients: the ReLU-derivative bits of a batch's output images ride in with the batch (their own DMA instructions,
  // behind the image in the slot) and are read from LDS.  Loaded from global memory inside the epilogue they cost far more than
  // their bytes: vmcnt counts in order, so waiting for a mask word waited for every DMA instruction of the batch prefetched
  // just before it (waves parked in s_waitcnt 48-57 % of their cycles, matrix pipe 37-46 % busy)
  constexpr int MREG = MASK_IN ? (GE::G * MIMGB + 1023) / 1024 * 1024 : 0;
  constexpr int SLOTM = SLOT + MREG;                   // a slot with its mask region
  constexpr int NDM = MREG / 1024;                     // mask DMA instructions per batch
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg0 = (wid % NWG) * CGW, ph = wid / NWG;   // first channel group of this wavefront, its position half
  const int p = lane & 15, quad = lane >> 4;
  const bool up = quad >> 1;

  // ---- this wavefront's weights: rows cg*16 + p, every k-block, both pieces: lane (channel p, group quad)
  h2_f16x8 wf[CGW][NKB][2];
  {
#pragma unroll
    for (int c = 0; c < CGW; ++c) {
      const uint8_t* wr = static_cast<const uint8_t*>(a.w) + (size_t)((cg0 + c) * 16 + p) * KROW + quad * 32;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        wf[c][kb][0] = *reinterpret_cast<const h2_f16x8*>(wr + kb * 128);
        wf[c][kb][1] = *reinterpret_cast<const h2_f16x8*>(wr + kb * 128 + 16);
      }
    }
    // the loads are waited for HERE: otherwise the compiler places its counted vmcnt waits at the first use of each
    // fragment, inside the batch loop, where they would also wait for the DMA of the NEXT batch (issued just before)
#pragma unroll
    for (int c = 0; c < CGW; ++c)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(wf[c][kb][0]), "+v"(wf[c][kb][1]));
  }
  const float inv = 1.f / (*a.sx * *a.sw);
  float oscale = 1.f;
  if (!ROUTED) {
    float bound = *a.bound_in * *a.bound_w;
    if (a.bound_b) bound += *a.bound_b;
    oscale = h2_scale_for(bound);
    if (blockIdx.x == 0 && tid == 0) *a.out_scale = oscale;
  }
  float bq[CGW][4];  // bias of channels (cg0 + c) * 16 + 4 quad + r
#pragma unroll
  for (int c = 0; c < CGW; ++c) {
    bq[c][0] = bq[c][1] = bq[c][2] = bq[c][3] = 0.f;
    if (a.bias) {
      const float4 b4 = *reinterpret_cast<const float4*>(a.bias + (cg0 + c) * 16 + 4 * quad);
      bq[c][0] = b4.x; bq[c][1] = b4.y; bq[c][2] = b4.z; bq[c][3] = b4.w;
    }
  }
  const float lo = a.act == 1 ? 0.f : -3.0e38f;   // relu as a clamp: no branch in the epilogue

  if (GE::ZERO_BORDER) {  // borders are never written by the DMA (masked lanes): zero every slot once
    for (int i = tid; i < NSLOT * SLOTM / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }

  const h2_i32x4 rx = h2_rsrc(a.x);
  h2_i32x4 rmd = h2_rsrc(MASK_IN ? a.mask_in : a.x);   // the mask as a DMA source, bounded: lanes beyond the last image read zeros
  rmd[2] = MASK_IN ? (int)(a.n * MIMGB) : 0;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  const long nbatch = (a.n + GE::G - 1) / GE::G;

  // ---- DMA: instruction j = wid + 8 q fills LDS entries 64 j .. 64 j + 63 of a slot; entry = plane * PLANE_E + e,
  // e = image * NPIX_L + grid cell.  Per-lane source offsets (relative to the batch's first image) and image numbers are
  // batch-invariant: computed once.
  uint32_t dvoff[NDW];
  int dimg[NDW];   // image of the batch this lane's entry belongs to; -1: never loaded (border / padding)
#pragma unroll
  for (int q = 0; q < NDW; ++q) {
    const int j = wid + 8 * q;
    const int nent = j * 64 + lane;
    if (GE::SRC == H2S_LINEAR) {
      dvoff[q] = (uint32_t)(nent * 16);
      dimg[q] = j < NDMA ? 0 : -1;
    } else if (GE::SRC == H2S_ROWSWZ) {   // LDS slot (pixel P, position s) <- the pixel's slot s ^ sigma(P)
      const int P = nent >> 3, sl = nent & 7;
      dvoff[q] = (uint32_t)(P * 128 + 16 * (sl ^ (((P >> 1) & 1) | (((P >> 2) & 1) << 2))));
      dimg[q] = j < NDMA ? 0 : -1;
    } else {
      const int plane = nent / GE::PLANE_E, e = nent - plane * GE::PLANE_E;
      const int il = e / GE::NPIX_L, cell = e - il * GE::NPIX_L;
      const int gy = cell / GE::GW, gx = cell - gy * GE::GW;
      const int sy = gy - GE::PAD, sx = gx - GE::PAD;
      const bool ok = j < NDMA && il < GE::G && (unsigned)sy < (unsigned)GE::SH && (unsigned)sx < (unsigned)GE::SW;
      const int sp = sy * GE::SW + sx;
      if (GE::SRC == H2S_PLANAR) dvoff[q] = (uint32_t)(il * IMGB + plane * (GE::SPIX * 16) + sp * 16);
      else dvoff[q] = (uint32_t)((il * GE::SPIX + sp) * (CB * 128) + plane * 16);  // rows: the plane index IS the 16-byte slot of the pixel's row
      dimg[q] = ok ? il : -1;
      if (!ok) dvoff[q] = 0;
    }
  }
  auto issue = [&](long b, int s) {
    const long img0 = b * GE::G;
    const int left = (int)(a.n - img0 < GE::G ? a.n - img0 : GE::G);  // images of this batch that exist
    const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)(img0 * (long)IMGB));
#pragma unroll
    for (int q = 0; q < NDW; ++q) {
      const uint32_t ldsa = __builtin_amdgcn_readfirstlane(lds0 + s * SLOTM + (wid + 8 * q) * 1024);
      if (dimg[q] >= 0 && dimg[q] < left)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsa), "v"(dvoff[q]), "s"(rx), "s"(soff) : "memory");
    }
    if (MASK_IN && wid < NDM) {   // the batch's mask bytes, linear
      const uint32_t ldsa = __builtin_amdgcn_readfirstlane(lds0 + s * SLOTM + SLOT + wid * 1024);
      const uint32_t msoff = __builtin_amdgcn_readfirstlane((uint32_t)(img0 * (long)MIMGB));
      const uint32_t mvoff = (uint32_t)(wid * 1024 + lane * 16);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsa), "v"(mvoff), "s"(rmd), "s"(msoff) : "memory");
    }
  };

  // ---- per-lane output constants, per channel group c of the wavefront
  uint32_t lane_out[CGW];      // byte offset of this lane's store inside an output image, pixel (0, 0)
  uint32_t lane_mout[CGW];     // byte offset of this lane's mask byte inside an image's mask, pixel 0
  int mi_shift[CGW], mi_pix[CGW];  // data gradients, natural-order mask_in: bit shift inside the pixel's word, pixel offset of the class
  int pix_step_y, pix_step_x;   // output pixel index = oy * pix_step_y + ox * pix_step_x (+ class offset, folded into lane_out)
#pragma unroll
  for (int c = 0; c < CGW; ++c) {
    const int cg = cg0 + c;
    const int g_out = 2 * (quad & 1) + (cg & 1), ocb = cg >> 1;
    if (ROUTED) {
      const int cls = cg >> 1, py = cls >> 1, px = cls & 1;
      lane_out[c] = (uint32_t)(((py * 20 + px) * GE::OCH + 16 * (cg & 1) + 4 * quad) * 4);
      lane_mout[c] = 0;
      mi_shift[c] = 16 * (cg & 1) + 4 * quad;
      mi_pix[c] = py * 20 + px;
    } else {
      lane_out[c] = GE::DST == H2D_PLANAR ? (uint32_t)((((ocb * 4 + g_out) * 2 + (up ? 1 : 0)) * GE::OPIX) * 16)
                                          : (uint32_t)(ocb * 128 + g_out * 32 + (up ? 16 : 0));
      lane_mout[c] = (uint32_t)(ocb * 4 + g_out);
      mi_shift[c] = 0;
      mi_pix[c] = 0;
    }
  }
  if (ROUTED) { pix_step_y = 40; pix_step_x = 2; }
  else { pix_step_y = GE::OW; pix_step_x = 1; }
  constexpr int PIXB = ROUTED ? GE::OCH * 4 : (GE::DST == H2D_PLANAR ? 16 : GE::OCH * 4);   // bytes per output pixel step
  const uint32_t lane_base = GE::SRC == H2S_ROWSWZ ? (uint32_t)(p * 128 + quad * 32) : (uint32_t)(quad * 2 * PLANE_B + p * 16);
  // ROWSWZ: an entry U pixels on from the block's base lies at (base + 128 U) ^ x, x = 16 sigma(P) of the pixel P = base + p + U
  // (mod 8): bits 4 and 6 of an address whose other low bits belong to the lane (bit 4 clear, bit 6 = quad >> 1; block, tap,
  // stage and image offsets are multiples of 128) -- so per U mod 8 the XOR is a per-lane CONSTANT displacement, for both
  // pieces (x and x ^ 16).  Eight displacements per piece, once per launch; a block adds its base to them (16 additions) and
  // every fragment read is table register + immediate.  (Computed per read -- shift, mask, two XORs, two subtractions, two
  // additions -- the address arithmetic was 300 of the 700 vector instructions beside a block's 144 MFMAs, and the vector
  // and matrix instructions of a SIMD do not overlap here: obs_h2.h.)
  int swz_d0[8], swz_d1[8];
  if (GE::SRC == H2S_ROWSWZ) {
    static_assert(GE::SRC != H2S_ROWSWZ || SLOTM % 128 == 0, "slot stride must keep address bits 4-6");
    const uint32_t a_ref = (uint32_t)(uintptr_t)(lds + lane_base);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = (p + u) & 7;
      const uint32_t x = (uint32_t)((((k >> 1) & 1) | (((k >> 2) & 1) << 2)) * 16);
      swz_d0[u] = u * 128 + (int)((a_ref ^ x) - a_ref);
      swz_d1[u] = u * 128 + (int)((a_ref ^ x ^ 16u) - a_ref);
    }
  }
  float amax = 0.f;

  // ---- the deferred epilogue of a block: what the NEXT block's MFMA chain runs beside
  struct Meta {
    uint32_t pixoff;   // pixel index of this lane's position inside its image
    bool ok;           // lane holds a real output
    long img;          // image (uniform)
    uint32_t bits[CGW];   // data gradients: the ReLU-derivative bits of this lane's four channels per channel group (bits 0..3)
  };
  // Stores go through buffer descriptors whose extent is the tensor: a lane without a real output uses an out-of-range offset,
  // which the hardware drops -- no divergent branch in the epilogue.
  // The image's base goes into the VECTOR offset and soffset is the constant 0, on purpose: a buffer store of more than 64 bits
  // needs a wait state before its data registers are overwritten, and the compiler (ROCm 7.2) pads that hazard only when
  // soffset is NOT a register.  With the base in an SGPR soffset it emitted `buffer_store_dwordx4 v[138:141], ..., s87 offen`
  // directly followed by `v_max_f32 v140, ...` / an MFMA writing v[138:141], and one 64-byte segment in ~10^5 was lost or
  // stale, differently from run to run (found as 16 wrong values of one pixel; the same pattern was behind the "wrong results
  // with sched_group_barrier" of the first version of this kernel).
  const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)(a.n * OIMGB), 0x00020000);
  const __amdgpu_buffer_rsrc_t r_mo = __builtin_amdgcn_make_buffer_rsrc(MASK_OUT ? (void*)a.mask_out : a.out, 0, (int)(a.n * MIMGB), 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;
  auto epilogue1 = [&](const h2_f32x4& acc, const Meta& m, const float (&bqc)[4], uint32_t lane_out_c, uint32_t lane_mout_c, uint32_t mbits) {
    float v[4];
    constexpr bool CLAMP = ID == H2C_F2 || ID == H2C_F3;   // data gradients take no activation (srl_h2_conv checks)
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = CLAMP ? fmaxf(acc[r] * inv + bqc[r], lo) : acc[r] * inv + bqc[r];
    const bool ok = m.ok && !(SRL_H2C_DBG & 4);
    const uint32_t img32 = (uint32_t)m.img;
    if (MASK_IN) {   // bit r -> all ones / zero (v_bfe_i32), and: two vector instructions per value
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __uint_as_float(__float_as_uint(v[r]) & (uint32_t)__builtin_amdgcn_sbfe((int)mbits, r, 1));
    }
    {
      // (vector instructions cost what matrix instructions cost here, DESIGN 4: max3, and the exchanges below without selects)
      const float m3 = __builtin_fmaxf(__builtin_fmaxf(fabsf(v[0]), fabsf(v[1])), fabsf(v[2]));
      const float m5 = __builtin_fmaxf(__builtin_fmaxf(m3, fabsf(v[3])), amax);
      amax = m.ok ? m5 : amax;
    }
    const uint32_t ooff = ok ? m.pixoff * PIXB + lane_out_c : OOB;
    typedef uint32_t h2_u32x4 __attribute__((ext_vector_type(4)));
    if (ROUTED) {
      h2_u32x4 d = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      __builtin_amdgcn_raw_buffer_store_b128(d, r_out, ooff + img32 * (uint32_t)OIMGB, 0, 0);
    } else {
      // two pieces of the four values; lanes (quad, quad ^ 2) = (L, L + 32) complete each other's 16-byte chunks: the lower one
      // ends up with the first pieces of elements 0..7 of group g_out, the upper one with the second pieces.
      // v_permlane32_swap a, b exchanges a's upper 32 lanes with b's lower 32: with a = first pieces, b = second pieces, the
      // lower lane holds (own first, partner's first) and the upper lane (partner's second, own second) afterwards -- no selects.
      uint32_t h0a, h0b, h1a, h1b;
      h2_split_pair(v[0], v[1], oscale, h0a, h1a);
      h2_split_pair(v[2], v[3], oscale, h0b, h1b);
      h2_u32x4 chunk;
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
      typedef unsigned h2_u32x2 __attribute__((ext_vector_type(2)));
      const h2_u32x2 sa = __builtin_amdgcn_permlane32_swap(h0a, h1a, false, false);
      const h2_u32x2 sb = __builtin_amdgcn_permlane32_swap(h0b, h1b, false, false);
      chunk[0] = sa[0]; chunk[1] = sb[0]; chunk[2] = sa[1]; chunk[3] = sb[1];
#else
      const uint32_t r0 = h2_xor32(up ? h0a : h1a, up), r1 = h2_xor32(up ? h0b : h1b, up);
      chunk[0] = up ? r0 : h0a; chunk[1] = up ? r1 : h0b; chunk[2] = up ? h1a : r0; chunk[3] = up ? h1b : r1;
#endif
      __builtin_amdgcn_raw_buffer_store_b128(chunk, r_out, ooff + img32 * (uint32_t)OIMGB, 0, 0);
      if (MASK_OUT) {
        uint32_t nib = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
        const h2_u32x2 sn = __builtin_amdgcn_permlane32_swap(nib, nib, false, false);   // lower lanes: (own, partner's)
        const uint32_t byte = sn[0] | (sn[1] << 4);
#else
        const uint32_t byte = nib | (h2_xor32(nib, up) << 4);
#endif
        const uint32_t moff = (ok && !up) ? m.pixoff * (GE::OCH / 8) + lane_mout_c : OOB;
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)byte, r_mo, moff + img32 * (uint32_t)MIMGB, 0, 0);
      }
    }
  };
  auto epilogue = [&](const h2_f32x4 (&accs)[CGW], const Meta& m) {
#pragma unroll
    for (int c = 0; c < CGW; ++c) epilogue1(accs[c], m, bq[c], lane_out[c], lane_mout[c], m.bits[c]);
  };

  // ---- one block: 3 NKB MFMAs on the block's entries, fragments fetched PD k-blocks ahead (LDS latency is ~4-8 MFMAs), with
  // the PREVIOUS block's epilogue inside the chain -- early in it for wavefronts 0-3, in the middle for wavefronts 4-7: the two
  // wavefronts of a SIMD (w, w + 4) run the same code between the same barriers, and what overlaps one's vector work and
  // stores is the other's matrix work.  (sched_group_barrier was tried for a per-instruction interleave: with the directives
  // in the block the compiler placed a vector write to a store's data register directly behind the store -- a missing wait
  // state, wrong results -- so the order is written out in the source instead.)
  // (conv2's data gradient, two channel groups per wavefront: one k-block ahead -- with two, the two instantiations of the
  // position half below spill 20 bytes and the kernel runs at 352 us instead of 286)
  constexpr int PD = CB == 1 ? 3 : (ID == H2C_D2 ? 1 : 2);
  const bool late = wid >= 4;
  // rows_c: the tap rows that can contribute to this block (a data gradient's first and last blocks of an image see only
  // zero border through some: conv3's 9 x 9 outputs on the 11-wide grid skip 12 of their 63 tap-blocks); all rows otherwise
  auto block = [&](auto rows_c, const uint8_t* xb, h2_f32x4 (&pacc)[CGW], Meta& pm, const Meta& nm) {
    constexpr int TPR = ID == H2C_D3 ? 3 : (ID == H2C_D2 ? 2 : GE::NTAP);   // taps per tap row (one row = everything, where nothing is skipped)
    constexpr H2ActiveKb<NKB> AK = h2_active_kbs<decltype(rows_c)::value, GE::NTAP, CB, TPR>();
    h2_f32x4 acc[CGW];
#pragma unroll
    for (int c = 0; c < CGW; ++c) acc[c] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
    h2_f16x8 xr[PD + 1][2];
    const uint8_t* tq0[8];
    const uint8_t* tq1[8];
    if (GE::SRC == H2S_ROWSWZ) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        tq0[u] = xb + swz_d0[u];
        tq1[u] = xb + swz_d1[u];
      }
    }
    auto fetch = [&](int i) {   // i-th active k-block
      const int kb = AK.kb[i];
      const int t = kb / CB, c = kb - t * CB;
      if (GE::SRC == H2S_ROWSWZ) {
        const int U = GE::tap_u(t), u8 = U & 7;
        xr[i % (PD + 1)][0] = *reinterpret_cast<const h2_f16x8*>(tq0[u8] + (U - u8) * 128);
        xr[i % (PD + 1)][1] = *reinterpret_cast<const h2_f16x8*>(tq1[u8] + (U - u8) * 128);
      } else {
        const int off = (c * 8) * PLANE_B + GE::tap_u(t) * 16;
        xr[i % (PD + 1)][0] = *reinterpret_cast<const h2_f16x8*>(xb + off);
        xr[i % (PD + 1)][1] = *reinterpret_cast<const h2_f16x8*>(xb + off + PLANE_B);
      }
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) fetch(i);
#pragma unroll
    for (int i = 0; i < AK.n; ++i) {
      if (i + PD < AK.n) fetch(i + PD);
      if (i == 1 && !late) epilogue(pacc, pm);
      if (i == AK.n / 2 + 1 && late) epilogue(pacc, pm);
      const int kb = AK.kb[i];
      const int t = kb / CB, c = kb - t * CB;
      const int kw = GE::tap_k(t) * CB + c;
#pragma unroll
      for (int cc = 0; cc < CGW; ++cc) {
        acc[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cc][kw][1], xr[i % (PD + 1)][0], acc[cc], 0, 0, 0);
        acc[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cc][kw][0], xr[i % (PD + 1)][1], acc[cc], 0, 0, 0);
        acc[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cc][kw][0], xr[i % (PD + 1)][0], acc[cc], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < CGW; ++c) pacc[c] = acc[c];
    pm = nm;
  };

  h2_f32x4 pacc[CGW];
#pragma unroll
  for (int c = 0; c < CGW; ++c) pacc[c] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
  Meta pm = {};
  pm.ok = false;
  // this lane's entry inside a block, walked block by block: (oy, ox) of entry 16 bi + p
  // phc_c: the wavefront's position half as a constant (two instantiations, chosen by a uniform branch per batch): every block's
  // index inside its image is then known at compile time, and with it the entry arithmetic and the row masks (conv2 forward
  // 302 -> 277 us, conv3 forward 198 -> 188, conv3 data gradient 278 -> 267, conv2 data gradient 290 -> 286)
  auto compute = [&](auto phc_c, long b, int s) {
    constexpr int phc = decltype(phc_c)::value;
    const uint8_t* slot = lds + s * SLOTM + lane_base;
    const uint8_t* mreg = lds + s * SLOTM + SLOT;
#pragma unroll
    for (int jj = 0; jj < NBT / NPH; ++jj) {   // unrolled: a block's accumulators are handed to the next block's chain by renaming
      const int jb = phc * (NBT / NPH) + jj;
      const int il = jb / NB_IMG, bi = jb - il * NB_IMG;
      const int e = bi * 16 + p;
      constexpr int EW = GE::DENSE ? GE::OW : GE::GW;   // entries per row of the enumeration
      const int oy = (e * (65536 / EW + 1)) >> 16, ox = e - oy * EW;   // e < 128
      Meta nm;
      nm.img = b * GE::G + il;
      nm.ok = ox < GE::OW && oy < GE::OH && nm.img < a.n;
      nm.pixoff = (uint32_t)(oy * pix_step_y + ox * pix_step_x);
#pragma unroll
      for (int c = 0; c < CGW; ++c) {
        nm.bits[c] = 0u;
        if (MASK_IN) {   // from the slot's mask region (idle lanes read some other in-region byte: their outputs are dropped)
          const uint32_t po = nm.ok ? nm.pixoff : 0u;
          if (ID == H2C_D3) nm.bits[c] = (uint32_t)mreg[il * MIMGB + po * (GE::OCH / 8) + lane_mout[c]] >> (up ? 4 : 0);
          else nm.bits[c] = *reinterpret_cast<const uint32_t*>(mreg + il * MIMGB + (po + mi_pix[c]) * 4) >> mi_shift[c];
        }
      }
      if (SRL_H2C_DBG & 2) { epilogue(pacc, pm); pm = nm; continue; }
      // (dense enumeration: output (oy, ox) sits at grid entry oy * GW + ox = e + oy * (GW - OW))
      const uint8_t* xb_ = slot + (il * GE::NPIX_L + bi * 16) * (GE::SRC == H2S_ROWSWZ ? 128 : 16) +
                           (GE::DENSE ? oy * ((GE::GW - GE::OW) * 16) : 0);
      if (ID == H2C_D3 && !(SRL_H2C_DBG & 8)) {
        // output rows of block bi: outputs 16 bi .. + 15 of the 9-wide image; tap row ky reads dz row y - ky, which exists for
        // 0 <= y - ky <= 6: block 0 (y = 0, 1) never through ky = 2, block 4 (y = 7, 8) never through ky = 0, block 5 (y = 8) only
        // through ky = 2
        static_assert(ID != H2C_D3 || (NB_IMG == 6 && GE::GW == 11 && GE::OW == 9 && GE::NTAP == 9 && GE::SRC != H2S_ROWSWZ),
                      "row masks below are conv3's");
        if (bi == 0) block(std::integral_constant<int, 0b011>{}, xb_, pacc, pm, nm);
        else if (bi == 4) block(std::integral_constant<int, 0b110>{}, xb_, pacc, pm, nm);
        else if (bi == 5) block(std::integral_constant<int, 0b100>{}, xb_, pacc, pm, nm);
        else block(std::integral_constant<int, 0b111>{}, xb_, pacc, pm, nm);
      } else if (ID == H2C_D2 && !(SRL_H2C_DBG & 8)) {
        // class-grid rows of block bi: outputs 16 bi .. + 15 of the 10-wide class image; tap row dy reads dz row a - dy, which
        // exists for 0 <= a - dy <= 8: block 6 (a = 9 alone) only through dy = 1
        static_assert(ID != H2C_D2 || (NB_IMG == 7 && GE::GW == 11 && GE::OW == 10 && GE::NTAP == 4), "row mask below is conv2's");
        if (bi == 6) block(std::integral_constant<int, 0b10>{}, xb_, pacc, pm, nm);
        else block(std::integral_constant<int, 0b11>{}, xb_, pacc, pm, nm);
      } else {
        block(std::integral_constant<int, (1 << 30) - 1>{}, xb_, pacc, pm, nm);
      }
    }
  };

  // ring of NSLOT slots: batches b, b + grid, ... of this workgroup; the DMA of the batch NSLOT-1 ahead is issued right
  // after the barrier that frees its slot.  The last block's stores of a batch are issued inside the NEXT batch's first
  // block, so the vmcnt(0) at the top of a batch rarely waits for a store.
  long b = blockIdx.x;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s)
    if (b + (long)s * gridDim.x < nbatch) issue(b + (long)s * gridDim.x, s);
  int s_cur = 0, s_nxt = NSLOT - 1;
  for (; b < nbatch; b += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long bn = b + (long)(NSLOT - 1) * gridDim.x;
    if (bn < nbatch && !(SRL_H2C_DBG & 1)) issue(bn, s_nxt);
    if (ph == 0) compute(std::integral_constant<int, 0>{}, b, s_cur);
    else compute(std::integral_constant<int, NPH - 1>{}, b, s_cur);
    s_cur = s_cur + 1 == NSLOT ? 0 : s_cur + 1;
    s_nxt = s_nxt + 1 == NSLOT ? 0 : s_nxt + 1;
  }
  epilogue(pacc, pm);
  {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0) {
      const float cur = __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (amax > cur) atomicMax(reinterpret_cast<int*>(a.out_absmax), __float_as_int(amax));
    }
  }
}

template <int ID, int NSLOT = 2>
inline int h2conv_launch(hipStream_t st, const H2ConvArgs& a, int max_blocks = 256) {
  constexpr bool MASK_IN_L = ID == H2C_D3 || ID == H2C_D2;
  constexpr int SLOT = h2c_slot_bytes<ID>() + (MASK_IN_L ? (H2Geo<ID>::G * (H2Geo<ID>::OPIX * H2Geo<ID>::OCH / 8) + 1023) / 1024 * 1024 : 0);
  static_assert(NSLOT * SLOT <= 160 * 1024, "LDS");
  static bool attr_set = false;
  auto kern = h2conv_kernel<ID, NSLOT>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
    attr_set = true;
  }
  const long nbatch = (a.n + H2Geo<ID>::G - 1) / H2Geo<ID>::G;
  if (nbatch <= 0) return 0;
  const long grid = nbatch < max_blocks ? nbatch : max_blocks;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), NSLOT * SLOT, st, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// float32 NHWC [n, H, W, C] -> planar h2 images [n][C/32][4][2][NPIX][16 B]; order 0: entry = y * W + x; order 2: parity-class
// major (the layout a stride-2 consumer reads): entry = ((y & 1) * 2 + (x & 1)) * (H/2 * W/2) + (y >> 1) * (W/2) + (x >> 1)
__host__ __device__ inline int h2_entry(int y, int x, int H, int W, int order) {
  return order == 2 ? ((y & 1) * 2 + (x & 1)) * ((H / 2) * (W / 2)) + (y >> 1) * (W / 2) + (x >> 1) : y * W + x;
}
static __global__ void h2_pack_planar_kernel(const float* __restrict__ src, int64_t n, int H, int W, int C, int order, const float* absmax,
                                      const float* scale_in, float* scale_out, uint8_t* __restrict__ dst) {
  const float scale = scale_in ? *scale_in : h2_scale_for(*absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  const int64_t ngrp = n * H * W * (C / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngrp; t += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(t % (C / 8));
    int64_t r = t / (C / 8);
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int64_t img = r / H;
    const int blk = gi >> 2, g = gi & 3;
    const float* s = src + ((img * H + y) * W + x) * C + blk * 32;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[h2p_elem(g, j)];
    uint4 h0, h1;
    h2_split_pair(v[0], v[1], scale, h0.x, h1.x);
    h2_split_pair(v[2], v[3], scale, h0.y, h1.y);
    h2_split_pair(v[4], v[5], scale, h0.z, h1.z);
    h2_split_pair(v[6], v[7], scale, h0.w, h1.w);
    const int NP = H * W, e = h2_entry(y, x, H, W, order);
    uint8_t* d = dst + img * (int64_t)(NP * C * 4);
    *reinterpret_cast<uint4*>(d + ((int64_t)((blk * 4 + g) * 2 + 0) * NP + e) * 16) = h0;
    *reinterpret_cast<uint4*>(d + ((int64_t)((blk * 4 + g) * 2 + 1) * NP + e) * 16) = h1;
  }
}
// float32 NHWC -> h2p rows [n][NPIX][C] with the pixels of an image in `order` (h2_entry)
static __global__ void h2_pack_pixrows_kernel(const float* __restrict__ src, int64_t n, int H, int W, int C, int order, const float* absmax,
                                       const float* scale_in, float* scale_out, uint8_t* __restrict__ dst) {
  const float scale = scale_in ? *scale_in : h2_scale_for(*absmax);
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
  const int64_t ngrp = n * H * W * (C / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngrp; t += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(t % (C / 8));
    int64_t r = t / (C / 8);
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int64_t img = r / H;
    const int blk = gi >> 2, g = gi & 3;
    const float* s = src + ((img * H + y) * W + x) * C + blk * 32;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = s[h2p_elem(g, j)];
    uint4 h0, h1;
    h2_split_pair(v[0], v[1], scale, h0.x, h1.x);
    h2_split_pair(v[2], v[3], scale, h0.y, h1.y);
    h2_split_pair(v[4], v[5], scale, h0.z, h1.z);
    h2_split_pair(v[6], v[7], scale, h0.w, h1.w);
    uint4* d = reinterpret_cast<uint4*>(dst + ((img * (int64_t)(H * W) + h2_entry(y, x, H, W, order)) * C + blk * 32) * 4 + g * 32);
    d[0] = h0;
    d[1] = h1;
  }
}
static __global__ void h2_unpack_planar_kernel(const uint8_t* __restrict__ src, int64_t n, int H, int W, int C, int order, const float* scale,
                                        float* __restrict__ dst) {
  const float inv = 1.f / *scale;
  const int64_t ngrp = n * H * W * (C / 8);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < ngrp; t += (int64_t)gridDim.x * blockDim.x) {
    const int gi = (int)(t % (C / 8));
    int64_t r = t / (C / 8);
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int64_t img = r / H;
    const int blk = gi >> 2, g = gi & 3;
    const int NP = H * W, e = h2_entry(y, x, H, W, order);
    const uint8_t* d = src + img * (int64_t)(NP * C * 4);
    const _Float16* p0 = reinterpret_cast<const _Float16*>(d + ((int64_t)((blk * 4 + g) * 2 + 0) * NP + e) * 16);
    const _Float16* p1 = reinterpret_cast<const _Float16*>(d + ((int64_t)((blk * 4 + g) * 2 + 1) * NP + e) * 16);
    float* o = dst + ((img * H + y) * W + x) * C + blk * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[h2p_elem(g, j)] = ((float)p0[j] + (float)p1[j]) * inv;
  }
}

#endif  // __HIPCC__

// =====================================================================================================================
// Weight gradients, image-stationary:  dW[cout][tap][cin] += sum over images and positions dz[pos][cout] x[pos + tap][cin].
// The summation index is the POSITION, which is the slow index of both operands as they lie in LDS (entries of 8 channels):
// both MFMA operands are fetched with ds_read_b64_tr_b16 (4 positions x 16 channels per 16 lanes, delivered per channel).
//   v_mfma_f32_16x16x32_f16: A = dz^T (16 output channels x 32 positions), B = x (32 positions x 16 input channels of one tap),
//   D = 16 cout x 16 cin of that tap.  The 32 positions of a K-chunk are 32 consecutive entries of the INPUT grid's numbering
//   (as in the forward kernel); dz is staged on that same grid with its unused columns zero, so a position that is not an
//   output contributes nothing and needs no test.  Entry order inside a chunk: octet o, element e -> entry 4 o + e (e < 4),
//   16 + 4 o + e - 4 (e >= 4): the 32 lanes of a transposed read then cover 8 consecutive entries of two channel groups,
//   and with the planes of groups 2, 3 skewed by 128 bytes those are 32 distinct 8-byte bank slots.
// Accumulators stay in registers over ALL images of a (persistent) workgroup: 16-18 tiles of 16x16 per wavefront; a launch
// ends with one float32 slab [cout][K] (+ the bias gradient) per workgroup, summed by h2_wgrad_reduce_kernel.
enum { H2W_C2 = 0, H2W_C3 = 1 };

template <int ID> struct H2WGeo;
// conv2: x = a1, h2p rows in parity-class-major pixel order (pixel-major in LDS, sigma-swizzled like the forward kernel);
// dz = dz2 planar 9x9x64, staged on the 10-wide class grid.  Wavefront w: every cout block, taps 2w, 2w+1, both cin blocks.
template <> struct H2WGeo<H2W_C2> {
  static constexpr int CIN = 32, NTAP = 16, GW = 10, OH = 9, OW = 9, XPIX = 400, ZPIX = 81, ZW = 9;
  static constexpr bool X_ROWS = true, Z_ROWS = false;
  static constexpr int MC = 4, NN = 4;   // cout blocks x (tap, cin block) pairs per wavefront
  __host__ __device__ static constexpr int tap_u(int t) { return (((t >> 2) & 1) * 2 + (t & 1)) * 100 + (t >> 3) * 10 + ((t & 3) >> 1); }
};
// conv3: x = a2 planar 9x9x64; dz = dz3, h2p rows [49][64], staged planar on the 9-wide grid.  Wavefront w: cout half w & 1,
// 9 of the 36 (tap, cin block) pairs.
template <> struct H2WGeo<H2W_C3> {
  static constexpr int CIN = 64, NTAP = 9, GW = 9, OH = 7, OW = 7, XPIX = 81, ZPIX = 49, ZW = 7;
  static constexpr bool X_ROWS = false, Z_ROWS = true;
  static constexpr int MC = 2, NN = 9;
  __host__ __device__ static constexpr int tap_u(int t) { return (t / 3) * 9 + t % 3; }
};

struct H2WgradArgs {
  const void* x;    // activations of the layer's input
  const void* dz;   // gradient of the layer's output (64 channels)
  const float* sx;
  const float* sz;
  int64_t n;
  float* slabs;     // [grid][64 * K + 64] float32: partial dW ([cout][tap][cin]) and db of each workgroup
};

#ifdef __HIPCC__

typedef short h2_s16x4 __attribute__((ext_vector_type(4)));

// byte offset of plane index (cb * 4 + g) * 2 + piece inside a staged planar image whose planes are `plane` bytes: groups 2, 3
// lie 128 bytes (mod 256) off groups 0, 1 of the same block -- the 32 lanes of a transposed read take both -- and each block
// starts 256 bytes later so that the skewed planes do not run into the next block
__host__ __device__ constexpr int h2w_plane_base(int idx, int plane) { return idx * plane + (idx >> 3) * 256 + (((idx >> 1) & 3) >= 2 ? 128 : 0); }

__device__ __forceinline__ h2_s16x4 h2_tr_read(const uint8_t* p) {
  typedef __attribute__((address_space(3))) h2_s16x4 lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

template <int ID, int NSLOT>
__global__ __launch_bounds__(512, 2) void h2wgrad_kernel(H2WgradArgs a) {
  using GE = H2WGeo<ID>;
  constexpr int CBX = GE::CIN / 32;                       // 32-channel blocks of x
  constexpr int NE = (GE::OH - 1) * GE::GW + GE::OW;      // grid entries that can be outputs
  constexpr int NCHUNK = (NE + 31) / 32;
  constexpr int ZE = NCHUNK * 32;                         // dz grid entries staged per image (zero beyond the outputs)
  constexpr int ZPLANE = ZE * 16;                         // bytes of one dz plane
  constexpr int ZBYTES = 16 * ZPLANE + 512;               // 2 blocks x 4 groups x 2 pieces; groups 2, 3 skewed by 128 bytes, blocks by 256
  constexpr int XE = GE::X_ROWS ? GE::XPIX : ((GE::XPIX + 15) / 16) * 16;   // x entries per plane (planar) / pixels (rows)
  constexpr int XPLANE = XE * 16;
  constexpr int XBYTES = GE::X_ROWS ? GE::XPIX * 128 : CBX * 8 * XPLANE + CBX * 256;
  constexpr int XB_AL = (XBYTES + 1023) / 1024 * 1024, ZB_AL = (ZBYTES + 1023) / 1024 * 1024;
  constexpr int SLOT = XB_AL + ZB_AL;
  constexpr int NDX = XB_AL / 1024, NDZ = ZB_AL / 1024, NDMA = NDX + NDZ, NDW = (NDMA + 7) / 8;
  constexpr int K = GE::NTAP * GE::CIN;
  constexpr int XIMGB = GE::XPIX * GE::CIN * 4, ZIMGB = GE::ZPIX * 64 * 4;
  static_assert(NSLOT * SLOT <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;

  // zero every slot once: the dz grid's unused entries (and the skew gaps) are never written by the DMA
  for (int i = tid; i < NSLOT * SLOT / 16; i += 512) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // ---- DMA: instruction j = wid + 8 qd; j < NDX fills x's region, the rest dz's.  Per-lane source offsets are
  // batch-invariant; -1 marks lanes that never load.
  const h2_i32x4 rx = h2_rsrc(a.x), rz = h2_rsrc(a.dz);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
  int dvoff[NDW];
#pragma unroll
  for (int qd = 0; qd < NDW; ++qd) {
    const int j = wid + 8 * qd;
    int off = -1;
    if (j < NDX) {
      const int b = j * 1024 + lane * 16;   // byte inside x's region
      if (GE::X_ROWS) {
        const int P = b >> 7, sl = (b >> 4) & 7;
        if (P < GE::XPIX) off = P * 128 + 16 * (sl ^ (((P >> 1) & 1) | (((P >> 2) & 1) << 2)));
      } else {
        // plane pl_i = (cb * 4 + g) * 2 + piece lies at h2w_plane_base(pl_i, XPLANE)
#pragma unroll
        for (int pl_i = 0; pl_i < CBX * 8; ++pl_i) {
          const int rel = b - h2w_plane_base(pl_i, XPLANE);
          if (rel >= 0 && rel < XPLANE && (rel >> 4) < GE::XPIX) off = pl_i * (GE::XPIX * 16) + (rel >> 4) * 16;
        }
      }
    } else if (j < NDMA) {
      const int b = (j - NDX) * 1024 + lane * 16;
      // dz planes at h2w_plane_base(pl_i, ZPLANE); entry = grid cell (oy, ox) of the GW-wide grid
#pragma unroll
      for (int pl_i = 0; pl_i < 16; ++pl_i) {
        const int rel = b - h2w_plane_base(pl_i, ZPLANE);
        if (rel >= 0 && rel < ZPLANE) {
          const int e = rel >> 4, oy = e / GE::GW, ox = e - oy * GE::GW;
          if (ox < GE::OW && oy < GE::OH) {
            const int zp = oy * GE::ZW + ox;
            off = GE::Z_ROWS ? zp * 256 + pl_i * 16 : pl_i * (GE::ZPIX * 16) + zp * 16;
          }
        }
      }
    }
    dvoff[qd] = off;
  }
  auto issue = [&](long img, int s) {
    const uint32_t sox = __builtin_amdgcn_readfirstlane((uint32_t)(img * (long)XIMGB));
    const uint32_t soz = __builtin_amdgcn_readfirstlane((uint32_t)(img * (long)ZIMGB));
#pragma unroll
    for (int qd = 0; qd < NDW; ++qd) {
      const int j = wid + 8 * qd;
      const uint32_t ldsa = __builtin_amdgcn_readfirstlane(lds0 + s * SLOT + j * 1024);
      if (dvoff[qd] >= 0) {
        if (j < NDX) asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsa), "v"(dvoff[qd]), "s"(rx), "s"(sox) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(ldsa), "v"(dvoff[qd]), "s"(rz), "s"(soz) : "memory");
      }
    }
  };

  // ---- operand addresses of this lane inside a slot.  A transposed read: lane 4 q + p of octet o names position 4 o + q
  // (first read; + 16: second) and the 8 bytes of channels 4 p .. 4 p + 3 of a 16-channel block c16: group 2 (p & 1) + (c16 & 1),
  // elements 4 (p >> 1) .. of its entry.
  auto zplane = [&](int cb32, int g, int pl) { return h2w_plane_base((cb32 * 4 + g) * 2 + pl, ZPLANE); };
  auto xplane = [&](int cb32, int g, int pl) { return h2w_plane_base((cb32 * 4 + g) * 2 + pl, XPLANE); };
  const int ent = 4 * o + q, half8 = 8 * (p >> 1);
  // cout blocks of this wavefront
  const int mc0 = GE::MC == 4 ? 0 : (wid & 1) * 2;
  const int nn0 = GE::MC == 4 ? wid * GE::NN : (wid >> 1) * GE::NN;   // first (tap, cin block) pair
  uint32_t za[GE::MC];   // piece 0, chunk 0, read 0
#pragma unroll
  for (int m = 0; m < GE::MC; ++m) {
    const int cbk = mc0 + m;
    za[m] = (uint32_t)(XB_AL + zplane(cbk >> 1, 2 * (p & 1) + (cbk & 1), 0) + ent * 16 + half8);
  }
  // x: per cin block c16x of the pair (pair index nn -> tap = nn / (CIN/16), c16x = nn % (CIN/16))
  constexpr int CPB = GE::CIN / 16;
  uint32_t swz_lo = 0, swz_hi = 0;
  if (GE::X_ROWS) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = (ent + u) & 7;
      const uint32_t t = (uint32_t)((((k >> 1) & 1) | (((k >> 2) & 1) << 2)) * 16);
      if (u < 4) swz_lo |= t << (8 * u);
      else swz_hi |= t << (8 * (u - 4));
    }
  }

  h2_f32x4 acc[GE::MC][GE::NN];
#pragma unroll
  for (int m = 0; m < GE::MC; ++m)
#pragma unroll
    for (int n2 = 0; n2 < GE::NN; ++n2) acc[m][n2] = h2_f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient: D[cout i][any column] of the products with ones -- lane l, register r: cout 4 (l >> 4) + r of the block.
  // conv2, wavefronts 0..3: block wid in [0]; conv3, wavefronts 0, 1: their two blocks
  h2_f32x4 dbacc[2] = {h2_f32x4{0.f, 0.f, 0.f, 0.f}, h2_f32x4{0.f, 0.f, 0.f, 0.f}};
  h2_f16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

  auto frag = [&](const uint8_t* p0, const uint8_t* p1) {
    union { h2_s16x4 s[2]; h2_f16x8 v; } f;
    f.s[0] = h2_tr_read(p0);
    f.s[1] = h2_tr_read(p1);
    return f.v;
  };
  auto compute = [&](int s) {
    const uint8_t* slot = lds + s * SLOT;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      h2_f16x8 af[GE::MC][2];
#pragma unroll
      for (int m = 0; m < GE::MC; ++m)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const uint8_t* b = slot + za[m] + pl * ZPLANE + c * 512;
          af[m][pl] = frag(b, b + 256);
        }
      // bias gradient from the fragments a wavefront holds anyway: both pieces against a fragment of ones on the matrix cores
      // (two MFMAs per cout block and chunk).  As 16 conversions + 16 additions per piece pair on the vector unit it made the
      // two (conv3) / four (conv2) wavefronts that own the sums ~ 20-30 % longer than the others, which wait for them at every
      // image's barrier -- and vector time is not hidden under matrix time here.
      {
        const bool mine = GE::MC == 4 ? wid < 4 : wid < 2;
        if (mine) {
#pragma unroll
          for (int m = 0; m < GE::MC; ++m) {
            if (GE::MC == 4 && m != (wid & 3)) continue;
            h2_f32x4& d = dbacc[GE::MC == 4 ? 0 : (m & 1)];
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][0], ones, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][1], ones, d, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int n2 = 0; n2 < GE::NN; ++n2) {
        const int nn = nn0 + n2;   // wavefront-uniform at run time for conv3 (nn0 runtime) -> addresses computed, not immediates
        const int tap = nn / CPB, c16x = nn - tap * CPB;
        h2_f16x8 xf[2];
        if (GE::X_ROWS) {
          // tap_u needs a compile-time tap for the table index: conv2 has nn0 = 4 wid -> tap = 2 wid + (n2 >> 1): runtime.
          // U mod 8 at run time: table byte selected with a shift
          const int U = GE::tap_u(tap);
          const int P0 = c * 32 + U;   // + ent (+16)
          const uint32_t u8 = (uint32_t)(U & 7);
          const uint32_t x = (uint32_t)(((((uint64_t)swz_hi << 32) | swz_lo) >> (8 * u8)) & 0xffu);
          const uint32_t g0 = (uint32_t)((2 * (2 * (p & 1) + c16x)) * 16);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const uint32_t a0 = (uint32_t)((P0 + ent) * 128) + ((g0 + pl * 16) ^ x) + half8;
            const uint32_t a1 = (uint32_t)((P0 + ent + 16) * 128) + ((g0 + pl * 16) ^ x) + half8;   // + 16 pixels: same low bits
            xf[pl] = frag(slot + a0, slot + a1);
          }
        } else {
          const int U = GE::tap_u(tap);
          const int cb32 = c16x >> 1;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const uint32_t a0 = (uint32_t)(xplane(cb32, 2 * (p & 1) + (c16x & 1), pl) + (c * 32 + U + ent) * 16 + half8);
            xf[pl] = frag(slot + a0, slot + a0 + 256);
          }
        }
#pragma unroll
        for (int m = 0; m < GE::MC; ++m) {
          acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][1], xf[0], acc[m][n2], 0, 0, 0);
          acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][0], xf[1], acc[m][n2], 0, 0, 0);
          acc[m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m][0], xf[0], acc[m][n2], 0, 0, 0);
        }
      }
    }
  };

  long img = blockIdx.x;
#pragma unroll
  for (int s = 0; s < NSLOT - 1; ++s)
    if (img + (long)s * gridDim.x < a.n) issue(img + (long)s * gridDim.x, s);
  int s_cur = 0, s_nxt = NSLOT - 1;
  for (; img < a.n; img += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long in = img + (long)(NSLOT - 1) * gridDim.x;
    if (in < a.n) issue(in, s_nxt);
    compute(s_cur);
    s_cur = s_cur + 1 == NSLOT ? 0 : s_cur + 1;
    s_nxt = s_nxt + 1 == NSLOT ? 0 : s_nxt + 1;
  }

  // ---- slab of this workgroup: dW[cout][tap][cin] (cout = 16 block + 4 (lane >> 4) + r, cin = 16 c16x + (lane & 15)), then db
  const float inv = 1.f / (*a.sx * *a.sz);
  float* slab = a.slabs + (size_t)blockIdx.x * (64 * K + 64);
#pragma unroll
  for (int m = 0; m < GE::MC; ++m)
#pragma unroll
    for (int n2 = 0; n2 < GE::NN; ++n2) {
      const int nn = nn0 + n2, tap = nn / CPB, c16x = nn - tap * CPB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cout = 16 * (mc0 + m) + 4 * (lane >> 4) + r;
        slab[(size_t)cout * K + tap * GE::CIN + c16x * 16 + (lane & 15)] = acc[m][n2][r] * inv;
      }
    }
  {
    // db: cout i = lane (< 16) of the block sits in register i & 3 of the lanes 16 (i >> 2) .. + 15 (every column the same sum)
    const bool mine = GE::MC == 4 ? wid < 4 : wid < 2;
#pragma unroll
    for (int m = 0; m < (GE::MC == 4 ? 1 : 2); ++m) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __shfl(dbacc[m][r], 16 * ((lane & 15) >> 2));
        if ((lane & 3) == r) t = v;
      }
      if (mine && lane < 16) slab[64 * K + 16 * (GE::MC == 4 ? wid : mc0 + m) + lane] = t / *a.sz;
    }
  }
}

// dst[i] += sum over slabs; i < nw: weight gradient, then the bias gradients.  64 outputs per workgroup, four groups of 64
// threads each taking every fourth slab with four loads in flight (one thread per output walking 256 slabs one after the other
// was a chain of dependent-latency loads: 63 us per call)
static __global__ __launch_bounds__(256) void h2_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslab, int per_slab, int nw,
                                                                     float* __restrict__ gw, float* __restrict__ gb) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < per_slab) {
    int k = sg;
    for (; k + 12 < nslab; k += 16) {
      s0 += slabs[(size_t)k * per_slab + i];
      s1 += slabs[(size_t)(k + 4) * per_slab + i];
      s2 += slabs[(size_t)(k + 8) * per_slab + i];
      s3 += slabs[(size_t)(k + 12) * per_slab + i];
    }
    for (; k < nslab; k += 4) s0 += slabs[(size_t)k * per_slab + i];
  }
  part[sg][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sg == 0 && i < per_slab) {
    const float s = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
    if (i < nw) gw[i] += s;
    else if (gb) gb[i - nw] += s;
  }
}

template <int ID, int NSLOT>
inline int h2wgrad_launch(hipStream_t st, const H2WgradArgs& a, int grid) {
  using GE = H2WGeo<ID>;
  auto kern = h2wgrad_kernel<ID, NSLOT>;
  constexpr int CBX = GE::CIN / 32;
  constexpr int NE = (GE::OH - 1) * GE::GW + GE::OW, NCHUNK = (NE + 31) / 32, ZE = NCHUNK * 32;
  constexpr int ZBYTES = 16 * ZE * 16 + 512;
  constexpr int XE = GE::X_ROWS ? GE::XPIX : ((GE::XPIX + 15) / 16) * 16;
  constexpr int XBYTES = GE::X_ROWS ? GE::XPIX * 128 : CBX * 8 * XE * 16 + CBX * 256;
  constexpr int SLOT = (XBYTES + 1023) / 1024 * 1024 + (ZBYTES + 1023) / 1024 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), NSLOT * SLOT, st, a);
  return 0;
}

#endif  // __HIPCC__ (weight gradients)

}  // namespace srlh2
