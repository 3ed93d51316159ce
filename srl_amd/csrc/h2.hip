// C ABI of the pre-split ("h2") kernels: h2gemm.h (LDS-DMA ring GEMM) and h2conv.h (image-stationary convolutions, data and
// weight gradients).  See include/srl_hip.h for the contracts.
#include "../../include/srl_hip.h"
#include "h2conv.h"
#include "h2tn.h"
#include "h2gemmp.h"
#include "gemm_core.h"
#include "srl_common.h"

using namespace srlh2;

namespace {

// one workgroup per row of the h2p weight matrix dst [rows][K]: element (r, k) of the matrix is w[index(r, k)]
//   mode 0: w [rows][K];  mode 1: w [K][rows] (transposed);  mode 2: data-gradient regrouping of a convolution weight
//   w [Cout][KH][KW][Cin]: row = cls * Cin + ci, k = t * Cout + co with cls = py * st + px, t = dy * (KW / st) + dx ->
//   w[co][py + st dy][px + st dx][ci]
__global__ __launch_bounds__(256) void h2_weights_kernel(const float* __restrict__ w, int rows, int K, int mode, int Cin, int KH, int KW,
                                                         int st, int Cout, const float* absmax, float* scale_out, float* rownorm,
                                                         uint8_t* __restrict__ dst) {
  const int r = blockIdx.x;
  const float scale = h2_scale_for(*absmax);
  if (r == 0 && threadIdx.x == 0) *scale_out = scale;
  auto at = [&](int k) -> float {
    if (mode == 0) return w[(long)r * K + k];
    if (mode == 1) return w[(long)k * rows + r];
    const int TW = KW / st;
    const int co = k % Cout, t = k / Cout, dy = t / TW, dx = t - dy * TW;
    const int ci = r % Cin, cls = r / Cin, py = cls / st, px = cls - py * st;
    return w[((long)(co * KH + py + st * dy) * KW + px + st * dx) * Cin + ci];
  };
  float l1 = 0.f;
  for (int gi = threadIdx.x; gi < K / 8; gi += 256) {
    const int blk = gi >> 2, g = gi & 3;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = at(blk * 32 + h2p_elem(g, j)); l1 += fabsf(v[j]); }
    uint4 h0, h1;
    h2_split_pair(v[0], v[1], scale, h0.x, h1.x);
    h2_split_pair(v[2], v[3], scale, h0.y, h1.y);
    h2_split_pair(v[4], v[5], scale, h0.z, h1.z);
    h2_split_pair(v[6], v[7], scale, h0.w, h1.w);
    uint4* d = reinterpret_cast<uint4*>(dst + ((long)r * K + blk * 32) * 4 + g * 32);
    d[0] = h0;
    d[1] = h1;
  }
  __shared__ float sh[4];
  for (int o = 32; o; o >>= 1) l1 += __shfl_xor(l1, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = l1;
  __syncthreads();
  if (threadIdx.x == 0 && rownorm) atomicMax(reinterpret_cast<int*>(rownorm), __float_as_int((sh[0] + sh[1] + sh[2] + sh[3]) * 1.0001f));
}

}  // namespace

extern "C" int srl_h2_pack_rows(void* stream, const float* src, int64_t ld, int64_t rows, int32_t C, const float* absmax,
                                const float* scale_in, float* scale_out, void* dst) {
  SRL_CHECK_ARG(src && dst && (absmax || scale_in) && rows >= 0 && C > 0 && C % 32 == 0 && ld >= C, "null tensor / C not a multiple of 32");
  if (rows == 0) return 0;
  long blocks = srl_ceil_div(rows * (C / 8), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(h2_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (int64_t)ld, rows, (int)C, absmax, scale_in,
                     scale_out, static_cast<uint8_t*>(dst));
  SRL_LAUNCH_CHECK();
  return 0;
}

// srl_h2_pack_rows with the column sums of src added into colsum [C] (accumulate != 0) or written there: slabs per workgroup in
// `workspace` (srl_h2_pack_rows_colsum_workspace floats), summed in a fixed order.  C <= 2048.
static int h2_pack_colsum_blocks(int64_t rows) {
  int64_t b = srl_ceil_div(rows, 32L);
  return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}
extern "C" int64_t srl_h2_pack_rows_colsum_workspace(int64_t rows, int32_t C) { return rows > 0 && C > 0 ? (int64_t)h2_pack_colsum_blocks(rows) * C : 0; }
extern "C" int srl_h2_pack_rows_colsum(void* stream, const float* src, int64_t ld, int64_t rows, int32_t C, const float* absmax,
                                       const float* scale_in, float* scale_out, void* dst, float* workspace, float* colsum, int32_t accumulate) {
  SRL_CHECK_ARG(src && dst && workspace && colsum && (absmax || scale_in) && rows >= 0 && C > 0 && C % 32 == 0 && C <= 2048 && ld >= C && ld % 4 == 0 &&
                    ((uintptr_t)src & 15) == 0,
                "null tensor / C not a multiple of 32 or above 2048 / unaligned rows");
  if (rows == 0) return 0;
  const int nb = h2_pack_colsum_blocks(rows), lanes = 256 / (C / 8);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(h2_pack_colsum_kernel, dim3((unsigned)nb), dim3(256), (size_t)lanes * C * sizeof(float), st, src, (int64_t)ld, rows, (int)C, absmax,
                     scale_in, scale_out, static_cast<uint8_t*>(dst), workspace);
  hipLaunchKernelGGL(h2_colsum_finish_kernel, dim3((unsigned)srl_ceil_div(C, 32)), dim3(256), 0, st, workspace, nb, (int)C, colsum, (int)accumulate);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_unpack_rows(void* stream, const void* src, int64_t rows, int32_t C, const float* scale, float* dst, int64_t ld) {
  SRL_CHECK_ARG(src && dst && scale && rows >= 0 && C > 0 && C % 32 == 0 && ld >= C, "null tensor / C not a multiple of 32");
  if (rows == 0) return 0;
  long blocks = srl_ceil_div(rows * (C / 8), 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(h2_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const uint8_t*>(src), rows, (int)C,
                     scale, dst, (int64_t)ld);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_pack_image(void* stream, const float* src, int64_t n, int32_t H, int32_t W, int32_t C, int32_t layout,
                                 const float* absmax, const float* scale_in, float* scale_out, void* dst) {
  SRL_CHECK_ARG(src && dst && (absmax || scale_in) && n >= 0 && C > 0 && C % 32 == 0 && layout >= 0 && layout <= 2, "bad argument");
  SRL_CHECK_ARG(layout != 2 || (H % 2 == 0 && W % 2 == 0), "parity-class order needs even extents");
  if (n == 0) return 0;
  long blocks = srl_ceil_div(n * H * W * (C / 8), 256);
  if (blocks > 8192) blocks = 8192;
  if (layout == 0)
    hipLaunchKernelGGL(h2_pack_planar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, n, (int)H, (int)W, (int)C, 0, absmax,
                       scale_in, scale_out, static_cast<uint8_t*>(dst));
  else
    hipLaunchKernelGGL(h2_pack_pixrows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, n, (int)H, (int)W, (int)C,
                       layout == 2 ? 2 : 0, absmax, scale_in, scale_out, static_cast<uint8_t*>(dst));
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_unpack_image(void* stream, const void* src, int64_t n, int32_t H, int32_t W, int32_t C, int32_t layout,
                                   const float* scale, float* dst) {
  SRL_CHECK_ARG(src && dst && scale && n >= 0 && C > 0 && C % 32 == 0 && (layout == 0 || layout == 1), "planar (0) or raster rows (1)");
  if (n == 0) return 0;
  long blocks = srl_ceil_div(n * H * W * (C / 8), 256);
  if (blocks > 8192) blocks = 8192;
  if (layout == 0)
    hipLaunchKernelGGL(h2_unpack_planar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const uint8_t*>(src), n,
                       (int)H, (int)W, (int)C, 0, scale, dst);
  else
    hipLaunchKernelGGL(h2_unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const uint8_t*>(src),
                       n * H * W, (int)C, scale, dst, (int64_t)C);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_weights(void* stream, const float* w, int32_t rows, int32_t K, int32_t mode, const srl_conv_desc* d,
                              const float* absmax, float* scale_out, float* rownorm_out, void* dst) {
  SRL_CHECK_ARG(w && dst && absmax && scale_out && rows > 0 && K > 0 && K % 32 == 0 && mode >= 0 && mode <= 2, "bad argument");
  int Cin = 0, KH = 0, KW = 0, st = 1, Cout = 0;
  if (mode == 2) {
    SRL_CHECK_ARG(d && d->KH % d->stride == 0 && d->KW % d->stride == 0 && rows == d->stride * d->stride * d->Cin &&
                      K == (d->KH / d->stride) * (d->KW / d->stride) * d->Cout,
                  "regrouping: rows = stride^2 Cin, K = (KH / stride)(KW / stride) Cout");
    Cin = d->Cin; KH = d->KH; KW = d->KW; st = d->stride; Cout = d->Cout;
  }
  if (rownorm_out) (void)hipMemsetAsync(rownorm_out, 0, sizeof(float), (hipStream_t)stream);
  hipLaunchKernelGGL(h2_weights_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, w, (int)rows, (int)K, (int)mode, Cin, KH, KW, st,
                     Cout, absmax, scale_out, rownorm_out, static_cast<uint8_t*>(dst));
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_conv(void* stream, int32_t kind, const srl_h2_conv_args* p) {
  SRL_CHECK_ARG(p && p->x && p->w && p->sx && p->sw && p->out && p->out_absmax && p->n >= 0, "null argument");
  SRL_CHECK_ARG(kind >= 0 && kind <= 3, "kind: 0 conv2 forward, 1 conv3 forward, 2 conv3 data gradient, 3 conv2 data gradient");
  SRL_CHECK_ARG(p->n * 20 * 20 * 32 * 4 < 0x7fffffffL, "tensors of 2 GiB and more: split the batch");
  if (p->n == 0) return 0;
  H2ConvArgs a{};
  a.x = p->x; a.w = p->w; a.sx = p->sx; a.sw = p->sw; a.n = p->n; a.bias = p->bias; a.act = p->act; a.out = p->out;
  a.out_scale = p->out_scale; a.bound_in = p->bound_in; a.bound_w = p->bound_w; a.bound_b = p->bound_b; a.out_absmax = p->out_absmax;
  a.mask_out = static_cast<uint8_t*>(p->mask_out); a.mask_in = p->mask_in;
  hipStream_t st = (hipStream_t)stream;
  const bool fwd = kind == H2C_F2 || kind == H2C_F3;
  SRL_CHECK_ARG(fwd ? a.mask_out != nullptr : a.mask_in != nullptr, "forward kinds write mask_out, data gradients read mask_in");
  SRL_CHECK_ARG(kind == H2C_D2 || (a.out_scale && a.bound_in && a.bound_w), "h2 outputs need out_scale and the bound's factors");
  SRL_CHECK_ARG((kind != H2C_D2 && kind != H2C_D3) || a.act == 0, "data gradients take no activation (act = 0)");
  srl_count_dispatch(SRL_DISP_H2, 1, kind, fwd ? 3 : 2);
  switch (kind) {
    case H2C_F2: h2conv_launch<H2C_F2, 3>(st, a); break;
    case H2C_F3: h2conv_launch<H2C_F3, 3>(st, a); break;
    case H2C_D3: h2conv_launch<H2C_D3, 2>(st, a); break;
    default: h2conv_launch<H2C_D2, 2>(st, a); break;
  }
  SRL_LAUNCH_CHECK();
  return 0;
}

static constexpr int kWgradGrid = 256;
extern "C" int64_t srl_h2_wgrad_workspace(int32_t kind) {
  const int K = kind == H2W_C2 ? 16 * 32 : 9 * 64;
  return (int64_t)kWgradGrid * (64 * K + 64);
}

extern "C" int srl_h2_wgrad(void* stream, int32_t kind, const void* x, const void* dz, const float* sx, const float* sz, int64_t n,
                            float* workspace, float* gw, float* gb) {
  SRL_CHECK_ARG(x && dz && sx && sz && workspace && gw && n >= 0 && (kind == H2W_C2 || kind == H2W_C3), "bad argument");
  SRL_CHECK_ARG(n * 20 * 20 * 32 * 4 < 0x7fffffffL, "tensors of 2 GiB and more: split the batch");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  H2WgradArgs a{};
  a.x = x; a.dz = dz; a.sx = sx; a.sz = sz; a.n = n; a.slabs = workspace;
  const int grid = n < kWgradGrid ? (int)n : kWgradGrid;
  const int K = kind == H2W_C2 ? 16 * 32 : 9 * 64, per = 64 * K + 64;
  srl_count_dispatch(SRL_DISP_H2, 2, kind, kind == H2W_C2 ? 2 : 3);
  if (kind == H2W_C2) h2wgrad_launch<H2W_C2, 2>(st, a, grid);
  else h2wgrad_launch<H2W_C3, 3>(st, a, grid);
  hipLaunchKernelGGL(h2_wgrad_reduce_kernel, dim3((unsigned)srl_ceil_div(per, 64)), dim3(256), 0, st, workspace, grid, per, 64 * K, gw, gb);
  SRL_LAUNCH_CHECK();
  return 0;
}

// The dense product with its reduction split over `ksplits` workgroups per tile: slab s of out ([ksplits][M][NC] float32) receives
// the raw partial sums of k-range s; bias, activation, masks and the h2 output format are the consumer's (srl_ln_heads_fwd adds the
// slabs while it reads).  For row counts that leave most CUs without a tile: the Linear forward of an inference batch.
extern "C" int srl_h2_gemm_splitk(void* stream, const srl_h2_gemm_desc* d, int32_t ksplits, int32_t wide) {
  SRL_CHECK_ARG(d && d->x && d->w && d->sx && d->sw && d->out && d->M >= 0 && d->NC >= 128 && d->NC % 128 == 0 && d->K > 0 && d->K % 32 == 0,
                "null tensor / NC not a multiple of 128 / K not a multiple of 32");
  SRL_CHECK_ARG(!d->out_h2 && !d->bias && !d->act && !d->mask_out && !d->mask_in && !d->out_absmax, "split products write raw float32 partial sums");
  SRL_CHECK_ARG(ksplits >= 1 && ksplits <= d->K / 32, "1 <= ksplits <= K / 32");
  SRL_CHECK_ARG(d->M * (int64_t)d->K * 4 < 0xffffffffL && d->M * (int64_t)d->NC * 4 < 0xffffffffL, "operands of 4 GiB and more: split the rows");
  if (d->M == 0) return 0;
  H2Args a{};
  a.x = d->x; a.w = d->w; a.sx = d->sx; a.sw = d->sw; a.M = d->M; a.NC = d->NC; a.nk = d->K / 32;
  a.w_row_bytes = (uint32_t)d->K * 4u; a.x_row_bytes = (uint32_t)d->K * 4u;
  a.out_fmt = H2O_F32; a.out = d->out; a.out_row_bytes = (uint32_t)d->NC * 4u;
  a.ksplits = ksplits; a.slab_bytes = d->M * (int64_t)d->NC * 4;
  // wide: 256 channels per workgroup (a third less staged per multiply-add; NC a multiple of 256) on the two-stage ring -- for
  // row counts that fill the chip even so (the training chunk: 64 row tiles x 2 channel tiles x 2 k-ranges = 256 workgroups)
  SRL_CHECK_ARG(!wide || d->NC % 256 == 0, "wide tiles: NC a multiple of 256");
  srl_count_dispatch(SRL_DISP_H2, 3, wide ? 8 : 4, wide ? 2 : 3);
  const int rc = wide ? h2gemm_launch<8, H2X_DENSE, 2, false>((hipStream_t)stream, a) : h2gemm_launch<4, H2X_DENSE, 3>((hipStream_t)stream, a);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_h2_gemm(void* stream, const srl_h2_gemm_desc* d) {
  SRL_CHECK_ARG(d && d->x && d->w && d->sx && d->sw && d->out && d->M >= 0 && d->NC > 0 && d->NC % 32 == 0 && d->K > 0 && d->K % 32 == 0,
                "null tensor / NC, K not multiples of 32");
  SRL_CHECK_ARG(!d->out_h2 || (d->out_scale && d->bound_in && d->bound_w), "h2 output needs out_scale and the bound's factors");
  SRL_CHECK_ARG(d->M * (int64_t)d->K * 4 < 0xffffffffL && d->M * (int64_t)d->NC * 4 < 0xffffffffL, "operands of 4 GiB and more: split the rows");
  if (d->M == 0) return 0;
  H2Args a{};
  a.x = d->x; a.w = d->w; a.sx = d->sx; a.sw = d->sw; a.M = d->M; a.NC = d->NC; a.nk = d->K / 32;
  a.w_row_bytes = (uint32_t)d->K * 4u; a.x_row_bytes = (uint32_t)d->K * 4u;
  a.bias = d->bias; a.act = d->act; a.out_fmt = d->out_h2 ? H2O_H2P : H2O_F32; a.out = d->out; a.out_row_bytes = (uint32_t)d->NC * 4u;
  a.out_scale = d->out_scale; a.bound_in = d->bound_in; a.bound_w = d->bound_w; a.bound_b = d->bound_b; a.out_absmax = d->out_absmax;
  a.mask_out = d->mask_out; a.mask_in = d->mask_in; a.mask_in_h2 = d->mask_in_h2order;
  // a wide output over a short reduction (the Linear's data gradient: 3136 channels, K = 512): 256 channels per workgroup, both
  // k-halves in every wavefront -- a third less staged per multiply-add, on a two-stage ring (240 -> 220 us per 16 384 rows)
  static const bool wide_on = [] { const char* e = getenv("SRL_H2GEMM_WIDE"); return !(e && e[0] == '0'); }();
  const bool wide = wide_on && d->NC >= 1024 && d->K <= 1024;
  static const bool half_cnt = [] { const char* e = getenv("SRL_H2GEMM_HALF"); return e && e[0] == '1'; }();
  static const int dbg = [] { const char* e = getenv("SRL_H2G_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  int rc;
  const auto count_old = [&] { srl_count_dispatch(SRL_DISP_H2, 3, wide ? 8 : (d->NC >= 128 ? 4 : 2), wide && !half_cnt ? 2 : 3); };
  // (two 4-wavefront workgroups of 128 x 256 per CU on half k-steps instead of one 8-wavefront workgroup of 256 x 256: h2gemm.h HALF)
  // Alone 209.5 -> 192.3 us per 16 384 rows of the Linear's data gradient (same box; its leave-outs then overlap: no DMA 138, no
  // MFMAs 149, no stores 143) -- but inside the update, beside the three other row-chunk pipelines, the update gets SLOWER (91.1 /
  // 91.6 ms with the 8-wavefront kernel, 92.4 / 92.4 with this one, alternating runs on one box): the smaller tile stages 1.5 x the
  // bytes per multiply-add (613 against 452 MB of counted traffic per launch) and that is what the neighbours compete for.  Opt-in
  // (SRL_H2GEMM_HALF=1) for the record.
  static const bool half_on = [] { const char* e = getenv("SRL_H2GEMM_HALF"); return e && e[0] == '1'; }();
  // round 6: the wide product as a PERSISTENT kernel (h2gemmp.h: one workgroup per CU walks its tiles, the ring never drains, a
  // tile's stores are not waited for) -- same operands, same piece products in the same order: bit-identical output.  207 -> 191 us
  // per 16 384 rows of the Linear's data gradient on one box, launched alone.
  // Default: the LARGE products only.  For the Atari data gradient the persistent kernel wins alone (one pipeline) and not in the
  // update: four row-chunk pipelines share the chip there and fill each other's idle CUs -- alternating runs on two boxes 86.6 / 86.4
  // (one tile per workgroup) against 89.1 / 86.4, and 85.1 / 84.9 / 85.1 against 84.9 / 88.3 / 84.6 ms per update: no gain, and two of
  // five ring-fed runs 3.5 ms slower (a kernel that holds every CU's whole LDS for the length of the launch lets nobody else's
  // small kernels in).  SRL_H2GEMM_P=1: also there (the per-kernel table of DESIGN section 4 names both); SRL_H2GEMM_P=0: nowhere.
  static const int pers_mode = [] { const char* e = getenv("SRL_H2GEMM_P"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : 2; }();
  // ... and every product large enough that a 256 x 256 tile still fills the chip several times over (the football tower's layers:
  // 51 200 rows x 704 ... 22 528 channels): a third fewer operand bytes staged per multiply-add than the 256 x 128 tiles, 4-13 %
  // faster at those shapes (scripts/h2r6_probe.hip)
  // (chosen from the layer's widths and a row floor only: the pieces an encoder's rows are cut into must not change kernels --
  // `test_config4_football_per_gpu_size` holds an update to its own result under another cut)
  const bool big = d->M >= 4096 && d->NC >= 2048 && d->K >= 1024;
  if (((wide && pers_mode == 1) || (big && pers_mode >= 1)) && !half_on && !dbg) {
    srl_count_dispatch(SRL_DISP_H2, 5, 8, H2P_NSLOT);
    rc = h2gemmp_launch<0>((hipStream_t)stream, a);
    SRL_CHECK_ARG(rc == 0, "grid too large / rows of 8 MiB and more");
    SRL_LAUNCH_CHECK();
    return 0;
  }
  count_old();
  if (wide && half_on) rc = h2gemm_launch<8, H2X_DENSE, 3, false, true>((hipStream_t)stream, a);
  else if (wide) rc = h2gemm_launch<8, H2X_DENSE, 2, false>((hipStream_t)stream, a);
  else if (d->NC >= 128) rc = h2gemm_launch<4, H2X_DENSE, 3>((hipStream_t)stream, a);
  else rc = h2gemm_launch<2, H2X_DENSE, 3>((hipStream_t)stream, a);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  return 0;
}

// The dense weight gradient over h2p rows (h2tn.h): slabs of the row ranges, then their sum.
extern "C" int64_t srl_h2_wgrad_dense_workspace(int64_t M, int32_t NA, int32_t NB) {
  if (M <= 0 || NA <= 0 || NB <= 0) return 0;
  int splits; int64_t rows;
  h2tn_plan(M, ((NA + 255) / 256) * ((NB + 255) / 256), &splits, &rows);
  return (int64_t)splits * NA * NB;
}

extern "C" int srl_h2_wgrad_dense(void* stream, const void* a, const void* b, const float* sa, const float* sb, int64_t M, int32_t NA,
                                  int32_t NB, int64_t a_row_bytes, int64_t b_row_bytes, float* workspace, float* gw, int32_t accumulate) {
  SRL_CHECK_ARG(a && b && sa && sb && workspace && gw && M >= 0 && NA > 0 && NB > 0 && NA % 32 == 0 && NB % 32 == 0,
                "null tensor / NA, NB not multiples of 32");
  SRL_CHECK_ARG(a_row_bytes >= (int64_t)NA * 4 && b_row_bytes >= (int64_t)NB * 4 && a_row_bytes % 16 == 0 && b_row_bytes % 16 == 0,
                "row pitches: at least 4 bytes per channel, multiples of 16");
  SRL_CHECK_ARG(16 * a_row_bytes < 0x7fffffffL && 16 * b_row_bytes < 0x7fffffffL, "rows of 128 MiB and more");
  if (M == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  H2TnArgs g{};
  g.a = a; g.b = b; g.sa = sa; g.sb = sb; g.M = M; g.NA = NA; g.NB = NB; g.a_row_bytes = a_row_bytes; g.b_row_bytes = b_row_bytes;
  h2tn_plan(M, ((NA + 255) / 256) * ((NB + 255) / 256), &g.splits, &g.rows_per_split);
  g.slabs = g.splits == 1 && !accumulate ? gw : workspace;   // (one range, nothing to add to: the slab IS the result)
  srl_count_dispatch(SRL_DISP_H2, 4, 8, H2TN_NSLOT);
  SRL_CHECK_ARG(h2tn_launch<0>(st, g) == 0, "grid too large");
  if (g.slabs != gw) srlgemm::reduce_slabs(st, workspace, g.splits, 1L, (long)NA, (long)NB, gw, (long)NB, 0L, accumulate ? 1 : 0);
  SRL_LAUNCH_CHECK();
  return 0;
}
