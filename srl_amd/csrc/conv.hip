// Implicit-GEMM convolution entry points (padding 0, square or rectangular kernels, any stride):
// forward / weight gradient / data gradient on NHWC activations, and the first convolution on the NCHW
// observation with the whole-observation LayerNorm fused into the operand gather.  No patch matrix is ever
// written to memory; the contraction runs on the FP32 MFMA kernel of gemm_core.h, except the first layer on uint8
// channels-last frames, which runs exactly on the bf16 matrix cores (obs_bf16.h).
#include <stdlib.h>

#include "gemm_core.h"
#include "gemm_bf16x3.h"
#include "obs_bf16.h"
#include "obs_h2.h"

using namespace srlgemm;

// First-layer block kernels (obs_h2.h): XCD-contiguous numbering of the (position block, sample range) workgroups, so that the
// blocks an XCD works on are neighbours in the frame and share window rows / columns in ITS L2.  Measured (scripts/obs_xcd_traffic.sh,
// scripts/ab_bench.sh): the forward's fetched bytes per launch fall by a third (494 -> 326 counter MB), the weight gradient's by a
// sixth (872 -> 729) -- and the update gets 0.4 ms SLOWER (87.9 / 88.6 -> 88.3 / 89.0 ms, alternating runs on one box): these
// kernels are bound by their vector work, not by the frames' bytes.  Opt-in (SRL_OBS_XCD=1).
static int obs_xcd_order() {
  static const int v = [] { const char* e = getenv("SRL_OBS_XCD"); return (e && e[0] == '1') ? 1 : 0; }();
  return v;
}

namespace {

constexpr int K3 = 16;  // k-step depth of the bf16x3 kernels (48 KB of LDS per 128x128 workgroup: three per CU)

inline int conv_out(int in, int k, int s) { return (in - k) / s + 1; }

OutDesc plain_out(float* out, long ld) {
  OutDesc o{};
  o.out = out;
  o.ldo = ld;
  o.f_img = o.f_line = make_fastdiv(1);
  return o;
}

bool fits31(long v) { return v >= 0 && v < (1L << 31); }

// NHWC patch rows: r = (n, oh, ow) -> x[n, oh*S, ow*S, 0];  c = (kh, kw, ci) -> kh*W*C + kw*C + ci
SrcDesc conv_patch_src(const float* x, const srl_conv_desc* d, int OH, int OW) {
  SrcDesc s = plain_src(x, 0);
  s.f_img = make_fastdiv((uint32_t)(OH * OW));
  s.f_line = make_fastdiv((uint32_t)OW);
  s.img_stride = d->H * d->W * d->Cin;
  s.y_stride = d->stride * d->W * d->Cin;
  s.x_stride = d->stride * d->Cin;
  s.f_inner = make_fastdiv((uint32_t)(d->KW * d->Cin));
  s.k1_stride = d->W * d->Cin;
  return s;
}

// Observation patch rows with LayerNorm: r = (n, oh, ow).
//   planar (NCHW):        c = (ci, kh, kw) -> ci*H*W + kh*W + kw
//   channels-last (NHWC): c = (kh, kw, ci) -> kh*W*C + kw*C + ci   (e.g. a space-to-depth'd frame stack)
SrcDesc obs_patch_src(const void* obs, int is_u8, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, const srl_conv_desc* d, int rows_per_img, int OW, int channels_last) {
  SrcDesc s = plain_src(nullptr, 0);
  s.base = obs;
  s.f_img = make_fastdiv((uint32_t)rows_per_img);
  s.f_line = make_fastdiv((uint32_t)OW);
  s.img_stride = d->Cin * d->H * d->W;
  if (channels_last) {
    s.y_stride = d->stride * d->W * d->Cin;
    s.x_stride = d->stride * d->Cin;
    s.f_inner = make_fastdiv((uint32_t)(d->KW * d->Cin));
    s.f_tap = make_fastdiv((uint32_t)(d->KW * d->Cin));  // kh' = 0, kw' = offset inside the (kw, ci) run
    s.k1_stride = d->W * d->Cin;
    s.k2_stride = 0;
  } else {
    s.y_stride = d->stride * d->W;
    s.x_stride = d->stride;
    s.f_inner = make_fastdiv((uint32_t)(d->KH * d->KW));
    s.f_tap = make_fastdiv((uint32_t)d->KW);
    s.k1_stride = d->H * d->W;
    s.k2_stride = d->W;
  }
  s.mean = mean; s.rstd = rstd; s.gamma = gamma; s.beta = beta;
  s.is_u8 = is_u8;
  s.affine = gamma != nullptr;
  return s;
}

bool obs_geometry_ok(const srl_conv_desc* d, int channels_last) {
  // 4-wide gathers must stay aligned: every stride that enters an address is a multiple of 4
  if (channels_last) return d->Cin % 4 == 0;
  return d->KW % 4 == 0 && d->W % 4 == 0 && d->stride % 4 == 0 && (d->H * d->W) % 4 == 0;
}

// (k, pos) -> index into the observation-shaped tables, for both layouts
struct ObsIndex {
  int Cin, H, W, KH, KW, S, OW, cl;
  __host__ __device__ void split_k(int k, int& ci, int& kh, int& kw) const {
    if (cl) { ci = k % Cin; kw = (k / Cin) % KW; kh = k / (Cin * KW); }
    else { kw = k % KW; kh = (k / KW) % KH; ci = k / (KW * KH); }
  }
  __host__ __device__ int k_of(int ci, int kh, int kw) const {
    return cl ? (kh * KW + kw) * Cin + ci : (ci * KH + kh) * KW + kw;
  }
  __host__ __device__ int p_of(int ci, int y, int x) const { return cl ? (y * W + x) * Cin + ci : (ci * H + y) * W + x; }
  __host__ __device__ void split_p(int p, int& ci, int& y, int& x) const {
    if (cl) { ci = p % Cin; x = (p / Cin) % W; y = p / (Cin * W); }
    else { x = p % W; y = (p / W) % H; ci = p / (W * H); }
  }
};

int want_split(long rows, long tiles, long batch) {
  // a batched problem (one GEMM per output position) wants ~4096 workgroups, a single weight gradient ~1024
  long want = (batch > 1 ? 4096 : 1024) / (tiles * batch > 0 ? tiles * batch : 1);
  if (want < 1) want = 1;
  const long cap = rows / 2048 > 1 ? rows / 2048 : 1;
  return (int)(want < cap ? want : cap);
}

// per parity class (ph, pw) of the data gradient: taps th x tw, row grid ra x rb, offset of its weight block
struct DgradClass {
  int ph, pw, th, tw, ra, rb;
  long w_off;
};

int dgrad_classes(const srl_conv_desc* d, DgradClass* out /* stride*stride entries */) {
  int n = 0;
  long off = 0;
  for (int ph = 0; ph < d->stride; ++ph)
    for (int pw = 0; pw < d->stride; ++pw) {
      DgradClass c;
      c.ph = ph; c.pw = pw;
      c.th = ph < d->KH ? (d->KH - ph + d->stride - 1) / d->stride : 0;
      c.tw = pw < d->KW ? (d->KW - pw + d->stride - 1) / d->stride : 0;
      c.ra = ph < d->H ? (d->H - ph + d->stride - 1) / d->stride : 0;
      c.rb = pw < d->W ? (d->W - pw + d->stride - 1) / d->stride : 0;
      c.w_off = off;
      off += (long)c.th * c.tw * d->Cout * d->Cin;
      out[n++] = c;
    }
  return n;
}

// wt[(jh, jw, o) * ld + ci] = w[o][ph + jh*S][pw + jw*S][ci]: one class's block; ld = Cin when the classes are stored
// one after the other, ncls * Cin when they sit side by side along N (wt then points at the class's first column)
__global__ __launch_bounds__(256) void dgrad_repack_kernel(const float* w, float* wt, int Cout, int KH, int KW, int Cin,
                                                           int S, int ph, int pw, int th, int tw, int ld) {
  const int total = th * tw * Cout * Cin;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int ci = e % Cin;
    const int o = (e / Cin) % Cout;
    const int t = e / (Cin * Cout);
    const int jh = t / tw, jw = t % tw;
    wt[(long)(e / Cin) * ld + ci] = w[((o * KH + ph + jh * S) * KW + pw + jw * S) * Cin + ci];
  }
}

// every parity class has the same tap and row grids (stride divides the kernel and the image): the classes then share
// the A operand (the taps of dz that reach a pixel do not depend on the class) and are packed along N
bool dgrad_uniform(const DgradClass* cls, int nc) {
  bool u = nc > 1 && cls[0].th * cls[0].tw > 0 && cls[0].ra > 0 && cls[0].rb > 0;
  for (int i = 1; i < nc; ++i)
    u = u && cls[i].th == cls[0].th && cls[i].tw == cls[0].tw && cls[i].ra == cls[0].ra && cls[i].rb == cls[0].rb;
  return u;
}

// ---- finalisation of the first-layer backward from the per-position products Q and column sums R --------------------
// dw[o,k] += sum_pos gamma[p(pos,k)] * Q[pos,o,k] + beta[p(pos,k)] * R[pos,o];   db[o] += sum_pos R[pos,o]
// 32 outputs x 8 position-lanes per workgroup: one thread per output would walk the P positions as a serial chain of
// loads on 32 workgroups
constexpr int kDwEL = 32, kDwZL = 8;
// C[pos,o] (zeros on the float32 path): the mean correction the bf16 path leaves outside its contraction (obs_bf16.h),
// Q_true = Q - C
__global__ __launch_bounds__(256) void obs_dw_kernel(const float* Q, const float* R, const float* C, const float* gamma,
                                                     const float* beta, int P, int Cout, ObsIndex ix, float* dw,
                                                     float* db) {
  __shared__ float red_a[256], red_r[256];
  const int Kp = ix.Cin * ix.KH * ix.KW;
  const int ex = threadIdx.x % kDwEL, pz = threadIdx.x / kDwEL;
  const int e = blockIdx.x * kDwEL + ex;
  const bool live = e < Cout * Kp;
  const int k = live ? e % Kp : 0, o = live ? e / Kp : 0;
  int ci, kh, kw;
  ix.split_k(k, ci, kh, kw);
  float acc = 0.f, rsum = 0.f;
  if (live) {
#pragma unroll 4
    for (int pos = pz; pos < P; pos += kDwZL) {
      const int oh = pos / ix.OW, ow = pos % ix.OW;
      const int p = ix.p_of(ci, oh * ix.S + kh, ow * ix.S + kw);
      const float r = R[pos * Cout + o];
      acc += gamma[p] * (Q[((long)pos * Cout + o) * Kp + k] - C[pos * Cout + o]) + beta[p] * r;
      rsum += r;
    }
  }
  red_a[threadIdx.x] = acc;
  red_r[threadIdx.x] = rsum;
  __syncthreads();
  if (pz == 0 && live) {
    for (int z = 1; z < kDwZL; ++z) acc += red_a[z * kDwEL + ex], rsum += red_r[z * kDwEL + ex];
    dw[e] += acc;
    if (k == 0) db[o] += rsum;
  }
}

// dgamma[p] += sum_{(pos,k) -> p} sum_o w[o,k] Q[pos,o,k];   dbeta[p] += sum_{(pos,k) -> p} sum_o w[o,k] R[pos,o]
__global__ __launch_bounds__(256) void obs_affine_kernel(const float* Q, const float* R, const float* C, const float* w, int OH,
                                                         int Cout, ObsIndex ix, float* dgamma, float* dbeta) {
  const int Kp = ix.Cin * ix.KH * ix.KW;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= ix.Cin * ix.H * ix.W) return;
  int ci, y, x;
  ix.split_p(p, ci, y, x);
  const int S = ix.S;
  float ag = 0.f, ab = 0.f;
  for (int kh = y % S; kh < ix.KH && kh <= y; kh += S) {
    const int oh = (y - kh) / S;
    if (oh >= OH) continue;
    for (int kw = x % S; kw < ix.KW && kw <= x; kw += S) {
      const int ow = (x - kw) / S;
      if (ow >= ix.OW) continue;
      const int pos = oh * ix.OW + ow;
      const int k = ix.k_of(ci, kh, kw);
      for (int o = 0; o < Cout; ++o) {
        const float wv = w[o * Kp + k];
        ag += wv * (Q[((long)pos * Cout + o) * Kp + k] - C[pos * Cout + o]);
        ab += wv * R[pos * Cout + o];
      }
    }
  }
  dgamma[p] += ag;
  dbeta[p] += ab;
}

// ---- position-batched first layer: fold the LayerNorm affine into per-position weights ---------------------------------
// wg[pos][o][k] = w[o][k] * gamma[p(pos,k)];   b2[pos][o] = bias[o] + sum_k w[o][k] * beta[p(pos,k)]
__global__ __launch_bounds__(256) void obs_fold_affine_kernel(const float* w, const float* bias, const float* gamma,
                                                              const float* beta, int P, int Cout, ObsIndex ix, float* wg,
                                                              float* b2) {
  __shared__ float red[4];
  const int Kp = ix.Cin * ix.KH * ix.KW;
  const int pos = blockIdx.x / Cout, o = blockIdx.x % Cout;
  const int oh = pos / ix.OW, ow = pos % ix.OW;
  float acc = 0.f;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    int ci, kh, kw;
    ix.split_k(k, ci, kh, kw);
    const int p = ix.p_of(ci, oh * ix.S + kh, ow * ix.S + kw);
    const float wv = w[o * Kp + k];
    wg[((long)pos * Cout + o) * Kp + k] = wv * gamma[p];
    acc += wv * beta[p];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) b2[pos * Cout + o] = (bias ? bias[o] : 0.f) + red[0] + red[1] + red[2] + red[3];
}

// uint8 channels-last first layer on the bf16 matrix cores (obs_bf16.h): geometry it is written for
bool obs_bf16_ok(const srl_conv_desc* d, int is_u8, int channels_last, const void* obs) {
  const char* env = getenv("SRL_OBS_BF16");  // "0": force the float32 kernels (A/B timing, cross-check in the tests)
  const bool off = env && env[0] == '0';
  if (off || !is_u8 || !channels_last || d->Cout != srlobs::kCout) return false;
  const long run = (long)d->KW * d->Cin, Kp = run * d->KH;
  if (Kp != 256 || run < 16 || (run & (run - 1)) != 0) return false;
  const long img = (long)d->H * d->W * d->Cin;
  return img % 16 == 0 && ((long)d->stride * d->Cin) % 16 == 0 && aligned16(obs) && d->n >= 32;
}

srlobs::ObsGeom obs_geom(const srl_conv_desc* d, const void* obs, const float* mean, const float* rstd, int OW,
                         const int32_t* row_index = nullptr) {
  srlobs::ObsGeom g{};
  g.row_index = row_index;
  g.frames = static_cast<const uint8_t*>(obs);
  g.img_stride = (long)d->H * d->W * d->Cin;
  g.W = d->W; g.Cin = d->Cin; g.OW = OW; g.stride = d->stride;
  const int run = d->KW * d->Cin;
  g.run_shift = 31 - __builtin_clz((unsigned)run);
  g.run_stride = d->W * d->Cin;
  g.mean = mean; g.rstd = rstd; g.n = d->n;
  return g;
}

// sample ranges per output position: enough workgroups for a few rounds over the chip, >= 512 samples each
int obs_bf16_split(long n, int P, int per_cu) {
  const long slots = 256L * per_cu;
  long want = (4 * slots + P - 1) / P;
  static const long rows_min = [] { const char* e = getenv("SRL_OBS_SPLIT_ROWS"); return e ? atol(e) : 512L; }();  // tuning knob
  if (rows_min < 512) want = n / rows_min;
  const long cap = n / rows_min > 1 ? n / rows_min : 1;
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

int check_desc(const srl_conv_desc* d) {
  if (!d) return -1;
  if (d->n < 0 || d->H < 1 || d->W < 1 || d->Cin < 1 || d->Cout < 1 || d->KH < 1 || d->KW < 1 || d->stride < 1) return -1;
  if (d->KH > d->H || d->KW > d->W) return -1;
  return 0;
}

}  // namespace

extern "C" int srl_conv2d_supported(const srl_conv_desc* d, int first_layer) {
  if (check_desc(d) != 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  // one image must be addressable with 32-bit offsets; any NUMBER of images is fine (the entry points walk a large batch
  // in runs of images_per_launch: a 64-bit base per run, 32-bit offsets inside it)
  if (!fits31(4L * d->H * d->W * d->Cin) || !fits31(4L * OH * OW * d->Cout) || d->n < 0) return 0;
  if (first_layer) return obs_geometry_ok(d, first_layer == 2) ? 1 : 0;
  return (d->Cin % 4 == 0 && d->Cout % 4 == 0) ? 1 : 0;
}

// The kernels address every tensor of a launch with 32-bit offsets (bytes in the gathers, elements in the epilogue).  A batch
// whose activations exceed that -- 2 GiB on a 288 GB part is one 16 384-row chunk of a 128 x 128 x 16 feature map -- is walked
// in runs of this many images by the entry points below (whole 256-image groups, so that the position-grouped tiles of the
// data gradients stay full): a 64-bit base per run, the same kernels inside it.
static long images_per_launch(const srl_conv_desc* d, int in_esize) {
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long in_b = (long)d->H * d->W * d->Cin * in_esize, in_f = 4L * d->H * d->W * d->Cin, out_b = 4L * OH * OW * d->Cout;
  long big = in_b > out_b ? in_b : out_b;
  big = in_f > big ? in_f : big;  // the data gradient writes the input's shape in float32
  long lim = ((1L << 31) - 1) / big;
  const char* e = getenv("SRL_CONV_RUN_IMAGES");  // tests: force short runs
  if (e && atol(e) > 0 && atol(e) < lim) lim = atol(e);
  if (lim >= 512) lim &= ~255L;
  if (lim < 1) lim = 1;
  // equal runs rather than full ones and a remainder: a last run of a handful of images would fall below the sizes the byte
  // kernels (and a slot index) need, which the caller checked for the whole batch
  if (d->n > lim) {
    const long nruns = srl_ceil_div(d->n, lim);
    long r = srl_ceil_div(d->n, nruns);
    if (r >= 512) r = (r + 255) & ~255L;
    if (r < lim) lim = r;
  }
  return lim;
}

// Order of the 16-deep k-steps of a forward convolution (k = (kh, kw, c), c fastest).  Every tile re-reads its input
// patch region once per tap that reaches a pixel, and with ~100 tiles in flight per XCD the regions of all of them (10 MB
// for conv2 of the Atari stack) do not stay in the 4 MB L2 between two taps in ascending order.  Taps that read the SAME
// pixels are therefore made neighbours: for stride S the taps of one parity class (kh % S, kw % S) read one pixel class, so
// the classes are visited one after the other, the taps of a class back to back; inside, the two 16-channel blocks that
// share a 128-byte line stay adjacent, and for more than 32 channels the line index is the outermost key (each phase then
// touches 1 / lines-per-pixel of the region).  SRL_KPERM=0: ascending order (A/B switch).
static void fwd_kstep_order(const srl_conv_desc* d, GemmArgs* g) {
  static const bool on = [] { const char* e = getenv("SRL_KPERM"); return !(e && e[0] == '0'); }();
  const int C = (int)d->Cin, n = (int)(d->KH * d->KW * C / 16);
  if (!on || C % 16 != 0 || n < 2) return;
  g->kp_s = (int)d->stride; g->kp_kh = (int)d->KH; g->kp_kw = (int)d->KW; g->kp_cb = C / 16;
}

// 192 x 64 forward tiles (2 x 2 wavefronts of 96 x 32): 49 KB of LDS, 3 workgroups per CU where 256 x 64 fits 2 -- conv2 /
// conv3 forward 0.75 / 0.42 -> 0.73 / 0.40 ms (same box).  SRL_TILE192=0: the 256 x 64 tiles (A/B switch)
static bool tile192() {
  static const bool on = [] { const char* e = getenv("SRL_TILE192"); return !(e && e[0] == '0'); }();
  return on;
}

static int conv2d_nhwc_fwd_run(void* stream, const srl_conv_desc* d, const float* x, const float* w,
                               const float* bias, float* y, const float* x_absmax, const float* w_absmax,
                               float* y_absmax, uint32_t* y_mask, int w_presplit) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0), "unsupported geometry (needs Cin, Cout multiples of 4, < 2^31 elements)");
  SRL_CHECK_ARG(x && w && y && aligned16(x) && aligned16(w), "null / unaligned tensor");
  SRL_CHECK_ARG(!y_mask || (d->act == 1 && d->Cout % 32 == 0), "y_mask: ReLU layers with Cout a multiple of 32");
  if (d->n == 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long Kp = (long)d->KH * d->KW * d->Cin;
  GemmArgs g{};
  g.M = d->n * OH * OW; g.N = d->Cout; g.K = Kp;
  g.a = conv_patch_src(x, d, OH, OW);
  g.b = plain_src(w, Kp);
  g.o = plain_out(y, d->Cout);
  g.bias = bias; g.act = d->act;
  g.k_per_split = srl_ceil_div(Kp, BK) * BK;
  g.vec_a = 1; g.vec_b = 1;
  g.range_a = x_absmax; g.range_b = w_absmax; g.out_absmax = y_absmax;
  g.mask_out = y_mask;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  const bool x3 = use_bf16x3() && Kp >= 64;  // bf16 matrix cores, three exact pieces per float32 operand
  if (x3) fwd_kstep_order(d, &g);
  SRL_CHECK_ARG(!w_presplit || (x3 && x_absmax && w_absmax && use_f16x2() && d->Cout > 32),
                "w_presplit: only layers that take the two-piece kernel");
  g.b_presplit = w_presplit;
  if (x3 && x_absmax && w_absmax && use_f16x2() && d->Cout > 32) {
    // both operands' ranges are known: two f16 pieces each, three products (gemm_bf16x3.h, NP == 2)
    rc = d->Cout > 64 ? launch3<128, 128, 2, 2, false, false, SRC_CONV, SRC_PLAIN, K3, 2>(st, g, 1, 1)
                      : (tile192() ? launch3<192, 64, 2, 2, false, false, SRC_CONV, SRC_PLAIN, K3, 2>(st, g, 1, 1)
                                   : launch3<256, 64, 4, 1, false, false, SRC_CONV, SRC_PLAIN, K3, 2>(st, g, 1, 1));
  } else
  if (d->Cout > 64) rc = x3 ? launch3<128, 128, 2, 2, false, false, SRC_CONV, SRC_PLAIN, K3>(st, g, 1, 1)
                            : launch<128, 128, 2, 2, false, false, SRC_CONV, SRC_PLAIN>(st, g, 1, 1);
  else if (d->Cout > 32) rc = x3 ? (tile192() ? launch3<192, 64, 2, 2, false, false, SRC_CONV, SRC_PLAIN, K3>(st, g, 1, 1)
                                                : launch3<256, 64, 4, 1, false, false, SRC_CONV, SRC_PLAIN, K3>(st, g, 1, 1))
                                 : launch<256, 64, 4, 1, false, false, SRC_CONV, SRC_PLAIN>(st, g, 1, 1);
  else rc = launch<256, 32, 4, 1, false, false, SRC_CONV, SRC_PLAIN>(st, g, 1, 1);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_nhwc_fwd(void* stream, const srl_conv_desc* d, const float* x, const float* w,
                                   const float* bias, float* y, const float* x_absmax, const float* w_absmax,
                                   float* y_absmax, uint32_t* y_mask, int w_presplit) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0), "unsupported geometry (needs Cin, Cout multiples of 4)");
  const long run = images_per_launch(d, 4);
  const long in_e = (long)d->H * d->W * d->Cin;
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    const int rc = conv2d_nhwc_fwd_run(stream, &s, x + i0 * in_e, w, bias, y + i0 * out_e, x_absmax, w_absmax, y_absmax,
                                       y_mask ? y_mask + i0 * out_e / 32 : nullptr, w_presplit);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int64_t srl_conv2d_wgrad_workspace(const srl_conv_desc* d) {
  if (check_desc(d) != 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long Kp = (long)d->KH * d->KW * d->Cin;
  const int bm = d->Cout > 32 ? 64 : 32;
  const long tiles = srl_ceil_div(d->Cout, bm) * srl_ceil_div(Kp, 256);
  return (int64_t)want_split(d->n * OH * OW, tiles, 1) * d->Cout * Kp;
}

static int conv2d_nhwc_wgrad_run(void* stream, const srl_conv_desc* d, const float* x, const float* dz, float* dw,
                                 float* workspace, float* dbias, const float* x_absmax, const float* dz_absmax) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0), "unsupported geometry");
  SRL_CHECK_ARG(x && dz && dw && aligned16(x) && aligned16(dz), "null / unaligned tensor");
  if (d->n == 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long Kp = (long)d->KH * d->KW * d->Cin, rows = d->n * OH * OW;
  GemmArgs g{};
  g.M = d->Cout; g.N = Kp; g.K = rows;
  g.a = plain_src(dz, d->Cout);          // A(i = o, k = row) = dz[row*Cout + o]   (k-major)
  g.b = conv_patch_src(x, d, OH, OW);    // B(k = row, j) = patch(row)[j]           (k-major gather)
  // 64-row tiles when there are that many output channels: every patch row is then gathered once per 64
  // (not 32) channels, halving the dominant operand traffic
  const int bm = d->Cout > 32 ? 64 : 32;
  const long tiles = srl_ceil_div(d->Cout, bm) * srl_ceil_div(Kp, 256);
  int split = want_split(rows, tiles, 1);
  if (!workspace) split = 1;
  const int nsplit = plan_split(rows, split, &g.k_per_split);
  if (nsplit > 1) { g.o = plain_out(workspace, Kp); g.slab = (long)d->Cout * Kp; g.accumulate = 0; }
  else { g.o = plain_out(dw, Kp); g.accumulate = 1; }
  g.vec_a = 1; g.vec_b = 1;
  g.a_colsum = dbias;
  hipStream_t st = (hipStream_t)stream;
  int rc;
  g.range_a = dz_absmax; g.range_b = x_absmax;
  if (bm == 64 && use_bf16x3() && x_absmax && dz_absmax && use_f16x2())  // both ranges known: two f16 pieces per operand
    rc = launch3<64, 256, 1, 4, true, true, SRC_PLAIN, SRC_CONV, K3, 2>(st, g, 1, nsplit);
  else
  if (bm == 64) rc = use_bf16x3() ? launch3<64, 256, 1, 4, true, true, SRC_PLAIN, SRC_CONV, K3>(st, g, 1, nsplit)
                                  : launch<64, 256, 1, 4, true, true, SRC_PLAIN, SRC_CONV>(st, g, 1, nsplit);
  else rc = launch<32, 256, 1, 4, true, true, SRC_PLAIN, SRC_CONV>(st, g, 1, nsplit);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  if (nsplit > 1) {
    reduce_slabs(st, workspace, nsplit, 1L, (long)d->Cout, Kp, dw, Kp, 0L, 1);
    SRL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int srl_conv2d_nhwc_wgrad(void* stream, const srl_conv_desc* d, const float* x, const float* dz, float* dw,
                                     float* workspace, float* dbias, const float* x_absmax, const float* dz_absmax) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0), "unsupported geometry");
  const long run = images_per_launch(d, 4);
  const long in_e = (long)d->H * d->W * d->Cin;
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {  // every run adds into dw / dbias
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    const int rc = conv2d_nhwc_wgrad_run(stream, &s, x + i0 * in_e, dz + i0 * out_e, dw, workspace, dbias, x_absmax, dz_absmax);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int64_t srl_conv2d_dgrad_weight_elems(const srl_conv_desc* d) {
  if (check_desc(d) != 0) return 0;
  return (int64_t)d->KH * d->KW * d->Cin * d->Cout;  // the classes partition the taps
}

extern "C" int srl_conv2d_dgrad_repack(void* stream, const srl_conv_desc* d, const float* w, float* wt) {
  SRL_CHECK_ARG(check_desc(d) == 0 && w && wt && d->stride <= 8, "bad descriptor");
  DgradClass cls[64];
  const int nc = dgrad_classes(d, cls);
  const bool packed = dgrad_uniform(cls, nc) && (nc * d->Cin) % 4 == 0;
  for (int i = 0; i < nc; ++i) {
    const DgradClass& c = cls[i];
    const int total = c.th * c.tw * d->Cout * d->Cin;
    if (total == 0) continue;
    hipLaunchKernelGGL(dgrad_repack_kernel, dim3((unsigned)srl_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       packed ? wt + (long)i * d->Cin : wt + c.w_off, d->Cout, d->KH, d->KW, d->Cin, d->stride, c.ph, c.pw,
                       c.th, c.tw, packed ? nc * d->Cin : d->Cin);
  }
  SRL_LAUNCH_CHECK();
  return 0;
}

static int conv2d_nhwc_dgrad_run(void* stream, const srl_conv_desc* d, const float* dz, const float* wt,
                                 const float* x_act, int dact, float* dx, const float* dz_absmax, const float* w_absmax,
                                 float* dx_absmax, const uint32_t* x_mask, int wt_presplit) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0) && d->stride <= 8, "unsupported geometry");
  SRL_CHECK_ARG(!x_mask || (dact == 1 && !x_act && d->Cin % 32 == 0),
                "x_mask: the ReLU derivative, instead of x_act; Cin a multiple of 32");
  SRL_CHECK_ARG(dz && wt && dx && aligned16(dz) && aligned16(wt), "null / unaligned tensor");
  if (d->n == 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  hipStream_t st = (hipStream_t)stream;
  DgradClass cls[64];
  const int nc = dgrad_classes(d, cls);
  // When every parity class has the same tap and row grids (stride divides the kernel and the image), the classes
  // differ only in their weight block and in where their rows land.  They share the A operand (the taps of dz that
  // reach a pixel do not depend on its class), so they are packed side by side along N: ONE GEMM with
  // N = classes * Cin whose column groups land on the classes' pixels -- A is staged once for all of them and the
  // tile is 128 wide instead of 32.
  const bool uniform = dgrad_uniform(cls, nc) && (nc * d->Cin) % 4 == 0;
  for (int i = 0; i < (uniform ? 1 : nc); ++i) {
    const DgradClass& c = cls[i];
    if (c.ra == 0 || c.rb == 0) continue;
    GemmArgs g{};
    g.M = d->n * c.ra * c.rb; g.N = d->Cin; g.K = (long)c.th * c.tw * d->Cout;
    // rows (n, a, b) -> dz[n, a, b, 0]; columns (jh, jw, o) -> -(jh*OW + jw)*Cout + o, valid while inside the image
    SrcDesc a = plain_src(dz, 0);
    a.f_img = make_fastdiv((uint32_t)(c.ra * c.rb));
    a.f_line = make_fastdiv((uint32_t)c.rb);
    a.img_stride = OH * OW * d->Cout;
    a.y_stride = OW * d->Cout;
    a.x_stride = d->Cout;
    a.f_inner = make_fastdiv((uint32_t)d->Cout);
    a.f_tap = make_fastdiv((uint32_t)(c.tw > 0 ? c.tw : 1));
    a.OH = OH; a.OW = OW;
    g.a = a;
    g.b = plain_src(wt + c.w_off, d->Cin);  // B(k, j = ci) k-major
    OutDesc o{};
    o.out = dx + ((long)c.ph * d->W + c.pw) * d->Cin;
    o.rowmap = 1;
    o.f_img = make_fastdiv((uint32_t)(c.ra * c.rb));
    o.f_line = make_fastdiv((uint32_t)c.rb);
    o.img_stride = (long)d->H * d->W * d->Cin;
    o.y_stride = (long)d->stride * d->W * d->Cin;
    o.x_stride = (long)d->stride * d->Cin;
    g.o = o;
    if (x_act && dact) { g.dact_src = x_act + ((long)c.ph * d->W + c.pw) * d->Cin; g.dact = dact; }
    if (x_mask) { g.dact_mask = x_mask; g.mask_off = (uint32_t)(((long)c.ph * d->W + c.pw) * d->Cin); g.dact = 1; }
    const int batch = 1;
    if (uniform) {  // column group (ph, pw) = ph * stride + pw, Cin columns each
      g.N = (long)nc * d->Cin;
      g.b = plain_src(wt, nc * d->Cin);
      g.o.cg_width = d->Cin; g.o.cg_brw = d->stride; g.o.f_cg = make_fastdiv((uint32_t)d->Cin);
      g.o.cg_ystride = (long)d->W * d->Cin; g.o.cg_xstride = d->Cin;
    }
    const long ncols = g.N;
    g.k_per_split = srl_ceil_div(g.K > 0 ? g.K : 1, BK) * BK;
    g.vec_a = 1; g.vec_b = 1;
    // Position-grouped rows: a tile = one pixel position of BM images, so the taps that leave the gradient image are
    // known per tile and their k-steps are skipped (at the borders of a 9x9 image under a 3x3 kernel that is 40 % of
    // the multiply-adds).  Needs whole taps per k-step, a step mask that fits 64 bits, and enough images to fill tiles.
    const int bm = ncols > 64 ? 128 : 256, shift = ncols > 64 ? 7 : 8;
    if (g.K > 0 && d->Cout % BK == 0 && g.K / BK <= 64 && d->n >= 4L * bm) {
      const long groups = srl_ceil_div(d->n, (long)bm);
      g.M = groups * c.ra * c.rb * bm;
      g.a.grp_shift = g.o.grp_shift = shift;
      g.a.n_img = g.o.n_img = (int)d->n;
      SRL_CHECK_ARG(fits31(g.M), "unsupported geometry (too many rows)");
    }
    g.range_a = dz_absmax; g.range_b = w_absmax; g.out_absmax = dx_absmax;
    int rc;
    const bool x3 = use_bf16x3() && d->Cout % K3 == 0;  // the step mask skips whole taps: a k-step must not straddle two
    const bool two_piece = g.K > 0 && x3 && dz_absmax && w_absmax && use_f16x2() && ncols > 32;
    SRL_CHECK_ARG(!wt_presplit || two_piece, "wt_presplit: only layers that take the two-piece kernel");
    g.b_presplit = wt_presplit;
    if (g.K == 0) {  // no tap reaches this class: gradient is zero there (k loop is empty, epilogue writes 0)
      rc = launch<256, 32, 4, 1, false, true, SRC_DGRAD, SRC_PLAIN>(st, g, 1, 1);
    } else if (x3 && dz_absmax && w_absmax && use_f16x2() && ncols > 32) {  // both ranges known: two f16 pieces per operand
      rc = ncols > 64 ? launch3<128, 128, 2, 2, false, true, SRC_DGRAD, SRC_PLAIN, K3, 2>(st, g, batch, 1)
                      : launch3<256, 64, 4, 1, false, true, SRC_DGRAD, SRC_PLAIN, K3, 2>(st, g, batch, 1);
    } else if (ncols > 64) rc = x3 ? launch3<128, 128, 2, 2, false, true, SRC_DGRAD, SRC_PLAIN, K3>(st, g, batch, 1)
                                   : launch<128, 128, 2, 2, false, true, SRC_DGRAD, SRC_PLAIN>(st, g, batch, 1);
    else if (ncols > 32) rc = x3 ? launch3<256, 64, 4, 1, false, true, SRC_DGRAD, SRC_PLAIN, K3>(st, g, batch, 1)
                                 : launch<256, 64, 4, 1, false, true, SRC_DGRAD, SRC_PLAIN>(st, g, batch, 1);
    else rc = launch<256, 32, 4, 1, false, true, SRC_DGRAD, SRC_PLAIN>(st, g, batch, 1);
    SRL_CHECK_ARG(rc == 0, "grid too large");
  }
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_nhwc_dgrad(void* stream, const srl_conv_desc* d, const float* dz, const float* wt,
                                     const float* x_act, int dact, float* dx, const float* dz_absmax, const float* w_absmax,
                                     float* dx_absmax, const uint32_t* x_mask, int wt_presplit) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 0) && d->stride <= 8, "unsupported geometry");
  const long run = images_per_launch(d, 4);
  const long in_e = (long)d->H * d->W * d->Cin;
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    const int rc = conv2d_nhwc_dgrad_run(stream, &s, dz + i0 * out_e, wt, x_act ? x_act + i0 * in_e : nullptr, dact, dx + i0 * in_e,
                                         dz_absmax, w_absmax, dx_absmax, x_mask ? x_mask + i0 * in_e / 32 : nullptr, wt_presplit);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int64_t srl_conv2d_obs_fwd_workspace(const srl_conv_desc* d) {
  if (check_desc(d) != 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  // float32 path: wg [P][Cout][Kp] + b2 [P][Cout]; bf16 path: three bf16 planes (1.5 x wg) + S + b2
  return (int64_t)OH * OW * d->Cout * (2 * (long)d->Cin * d->KH * d->KW + 2) + 64;
}

extern "C" int srl_conv2d_obs_row_index_supported(const srl_conv_desc* d, int is_u8, int channels_last) {
  if (check_desc(d) != 0 || !srl_conv2d_supported(d, channels_last ? 2 : 1)) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  return obs_bf16_ok(d, is_u8, channels_last, nullptr) && ((long)OH * OW * d->Cout) % 4 == 0 ? 1 : 0;
}

// The first layer's block kernel (obs_h2.h) and its per-position folded weights: the geometry it serves, and the fold on its own
// (the weights depend on the parameters only: srl_conv2d_obs_fold_h2 lets the trainer enqueue it before the update's first chunk).
static bool obs_h2_block_geometry(const srl_conv_desc* d) {
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  return d->Cin == 64 && d->KH == 2 && d->KW == 2 && d->stride == 1 && d->Cout == 32 && OH % srlobs::kBlkH == 0 &&
         OW % srlobs::kBlkW == 0 && OH % 2 == 0 && OW % 2 == 0;
}

static void obs_h2_fold(hipStream_t st, const srl_conv_desc* d, const float* gamma, const float* beta, const float* w,
                        const float* bias, float* workspace) {
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride), P = OH * OW;
  const long Kp = (long)d->Cin * d->KH * d->KW;
  uint16_t* wq = reinterpret_cast<uint16_t*>(workspace);
  float* S = workspace + ((long)P * 3 * d->Cout * Kp) / 2;
  float* b2 = S + (long)P * d->Cout;
  float* bound = b2 + (long)P * d->Cout;
  float* winv = workspace + (long)P * d->Cout * Kp;
  const ObsIndex ix{d->Cin, d->H, d->W, d->KH, d->KW, d->stride, OW, 1};
  (void)hipMemsetAsync(bound, 0, sizeof(float), st);
  hipLaunchKernelGGL(srlobs::obs_fold_h2_kernel<ObsIndex>, dim3((unsigned)(P * d->Cout)), dim3(256), 0, st, w, bias, gamma, beta, P, ix,
                     wq, winv, S, b2, bound, sqrtf((float)((long)d->H * d->W * d->Cin)));
}

static int conv2d_obs_fwd_run(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                              const float* mean, const float* rstd, const float* gamma, const float* beta,
                              const float* w, const float* bias, float* y, float* workspace, const int32_t* row_index,
                              float* y_absmax, uint32_t* y_mask, int reuse_folded, uint8_t* y_h2 = nullptr, float* y_scale = nullptr,
                              int ent_order = 0, uint4* records = nullptr) {
  SRL_CHECK_ARG(!y_h2 || (y_scale && y_mask && y_absmax && workspace && obs_bf16_ok(d, is_u8, channels_last, obs)),
                "h2 output: byte kernels only, with y_scale, y_mask, y_absmax and the folded-weights workspace");
  SRL_CHECK_ARG(!y_mask || (d->act == 1 && d->Cout % 32 == 0), "y_mask: ReLU layers with Cout a multiple of 32");
  SRL_CHECK_ARG(row_index == nullptr || (workspace && srl_conv2d_obs_row_index_supported(d, is_u8, channels_last)),
                "row_index is served by the byte kernels only (srl_conv2d_obs_row_index_supported)");
  SRL_CHECK_ARG(srl_conv2d_supported(d, channels_last ? 2 : 1),
                "unsupported geometry (planar: KW, W, stride, H*W multiples of 4; channels-last: Cin multiple of 4)");
  SRL_CHECK_ARG(obs && mean && rstd && gamma && beta && w && (y || y_h2) && aligned16(obs) && aligned16(gamma) && aligned16(beta),
                "null / unaligned tensor");
  if (d->n == 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long Kp = (long)d->Cin * d->KH * d->KW;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs g{};
  g.K = Kp;
  g.act = d->act;
  g.k_per_split = srl_ceil_div(Kp, BK) * BK;
  g.vec_a = 1;
  g.out_absmax = y_absmax;
  g.mask_out = y_mask;
  int rc;
  if (workspace && aligned16(workspace) && obs_bf16_ok(d, is_u8, channels_last, obs)) {
    // bytes x (three exact bf16 planes of the folded weights) on the bf16 matrix cores: obs_bf16.h
    const int P = OH * OW;
    uint16_t* wq = reinterpret_cast<uint16_t*>(workspace);
    float* S = workspace + ((long)P * 3 * d->Cout * Kp) / 2;
    float* b2 = S + (long)P * d->Cout;
    const ObsIndex ix{d->Cin, d->H, d->W, d->KH, d->KW, d->stride, OW, 1};
    float* bound = b2 + (long)P * d->Cout;  // one of the workspace's 64 spare floats: upper bound of |y| from the folded weights
    // h2 output on the Atari geometry: blocks of 2 x 4 positions per workgroup, two f16 weight pieces in registers (obs_h2.h)
    static const bool blocks_on = [] { const char* e = getenv("SRL_OBS_H2BLOCK"); return !(e && e[0] == '0'); }();
    if (y_h2 && blocks_on && ent_order == 2 && obs_h2_block_geometry(d) && d->n * (long)P * 128 < 0x7fffffffL &&
        (records || 4 * (d->n + 32) + (long)P * d->Cout <= (long)P * d->Cout * Kp / 2)) {  // (the records fit behind the two planes)
      float* winv = workspace + (long)P * d->Cout * Kp;  // behind the two f16 planes, inside the room of the three bf16 ones
      if (!reuse_folded) obs_h2_fold(st, d, gamma, beta, w, bias, workspace);
      srlobs::FwdH2Args h{};
      const long n_pad = srl_ceil_div(d->n, (long)srlobs::kTile) * srlobs::kTile;
      // per-sample records of this launch: behind winv, or in the caller's own room when several streams share the folded weights
      uint4* meta = records ? records : reinterpret_cast<uint4*>(winv + (long)P * d->Cout);
      hipLaunchKernelGGL(srlobs::obs_meta_kernel, dim3((unsigned)srl_ceil_div(n_pad, 256L)), dim3(256), 0, st, row_index, mean, rstd,
                         (long)d->n, n_pad, meta);
      h.frames = static_cast<const uint8_t*>(obs); h.img_stride = (long)d->H * d->W * d->Cin; h.meta = meta;
      h.n = d->n; h.wq = reinterpret_cast<const uint4*>(wq); h.winv = winv; h.S = S; h.b2 = b2;
      h.y_h2 = y_h2; h.bound = bound; h.y_scale = y_scale; h.y_mask = y_mask; h.y_absmax = y_absmax;
      h.GW = d->W; h.OW = OW; h.OH = OH; h.P = P; h.act = d->act;
      // one workgroup per CU: every block of positions x as many ranges of the launch's tiles as make that many
      const long nblk = P / (srlobs::kBlkH * srlobs::kBlkW), ntiles = srl_ceil_div(d->n, (long)srlobs::kTile);
      long nsplit = nblk < 256 ? 256 / nblk : 1;
      if (nsplit > ntiles) nsplit = ntiles;
      h.nsplit = (int)nsplit;
      h.xcd = obs_xcd_order();
      const unsigned grid = (unsigned)(nblk * nsplit);
      constexpr int lds = srlobs::kStages * srlobs::kStageBytes + srlobs::kMeta * srlobs::kTile * 16 + srlobs::kWaves * 96 * 4 +
                          (SRL_OBS_LINE_STORES ? srlobs::kWaves * srlobs::kTile * 144 : 0);  // stages, records, tables, store rows
      auto go = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * srlobs::kWaves), lds, st, h);
      };
      srl_count_dispatch(SRL_DISP_OBS_FWD_BF16, 256, 2, (int)grid);
      const char* dbg = getenv("SRL_OBS_DBG");  // timing experiments (wrong results): see obs_h2.h
      switch (dbg ? atoi(dbg) : 0) {
        case 1: go(srlobs::obs_fwd_h2_kernel<1, 1>); break;
        case 2: go(srlobs::obs_fwd_h2_kernel<1, 2>); break;
        case 4: go(srlobs::obs_fwd_h2_kernel<1, 4>); break;
        case 7: go(srlobs::obs_fwd_h2_kernel<1, 7>); break;
        default:
          if (d->act == 1) go(srlobs::obs_fwd_h2_kernel<1>);
          else if (d->act == 2) go(srlobs::obs_fwd_h2_kernel<2>);
          else go(srlobs::obs_fwd_h2_kernel<0>);
      }
      SRL_LAUNCH_CHECK();
      return 0;
    }
    if (!reuse_folded) {
      (void)hipMemsetAsync(bound, 0, sizeof(float), st);
      hipLaunchKernelGGL(srlobs::obs_fold_split_kernel<ObsIndex>, dim3((unsigned)(P * d->Cout)), dim3(256), 0, st, w, bias, gamma,
                         beta, P, (int)Kp, ix, wq, S, b2, bound, sqrtf((float)((long)d->H * d->W * d->Cin)));
    }
    srlobs::FwdArgs a{};
    a.y_h2 = y_h2; a.bound = bound; a.y_scale = y_scale; a.ent_order = ent_order; a.OH = OH;
    a.g = obs_geom(d, obs, mean, rstd, OW, row_index);
    a.wq = reinterpret_cast<const uint4*>(wq);
    a.S = S; a.b2 = b2; a.y = y; a.P = P; a.act = d->act;
    a.y_absmax = y_absmax;
    a.y_mask = y_mask;
    a.nsplit = obs_bf16_split(d->n, P, 3);
    const char* dbg = getenv("SRL_OBS_DBG");  // timing experiments (wrong results): see obs_bf16.h
    const dim3 grid(srlobs::xcd_position_grid(P, a.nsplit));
    srl_count_dispatch(SRL_DISP_OBS_FWD_BF16, 256, y_h2 ? 1 : 0, a.nsplit);
    switch (dbg ? atoi(dbg) : 0) {
      case 3: hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 3>), grid, dim3(256), 0, st, a); break;
      case 4: hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 4>), grid, dim3(256), 0, st, a); break;
      case 8: hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 8>), grid, dim3(256), 0, st, a); break;
      case 12: hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 12>), grid, dim3(256), 0, st, a); break;
      default:
        if (y_h2) hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 0, true, true>), grid, dim3(256), 0, st, a);
        else if (y_mask) hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 0, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((srlobs::obs_fwd_bf16_kernel<256, 0>), grid, dim3(256), 0, st, a);
    }
    SRL_LAUNCH_CHECK();
    return 0;
  }
  if (workspace && aligned16(workspace) && Kp % 4 == 0 && d->n >= 64) {
    // Position-batched form: one GEMM per output position over the samples, with the LayerNorm affine folded into
    // per-position weights (w * gamma) and biases (bias + w . beta): the gather then needs no table lookups.
    const int P = OH * OW;
    float* wg = workspace;
    float* b2 = wg + (long)P * d->Cout * Kp;
    const ObsIndex ix{d->Cin, d->H, d->W, d->KH, d->KW, d->stride, OW, channels_last ? 1 : 0};
    if (!reuse_folded)
      hipLaunchKernelGGL(obs_fold_affine_kernel, dim3((unsigned)(P * d->Cout)), dim3(256), 0, st, w, bias, gamma, beta, P,
                         d->Cout, ix, wg, b2);
    g.M = d->n; g.N = d->Cout;
    g.a = obs_patch_src(obs, is_u8, mean, rstd, nullptr, nullptr, d, 1, 1, channels_last);  // rows = samples
    g.a.brw = OW;
    g.a.by_stride = channels_last ? d->stride * d->W * d->Cin : d->stride * d->W;
    g.a.bx_stride = channels_last ? d->stride * d->Cin : d->stride;
    g.b = plain_src(wg, Kp);
    g.b.brw = 1; g.b.by_stride = (int)(d->Cout * Kp); g.b.bx_stride = 0;
    g.o = plain_out(y, (long)P * d->Cout);
    g.o.batch_stride = d->Cout;
    g.bias = b2; g.bias_batch = d->Cout;
    g.vec_b = 1;
    if (d->Cout > 64) rc = launch<128, 128, 2, 2, false, false, SRC_OBSN, SRC_PLAIN>(st, g, P, 1);
    else if (d->Cout > 32) rc = launch<256, 64, 4, 1, false, false, SRC_OBSN, SRC_PLAIN>(st, g, P, 1);
    else if (is_u8) rc = launch<256, 32, 4, 1, false, false, SRC_OBSN, SRC_PLAIN, false, 2 * BK, true>(st, g, P, 1);  // bytes in LDS
    else rc = launch<256, 32, 4, 1, false, false, SRC_OBSN, SRC_PLAIN>(st, g, P, 1);
  } else {
    g.M = d->n * OH * OW; g.N = d->Cout;
    g.a = obs_patch_src(obs, is_u8, mean, rstd, gamma, beta, d, OH * OW, OW, channels_last);
    g.b = plain_src(w, Kp);
    g.o = plain_out(y, d->Cout);
    g.bias = bias;
    g.vec_b = 1;
    SRL_CHECK_ARG(Kp % 4 == 0, "Cin*KH*KW must be a multiple of 4");
    if (d->Cout > 64) rc = launch<128, 128, 2, 2, false, false, SRC_OBS, SRC_PLAIN>(st, g, 1, 1);
    else if (d->Cout > 32) rc = launch<256, 64, 4, 1, false, false, SRC_OBS, SRC_PLAIN>(st, g, 1, 1);
    else rc = launch<256, 32, 4, 1, false, false, SRC_OBS, SRC_PLAIN>(st, g, 1, 1);
  }
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_obs_fwd(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* w, const float* bias, float* y, float* workspace, const int32_t* row_index,
                                  float* y_absmax, uint32_t* y_mask, int reuse_folded) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, channels_last ? 2 : 1),
                "unsupported geometry (planar: KW, W, stride, H*W multiples of 4; channels-last: Cin multiple of 4)");
  const long run = images_per_launch(d, is_u8 ? 1 : 4);
  const long in_b = (long)d->H * d->W * d->Cin * (is_u8 ? 1 : 4);
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    // with a slot index the frames (and their statistics) are addressed through it: only the index moves on
    const long adv = row_index ? 0 : i0;
    const int rc = conv2d_obs_fwd_run(stream, &s, static_cast<const uint8_t*>(obs) + adv * in_b, is_u8, channels_last, mean + adv,
                                      rstd + adv, gamma, beta, w, bias, y + i0 * out_e, workspace,
                                      row_index ? row_index + i0 : nullptr, y_absmax, y_mask ? y_mask + i0 * out_e / 32 : nullptr,
                                      reuse_folded || i0 > 0);
    if (rc != 0) return rc;
  }
  return 0;
}

extern "C" int srl_conv2d_obs_fold_h2(void* stream, const srl_conv_desc* d, const float* gamma, const float* beta, const float* w,
                                      const float* bias, float* workspace) {
  static const bool blocks_on = [] { const char* e = getenv("SRL_OBS_H2BLOCK"); return !(e && e[0] == '0'); }();
  if (check_desc(d) != 0 || !blocks_on || !obs_h2_block_geometry(d)) return 1;  // not the block kernel's layer: nothing written
  SRL_CHECK_ARG(gamma && beta && w && workspace && aligned16(workspace) && aligned16(gamma) && aligned16(beta), "null / unaligned tensor");
  obs_h2_fold((hipStream_t)stream, d, gamma, beta, w, bias, workspace);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_obs_fwd_h2(void* stream, const srl_conv_desc* d, const void* obs, const float* mean, const float* rstd,
                                     const float* gamma, const float* beta, const float* w, const float* bias, void* y_h2,
                                     float* y_scale, float* workspace, const int32_t* row_index, float* y_absmax, uint32_t* y_mask,
                                     int reuse_folded, int ent_order, void* records) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, 2), "unsupported geometry");
  SRL_CHECK_ARG(!records || aligned16(records), "records: 16-byte aligned");
  SRL_CHECK_ARG(ent_order == 0 || (ent_order == 2 && conv_out(d->H, d->KH, d->stride) % 2 == 0 && conv_out(d->W, d->KW, d->stride) % 2 == 0),
                "ent_order: 0 (raster) or 2 (parity-class major: even output extents)");
  const long run = images_per_launch(d, 1);
  const long in_b = (long)d->H * d->W * d->Cin;
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    const long adv = row_index ? 0 : i0;
    const int rc = conv2d_obs_fwd_run(stream, &s, static_cast<const uint8_t*>(obs) + adv * in_b, 1, 1, mean + adv, rstd + adv, gamma,
                                      beta, w, bias, nullptr, workspace, row_index ? row_index + i0 : nullptr, y_absmax,
                                      y_mask ? y_mask + i0 * out_e / 32 : nullptr, reuse_folded || i0 > 0,
                                      static_cast<uint8_t*>(y_h2) + i0 * out_e * 4, y_scale, ent_order, static_cast<uint4*>(records));
    if (rc != 0) return rc;
  }
  return 0;
}

// sample ranges per block of positions of obs_bwd_h2_kernel: one workgroup per CU
static int obs_bwd_h2_split(long P) {
  const long blocks = P / (srlobs::kBlkH * srlobs::kBlkW);
  long s = blocks > 0 ? 256 / blocks : 1;
  return (int)(s < 1 ? 1 : s);
}

extern "C" int64_t srl_conv2d_obs_bwd_workspace(const srl_conv_desc* d) {
  if (check_desc(d) != 0) return 0;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const long P = (long)OH * OW, Kp = (long)d->Cin * d->KH * d->KW;
  const long tiles = srl_ceil_div(d->Cout, 32) * srl_ceil_div(Kp, 256);
  long split = want_split(d->n, tiles, P);
  const long split16 = obs_bf16_split(d->n, (int)P, 3);
  if (split16 > split) split = split16;
  const long split_h2 = obs_bwd_h2_split(P);
  if (split_h2 > split) split = split_h2;
  // + the per-sample records and the rstd bound of the block kernel (obs_h2.h)
  return (int64_t)((split + 1) * P * d->Cout * Kp + 2 * P * d->Cout + 64 + 4 * (d->n + 32) + 4);
}

static int conv2d_obs_bwd_run(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                              const float* mean, const float* rstd, const float* gamma, const float* beta,
                              const float* w, const float* dz, float* dw, float* db, float* dgamma, float* dbeta,
                              float* workspace, const int32_t* row_index, int phase = 3, const float* dz_absmax = nullptr) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, channels_last ? 2 : 1), "unsupported geometry");
  SRL_CHECK_ARG(row_index == nullptr || srl_conv2d_obs_row_index_supported(d, is_u8, channels_last),
                "row_index is served by the byte kernels only (srl_conv2d_obs_row_index_supported)");
  SRL_CHECK_ARG(obs && mean && rstd && gamma && beta && w && dz && dw && db && dgamma && dbeta && workspace,
                "null tensor");
  SRL_CHECK_ARG(aligned16(obs) && aligned16(dz) && aligned16(workspace) && d->Cout % 4 == 0, "unaligned tensor / Cout % 4");
  if (d->n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int OH = conv_out(d->H, d->KH, d->stride), OW = conv_out(d->W, d->KW, d->stride);
  const int P = OH * OW;
  const long Kp = (long)d->Cin * d->KH * d->KW;
  // workspace: Q [P][Cout][Kp] | R [P][Cout] (padded to 16 B) | split slabs
  float* Q = workspace;
  float* R = Q + (long)P * d->Cout * Kp;
  float* C = R + (long)P * d->Cout;  // bf16 path: mean correction of Q (zero otherwise)
  float* slabs = C + (((long)P * d->Cout + 3) & ~3L);
  // R[pos, o] = sum_n dz[(n, pos), o]: column sums of the A tiles the batched product below stages anyway
  // phase (the byte kernels with split slabs only; 3 otherwise): bit 0 = this call opens an accumulation (Q, R, C start from
  // zero), bit 1 = it closes one (dW, db, dgamma, dbeta are formed from the sums): dW etc. are linear in (Q, R, C), so the chunks
  // of one update can share ONE finalisation (two launches, 59 us per 16 384-frame chunk) instead of one each
  const bool first = (phase & 1) != 0, last = (phase & 2) != 0;
  if (first && hipMemsetAsync(R, 0, sizeof(float) * 2 * P * d->Cout, st) != hipSuccess) return -EIO;
  const ObsIndex ix{d->Cin, d->H, d->W, d->KH, d->KW, d->stride, OW, channels_last ? 1 : 0};
  // the Atari geometry with a measured bound of |dz|: blocks of 2 x 4 positions per workgroup, frames and dz by LDS-DMA (obs_h2.h)
  static const bool blocks_on = [] { const char* e = getenv("SRL_OBS_BWD_H2BLOCK"); return !(e && e[0] == '0'); }();
  if (dz_absmax && blocks_on && obs_bf16_ok(d, is_u8, channels_last, obs) && d->Cin == 64 && d->KH == 2 && d->KW == 2 &&
      d->stride == 1 && d->Cout == 32 && OH % srlobs::kBlkH == 0 && OW % srlobs::kBlkW == 0 && d->n * (long)P * 128 < 0xffffffffL) {
    srlobs::BwdH2Args a{};
    const long n_pad = srl_ceil_div(d->n, (long)srlobs::kTile) * srlobs::kTile;
    float* rstd_max = C + (((long)P * d->Cout + 3) & ~3L);
    uint4* meta = reinterpret_cast<uint4*>(rstd_max + 4);
    float* slabs_h2 = rstd_max + 4 + 4 * n_pad;
    if (hipMemsetAsync(rstd_max, 0, sizeof(float), st) != hipSuccess) return -EIO;
    hipLaunchKernelGGL(srlobs::obs_meta_kernel, dim3((unsigned)srl_ceil_div(n_pad, 256L)), dim3(256), 0, st, row_index, mean, rstd,
                       (long)d->n, n_pad, meta, rstd_max);
    a.frames = static_cast<const uint8_t*>(obs); a.img_stride = (long)d->H * d->W * d->Cin; a.meta = meta; a.n = d->n;
    a.dz = dz; a.dz_bound = dz_absmax; a.rstd_bound = rstd_max; a.Q = slabs_h2; a.slab = (long)P * d->Cout * Kp; a.R = R; a.C = C;
    a.GW = d->W; a.OW = OW; a.OH = OH; a.P = P; a.nsplit = obs_bwd_h2_split(P);
    a.xcd = obs_xcd_order();
    if (a.nsplit > srl_ceil_div(d->n, (long)srlobs::kTileB)) a.nsplit = (int)srl_ceil_div(d->n, (long)srlobs::kTileB);  // (a range of tiles each)
    const unsigned grid = (unsigned)((P / (srlobs::kBlkH * srlobs::kBlkW)) * a.nsplit);
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, srlobs::kLdsB);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * srlobs::kWaves), srlobs::kLdsB, st, a);
    };
    srl_count_dispatch(SRL_DISP_OBS_BWD_BF16, 256, 2, a.nsplit);
    const char* dbg = getenv("SRL_OBSB_DBG");  // timing experiments (wrong results): see obs_h2.h
    switch (dbg ? atoi(dbg) : 0) {
      case 1: go(srlobs::obs_bwd_h2_kernel<1>); break;
      case 2: go(srlobs::obs_bwd_h2_kernel<2>); break;
      case 3: go(srlobs::obs_bwd_h2_kernel<3>); break;
      case 4: go(srlobs::obs_bwd_h2_kernel<4>); break;
      case 7: go(srlobs::obs_bwd_h2_kernel<7>); break;
      case 8: go(srlobs::obs_bwd_h2_kernel<8>); break;
      case 19: go(srlobs::obs_bwd_h2_kernel<19>); break;
      case 23: go(srlobs::obs_bwd_h2_kernel<23>); break;
      case 12: go(srlobs::obs_bwd_h2_kernel<12>); break;
      case 15: go(srlobs::obs_bwd_h2_kernel<15>); break;
      default: go(srlobs::obs_bwd_h2_kernel<0>);
    }
    SRL_LAUNCH_CHECK();
    reduce_slabs(st, slabs_h2, a.nsplit, (long)P, (long)d->Cout, Kp, Q, Kp, (long)d->Cout * Kp, first ? 0 : 1);
    SRL_LAUNCH_CHECK();
    if (!last) return 0;
    hipLaunchKernelGGL(obs_dw_kernel, dim3((unsigned)srl_ceil_div(d->Cout * Kp, kDwEL)), dim3(256), 0, st, Q, R, C, gamma, beta,
                       P, d->Cout, ix, dw, db);
    hipLaunchKernelGGL(obs_affine_kernel, dim3((unsigned)srl_ceil_div(d->Cin * d->H * d->W, 256)), dim3(256), 0, st, Q, R, C, w,
                       OH, d->Cout, ix, dgamma, dbeta);
    SRL_LAUNCH_CHECK();
    return 0;
  }
  if (obs_bf16_ok(d, is_u8, channels_last, obs) && (P * d->Cout) % 4 == 0) {
    srlobs::BwdArgs a{};
    a.g = obs_geom(d, obs, mean, rstd, OW, row_index);
    a.dz = dz; a.R = R; a.C = C; a.P = P;
    a.nsplit = obs_bf16_split(d->n, P, 3);
    // an accumulation over calls goes through the slab + reduce even when one slab holds the call (a ragged last chunk)
    const bool via_slabs = a.nsplit > 1 || phase != 3;
    a.Q = via_slabs ? slabs : Q;
    a.slab = (long)P * d->Cout * Kp;
    srl_count_dispatch(SRL_DISP_OBS_BWD_BF16, 256, 0, a.nsplit);
    hipLaunchKernelGGL(srlobs::obs_bwd_bf16_kernel<256>, dim3(srlobs::xcd_position_grid(P, a.nsplit)), dim3(256), 0, st, a);
    SRL_LAUNCH_CHECK();
    if (via_slabs) {
      reduce_slabs(st, slabs, a.nsplit, (long)P, (long)d->Cout, Kp, Q, Kp, (long)d->Cout * Kp, first ? 0 : 1);
      SRL_LAUNCH_CHECK();
    }
    if (!last) return 0;
    hipLaunchKernelGGL(obs_dw_kernel, dim3((unsigned)srl_ceil_div(d->Cout * Kp, kDwEL)), dim3(256), 0, st, Q, R, C, gamma, beta,
                       P, d->Cout, ix, dw, db);
    hipLaunchKernelGGL(obs_affine_kernel, dim3((unsigned)srl_ceil_div(d->Cin * d->H * d->W, 256)), dim3(256), 0, st, Q, R, C, w,
                       OH, d->Cout, ix, dgamma, dbeta);
    SRL_LAUNCH_CHECK();
    return 0;
  }
  SRL_CHECK_ARG(phase == 3, "phase: only the byte kernels accumulate over calls");
  int rc;
  // Q[pos][o][k] = sum_n dz[(n,pos), o] * xhat[n, patch(pos)[k]]   — one batched GEMM over the P output positions
  GemmArgs g{};
  g.M = d->Cout; g.N = Kp; g.K = d->n;
  g.a = plain_src(dz, (long)P * d->Cout);  // A(i = o, k = n) = dz[n*P*Cout + pos*Cout + o]
  g.a.brw = OW; g.a.by_stride = OW * d->Cout; g.a.bx_stride = d->Cout;
  g.b = obs_patch_src(obs, is_u8, mean, rstd, nullptr, nullptr, d, 1, 1, channels_last);  // rows = samples; xhat
  g.b.brw = OW;
  g.b.by_stride = channels_last ? d->stride * d->W * d->Cin : d->stride * d->W;
  g.b.bx_stride = channels_last ? d->stride * d->Cin : d->stride;
  const long tiles = srl_ceil_div(d->Cout, 32) * srl_ceil_div(Kp, 256);
  const int nsplit = plan_split(d->n, want_split(d->n, tiles, P), &g.k_per_split);
  g.o = plain_out(nsplit > 1 ? slabs : Q, Kp);
  g.o.batch_stride = (long)d->Cout * Kp;
  g.slab = (long)P * d->Cout * Kp;
  g.vec_a = 1; g.vec_b = 1;
  g.a_colsum = R; g.a_colsum_batch = d->Cout;
  rc = is_u8 ? launch<32, 256, 1, 4, true, true, SRC_PLAIN, SRC_OBSN, false, 2 * BK, true>(st, g, P, nsplit)  // bytes in LDS
             : launch<32, 256, 1, 4, true, true, SRC_PLAIN, SRC_OBSN>(st, g, P, nsplit);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  if (nsplit > 1) {
    reduce_slabs(st, slabs, nsplit, (long)P, (long)d->Cout, Kp, Q, Kp, (long)d->Cout * Kp, 0);
    SRL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(obs_dw_kernel, dim3((unsigned)srl_ceil_div(d->Cout * Kp, kDwEL)), dim3(256), 0, st, Q, R, C, gamma, beta, P,
                     d->Cout, ix, dw, db);
  hipLaunchKernelGGL(obs_affine_kernel, dim3((unsigned)srl_ceil_div(d->Cin * d->H * d->W, 256)), dim3(256), 0, st, Q, R, C, w,
                     OH, d->Cout, ix, dgamma, dbeta);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_obs_bwd(void* stream, const srl_conv_desc* d, const void* obs, int is_u8, int channels_last,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  const float* w, const float* dz, float* dw, float* db, float* dgamma, float* dbeta,
                                  float* workspace, const int32_t* row_index, int phase, const float* dz_absmax) {
  SRL_CHECK_ARG(srl_conv2d_supported(d, channels_last ? 2 : 1), "unsupported geometry");
  SRL_CHECK_ARG(phase >= 0 && phase <= 3, "phase: bit 0 opens, bit 1 closes an accumulation over calls");
  const long run = images_per_launch(d, is_u8 ? 1 : 4);
  const long in_b = (long)d->H * d->W * d->Cin * (is_u8 ? 1 : 4);
  const long out_e = (long)conv_out(d->H, d->KH, d->stride) * conv_out(d->W, d->KW, d->stride) * d->Cout;
  for (long i0 = 0; i0 < d->n || i0 == 0; i0 += run) {  // every run adds into dw / db / dgamma / dbeta
    srl_conv_desc s = *d;
    s.n = d->n - i0 < run ? d->n - i0 : run;
    const long adv = row_index ? 0 : i0;
    // several runs per call: each closes its own accumulation, unless the caller accumulates over calls -- then the runs of a
    // call are links of that chain
    const bool only = d->n <= run || phase == 3;
    const int rc = conv2d_obs_bwd_run(stream, &s, static_cast<const uint8_t*>(obs) + adv * in_b, is_u8, channels_last, mean + adv,
                                      rstd + adv, gamma, beta, w, dz + i0 * out_e, dw, db, dgamma, dbeta, workspace,
                                      row_index ? row_index + i0 : nullptr,
                                      only ? phase : ((i0 == 0 ? phase & 1 : 0) | (i0 + run >= d->n ? phase & 2 : 0)), dz_absmax);
    if (rc != 0) return rc;
  }
  return 0;
}
