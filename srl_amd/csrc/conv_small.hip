// Direct convolutions for layers with 4 or 8 channels on both sides (round 6): NHWC float32, 3 x 3 or 5 x 5, stride 1, no padding -- the
// reference's DEFAULT convolution stack behind the first layer (modules/cnn.py:96-98: C -> 2C -> C channels, the football preset's
// (4, 96, 72) frames: 4 -> 8 and 8 -> 4).  As implicit GEMMs on the matrix cores (gemm_core.h, 256 x 32 tiles) a layer with 4 or 8
// output channels fills an eighth to a quarter of a tile's columns and gathers its patch rows element by element: 14 ms per
// launch of 2560 images, 2.5 TFLOP/s -- 560 of the 1090 ms of a football-sized update (256 envs x 200 steps), against ~20 GB of
// activations that the HBM moves in 4 ms.
//
// These layers are bandwidth work with a few hundred multiply-adds per pixel, so they run on the vector units:
//   * forward / data gradient: a thread per output pixel, its 3 x 3 window of 16- or 32-byte pixels from the L1 (neighbouring
//     lanes read neighbouring pixels: whole lines), the 288 weights as scalar operands (uniform addresses: s_load), 288 FMAs;
//   * weight gradient: a wavefront owns a 32-pixel-wide column block of two images and walks down the rows with a three-row
//     window in registers (one new row of the input and one of dz per step), every lane accumulating ALL 288 weight gradients
//     of its pixel column in registers (1 wavefront per SIMD); the lanes are summed once per wavefront at the end, the
//     wavefronts' sums by a second launch (slabs: no atomics, bit-reproducible).
#include "../../include/srl_hip.h"
#include "srl_common.h"
#include <type_traits>

namespace {

template <int C> __device__ __forceinline__ void px_load(const float* __restrict__ p, float (&v)[C]) {
#pragma unroll
  for (int q = 0; q < C / 4; ++q) {
    const float4 t = reinterpret_cast<const float4*>(p)[q];
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}
template <int C> __device__ __forceinline__ void px_store(float* __restrict__ p, const float (&v)[C]) {
#pragma unroll
  for (int q = 0; q < C / 4; ++q) reinterpret_cast<float4*>(p)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// y[n, oy, ox, :] = act(b + sum_{ky, kx, ci} x[n, oy + ky, ox + kx, ci] w[co, ky, kx, ci])
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(256) void conv_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, long nimg, int H, int W,
                                                             int act) {
  // grid: (pixel blocks of an image, images): a 64-bit division per thread cost more than the layer's 288 multiply-adds
  const int OH = H - K + 1, OW = W - K + 1;
  const int r = blockIdx.x * 256 + threadIdx.x;
  const long n = (long)blockIdx.y + (long)blockIdx.z * 65535;
  if (r >= OH * OW || n >= nimg) return;
  const int oy = r / OW, ox = r - oy * OW;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias ? bias[co] : 0.f;
  const float* xp = x + ((n * H + oy) * (long)W + ox) * CIN;
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      float v[CIN];
      px_load<CIN>(xp + (ky * W + kx) * CIN, v);
#pragma unroll
      for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[co] = __builtin_fmaf(v[ci], w[((co * K + ky) * K + kx) * CIN + ci], acc[co]);
    }
  if (act == 1) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = acc[co] > 0.f ? acc[co] : 0.f;
  } else if (act == 2) {
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = tanhf(acc[co]);
  }
  px_store<COUT>(y + (n * (long)(OH * OW) + r) * COUT, acc);
}

// dx[n, y, x, ci] = (sum_{ky, kx, co} dz[n, y - ky, x - kx, co] w[co, ky, kx, ci]) * act'(xact[n, y, x, ci])
template <int CIN, int COUT, int K>
__global__ __launch_bounds__(256) void conv_small_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                               const float* __restrict__ xact, int dact, float* __restrict__ dx, long nimg,
                                                               int H, int W) {
  const int OH = H - K + 1, OW = W - K + 1;
  const int r = blockIdx.x * 256 + threadIdx.x;
  const long n = (long)blockIdx.y + (long)blockIdx.z * 65535;
  if (r >= H * W || n >= nimg) return;
  const int yy = r / W, xx = r - yy * W;
  float acc[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
  const float* zp = dz + ((n * OH + yy) * (long)OW + xx) * COUT;   // dz[n, yy - ky, xx - kx, :] = zp - (ky * OW + kx) * COUT
  auto tap = [&](int ky, int kx) {
    float v[COUT];
    px_load<COUT>(zp - (ky * OW + kx) * COUT, v);
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[ci] = __builtin_fmaf(v[co], w[((co * K + ky) * K + kx) * CIN + ci], acc[ci]);
  };
  // (an untested copy of the loop for the pixels away from the border doubled the scalar weight loads: 573 SGPRs spilled, slower)
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
      if ((unsigned)(yy - ky) < (unsigned)OH && (unsigned)(xx - kx) < (unsigned)OW) tap(ky, kx);
  const long p = n * (long)(H * W) + r;
  if (xact && dact) {
    float a[CIN];
    px_load<CIN>(xact + p * CIN, a);
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      if (dact == 1) acc[ci] = a[ci] > 0.f ? acc[ci] : 0.f;
      else acc[ci] *= 1.f - a[ci] * a[ci];   // tanh: the activation's output is what was kept
    }
  }
  px_store<CIN>(dx + p * CIN, acc);
}

constexpr int kWgGrid = 256;   // workgroups of the weight gradient (4 wavefronts each): one per CU

// slab[wave][(co, ky, kx, ci)] and [COUT] bias sums behind them.  COB output channels per launch (co0 .. co0 + COB - 1): a lane
// keeps COB K K CIN sums -- 288 at most with 4 input channels, 144 with 8 (a 5 x 5 layer takes its output channels two at a
// time), re-reading the activations per launch.
template <int CIN, int COUT, int K, int COB>
__global__ __launch_bounds__(256, 1) void conv_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, long nimg, int H,
                                                                  int W, int co0, float* __restrict__ slabs) {
  constexpr int NW = COUT * K * K * CIN;
  const int OH = H - K + 1, OW = W - K + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l32 = lane & 31;
  const int ncb = (OW + 31) / 32;
  const long units = ((nimg + 1) / 2) * ncb;
  const long nwaves = (long)gridDim.x * 4, wg = (long)blockIdx.x * 4 + wave;
  float acc[COB][K * K][CIN];
  float accb[COB];
#pragma unroll
  for (int co = 0; co < COB; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[co][t][ci] = 0.f;
  }
  for (long u = wg; u < units; u += nwaves) {
    const long pair = u / ncb;
    const int cb = (int)(u - pair * ncb);
    const long img = pair * 2 + half;
    const int ox = cb * 32 + l32;
    const bool live = img < nimg && ox < OW;
    const long imgc = img < nimg ? img : nimg - 1;
    const int oxc = ox < OW ? ox : 0;
    const float* xb = x + (imgc * H * (long)W + oxc) * CIN;          // + (row * W + kx) * CIN
    const float* zb = dz + (imgc * OH * (long)OW + oxc) * COUT;      // + row * OW * COUT
    float win[K][K][CIN];   // rows oy .. oy + K - 1 (rotating: row r lives in win[r % K])
#pragma unroll
    for (int r = 0; r < K - 1; ++r)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) px_load<CIN>(xb + ((long)r * W + kx) * CIN, win[r][kx]);
    auto step = [&](int oy, auto rot_c) {
      constexpr int ROT = decltype(rot_c)::value;   // oy % K
      // the new row oy + K - 1 goes where row oy - 1 was: slot (ROT + K - 1) % K
#pragma unroll
      for (int kx = 0; kx < K; ++kx) px_load<CIN>(xb + ((long)(oy + K - 1) * W + kx) * CIN, win[(ROT + K - 1) % K][kx]);
      float zf[COUT];
      px_load<COUT>(zb + (long)oy * OW * COUT, zf);
#pragma unroll
      for (int co = 0; co < COB; ++co) {
        // (co0 is uniform: a select over the pixel's COUT values)
        float z = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < COUT; ++c2) z = (c2 == co0 + co) ? zf[c2] : z;
        z = live ? z : 0.f;
        accb[co] += z;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
          for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) acc[co][ky * K + kx][ci] = __builtin_fmaf(z, win[(ROT + ky) % K][kx][ci], acc[co][ky * K + kx][ci]);
      }
    };
    int oy = 0;
    for (; oy + K <= OH; oy += K) {
      step(oy, std::integral_constant<int, 0>{});
      step(oy + 1, std::integral_constant<int, 1>{});
      step(oy + 2, std::integral_constant<int, 2>{});
      if constexpr (K == 5) {
        step(oy + 3, std::integral_constant<int, 3>{});
        step(oy + 4, std::integral_constant<int, 4>{});
      }
    }
    if (oy < OH) { step(oy, std::integral_constant<int, 0>{}); ++oy; }
    if (oy < OH) { step(oy, std::integral_constant<int, 1>{}); ++oy; }
    if constexpr (K == 5) {
      if (oy < OH) { step(oy, std::integral_constant<int, 2>{}); ++oy; }
      if (oy < OH) { step(oy, std::integral_constant<int, 3>{}); ++oy; }
    }
  }
  // sum over the wavefront's lanes, lane 0 writes this launch's part of the slab
  float* slab = slabs + wg * (NW + COUT);
#pragma unroll
  for (int co = 0; co < COB; ++co) {
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        float v = acc[co][t][ci];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) slab[((co0 + co) * K * K + t) * CIN + ci] = v;
      }
    float v = accb[co];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) slab[NW + co0 + co] = v;
  }
}

// gw[i] += sum over slabs (i < nw), gb[i - nw] += ... : one workgroup per output
__global__ __launch_bounds__(256) void conv_small_reduce_kernel(const float* __restrict__ slabs, int nslab, int per, int nw, float* __restrict__ gw,
                                                                float* __restrict__ gb) {
  __shared__ float part[4];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < nslab; k += 256) s += slabs[(long)k * per + i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (part[0] + part[1]) + (part[2] + part[3]);
    if (i < nw) gw[i] += t;
    else if (gb) gb[i - nw] += t;
  }
}

bool small_ok(const srl_conv_desc* d) {
  // (5 x 5 with 8 input channels: the weight gradient's window alone is 200 registers -- left to the implicit GEMMs)
  return d && d->KH == d->KW && (d->KH == 3 || (d->KH == 5 && d->Cin == 4)) && d->stride == 1 && d->H >= d->KH && d->W >= d->KW &&
         (d->Cin == 4 || d->Cin == 8) && (d->Cout == 4 || d->Cout == 8) && d->act >= 0 && d->act <= 2;
}

// one instantiation per (Cin, Cout, K)
template <class F> bool small_dispatch(const srl_conv_desc* d, F&& f) {
#define SRL_CS_CASE(CI, CO, KK) if (d->Cin == CI && d->Cout == CO && d->KH == KK) { f(std::integral_constant<int, CI>{}, std::integral_constant<int, CO>{}, std::integral_constant<int, KK>{}); return true; }
  SRL_CS_CASE(4, 4, 3) SRL_CS_CASE(4, 8, 3) SRL_CS_CASE(8, 4, 3) SRL_CS_CASE(8, 8, 3)
  SRL_CS_CASE(4, 4, 5) SRL_CS_CASE(4, 8, 5)
#undef SRL_CS_CASE
  return false;
}

}  // namespace

extern "C" int srl_conv2d_small_supported(const srl_conv_desc* d) { return small_ok(d) ? 1 : 0; }

extern "C" int srl_conv2d_small_fwd(void* stream, const srl_conv_desc* d, const float* x, const float* w, const float* bias, float* y) {
  SRL_CHECK_ARG(small_ok(d) && x && w && y && d->n >= 0, "geometry not supported (srl_conv2d_small_supported) / null tensor");
  SRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "unaligned tensor");
  if (d->n == 0) return 0;
  const dim3 grid((unsigned)srl_ceil_div((long)(d->H - d->KH + 1) * (d->W - d->KW + 1), 256L), (unsigned)(d->n < 65535 ? d->n : 65535),
                  (unsigned)srl_ceil_div(d->n, 65535L));
  hipStream_t st = (hipStream_t)stream;
  small_dispatch(d, [&](auto ci, auto co, auto k) {
    hipLaunchKernelGGL((conv_small_fwd_kernel<decltype(ci)::value, decltype(co)::value, decltype(k)::value>), grid, dim3(256), 0, st, x, w, bias,
                       y, (long)d->n, (int)d->H, (int)d->W, (int)d->act);
  });
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_conv2d_small_dgrad(void* stream, const srl_conv_desc* d, const float* dz, const float* w, const float* x_act, int32_t dact,
                                      float* dx) {
  SRL_CHECK_ARG(small_ok(d) && dz && w && dx && d->n >= 0 && dact >= 0 && dact <= 2, "geometry not supported / null tensor");
  SRL_CHECK_ARG((((uintptr_t)dz | (uintptr_t)dx | (uintptr_t)x_act) & 15) == 0, "unaligned tensor");
  if (d->n == 0) return 0;
  const dim3 grid((unsigned)srl_ceil_div((long)d->H * d->W, 256L), (unsigned)(d->n < 65535 ? d->n : 65535), (unsigned)srl_ceil_div(d->n, 65535L));
  hipStream_t st = (hipStream_t)stream;
  small_dispatch(d, [&](auto ci, auto co, auto k) {
    hipLaunchKernelGGL((conv_small_dgrad_kernel<decltype(ci)::value, decltype(co)::value, decltype(k)::value>), grid, dim3(256), 0, st, dz, w,
                       x_act, (int)dact, dx, (long)d->n, (int)d->H, (int)d->W);
  });
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t srl_conv2d_small_wgrad_workspace(const srl_conv_desc* d) {
  if (!small_ok(d)) return 0;
  return (int64_t)kWgGrid * 4 * (d->Cout * d->KH * d->KW * d->Cin + d->Cout);
}

extern "C" int srl_conv2d_small_wgrad(void* stream, const srl_conv_desc* d, const float* x, const float* dz, float* workspace, float* gw,
                                      float* gb) {
  SRL_CHECK_ARG(small_ok(d) && x && dz && workspace && gw && d->n >= 0, "geometry not supported / null tensor");
  SRL_CHECK_ARG((((uintptr_t)x | (uintptr_t)dz) & 15) == 0, "unaligned tensor");
  if (d->n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nw = d->Cout * d->KH * d->KW * d->Cin, per = nw + d->Cout;
  small_dispatch(d, [&](auto ci, auto co, auto k) {
    constexpr int CI = decltype(ci)::value, CO = decltype(co)::value, KK = decltype(k)::value;
    // sums per lane: 288 with 4-channel pixels; 144 with 8 (their three-row window is 72 registers: at 288 the compiler spilled 56).
    // One wavefront per SIMD, accumulators partly in AGPRs.  (Tried: 144 sums at two wavefronts per SIMD with the next row
    // prefetched -- the unrolled steps spilled 230-400 registers at the 256 cap: football's weight gradients 96 -> 238 ms.)
    constexpr int cob0 = (CI == 4 ? 288 : 144) / (KK * KK * CI), COB = cob0 < 1 ? 1 : (cob0 > CO ? CO : cob0);
    static_assert(CO % COB == 0, "");
    for (int c0 = 0; c0 < CO; c0 += COB)
      hipLaunchKernelGGL((conv_small_wgrad_kernel<CI, CO, KK, COB>), dim3(kWgGrid), dim3(256), 0, st, x, dz, (long)d->n, (int)d->H, (int)d->W, c0,
                         workspace);
  });
  hipLaunchKernelGGL(conv_small_reduce_kernel, dim3((unsigned)per), dim3(256), 0, st, workspace, kWgGrid * 4, per, nw, gw, gb);
  SRL_LAUNCH_CHECK();
  return 0;
}
