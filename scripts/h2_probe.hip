// Standalone probe of csrc/h2gemm.h: correctness against float64 reference kernels and timing, per shape of the benchmarked
// network.  Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o scripts/h2_probe scripts/h2_probe.hip ; run on the GPU box.
//   h2_probe [n_check] [n_time] [stages]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include "../srl_amd/csrc/h2conv.h"
using namespace srlh2;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (uint32_t)x;
}
// kind 0: uniform [-amp, amp); 1: relu-like (half zeros, rest |.| * amp, a few large)
__global__ void fill_kernel(float* p, int64_t n, uint64_t seed, float amp, int kind) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32(seed * 0x9e3779b97f4a7c15ull + i);
    float u = (float)(h >> 8) * (1.f / 16777216.f) * 2.f - 1.f;
    if (kind == 1) { u = u < 0.f ? 0.f : u * u * 3.f; }
    if (kind == 2) { u = fabsf(u) + 0.1f; }
    p[i] = u * amp;
  }
}
__global__ void absmax_kernel(const float* p, int64_t n, float* out) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(p[i]));
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) atomicMax((int*)out, __float_as_int(m));
}
__global__ void rownorm1_kernel(const float* w, int rows, int cols, float* out) {
  const int r = blockIdx.x;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) s += fabsf(w[(int64_t)r * cols + c]);
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  __shared__ float sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    atomicMax((int*)out, __float_as_int(t));
  }
}
// references (float64 accumulate from the float32 tensors)
__global__ void ref_dense(const float* x, const float* w, const float* bias, int64_t M, int NC, int K, int relu, float* y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * NC) return;
  const int64_t m = i / NC; const int c = (int)(i % NC);
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)x[m * K + k] * (double)w[(int64_t)c * K + k];
  if (bias) s += bias[c];
  if (relu && s < 0) s = 0;
  y[i] = (float)s;
}
__global__ void ref_conv(const float* x, const float* w, const float* bias, int64_t n, int H, int W, int C, int KH, int KW, int st,
                         int Cout, int relu, float* y) {
  const int OH = (H - KH) / st + 1, OW = (W - KW) / st + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * OH * OW * Cout) return;
  const int co = (int)(i % Cout); int64_t r = i / Cout;
  const int ox = (int)(r % OW); r /= OW; const int oy = (int)(r % OH); const int64_t img = r / OH;
  double s = 0;
  for (int ky = 0; ky < KH; ++ky) for (int kx = 0; kx < KW; ++kx) for (int c = 0; c < C; ++c)
    s += (double)x[((img * H + oy * st + ky) * W + ox * st + kx) * C + c] * (double)w[((int64_t)(co * KH + ky) * KW + kx) * C + c];
  if (bias) s += bias[co];
  if (relu && s < 0) s = 0;
  y[i] = (float)s;
}
// dx[img,y,x,ci] = sum dz[img,oy,ox,co] w[co,ky,kx,ci], y = oy st + ky;  * (act[img,y,x,ci] > 0) if act
__global__ void ref_dgrad(const float* dz, const float* w, const float* act, int64_t n, int H, int W, int C, int KH, int KW, int st,
                          int Cout, float* dx) {
  const int OH = (H - KH) / st + 1, OW = (W - KW) / st + 1;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * H * W * C) return;
  const int ci = (int)(i % C); int64_t r = i / C;
  const int x = (int)(r % W); r /= W; const int y = (int)(r % H); const int64_t img = r / H;
  double s = 0;
  for (int ky = 0; ky < KH; ++ky) {
    const int ty = y - ky; if (ty < 0 || ty % st) continue; const int oy = ty / st; if (oy >= OH) continue;
    for (int kx = 0; kx < KW; ++kx) {
      const int tx = x - kx; if (tx < 0 || tx % st) continue; const int ox = tx / st; if (ox >= OW) continue;
      for (int co = 0; co < Cout; ++co)
        s += (double)dz[((img * OH + oy) * OW + ox) * Cout + co] * (double)w[((int64_t)(co * KH + ky) * KW + kx) * C + ci];
    }
  }
  if (act && !(act[i] > 0.f)) s = 0;
  dx[i] = (float)s;
}
// regrouped weights for the data gradient: wg[(cls, ci)][(t, co)] = w[co][py + st dy][px + st dx][ci]
__global__ void regroup_kernel(const float* w, int C, int KH, int KW, int st, int Cout, float* wg) {
  const int TH = KH / st, TW = KW / st, K = TH * TW * Cout;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= st * st * C * K) return;
  const int k = i % K, row = i / K;
  const int co = k % Cout, t = k / Cout, dy = t / TW, dx = t % TW;
  const int ci = row % C, cls = row / C, py = cls / st, px = cls % st;
  wg[i] = w[((int64_t)(co * KH + py + st * dy) * KW + px + st * dx) * C + ci];
}
__global__ void mask_kernel(const float* a, int64_t n32, uint32_t* m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n32) return;
  uint32_t b = 0;
  for (int j = 0; j < 32; ++j) b |= (a[i * 32 + j] > 0.f ? 1u : 0u) << j;
  m[i] = b;
}
// h2-order sign bytes of a [pixels, C] float tensor: byte (pixel, block, group), bit j = element h2p_elem(g, j) > 0
__global__ void mask_h2_kernel(const float* a, int64_t npix, int C, uint8_t* m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix * (C / 8)) return;
  const int64_t pix = i / (C / 8); const int gi = (int)(i % (C / 8)), blk = gi >> 2, g = gi & 3;
  uint32_t b = 0;
  for (int j = 0; j < 8; ++j) b |= (a[pix * C + blk * 32 + h2p_elem(g, j)] > 0.f ? 1u : 0u) << j;
  m[i] = (uint8_t)b;
}
__global__ void cmp_kernel(const float* a, const float* b, int64_t n, double* out /* max abs diff, max abs ref, sum sq diff, sum sq ref */) {
  double md = 0, mr = 0, sd = 0, sr = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = fabs((double)a[i] - (double)b[i]);
    md = fmax(md, d); mr = fmax(mr, fabs((double)b[i])); sd += d * d; sr += (double)b[i] * b[i];
  }
  atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(md));
  atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(mr));
  atomicAdd(&out[2], sd); atomicAdd(&out[3], sr);
}
__global__ void cmp_mask_kernel(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* out) {
  unsigned long long d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d += __popc(a[i] ^ b[i]);
  if (d) atomicAdd(out, d);
}

template <class T> T* dalloc(int64_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static void fill(float* p, int64_t n, uint64_t seed, float amp, int kind = 0) { fill_kernel<<<2048, 256>>>(p, n, seed, amp, kind); }
static float* absmax_of(const float* p, int64_t n) { float* o = dalloc<float>(1); absmax_kernel<<<1024, 256>>>(p, n, o); return o; }
static float* pack(const float* src, int64_t rows, int C, const float* amax, float** scale_out) {
  uint8_t* d = (uint8_t*)dalloc<float>(rows * C);
  *scale_out = dalloc<float>(1);
  h2_pack_kernel<<<2048, 256>>>(src, C, rows, C, amax, nullptr, *scale_out, d);
  return (float*)d;
}
static bool report(const char* what, const float* got, const float* ref, int64_t n, double tol) {
  double* o = dalloc<double>(4);
  cmp_kernel<<<1024, 256>>>(got, ref, n, o);
  double h[4]; CK(hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost)); CK(hipFree(o));
  const double rel_max = h[0] / (h[1] > 0 ? h[1] : 1), rel_rms = sqrt(h[2] / (h[3] > 0 ? h[3] : 1));
  const bool ok = rel_max < tol && rel_rms == rel_rms;
  printf("  %-34s max|d|/max|ref| %.3e  rms rel %.3e  %s\n", what, rel_max, rel_rms, ok ? "ok" : "FAIL");
  return ok;
}
template <class F> static double time_ms(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

static int g_stages = 3;
template <int NCB, int XMODE> static int launch(H2Args a) {
  return g_stages == 4 ? h2gemm_launch<NCB, XMODE, 4>(0, a) : h2gemm_launch<NCB, XMODE, 3>(0, a);
}

static bool g_all_ok = true;

static void run_dense(const char* name, int64_t M, int NC, int K, bool h2out, int64_t Mt, int ncb = 0) {
  printf("%s: dense M=%ld NC=%d K=%d out=%s\n", name, (long)M, NC, K, h2out ? "h2p" : "f32");
  const int64_t Mx = M > Mt ? M : Mt;
  float* x = dalloc<float>(Mx * K); float* w = dalloc<float>((int64_t)NC * K); float* b = dalloc<float>(NC);
  fill(x, Mx * K, 1, 2.f, 1); fill(w, (int64_t)NC * K, 2, 0.03f); fill(b, NC, 3, 0.1f);
  float *sx, *sw; float* ax = absmax_of(x, Mx * K); float* aw = absmax_of(w, (int64_t)NC * K);
  float* xp = pack(x, Mx, K, ax, &sx); float* wp = pack(w, NC, K, aw, &sw);
  float* wn = dalloc<float>(1); rownorm1_kernel<<<NC, 256>>>(w, NC, K, wn);
  float* ab = absmax_of(b, NC);
  float* y = dalloc<float>(Mx * NC); float* yr = dalloc<float>(M * NC); float* yu = dalloc<float>(M * NC);
  float* osc = dalloc<float>(1); float* oam = dalloc<float>(1);
  uint32_t* mk = dalloc<uint32_t>(Mx * NC / 32); uint32_t* mkr = dalloc<uint32_t>(M * NC / 32);
  H2Args a = {};
  a.x = xp; a.w = wp; a.sx = sx; a.sw = sw; a.M = M; a.NC = NC; a.nk = K / 32; a.w_row_bytes = K * 4; a.x_row_bytes = K * 4;
  a.bias = b; a.act = 1; a.out_fmt = h2out ? H2O_H2P : H2O_F32; a.out = y; a.out_row_bytes = NC * 4;
  a.out_scale = osc; a.bound_in = ax; a.bound_w = wn; a.bound_b = ab; a.out_absmax = oam; a.mask_out = mk;
  auto go = [&] { if (ncb == 8) h2gemm_launch<8, H2X_DENSE, 2, false>(0, a); else if (ncb == 4 || (ncb == 0 && NC % 128 == 0)) launch<4, H2X_DENSE>(a); else launch<2, H2X_DENSE>(a); };
  go();
  CK(hipDeviceSynchronize());
  ref_dense<<<(unsigned)((M * NC + 255) / 256), 256>>>(x, w, b, M, NC, K, 1, yr);
  const float* got = y;
  if (h2out) { h2_unpack_kernel<<<2048, 256>>>((const uint8_t*)y, M, NC, osc, yu, NC); got = yu; }
  g_all_ok &= report("output vs float64", got, yr, M * NC, 2e-6);
  mask_kernel<<<(unsigned)((M * NC / 32 + 255) / 256), 256>>>(yr, M * NC / 32, mkr);
  unsigned long long* md = dalloc<unsigned long long>(1); cmp_mask_kernel<<<512, 256>>>(mk, mkr, M * NC / 32, md);
  unsigned long long mdh; CK(hipMemcpy(&mdh, md, 8, hipMemcpyDeviceToHost));
  float oamh, rmh; float* rm = absmax_of(yr, M * NC); CK(hipMemcpy(&oamh, oam, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&rmh, rm, 4, hipMemcpyDeviceToHost));
  float osch, axh, wnh; CK(hipMemcpy(&osch, osc, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&axh, ax, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&wnh, wn, 4, hipMemcpyDeviceToHost));
  printf("  mask bits differing %llu of %ld (sign of values near 0); absmax %.5g (ref %.5g); out scale %g, bound %.4g\n", mdh, (long)(M * NC), oamh, rmh, osch, axh * wnh);
  if (Mt > 0) {
    a.M = Mt; a.mask_out = mk;
    const double ms = time_ms(go, 20);
    printf("  TIME M=%ld: %.1f us  (%.1f TFLOP/s float32-equivalent)\n", (long)Mt, ms * 1e3, 2.0 * Mt * NC * K / ms * 1e-9);
  }
  for (void* p : {(void*)x, (void*)w, (void*)b, (void*)xp, (void*)wp, (void*)y, (void*)yr, (void*)yu, (void*)mk, (void*)mkr}) CK(hipFree(p));
}

static void fill_conv_koff(H2Args& a, int W, int C, int KH, int KW, bool parity_order, int st) {
  const int cb = C / 32;
  int t = 0;
  auto put = [&](int ky, int kx) {
    for (int c = 0; c < cb; ++c, ++t) {
      a.koff_x[t] = (uint32_t)(((ky * W + kx) * C + c * 32) * 4);
      a.koff_w[t] = (uint32_t)((((ky * KW + kx) * cb) + c) * 128);
    }
  };
  if (parity_order && st == 2) {
    for (int py = 0; py < 2; ++py) for (int px = 0; px < 2; ++px)
      for (int ky = py; ky < KH; ky += 2) for (int kx = px; kx < KW; kx += 2) put(ky, kx);
  } else {
    for (int ky = 0; ky < KH; ++ky) for (int kx = 0; kx < KW; ++kx) put(ky, kx);
  }
  a.nk = t;
}

static void run_conv(const char* name, int64_t n, int H, int W, int C, int KH, int KW, int st, int Cout, int64_t nt, bool parity) {
  const int OH = (H - KH) / st + 1, OW = (W - KW) / st + 1, K = KH * KW * C;
  printf("%s: conv fwd n=%ld %dx%dx%d k%d s%d -> %dx%dx%d (K=%d)%s\n", name, (long)n, H, W, C, KH, st, OH, OW, Cout, K, parity ? " parity-ordered taps" : "");
  const int64_t nx = n > nt ? n : nt;
  float* x = dalloc<float>(nx * H * W * C); float* w = dalloc<float>((int64_t)Cout * K); float* b = dalloc<float>(Cout);
  fill(x, nx * H * W * C, 11, 2.f, 1); fill(w, (int64_t)Cout * K, 12, 0.05f); fill(b, Cout, 13, 0.1f);
  float *sx, *sw; float* ax = absmax_of(x, nx * H * W * C); float* aw = absmax_of(w, (int64_t)Cout * K);
  float* xp = pack(x, nx * H * W, C, ax, &sx);
  float* wp = pack(w, (int64_t)Cout * KH * KW, C, aw, &sw);
  float* wn = dalloc<float>(1); rownorm1_kernel<<<Cout, 256>>>(w, Cout, K, wn);
  float* ab = absmax_of(b, Cout);
  const int64_t Mo = n * OH * OW, Mx = nx * OH * OW;
  float* y = dalloc<float>(Mx * Cout); float* yr = dalloc<float>(Mo * Cout); float* yu = dalloc<float>(Mo * Cout);
  float* osc = dalloc<float>(1); float* oam = dalloc<float>(1);
  uint32_t* mk = dalloc<uint32_t>(Mx * Cout / 32); uint32_t* mkr = dalloc<uint32_t>(Mo * Cout / 32);
  H2Args a = {};
  a.x = xp; a.w = wp; a.sx = sx; a.sw = sw; a.M = Mo; a.NC = Cout; a.w_row_bytes = K * 4; a.x_row_bytes = H * W * C * 4;
  a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.stride = st; a.KH = KH; a.KW = KW;
  fill_conv_koff(a, W, C, KH, KW, parity, st);
  a.bias = b; a.act = 1; a.out_fmt = H2O_H2P; a.out = y; a.out_row_bytes = Cout * 4;
  a.out_scale = osc; a.bound_in = ax; a.bound_w = wn; a.bound_b = ab; a.out_absmax = oam; a.mask_out = mk;
  launch<2, H2X_CONV>(a);
  CK(hipDeviceSynchronize());
  ref_conv<<<(unsigned)((Mo * Cout + 255) / 256), 256>>>(x, w, b, n, H, W, C, KH, KW, st, Cout, 1, yr);
  h2_unpack_kernel<<<2048, 256>>>((const uint8_t*)y, Mo, Cout, osc, yu, Cout);
  g_all_ok &= report("output (h2p) vs float64", yu, yr, Mo * Cout, 2e-6);
  mask_kernel<<<(unsigned)((Mo * Cout / 32 + 255) / 256), 256>>>(yr, Mo * Cout / 32, mkr);
  unsigned long long* md = dalloc<unsigned long long>(1); cmp_mask_kernel<<<512, 256>>>(mk, mkr, Mo * Cout / 32, md);
  unsigned long long mdh; CK(hipMemcpy(&mdh, md, 8, hipMemcpyDeviceToHost));
  float oamh, rmh, osch; float* rm = absmax_of(yr, Mo * Cout);
  CK(hipMemcpy(&oamh, oam, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&rmh, rm, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&osch, osc, 4, hipMemcpyDeviceToHost));
  printf("  mask bits differing %llu of %ld; absmax %.5g (ref %.5g); out scale %g\n", mdh, (long)(Mo * Cout), oamh, rmh, osch);
  if (nt > 0) {
    a.M = Mx;
    const double ms = time_ms([&] { launch<2, H2X_CONV>(a); }, 20);
    printf("  TIME n=%ld: %.1f us  (%.1f TFLOP/s float32-equivalent; %.2f TB/s of in+out bytes)\n", (long)nt, ms * 1e3,
           2.0 * Mx * Cout * K / ms * 1e-9, (nx * H * W * C * 4.0 + Mx * Cout * 4.0) / ms * 1e-9);
  }
  for (void* p : {(void*)x, (void*)w, (void*)b, (void*)xp, (void*)wp, (void*)y, (void*)yr, (void*)yu, (void*)mk, (void*)mkr}) CK(hipFree(p));
}

static void run_dgrad(const char* name, int64_t n, int H, int W, int C, int KH, int KW, int st, int Cout, int64_t nt, bool h2out) {
  const int OH = (H - KH) / st + 1, OW = (W - KW) / st + 1;
  const int TH = KH / st, TW = KW / st, Kg = TH * TW * Cout, NC = st * st * C;
  printf("%s: conv dgrad n=%ld dz %dx%dx%d -> dx %dx%dx%d k%d s%d (NC=%d, K<=%d) out=%s\n", name, (long)n, OH, OW, Cout, H, W, C, KH, st, NC, Kg, h2out ? "h2p" : "f32");
  const int64_t nx = n > nt ? n : nt;
  float* dz = dalloc<float>(nx * OH * OW * Cout); float* w = dalloc<float>((int64_t)Cout * KH * KW * C); float* act = dalloc<float>(nx * H * W * C);
  fill(dz, nx * OH * OW * Cout, 21, 1e-3f, 0); fill(w, (int64_t)Cout * KH * KW * C, 22, 0.05f); fill(act, nx * H * W * C, 23, 1.f, 1);
  float* wg = dalloc<float>((int64_t)NC * Kg);
  regroup_kernel<<<(NC * Kg + 255) / 256, 256>>>(w, C, KH, KW, st, Cout, wg);
  float *sx, *sw; float* ax = absmax_of(dz, nx * OH * OW * Cout); float* aw = absmax_of(wg, (int64_t)NC * Kg);
  float* xp = pack(dz, nx * OH * OW, Cout, ax, &sx);
  float* wp = pack(wg, (int64_t)NC * TH * TW, Cout, aw, &sw);
  float* wn = dalloc<float>(1); rownorm1_kernel<<<NC, 256>>>(wg, NC, Kg, wn);
  uint32_t* mk = dalloc<uint32_t>(nx * H * W * C / 32);
  mask_kernel<<<(unsigned)((nx * H * W * C / 32 + 255) / 256), 256>>>(act, nx * H * W * C / 32, mk);
  float* y = dalloc<float>(nx * H * W * C); float* yr = dalloc<float>(n * H * W * C); float* yu = dalloc<float>(n * H * W * C);
  float* osc = dalloc<float>(1); float* oam = dalloc<float>(1);
  H2Args a = {};
  a.x = xp; a.w = wp; a.sx = sx; a.sw = sw; a.M = n; a.NC = NC; a.w_row_bytes = Kg * 4; a.x_row_bytes = OH * OW * Cout * 4;
  a.C = Cout; a.OH = OH; a.OW = OW; a.stride = st; a.KH = KH; a.KW = KW; a.GH = H / st; a.GW = W / st; a.nk = TH * TW * (Cout / 32);
  a.out_fmt = h2out ? H2O_H2P : H2O_F32; a.out = y; a.out_row_bytes = H * W * C * 4; a.out_C = C; a.out_H = H; a.out_W = W;
  a.out_scale = osc; a.bound_in = ax; a.bound_w = wn; a.out_absmax = oam; a.mask_in = mk;
  auto go = [&] { if (NC % 128 == 0) launch<4, H2X_GROUPED>(a); else launch<2, H2X_GROUPED>(a); };
  go();
  CK(hipDeviceSynchronize());
  ref_dgrad<<<(unsigned)((n * H * W * C + 255) / 256), 256>>>(dz, w, act, n, H, W, C, KH, KW, st, Cout, yr);
  const float* got = y;
  if (h2out) { h2_unpack_kernel<<<2048, 256>>>((const uint8_t*)y, n * H * W, C, osc, yu, C); got = yu; }
  g_all_ok &= report("dx vs float64", got, yr, n * H * W * C, 2e-6);
  if (getenv("H2_PERIMG") && n <= 64) {
    std::vector<float> hg(n * H * W * C), hr(n * H * W * C);
    CK(hipMemcpy(hg.data(), got, hg.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), yr, hr.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t img = 0; img < n; ++img) {
      int bad = 0, first = -1;
      for (int i = 0; i < H * W * C; ++i) if (fabsf(hg[img * H * W * C + i] - hr[img * H * W * C + i]) > 1e-7f) { if (first < 0) first = i; ++bad; }
      printf("    image %ld: %d wrong of %d (first at pixel %d ch %d)\n", (long)img, bad, H * W * C, first / C, first % C);
    }
  }
  float oamh, rmh; float* rm = absmax_of(yr, n * H * W * C);
  CK(hipMemcpy(&oamh, oam, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&rmh, rm, 4, hipMemcpyDeviceToHost));
  printf("  absmax %.5g (ref %.5g)\n", oamh, rmh);
  if (nt > 0) {
    a.M = nx;
    const double ms = time_ms(go, 20);
    printf("  TIME n=%ld: %.1f us\n", (long)nt, ms * 1e3);
  }
  for (void* p : {(void*)dz, (void*)w, (void*)act, (void*)wg, (void*)xp, (void*)wp, (void*)y, (void*)yr, (void*)yu, (void*)mk}) CK(hipFree(p));
}

static float* pack_planar(const float* src, int64_t n, int H, int W, int C, int order, const float* amax, float** scale_out) {
  uint8_t* d = (uint8_t*)dalloc<float>(n * H * W * C);
  *scale_out = dalloc<float>(1);
  h2_pack_planar_kernel<<<2048, 256>>>(src, n, H, W, C, order, amax, nullptr, *scale_out, d);
  return (float*)d;
}

// image-stationary forward convolutions (ID = H2C_F2 / H2C_F3)
template <int ID> static void run_is_fwd(const char* name, int64_t n, int64_t nt) {
  constexpr bool F2 = ID == H2C_F2;
  const int H = F2 ? 20 : 9, W = H, C = F2 ? 32 : 64, KH = F2 ? 4 : 3, KW = KH, st = F2 ? 2 : 1, Cout = 64;
  const int OH = (H - KH) / st + 1, OW = OH, K = KH * KW * C;
  printf("%s: IS conv fwd n=%ld %dx%dx%d k%d s%d -> %dx%dx%d\n", name, (long)n, H, W, C, KH, st, OH, OW, Cout);
  const int64_t nx = n > nt ? n : nt;
  float* x = dalloc<float>(nx * H * W * C); float* w = dalloc<float>((int64_t)Cout * K); float* b = dalloc<float>(Cout);
  fill(x, nx * H * W * C, 31, 2.f, 1); fill(w, (int64_t)Cout * K, 32, 0.05f); fill(b, Cout, 33, 0.1f);
  float *sx, *sw; float* ax = absmax_of(x, nx * H * W * C); float* aw = absmax_of(w, (int64_t)Cout * K);
  float* xp;
  if (F2) { xp = dalloc<float>(nx * H * W * C); sx = dalloc<float>(1); h2_pack_pixrows_kernel<<<2048, 256>>>(x, nx, H, W, C, 2, ax, nullptr, sx, (uint8_t*)xp); }
  else xp = pack_planar(x, nx, H, W, C, 0, ax, &sx);
  float* wp = pack(w, (int64_t)Cout * KH * KW, C, aw, &sw);
  float* wn = dalloc<float>(1); rownorm1_kernel<<<Cout, 256>>>(w, Cout, K, wn);
  float* ab = absmax_of(b, Cout);
  const int64_t Mo = n * OH * OW, Mx = nx * OH * OW;
  float* y = dalloc<float>(Mx * Cout); float* yr = dalloc<float>(Mo * Cout); float* yu = dalloc<float>(Mo * Cout);
  float* osc = dalloc<float>(1); float* oam = dalloc<float>(1);
  uint32_t* mk = dalloc<uint32_t>(Mx * Cout / 32); uint32_t* mkr = dalloc<uint32_t>(Mo * Cout / 32);
  H2ConvArgs a = {};
  a.x = xp; a.w = wp; a.sx = sx; a.sw = sw; a.n = n; a.bias = b; a.act = 1; a.out = y; a.out_scale = osc;
  a.bound_in = ax; a.bound_w = wn; a.bound_b = ab; a.out_absmax = oam; a.mask_out = (uint8_t*)mk;
  h2conv_launch<ID>(0, a);
  CK(hipDeviceSynchronize());
  ref_conv<<<(unsigned)((Mo * Cout + 255) / 256), 256>>>(x, w, b, n, H, W, C, KH, KW, st, Cout, 1, yr);
  if (F2) h2_unpack_planar_kernel<<<2048, 256>>>((const uint8_t*)y, n, OH, OW, Cout, 0, osc, yu);
  else h2_unpack_kernel<<<2048, 256>>>((const uint8_t*)y, Mo, Cout, osc, yu, Cout);
  g_all_ok &= report("output (h2) vs float64", yu, yr, Mo * Cout, 2e-6);
  mask_h2_kernel<<<(unsigned)((Mo * Cout / 8 + 255) / 256), 256>>>(yr, Mo, Cout, (uint8_t*)mkr);
  unsigned long long* md = dalloc<unsigned long long>(1); cmp_mask_kernel<<<512, 256>>>(mk, mkr, Mo * Cout / 32, md);
  unsigned long long mdh; CK(hipMemcpy(&mdh, md, 8, hipMemcpyDeviceToHost));
  float oamh, rmh, osch; float* rm = absmax_of(yr, Mo * Cout);
  CK(hipMemcpy(&oamh, oam, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&rmh, rm, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&osch, osc, 4, hipMemcpyDeviceToHost));
  printf("  mask bits differing %llu of %ld; absmax %.5g (ref %.5g); out scale %g\n", mdh, (long)(Mo * Cout), oamh, rmh, osch);
  if (mdh > 8 || fabsf(oamh - rmh) > 1e-5f * rmh) g_all_ok = false;
  if (nt > 0) {
    a.n = nx;
    const double ms = time_ms([&] { h2conv_launch<ID>(0, a); }, 20);
    printf("  TIME n=%ld: %.1f us  (%.1f TFLOP/s float32-equivalent; %.2f TB/s of in+out bytes)\n", (long)nt, ms * 1e3,
           2.0 * Mx * Cout * K / ms * 1e-9, (nx * H * W * C * 4.0 + Mx * Cout * 4.0) / ms * 1e-9);
    if constexpr (ID == H2C_F2 || ID == H2C_F3) printf("  3 slots: %.1f us\n", 1e3 * time_ms([&] { h2conv_launch<ID, 3>(0, a); }, 20));
    printf("  128 workgroups: %.1f us\n", 1e3 * time_ms([&] { h2conv_launch<ID>(0, a, 128); }, 20));

  }
  for (void* p : {(void*)x, (void*)w, (void*)b, (void*)xp, (void*)wp, (void*)y, (void*)yr, (void*)yu, (void*)mk, (void*)mkr}) CK(hipFree(p));
}

// image-stationary data gradients (ID = H2C_D3 / H2C_D2)
template <int ID> static void run_is_dgrad(const char* name, int64_t n, int64_t nt) {
  constexpr bool D3 = ID == H2C_D3;
  const int H = D3 ? 9 : 20, W = H, C = D3 ? 64 : 32, KH = D3 ? 3 : 4, KW = KH, st = D3 ? 1 : 2, Cout = 64;
  const int OH = (H - KH) / st + 1, OW = OH;
  const int TH = KH / st, TW = KW / st, Kg = TH * TW * Cout, NC = st * st * C;
  printf("%s: IS conv dgrad n=%ld dz %dx%dx%d -> dx %dx%dx%d k%d s%d\n", name, (long)n, OH, OW, Cout, H, W, C, KH, st);
  const int64_t nx = n > nt ? n : nt;
  float* dz = dalloc<float>(nx * OH * OW * Cout); float* w = dalloc<float>((int64_t)Cout * KH * KW * C); float* act = dalloc<float>(nx * H * W * C);
  fill(dz, nx * OH * OW * Cout, 41, 1e-3f, 0); fill(w, (int64_t)Cout * KH * KW * C, 42, 0.05f); fill(act, nx * H * W * C, 43, 1.f, getenv("H2_ALLPOS") ? 2 : 1);
  float* wg = dalloc<float>((int64_t)NC * Kg);
  regroup_kernel<<<(NC * Kg + 255) / 256, 256>>>(w, C, KH, KW, st, Cout, wg);
  float *sx, *sw; float* ax = absmax_of(dz, nx * OH * OW * Cout); float* aw = absmax_of(wg, (int64_t)NC * Kg);
  float* xp = D3 ? pack(dz, nx * OH * OW, Cout, ax, &sx) : pack_planar(dz, nx, OH, OW, Cout, 0, ax, &sx);
  float* wp = pack(wg, (int64_t)NC * TH * TW, Cout, aw, &sw);
  float* wn = dalloc<float>(1); rownorm1_kernel<<<NC, 256>>>(wg, NC, Kg, wn);
  uint32_t* mk = dalloc<uint32_t>(nx * H * W * C / 32);
  mask_kernel<<<(unsigned)((nx * H * W * C / 32 + 255) / 256), 256>>>(act, nx * H * W * C / 32, mk);
  float* y = dalloc<float>(nx * H * W * C); float* yr = dalloc<float>(n * H * W * C); float* yu = dalloc<float>(n * H * W * C);
  float* osc = dalloc<float>(1); float* oam = dalloc<float>(1);
  H2ConvArgs a = {};
  a.x = xp; a.w = wp; a.sx = sx; a.sw = sw; a.n = n; a.out = y; a.out_scale = osc;
  if (D3) mask_h2_kernel<<<(unsigned)((nx * H * W * C / 8 + 255) / 256), 256>>>(act, nx * H * W, C, (uint8_t*)mk);
  a.bound_in = ax; a.bound_w = wn; a.out_absmax = oam; a.mask_in = mk;
  h2conv_launch<ID>(0, a);
  CK(hipDeviceSynchronize());
  ref_dgrad<<<(unsigned)((n * H * W * C + 255) / 256), 256>>>(dz, w, act, n, H, W, C, KH, KW, st, Cout, yr);
  const float* got = y;
  if (D3) { h2_unpack_planar_kernel<<<2048, 256>>>((const uint8_t*)y, n, H, W, C, 0, osc, yu); got = yu; }
  g_all_ok &= report("dx vs float64", got, yr, n * H * W * C, 2e-6);
  if (getenv("H2_PERIMG") && n <= 64) {
    std::vector<float> hg(n * H * W * C), hr(n * H * W * C);
    CK(hipMemcpy(hg.data(), got, hg.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), yr, hr.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t img = 0; img < n; ++img) {
      int bad = 0, first = -1;
      for (int i = 0; i < H * W * C; ++i) if (fabsf(hg[img * H * W * C + i] - hr[img * H * W * C + i]) > 1e-7f) { if (first < 0) first = i; ++bad; }
      printf("    image %ld: %d wrong of %d (first at pixel %d ch %d)\n", (long)img, bad, H * W * C, first / C, first % C);
    }
  }
  float oamh, rmh; float* rm = absmax_of(yr, n * H * W * C);
  CK(hipMemcpy(&oamh, oam, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&rmh, rm, 4, hipMemcpyDeviceToHost));
  printf("  absmax %.5g (ref %.5g)\n", oamh, rmh);
  if (nt > 0) {
    a.n = nx;
    const double ms = time_ms([&] { h2conv_launch<ID>(0, a); }, 20);
    printf("  TIME n=%ld: %.1f us  (%.2f TB/s of in+out bytes)\n", (long)nt, ms * 1e3, (nx * OH * OW * Cout * 4.0 + nx * H * W * C * 4.0) / ms * 1e-9);

  }
  for (void* p : {(void*)dz, (void*)w, (void*)act, (void*)wg, (void*)xp, (void*)wp, (void*)y, (void*)yr, (void*)yu, (void*)mk}) CK(hipFree(p));
}

// dw[co][ky][kx][ci] = sum_{img,oy,ox} dz[img,oy,ox,co] x[img, oy st + ky, ox st + kx, ci]; db[co] = sum dz
__global__ void ref_wgrad(const float* x, const float* dz, int64_t n, int H, int W, int C, int KH, int KW, int st, int Cout, float* dw, float* db) {
  const int OH = (H - KH) / st + 1, OW = (W - KW) / st + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int K = KH * KW * C;
  if (i >= Cout * K + Cout) return;
  double s = 0;
  if (i >= Cout * K) {
    const int co = i - Cout * K;
    for (int64_t r = 0; r < n * OH * OW; ++r) s += dz[r * Cout + co];
    db[co] = (float)s;
    return;
  }
  const int ci = i % C; int r = i / C; const int kx = r % KW; r /= KW; const int ky = r % KH; const int co = r / KH;
  for (int64_t img = 0; img < n; ++img)
    for (int oy = 0; oy < OH; ++oy) for (int ox = 0; ox < OW; ++ox)
      s += (double)dz[((img * OH + oy) * OW + ox) * Cout + co] * (double)x[((img * H + oy * st + ky) * W + ox * st + kx) * C + ci];
  dw[i] = (float)s;
}

template <int ID> static void run_is_wgrad(const char* name, int64_t n, int64_t nt) {
  constexpr bool C2 = ID == H2W_C2;
  const int H = C2 ? 20 : 9, W = H, C = C2 ? 32 : 64, KH = C2 ? 4 : 3, KW = KH, st = C2 ? 2 : 1, Cout = 64;
  const int OH = (H - KH) / st + 1, OW = OH, K = KH * KW * C;
  printf("%s: IS conv wgrad n=%ld x %dx%dx%d, dz %dx%dx%d -> dw [%d][%d]\n", name, (long)n, H, W, C, OH, OW, Cout, Cout, K);
  const int64_t nx = n > nt ? n : nt;
  float* x = dalloc<float>(nx * H * W * C); float* dz = dalloc<float>(nx * OH * OW * Cout);
  fill(x, nx * H * W * C, 51, 2.f, 1); fill(dz, nx * OH * OW * Cout, 52, 1e-3f, 0);
  float *sx, *sz; float* ax = absmax_of(x, nx * H * W * C); float* az = absmax_of(dz, nx * OH * OW * Cout);
  float* xp;
  if (C2) { xp = dalloc<float>(nx * H * W * C); sx = dalloc<float>(1); h2_pack_pixrows_kernel<<<2048, 256>>>(x, nx, H, W, C, 2, ax, nullptr, sx, (uint8_t*)xp); }
  else xp = pack_planar(x, nx, H, W, C, 0, ax, &sx);
  float* zp = C2 ? pack_planar(dz, nx, OH, OW, Cout, 0, az, &sz) : pack(dz, nx * OH * OW, Cout, az, &sz);
  const int grid = 256, per = Cout * K + Cout;
  float* slabs = dalloc<float>((int64_t)grid * per);
  float* gw = dalloc<float>(Cout * K); float* gb = dalloc<float>(Cout); float* rw = dalloc<float>(Cout * K); float* rb = dalloc<float>(Cout);
  H2WgradArgs a = {};
  a.x = xp; a.dz = zp; a.sx = sx; a.sz = sz; a.n = n; a.slabs = slabs;
  auto go = [&] { h2wgrad_launch<ID, C2 ? 2 : 3>(0, a, grid); };
  go();
  h2_wgrad_reduce_kernel<<<(per + 63) / 64, 256>>>(slabs, grid, per, Cout * K, gw, gb);
  CK(hipDeviceSynchronize());
  ref_wgrad<<<(per + 63) / 64, 64>>>(x, dz, n, H, W, C, KH, KW, st, Cout, rw, rb);
  g_all_ok &= report("dw vs float64", gw, rw, Cout * K, 2e-6);
  g_all_ok &= report("db vs float64", gb, rb, Cout, 2e-6);
  if (nt > 0) {
    a.n = nx;
    const double ms = time_ms(go, 20);
    printf("  TIME n=%ld: %.1f us  (%.2f TB/s of x + dz bytes)\n", (long)nt, ms * 1e3, (nx * H * W * C * 4.0 + nx * OH * OW * Cout * 4.0) / ms * 1e-9);
  }
  for (void* p : {(void*)x, (void*)dz, (void*)xp, (void*)zp, (void*)slabs}) CK(hipFree(p));
}

int main(int argc, char** argv) {
  const int64_t nc = argc > 1 ? atol(argv[1]) : 1000;
  const int64_t nt = argc > 2 ? atol(argv[2]) : 16384;
  g_stages = argc > 3 ? atoi(argv[3]) : 3;
  const std::string only = argc > 4 ? argv[4] : "";
  printf("h2_probe: check n=%ld, time n=%ld, stages=%d\n", (long)nc, (long)nt, g_stages);
  auto want = [&](const char* k) { return only.empty() || ("," + only + ",").find(std::string(",") + k + ",") != std::string::npos; };
  if (want("fc")) run_dense("FC forward", nc, 512, 3136, false, nt);
  if (want("fcd")) run_dense("FC dgrad-shaped", nc, 3136 - 3136 % 64, 512, true, nt);
  if (want("fcd4")) run_dense("FC dgrad-shaped, 128-channel tiles", nc, 3136, 512, true, nt, 4);
  if (want("fcd8")) run_dense("FC dgrad-shaped, 256-channel tiles unsplit", nc, 3136, 512, true, nt, 8);
  if (want("fc8")) run_dense("FC forward, 256-channel tiles unsplit", nc, 512, 3136, false, nt, 8);
  if (want("c2")) run_conv("conv2", nc, 20, 20, 32, 4, 4, 2, 64, nt, false);
  if (want("c2p")) run_conv("conv2", nc, 20, 20, 32, 4, 4, 2, 64, nt, true);
  if (want("c3")) run_conv("conv3", nc, 9, 9, 64, 3, 3, 1, 64, nt, false);
  if (want("d3")) run_dgrad("conv3", nc, 9, 9, 64, 3, 3, 1, 64, nt, true);
  if (want("d2")) run_dgrad("conv2", nc, 20, 20, 32, 4, 4, 2, 64, nt, false);
  if (want("if2")) run_is_fwd<H2C_F2>("conv2", nc, nt);
  if (want("if3")) run_is_fwd<H2C_F3>("conv3", nc, nt);
  if (want("id3")) run_is_dgrad<H2C_D3>("conv3", nc, nt);
  if (want("id2")) run_is_dgrad<H2C_D2>("conv2", nc, nt);
  if (want("iw3")) run_is_wgrad<H2W_C3>("conv3", nc > 256 ? 256 + 37 : nc, nt);
  if (want("iw2")) run_is_wgrad<H2W_C2>("conv2", nc > 256 ? 256 + 37 : nc, nt);
  printf(g_all_ok ? "ALL OK\n" : "SOME FAILED\n");
  return g_all_ok ? 0 : 1;
}
