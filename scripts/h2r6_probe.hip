// Round-6 standalone probe: the TN weight-gradient kernel (csrc/h2tn.h) and the wide-tile NT kernel (csrc/h2g16.h) against
// float64 references, with timings at the shapes of the benchmarked network.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o scripts/h2r6_probe scripts/h2r6_probe.hip ; run on the GPU box
//   h2r6_probe [what]      what: tn | nt | all (default all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../srl_amd/csrc/h2tn.h"
#ifdef HAVE_G16
#include "../srl_amd/csrc/h2gemmp.h"
#endif
using namespace srlh2;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return (uint32_t)x;
}
// kind 0: uniform [-amp, amp); 1: relu-like (half zeros, rest squared)
__global__ void fill_kernel(float* p, int64_t n, uint64_t seed, float amp, int kind) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = hash32(seed * 0x9e3779b97f4a7c15ull + i);
    float u = (float)(h >> 8) * (1.f / 16777216.f) * 2.f - 1.f;
    if (kind == 1) u = u < 0.f ? 0.f : u * u * 3.f;
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
// dW[o][c] = sum_m a[m][o] b[m][c] for a sampled set of outputs: thread t -> output idx[t]
__global__ void ref_tn(const float* a, const float* b, int64_t M, int NA, int NB, const int64_t* idx, int64_t nidx, float* out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nidx) return;
  const int64_t e = idx[t];
  const int o = (int)(e / NB), c = (int)(e % NB);
  double s = 0;
  for (int64_t m = 0; m < M; ++m) s += (double)a[m * NA + o] * (double)b[m * NB + c];
  out[t] = (float)s;
}
__global__ void gather_kernel(const float* src, const int64_t* idx, int64_t n, float* out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = src[idx[t]];
}
__global__ void ref_dense(const float* x, const float* w, const float* bias, int64_t M, int NC, int K, int relu, const int64_t* idx, int64_t nidx, float* y) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nidx) return;
  const int64_t i = idx[t];
  const int64_t m = i / NC; const int c = (int)(i % NC);
  double s = 0;
  for (int k = 0; k < K; ++k) s += (double)x[m * K + k] * (double)w[(int64_t)c * K + k];
  if (bias) s += bias[c];
  if (relu && s < 0) s = 0;
  y[t] = (float)s;
}
__global__ void sum_slabs(const float* slabs, int n, int64_t per, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < n; ++z) s += slabs[(int64_t)z * per + i];
    out[i] = s;
  }
}
__global__ void cmp_kernel(const float* a, const float* b, int64_t n, double* out) {
  double md = 0, mr = 0, sd = 0, sr = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = fabs((double)a[i] - (double)b[i]);
    md = fmax(md, d); mr = fmax(mr, fabs((double)b[i])); sd += d * d; sr += (double)b[i] * b[i];
  }
  atomicMax((unsigned long long*)&out[0], (unsigned long long)__double_as_longlong(md));
  atomicMax((unsigned long long*)&out[1], (unsigned long long)__double_as_longlong(mr));
  atomicAdd(&out[2], sd); atomicAdd(&out[3], sr);
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
static bool g_all_ok = true;
static bool report(const char* what, const float* got, const float* ref, int64_t n, double tol) {
  double* o = dalloc<double>(4);
  cmp_kernel<<<256, 256>>>(got, ref, n, o);
  double h[4]; CK(hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost)); CK(hipFree(o));
  const double rel_max = h[0] / (h[1] > 0 ? h[1] : 1), rel_rms = sqrt(h[2] / (h[3] > 0 ? h[3] : 1));
  const bool ok = rel_max < tol && rel_rms == rel_rms && h[1] > 0;
  printf("  %-40s max|d|/max|ref| %.3e  rms rel %.3e  (max|ref| %.4g) %s\n", what, rel_max, rel_rms, h[1], ok ? "ok" : "FAIL");
  g_all_ok &= ok;
  return ok;
}
template <class F> static double time_us(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1e3;
}
static int64_t* sample_idx(int64_t total, int64_t n, uint64_t seed) {
  std::vector<int64_t> h(n);
  uint64_t s = seed * 0x9e3779b97f4a7c15ull + 12345;
  for (int64_t i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; h[i] = (int64_t)((s >> 17) % (uint64_t)total); }
  // always include the corners
  h[0] = 0; h[1] = total - 1;
  int64_t* d = dalloc<int64_t>(n);
  CK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
  return d;
}

static void run_tn(const char* name, int64_t M, int NA, int NB, bool timeit, int dbg = 0) {
  printf("%s: TN  dW[%d, %d] = sum over %ld rows\n", name, NA, NB, (long)M);
  float* a = dalloc<float>(M * NA); float* b = dalloc<float>(M * NB);
  fill(a, M * NA, 41, 1e-3f, 0); fill(b, M * NB, 42, 2.f, 1);
  float *sa, *sb; float* aa = absmax_of(a, M * NA); float* ab = absmax_of(b, M * NB);
  float* ap = pack(a, M, NA, aa, &sa); float* bp = pack(b, M, NB, ab, &sb);
  H2TnArgs g = {};
  g.a = ap; g.b = bp; g.sa = sa; g.sb = sb; g.M = M; g.NA = NA; g.NB = NB; g.a_row_bytes = (int64_t)NA * 4; g.b_row_bytes = (int64_t)NB * 4;
  const int tiles = ((NA + 255) / 256) * ((NB + 255) / 256);
  h2tn_plan(M, tiles, &g.splits, &g.rows_per_split);
  const int64_t per = (int64_t)NA * NB;
  float* slabs = dalloc<float>(per * g.splits); float* out = dalloc<float>(per);
  g.slabs = slabs;
  auto go = [&] { switch (dbg) { case 1: h2tn_launch<1>(0, g); break; case 2: h2tn_launch<2>(0, g); break; case 3: h2tn_launch<3>(0, g); break; case 4: h2tn_launch<4>(0, g); break; default: h2tn_launch<0>(0, g); } };
  printf("  tiles %d, row ranges %d x %ld rows, slabs %.1f MB\n", tiles, g.splits, (long)g.rows_per_split, per * g.splits * 4e-6);
  CK(hipMemset(slabs, 0xff, per * g.splits * 4));   // NaN: every slab element must be written
  go();
  sum_slabs<<<2048, 256>>>(slabs, g.splits, per, out);
  CK(hipDeviceSynchronize());
  const int64_t ns = 8192;
  int64_t* idx = sample_idx(per, ns, 7);
  float* ref = dalloc<float>(ns); float* got = dalloc<float>(ns);
  ref_tn<<<(unsigned)((ns + 63) / 64), 64>>>(a, b, M, NA, NB, idx, ns, ref);
  gather_kernel<<<(unsigned)((ns + 255) / 256), 256>>>(out, idx, ns, got);
  if (!dbg) report("sampled outputs vs float64", got, ref, ns, 2e-6);
  if (timeit) {
    if (getenv("H2TN_ROW0")) {   // diagnostic: every k-step re-reads the same rows (L2-resident operands)
      H2TnArgs keep = g;
      g.a_row_bytes = 0; g.b_row_bytes = 0;
      printf("  TIME with every row at offset 0 (operands from L2): %.1f us\n", time_us(go, 20));
      g.b_row_bytes = keep.b_row_bytes;
      printf("  TIME with A's rows at offset 0: %.1f us\n", time_us(go, 20));
      g = keep;
    }
    const double us = time_us(go, 20);
    const double us2 = time_us([&] { sum_slabs<<<2048, 256>>>(slabs, g.splits, per, out); }, 20);
    printf("  TIME: kernel %.1f us (%.1f TFLOP/s float32-equivalent), slab sum %.1f us\n", us, 2.0 * M * NA * NB / us * 1e-6, us2);
  }
  for (void* p : {(void*)a, (void*)b, (void*)ap, (void*)bp, (void*)slabs, (void*)out, (void*)idx, (void*)ref, (void*)got}) CK(hipFree(p));
}

#ifdef HAVE_G16
#include "h2r6_probe_nt.inc"
#endif

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "all";
  const bool tn = !strcmp(what, "all") || !strcmp(what, "tn"), nt = !strcmp(what, "all") || !strcmp(what, "nt");
  if (tn) {
    run_tn("small ragged", 1000, 96, 160, false);
    run_tn("one tile, 37 rows", 37, 256, 256, false);
    run_tn("narrow tiles", 2085, 512, 3136, false);
    run_tn("Atari Linear wgrad", 16384, 512, 3136, true);
    if (argc > 2) {
      for (int d : {1, 2, 4, 3}) { printf("-- leave-out %d\n", d); run_tn("Atari Linear wgrad", 16384, 512, 3136, true, d); }
    }
    run_tn("football tower wgrad (1/8 of the rows)", 6400, 11264, 22528, true);
  }
#ifdef HAVE_G16
  if (nt) run_nt_all(argc > 2);
#endif
  printf(g_all_ok ? "ALL OK\n" : "FAILURES\n");
  return g_all_ok ? 0 : 1;
}
