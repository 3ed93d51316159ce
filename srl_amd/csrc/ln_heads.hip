// LayerNorm over the trunk's features and the heads that read it -- the tail of the reference's shared-backbone actor-critic
// (actor_critic_policy.py:117-140: features -> actor head logits, critic head value; the CNN base ends in a LayerNorm,
// modules/cnn.py + modules/utils.py:154-161) -- in ONE launch per direction (srl_ln_heads_fwd / srl_ln_heads_bwd).
//
// Layer by layer this tail was, per 16 384-row chunk of the Atari update, a LayerNorm kernel and two skinny products forward
// (54 us) and two skinny data gradients, two skinny weight gradients and the LayerNorm's backward (116 us): nine passes over the
// [n, 512] features for 14 flops per byte.  Here a wavefront takes a row at a time: its 512 features are 8 registers per lane, the
// statistics and the heads' 7 dot products are wave reductions, and the normalised features are never written -- the backward pass
// forms them again from x and the stored (mean, rstd).  Forward: x is read once (33.5 MB).  Backward: x read, dx written, and
// every parameter gradient of the LayerNorm and the heads summed in registers over the rows a wavefront walks, met in LDS per
// workgroup, one atomic per parameter and workgroup at the end.
#include "srl_common.h"

#include "../../include/srl_hip.h"

namespace {

constexpr float kEps = 1e-5f;       // nn.LayerNorm default
constexpr int kMaxOut = SRL_LN_HEADS_MAX_OUT;   // outputs of all heads together at most
constexpr int kWaves = 8;           // wavefronts per workgroup
constexpr int kDbGroups = 8;        // workgroups that form the heads' bias gradients

struct LhArgs {
  const float* x;
  long ldx, n;
  const float *gamma, *beta;
  const float* W[2];      // head weights [A, D] row-major
  const float* b[2];
  int A[2];
  float* y[2];
  long ldy[2];
  float *mean, *rstd;
  // backward
  const float* dy[2];
  long lddy[2];
  int in_act;
  float* dx;
  long lddx;
  float *dgamma, *dbeta;
  float* dW[2];
  float* db[2];
  float* dx_absmax;
  // forward: x = act(sum of x_slabs slabs (x_slab_stride floats apart) + x_bias) -- the raw partial sums of a split product
  // (srl_h2_gemm_splitk) finished while they are read
  int x_slabs;
  long x_slab_stride;
  const float* x_bias;
  int x_act;
  float* x_out;      // (optional) the finished x is also written here [n, ldxo]: what a backward pass will read
  long ldxo;
  int dbg;   // timing experiments (wrong results; SRL_LNH_DBG): 1 no final atomics, 2 no LDS meeting either, 4 no row loop
};

__device__ __forceinline__ float allsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// the sums over the wavefront of 8 values per lane with 10 exchanges instead of 48: three halving steps (a lane keeps half of its
// values and takes the partner's partial sums of them), then three plain steps.  Afterwards every lane holds the total of value
// 4 * bit0 + 2 * bit1 + bit2 of its lane number.
__device__ __forceinline__ float allsum8(const float (&v)[8], int lane) {
  float k4[4], k2[2], k1;
  {
    const bool hi = lane & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float send = hi ? v[i] : v[i + 4];
      k4[i] = (hi ? v[i + 4] : v[i]) + __shfl_xor(send, 1, 64);
    }
  }
  {
    const bool hi = lane & 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float send = hi ? k4[i] : k4[i + 2];
      k2[i] = (hi ? k4[i + 2] : k4[i]) + __shfl_xor(send, 2, 64);
    }
  }
  {
    const bool hi = lane & 4;
    const float send = hi ? k2[0] : k2[1];
    k1 = (hi ? k2[1] : k2[0]) + __shfl_xor(send, 4, 64);
  }
  k1 += __shfl_xor(k1, 8, 64);
  k1 += __shfl_xor(k1, 16, 64);
  k1 += __shfl_xor(k1, 32, 64);
  return k1;
}
__device__ __forceinline__ int slot8(int lane) { return 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1); }

// lane l holds channels 256 j + 4 l + (0..3), j < NV / 4: float4 accesses, a wavefront reads whole kilobytes
template <int NV>
__device__ __forceinline__ void load_row(const float* p, int lane, float (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV / 4; ++j) {
    const float4 q = *reinterpret_cast<const float4*>(p + 256 * j + 4 * lane);
    v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
  }
}
template <int NV>
__device__ __forceinline__ void store_row(float* p, int lane, const float (&v)[NV]) {
#pragma unroll
  for (int j = 0; j < NV / 4; ++j)
    *reinterpret_cast<float4*>(p + 256 * j + 4 * lane) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

template <int NV>
__global__ __launch_bounds__(64 * kWaves) void ln_heads_fwd_kernel(LhArgs a) {
  constexpr int D = 64 * NV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int AT = a.A[0] + a.A[1];
  float g[NV], be[NV], w[kMaxOut][NV];
  load_row<NV>(a.gamma, lane, g);
  load_row<NV>(a.beta, lane, be);
#pragma unroll
  for (int o = 0; o < kMaxOut; ++o) {
    if (o < AT) load_row<NV>(o < a.A[0] ? a.W[0] + (long)o * D : a.W[1] + (long)(o - a.A[0]) * D, lane, w[o]);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) w[o][k] = 0.f;
    }
  }
  const int so = slot8(lane);
  const float bias = so < AT ? (so < a.A[0] ? (a.b[0] ? a.b[0][so] : 0.f) : (a.b[1] ? a.b[1][so - a.A[0]] : 0.f)) : 0.f;
  float xb[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) xb[k] = 0.f;
  if (a.x_bias) load_row<NV>(a.x_bias, lane, xb);
  for (long row = (long)blockIdx.x * kWaves + wave; row < a.n; row += (long)gridDim.x * kWaves) {
    float x[NV];
    load_row<NV>(a.x + row * a.ldx, lane, x);
    if (a.x_slabs > 1 || a.x_bias || a.x_act) {
      for (int sl = 1; sl < a.x_slabs; ++sl) {
        float t[NV];
        load_row<NV>(a.x + sl * a.x_slab_stride + row * a.ldx, lane, t);
#pragma unroll
        for (int k = 0; k < NV; ++k) x[k] += t[k];
      }
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        x[k] += xb[k];
        x[k] = a.x_act == 1 ? fmaxf(x[k], 0.f) : (a.x_act == 2 ? tanhf(x[k]) : x[k]);
      }
      if (a.x_out) store_row<NV>(a.x_out + row * a.ldxo, lane, x);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += x[k];
    const float mean = allsum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      x[k] -= mean;
      q = fmaf(x[k], x[k], q);
    }
    const float rstd = rsqrtf(allsum(q) / (float)D + kEps);
    float p[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) p[o] = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float f = fmaf(x[k] * rstd, g[k], be[k]);
#pragma unroll
      for (int o = 0; o < kMaxOut; ++o) p[o] = fmaf(f, w[o][k], p[o]);
    }
    const float tot = allsum8(p, lane) + bias;
    if (lane < 8 && so < AT) {   // lanes 0..7 hold the 8 slots
      if (so < a.A[0]) a.y[0][row * a.ldy[0] + so] = tot;
      else a.y[1][row * a.ldy[1] + so - a.A[0]] = tot;
    }
    if (lane == 0) {
      a.mean[row] = mean;
      a.rstd[row] = rstd;
    }
  }
}

// Backward.  LDS: D floats per wavefront, for the parameter sums at the end.
template <int NV>
__global__ __launch_bounds__(64 * kWaves) void ln_heads_bwd_kernel(LhArgs a) {
  constexpr int D = 64 * NV;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
  const int A0 = a.A[0], AT = a.A[0] + a.A[1];
  float g[NV], be[NV], w[kMaxOut][NV];
  load_row<NV>(a.gamma, lane, g);
  load_row<NV>(a.beta, lane, be);
#pragma unroll
  for (int o = 0; o < kMaxOut; ++o) {
    if (o < AT) load_row<NV>(o < A0 ? a.W[0] + (long)o * D : a.W[1] + (long)(o - A0) * D, lane, w[o]);
    else {
#pragma unroll
      for (int k = 0; k < NV; ++k) w[o][k] = 0.f;
    }
  }
  float accw[kMaxOut][NV], accg[NV], accb[NV];   // dW, dgamma, dbeta of this lane's channels
#pragma unroll
  for (int k = 0; k < NV; ++k) accg[k] = accb[k] = 0.f;
#pragma unroll
  for (int o = 0; o < kMaxOut; ++o) {
#pragma unroll
    for (int k = 0; k < NV; ++k) accw[o][k] = 0.f;
  }
  float amax = 0.f;
  for (long row = (long)blockIdx.x * kWaves + wave; row < ((a.dbg & 4) ? 0 : a.n); row += (long)gridDim.x * kWaves) {
    float x[NV];
    load_row<NV>(a.x + row * a.ldx, lane, x);
    const float mean = a.mean[row], rstd = a.rstd[row];
    float dyv[kMaxOut];   // the row's head gradients (the same in every lane)
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o)
      dyv[o] = o < AT ? (o < A0 ? a.dy[0][row * a.lddy[0] + o] : a.dy[1][row * a.lddy[1] + o - A0]) : 0.f;
    float xh[NV], gg[NV];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      xh[k] = (x[k] - mean) * rstd;
      const float f = fmaf(xh[k], g[k], be[k]);
      float df = 0.f;
#pragma unroll
      for (int o = 0; o < kMaxOut; ++o) {
        df = fmaf(dyv[o], w[o][k], df);
        accw[o][k] = fmaf(dyv[o], f, accw[o][k]);
      }
      accb[k] += df;
      accg[k] = fmaf(df, xh[k], accg[k]);
      gg[k] = df * g[k];
      m1 += gg[k];
      m2 = fmaf(gg[k], xh[k], m2);
    }
    m1 = allsum(m1) / (float)D;
    m2 = allsum(m2) / (float)D;
    float dx[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float der = a.in_act == 1 ? (x[k] > 0.f ? 1.f : 0.f) : (a.in_act == 2 ? 1.f - x[k] * x[k] : 1.f);
      dx[k] = rstd * (gg[k] - m1 - xh[k] * m2) * der;
      amax = fmaxf(amax, fabsf(dx[k]));
    }
    store_row<NV>(a.dx + row * a.lddx, lane, dx);
  }
  if (a.dx_absmax) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0 && amax > 0.f) atomicMax(reinterpret_cast<int*>(a.dx_absmax), __float_as_int(amax));
  }
  // every wavefront parks its sums in a slot of its own (plain stores), then the workgroup's threads add the slots and send
  // one atomic per parameter
  if (a.dbg & 2) return;   // (uniform)
  // The workgroup's sums meet in LDS one PLANE (one head output's dW row set, dgamma, dbeta) at a time: every wavefront parks its D
  // sums of the plane (plain stores), the threads add the eight slots and send one atomic per parameter.  (All planes at once
  // were 147 KB of LDS per workgroup: inside the update such a workgroup waits for a CU that a persistent kernel of another
  // row-chunk pipeline has left entirely.)
#pragma unroll
  for (int o = 0; o < kMaxOut + 2; ++o) {
    if (o < kMaxOut && o >= AT) continue;   // (uniform)
    float* mine = sm + wave * D;
    if (o < kMaxOut) store_row<NV>(mine, lane, accw[o < kMaxOut ? o : 0]);
    else if (o == kMaxOut) store_row<NV>(mine, lane, accg);
    else store_row<NV>(mine, lane, accb);
    __syncthreads();
    for (int c = tid; c < D; c += 64 * kWaves) {
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < kWaves; ++wv) sum += sm[wv * D + c];
      if (a.dbg & 1) continue;
      float* dst = o == kMaxOut ? a.dgamma + c : (o == kMaxOut + 1 ? a.dbeta + c : (o < A0 ? a.dW[0] + (long)o * D + c : a.dW[1] + (long)(o - A0) * D + c));
      atomicAdd(dst, sum);
    }
    __syncthreads();
  }
  // bias gradients = column sums of dy.  NOT one atomic per wavefront from the sums the row loop could keep: 2048 atomics on one
  // address take ~80 ns each, one after the other (160 us of a 17 us kernel).  The first kDbGroups workgroups sum a slice of dy's
  // rows each -- 460 KB in all -- and send one atomic per output.
  const int G = gridDim.x < kDbGroups ? (int)gridDim.x : kDbGroups;
  if ((int)blockIdx.x < G && (a.db[0] || a.db[1])) {
    __syncthreads();   // the slots above are read
    float s[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) s[o] = 0.f;
    const long r0 = a.n * blockIdx.x / G, r1 = a.n * (blockIdx.x + 1) / G;
    for (long row = r0 + tid; row < r1; row += 64 * kWaves) {
#pragma unroll
      for (int o = 0; o < kMaxOut; ++o)
        if (o < AT) s[o] += o < A0 ? a.dy[0][row * a.lddy[0] + o] : a.dy[1][row * a.lddy[1] + o - A0];
    }
    const float tot = allsum8(s, lane);   // lanes 0..7: the wavefront's total of slot8(lane)
    if (lane < 8) sm[wave * 8 + slot8(lane)] = tot;
    __syncthreads();
    if (tid < AT) {
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < kWaves; ++wv) v += sm[wv * 8 + tid];
      float* dst = tid < A0 ? a.db[0] : a.db[1];
      if (dst) atomicAdd(dst + (tid < A0 ? tid : tid - A0), v);
    }
  }
}

int check_common(const float* x, int64_t ldx, int D, const float* gamma, const float* beta, const float* const* W, const int32_t* A,
                 int nh) {
  if (!x || !gamma || !beta || !W || !A || nh < 1 || nh > 2) return -1;
  if (!(D == 256 || D == 512 || D == 1024) || ldx % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15)) return -1;
  int at = 0;
  for (int h = 0; h < nh; ++h) {
    if (!W[h] || A[h] < 1 || (reinterpret_cast<uintptr_t>(W[h]) & 15)) return -1;
    at += A[h];
  }
  if ((reinterpret_cast<uintptr_t>(gamma) & 15) || (reinterpret_cast<uintptr_t>(beta) & 15)) return -1;
  return at <= kMaxOut ? at : -1;
}

}  // namespace

static long bwd_lds_bytes(int D, int at);
extern "C" int srl_ln_heads_supported(int D, int n_heads, const int32_t* head_dims) {
  if (!(D == 256 || D == 512 || D == 1024) || n_heads < 1 || n_heads > 2 || !head_dims) return 0;
  int at = 0;
  for (int h = 0; h < n_heads; ++h) {
    if (head_dims[h] < 1) return 0;
    at += head_dims[h];
  }
  return at <= kMaxOut && bwd_lds_bytes(D, at) <= 160 * 1024;
}

extern "C" int srl_ln_heads_fwd(void* stream, const float* x, int64_t ldx, int64_t n, int D, const float* gamma, const float* beta,
                                int n_heads, const float* const* W, const float* const* b, const int32_t* head_dims,
                                float* const* y, const int64_t* ldy, float* mean, float* rstd, int x_slabs, int64_t x_slab_stride,
                                const float* x_bias, int x_act, float* x_out, int64_t ldxo) {
  SRL_CHECK_ARG(check_common(x, ldx, D, gamma, beta, W, head_dims, n_heads) > 0,
                "unsupported (D in 256 | 512 | 1024, 1-2 heads, <= SRL_LN_HEADS_MAX_OUT outputs, 16-byte aligned rows)");
  SRL_CHECK_ARG(y && ldy && mean && rstd && n >= 0, "null output");
  SRL_CHECK_ARG(x_slabs >= 1 && (x_slabs == 1 || x_slab_stride % 4 == 0) && x_act >= 0 && x_act <= 2 &&
                (!x_bias || (reinterpret_cast<uintptr_t>(x_bias) & 15) == 0), "x_slabs >= 1, aligned slabs / bias, x_act 0..2");
  if (n == 0) return 0;
  LhArgs a{};
  a.x = x; a.ldx = ldx; a.n = n; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
  a.x_slabs = x_slabs; a.x_slab_stride = x_slab_stride; a.x_bias = x_bias; a.x_act = x_act;
  SRL_CHECK_ARG(!x_out || ((x_slabs > 1 || x_bias || x_act) && ldxo % 4 == 0 && (reinterpret_cast<uintptr_t>(x_out) & 15) == 0),
                "x_out: only with a finishing step (slabs / bias / activation), 16-byte aligned rows");
  a.x_out = x_out; a.ldxo = ldxo;
  for (int h = 0; h < n_heads; ++h) {
    SRL_CHECK_ARG(y[h] && ldy[h] >= head_dims[h], "null head output / short rows");
    a.W[h] = W[h]; a.b[h] = b ? b[h] : nullptr; a.A[h] = head_dims[h]; a.y[h] = y[h]; a.ldy[h] = ldy[h];
  }
  const long groups = srl_ceil_div(n, (long)kWaves);
  const dim3 grid((unsigned)(groups < 1024 ? groups : 1024)), block(64 * kWaves);
  hipStream_t st = (hipStream_t)stream;
  if (D == 256) hipLaunchKernelGGL(ln_heads_fwd_kernel<4>, grid, block, 0, st, a);
  else if (D == 512) hipLaunchKernelGGL(ln_heads_fwd_kernel<8>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(ln_heads_fwd_kernel<16>, grid, block, 0, st, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

static long bwd_lds_bytes(int D, int at) { (void)at; return (long)kWaves * D * 4; }

template <int NV>
static void launch_bwd(const LhArgs& a, hipStream_t st) {
  const int lds = (int)bwd_lds_bytes(64 * NV, a.A[0] + a.A[1]);
  static int attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_heads_bwd_kernel<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = lds;
  }
  const long groups = srl_ceil_div(a.n, (long)kWaves);
  hipLaunchKernelGGL(ln_heads_bwd_kernel<NV>, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kWaves), lds, st, a);
}

extern "C" int srl_ln_heads_bwd(void* stream, const float* x, int64_t ldx, int64_t n, int D, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, int n_heads, const float* const* W,
                                const int32_t* head_dims, const float* const* dy, const int64_t* lddy, int in_act, float* dx,
                                int64_t lddx, float* dgamma, float* dbeta, float* const* dW, float* const* db,
                                float* dx_absmax) {
  SRL_CHECK_ARG(check_common(x, ldx, D, gamma, beta, W, head_dims, n_heads) > 0, "unsupported (see srl_ln_heads_fwd)");
  SRL_CHECK_ARG(mean && rstd && dy && lddy && dx && dgamma && dbeta && dW && n >= 0, "null tensor");
  SRL_CHECK_ARG(lddx % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && in_act >= 0 && in_act <= 2, "dx rows / activation");
  {
    int at = 0;
    for (int h = 0; h < n_heads; ++h) at += head_dims[h];
    SRL_CHECK_ARG(bwd_lds_bytes(D, at) <= 160 * 1024, "the heads' outputs times D do not fit the backward kernel's LDS");
  }
  if (n == 0) return 0;
  LhArgs a{};
  a.x = x; a.ldx = ldx; a.n = n; a.gamma = gamma; a.beta = beta; a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd);
  a.in_act = in_act; a.dx = dx; a.lddx = lddx; a.dgamma = dgamma; a.dbeta = dbeta; a.dx_absmax = dx_absmax;
  for (int h = 0; h < n_heads; ++h) {
    SRL_CHECK_ARG(dy[h] && lddy[h] >= head_dims[h] && dW[h], "null head gradient");
    a.W[h] = W[h]; a.A[h] = head_dims[h]; a.dy[h] = dy[h]; a.lddy[h] = lddy[h]; a.dW[h] = dW[h]; a.db[h] = db ? db[h] : nullptr;
  }
  static const int dbg = [] { const char* e = getenv("SRL_LNH_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  hipStream_t st = (hipStream_t)stream;
  if (D == 256) launch_bwd<4>(a, st);
  else if (D == 512) launch_bwd<8>(a, st);
  else launch_bwd<16>(a, st);
  SRL_LAUNCH_CHECK();
  return 0;
}
