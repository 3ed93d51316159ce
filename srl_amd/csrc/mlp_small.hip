// A whole small MLP chain -- LayerNorm / Linear (+ ReLU / tanh) layers no wider than 128 -- in ONE launch forward and ONE launch
// backward (srl_mlp_fwd / srl_mlp_bwd).
//
// The CartPole-sized configurations (BASELINE configs[0]: 8 envs x 32 steps, separate 2 x 64 MLPs) are not launch-bound behind
// the captured graph any more; they are LATENCY-bound: 38 kernels of 3-10 us, each a chain of one or two dependent trips to
// memory for a few kilobytes of work.  Here a workgroup takes 16 rows through every layer: the rows live in LDS, a layer's
// weights are staged into LDS once per workgroup (they are L2-resident: every workgroup reads the same 16 KB), and the only
// global traffic besides x and y is the tape -- every layer's input, which the backward pass reads back.  The backward pass
// walks the same rows through the layers in reverse: weight / bias / LayerNorm-affine gradients are summed over the
// workgroup's 16 rows and added to the gradient buffer with atomics, the data gradient stays in LDS.
// Arithmetic: plain float32 FMAs (the layers are far too small for the matrix cores to matter), sums in a fixed order
// inside a workgroup; across workgroups the atomics' order is not fixed, like the LayerNorm backward kernels' already.
// Reference: the nn.Sequential of modules/utils.py:154-161 (mlp: LayerNorm -> Linear -> activation ...) and the heads of
// actor_critic_policy.py:92-107, as one kernel per direction.
#include "srl_common.h"

#include "../../include/srl_hip.h"

namespace {

constexpr int kRB = 16;     // rows per workgroup (16 lanes per row)
constexpr int kMaxD = 128;  // widest layer
constexpr int kLd = kMaxD + 1;
constexpr float kLnEps = 1e-5f;  // nn.LayerNorm default
constexpr int kMaxP = 12288;     // parameter-gradient floats a backward workgroup can keep in LDS (48 KB)
constexpr int kBwdGroups = 512;  // backward workgroups at most: each walks its 16-row blocks and flushes its sums once

struct Layer {
  int kind, in, out, act;  // kind 0: LayerNorm over `in` (w = gamma, b = beta); 1: Linear [out][in] (+ act: 1 relu, 2 tanh)
  const float* w;
  const float* b;
  float* gw;
  float* gb;
  int toff;  // offset of this layer's INPUT inside a tape row (layer 0 reads x itself)
  int pw, pb;  // offsets of this layer's weight / bias gradient sums in the backward kernel's LDS accumulator
};

struct Args {
  Layer L[SRL_MLP_MAX_LAYERS];
  int n;
  const float* x;
  long ldx, rows;
  float* tape;
  long tld;
  float* y;
  long ldy;
  const float* dy;
  long lddy;
  float* dx;   // backward, matrix-core chains only: d loss / d x [rows, lddx] is stored here (null: not formed)
  long lddx;
  int lds_acc;  // backward: parameter gradients summed in LDS over the workgroup's row blocks (they fit: <= kMaxP floats)
};

__device__ __forceinline__ float sum16(float v) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
  return v;
}
__device__ __forceinline__ float act_fwd(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? tanhf(v) : v); }
// derivative of the activation from its OUTPUT value
__device__ __forceinline__ float act_der(float y, int act) { return act == 1 ? (y > 0.f ? 1.f : 0.f) : (act == 2 ? 1.f - y * y : 1.f); }

// out[r][c + 16 j] = bias + sum_k cur[r][k] * Wt[k][c + 16 j], JN column groups per lane
template <int JN>
__device__ __forceinline__ void linear_rows(const float (*cur)[kLd], const float* Wt, const Layer& L, int r, int c, float* res) {
  float acc[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) acc[j] = (L.b && c + 16 * j < L.out) ? L.b[c + 16 * j] : 0.f;
  // (unrolled: with one or two workgroups per CU nothing else hides the LDS latency of a 5-read iteration)
#pragma unroll 8
  for (int k = 0; k < L.in; ++k) {
    const float xv = cur[r][k];
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j] = fmaf(xv, Wt[k * kLd + c + 16 * j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) res[j] = acc[j];
}

__global__ __launch_bounds__(256) void mlp_fwd_kernel(Args a) {
  __shared__ float cur[kRB][kLd];
  __shared__ float Wt[kMaxD * kLd];  // a Linear's weights, transposed: Wt[k][o]
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const long row = (long)blockIdx.x * kRB + r;
  const bool rok = row < a.rows;
  for (int k = c; k < a.L[0].in; k += 16) cur[r][k] = rok ? a.x[row * a.ldx + k] : 0.f;
  int dim = a.L[0].in;
  for (int i = 0; i < a.n; ++i) {
    const Layer L = a.L[i];
    if (L.kind == 0) {  // the row's 16 lanes sit in one wavefront: LDS reads and writes of a row need no barrier
      float s = 0.f;
      for (int k = c; k < L.in; k += 16) s += cur[r][k];
      const float mean = sum16(s) / (float)L.in;
      float q = 0.f;
      for (int k = c; k < L.in; k += 16) {
        const float d = cur[r][k] - mean;
        q = fmaf(d, d, q);
      }
      const float rstd = rsqrtf(sum16(q) / (float)L.in + kLnEps);
      for (int k = c; k < L.in; k += 16) cur[r][k] = (cur[r][k] - mean) * rstd * L.w[k] + L.b[k];
      dim = L.in;
    } else {
      __syncthreads();  // the previous Linear's readers of Wt are done
      for (int e = tid; e < L.out * L.in; e += 256) {
        const int o = e / L.in, k = e - o * L.in;
        Wt[k * kLd + o] = L.w[e];
      }
      __syncthreads();
      float res[8];
      const int jn = (L.out + 15) >> 4;
      if (jn <= 1) linear_rows<1>(cur, Wt, L, r, c, res);
      else if (jn <= 2) linear_rows<2>(cur, Wt, L, r, c, res);
      else if (jn <= 4) linear_rows<4>(cur, Wt, L, r, c, res);
      else linear_rows<8>(cur, Wt, L, r, c, res);
      // every lane of the row has finished reading cur[r][*] (same wavefront, program order): overwrite it
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < jn && c + 16 * j < L.out) cur[r][c + 16 * j] = act_fwd(res[j], L.act);
      dim = L.out;
    }
    if (i + 1 < a.n && rok)  // this layer's output = the next layer's input: what the backward pass reads back
      for (int k = c; k < dim; k += 16) a.tape[row * a.tld + a.L[i + 1].toff + k] = cur[r][k];
  }
  if (rok)
    for (int k = c; k < dim; k += 16) a.y[row * a.ldy + k] = cur[r][k];
}

template <int JN>
__device__ __forceinline__ void dgrad_rows(const float (*g)[kLd], const float* Ws, const Layer& L, int r, int c, float* res) {
  float acc[JN];
#pragma unroll
  for (int j = 0; j < JN; ++j) acc[j] = 0.f;
#pragma unroll 8
  for (int o = 0; o < L.out; ++o) {
    const float gv = g[r][o];
#pragma unroll
    for (int j = 0; j < JN; ++j) acc[j] = fmaf(gv, Ws[o * kLd + c + 16 * j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < JN; ++j) res[j] = acc[j];
}

__global__ __launch_bounds__(256) void mlp_bwd_kernel(Args a) {
  __shared__ float g[kRB][kLd];    // gradient w.r.t. the current layer's (pre-activation) output
  __shared__ float xin[kRB][kLd];  // the current layer's input
  __shared__ float Ws[kMaxD * kLd];  // a Linear's weights Ws[o][k]; scratch of the LayerNorm's affine gradients
  // Parameter-gradient sums of this workgroup over ALL its row blocks: an element is always touched by the same thread, so
  // no barrier guards it, and it reaches the gradient buffer with ONE atomic per workgroup at the end.  (Atomics per 16-row
  // block were fine for 256 rows and a wall at 65 536: 35 M float atomics, 1.25 ms against 1.03 layer by layer.)
  __shared__ float pacc[kMaxP];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  if (a.lds_acc)
    for (int e = tid; e < kMaxP; e += 256) pacc[e] = 0.f;
  const long nblk = (a.rows + kRB - 1) / kRB;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long row = blk * kRB + r;
    const bool rok = row < a.rows;
    {
      const Layer& last = a.L[a.n - 1];
      const int dout = last.kind == 1 ? last.out : last.in;
      for (int k = c; k < dout; k += 16) g[r][k] = rok ? a.dy[row * a.lddy + k] : 0.f;
    }
    for (int i = a.n - 1; i >= 0; --i) {
      const Layer L = a.L[i];
      for (int k = c; k < L.in; k += 16)
        xin[r][k] = rok ? (i == 0 ? a.x[row * a.ldx + k] : a.tape[row * a.tld + L.toff + k]) : 0.f;
      // the activation that produced this input: its derivative (from the input's value) closes the data gradient
      const int pact = (i > 0 && a.L[i - 1].kind == 1) ? a.L[i - 1].act : 0;
      if (L.kind == 1) {
        __syncthreads();  // g and xin of every row are in LDS; Ws is free
        if (i > 0)
          for (int e = tid; e < L.out * L.in; e += 256) {
            const int o = e / L.in, k = e - o * L.in;
            Ws[o * kLd + k] = L.w[e];
          }
        // weight and bias gradients of the block's rows (rows past the end carry zeros)
        for (int e = tid; e < L.out * L.in; e += 256) {
          const int o = e / L.in, k = e - o * L.in;
          float s = 0.f;
#pragma unroll
          for (int rr = 0; rr < kRB; ++rr) s = fmaf(g[rr][o], xin[rr][k], s);
          if (a.lds_acc) pacc[L.pw + e] += s;
          else atomicAdd(L.gw + e, s);
        }
        if (tid < L.out && L.gb) {
          float s = 0.f;
#pragma unroll
          for (int rr = 0; rr < kRB; ++rr) s += g[rr][tid];
          if (a.lds_acc) pacc[L.pb + tid] += s;
          else atomicAdd(L.gb + tid, s);
        }
        __syncthreads();  // Ws staged; every reader of g is done
        if (i > 0) {
          float res[8];
          const int jn = (L.in + 15) >> 4;
          if (jn <= 1) dgrad_rows<1>(g, Ws, L, r, c, res);
          else if (jn <= 2) dgrad_rows<2>(g, Ws, L, r, c, res);
          else if (jn <= 4) dgrad_rows<4>(g, Ws, L, r, c, res);
          else dgrad_rows<8>(g, Ws, L, r, c, res);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < jn && c + 16 * j < L.in) g[r][c + 16 * j] = res[j] * act_der(xin[r][c + 16 * j], pact);
        }
      } else {
        // LayerNorm: statistics recomputed from the input (cheaper than a trip to memory)
        float s = 0.f;
        for (int k = c; k < L.in; k += 16) s += xin[r][k];
        const float mean = sum16(s) / (float)L.in;
        float q = 0.f;
        for (int k = c; k < L.in; k += 16) {
          const float d = xin[r][k] - mean;
          q = fmaf(d, d, q);
        }
        const float rstd = rsqrtf(sum16(q) / (float)L.in + kLnEps);
        __syncthreads();  // Ws (scratch) is free
        float m1 = 0.f, m2 = 0.f;
        for (int k = c; k < L.in; k += 16) {
          const float xh = (xin[r][k] - mean) * rstd, gy = g[r][k], gg = gy * L.w[k];
          Ws[r * kLd + k] = gy * xh;
          Ws[(kRB + r) * kLd + k] = gy;
          m1 += gg;
          m2 = fmaf(gg, xh, m2);
        }
        m1 = sum16(m1) / (float)L.in;
        m2 = sum16(m2) / (float)L.in;
        __syncthreads();
        if (tid < L.in) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int rr = 0; rr < kRB; ++rr) {
            s1 += Ws[rr * kLd + tid];
            s2 += Ws[(kRB + rr) * kLd + tid];
          }
          if (a.lds_acc) {
            pacc[L.pw + tid] += s1;
            pacc[L.pb + tid] += s2;
          } else {
            atomicAdd(L.gw + tid, s1);
            atomicAdd(L.gb + tid, s2);
          }
        }
        if (i > 0)
          for (int k = c; k < L.in; k += 16) {
            const float xh = (xin[r][k] - mean) * rstd;
            g[r][k] = rstd * (g[r][k] * L.w[k] - m1 - xh * m2) * act_der(xin[r][k], pact);
          }
      }
    }
    __syncthreads();  // the next block's loads overwrite g / xin, which layer 0's sums read across rows
  }
  if (a.lds_acc)  // every element by the thread that summed it
    for (int i = 0; i < a.n; ++i) {
      const Layer L = a.L[i];
      if (L.kind == 1) {
        for (int e = tid; e < L.out * L.in; e += 256) atomicAdd(L.gw + e, pacc[L.pw + e]);
        if (tid < L.out && L.gb) atomicAdd(L.gb + tid, pacc[L.pb + tid]);
      } else if (tid < L.in) {
        atomicAdd(L.gw + tid, pacc[L.pw + tid]);
        atomicAdd(L.gb + tid, pacc[L.pb + tid]);
      }
    }
}

int fill(Args& a, const srl_mlp_layer* layers, int n) {
  if (!layers || n < 1 || n > SRL_MLP_MAX_LAYERS) return -1;
  int dim = layers[0].in, toff = 0;
  for (int i = 0; i < n; ++i) {
    const srl_mlp_layer& s = layers[i];
    if (s.kind != 0 && s.kind != 1) return -1;
    if (s.in != dim || s.in < 1 || s.in > kMaxD || !s.w) return -1;
    if (s.kind == 1 && (s.out < 1 || s.out > kMaxD || s.act < 0 || s.act > 2)) return -1;
    if (s.kind == 0 && !s.b) return -1;
    Layer& L = a.L[i];
    L.kind = s.kind; L.in = s.in; L.out = s.kind == 1 ? s.out : s.in; L.act = s.kind == 1 ? s.act : 0;
    L.w = s.w; L.b = s.b; L.gw = s.gw; L.gb = s.gb;
    L.toff = toff;
    if (i > 0) toff += s.in;  // layer 0 reads x
    else toff = 0;
    dim = L.out;
  }
  // tape rows: the inputs of layers 1 .. n-1
  toff = 0;
  for (int i = 1; i < n; ++i) {
    a.L[i].toff = toff;
    toff += a.L[i].in;
  }
  int poff = 0;
  for (int i = 0; i < n; ++i) {
    Layer& L = a.L[i];
    L.pw = poff;
    poff += L.kind == 1 ? L.out * L.in : L.in;
    L.pb = poff;
    poff += L.kind == 1 ? L.out : L.in;
  }
  a.lds_acc = poff <= kMaxP ? 1 : 0;
  a.n = n;
  return toff;
}

}  // namespace

#include "mlp_mfma.h"
#include "mlp_sig.h"
#include "mlp_sigh.h"

// rows from which the matrix-core chain (mlp_mfma.h) takes over from the FMA chain; SRL_MLP_MFMA=0 switches it off (A/B)
static long mfma_min_rows() {
  static const long v = [] { const char* e = getenv("SRL_MLP_MFMA"); return e ? (e[0] == '0' ? (1L << 62) : atol(e)) : 512L; }();
  return v;
}

extern "C" int64_t srl_mlp_bwd_max_rows(const srl_mlp_layer* layers, int n) {
  Args a{};
  if (fill(a, layers, n) < 0) return 0;
  {
    MArgs m{};
    m.a = a;
    if (mfma_min_rows() < (1L << 40) && mm_plan(m)) return 1L << 30;  // the matrix-core chain: any row count
  }
  // Measured against the layer-by-layer kernels on 2 x 64 nets (scripts/mlp_rows_sweep.py): 0.30 against 0.64 ms per update at
  // 256 rows, 0.54 against 0.89 at 16 384, 1.30 against 1.03 at 65 536 -- 256 threads per CU are too few once the row count
  // fills the chip (scripts/mlp_probe.py: a 64 x 64 Linear's backward 105 us, a LayerNorm's 84 us at 65 536 rows).  Without
  // the LDS sums every 16-row block adds its partial sums with float atomics: a few thousand rows only.
  return a.lds_acc ? 32768 : 8192;
}

extern "C" int64_t srl_mlp_tape_floats(const srl_mlp_layer* layers, int n) {
  Args a{};
  const int t = fill(a, layers, n);
  return t < 0 ? -1 : (t > 0 ? t : 1);
}

// the matrix-core chain keeps no tape: its backward pass walks the chain forward again from x
static bool mfma_takes(const Args& a, long rows, MArgs& m) {
  m = MArgs{};
  m.a = a;
  return rows >= mfma_min_rows() && mm_plan(m);
}

extern "C" int64_t srl_mlp_tape_floats_at(const srl_mlp_layer* layers, int n, int64_t rows) {
  Args a{};
  const int t = fill(a, layers, n);
  if (t < 0) return -1;
  MArgs m;
  return mfma_takes(a, rows, m) ? 0 : (t > 0 ? t : 1);
}

extern "C" int srl_mlp_fwd(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                           float* tape, int64_t tape_ld, float* y, int64_t ldy) {
  Args a{};
  const int t = fill(a, layers, n);
  SRL_CHECK_ARG(t >= 0, "unsupported chain (LayerNorm / Linear layers, widths 1..128, at most SRL_MLP_MAX_LAYERS)");
  SRL_CHECK_ARG(x && y && rows >= 0, "null tensor");
  if (rows == 0) return 0;
  a.x = x; a.ldx = ldx; a.rows = rows; a.tape = tape; a.tld = tape_ld; a.y = y; a.ldy = ldy;
  MArgs m;
  if (mfma_takes(a, rows, m)) {
    if (hx_fwd(a, (hipStream_t)stream) || sx_fwd(a, (hipStream_t)stream)) {   // a shape with a kernel of its own (mlp_sigh.h / mlp_sig.h)
      SRL_LAUNCH_CHECK();
      return 0;
    }
    const long tiles4 = srl_ceil_div(rows, 128L);
    const int lds = 4 * m.par_floats;
    static int attr = 0;
    if (lds > attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = lds;
    }
    hipLaunchKernelGGL(mlp_fwd_mfma_kernel, dim3((unsigned)(tiles4 < 512 ? tiles4 : 512)), dim3(256), lds, (hipStream_t)stream, m);
    SRL_LAUNCH_CHECK();
    return 0;
  }
  SRL_CHECK_ARG((n == 1 || tape) && tape_ld >= t, "null tape / short tape rows");
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((unsigned)srl_ceil_div(rows, (long)kRB)), dim3(256), 0, (hipStream_t)stream, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

template <int NL>
static void launch_bwd_mfma(const MArgs& m, long rows, hipStream_t st) {
  const long groups = srl_ceil_div(rows, 32L * kBwdWaves);
  const int lds = (int)mm_bwd_lds_bytes(m);
  static int attr = 0;
  if (lds > attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_bwd_mfma_kernel<NL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = lds;
  }
  hipLaunchKernelGGL(mlp_bwd_mfma_kernel<NL>, dim3((unsigned)(groups < 256 ? groups : 256)), dim3(64 * kBwdWaves), lds, st, m);
}

static int mlp_bwd_impl(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                        const float* tape, int64_t tape_ld, const float* dy, int64_t lddy, float* dx, int64_t lddx);

extern "C" int srl_mlp_bwd(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                           const float* tape, int64_t tape_ld, const float* dy, int64_t lddy) {
  return mlp_bwd_impl(stream, layers, n, x, ldx, rows, tape, tape_ld, dy, lddy, nullptr, 0);
}

extern "C" int srl_mlp_bwd_dx(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                              const float* dy, int64_t lddy, float* dx, int64_t lddx) {
  SRL_CHECK_ARG(dx && layers && n >= 1 && lddx >= layers[0].in, "null dx / short dx rows");
  return mlp_bwd_impl(stream, layers, n, x, ldx, rows, nullptr, 0, dy, lddy, dx, lddx);
}

static int mlp_bwd_impl(void* stream, const srl_mlp_layer* layers, int n, const float* x, int64_t ldx, int64_t rows,
                        const float* tape, int64_t tape_ld, const float* dy, int64_t lddy, float* dx, int64_t lddx) {
  Args a{};
  const int t = fill(a, layers, n);
  SRL_CHECK_ARG(t >= 0, "unsupported chain");
  SRL_CHECK_ARG(x && dy && rows >= 0, "null tensor");
  for (int i = 0; i < n; ++i) SRL_CHECK_ARG(a.L[i].gw && (a.L[i].gb || (a.L[i].kind == 1 && !a.L[i].b)), "null gradient");
  if (rows == 0) return 0;
  a.x = x; a.ldx = ldx; a.rows = rows; a.tape = const_cast<float*>(tape); a.tld = tape_ld; a.dy = dy; a.lddy = lddy;
  a.dx = dx; a.lddx = lddx;
  MArgs m;
  if (mfma_takes(a, rows, m)) {
    static const int dbg = [] { const char* e = getenv("SRL_MLP_DBG"); return e ? atoi(e) : 0; }();
    m.dbg = dbg;
    if (hx_bwd(a, dbg, (hipStream_t)stream) || sx_bwd(a, dbg, (hipStream_t)stream)) {
      SRL_LAUNCH_CHECK();
      return 0;
    }
    if (n <= 4) launch_bwd_mfma<4>(m, rows, (hipStream_t)stream);
    else if (n <= 6) launch_bwd_mfma<6>(m, rows, (hipStream_t)stream);
    else launch_bwd_mfma<8>(m, rows, (hipStream_t)stream);
    SRL_LAUNCH_CHECK();
    return 0;
  }
  SRL_CHECK_ARG(!dx, "srl_mlp_bwd_dx: only chains and row counts that keep no tape (srl_mlp_tape_floats_at == 0) form dx");
  SRL_CHECK_ARG((n == 1 || tape) && tape_ld >= t, "null tape / short tape rows");
  long groups = srl_ceil_div(rows, (long)kRB);
  if (a.lds_acc && groups > kBwdGroups) groups = kBwdGroups;
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3((unsigned)groups), dim3(256), 0, (hipStream_t)stream, a);
  SRL_LAUNCH_CHECK();
  return 0;
}

#ifdef SRL_MLP_PROF
// variant builds only: the phase stamps of the last mlp_bwd_mfma_kernel launch (scripts/mlp_prof.sh)
extern "C" int mlp_prof_dump(long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mlp_prof), sizeof(long long) * n);
}
#endif
