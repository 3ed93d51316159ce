// GAE / V-trace segmented reverse scan + advantage statistics + masked normalisation (gfx950).
//
// Data layout: every leaf is time-major [T(+1), B, Nc]; one "column" = one (env, value-channel)
// pair whose successive time steps are B*Nc floats apart.  A workgroup owns COLS adjacent columns
// so that each time row of its tile is one contiguous COLS*4-byte segment (coalesced along B),
// stages the whole [T, COLS] tile through LDS, and splits the T axis of every column over
// SEG = 256/COLS lanes.  The recurrence g_t = delta_t + m_t * g_{t+1} is an affine map in g_{t+1};
// affine maps compose associatively ((m,d) o (m',d') = (m m', d + m d')), and m_t = 0 at episode
// boundaries makes the scan "segmented" for free.  Three passes over the LDS tile:
//   1. every lane folds its T-chunk into one affine map (M, D)           [float64]
//   2. every lane folds the maps of the chunks behind it into its carry-in g
//   3. every lane replays its chunk from the carry-in, emitting adv, ret and the masked sums
// Time is processed in tiles of TT rows from the end, the carry crossing tiles through LDS, so T is
// unbounded.  All arithmetic that the reference does in float64 (gae.py:46-48) is float64 here.
#include "srl_common.h"

namespace {

struct GaeParams {
  const float* reward;
  const float* value;
  const uint8_t* done;
  const uint8_t* truncated;
  const uint8_t* on_reset;
  const float* ratio;
  float* adv;
  float* ret;
  double* stats;
  int T, B, Nc, TT;
  double gamma, lambda, rho, c;
};

// LDS carve (dynamic, 16-byte aligned pieces):
//   double segM[SEG*COLS], segD[SEG*COLS], carry[COLS], red[4*3]
//   float  r[TT*COLS], v[(TT+1)*COLS], (ratio[TT*COLS])
//   uint8  tr[(TT+1)*COLS], orr[(TT+1)*COLS]
template <int COLS, bool VTRACE>
__global__ __launch_bounds__(256) void gae_scan_kernel(GaeParams p) {
  constexpr int SEG = 256 / COLS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* segM = reinterpret_cast<double*>(smem);
  double* segD = segM + SEG * COLS;
  double* carry = segD + SEG * COLS;
  double* red = carry + COLS;
  float* s_r = reinterpret_cast<float*>(red + 12);
  float* s_v = s_r + p.TT * COLS;
  float* s_q = s_v + (p.TT + 1) * COLS;  // ratio (only when VTRACE)
  uint8_t* s_tr = reinterpret_cast<uint8_t*>(s_q + (VTRACE ? p.TT * COLS : 0));
  uint8_t* s_or = s_tr + (p.TT + 1) * COLS;

  const int tid = threadIdx.x;
  const int lc = tid % COLS;   // local column
  const int seg = tid / COLS;  // which T-chunk of the column this lane owns
  const long ncols = (long)p.B * p.Nc;
  const long col0 = (long)blockIdx.x * COLS;
  const long col = col0 + lc;
  const bool col_ok = col < ncols;
  const long fcol = col_ok ? col / p.Nc : 0;  // flag column (flags have a trailing dim of 1)
  const double gl = p.gamma * p.lambda;

  if (tid < COLS) carry[tid] = 0.0;  // adv[T] = 0 (gae.py:80)
  double acc[3] = {0.0, 0.0, 0.0};   // n, sum, sumsq

  for (int t1 = p.T; t1 > 0; t1 -= p.TT) {
    const int t0 = t1 - p.TT > 0 ? t1 - p.TT : 0;
    const int len = t1 - t0;
    __syncthreads();  // previous tile fully consumed (and carry[] initialised / updated)
    // ---- stage rows t0..t1 (value/flags: one extra row) ------------------------------------------
    for (int r = seg; r <= len; r += SEG) {
      const long t = t0 + r;
      float v = 0.f;
      uint8_t tr = 0, orr = 0;
      if (col_ok) {
        const uint8_t dn = p.done[t * p.B + fcol];
        v = p.value[t * ncols + col];
        v = dn ? 0.0f : v;  // value * (1 - done), mappo.py:124
        tr = p.truncated[t * p.B + fcol];
        orr = p.on_reset[t * p.B + fcol];
      }
      s_v[r * COLS + lc] = v;
      s_tr[r * COLS + lc] = tr;
      s_or[r * COLS + lc] = orr;
      if (r < len) {
        s_r[r * COLS + lc] = col_ok ? p.reward[t * ncols + col] : 0.f;
        if (VTRACE) s_q[r * COLS + lc] = col_ok ? p.ratio[t * p.B + fcol] : 1.f;
      }
    }
    __syncthreads();

    const int L = (len + SEG - 1) / SEG;
    const int a = seg * L < len ? seg * L : len;
    const int b = (seg + 1) * L < len ? (seg + 1) * L : len;

    auto step = [&](int r, double& delta, double& m) {
      const double v0 = (double)s_v[r * COLS + lc];
      const double v1 = (double)s_v[(r + 1) * COLS + lc];
      const double nr = 1.0 - (double)s_or[(r + 1) * COLS + lc];
      const double nt = 1.0 - (double)s_tr[(r + 1) * COLS + lc];
      delta = (double)s_r[r * COLS + lc] + p.gamma * v1 * nr - v0;  // gae.py:63
      m = gl * nr * nt;                                             // gae.py:87
      if (VTRACE) {
        const double q = (double)s_q[r * COLS + lc];
        delta *= fmin(q, p.rho);  // gae.py:65
        m *= fmin(q, p.c);        // gae.py:89
      }
    };

    // ---- pass 1: fold my chunk into (M, D): g_a = D + M * g_b ------------------------------------
    double M = 1.0, D = 0.0;
    for (int r = b - 1; r >= a; --r) {
      double delta, m;
      step(r, delta, m);
      D = delta + m * D;
      M = m * M;
    }
    segM[seg * COLS + lc] = M;
    segD[seg * COLS + lc] = D;
    __syncthreads();
    // ---- pass 2: carry-in of my chunk = fold of all later chunks applied to the tile carry --------
    double g = carry[lc];
    for (int s = SEG - 1; s > seg; --s) g = segD[s * COLS + lc] + segM[s * COLS + lc] * g;
    const double g_tile_out = D + M * g;  // only meaningful for seg == 0
    // ---- pass 3: replay ---------------------------------------------------------------------------
    for (int r = b - 1; r >= a; --r) {
      double delta, m;
      step(r, delta, m);
      g = delta + m * g;
      const float advf = (float)g;  // gae.py:97
      if (col_ok) {
        const long t = t0 + r;
        p.adv[t * ncols + col] = advf;
        p.ret[t * ncols + col] = advf + s_v[r * COLS + lc];  // mappo.py:143 (float32 add)
        const double mask = 1.0 - (double)s_or[(r + 1) * COLS + lc];  // mappo.py:260
        const double x = (double)advf * mask;                           // utils.py:52
        acc[0] += mask;
        acc[1] += x;
        acc[2] += x * x;
      }
    }
    __syncthreads();  // everyone has read carry[] / seg*[]
    if (seg == 0) carry[lc] = g_tile_out;
  }

  if (p.stats != nullptr) {
    __syncthreads();
    block_sum<3, 256>(acc, red);
    if (tid == 0) {
      atomicAdd(&p.stats[0], acc[0]);
      atomicAdd(&p.stats[1], acc[1]);
      atomicAdd(&p.stats[2], acc[2]);
    }
  }
}

template <int COLS>
size_t gae_lds_bytes(int TT, bool vtrace) {
  constexpr int SEG = 256 / COLS;
  size_t b = sizeof(double) * (2 * SEG * COLS + COLS + 12);
  b += sizeof(float) * ((size_t)TT * COLS + (size_t)(TT + 1) * COLS + (vtrace ? (size_t)TT * COLS : 0));
  b += 2 * (size_t)(TT + 1) * COLS;
  return (b + 15) & ~size_t(15);
}

template <int COLS>
int launch_gae(hipStream_t st, GaeParams p, bool vtrace) {
  // largest time tile that keeps the LDS tile under ~60 KiB (>= 2 workgroups per CU)
  const size_t budget = 60 * 1024;
  int TT = p.T;
  while (TT > 8 && gae_lds_bytes<COLS>(TT, vtrace) > budget) TT = (TT + 1) / 2;
  p.TT = TT;
  const size_t lds = gae_lds_bytes<COLS>(TT, vtrace);
  const long ncols = (long)p.B * p.Nc;
  dim3 grid((unsigned)srl_ceil_div(ncols, COLS));
  if (vtrace)
    hipLaunchKernelGGL((gae_scan_kernel<COLS, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((gae_scan_kernel<COLS, false>), grid, dim3(256), lds, st, p);
  SRL_LAUNCH_CHECK();
  return 0;
}

// ---- masked statistics / normalisation -------------------------------------------------------------
__global__ __launch_bounds__(256) void masked_stats_kernel(const float* x, const uint8_t* mask, int invert, long n,
                                                           double* stats) {
  __shared__ double red[12];
  double acc[3] = {0.0, 0.0, 0.0};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double m = 1.0;
    if (mask) m = invert ? 1.0 - (double)mask[i] : (double)mask[i];
    const double v = (double)x[i] * m;
    acc[0] += m;
    acc[1] += v;
    acc[2] += v * v;
  }
  block_sum<3, 256>(acc, red);
  if (threadIdx.x == 0) {
    atomicAdd(&stats[0], acc[0]);
    atomicAdd(&stats[1], acc[1]);
    atomicAdd(&stats[2], acc[2]);
  }
}

__global__ __launch_bounds__(256) void masked_normalize_kernel(const float* x, const uint8_t* mask, int invert, long n,
                                                               const double* stats, double eps, int unbiased,
                                                               float* out) {
  const double cnt = stats[0];
  const double mean = stats[1] / cnt;
  double var = stats[2] / cnt - mean * mean;  // utils.py:62-64
  if (unbiased) var *= cnt / (cnt - 1.0);
  const double denom = sqrt(var) + eps;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double m = 1.0;
    if (mask) m = invert ? 1.0 - (double)mask[i] : (double)mask[i];
    out[i] = (float)(((double)x[i] * m - mean) / denom);  // utils.py:67
  }
}

}  // namespace

extern "C" int srl_gae_scan(void* stream, const float* reward, const float* value, const uint8_t* done,
                            const uint8_t* truncated, const uint8_t* on_reset, const float* imp_ratio, int T, int B,
                            int Nc, double gamma, double lambda, double rho, double c, float* adv, float* ret,
                            double* stats) {
  SRL_CHECK_ARG(T >= 0 && B >= 0 && Nc >= 1, "T, B >= 0 and Nc >= 1 required");
  hipStream_t st = (hipStream_t)stream;
  if (stats) SRL_HIP_TRY(hipMemsetAsync(stats, 0, 3 * sizeof(double), st));
  if (T == 0 || B == 0) return 0;  // empty batch: nothing to scan (tensors may be null)
  SRL_CHECK_ARG(reward && value && done && truncated && on_reset && adv && ret, "null tensor");
  GaeParams p{reward, value, done, truncated, on_reset, imp_ratio, adv, ret, stats, T, B, Nc, 0,
              gamma,  lambda, rho,  c};
  const long ncols = (long)B * Nc;
  // narrow tiles while the grid would not cover the chip; wide (256-byte rows) once it does
  if (ncols >= 64L * 512) return launch_gae<64>(st, p, imp_ratio != nullptr);
  return launch_gae<16>(st, p, imp_ratio != nullptr);
}

extern "C" int srl_masked_stats(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n,
                                double* stats) {
  SRL_CHECK_ARG(x && stats && n >= 0, "null tensor");
  hipStream_t st = (hipStream_t)stream;
  SRL_HIP_TRY(hipMemsetAsync(stats, 0, 3 * sizeof(double), st));
  if (n == 0) return 0;
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(masked_stats_kernel, dim3(grid), dim3(256), 0, st, x, mask, mask_invert, n, stats);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_masked_normalize(void* stream, const float* x, const uint8_t* mask, int mask_invert, long n,
                                    const double* stats, double eps, int unbiased, float* out) {
  SRL_CHECK_ARG(x && stats && out && n >= 0, "null tensor");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(srl_ceil_div(n, 256) < 2048 ? srl_ceil_div(n, 256) : 2048);
  hipLaunchKernelGGL(masked_normalize_kernel, dim3(grid), dim3(256), 0, st, x, mask, mask_invert, n, stats, eps,
                     unbiased, out);
  SRL_LAUNCH_CHECK();
  return 0;
}
