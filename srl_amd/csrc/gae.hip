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
// Two kernels share that scheme: gae_scan_reg_kernel (one value channel, B % 4 == 0 -- the trainer's
// layout; register-resident quads of columns, 16-byte loads, log-step scan over chunk maps) and
// gae_scan_kernel (any Nc / alignment; LDS-staged tile).
#include "srl_common.h"

namespace {

struct GaeParams {
  const float* reward;
  const float* value;
  const uint8_t* done;
  const uint8_t* truncated;
  const uint8_t* on_reset;
  const float* ratio;
  const float* gamma_t;   // per-step discount [T, B, 1] or null (gae.py:51-55)
  const float* lambda_t;  // per-step lambda [T, B, 1] or null (gae.py:56-60)
  float* adv;
  float* ret;
  double* stats;
  int T, B, Nc, TT;
  double gamma, lambda, rho, c;
  void* ws;  // null: stats zeroed by a memset and accumulated with atomics; else see stats_finish()
};

// ---- the three advantage sums without a zeroing launch and without float64 atomics ---------------------------------
// workspace = { uint32 ticket, pad, double partial[grid][3] }.  Every workgroup stores its partial sums, releases
// them (agent scope) and takes a ticket; the workgroup that draws the last ticket acquires, adds the partials up in
// workgroup order -- so the result does not depend on which workgroup finished when: bitwise reproducible -- writes
// stats[0..2] and puts the ticket back to zero for the next launch.  The caller zeroes the workspace once.
struct StatsWs {
  unsigned int ticket;
  unsigned int pad;
  double partial[1];
};

__device__ __forceinline__ void stats_finish(const GaeParams& p, double (&acc)[3], double* red) {
  if (p.stats == nullptr) return;
  __syncthreads();
  block_sum<3, 256>(acc, red);
  if (p.ws == nullptr) {
    if (threadIdx.x == 0) {
      atomicAdd(&p.stats[0], acc[0]);
      atomicAdd(&p.stats[1], acc[1]);
      atomicAdd(&p.stats[2], acc[2]);
    }
    return;
  }
  StatsWs* ws = static_cast<StatsWs*>(p.ws);
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    double* mine = ws->partial + 3 * (size_t)blockIdx.x;
    mine[0] = acc[0], mine[1] = acc[1], mine[2] = acc[2];
    __threadfence();  // release: the partials are visible device-wide before the ticket is
    const unsigned int t = atomicAdd(&ws->ticket, 1u);
    is_last = t == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();  // acquire
  double tot[3] = {0.0, 0.0, 0.0};
  for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      tot[i] += __hip_atomic_load(ws->partial + 3 * (size_t)b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();  // `red` was read by thread 0 above
  block_sum<3, 256>(tot, red);
  if (threadIdx.x == 0) {
    p.stats[0] = tot[0], p.stats[1] = tot[1], p.stats[2] = tot[2];
    __hip_atomic_store(&ws->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// LDS carve (dynamic, 16-byte aligned pieces):
//   double segM[SEG*COLS], segD[SEG*COLS], carry[COLS], red[4*3]
//   float  r[TT*COLS], v[(TT+1)*COLS], (ratio[TT*COLS])
//   uint8  tr[(TT+1)*COLS], orr[(TT+1)*COLS]
template <int COLS, bool VTRACE, bool GLT>
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
  float* s_g = s_q + (VTRACE ? p.TT * COLS : 0);  // per-step gamma, lambda (only when GLT)
  float* s_l = s_g + (GLT ? p.TT * COLS : 0);
  uint8_t* s_tr = reinterpret_cast<uint8_t*>(s_l + (GLT ? p.TT * COLS : 0));
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
        if (GLT) {
          s_g[r * COLS + lc] = col_ok && p.gamma_t ? p.gamma_t[t * p.B + fcol] : (float)p.gamma;
          s_l[r * COLS + lc] = col_ok && p.lambda_t ? p.lambda_t[t * p.B + fcol] : (float)p.lambda;
        }
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
      // per-step tensors are float32 widened to float64 (gae.py:54,59); a scalar stays the caller's double
      const double gam = GLT && p.gamma_t ? (double)s_g[r * COLS + lc] : p.gamma;
      const double lam = GLT && p.lambda_t ? (double)s_l[r * COLS + lc] : p.lambda;
      delta = (double)s_r[r * COLS + lc] + gam * v1 * nr - v0;  // gae.py:63
      m = (GLT ? gam * lam : gl) * nr * nt;                      // gae.py:87
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
        // the mask is [T, B, 1]: its sum counts every (step, env) once, whatever the number of value channels (utils.py:41-47)
        if (col % p.Nc == 0) acc[0] += mask;
        acc[1] += x;
        acc[2] += x * x;
      }
    }
    __syncthreads();  // everyone has read carry[] / seg*[]
    if (seg == 0) carry[lc] = g_tile_out;
  }

  stats_finish(p, acc, red);
}

template <int COLS>
size_t gae_lds_bytes(int TT, bool vtrace, bool glt) {
  constexpr int SEG = 256 / COLS;
  size_t b = sizeof(double) * (2 * SEG * COLS + COLS + 12);
  b += sizeof(float) * ((size_t)TT * COLS + (size_t)(TT + 1) * COLS + (vtrace ? (size_t)TT * COLS : 0) +
                        (glt ? 2 * (size_t)TT * COLS : 0));
  b += 2 * (size_t)(TT + 1) * COLS;
  return (b + 15) & ~size_t(15);
}

template <int COLS>
int launch_gae(hipStream_t st, GaeParams p, bool vtrace) {
  const bool glt = p.gamma_t != nullptr || p.lambda_t != nullptr;
  // largest time tile that keeps the LDS tile under ~60 KiB (>= 2 workgroups per CU)
  const size_t budget = 60 * 1024;
  int TT = p.T;
  while (TT > 8 && gae_lds_bytes<COLS>(TT, vtrace, glt) > budget) TT = (TT + 1) / 2;
  p.TT = TT;
  const size_t lds = gae_lds_bytes<COLS>(TT, vtrace, glt);
  const long ncols = (long)p.B * p.Nc;
  dim3 grid((unsigned)srl_ceil_div(ncols, COLS));
  if (vtrace && glt)
    hipLaunchKernelGGL((gae_scan_kernel<COLS, true, true>), grid, dim3(256), lds, st, p);
  else if (vtrace)
    hipLaunchKernelGGL((gae_scan_kernel<COLS, true, false>), grid, dim3(256), lds, st, p);
  else if (glt)
    hipLaunchKernelGGL((gae_scan_kernel<COLS, false, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((gae_scan_kernel<COLS, false, false>), grid, dim3(256), lds, st, p);
  SRL_LAUNCH_CHECK();
  return 0;
}

// ---- register-resident variant (Nc == 1, B % 4 == 0, 16-byte aligned leaves) ------------------------
// The common layout: one value channel, so flags and floats share the column index.  Each lane owns
// a quad of adjacent columns x L consecutive time rows and reads them straight into registers with
// 16-byte (floats) / 4-byte (flags) loads: a time row of the workgroup's tile is one contiguous
// CQ*16-byte segment.  Nothing but the per-chunk affine maps goes through LDS, and the carry-in of
// every chunk comes from a log-step suffix scan over those maps (SEG = 256/CQ chunks per column).
template <int CQ, int L, bool VTRACE>
__global__ __launch_bounds__(256) void gae_scan_reg_kernel(GaeParams p) {
  constexpr int COLS = CQ * 4;
  constexpr int SEG = 256 / CQ;
  constexpr int TT = SEG * L;
  __shared__ __attribute__((aligned(16))) double sM[SEG * COLS];
  __shared__ __attribute__((aligned(16))) double sD[SEG * COLS];
  __shared__ double red[12];

  const int tid = threadIdx.x;
  const int cq = tid % CQ;
  const int seg = tid / CQ;
  const long B = p.B;
  const long col = (long)blockIdx.x * COLS + 4 * cq;
  const bool col_ok = col < B;
  const double gl = p.gamma * p.lambda;
  const int lbase = seg * COLS + 4 * cq;

  double carry[4] = {0.0, 0.0, 0.0, 0.0};  // adv[T] = 0 (gae.py:80)
  double acc[3] = {0.0, 0.0, 0.0};

  for (int t1 = p.T; t1 > 0; t1 -= TT) {
    const int t0 = t1 - TT > 0 ? t1 - TT : 0;
    const int len = t1 - t0;
    const int a = seg * L;

    float v[L + 1][4], rw[L][4], q[L][4];
    uint32_t orw[L + 1], trw[L + 1];
#pragma unroll
    for (int i = 0; i <= L; ++i) {
      float4 vv = make_float4(0.f, 0.f, 0.f, 0.f);
      uint32_t dn = 0;
      orw[i] = 0;
      trw[i] = 0;
      if (col_ok && a + i <= len) {
        const long o = (long)(t0 + a + i) * B + col;
        vv = *reinterpret_cast<const float4*>(p.value + o);
        dn = *reinterpret_cast<const uint32_t*>(p.done + o);
        trw[i] = *reinterpret_cast<const uint32_t*>(p.truncated + o);
        orw[i] = *reinterpret_cast<const uint32_t*>(p.on_reset + o);
      }
      v[i][0] = (dn & 0xffu) ? 0.f : vv.x;  // value * (1 - done), mappo.py:124
      v[i][1] = (dn & 0xff00u) ? 0.f : vv.y;
      v[i][2] = (dn & 0xff0000u) ? 0.f : vv.z;
      v[i][3] = (dn & 0xff000000u) ? 0.f : vv.w;
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
      float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 qq = make_float4(1.f, 1.f, 1.f, 1.f);
      if (col_ok && a + i < len) {
        const long o = (long)(t0 + a + i) * B + col;
        rr = *reinterpret_cast<const float4*>(p.reward + o);
        if (VTRACE) qq = *reinterpret_cast<const float4*>(p.ratio + o);
      }
      rw[i][0] = rr.x, rw[i][1] = rr.y, rw[i][2] = rr.z, rw[i][3] = rr.w;
      q[i][0] = qq.x, q[i][1] = qq.y, q[i][2] = qq.z, q[i][3] = qq.w;
    }

    auto step = [&](int i, int c, double& delta, double& m) {
      const double nr = 1.0 - (double)((orw[i + 1] >> (8 * c)) & 0xffu);
      const double nt = 1.0 - (double)((trw[i + 1] >> (8 * c)) & 0xffu);
      delta = (double)rw[i][c] + p.gamma * (double)v[i + 1][c] * nr - (double)v[i][c];  // gae.py:63
      m = gl * nr * nt;                                                                 // gae.py:87
      if (VTRACE) {
        delta *= fmin((double)q[i][c], p.rho);  // gae.py:65
        m *= fmin((double)q[i][c], p.c);        // gae.py:89
      }
      if (a + i >= len) delta = 0.0, m = 1.0;  // rows past the (partial) tile: identity map
    };

    // ---- pass 1: fold my chunk into (M, D) per column --------------------------------------------
    double M[4], D[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      M[c] = 1.0, D[c] = 0.0;
#pragma unroll
      for (int i = L - 1; i >= 0; --i) {
        double delta, m;
        step(i, c, delta, m);
        D[c] = delta + m * D[c];
        M[c] = m * M[c];
      }
    }
    // ---- pass 2: inclusive suffix scan of the chunk maps over seg (log steps, in place) ----------
    __syncthreads();  // previous tile's readers are done
#pragma unroll
    for (int c = 0; c < 4; ++c) sM[lbase + c] = M[c], sD[lbase + c] = D[c];
    __syncthreads();
#pragma unroll
    for (int off = 1; off < SEG; off <<= 1) {
      double M2[4], D2[4];
      const bool has = seg + off < SEG;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        M2[c] = has ? sM[lbase + off * COLS + c] : 1.0;
        D2[c] = has ? sD[lbase + off * COLS + c] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        D[c] = D[c] + M[c] * D2[c];
        M[c] = M[c] * M2[c];
        sM[lbase + c] = M[c], sD[lbase + c] = D[c];
      }
      __syncthreads();
    }
    // carry-in of my chunk: the composite of every later chunk applied to the tile carry
    double g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool has = seg + 1 < SEG;
      const double Mn = has ? sM[lbase + COLS + c] : 1.0;
      const double Dn = has ? sD[lbase + COLS + c] : 0.0;
      g[c] = Dn + Mn * carry[c];
      carry[c] = sD[4 * cq + c] + sM[4 * cq + c] * carry[c];  // whole tile applied: next tile's carry
    }
    // ---- pass 3: replay -------------------------------------------------------------------------
    float ao[L][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = L - 1; i >= 0; --i) {
        double delta, m;
        step(i, c, delta, m);
        g[c] = delta + m * g[c];
        const float advf = (float)g[c];  // gae.py:97
        ao[i][c] = advf;
        if (col_ok && a + i < len) {
          const double mask = 1.0 - (double)((orw[i + 1] >> (8 * c)) & 0xffu);  // mappo.py:260
          const double x = (double)advf * mask;                                  // utils.py:52
          acc[0] += mask;
          acc[1] += x;
          acc[2] += x * x;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < L; ++i) {
      if (col_ok && a + i < len) {
        const long o = (long)(t0 + a + i) * B + col;
        *reinterpret_cast<float4*>(p.adv + o) = make_float4(ao[i][0], ao[i][1], ao[i][2], ao[i][3]);
        // mappo.py:143 (float32 add)
        *reinterpret_cast<float4*>(p.ret + o) =
            make_float4(ao[i][0] + v[i][0], ao[i][1] + v[i][1], ao[i][2] + v[i][2], ao[i][3] + v[i][3]);
      }
    }
  }

  stats_finish(p, acc, red);
}

template <int CQ, int L>
int launch_gae_reg(hipStream_t st, const GaeParams& p, bool vtrace) {
  dim3 grid((unsigned)srl_ceil_div((long)p.B, (long)CQ * 4));
  if (vtrace)
    hipLaunchKernelGGL((gae_scan_reg_kernel<CQ, L, true>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((gae_scan_reg_kernel<CQ, L, false>), grid, dim3(256), 0, st, p);
  SRL_LAUNCH_CHECK();
  return 0;
}

template <int CQ>
int launch_gae_reg_l(hipStream_t st, const GaeParams& p, bool vtrace, int lmax = 4) {
  constexpr int SEG = 256 / CQ;
  const int need = (p.T + SEG - 1) / SEG;  // rows per lane if one time tile covered all of T
  if (need <= 1 || lmax == 1) return launch_gae_reg<CQ, 1>(st, p, vtrace);
  if (need <= 2 || lmax == 2) return launch_gae_reg<CQ, 2>(st, p, vtrace);
  return launch_gae_reg<CQ, 4>(st, p, vtrace);  // longer T: several time tiles, carry in registers
}

inline bool aligned_to(const void* ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) & (a - 1)) == 0; }

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

// widest grid any of the tilings above launches: one workgroup per 16 columns
static long gae_max_grid(long B, long Nc) { return srl_ceil_div(B * Nc, 16); }

extern "C" long srl_gae_scan_workspace_bytes(int B, int Nc) {
  if (B < 0 || Nc < 1) return -EINVAL;
  return (long)(16 + 24 * gae_max_grid(B, Nc));
}

extern "C" int srl_gae_scan(void* stream, const float* reward, const float* value, const uint8_t* done,
                            const uint8_t* truncated, const uint8_t* on_reset, const float* imp_ratio,
                            const float* gamma_t, const float* lambda_t, int T, int B, int Nc, double gamma,
                            double lambda, double rho, double c, float* adv, float* ret, double* stats,
                            void* workspace) {
  SRL_CHECK_ARG(T >= 0 && B >= 0 && Nc >= 1, "T, B >= 0 and Nc >= 1 required");
  hipStream_t st = (hipStream_t)stream;
  SRL_CHECK_ARG(!workspace || aligned_to(workspace, 8), "workspace must be 8-byte aligned");
  // The ticketed reduction pays a release fence per workgroup (its own stores must have landed before the ticket): worth it
  // where the zeroing launch is a large part of the call (<= 768 workgroups), not on grids that stream for hundreds of
  // microseconds (measured at 2^20 environments: 547 us against 505 with the memset + atomics)
  if ((long)B * Nc >= 98304) workspace = nullptr;
  if (stats && (!workspace || T == 0 || B == 0)) SRL_HIP_TRY(hipMemsetAsync(stats, 0, 3 * sizeof(double), st));
  if (T == 0 || B == 0) return 0;  // empty batch: nothing to scan (tensors may be null)
  SRL_CHECK_ARG(reward && value && done && truncated && on_reset && adv && ret, "null tensor");
  GaeParams p{reward, value, done, truncated, on_reset, imp_ratio, gamma_t, lambda_t, adv, ret, stats, T, B, Nc, 0,
              gamma,  lambda, rho,  c, stats ? workspace : nullptr};
  const long ncols = (long)B * Nc;
  const bool quads = Nc == 1 && B % 4 == 0 && !gamma_t && !lambda_t && aligned_to(reward, 16) && aligned_to(value, 16) &&
                     aligned_to(adv, 16) && aligned_to(ret, 16) && aligned_to(done, 4) &&
                     aligned_to(truncated, 4) && aligned_to(on_reset, 4) && (!imp_ratio || aligned_to(imp_ratio, 16));
  if (quads) {
    const bool vt = imp_ratio != nullptr;
    static const char* tune = getenv("SRL_GAE_TILE");  // "<cq><l>" tuning knob for scripts/gae_sweep.py
    if (tune && tune[0] && tune[1]) {
      const int lmax = tune[1] - '0';
      if (tune[0] == '4') return launch_gae_reg_l<4>(st, p, vt, lmax);
      if (tune[0] == '8') return launch_gae_reg_l<8>(st, p, vt, lmax);
      if (tune[0] == '3') return launch_gae_reg_l<32>(st, p, vt, lmax);
      if (tune[0] == '6') return launch_gae_reg_l<64>(st, p, vt, lmax);
      if (tune[0] == 'a') return launch_gae_reg_l<128>(st, p, vt, lmax);
      if (tune[0] == 'b') return launch_gae_reg_l<256>(st, p, vt, lmax);
      return launch_gae_reg_l<16>(st, p, vt, lmax);
    }
    // Row width per workgroup grows with the batch (measured on MI355X, scripts/gae_sweep.py): wide
    // rows stream best (1 KiB = 8 whole cache lines per wave load) but need B/256 >> 256 workgroups.
    if (ncols < 2048) return launch_gae_reg_l<4>(st, p, vt);
    if (ncols < 16384) return launch_gae_reg_l<8>(st, p, vt);
    if (ncols < 98304) return launch_gae_reg_l<32>(st, p, vt);
    if (ncols < 393216) return launch_gae_reg_l<64>(st, p, vt);
    if (ncols < 786432) return launch_gae_reg_l<128>(st, p, vt, 2);
    return launch_gae_reg_l<256>(st, p, vt, 2);  // one lane per column quad, pure streaming over T
  }
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
