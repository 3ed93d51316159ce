// The recurrent time loop of a chunk inside ONE launch (round 4): LSTM / GRU layers of the auto-reset backbone
// (reference legacy/algorithm/modules/autoreset_rnn.py:42-66 around torch.nn.LSTM / nn.GRU, recurrent_backbone.py:61-66).
//
// gru.hip leaves the loop on the host: per time step one srl_gemm (W_hh h) and one cell kernel, forward and backward -- 4 C
// launches of a few microseconds of work per layer, which is what a SMAC-sized update (97 launches, 6.4 ms in round 3) mostly
// consists of.  The steps are serial but the ROWS are independent: here a wavefront takes 32 rows through all C steps of the
// chunk on the float32 matrix cores (v_mfma_f32_32x32x2_f32: exact float32 multiply-adds), the way csrc/mlp_mfma.h runs an MLP
// chain: D[gate unit][row] = W_hh . h^T with W_hh as the A operand (fragments resident in LDS, staged once per workgroup) and
// the state h as the B operand -- and since the accumulator layout (lane = row, registers = units) IS the operand layout of the
// next step's product, the state never leaves registers; the four (three) gates of a unit land in the same lane and register
// index of their gate blocks, so the cell is lane-local.  The data gradient runs the same way with W_hh^T fragments.
// Buffers, layouts and saved values are exactly those of the per-step path (gru.hip): the W_ih products for all steps and the
// parameter-gradient products over all steps stay on srl_gemm before / after these kernels, and either path can run the other's
// backward.  (A scalar-FMA version of this loop -- 16 rows per workgroup, W_hh in LDS -- was LDS-bound at 815 us per launch on
// 30 720 rows x 10 steps where the per-step path takes ~200: the matrix cores are what makes the fusion pay.)
#include "srl_common.h"

#include "../../include/srl_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The gates on the hardware's exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm's expf / tanhf and an IEEE
// division: a step's 5 transcendentals per unit were ~4 800 vector instructions per 32 rows, as long as the step's 256 MFMAs, and
// with one wavefront per SIMD the two add up.  |error| <= ~2e-7 absolute on values in [-1, 1] (tanh near 0 through 1 - 2 / (1 +
// e^2x): absolute, not relative); SRL_RNN_FASTMATH=0 at build time restores libm.
#ifndef SRL_RNN_FASTMATH
#define SRL_RNN_FASTMATH 1
#endif
#if SRL_RNN_FASTMATH
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177792681f * x)); }
#else
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return tanhf(x); }
#endif
__device__ __forceinline__ int ch_of(int e, int hb) { return (e & 3) + 8 * (e >> 2) + 4 * hb; }

struct SeqArgs {
  float* pre;            // LSTM: [C][N][4H] = W_ih x + b_ih on entry, activated gates on exit (forward) / d pre (backward)
                         // GRU: gi [C][N][3H] likewise (gates r, z, n / d gi)
  float* gh;             // GRU: [C][N][3H]: written by the forward (its n part is kept for the backward), d gh by the backward
  const float* w_hh;     // [G H][H]
  const float* b_hh;     // [G H] or NULL
  float* hin;            // [C][N][H]: hin[0] filled by the caller (masked initial state); steps 1.. written here
  float* cin;            // LSTM: [C][N][H] likewise
  float* y;              // [C][N][H]
  float* cnew;           // LSTM: [C][N][H]
  const uint8_t* reset;  // [C][N] or NULL: reset[c] masks the state ENTERING step c
  const float* dy;       // backward: [C][N][ld_dy] gradient w.r.t. y
  long ld_dy;
  long N;
  int H, C;
};

// 32 consecutive columns [col0, col0 + 32) of row `p` (row-major) <-> the accumulator layout: register 4 j + q = column 8 j + 4 hb + q
__device__ __forceinline__ void blk_load(const float* p, int col0, int hb, bool ok, float (&v)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) q = *reinterpret_cast<const float4*>(p + col0 + 8 * j + 4 * hb);
    v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
  }
}
__device__ __forceinline__ void blk_store(float* p, int col0, int hb, bool ok, const float (&v)[16]) {
  if (!ok) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(p + col0 + 8 * j + 4 * hb) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}

// W_hh as A-operand fragments.  Forward: out block ob (32 of the G H gate columns), in block ib (32 state units), register e:
// lane (o, kk) holds W[32 ob + o][32 ib + ch_of(e, kk)].  Backward (transposed): out block ub (state units), in block jb (gate
// columns): lane (u, kk) holds W[32 jb + ch_of(e, kk)][32 ub + u].
template <int G, int H>
__device__ __forceinline__ void stage_w(const float* w, float* sm, int tid, bool transposed) {
  constexpr int NB = H / 32, NO = G * NB;
  for (int e = tid; e < NO * NB * 1024; e += 256) {
    const int l = e & 63, e16 = (e >> 6) & 15, blk = e >> 10;
    if (!transposed) {
      const int ib = blk % NB, ob = blk / NB;
      sm[e] = w[(32 * ob + (l & 31)) * H + 32 * ib + ch_of(e16, l >> 5)];
    } else {
      const int jb = blk % NO, ub = blk / NO;
      sm[e] = w[(32 * jb + ch_of(e16, l >> 5)) * H + 32 * ub + (l & 31)];
    }
  }
}

// KIND 1: LSTM (gates i | f | g | o), 0: GRU (r | z | n)
template <int H, int KIND>
__global__ __launch_bounds__(256, 1) void rnn_seq_fwd_kernel(SeqArgs a) {
  constexpr int G = KIND ? 4 : 3, NB = H / 32;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* bt = sm + G * NB * NB * 1024;  // bias table [G H]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  stage_w<G, H>(a.w_hh, sm, tid, false);
  for (int c = tid; c < G * H; c += 256) bt[c] = a.b_hh ? a.b_hh[c] : 0.f;
  __syncthreads();
  const long ntiles = (a.N + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool ok = row < a.N;
    float h[NB][16], cs[NB][16];
#pragma unroll
    for (int ub = 0; ub < NB; ++ub) {
      blk_load(a.hin + row * H, 32 * ub, hb, ok, h[ub]);
      if (KIND) blk_load(a.cin + row * H, 32 * ub, hb, ok, cs[ub]);
    }
    for (int c = 0; c < a.C; ++c) {
      const long base = (long)c * a.N + row;
      const bool nxt = c + 1 < a.C;
      const bool rs = nxt && a.reset && ok && a.reset[(long)(c + 1) * a.N + row];
      float hn[NB][16];
#pragma unroll
      for (int ub = 0; ub < NB; ++ub) {
        // the G gate blocks of this unit block: W_hh h + b_hh
        f32x16 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int ob = g * NB + ub;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 b4 = *reinterpret_cast<const float4*>(bt + 32 * ob + 8 * j + 4 * hb);
            acc[g][4 * j] = b4.x; acc[g][4 * j + 1] = b4.y; acc[g][4 * j + 2] = b4.z; acc[g][4 * j + 3] = b4.w;
          }
#pragma unroll
          for (int ib = 0; ib < NB; ++ib) {
            const float* wfr = sm + (ob * NB + ib) * 1024 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(wfr[64 * e], h[ib][e], acc[g], 0, 0, 0);
          }
        }
        float p[G][16];
#pragma unroll
        for (int g = 0; g < G; ++g) blk_load(a.pre + base * G * H, g * H + 32 * ub, hb, ok, p[g]);
        float yv[16], c2[16];
        if (KIND) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float gi = sigm(p[0][e] + acc[0][e]), gf = sigm(p[1][e] + acc[1][e]);
            const float gg = tanh_(p[2][e] + acc[2][e]), go = sigm(p[3][e] + acc[3][e]);
            c2[e] = gf * cs[ub][e] + gi * gg;
            yv[e] = go * tanh_(c2[e]);
            p[0][e] = gi; p[1][e] = gf; p[2][e] = gg; p[3][e] = go;
          }
          blk_store(a.cnew + base * H, 32 * ub, hb, ok, c2);
#pragma unroll
          for (int e = 0; e < 16; ++e) cs[ub][e] = rs ? 0.f : c2[e];
          if (nxt) blk_store(a.cin + ((long)(c + 1) * a.N + row) * H, 32 * ub, hb, ok, cs[ub]);
        } else {
          float q[3][16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float rr = sigm(p[0][e] + acc[0][e]), z = sigm(p[1][e] + acc[1][e]);
            const float n = tanh_(p[2][e] + rr * acc[2][e]);
            yv[e] = (1.0f - z) * n + z * h[ub][e];
            p[0][e] = rr; p[1][e] = z; p[2][e] = n;
            q[0][e] = acc[0][e]; q[1][e] = acc[1][e]; q[2][e] = acc[2][e];
          }
#pragma unroll
          for (int g = 0; g < 3; ++g) blk_store(a.gh + base * 3 * H, g * H + 32 * ub, hb, ok, q[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) blk_store(a.pre + base * G * H, g * H + 32 * ub, hb, ok, p[g]);
        blk_store(a.y + base * H, 32 * ub, hb, ok, yv);
#pragma unroll
        for (int e = 0; e < 16; ++e) hn[ub][e] = rs ? 0.f : yv[e];
        if (nxt) blk_store(a.hin + ((long)(c + 1) * a.N + row) * H, 32 * ub, hb, ok, hn[ub]);
      }
#pragma unroll
      for (int ub = 0; ub < NB; ++ub)
#pragma unroll
        for (int e = 0; e < 16; ++e) h[ub][e] = hn[ub][e];
    }
  }
}

template <int H, int KIND>
__global__ __launch_bounds__(256, 1) void rnn_seq_bwd_kernel(SeqArgs a) {
  constexpr int G = KIND ? 4 : 3, NB = H / 32, NO = G * NB;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hb = lane >> 5;
  stage_w<G, H>(a.w_hh, sm, tid, true);
  __syncthreads();
  const long ntiles = (a.N + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
    const long row = tile * 32 + r;
    const bool ok = row < a.N;
    float ch[NB][16], cc[NB][16];  // d loss / d (h_in, c_in) of the step after the current one
#pragma unroll
    for (int ub = 0; ub < NB; ++ub)
#pragma unroll
      for (int e = 0; e < 16; ++e) ch[ub][e] = cc[ub][e] = 0.f;
    for (int c = a.C - 1; c >= 0; --c) {
      const long base = (long)c * a.N + row;
      const bool keep = c + 1 < a.C && !(a.reset && ok && a.reset[(long)(c + 1) * a.N + row]);
      float dq[NO][16];      // what multiplies W_hh: d pre (LSTM) / d gh (GRU), gate block g * NB + ub
      float direct[NB][16];  // GRU: the part of d h_in that does not go through W_hh
#pragma unroll
      for (int ub = 0; ub < NB; ++ub) {
        float dyv[16], g[G][16];
        blk_load(a.dy ? a.dy + base * a.ld_dy : a.pre, 32 * ub, hb, ok && a.dy, dyv);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) blk_load(a.pre + base * G * H, gg * H + 32 * ub, hb, ok, g[gg]);
        if (KIND) {
          float cn[16], ci[16];
          blk_load(a.cnew + base * H, 32 * ub, hb, ok, cn);
          blk_load(a.cin + base * H, 32 * ub, hb, ok, ci);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float dh = dyv[e], dc = 0.f;
            if (keep) { dh += ch[ub][e]; dc = cc[ub][e]; }
            const float gi = g[0][e], gf = g[1][e], gc = g[2][e], go = g[3][e];
            const float tc = tanh_(cn[e]);
            dc += dh * go * (1.0f - tc * tc);
            g[0][e] = dc * gc * gi * (1.0f - gi);
            g[1][e] = dc * ci[e] * gf * (1.0f - gf);
            g[2][e] = dc * gi * (1.0f - gc * gc);
            g[3][e] = dh * tc * go * (1.0f - go);
            cc[ub][e] = dc * gf;
          }
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) {
            blk_store(a.pre + base * 4 * H, gg * H + 32 * ub, hb, ok, g[gg]);
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[gg * NB + ub][e] = g[gg][e];
          }
        } else {
          float qn[16], hi[16];
          blk_load(a.gh + base * 3 * H, 2 * H + 32 * ub, hb, ok, qn);
          blk_load(a.hin + base * H, 32 * ub, hb, ok, hi);
          float q[3][16];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float dh = dyv[e];
            if (keep) dh += ch[ub][e];
            const float rr = g[0][e], z = g[1][e], n = g[2][e];
            const float dn = dh * (1.0f - z);
            const float dz = dh * (hi[e] - n);
            const float dpn = dn * (1.0f - n * n);
            const float dpr = dpn * qn[e] * rr * (1.0f - rr);
            const float dpz = dz * z * (1.0f - z);
            g[0][e] = dpr; g[1][e] = dpz; g[2][e] = dpn;
            q[0][e] = dpr; q[1][e] = dpz; q[2][e] = dpn * rr;
            direct[ub][e] = dh * z;
          }
#pragma unroll
          for (int gg = 0; gg < 3; ++gg) {
            blk_store(a.pre + base * 3 * H, gg * H + 32 * ub, hb, ok, g[gg]);
            blk_store(a.gh + base * 3 * H, gg * H + 32 * ub, hb, ok, q[gg]);
#pragma unroll
            for (int e = 0; e < 16; ++e) dq[gg * NB + ub][e] = q[gg][e];
          }
        }
      }
      // d h_in(c)[unit] = (GRU: direct +) sum over gate columns j of dq[j] W_hh[j][unit]
#pragma unroll
      for (int ub = 0; ub < NB; ++ub) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = KIND ? 0.f : direct[ub][e];
#pragma unroll
        for (int jb = 0; jb < NO; ++jb) {
          const float* wfr = sm + (ub * NO + jb) * 1024 + lane;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wfr[64 * e], dq[jb][e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) ch[ub][e] = acc[e];
      }
    }
  }
}

template <int H, int KIND>
int launch_seq(void* stream, const SeqArgs& a, bool bwd) {
  constexpr int G = KIND ? 4 : 3, NB = H / 32;
  const int lds = (int)sizeof(float) * (G * NB * NB * 1024 + G * H);
  long blocks = srl_ceil_div(a.N, 128L);
  if (blocks > 256) blocks = 256;
  if (bwd) {
    auto kern = rnn_seq_bwd_kernel<H, KIND>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
  } else {
    auto kern = rnn_seq_fwd_kernel<H, KIND>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, a);
  }
  return 0;
}

}  // namespace

extern "C" int srl_rnn_seq_supported(int kind, int H) { return (kind == 0 || kind == 1) && (H == 32 || H == 64); }

extern "C" int srl_lstm_seq_fwd(void* stream, float* pre, const float* w_hh, const float* b_hh, float* hin, float* cin,
                                const uint8_t* reset, int64_t N, int H, int C, float* y, float* cnew) {
  SRL_CHECK_ARG(srl_rnn_seq_supported(1, H) && N >= 0 && C >= 1, "H must be 32 or 64");
  SRL_CHECK_ARG(pre && w_hh && hin && cin && y && cnew, "null tensor");
  if (N == 0) return 0;
  SeqArgs a{};
  a.pre = pre; a.w_hh = w_hh; a.b_hh = b_hh; a.hin = hin; a.cin = cin; a.reset = reset; a.N = N; a.H = H; a.C = C; a.y = y; a.cnew = cnew;
  if (H == 32) launch_seq<32, 1>(stream, a, false);
  else launch_seq<64, 1>(stream, a, false);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_lstm_seq_bwd(void* stream, const float* dy, int64_t ld_dy, float* gates, const float* w_hh, const float* cin,
                                const float* cnew, const uint8_t* reset, int64_t N, int H, int C) {
  SRL_CHECK_ARG(srl_rnn_seq_supported(1, H) && N >= 0 && C >= 1, "H must be 32 or 64");
  SRL_CHECK_ARG(gates && w_hh && cin && cnew && (!dy || ld_dy >= H), "null tensor");
  if (N == 0) return 0;
  SeqArgs a{};
  a.pre = gates; a.w_hh = w_hh; a.cin = const_cast<float*>(cin); a.cnew = const_cast<float*>(cnew); a.reset = reset; a.dy = dy;
  a.ld_dy = ld_dy; a.N = N; a.H = H; a.C = C;
  if (H == 32) launch_seq<32, 1>(stream, a, true);
  else launch_seq<64, 1>(stream, a, true);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gru_seq_fwd(void* stream, float* gi, float* gh, const float* w_hh, const float* b_hh, float* hin,
                               const uint8_t* reset, int64_t N, int H, int C, float* y) {
  SRL_CHECK_ARG(srl_rnn_seq_supported(0, H) && N >= 0 && C >= 1, "H must be 32 or 64");
  SRL_CHECK_ARG(gi && gh && w_hh && hin && y, "null tensor");
  if (N == 0) return 0;
  SeqArgs a{};
  a.pre = gi; a.gh = gh; a.w_hh = w_hh; a.b_hh = b_hh; a.hin = hin; a.reset = reset; a.N = N; a.H = H; a.C = C; a.y = y;
  if (H == 32) launch_seq<32, 0>(stream, a, false);
  else launch_seq<64, 0>(stream, a, false);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gru_seq_bwd(void* stream, const float* dy, int64_t ld_dy, float* gates, float* gh, const float* w_hh,
                               const float* hin, const uint8_t* reset, int64_t N, int H, int C) {
  SRL_CHECK_ARG(srl_rnn_seq_supported(0, H) && N >= 0 && C >= 1, "H must be 32 or 64");
  SRL_CHECK_ARG(gates && gh && w_hh && hin && (!dy || ld_dy >= H), "null tensor");
  if (N == 0) return 0;
  SeqArgs a{};
  a.pre = gates; a.gh = gh; a.w_hh = w_hh; a.hin = const_cast<float*>(hin); a.reset = reset; a.dy = dy; a.ld_dy = ld_dy; a.N = N;
  a.H = H; a.C = C;
  if (H == 32) launch_seq<32, 0>(stream, a, true);
  else launch_seq<64, 0>(stream, a, true);
  SRL_LAUNCH_CHECK();
  return 0;
}
