// GRU cell kernels for the recurrent backbone (reference legacy/algorithm/modules/autoreset_rnn.py:42-66 wrapping
// torch.nn.GRU; recurrent_backbone.py:61-66).  The matrix products (W_ih x for all steps at once, W_hh h per step,
// and the three gradient GEMMs) run on srl_gemm; these kernels do what sits between them, one time step at a time:
//   forward   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r*gh_n), h = (1-z)*n + z*h_in
//             and the auto-reset of the NEXT step's input state, h_in' = h * (1 - on_reset')
//   backward  the same cell differentiated by hand, leaving d gi / d gh in place of the saved gates.
// Layout: every per-step block is [N, 3H] (gates r|z|n, torch order) or [N, H], rows = environment columns of
// the chunked batch; the time loop lives on the host (it is inherently serial), so each launch is fully parallel.
#include "srl_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_mask_state_kernel(const float* h, const uint8_t* reset, long N, int H,
                                                             float* out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * H; i += (long)gridDim.x * 256) {
    const long row = i / H;
    out[i] = (reset && reset[row]) ? 0.f : h[i];  // h * (1 - on_reset), autoreset_rnn.py:59
  }
}

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(float* gi, float* gh, const float* hin,
                                                           const uint8_t* reset_next, long N, int H, float* y,
                                                           float* hin_next) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * H; i += (long)gridDim.x * 256) {
    const long row = i / H;
    const int j = (int)(i - row * H);
    float* gi_r = gi + row * 3 * H;
    const float* gh_r = gh + row * 3 * H;
    const float r = sigmoidf_(gi_r[j] + gh_r[j]);
    const float z = sigmoidf_(gi_r[H + j] + gh_r[H + j]);
    const float n = tanhf(gi_r[2 * H + j] + r * gh_r[2 * H + j]);
    const float h = (1.0f - z) * n + z * hin[i];
    gi_r[j] = r, gi_r[H + j] = z, gi_r[2 * H + j] = n;  // saved for the backward pass (gh keeps its n part)
    y[i] = h;
    if (hin_next) hin_next[i] = (reset_next && reset_next[row]) ? 0.f : h;
  }
}

__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* dy, const float* carry,
                                                           const uint8_t* reset_next, float* gates, float* gh,
                                                           const float* hin, long N, int H, float* dh_direct) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * H; i += (long)gridDim.x * 256) {
    const long row = i / H;
    const int j = (int)(i - row * H);
    float* g = gates + row * 3 * H;
    float* q = gh + row * 3 * H;
    float dh = dy ? dy[i] : 0.f;
    if (carry && !(reset_next && reset_next[row])) dh += carry[i];  // d h_in(t+1) reaches h(t) unless t+1 was reset
    const float r = g[j], z = g[H + j], n = g[2 * H + j], ghn = q[2 * H + j];
    const float dn = dh * (1.0f - z);
    const float dz = dh * (hin[i] - n);
    const float dpn = dn * (1.0f - n * n);
    const float dpr = dpn * ghn * r * (1.0f - r);
    const float dpz = dz * z * (1.0f - z);
    g[j] = dpr, g[H + j] = dpz, g[2 * H + j] = dpn;      // d gi
    q[j] = dpr, q[H + j] = dpz, q[2 * H + j] = dpn * r;  // d gh
    dh_direct[i] = dh * z;                                // the part of d h_in that does not go through W_hh
  }
}

// LSTM (torch.nn.LSTM gate order i|f|g|o).  pre = W_ih x + b_ih + W_hh h_in + b_hh (both GEMMs accumulate into one
// block).  c' = s(f)*c_in + s(i)*tanh(g), h = s(o)*tanh(c').  pre is overwritten with the activated gates.
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float* pre, const float* cin, const uint8_t* reset_next,
                                                            long N, int H, float* y, float* cnew, float* hin_next,
                                                            float* cin_next) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * H; i += (long)gridDim.x * 256) {
    const long row = i / H;
    const int j = (int)(i - row * H);
    float* p = pre + row * 4 * H;
    const float gi = sigmoidf_(p[j]), gf = sigmoidf_(p[H + j]), gg = tanhf(p[2 * H + j]), go = sigmoidf_(p[3 * H + j]);
    const float c2 = gf * cin[i] + gi * gg;
    const float h = go * tanhf(c2);
    p[j] = gi, p[H + j] = gf, p[2 * H + j] = gg, p[3 * H + j] = go;
    y[i] = h;
    cnew[i] = c2;
    if (hin_next) {
      const bool rs = reset_next && reset_next[row];
      hin_next[i] = rs ? 0.f : h;
      cin_next[i] = rs ? 0.f : c2;
    }
  }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* dy, const float* carry_h, const float* carry_c,
                                                            const uint8_t* reset_next, float* gates, const float* cin,
                                                            const float* cnew, long N, int H, float* dc_in) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N * H; i += (long)gridDim.x * 256) {
    const long row = i / H;
    const int j = (int)(i - row * H);
    float* g = gates + row * 4 * H;
    const bool cut = reset_next && reset_next[row];
    float dh = dy ? dy[i] : 0.f;
    float dc = 0.f;
    if (carry_h && !cut) { dh += carry_h[i]; dc = carry_c[i]; }
    const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
    const float tc = tanhf(cnew[i]);
    dc += dh * go * (1.0f - tc * tc);
    g[j] = dc * gg * gi * (1.0f - gi);
    g[H + j] = dc * cin[i] * gf * (1.0f - gf);
    g[2 * H + j] = dc * gi * (1.0f - gg * gg);
    g[3 * H + j] = dh * tc * go * (1.0f - go);
    dc_in[i] = dc * gf;
  }
}

// dst row (c, k*B + b) <- src row ((k*C + c)*B + b)   (inverse: the other way round); rows of D floats
__global__ __launch_bounds__(256) void chunk_rows_kernel(const float* src, float* dst, int T, int B, int C, int D,
                                                         int inverse) {
  const long total = (long)T * B * D;
  const int K = T / C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D;
    const int d = (int)(i - row * D);
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    const int k = t / C, c = t - k * C;
    const long crow = (long)c * K * B + (long)k * B + b;
    if (inverse) dst[row * D + d] = src[crow * D + d];
    else dst[crow * D + d] = src[row * D + d];
  }
}

inline unsigned grid_for(long n) { return (unsigned)(srl_ceil_div(n, 256L) < 4096 ? srl_ceil_div(n, 256L) : 4096); }

}  // namespace

extern "C" int srl_gru_mask_state(void* stream, const float* h, const uint8_t* reset, long N, int H, float* out) {
  SRL_CHECK_ARG(N >= 0 && H >= 1, "bad extents");
  if (N == 0) return 0;
  SRL_CHECK_ARG(h && out, "null tensor");
  hipLaunchKernelGGL(gru_mask_state_kernel, dim3(grid_for(N * H)), dim3(256), 0, (hipStream_t)stream, h, reset, N, H, out);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gru_cell_fwd(void* stream, float* gi, float* gh, const float* hin, const uint8_t* reset_next, long N,
                                int H, float* y, float* hin_next) {
  SRL_CHECK_ARG(N >= 0 && H >= 1, "bad extents");
  if (N == 0) return 0;
  SRL_CHECK_ARG(gi && gh && hin && y, "null tensor");
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(grid_for(N * H)), dim3(256), 0, (hipStream_t)stream, gi, gh, hin,
                     reset_next, N, H, y, hin_next);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gru_cell_bwd(void* stream, const float* dy, const float* carry, const uint8_t* reset_next,
                                float* gates, float* gh, const float* hin, long N, int H, float* dh_direct) {
  SRL_CHECK_ARG(N >= 0 && H >= 1, "bad extents");
  if (N == 0) return 0;
  SRL_CHECK_ARG(gates && gh && hin && dh_direct, "null tensor");
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(grid_for(N * H)), dim3(256), 0, (hipStream_t)stream, dy, carry,
                     reset_next, gates, gh, hin, N, H, dh_direct);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_lstm_cell_fwd(void* stream, float* pre, const float* cin, const uint8_t* reset_next, long N, int H,
                                 float* y, float* cnew, float* hin_next, float* cin_next) {
  SRL_CHECK_ARG(N >= 0 && H >= 1, "bad extents");
  if (N == 0) return 0;
  SRL_CHECK_ARG(pre && cin && y && cnew && (!hin_next == !cin_next), "null tensor");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for(N * H)), dim3(256), 0, (hipStream_t)stream, pre, cin, reset_next,
                     N, H, y, cnew, hin_next, cin_next);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_lstm_cell_bwd(void* stream, const float* dy, const float* carry_h, const float* carry_c,
                                 const uint8_t* reset_next, float* gates, const float* cin, const float* cnew, long N,
                                 int H, float* dc_in) {
  SRL_CHECK_ARG(N >= 0 && H >= 1, "bad extents");
  if (N == 0) return 0;
  SRL_CHECK_ARG(gates && cin && cnew && dc_in && (!carry_h == !carry_c), "null tensor");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for(N * H)), dim3(256), 0, (hipStream_t)stream, dy, carry_h, carry_c,
                     reset_next, gates, cin, cnew, N, H, dc_in);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_chunk_rows(void* stream, const float* src, float* dst, int T, int B, int C, int D, int inverse) {
  SRL_CHECK_ARG(T >= 0 && B >= 0 && C >= 1 && D >= 1 && T % C == 0, "T must be a multiple of the chunk length");
  if (T == 0 || B == 0) return 0;
  SRL_CHECK_ARG(src && dst && src != dst, "null / aliased tensor");
  hipLaunchKernelGGL(chunk_rows_kernel, dim3(grid_for((long)T * B * D)), dim3(256), 0, (hipStream_t)stream, src, dst, T,
                     B, C, D, inverse);
  SRL_LAUNCH_CHECK();
  return 0;
}
