// Row gathers behind the HBM observation ring (srl_amd/runtime/obs_ring.py): the frames a rollout uploaded stay in a
// ring of fixed-size rows; a training sample names its rows by ring slot and the chunk's rows are gathered from there
// instead of crossing PCIe a second time.  HBM-bound copies: 2 x row_bytes per row.
#include "srl_common.h"

namespace {

// one workgroup per destination row, 16-byte pieces, the slot read once per row on the scalar unit
__global__ __launch_bounds__(256) void gather_rows16_kernel(const uint4* __restrict__ src, long row_vec,
                                                            const int32_t* __restrict__ index, long n,
                                                            uint4* __restrict__ dst) {
  for (long r = blockIdx.x; r < n; r += gridDim.x) {
    const long s = index[r];
    const uint4* in = src + s * row_vec;
    uint4* out = dst + r * row_vec;
    for (long j = threadIdx.x; j < row_vec; j += 256) out[j] = in[j];
  }
}

// narrow rows (vector observations, per-row statistics): one thread per 4-byte word
__global__ __launch_bounds__(256) void gather_rows4_kernel(const uint32_t* __restrict__ src, int row_words,
                                                           const int32_t* __restrict__ index, long total,
                                                           uint32_t* __restrict__ dst) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long r = e / row_words;
    const int j = (int)(e - r * row_words);
    dst[e] = src[(long)index[r] * row_words + j];
  }
}

// ring sequence numbers -> storage slots
__global__ __launch_bounds__(256) void ring_slots_kernel(const int64_t* __restrict__ refs, long n, long capacity, long base,
                                                         int32_t* __restrict__ slots) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) slots[i] = (int32_t)((refs[i] - base) % capacity);
}

// A frame-stacked observation (atari_wrappers.py:211-242: the k latest frames, the newest last; reset() fills the stack with k
// copies of the first frame) differs from the previous one of the same environment by ONE plane.  One workgroup per new row of
// the ring, in the space-to-depth layout the first layer reads ([H/4, W/4, (c, 4, 4)] bytes: 16 bytes per 4x4 block and
// channel): channels 0..C-2 are channels 1..C-1 of the previous row (read in place from the ring), channel C-1 is the uploaded
// plane, re-tiled through LDS; prev < 0: every channel is the plane.  The whole-observation LayerNorm statistics come out of
// the same pass (integer sums, exact: bit-identical to srl_obs_space_to_depth on the assembled stack).
__global__ __launch_bounds__(256) void stack_push_kernel(uint8_t* __restrict__ store, const uint8_t* __restrict__ planes,
                                                         const int32_t* __restrict__ prev, long slot0, int C, int H, int W,
                                                         float* __restrict__ mean, float* __restrict__ rstd, float eps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t plane_lds[];
  __shared__ double red[8];
  const long r = blockIdx.x;
  const int Wb = W / 4, HW = H * W, D = C * HW;
  const uint4* src4 = reinterpret_cast<const uint4*>(planes + r * HW);
  uint4* st4 = reinterpret_cast<uint4*>(plane_lds);
  for (int e = threadIdx.x; e < HW / 16; e += 256) st4[e] = src4[e];
  __syncthreads();
  const long p = prev[r];
  const uint4* old4 = reinterpret_cast<const uint4*>(store + (p < 0 ? 0 : p) * D);
  uint4* dst4 = reinterpret_cast<uint4*>(store + (slot0 + r) * D);
  uint32_t a = 0, b = 0;
  auto tally = [&](uint32_t w) {
    const unsigned b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
    a += b0 + b1 + b2 + b3;
    b += b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
  };
  for (int o4 = threadIdx.x; o4 < D / 16; o4 += 256) {  // o4 = (ab*Wb + bq)*C + c
    const int c = o4 % C, blk = o4 / C;
    uint4 q;
    if (c == C - 1 || p < 0) {
      const int bq = blk % Wb, ab = blk / Wb;
      const uint32_t* sp = plane_lds + (ab * 4) * Wb + bq;
      q = make_uint4(sp[0], sp[Wb], sp[2 * Wb], sp[3 * Wb]);
    } else {
      q = old4[o4 + 1];
    }
    dst4[o4] = q;
    tally(q.x); tally(q.y); tally(q.z); tally(q.w);
  }
  double acc[2] = {(double)a, (double)b};
  block_sum<2, 256>(acc, red);
  if (threadIdx.x == 0) {
    const double mu = acc[0] / D;
    double var = acc[1] / D - mu * mu;
    var = var > 0.0 ? var : 0.0;
    mean[slot0 + r] = (float)mu;
    rstd[slot0 + r] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

}  // namespace

extern "C" int srl_ring_stack_push(void* stream, void* store, const void* planes, const int32_t* prev, int64_t slot0, int64_t n,
                                   int C, int H, int W, float* mean, float* rstd) {
  SRL_CHECK_ARG(store && planes && prev && mean && rstd && n >= 0 && slot0 >= 0, "null tensor");
  SRL_CHECK_ARG(C >= 1 && H % 4 == 0 && W % 4 == 0 && (long)H * W % 16 == 0 && (long)H * W <= 60 * 1024 &&
                    (long)C * H * W * 255L * 255L < 0xffffffffL,
                "uint8 planes with H, W multiples of 4, at most 60 KB each, the stack's sum of squares below 2^32");
  SRL_CHECK_ARG((((uintptr_t)store | (uintptr_t)planes) & 15) == 0, "unaligned tensor");
  if (n == 0) return 0;
  hipLaunchKernelGGL(stack_push_kernel, dim3((unsigned)n), dim3(256), (size_t)H * W, (hipStream_t)stream, static_cast<uint8_t*>(store),
                     static_cast<const uint8_t*>(planes), prev, (long)slot0, C, H, W, mean, rstd, 1e-5f);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_ring_slots(void* stream, const int64_t* refs, int64_t n, int64_t capacity, int64_t base, int32_t* slots) {
  SRL_CHECK_ARG(refs && slots && n >= 0 && capacity > 0 && capacity <= 0x7fffffffL, "null tensor / capacity out of range");
  if (n == 0) return 0;
  long blocks = srl_ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ring_slots_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, refs, (long)n, (long)capacity, (long)base, slots);
  SRL_LAUNCH_CHECK();
  return 0;
}

extern "C" int srl_gather_rows(void* stream, const void* src, int64_t row_bytes, const int32_t* index, int64_t n, void* dst) {
  SRL_CHECK_ARG(src && index && dst, "null tensor");
  SRL_CHECK_ARG(row_bytes > 0 && row_bytes % 4 == 0 && n >= 0, "row_bytes must be a positive multiple of 4");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 && row_bytes >= 1024;
  if (vec) {
    const long blocks = n < (1L << 20) ? n : (1L << 20);
    hipLaunchKernelGGL(gather_rows16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint4*>(src),
                       (long)(row_bytes / 16), index, (long)n, static_cast<uint4*>(dst));
  } else {
    SRL_CHECK_ARG(row_bytes / 4 <= 0x7fffffffL, "row too wide for the word gather");
    const long total = n * (row_bytes / 4);
    long blocks = srl_ceil_div(total, 256);
    if (blocks > (1L << 20)) blocks = 1L << 20;
    hipLaunchKernelGGL(gather_rows4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint32_t*>(src),
                       (int)(row_bytes / 4), index, total, static_cast<uint32_t*>(dst));
  }
  SRL_LAUNCH_CHECK();
  return 0;
}
