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

}  // namespace

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
