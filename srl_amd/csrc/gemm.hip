// FP32 contraction on the CDNA4 matrix cores: C[M,N] = epilogue(sum_k A(i,k) B(k,j)).
//
// v_mfma_f32_32x32x2_f32 (exact float32 FMA chains, 64 FLOP/clk/SIMD = the FP32 peak of gfx950):
// lane l of a wavefront supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the 32x32 result
// sits in 16 accumulator registers per lane with col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// A workgroup of 4 wavefronts (WM x WN) owns a BM x BN output tile and walks K in steps of BK = 32.
// Both operand tiles are kept "k-major" in LDS ([BK][BM+pad], [BK][BN+pad]) whatever their layout in
// memory, so the MFMA feed is always one conflict-free ds_read_b32 per operand per step (32
// consecutive floats per half-wave).  The orientation of an operand only changes how it is staged:
//   k-contiguous source (rows = output index): float4 along k, transposed on the LDS store
//                                              (row pitch BM+1: conflict-free scalar stores)
//   k-major source      (rows = k):            float4 along the output index, ds_write_b128
// That one kernel therefore serves forward (X W^T), data gradient (dZ W) and weight gradient
// (dZ^T X) without transposed copies of anything.  Global loads for tile t+1 are issued before the
// MFMAs of tile t and written to LDS after them (register double buffering).
#include "srl_common.h"

static_assert(sizeof(srl_gemm_desc) == 144 && sizeof(srl_ppo_hparams) == 44, "ABI struct layout (mirrored in srl_amd/hip.py)");

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  long M, N, K;
  const float* A; long lda;
  const float* B; long ldb;
  float* out; long ldo; long slab;  // out + z*slab
  const float* bias;
  const float* dact_src; long ld_dact;
  int act, dact, accumulate;
  long k_per_split;
  int vec_a, vec_b;
  int tiles_n;
};

constexpr int BK = 32;

// ---- staging of one operand tile: BX output indices x BK reduction indices ------------------------
template <int BX, bool KMAJOR>
struct Stage {
  static constexpr int LD = KMAJOR ? BX + 4 : BX + 1;  // LDS row pitch (floats)
  static constexpr int NF = BX * BK / 256;             // floats per thread
  static constexpr int NV = NF / 4;                    // float4 per thread
  float r[NF];

  // src: operand base; ld: its row pitch; x0: first output index of the tile; xn: extent of that dim;
  // k0: first k of the tile; kend: end of this split's k range.
  __device__ __forceinline__ void load(const float* __restrict__ src, long ld, long x0, long xn, long k0, long kend,
                                       bool vec) {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * 256;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!KMAJOR) {
          const long x = x0 + u / 8, k = k0 + (u % 8) * 4;
          if (x < xn && k < kend) v = *reinterpret_cast<const float4*>(src + x * ld + k);
        } else {
          const long k = k0 + u / (BX / 4), x = x0 + (u % (BX / 4)) * 4;
          if (x < xn && k < kend) v = *reinterpret_cast<const float4*>(src + k * ld + x);
        }
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int e = tid + q * 256;
        float v = 0.f;
        if (!KMAJOR) {
          const long x = x0 + e / BK, k = k0 + e % BK;
          if (x < xn && k < kend) v = src[x * ld + k];
        } else {
          const long k = k0 + e / BX, x = x0 + e % BX;
          if (x < xn && k < kend) v = src[k * ld + x];
        }
        r[q] = v;
      }
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ lds, bool vec) const {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int u = tid + q * 256;
        if (!KMAJOR) {
          const int x = u / 8, k = (u % 8) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) lds[(k + j) * LD + x] = r[4 * q + j];
        } else {
          const int k = u / (BX / 4), x = (u % (BX / 4)) * 4;
          *reinterpret_cast<float4*>(lds + k * LD + x) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NF; ++q) {
        const int e = tid + q * 256;
        if (!KMAJOR) lds[(e % BK) * LD + e / BK] = r[q];
        else lds[(e / BX) * LD + e % BX] = r[q];
      }
    }
  }
};

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 wavefronts per workgroup");
  using SA = Stage<BM, AKM>;
  using SB = Stage<BN, BKM>;
  __shared__ __attribute__((aligned(16))) float lds[BK * SA::LD + BK * SB::LD + 8];
  float* As = lds;
  float* Bs = lds + ((BK * SA::LD + 3) & ~3);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const long tile_m = blockIdx.x / g.tiles_n, tile_n = blockIdx.x % g.tiles_n;
  const long m0 = tile_m * BM, n0 = tile_n * BN;
  const long kbeg = (long)blockIdx.z * g.k_per_split;
  const long kend = (kbeg + g.k_per_split < g.K) ? kbeg + g.k_per_split : g.K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  SA sa;
  SB sb;
  const bool va = g.vec_a != 0, vb = g.vec_b != 0;
  sa.load(g.A, g.lda, m0, g.M, kbeg, kend, va);
  sb.load(g.B, g.ldb, n0, g.N, kbeg, kend, vb);

  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile's LDS reads are done
    sa.store(As, va);
    sb.store(Bs, vb);
    __syncthreads();
    if (k0 + BK < kend) {  // prefetch the next tile; latency hides under the MFMAs below
      sa.load(g.A, g.lda, m0, g.M, k0 + BK, kend, va);
      sb.load(g.B, g.ldb, n0, g.N, k0 + BK, kend, vb);
    }
    const float* ap = As + h * SA::LD + wm * (TM * 32) + l31;
    const float* bp = Bs + h * SB::LD + wn * (TN * 32) + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = ap[kk * 2 * SA::LD + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = bp[kk * 2 * SB::LD + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  float* out = g.out + (long)blockIdx.z * g.slab;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long col = n0 + wn * (TN * 32) + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long row = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        v = act_apply(v, g.act);
        if (g.dact_src) v *= act_grad_from_output(g.dact_src[row * g.ld_dact + col], g.dact);
        float* dst = out + row * g.ldo + col;
        if (g.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* ws, int nslab, long M, long N, float* C,
                                                           long ldc, int accumulate) {
  const long total = M * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < nslab; ++z) s += ws[(long)z * total + e];
    float* dst = C + (e / N) * ldc + (e % N);
    *dst = accumulate ? *dst + s : s;
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(hipStream_t st, const GemmArgs& g, int akm, int bkm, int split) {
  const long tiles_m = srl_ceil_div(g.M, BM);
  GemmArgs a = g;
  a.tiles_n = (int)srl_ceil_div(g.N, BN);
  const long nblk = tiles_m * a.tiles_n;
  if (nblk > 0x7fffffffL) return -EINVAL;
  dim3 grid((unsigned)nblk, 1, (unsigned)split);
  if (!akm && !bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, a);
  else if (!akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, a);
  else if (akm && bkm) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, a);
  return 0;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int srl_gemm(void* stream, const srl_gemm_desc* d) {
  SRL_CHECK_ARG(d != nullptr, "null descriptor");
  SRL_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "negative extent");
  SRL_CHECK_ARG(d->A && d->B && d->C, "null matrix");
  const int split = d->split_k > 1 ? d->split_k : 1;
  SRL_CHECK_ARG(split == 1 || d->workspace, "split_k > 1 needs a workspace");
  SRL_CHECK_ARG(split == 1 || (!d->bias && !d->act && !d->dact_src), "split_k supports only the accumulate epilogue");
  if (d->M == 0 || d->N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;

  GemmArgs g{};
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.A = d->A; g.lda = d->lda;
  g.B = d->B; g.ldb = d->ldb;
  g.bias = d->bias; g.act = d->act;
  g.dact_src = d->dact_src; g.ld_dact = d->ld_dact; g.dact = d->dact;
  // k range per split: a multiple of BK so that float4 loads never straddle a split boundary
  long kps = srl_ceil_div(srl_ceil_div(d->K, split), BK) * BK;
  if (kps == 0) kps = BK;
  const int nsplit = (int)(srl_ceil_div(d->K, kps) > 0 ? srl_ceil_div(d->K, kps) : 1);
  g.k_per_split = kps;
  if (nsplit > 1) {
    g.out = d->workspace; g.ldo = d->N; g.slab = d->M * d->N; g.accumulate = 0;
  } else {
    g.out = d->C; g.ldo = d->ldc; g.slab = 0; g.accumulate = d->accumulate;
  }
  // float4 staging needs 16-byte aligned rows and a contiguous extent that is a multiple of 4
  const long a_contig = d->a_kmajor ? d->M : d->K, b_contig = d->b_kmajor ? d->N : d->K;
  g.vec_a = aligned16(d->A) && d->lda % 4 == 0 && a_contig % 4 == 0;
  g.vec_b = aligned16(d->B) && d->ldb % 4 == 0 && b_contig % 4 == 0;

  int rc;
  if (d->N > 64 && d->M > 64) rc = launch_cfg<128, 128, 2, 2>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else if (d->N > 64) rc = launch_cfg<32, 256, 1, 4>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else if (d->N > 32) rc = launch_cfg<256, 64, 4, 1>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else rc = launch_cfg<256, 32, 4, 1>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  SRL_CHECK_ARG(rc == 0, "grid too large");
  SRL_LAUNCH_CHECK();
  if (nsplit > 1) {
    const long total = d->M * d->N;
    const unsigned grid = (unsigned)(srl_ceil_div(total, 256) < 4096 ? srl_ceil_div(total, 256) : 4096);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid), dim3(256), 0, st, (const float*)d->workspace, nsplit, d->M,
                       d->N, d->C, d->ldc, d->accumulate);
    SRL_LAUNCH_CHECK();
  }
  return 0;
}
