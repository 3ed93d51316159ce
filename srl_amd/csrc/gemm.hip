// Dense float32 GEMM entry point (srl_gemm).  Kernels: gemm_bf16x3.h (bf16 matrix cores, three exact pieces per
// operand) for float4-stageable shapes, gemm_core.h (float32 MFMA) otherwise, skinny.h for extents <= 16.
#include <stdlib.h>

#include "gemm_core.h"
#include "gemm_bf16x3.h"
#include "skinny.h"

static_assert(sizeof(srl_gemm_desc) == 208 && sizeof(srl_ppo_hparams) == 44, "ABI struct layout (mirrored in srl_amd/hip.py)");

using namespace srlgemm;

namespace {

template <int BM, int BN, int WM, int WN, bool GEN>
int launch_or(hipStream_t st, const GemmArgs& g, int akm, int bkm, int split) {
  if (!akm && !bkm) return launch<BM, BN, WM, WN, false, false, SRC_PLAIN, SRC_PLAIN, GEN>(st, g, 1, split);
  if (!akm && bkm) return launch<BM, BN, WM, WN, false, true, SRC_PLAIN, SRC_PLAIN, GEN>(st, g, 1, split);
  if (akm && bkm) return launch<BM, BN, WM, WN, true, true, SRC_PLAIN, SRC_PLAIN, GEN>(st, g, 1, split);
  return launch<BM, BN, WM, WN, true, false, SRC_PLAIN, SRC_PLAIN, GEN>(st, g, 1, split);
}

// float32 operands on the bf16 matrix cores (gemm_bf16x3.h): float4-stageable operands only.  k-steps of 16 (48 KB of LDS per
// 128x128 workgroup, three per CU): k-steps of 32 measured 10-25 % slower, 256x128 tiles +12 % on 4096^3 but not on the
// network's shapes, 128x64 / 64x128 tiles (full occupancy for the FC forward) -8 %.
template <int BM, int BN, int WM, int WN, int NP = 3, int KB = 16>
int launch3_or(hipStream_t st, const GemmArgs& g, int akm, int bkm, int split) {
  if (!akm && !bkm) return launch3<BM, BN, WM, WN, false, false, SRC_PLAIN, SRC_PLAIN, KB, NP>(st, g, 1, split);
  if (!akm && bkm) return launch3<BM, BN, WM, WN, false, true, SRC_PLAIN, SRC_PLAIN, KB, NP>(st, g, 1, split);
  if (akm && bkm) return launch3<BM, BN, WM, WN, true, true, SRC_PLAIN, SRC_PLAIN, KB, NP>(st, g, 1, split);
  return launch3<BM, BN, WM, WN, true, false, SRC_PLAIN, SRC_PLAIN, KB, NP>(st, g, 1, split);
}

// both operands float4-loadable (aligned, leading dimensions multiples of 4): the lean instantiation
template <int BM, int BN, int WM, int WN>
int launch_cfg(hipStream_t st, const GemmArgs& g, int akm, int bkm, int split) {
  if (g.vec_a && g.vec_b) return launch_or<BM, BN, WM, WN, false>(st, g, akm, bkm, split);
  return launch_or<BM, BN, WM, WN, true>(st, g, akm, bkm, split);
}

inline bool small_gemm() {  // SRL_SMALL_GEMM=0: the tiny products on the general kernels (A/B)
  const char* e = getenv("SRL_SMALL_GEMM");
  return !(e && e[0] == '0');
}

}  // namespace

extern "C" int srl_gemm(void* stream, const srl_gemm_desc* d) {
  SRL_CHECK_ARG(d != nullptr, "null descriptor");
  SRL_CHECK_ARG(d->M >= 0 && d->N >= 0 && d->K >= 0, "negative extent");
  SRL_CHECK_ARG(d->A && d->B && d->C, "null matrix");
  const int split = d->split_k > 1 ? d->split_k : 1;
  SRL_CHECK_ARG(split == 1 || d->workspace, "split_k > 1 needs a workspace");
  SRL_CHECK_ARG(split == 1 || (!d->bias && !d->act && !d->dact_src && !d->dact_mask && !d->mask_out),
                "split_k supports only the accumulate epilogue");
  SRL_CHECK_ARG(!d->mask_out || (d->act == 1 && d->N % 32 == 0 && d->ldc % 32 == 0),
                "mask_out: ReLU outputs with N and ldc multiples of 32");
  SRL_CHECK_ARG(!d->dact_mask || (d->dact == 1 && !d->dact_src && d->N % 32 == 0 && d->ld_dact % 32 == 0),
                "dact_mask: the ReLU derivative, instead of dact_src; N and ld_dact multiples of 32");
  if (d->M == 0 || d->N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (srlskinny::try_skinny(st, d)) {  // one extent <= 16: bandwidth kernels instead of padded MFMA tiles
    srl_count_dispatch(SRL_DISP_SKINNY);
    SRL_LAUNCH_CHECK();
    return 0;
  }

  GemmArgs g{};
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.a = plain_src(d->A, d->lda);
  g.b = plain_src(d->B, d->ldb);
  g.bias = d->bias; g.act = d->act;
  g.range_a = d->a_absmax; g.range_b = d->b_absmax;
  g.dact_src = d->dact_src; g.ld_dact = d->ld_dact; g.dact = d->dact;
  g.mask_out = d->mask_out; g.dact_mask = d->dact_mask;
  g.b_presplit = d->b_presplit;
  g.b_h2_scale = d->b_h2_scale;
  const int nsplit = plan_split(d->K, split, &g.k_per_split);
  g.o = OutDesc{};
  g.o.f_img = g.o.f_line = make_fastdiv(1);
  if (nsplit > 1) {
    g.o.out = d->workspace; g.o.ldo = d->N; g.slab = d->M * d->N; g.accumulate = 0;
  } else {
    g.o.out = d->C; g.o.ldo = d->ldc; g.slab = 0; g.accumulate = d->accumulate;
  }
  // float4 staging needs 16-byte aligned rows and a contiguous extent that is a multiple of 4
  const long a_contig = d->a_kmajor ? d->M : d->K, b_contig = d->b_kmajor ? d->N : d->K;
  g.vec_a = aligned16(d->A) && d->lda % 4 == 0 && a_contig % 4 == 0;
  g.vec_b = aligned16(d->B) && d->ldb % 4 == 0 && b_contig % 4 == 0;
  if (d->a_colsum) {
    SRL_CHECK_ARG(d->a_kmajor && g.vec_a && g.vec_b,
                  "a_colsum needs a k-major A and float4-stageable operands (16-byte aligned, pitches % 4 == 0)");
    g.a_colsum = d->a_colsum;
  }

  SRL_CHECK_ARG(!(d->mask_out || d->dact_mask) || (g.vec_a && g.vec_b && !d->a_kmajor && (d->mask_out ? !d->b_kmajor : d->b_kmajor)),
                "sign masks: float4-stageable operands; mask_out in the forward orientation (a_kmajor 0, b_kmajor 0), dact_mask in "
                "the data-gradient one (0, 1)");
  int rc;
  if (nsplit == 1) g.out_absmax = d->out_absmax;
  const bool small = use_bf16x3() && small_gemm() && g.vec_a && g.vec_b && d->M * d->N <= 65536 && d->K >= 4 && d->K <= 512;
  const bool two = !small && use_bf16x3() && d->a_absmax && (d->b_absmax || d->b_h2_scale) && use_f16x2() && g.vec_a && g.vec_b &&
                   d->M > 64 && d->N > 64 && d->K >= 64;
  SRL_CHECK_ARG(!d->b_h2_scale || (two && d->b_kmajor && d->N % 32 == 0 && d->ldb == d->N && !d->b_presplit),
                "b_h2_scale: a k-major dense B of h2p rows (N a multiple of 32, ldb == N) on the two-piece kernel (a_absmax given)");
  SRL_CHECK_ARG(!d->b_presplit || (two && nsplit >= 1),
                "b_presplit: only products that take the two-piece kernel (both ranges, M > 64, N > 64, K >= 64, aligned operands, "
                "more than 65 536 outputs or K > 512)");
  if (small) {
    // A product of a few thousand outputs (the layers of the CartPole-sized configurations) is a latency chain: on 256 x 64
    // tiles it is ONE workgroup walking K in 16-deep steps, a memory latency each (9-11 us for 256 x 64 x 64).  64 x 64
    // tiles with 64-deep steps: several workgroups, and K <= 64 arrives with one round of loads.
    rc = launch3_or<64, 64, 2, 2, 3, 64>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  } else
  if (two) {
    // the caller knows both operands' ranges: two f16 pieces per operand, three products (gemm_bf16x3.h, NP == 2)
    rc = launch3_or<128, 128, 2, 2, 2>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  } else
  if (use_bf16x3() && g.vec_a && g.vec_b && d->M > 64 && d->N > 32 && d->K >= 64) {
    // bf16 matrix cores, three exact pieces per float32 operand (2.67x fewer matrix-pipe cycles)
    rc = d->N > 64 ? launch3_or<128, 128, 2, 2>(st, g, d->a_kmajor, d->b_kmajor, nsplit)
                   : launch3_or<256, 64, 4, 1>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  } else
  // long reductions over big outputs: 8-wavefront workgroups on 256x128 tiles (measured +6 % at K = 3136; on the
  // K = 512 shapes the 4-wavefront 128x128 tiles are faster)
  if (d->N > 64 && d->M >= 4096 && d->K >= 2048 && g.vec_a && g.vec_b && nsplit == 1)
    rc = launch_or<256, 128, 4, 2, false>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else if (d->N > 64 && d->M > 64) rc = launch_cfg<128, 128, 2, 2>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else if (d->N > 64) rc = launch_cfg<32, 256, 1, 4>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else if (d->N > 32) rc = launch_cfg<256, 64, 4, 1>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  else rc = launch_cfg<256, 32, 4, 1>(st, g, d->a_kmajor, d->b_kmajor, nsplit);
  SRL_CHECK_ARG(rc == 0, "grid too large (or sign masks on a kernel without them)");
  SRL_LAUNCH_CHECK();
  if (nsplit > 1) {
    reduce_slabs(st, d->workspace, nsplit, 1L, d->M, d->N, d->C, d->ldc, 0L, d->accumulate);
    SRL_LAUNCH_CHECK();
  }
  return 0;
}
