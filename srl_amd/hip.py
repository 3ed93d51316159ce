"""ctypes binding of ``libsrlhip.so`` (the C ABI declared in ``include/srl_hip.h``).

PyTorch-ROCm is used for plumbing only: it owns device memory (``torch.empty(..., device="cuda")``) and
the HIP stream; every kernel is launched through the C ABI with raw device pointers on torch's
*current* stream.  There is no fallback: if the library is missing or a call fails, a ``HipError`` is
raised (the product path never computes on the CPU).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_long, c_uint64,
                    c_void_p)
from typing import Optional, Sequence

import torch

# SRL_HIP_LIB: another build of the same C ABI (kernel experiments, scripts/build_variant.sh); the default is the in-tree one
_LIB_PATH = os.environ.get("SRL_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsrlhip.so")
_lib = None

# The update runs on up to six streams (two row-chunk pipelines, each with a weight-gradient stream; the ingest copy stream;
# the collectives' stream).  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that
# share a queue serialise: with 4 the ingest ring's H2D copy queued behind a pipeline (host-fed update 337 instead of 264 ms),
# with 8 it does not, and the ring-fed update gains ~1 %.  The runtime reads it when libamdhip64 is LOADED (`import torch`), so
# this default only takes effect in processes that import srl_amd before torch; trainer entry points export it themselves
# (bench.py does, before its imports) -- INTEGRATION.md lists it with the other switches.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

ABI_VERSION = 18

# loss-term slots (srl_hip.h: SRL_LT_*)
LT_POLICY, LT_VALUE, LT_ENTROPY, LT_CLIP, LT_RATIO, LT_ADV, LT_RET, LT_MASK, LT_DONE, LT_TRUNC, LT_COUNT = range(11)
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
VALUE_LOSS_KINDS = {"mse": 0, "huber": 1, "smoothl1": 2}
MAX_HEADS = 8


class HipError(RuntimeError):
    pass


class PpoHparams(Structure):
    _fields_ = [("eps_clip", c_float), ("c_clip", c_float), ("value_eps_clip", c_float),
                ("value_loss_weight", c_float), ("entropy_bonus_weight", c_float), ("huber_delta", c_float),
                ("norm_eps", c_float), ("dual_clip", c_int32), ("clip_value", c_int32), ("value_loss", c_int32),
                ("mask_invert", c_int32)]


class GemmDesc(Structure):
    _fields_ = [("M", c_int64), ("N", c_int64), ("K", c_int64), ("A", c_void_p), ("lda", c_int64),
                ("a_kmajor", c_int32), ("B", c_void_p), ("ldb", c_int64), ("b_kmajor", c_int32), ("C", c_void_p),
                ("ldc", c_int64), ("bias", c_void_p), ("act", c_int32), ("dact_src", c_void_p), ("ld_dact", c_int64),
                ("dact", c_int32), ("accumulate", c_int32), ("split_k", c_int32), ("workspace", c_void_p),
                ("a_colsum", c_void_p), ("a_absmax", c_void_p), ("b_absmax", c_void_p), ("out_absmax", c_void_p),
                ("mask_out", c_void_p), ("dact_mask", c_void_p), ("b_presplit", c_int32), ("b_h2_scale", c_void_p)]


class MlpLayer(Structure):
    _fields_ = [("kind", c_int32), ("in_", c_int32), ("out", c_int32), ("act", c_int32), ("w", c_void_p), ("b", c_void_p),
                ("gw", c_void_p), ("gb", c_void_p)]


class ConvDesc(Structure):
    _fields_ = [("n", c_int64), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("KH", c_int32), ("KW", c_int32),
                ("stride", c_int32), ("Cout", c_int32), ("act", c_int32)]


_CD = POINTER(ConvDesc)

class H2ConvArgs(Structure):
    """``srl_h2_conv_args`` (include/srl_hip.h)."""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("sx", c_void_p), ("sw", c_void_p), ("n", c_int64), ("bias", c_void_p),
                ("act", c_int32), ("out", c_void_p), ("out_scale", c_void_p), ("bound_in", c_void_p), ("bound_w", c_void_p),
                ("bound_b", c_void_p), ("out_absmax", c_void_p), ("mask_out", c_void_p), ("mask_in", c_void_p)]


class H2GemmDesc(Structure):
    """``srl_h2_gemm_desc`` (include/srl_hip.h)."""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("sx", c_void_p), ("sw", c_void_p), ("M", c_int64), ("NC", c_int32), ("K", c_int32),
                ("bias", c_void_p), ("act", c_int32), ("out_h2", c_int32), ("out", c_void_p), ("out_scale", c_void_p),
                ("bound_in", c_void_p), ("bound_w", c_void_p), ("bound_b", c_void_p), ("out_absmax", c_void_p),
                ("mask_out", c_void_p), ("mask_in", c_void_p), ("mask_in_h2order", c_int32)]


_SIGNATURES = {
    "srl_conv2d_supported": (c_int, [_CD, c_int]),
    "srl_conv2d_nhwc_fwd": (c_int, [c_void_p, _CD] + [c_void_p] * 8 + [c_int]),
    "srl_presplit": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "srl_absmax": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "srl_relu_mask": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "srl_conv2d_small_supported": (c_int, [POINTER(ConvDesc)]),
    "srl_conv2d_small_fwd": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_conv2d_small_dgrad": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "srl_conv2d_small_wgrad_workspace": (c_int64, [POINTER(ConvDesc)]),
    "srl_conv2d_small_wgrad": (c_int, [c_void_p, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_dispatch_tiles": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "srl_mlp_tape_floats": (c_int64, [POINTER(MlpLayer), c_int]),
    "srl_mlp_tape_floats_at": (c_int64, [POINTER(MlpLayer), c_int, c_int64]),
    "srl_mlp_bwd_max_rows": (c_int64, [POINTER(MlpLayer), c_int]),
    "srl_mlp_fwd": (c_int, [c_void_p, POINTER(MlpLayer), c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "srl_mlp_bwd": (c_int, [c_void_p, POINTER(MlpLayer), c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "srl_mlp_bwd_dx": (c_int, [c_void_p, POINTER(MlpLayer), c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64]),
    "srl_conv2d_wgrad_workspace": (c_int64, [_CD]),
    "srl_conv2d_nhwc_wgrad": (c_int, [c_void_p, _CD, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_conv2d_dgrad_weight_elems": (c_int64, [_CD]),
    "srl_conv2d_dgrad_repack": (c_int, [c_void_p, _CD, c_void_p, c_void_p]),
    "srl_conv2d_nhwc_dgrad": (c_int, [c_void_p, _CD, c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 5 + [c_int]),
    "srl_conv2d_obs_fwd": (c_int, [c_void_p, _CD, c_void_p, c_int, c_int] + [c_void_p] * 11 + [c_int]),
    "srl_conv2d_obs_row_index_supported": (c_int, [_CD, c_int, c_int]),
    "srl_conv2d_obs_fwd_workspace": (c_int64, [_CD]),
    "srl_obs_space_to_depth": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p]),
    "srl_conv2d_obs_bwd_workspace": (c_int64, [_CD]),
    "srl_conv2d_obs_bwd": (c_int, [c_void_p, _CD, c_void_p, c_int, c_int] + [c_void_p] * 12 + [c_int, c_void_p]),
    "srl_abi_version": (c_int, []),
    "srl_last_error": (c_char_p, []),
    "srl_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "srl_dispatch_counts": (c_int, [POINTER(c_int64), c_int, c_int]),
    "srl_gae_scan": (c_int, [c_void_p] + [c_void_p] * 8 + [c_int, c_int, c_int, c_double, c_double, c_double,
                                                             c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_gae_scan_workspace_bytes": (c_long, [c_int, c_int]),
    "srl_masked_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "srl_masked_normalize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_double, c_int,
                                      c_void_p]),
    "srl_masked_stats_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "srl_fold_col_stats": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "srl_importance_ratio": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p]),
    "srl_popart_update": (c_int, [c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                  c_int]),
    "srl_popart_map": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_double, c_int, c_void_p]),
    "srl_gaussian_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "srl_gaussian_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "srl_gaussian_sample": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, c_uint64,
                                    c_uint64, c_void_p, c_void_p, c_int64]),
    "srl_gru_mask_state": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "srl_rnn_seq_supported": (c_int, [c_int, c_int]),
    "srl_lstm_seq_fwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p, c_void_p]),
    "srl_lstm_seq_bwd": (c_int, [c_void_p, c_void_p, c_int64] + [c_void_p] * 5 + [c_int64, c_int, c_int]),
    "srl_gru_seq_fwd": (c_int, [c_void_p] * 7 + [c_int64, c_int, c_int, c_void_p]),
    "srl_gru_seq_bwd": (c_int, [c_void_p, c_void_p, c_int64] + [c_void_p] * 5 + [c_int64, c_int, c_int]),
    "srl_gru_cell_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "srl_gru_cell_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int,
                                 c_void_p]),
    "srl_lstm_cell_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "srl_lstm_cell_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long,
                                  c_int, c_void_p]),
    "srl_chunk_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "srl_ppo_loss_fwd_bwd": (c_int, [c_void_p] + [c_void_p] * 8 + [c_long, c_int, POINTER(PpoHparams)] + [c_void_p] * 8),
    "srl_categorical_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_int, POINTER(c_int32),
                                     c_void_p, c_void_p]),
    "srl_categorical_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_int, POINTER(c_int32),
                                     c_void_p, c_void_p, c_void_p, c_int]),
    "srl_categorical_sample": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_long, c_int,
                                        POINTER(c_int32), c_uint64, c_uint64, c_void_p, c_void_p, c_int64]),
    "srl_categorical_log_softmax": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_long, c_int, POINTER(c_int32), c_void_p, c_int]),
    "srl_ppg_aux_loss_fwd_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_long, c_int, POINTER(c_int32), c_void_p,
                                          c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p]),
    "srl_gemm": (c_int, [c_void_p, POINTER(GemmDesc)]),
    "srl_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                   c_void_p, c_void_p]),
    "srl_ln_heads_supported": (c_int, [c_int, c_int, POINTER(c_int32)]),
    "srl_ln_heads_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, POINTER(c_void_p),
                                 POINTER(c_void_p), POINTER(c_int32), POINTER(c_void_p), POINTER(c_int64), c_void_p, c_void_p, c_int, c_int64,
                                 c_void_p, c_int, c_void_p, c_int64]),
    "srl_ln_heads_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 POINTER(c_void_p), POINTER(c_int32), POINTER(c_void_p), POINTER(c_int64), c_int, c_void_p, c_int64,
                                 c_void_p, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p]),
    "srl_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "srl_obs_ln_stats": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "srl_im2col_obs_ln": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64] +
                          [c_int] * 6 + [c_void_p]),
    "srl_im2col_nhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 6 + [c_void_p]),
    "srl_col2im_nhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 6 + [c_void_p, c_int, c_void_p]),
    "srl_obs_ln_affine_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64] + [c_int] * 6 +
                              [c_void_p, c_void_p]),
    "srl_colsum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int]),
    "srl_copy2d": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int]),
    "srl_u8_to_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "srl_gather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "srl_ring_slots": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "srl_ring_stack_push": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "srl_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "srl_accumulate_n": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int32, c_int64]),
    "srl_grad_sumsq": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "srl_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                               c_float, c_float, c_int, c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "srl_obs_ln_nhwc": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "srl_obs_ln_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                     c_void_p, c_void_p]),
    "srl_pad_nhwc": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "srl_crop_nhwc": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "srl_maxpool2_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "srl_maxpool2_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "srl_pad_ndhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 8 + [c_void_p]),
    "srl_crop_ndhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 8 + [c_void_p]),
    "srl_maxpool_ndhwc_fwd": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 7 + [c_void_p]),
    "srl_maxpool_ndhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64] + [c_int] * 8 + [c_void_p]),
    "srl_im2col_ndhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 8 + [c_void_p]),
    "srl_col2im_ndhwc": (c_int, [c_void_p, c_void_p, c_int64] + [c_int] * 8 + [c_void_p, c_int, c_void_p]),
    "srl_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_int, c_int,
                              c_float, c_float, c_void_p, c_void_p]),
    "srl_rmsprop_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                                  c_float, c_float, c_int, c_float, c_float, c_void_p, c_void_p]),
    "srl_step_plan_create": (c_int, [POINTER(c_void_p), c_void_p]),
    "srl_step_plan_add_input": (c_int, [c_void_p, c_void_p, c_int64]),
    "srl_step_plan_add_output": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "srl_step_plan_run": (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int, POINTER(c_void_p), c_int, c_int]),
    "srl_step_plan_destroy": (c_int, [c_void_p]),
    "srl_h2_pack_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_h2_pack_rows_colsum_workspace": (c_int64, [c_int64, c_int32]),
    "srl_h2_pack_rows_colsum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32]),
    "srl_h2_unpack_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64]),
    "srl_h2_pack_image": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_h2_unpack_image": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "srl_h2_weights": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "srl_h2_conv": (c_int, [c_void_p, c_int32, POINTER(H2ConvArgs)]),
    "srl_h2_wgrad_workspace": (c_int64, [c_int32]),
    "srl_h2_wgrad": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "srl_h2_wgrad_dense_workspace": (c_int64, [c_int64, c_int32, c_int32]),
    "srl_h2_wgrad_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int64, c_int64, c_void_p, c_void_p,
                                   c_int32]),
    "srl_h2_gemm": (c_int, [c_void_p, POINTER(H2GemmDesc)]),
    "srl_h2_gemm_splitk": (c_int, [c_void_p, POINTER(H2GemmDesc), c_int32, c_int32]),
    "srl_conv2d_obs_fwd_h2": (c_int, [c_void_p, POINTER(ConvDesc)] + [c_void_p] * 13 + [c_int, c_int, c_void_p]),
    "srl_conv2d_obs_fold_h2": (c_int, [c_void_p, POINTER(ConvDesc)] + [c_void_p] * 5),
    "srl_comm_available": (c_int, []),
    "srl_comm_unique_id": (c_int, [c_void_p]),
    "srl_comm_init": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_int]),
    "srl_comm_world": (c_int, [c_void_p, POINTER(c_int)]),
    "srl_comm_destroy": (c_int, [c_void_p]),
    "srl_allreduce_stats_f64x3": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "srl_allreduce_grads": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "srl_broadcast_params": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path() -> str:
    return _LIB_PATH


def lib():
    """Load (once) and return the ctypes library; raises ``HipError`` if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipError(f"{_LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                           "there is no CPU fallback for the hot path")
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        if handle.srl_abi_version() != ABI_VERSION:
            raise HipError(f"ABI mismatch: library {handle.srl_abi_version()} vs binding {ABI_VERSION}")
        _lib = handle
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise HipError(f"{what} failed ({rc}): {lib().srl_last_error().decode(errors='replace')}")


def require_gpu():
    if not torch.cuda.is_available():
        raise HipError("no MI355X / ROCm device visible: the hot path has no CPU fallback")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """The current HIP stream of the current device, as the integer the C ABI takes.  (Through torch's C entry points: the Python
    route -- torch.cuda.current_stream().cuda_stream -- costs ~8 us, eleven times per rollout call and ~550 times per update.)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError(f"{name}: expected a device tensor, got {t.device}")
    if dtype is not None and t.dtype != dtype:
        raise HipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HipError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def _i32_array(vals: Sequence[int]):
    return (c_int32 * len(vals))(*[int(v) for v in vals])


class KernelProfile:
    """Per-kernel device timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py in an untimed pass; off (``None``) otherwise."""

    def __init__(self):
        # (name, start_event, end_event, work, executed): work = algorithmic flops or bytes; executed = matrix-core flops
        # issued for it (algorithmic x piece products of the kernel that took the call)
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, a, b, work, executed in self.records:
            e = out.setdefault(name, dict(calls=0, ms=0.0, work=0.0, executed=0.0))
            e["calls"] += 1
            e["ms"] += a.elapsed_time(b)
            e["work"] += work
            e["executed"] += executed
        return out


_prof: Optional[KernelProfile] = None


def set_profile(p: Optional[KernelProfile]):
    global _prof
    _prof = p


def piece_products(kind: str = "x3") -> float:
    """Matrix-core products issued per algorithmic multiply-add by the kernel family that takes a call: 6 for float32
    operands as three bf16 pieces each (``x3``), 3 for two f16 pieces each (``2h``) and for the byte-operand first layer
    (``obs``; ``obs2``: 2, the block kernel with two f16 weight pieces), 1 for the float32 MFMA kernels (SRL_MFMA=f32 / SRL_OBS_BF16=0)."""
    if kind == "f32":  # plain float32 FMA kernels (csrc/mlp_small.hip)
        return 1.0
    if kind == "obs":
        return 1.0 if os.environ.get("SRL_OBS_BF16", "")[:1] == "0" else 3.0
    if kind == "obs2":  # csrc/obs_h2.h: byte operand x two f16 weight pieces
        return 2.0 if os.environ.get("SRL_OBS_H2BLOCK", "")[:1] != "0" else 3.0
    if os.environ.get("SRL_MFMA", "")[:1] == "f":
        return 1.0
    return 3.0 if kind == "2h" else 6.0


def f16x2_enabled() -> bool:
    return os.environ.get("SRL_F16X2", "")[:1] != "0" and os.environ.get("SRL_MFMA", "")[:1] != "f"


class _scope:

    def __init__(self, name, work=0.0, kind="x3"):
        if _prof is None:   # (the common case: no environment look-ups, nothing kept, on ~550 launches per update)
            return
        self.name, self.work, self.executed = name, work, work * piece_products(kind) if work else 0.0

    def __enter__(self):
        if _prof is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _prof is not None:
            self.b.record()
            _prof.records.append((self.name, self.a, self.b, self.work, self.executed))
        return False


# ------------------------------------------------------------------------------------------------
def device_info():
    n, l = c_int(0), c_int(0)
    buf = ctypes.create_string_buffer(128)
    _check(lib().srl_device_info(ctypes.byref(n), ctypes.byref(l), buf, 128), "srl_device_info")
    return dict(num_cus=n.value, lds_bytes_per_cu=l.value, arch=buf.value.decode())


DISPATCH_FAMILIES = ("gemm3", "gemm_f32", "skinny", "obs_fwd_bf16", "obs_bwd_bf16", "gemm2h", "h2")


def dispatch_counts(reset: bool = False) -> dict:
    """Launches per kernel family since the last reset (``srl_dispatch_counts``): which kernels a step really ran on."""
    buf = (c_int64 * 8)()
    _check(lib().srl_dispatch_counts(buf, 8, int(bool(reset))), "srl_dispatch_counts")
    return {name: int(buf[i]) for i, name in enumerate(DISPATCH_FAMILIES)}


def dispatch_tiles(reset: bool = False) -> dict:
    """Launches per kernel INSTANTIATION since the last reset (``srl_dispatch_tiles``), keyed ``family:p0xp1:k<p2>[:flags]`` --
    e.g. ``gemm2h:64x256:k8:f7`` (tile, split-K factor, operand modes), ``h2:conv:0:s3`` (kind, ring depth), ``h2:gemm:4:s3``, ``h2:tn:8:s4`` (the dense weight gradient)."""
    keys, counts = (ctypes.c_uint64 * 256)(), (c_int64 * 256)()
    n = lib().srl_dispatch_tiles(keys, counts, 256, int(bool(reset)))
    if n < 0:
        raise HipError(f"srl_dispatch_tiles: {lib().srl_last_error().decode(errors='replace')}")
    out = {}
    for i in range(min(n, 256)):
        k = int(keys[i])
        fam, p0, p1, p2, fl = k >> 56, (k >> 40) & 0xffff, (k >> 24) & 0xffff, (k >> 8) & 0xffff, k & 0xff
        name = DISPATCH_FAMILIES[fam] if fam < len(DISPATCH_FAMILIES) else f"family{fam}"
        if name == "h2":
            label = f"h2:{ {1: 'conv', 2: 'wgrad', 3: 'gemm', 4: 'tn', 5: 'gemmp'}.get(p0, p0)}:{p1}:s{p2}"
        elif name.startswith("obs_"):
            label = f"{name}:k{p0}:{('f32', 'h2', 'h2blk')[min(p1, 2)]}:split{p2}"  # h2blk: obs_h2.h, p2 = its persistent workgroups
        else:
            label = f"{name}:{p0}x{p1}:k{p2}:f{fl}"
        out[label] = out.get(label, 0) + int(counts[i])
    return out


def gae_scan_workspace(B, Nc, device) -> torch.Tensor:
    """Zeroed workspace for ``gae_scan(..., workspace=)``: statistics without a memset launch or float64 atomics."""
    n = int(lib().srl_gae_scan_workspace_bytes(int(B), int(Nc)))
    if n < 0:
        raise HipError("srl_gae_scan_workspace_bytes: invalid argument")
    return torch.zeros((n + 7) // 8, dtype=torch.int64, device=device)


def gae_scan(reward, value, done, truncated, on_reset, gamma, lmbda, adv, ret, stats=None, imp_ratio=None, rho=1.0,
             c=1.0, workspace=None):
    """reward [>=T,B,Nc] f32, value/done/truncated/on_reset [T+1,B,*]; writes adv/ret rows [0,T); stats f64[3].
    ``gamma`` / ``lmbda``: python floats or float32 device tensors [T, B, 1] (reference gae.py:51-60)."""
    Tp1, B = on_reset.shape[0], on_reset.shape[1]
    T = Tp1 - 1
    Nc = value.shape[2] if value.dim() > 2 else 1
    assert value.shape[0] == Tp1 and reward.shape[0] >= T and adv.shape[0] >= T and ret.shape[0] >= T
    gamma_t = lambda_t = None
    if isinstance(gamma, torch.Tensor):
        if tuple(gamma.shape) != (T, B, 1):
            raise HipError(f"gamma tensor: expected shape {(T, B, 1)}, got {tuple(gamma.shape)}")  # gae.py:53
        gamma_t, gamma = gamma, 0.0
    if isinstance(lmbda, torch.Tensor):
        if tuple(lmbda.shape) != (T, B, 1):
            raise HipError(f"lmbda tensor: expected shape {(T, B, 1)}, got {tuple(lmbda.shape)}")  # gae.py:58
        lambda_t, lmbda = lmbda, 0.0
    if workspace is not None and workspace.numel() * workspace.element_size() < lib().srl_gae_scan_workspace_bytes(B, Nc):
        raise HipError("gae_scan: workspace too small for this batch")
    # algorithmic bytes: r,v f32 + 3 flag bytes in, adv,ret f32 out = 19 B per env-step (+ bootstrap row)
    with _scope("gae_scan", 19.0 * T * B * Nc + 7.0 * B * Nc):
      _check(
        lib().srl_gae_scan(_stream(), _ptr(reward, torch.float32, "reward"), _ptr(value, torch.float32, "value"),
                           _ptr(done, torch.uint8, "done"), _ptr(truncated, torch.uint8, "truncated"),
                           _ptr(on_reset, torch.uint8, "on_reset"), _ptr(imp_ratio, torch.float32, "imp_ratio"),
                           _ptr(gamma_t, torch.float32, "gamma"), _ptr(lambda_t, torch.float32, "lmbda"), T, B,
                           Nc, float(gamma), float(lmbda), float(rho), float(c), _ptr(adv, torch.float32, "adv"),
                           _ptr(ret, torch.float32, "ret"), _ptr(stats, torch.float64, "stats"),
                           _ptr(workspace, None, "workspace")), "srl_gae_scan")


def masked_stats(x, mask, stats, mask_invert=False):
    _check(
        lib().srl_masked_stats(_stream(), _ptr(x, torch.float32, "x"), _ptr(mask, torch.uint8, "mask"),
                               int(mask_invert), x.numel(), _ptr(stats, torch.float64, "stats")), "srl_masked_stats")


def masked_normalize(x, mask, stats, out, mask_invert=False, eps=1e-5, unbiased=False):
    _check(
        lib().srl_masked_normalize(_stream(), _ptr(x, torch.float32, "x"), _ptr(mask, torch.uint8, "mask"),
                                   int(mask_invert), x.numel(), _ptr(stats, torch.float64, "stats"), float(eps),
                                   int(unbiased), _ptr(out, torch.float32, "out")), "srl_masked_normalize")


def gaussian_fwd(mean, log_std_ptr, ld_ls, action, logp, ent):
    n, A = mean.shape
    _check(lib().srl_gaussian_fwd(_stream(), _ptr(mean, torch.float32, "mean"), mean.stride(0), log_std_ptr, int(ld_ls),
                                  _ptr(action, torch.float32, "action"), n, A, _ptr(logp, torch.float32, "logp"),
                                  _ptr(ent, torch.float32, "ent")), "srl_gaussian_fwd")


def gaussian_bwd(mean, log_std_ptr, ld_ls, action, d_logp, d_ent, d_mean, d_log_std):
    n, A = mean.shape
    f = torch.float32
    _check(lib().srl_gaussian_bwd(_stream(), _ptr(mean, f, "mean"), mean.stride(0), log_std_ptr, int(ld_ls),
                                  _ptr(action, f, "action"), n, A, _ptr(d_logp, f, "d_logp"), _ptr(d_ent, f, "d_ent"),
                                  _ptr(d_mean, f, "d_mean"), _ptr(d_log_std, f, "d_log_std")), "srl_gaussian_bwd")


def gaussian_sample(mean, log_std_ptr, ld_ls, is_eval, seed, offset, action, logp, row0=0):
    n, A = mean.shape
    _check(lib().srl_gaussian_sample(_stream(), _ptr(mean, torch.float32, "mean"), mean.stride(0), log_std_ptr, int(ld_ls),
                                     _ptr(is_eval, torch.uint8, "is_eval"), n, A, int(seed) & (2**64 - 1),
                                     int(offset) & (2**64 - 1), _ptr(action, torch.float32, "action"),
                                     _ptr(logp, torch.float32, "logp"), int(row0)), "srl_gaussian_sample")


def rnn_seq_supported(kind: str, H: int) -> bool:
    """Whether the time loop of a chunk runs inside one launch for this cell and width (``srl_rnn_seq_supported``; SRL_RNN_SEQ=0
    switches it off for an A/B)."""
    return os.environ.get("SRL_RNN_SEQ", "1")[:1] != "0" and bool(lib().srl_rnn_seq_supported(1 if kind == "lstm" else 0, int(H)))


def lstm_seq_fwd(pre_ptr, w_hh, b_hh, hin_ptr, cin_ptr, reset_ptr, N, H, C, y_ptr, cnew_ptr):
    _check(lib().srl_lstm_seq_fwd(_stream(), pre_ptr, w_hh, b_hh, hin_ptr, cin_ptr, reset_ptr, int(N), int(H), int(C), y_ptr, cnew_ptr),
           "srl_lstm_seq_fwd")


def lstm_seq_bwd(dy_ptr, ld_dy, gates_ptr, w_hh, cin_ptr, cnew_ptr, reset_ptr, N, H, C):
    _check(lib().srl_lstm_seq_bwd(_stream(), dy_ptr, int(ld_dy), gates_ptr, w_hh, cin_ptr, cnew_ptr, reset_ptr, int(N), int(H), int(C)),
           "srl_lstm_seq_bwd")


def gru_seq_fwd(gi_ptr, gh_ptr, w_hh, b_hh, hin_ptr, reset_ptr, N, H, C, y_ptr):
    _check(lib().srl_gru_seq_fwd(_stream(), gi_ptr, gh_ptr, w_hh, b_hh, hin_ptr, reset_ptr, int(N), int(H), int(C), y_ptr), "srl_gru_seq_fwd")


def gru_seq_bwd(dy_ptr, ld_dy, gates_ptr, gh_ptr, w_hh, hin_ptr, reset_ptr, N, H, C):
    _check(lib().srl_gru_seq_bwd(_stream(), dy_ptr, int(ld_dy), gates_ptr, gh_ptr, w_hh, hin_ptr, reset_ptr, int(N), int(H), int(C)),
           "srl_gru_seq_bwd")


def gru_mask_state(h_ptr, reset_ptr, N, H, out_ptr):
    _check(lib().srl_gru_mask_state(_stream(), h_ptr, reset_ptr, int(N), int(H), out_ptr), "srl_gru_mask_state")


def gru_cell_fwd(gi_ptr, gh_ptr, hin_ptr, reset_next_ptr, N, H, y_ptr, hin_next_ptr):
    _check(lib().srl_gru_cell_fwd(_stream(), gi_ptr, gh_ptr, hin_ptr, reset_next_ptr, int(N), int(H), y_ptr, hin_next_ptr),
           "srl_gru_cell_fwd")


def gru_cell_bwd(dy_ptr, carry_ptr, reset_next_ptr, gates_ptr, gh_ptr, hin_ptr, N, H, dh_direct_ptr):
    _check(lib().srl_gru_cell_bwd(_stream(), dy_ptr, carry_ptr, reset_next_ptr, gates_ptr, gh_ptr, hin_ptr, int(N), int(H),
                                  dh_direct_ptr), "srl_gru_cell_bwd")


def lstm_cell_fwd(pre_ptr, cin_ptr, reset_next_ptr, N, H, y_ptr, cnew_ptr, hin_next_ptr, cin_next_ptr):
    _check(lib().srl_lstm_cell_fwd(_stream(), pre_ptr, cin_ptr, reset_next_ptr, int(N), int(H), y_ptr, cnew_ptr, hin_next_ptr,
                                   cin_next_ptr), "srl_lstm_cell_fwd")


def lstm_cell_bwd(dy_ptr, carry_h_ptr, carry_c_ptr, reset_next_ptr, gates_ptr, cin_ptr, cnew_ptr, N, H, dc_in_ptr):
    _check(lib().srl_lstm_cell_bwd(_stream(), dy_ptr, carry_h_ptr, carry_c_ptr, reset_next_ptr, gates_ptr, cin_ptr, cnew_ptr,
                                   int(N), int(H), dc_in_ptr), "srl_lstm_cell_bwd")


def chunk_rows(src_ptr, dst_ptr, T, B, C, D, inverse=False):
    _check(lib().srl_chunk_rows(_stream(), src_ptr, dst_ptr, int(T), int(B), int(C), int(D), int(inverse)), "srl_chunk_rows")


def masked_stats_cols(x, mask, stats, vd, mask_invert=False):
    """x float32 [n, vd] -> stats float64 [vd, 3] = per column {sum mask, sum x*mask, sum (x*mask)^2}."""
    _check(
        lib().srl_masked_stats_cols(_stream(), _ptr(x, torch.float32, "x"), _ptr(mask, torch.uint8, "mask"),
                                    int(mask_invert), x.numel() // vd, int(vd), _ptr(stats, torch.float64, "stats")),
        "srl_masked_stats_cols")


def fold_col_stats(col_stats, vd, stats):
    _check(lib().srl_fold_col_stats(_stream(), _ptr(col_stats, torch.float64, "col_stats"), int(vd),
                                    _ptr(stats, torch.float64, "stats")), "srl_fold_col_stats")


def importance_ratio(new_lp, old_lp, out):
    _check(lib().srl_importance_ratio(_stream(), _ptr(new_lp, torch.float32, "new_lp"), _ptr(old_lp, torch.float32, "old_lp"),
                                      new_lp.numel(), _ptr(out, torch.float32, "ratio")), "srl_importance_ratio")


def popart_update(stats, rms, vd, beta, eps, w_ptr=None, b_ptr=None, in_features=0, rescale=False):
    _check(
        lib().srl_popart_update(_stream(), _ptr(stats, torch.float64, "stats"), float(beta), float(eps), int(vd),
                                _ptr(rms, torch.float64, "rms"), w_ptr, b_ptr, int(in_features), int(rescale)),
        "srl_popart_update")


def popart_map(x, rms, vd, out, normalize, eps):
    _check(
        lib().srl_popart_map(_stream(), _ptr(x, torch.float32, "x"), x.numel() // vd, int(vd),
                             _ptr(rms, torch.float64, "rms"), float(eps), int(normalize), _ptr(out, torch.float32, "out")),
        "srl_popart_map")


def ppo_loss_fwd_bwd(new_lp, old_lp, value, old_value, adv, ret, entropy, mask, hp: PpoHparams, norm_stats, local_n,
                     d_new_lp, d_value, d_entropy, loss_terms, done=None, truncated=None, value_dim=1):
    """value / old_value / adv / ret / d_value hold ``value_dim`` channels per row of new_lp."""
    f = torch.float32
    n = new_lp.numel()
    if value.numel() != n * value_dim or adv.numel() != n * value_dim or d_value.numel() != n * value_dim:
        raise HipError("ppo_loss_fwd_bwd: value tensors must hold value_dim channels per row")
    _check(
        lib().srl_ppo_loss_fwd_bwd(_stream(), _ptr(new_lp, f, "new_lp"), _ptr(old_lp, f, "old_lp"),
                                   _ptr(value, f, "value"), _ptr(old_value, f, "old_value"), _ptr(adv, f, "adv"),
                                   _ptr(ret, f, "ret"), _ptr(entropy, f, "entropy"), _ptr(mask, torch.uint8, "mask"),
                                   n, int(value_dim), ctypes.byref(hp), _ptr(norm_stats, torch.float64, "norm_stats"),
                                   _ptr(local_n, torch.float64, "local_n"), _ptr(done, torch.uint8, "done"),
                                   _ptr(truncated, torch.uint8, "truncated"), _ptr(d_new_lp, f, "d_new_lp"),
                                   _ptr(d_value, f, "d_value"), _ptr(d_entropy, f, "d_entropy"),
                                   _ptr(loss_terms, torch.float64, "loss_terms")), "srl_ppo_loss_fwd_bwd")


def categorical_fwd(logits, action, avail, head_dims, logp, entropy):
    n = logits.shape[0]
    _check(
        lib().srl_categorical_fwd(_stream(), _ptr(logits, torch.float32, "logits"), logits.shape[1],
                                  _ptr(action, torch.int32, "action"), _ptr(avail, torch.uint8, "avail"), n,
                                  len(head_dims), _i32_array(head_dims), _ptr(logp, torch.float32, "logp"),
                                  _ptr(entropy, torch.float32, "entropy")), "srl_categorical_fwd")


def categorical_bwd(logits, action, avail, head_dims, d_logp, d_entropy, d_logits):
    n = logits.shape[0]
    _check(
        lib().srl_categorical_bwd(_stream(), _ptr(logits, torch.float32, "logits"), logits.shape[1],
                                  _ptr(action, torch.int32, "action"), _ptr(avail, torch.uint8, "avail"), n,
                                  len(head_dims), _i32_array(head_dims), _ptr(d_logp, torch.float32, "d_logp"),
                                  _ptr(d_entropy, torch.float32, "d_entropy"),
                                  _ptr(d_logits, torch.float32, "d_logits"), d_logits.shape[1]), "srl_categorical_bwd")


def categorical_sample(logits, avail, is_eval, head_dims, seed, offset, action_out, logp, row0=0):
    n = logits.shape[0]
    _check(
        lib().srl_categorical_sample(_stream(), _ptr(logits, torch.float32, "logits"), logits.shape[1],
                                     _ptr(avail, torch.uint8, "avail"), _ptr(is_eval, torch.uint8, "is_eval"), n,
                                     len(head_dims), _i32_array(head_dims), int(seed), int(offset),
                                     _ptr(action_out, torch.int64, "action_out"), _ptr(logp, torch.float32, "logp"), int(row0)),
        "srl_categorical_sample")


def categorical_log_softmax(logits, avail, head_dims, out):
    """out[i, head] = masked logits - logsumexp: ``Categorical(logits=...).logits`` per action head (srl_hip.h, PPG)."""
    n = logits.shape[0]
    _check(
        lib().srl_categorical_log_softmax(_stream(), _ptr(logits, torch.float32, "logits"), logits.shape[1],
                                          _ptr(avail, torch.uint8, "avail"), n, len(head_dims), _i32_array(head_dims),
                                          _ptr(out, torch.float32, "out"), out.shape[1]), "srl_categorical_log_softmax")


def ppg_aux_loss_fwd_bwd(logq_old, logits, avail, head_dims, aux_value, pred_value, target, done, undone_count, beta_clone,
                         value_head_weight, d_logits, d_aux, d_pred, terms):
    """The auxiliary phase's joint loss and its gradient with respect to (raw logits, auxiliary value, critic value)
    (phasic_policy_gradient.py:262-280; srl_hip.h).  ``terms`` float64[3]: auxiliary value loss, value head loss, policy distance."""
    n = logits.shape[0]
    vd = aux_value.shape[1] if aux_value.dim() > 1 else 1
    _check(
        lib().srl_ppg_aux_loss_fwd_bwd(_stream(), _ptr(logq_old, torch.float32, "logq_old"), logq_old.shape[1],
                                       _ptr(logits, torch.float32, "logits"), logits.shape[1], _ptr(avail, torch.uint8, "avail"), n,
                                       len(head_dims), _i32_array(head_dims), _ptr(aux_value, torch.float32, "aux_value"),
                                       _ptr(pred_value, torch.float32, "pred_value"), _ptr(target, torch.float32, "target"), vd,
                                       _ptr(done, torch.uint8, "done"), _ptr(undone_count, torch.float64, "undone_count"),
                                       float(beta_clone), float(value_head_weight), _ptr(d_logits, torch.float32, "d_logits"),
                                       d_logits.shape[1], _ptr(d_aux, torch.float32, "d_aux"), _ptr(d_pred, torch.float32, "d_pred"),
                                       _ptr(terms, torch.float64, "terms")), "srl_ppg_aux_loss_fwd_bwd")


def gemm_two_piece(M, N, K, A, lda, B, ldb, a_absmax, b_absmax) -> bool:
    """Whether ``gemm`` sends this product to the two-plane f16 kernel (mirrors gemm.hip): only then may B be handed over
    pre-split (``presplit``)."""
    small = M * N <= 65536 and 4 <= K <= 512 and os.environ.get("SRL_SMALL_GEMM", "1")[:1] != "0"  # 64 x 64 tiles, three bf16 pieces
    return (a_absmax is not None and b_absmax is not None and M > 64 and N > 64 and K >= 64 and A % 16 == 0 and B % 16 == 0 and
            lda % 4 == 0 and ldb % 4 == 0 and f16x2_enabled() and not small and os.environ.get("SRL_MFMA", "")[:1] != "f")


def presplit(src_ptr, absmax_ptr, dst_ptr, n):
    """dst = the two f16 pieces of src (float32, n a multiple of 4) under the scale of *absmax, 16 bytes per 4 elements: a B
    operand split once instead of in every tile that stages it (``gemm(..., b_presplit=True)``, ``conv2d_nhwc_fwd /
    dgrad(..., presplit=True)``)."""
    _check(lib().srl_presplit(_stream(), src_ptr, absmax_ptr, dst_ptr, n), "srl_presplit")


def gemm(M, N, K, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, bias=None, act=ACT_NONE, dact_src=None, ld_dact=0,
         dact=ACT_NONE, accumulate=False, split_k=1, workspace=None, a_colsum=None, a_absmax=None, b_absmax=None,
         out_absmax=None, mask_out=None, dact_mask=None, b_presplit=False, b_h2_scale=None):
    """Raw-pointer GEMM (``A``/``B``/``C``/... are ints from ``data_ptr()`` possibly with byte offsets).
    ``a_colsum``: [M] += sum_k A(i, k) as a by-product (k-major, float4-stageable A; see ``gemm_colsum_ok``).
    ``a_absmax`` / ``b_absmax``: device floats bounding max |A|, max |B| (both given: the two-plane f16 forward kernel);
    ``out_absmax``: device float folded with max |C|.  ``mask_out`` / ``dact_mask``: sign-bit masks of a ReLU output
    (written by the forward product; read by a data gradient instead of ``dact_src``'s floats: srl_hip.h).
    ``b_presplit``: B is ``presplit``'s output (only where ``gemm_two_piece`` says so)."""
    d = GemmDesc(M, N, K, A, lda, int(a_kmajor), B, ldb, int(b_kmajor), C, ldc, bias, int(act), dact_src, ld_dact,
                 int(dact), int(accumulate), int(split_k), workspace, a_colsum, a_absmax, b_absmax, out_absmax,
                 mask_out, dact_mask, int(bool(b_presplit)), b_h2_scale)
    two = gemm_two_piece(M, N, K, A, lda, B, ldb, a_absmax, b_absmax)
    with _scope("gemm", 2.0 * M * N * K, "2h" if two else "x3"):
        _check(lib().srl_gemm(_stream(), ctypes.byref(d)), "srl_gemm")


def gemm_colsum_ok(M, N, K, A, lda, B, ldb, b_kmajor) -> bool:
    """Whether ``gemm(..., a_kmajor=1, a_colsum=...)`` is accepted: both operands float4-stageable (srl_hip.h)."""
    b_contig = N if b_kmajor else K
    if M <= 16 and K >= 256 and b_kmajor and N % 4 == 0 and ldb % 4 == 0 and B % 16 == 0:
        return True  # the narrow-head kernel (csrc/skinny.h) sums A's columns whatever M's alignment
    return A % 16 == 0 and B % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0 and M % 4 == 0 and b_contig % 4 == 0


def layernorm_fwd(x_ptr, ldx, gamma_ptr, beta_ptr, rows, D, y_ptr, ldy, mean_ptr, rstd_ptr):
    _check(lib().srl_layernorm_fwd(_stream(), x_ptr, ldx, gamma_ptr, beta_ptr, rows, D, y_ptr, ldy, mean_ptr, rstd_ptr),
           "srl_layernorm_fwd")


def layernorm_bwd(dy_ptr, lddy, x_ptr, ldx, gamma_ptr, mean_ptr, rstd_ptr, rows, D, dx_ptr, lddx, dact, dgamma_ptr,
                  dbeta_ptr, dx_absmax=None):
    _check(
        lib().srl_layernorm_bwd(_stream(), dy_ptr, lddy, x_ptr, ldx, gamma_ptr, mean_ptr, rstd_ptr, rows, D, dx_ptr,
                                lddx, int(dact), dgamma_ptr, dbeta_ptr, dx_absmax), "srl_layernorm_bwd")


def _ptr_array(ptrs):
    return (c_void_p * len(ptrs))(*[p if p else None for p in ptrs])


def ln_heads_supported(D: int, head_dims) -> bool:
    """Whether LayerNorm(D) + these heads run as one launch per direction (csrc/ln_heads.hip)."""
    return bool(lib().srl_ln_heads_supported(int(D), len(head_dims), _i32_array(head_dims)))


def ln_heads_fwd(x_ptr, ldx, n, D, gamma_ptr, beta_ptr, w_ptrs, b_ptrs, head_dims, y_ptrs, ldys, mean_ptr, rstd_ptr, x_slabs=1,
                 x_slab_stride=0, x_bias=None, x_act=0, x_out=None, ldxo=0):
    """LayerNorm over D + the heads reading it: y[h] = LN(x) W[h]^T + b[h]; the normalised features are not stored.  ``x_slabs``
    > 1: x is the raw output of ``h2_gemm_splitk`` and is finished (slabs added, bias, activation) while it is read."""
    flops = 2.0 * n * D * sum(head_dims)
    with _scope("ln_heads_fwd", flops, "f32"):
        _check(lib().srl_ln_heads_fwd(_stream(), x_ptr, ldx, n, int(D), gamma_ptr, beta_ptr, len(head_dims), _ptr_array(w_ptrs), _ptr_array(b_ptrs),
                                      _i32_array(head_dims), _ptr_array(y_ptrs), (c_int64 * len(ldys))(*ldys), mean_ptr, rstd_ptr,
                                      int(x_slabs), int(x_slab_stride), x_bias, int(x_act), x_out, int(ldxo)), "srl_ln_heads_fwd")


def ln_heads_bwd(x_ptr, ldx, n, D, gamma_ptr, beta_ptr, mean_ptr, rstd_ptr, w_ptrs, head_dims, dy_ptrs, lddys, in_act, dx_ptr, lddx,
                 dgamma_ptr, dbeta_ptr, dw_ptrs, db_ptrs, dx_absmax=None):
    flops = 4.0 * n * D * sum(head_dims)
    with _scope("ln_heads_bwd", flops, "f32"):
        _check(lib().srl_ln_heads_bwd(_stream(), x_ptr, ldx, n, int(D), gamma_ptr, beta_ptr, mean_ptr, rstd_ptr, len(head_dims),
                                      _ptr_array(w_ptrs), _i32_array(head_dims), _ptr_array(dy_ptrs), (c_int64 * len(lddys))(*lddys), int(in_act),
                                      dx_ptr, lddx, dgamma_ptr, dbeta_ptr, _ptr_array(dw_ptrs), _ptr_array(db_ptrs), dx_absmax), "srl_ln_heads_bwd")


def obs_ln_stats(obs_ptr, is_u8, n, D, mean_ptr, rstd_ptr):
    _check(lib().srl_obs_ln_stats(_stream(), obs_ptr, int(is_u8), n, D, mean_ptr, rstd_ptr), "srl_obs_ln_stats")


def im2col_obs_ln(obs_ptr, is_u8, mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, n, C, H, W, KH, KW, stride, P_ptr):
    _check(
        lib().srl_im2col_obs_ln(_stream(), obs_ptr, int(is_u8), mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, n, C, H, W, KH,
                                KW, stride, P_ptr), "srl_im2col_obs_ln")


def im2col_nhwc(x_ptr, n, H, W, C, KH, KW, stride, P_ptr):
    _check(lib().srl_im2col_nhwc(_stream(), x_ptr, n, H, W, C, KH, KW, stride, P_ptr), "srl_im2col_nhwc")


def col2im_nhwc(dP_ptr, n, H, W, C, KH, KW, stride, y_ptr, dact, dX_ptr):
    _check(lib().srl_col2im_nhwc(_stream(), dP_ptr, n, H, W, C, KH, KW, stride, y_ptr, int(dact), dX_ptr),
           "srl_col2im_nhwc")


def obs_ln_affine_bwd(dP_ptr, obs_ptr, is_u8, mean_ptr, rstd_ptr, n, C, H, W, KH, KW, stride, dgamma_ptr, dbeta_ptr):
    _check(
        lib().srl_obs_ln_affine_bwd(_stream(), dP_ptr, obs_ptr, int(is_u8), mean_ptr, rstd_ptr, n, C, H, W, KH, KW,
                                    stride, dgamma_ptr, dbeta_ptr), "srl_obs_ln_affine_bwd")


def colsum(x_ptr, ld, rows, cols, out_ptr, accumulate=True):
    _check(lib().srl_colsum(_stream(), x_ptr, ld, rows, cols, out_ptr, int(accumulate)), "srl_colsum")


def copy2d(src_ptr, lds, dst_ptr, ldd, rows, cols):
    _check(lib().srl_copy2d(_stream(), src_ptr, lds, dst_ptr, ldd, rows, cols), "srl_copy2d")


def u8_to_f32(src, dst):
    _check(lib().srl_u8_to_f32(_stream(), _ptr(src, torch.uint8, "src"), _ptr(dst, torch.float32, "dst"), src.numel()),
           "srl_u8_to_f32")


ACCUMULATE_MAX = 7


def accumulate_n(dst, srcs):
    """dst += srcs[0] + srcs[1] + ... (``srl_accumulate_n``: one launch, added left to right), float32 tensors of equal size."""
    srcs = list(srcs)
    while srcs:
        part, srcs = srcs[:ACCUMULATE_MAX], srcs[ACCUMULATE_MAX:]
        arr = (c_void_p * len(part))(*[_ptr(t, torch.float32, "src") for t in part])
        _check(lib().srl_accumulate_n(_stream(), _ptr(dst, torch.float32, "dst"), arr, len(part), dst.numel()), "srl_accumulate_n")


def accumulate(dst, src):
    """dst += src (``srl_accumulate``), float32 tensors of equal size."""
    _check(lib().srl_accumulate(_stream(), _ptr(dst, torch.float32, "dst"), _ptr(src, torch.float32, "src"), dst.numel()),
           "srl_accumulate")


def grad_sumsq(g, sumsq):
    _check(lib().srl_grad_sumsq(_stream(), _ptr(g, torch.float32, "g"), g.numel(), _ptr(sumsq, torch.float64, "sumsq")),
           "srl_grad_sumsq")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, adamw, step, grad_scale=1.0, max_norm=-1.0, sumsq=None,
              grad_norm_out=None, step_scalars=None):
    f = torch.float32
    _check(
        lib().srl_adam_step(_stream(), _ptr(p, f, "p"), _ptr(g, f, "g"), _ptr(m, f, "m"), _ptr(v, f, "v"), p.numel(),
                            float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(adamw),
                            int(step), float(grad_scale), float(max_norm), _ptr(sumsq, torch.float64, "sumsq"),
                            _ptr(grad_norm_out, f, "grad_norm_out"), _ptr(step_scalars, f, "step_scalars")),
        "srl_adam_step")


def obs_ln_nhwc(obs_ptr, is_u8, mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, n, C, H, W, y_ptr):
    _check(lib().srl_obs_ln_nhwc(_stream(), obs_ptr, int(is_u8), mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, n, C, H, W, y_ptr),
           "srl_obs_ln_nhwc")


def obs_ln_nhwc_bwd(dy_ptr, obs_ptr, is_u8, mean_ptr, rstd_ptr, n, C, H, W, dgamma_ptr, dbeta_ptr):
    _check(lib().srl_obs_ln_nhwc_bwd(_stream(), dy_ptr, obs_ptr, int(is_u8), mean_ptr, rstd_ptr, n, C, H, W, dgamma_ptr,
                                     dbeta_ptr), "srl_obs_ln_nhwc_bwd")


def pad_nhwc(x_ptr, n, H, W, C, pad, y_ptr):
    _check(lib().srl_pad_nhwc(_stream(), x_ptr, n, H, W, C, pad, y_ptr), "srl_pad_nhwc")


def crop_nhwc(yp_ptr, n, H, W, C, pad, x_ptr):
    _check(lib().srl_crop_nhwc(_stream(), yp_ptr, n, H, W, C, pad, x_ptr), "srl_crop_nhwc")


def maxpool2_nhwc_fwd(x_ptr, n, H, W, C, y_ptr):
    _check(lib().srl_maxpool2_nhwc_fwd(_stream(), x_ptr, n, H, W, C, y_ptr), "srl_maxpool2_nhwc_fwd")


def maxpool2_nhwc_bwd(dy_ptr, x_ptr, n, H, W, C, dact, dx_ptr):
    _check(lib().srl_maxpool2_nhwc_bwd(_stream(), dy_ptr, x_ptr, n, H, W, C, int(dact), dx_ptr), "srl_maxpool2_nhwc_bwd")


PAD_MODES = {"zeros": 0, "reflect": 1, "replicate": 2, "circular": 3}  # nn.ConvNd's padding_mode


def pad_ndhwc(x_ptr, n, vol, pads, y_ptr, mode=0):
    """vol = (D, H, W, C), pads = (pd, ph, pw), mode = PAD_MODES[padding_mode]."""
    _check(lib().srl_pad_ndhwc(_stream(), x_ptr, n, *vol, *pads, int(mode), y_ptr), "srl_pad_ndhwc")


def crop_ndhwc(yp_ptr, n, vol, pads, x_ptr, mode=0):
    """The adjoint of ``pad_ndhwc``: the crop for zero padding, border voxels summed back into their sources otherwise."""
    _check(lib().srl_crop_ndhwc(_stream(), yp_ptr, n, *vol, *pads, int(mode), x_ptr), "srl_crop_ndhwc")


def maxpool_ndhwc_fwd(x_ptr, n, vol, win, y_ptr):
    _check(lib().srl_maxpool_ndhwc_fwd(_stream(), x_ptr, n, *vol, *win, y_ptr), "srl_maxpool_ndhwc_fwd")


def maxpool_ndhwc_bwd(dy_ptr, x_ptr, n, vol, win, dact, dx_ptr):
    _check(lib().srl_maxpool_ndhwc_bwd(_stream(), dy_ptr, x_ptr, n, *vol, *win, int(dact), dx_ptr), "srl_maxpool_ndhwc_bwd")


def im2col_ndhwc(x_ptr, n, vol, kern, stride, P_ptr):
    _check(lib().srl_im2col_ndhwc(_stream(), x_ptr, n, *vol, *kern, stride, P_ptr), "srl_im2col_ndhwc")


def col2im_ndhwc(dP_ptr, n, vol, kern, stride, y_ptr, dact, dX_ptr):
    _check(lib().srl_col2im_ndhwc(_stream(), dP_ptr, n, *vol, *kern, stride, y_ptr, int(dact), dX_ptr), "srl_col2im_ndhwc")


def sgd_step(p, g, buf, lr, momentum, dampening, weight_decay, nesterov, first_step, grad_scale=1.0, max_norm=-1.0,
             sumsq=None, grad_norm_out=None):
    f = torch.float32
    _check(
        lib().srl_sgd_step(_stream(), _ptr(p, f, "p"), _ptr(g, f, "g"), _ptr(buf, f, "buf"), p.numel(), float(lr),
                           float(momentum), float(dampening), float(weight_decay), int(nesterov), int(first_step),
                           float(grad_scale), float(max_norm), _ptr(sumsq, torch.float64, "sumsq"),
                           _ptr(grad_norm_out, f, "grad_norm_out")), "srl_sgd_step")


def rmsprop_step(p, g, square_avg, buf, grad_avg, lr, alpha, eps, weight_decay, momentum, centered, grad_scale=1.0,
                 max_norm=-1.0, sumsq=None, grad_norm_out=None):
    f = torch.float32
    _check(
        lib().srl_rmsprop_step(_stream(), _ptr(p, f, "p"), _ptr(g, f, "g"), _ptr(square_avg, f, "square_avg"),
                               _ptr(buf, f, "buf"), _ptr(grad_avg, f, "grad_avg"), p.numel(), float(lr), float(alpha),
                               float(eps), float(weight_decay), float(momentum), int(centered), float(grad_scale),
                               float(max_norm), _ptr(sumsq, torch.float64, "sumsq"), _ptr(grad_norm_out, f, "grad_norm_out")),
        "srl_rmsprop_step")


class StepPlan:
    """A captured trainer step behind one C call (csrc/step_plan.hip): static input leaves, the executable graph, outputs."""

    def __init__(self, graph_exec: int):
        h = c_void_p()
        _check(lib().srl_step_plan_create(ctypes.byref(h), c_void_p(graph_exec)), "srl_step_plan_create")
        self._h = h
        self._n_in = self._n_out = 0
        self._keep = []  # tensors the plan points into

    def add_input(self, static_leaf: torch.Tensor) -> int:
        self._keep.append(static_leaf)
        rc = lib().srl_step_plan_add_input(self._h, static_leaf.data_ptr(), static_leaf.numel() * static_leaf.element_size())
        if rc < 0:
            _check(rc, "srl_step_plan_add_input")
        self._n_in += 1
        return rc

    def add_output(self, src: torch.Tensor, host_dst: Optional[torch.Tensor] = None) -> int:
        self._keep.extend([src, host_dst])
        rc = lib().srl_step_plan_add_output(self._h, src.data_ptr(), src.numel() * src.element_size(),
                                            None if host_dst is None else host_dst.data_ptr())
        if rc < 0:
            _check(rc, "srl_step_plan_add_output")
        self._n_out += 1
        return rc

    def run(self, srcs: Sequence[Optional[int]], host_dsts: Optional[Sequence[Optional[int]]] = None, sync: bool = True):
        a = (c_void_p * self._n_in)(*srcs)
        if host_dsts is None:
            d, nd = None, 0
        else:
            d, nd = (c_void_p * self._n_out)(*host_dsts), self._n_out
        _check(lib().srl_step_plan_run(self._h, _stream(), a, self._n_in, d, nd, int(sync)), "srl_step_plan_run")

    def __del__(self):
        try:
            if self._h:
                lib().srl_step_plan_destroy(self._h)
        except Exception:  # interpreter shutdown
            pass


def conv_desc(n, H, W, Cin, KH, KW, stride, Cout, act=ACT_NONE) -> ConvDesc:
    return ConvDesc(int(n), int(H), int(W), int(Cin), int(KH), int(KW), int(stride), int(Cout), int(act))


def _conv_flops(d: ConvDesc):
    oh, ow = (d.H - d.KH) // d.stride + 1, (d.W - d.KW) // d.stride + 1
    return 2.0 * d.n * oh * ow * d.Cout * d.KH * d.KW * d.Cin


def conv2d_supported(d: ConvDesc, first_layer) -> bool:
    """first_layer: 0 / False = NHWC activation layer, 1 / True = planar observation, 2 = channels-last observation."""
    return bool(lib().srl_conv2d_supported(ctypes.byref(d), int(first_layer)))


def conv2d_small_supported(d: ConvDesc) -> bool:
    """Whether the direct vector-unit kernels take this layer (3 x 3, stride 1, 4 / 8 channels: csrc/conv_small.hip)."""
    return bool(lib().srl_conv2d_small_supported(ctypes.byref(d)))


def conv2d_small_fwd(d: ConvDesc, x_ptr, w_ptr, bias_ptr, y_ptr):
    with _scope("conv_fwd", _conv_flops(d), "f32"):
        _check(lib().srl_conv2d_small_fwd(_stream(), ctypes.byref(d), x_ptr, w_ptr, bias_ptr, y_ptr), "srl_conv2d_small_fwd")


def conv2d_small_dgrad(d: ConvDesc, dz_ptr, w_ptr, x_act_ptr, dact, dx_ptr):
    with _scope("conv_dgrad", _conv_flops(d), "f32"):
        _check(lib().srl_conv2d_small_dgrad(_stream(), ctypes.byref(d), dz_ptr, w_ptr, x_act_ptr, int(dact), dx_ptr), "srl_conv2d_small_dgrad")


def conv2d_small_wgrad_workspace(d: ConvDesc) -> int:
    return int(lib().srl_conv2d_small_wgrad_workspace(ctypes.byref(d)))


def conv2d_small_wgrad(d: ConvDesc, x_ptr, dz_ptr, ws_ptr, gw_ptr, gb_ptr=None):
    with _scope("conv_wgrad", _conv_flops(d), "f32"):
        _check(lib().srl_conv2d_small_wgrad(_stream(), ctypes.byref(d), x_ptr, dz_ptr, ws_ptr, gw_ptr, gb_ptr), "srl_conv2d_small_wgrad")


def conv2d_fwd_two_piece(d: ConvDesc, x_absmax, w_absmax) -> bool:
    """Whether ``conv2d_nhwc_fwd`` takes the two-plane f16 kernel (mirrors conv.hip): only then may the weights be pre-split."""
    return (x_absmax is not None and w_absmax is not None and d.Cout > 32 and d.Cin * d.KH * d.KW >= 64 and f16x2_enabled() and
            os.environ.get("SRL_MFMA", "")[:1] != "f")


def conv2d_nhwc_fwd(d: ConvDesc, x_ptr, w_ptr, bias_ptr, y_ptr, x_absmax=None, w_absmax=None, y_absmax=None, y_mask=None,
                    presplit=False):
    two = conv2d_fwd_two_piece(d, x_absmax, w_absmax)
    with _scope("conv_fwd", _conv_flops(d), "2h" if two else "x3"):
        _check(lib().srl_conv2d_nhwc_fwd(_stream(), ctypes.byref(d), x_ptr, w_ptr, bias_ptr, y_ptr, x_absmax, w_absmax,
                                         y_absmax, y_mask, int(bool(presplit))), "srl_conv2d_nhwc_fwd")


MLP_MAX_LAYERS, MLP_MAX_WIDTH = 12, 128


def mlp_layers(layers):
    """ctypes array of srl_mlp_layer from (kind, in, out, act, w, b, gw, gb) tuples (raw pointers; gw / gb may be None for a
    forward-only chain)."""
    arr = (MlpLayer * len(layers))()
    for i, (kind, n_in, n_out, act, w, b, gw, gb) in enumerate(layers):
        arr[i] = MlpLayer(int(kind), int(n_in), int(n_out), int(act), w, b, gw, gb)
    return arr


def mlp_tape_floats(arr) -> int:
    return int(lib().srl_mlp_tape_floats(arr, len(arr)))


def mlp_tape_floats_at(arr, rows: int) -> int:
    """Floats per tape row at this row count: 0 when the pair keeps no tape (the matrix-core chain walks forward again)."""
    return int(lib().srl_mlp_tape_floats_at(arr, len(arr), int(rows)))


def mlp_bwd_max_rows(arr) -> int:
    return int(lib().srl_mlp_bwd_max_rows(arr, len(arr)))


def mlp_fwd(arr, x_ptr, ldx, rows, tape_ptr, tape_ld, y_ptr, ldy):
    flops = 2.0 * rows * sum(l.in_ * l.out for l in arr if l.kind == 1)
    with _scope("mlp_fwd", flops, "f32"):
        _check(lib().srl_mlp_fwd(_stream(), arr, len(arr), x_ptr, ldx, rows, tape_ptr, tape_ld, y_ptr, ldy), "srl_mlp_fwd")


def mlp_bwd(arr, x_ptr, ldx, rows, tape_ptr, tape_ld, dy_ptr, lddy):
    flops = 4.0 * rows * sum(l.in_ * l.out for l in arr if l.kind == 1)
    with _scope("mlp_bwd", flops, "f32"):
        _check(lib().srl_mlp_bwd(_stream(), arr, len(arr), x_ptr, ldx, rows, tape_ptr, tape_ld, dy_ptr, lddy), "srl_mlp_bwd")


def mlp_bwd_dx(arr, x_ptr, ldx, rows, dy_ptr, lddy, dx_ptr, lddx):
    """``mlp_bwd`` that also stores d loss / d x (chains that keep no tape only: ``mlp_tape_floats_at`` == 0)."""
    flops = 4.0 * rows * sum(l.in_ * l.out for l in arr if l.kind == 1)
    with _scope("mlp_bwd", flops, "f32"):
        _check(lib().srl_mlp_bwd_dx(_stream(), arr, len(arr), x_ptr, ldx, rows, dy_ptr, lddy, dx_ptr, lddx), "srl_mlp_bwd_dx")


def absmax(x_ptr, n, out_ptr):
    """*out = max(*out, max |x|) over n float32 (``srl_absmax``)."""
    _check(lib().srl_absmax(_stream(), x_ptr, int(n), out_ptr), "srl_absmax")


def relu_mask(x_ptr, n, mask_ptr):
    """Sign words of n float32 (``srl_relu_mask``): bit e & 31 of mask[e >> 5] = x[e] > 0."""
    _check(lib().srl_relu_mask(_stream(), x_ptr, int(n), mask_ptr), "srl_relu_mask")


def conv2d_wgrad_workspace(d: ConvDesc) -> int:
    return int(lib().srl_conv2d_wgrad_workspace(ctypes.byref(d)))


def conv2d_nhwc_wgrad(d: ConvDesc, x_ptr, dz_ptr, dw_ptr, ws_ptr, dbias_ptr=None, x_absmax=None, dz_absmax=None):
    """``dbias_ptr``: [Cout] += column sums of dz (the bias gradient), produced by the same kernel."""
    two = x_absmax is not None and dz_absmax is not None and d.Cout > 32 and f16x2_enabled()
    with _scope("conv_wgrad", _conv_flops(d), "2h" if two else "x3"):
        _check(lib().srl_conv2d_nhwc_wgrad(_stream(), ctypes.byref(d), x_ptr, dz_ptr, dw_ptr, ws_ptr, dbias_ptr, x_absmax,
                                           dz_absmax), "srl_conv2d_nhwc_wgrad")


def conv2d_dgrad_weight_elems(d: ConvDesc) -> int:
    return int(lib().srl_conv2d_dgrad_weight_elems(ctypes.byref(d)))


def conv2d_dgrad_repack(d: ConvDesc, w_ptr, wt_ptr):
    _check(lib().srl_conv2d_dgrad_repack(_stream(), ctypes.byref(d), w_ptr, wt_ptr), "srl_conv2d_dgrad_repack")


def conv2d_dgrad_two_piece(d: ConvDesc, dz_absmax, w_absmax) -> bool:
    """Whether ``conv2d_nhwc_dgrad`` takes the two-plane f16 kernel for every parity class (mirrors conv.hip)."""
    s = d.stride
    uniform = d.KH % s == 0 and d.KW % s == 0 and d.H % s == 0 and d.W % s == 0
    ncols = (s * s if uniform else 1) * d.Cin
    return (dz_absmax is not None and w_absmax is not None and d.Cout % 16 == 0 and ncols > 32 and f16x2_enabled() and uniform and
            os.environ.get("SRL_MFMA", "")[:1] != "f")


def conv2d_nhwc_dgrad(d: ConvDesc, dz_ptr, wt_ptr, x_act_ptr, dact, dx_ptr, dz_absmax=None, w_absmax=None, dx_absmax=None,
                      x_mask=None, presplit=False):
    two = dz_absmax is not None and w_absmax is not None and d.Cout % 16 == 0 and f16x2_enabled()
    with _scope("conv_dgrad", _conv_flops(d), "2h" if two else "x3"):
        _check(lib().srl_conv2d_nhwc_dgrad(_stream(), ctypes.byref(d), dz_ptr, wt_ptr, x_act_ptr, int(dact), dx_ptr,
                                           dz_absmax, w_absmax, dx_absmax, x_mask, int(bool(presplit))), "srl_conv2d_nhwc_dgrad")


def conv2d_obs_fwd_workspace(d: ConvDesc) -> int:
    return int(lib().srl_conv2d_obs_fwd_workspace(ctypes.byref(d)))


def conv2d_obs_row_index_supported(d: ConvDesc, is_u8, channels_last) -> bool:
    """Whether ``conv2d_obs_fwd / bwd`` honour ``row_index`` for this geometry (the byte kernels do)."""
    return bool(lib().srl_conv2d_obs_row_index_supported(ctypes.byref(d), int(is_u8), int(channels_last)))


def conv2d_obs_fwd(d: ConvDesc, obs_ptr, is_u8, mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, w_ptr, bias_ptr, y_ptr,
                   channels_last=False, ws_ptr=None, row_index: Optional[torch.Tensor] = None, y_absmax=None, y_mask=None,
                   reuse_folded=False):
    with _scope("conv_obs_fwd", _conv_flops(d), "obs"):
        _check(
            lib().srl_conv2d_obs_fwd(_stream(), ctypes.byref(d), obs_ptr, int(is_u8), int(channels_last), mean_ptr,
                                     rstd_ptr, gamma_ptr, beta_ptr, w_ptr, bias_ptr, y_ptr, ws_ptr,
                                     _ptr(row_index, torch.int32, "row_index"), y_absmax, y_mask, int(bool(reuse_folded))),
            "srl_conv2d_obs_fwd")


def obs_space_to_depth(obs_ptr, is_u8, n, C, H, W, s, out_ptr, mean_ptr, rstd_ptr):
    with _scope("obs_space_to_depth"):
        _check(lib().srl_obs_space_to_depth(_stream(), obs_ptr, int(is_u8), n, C, H, W, s, out_ptr, mean_ptr, rstd_ptr),
               "srl_obs_space_to_depth")


def gather_rows(src_ptr, row_bytes, index: torch.Tensor, n, dst_ptr):
    """dst[i, :] = src[index[i], :] (``srl_gather_rows``); ``index`` int32 [>= n] on the device."""
    with _scope("gather_rows"):
        _check(lib().srl_gather_rows(_stream(), src_ptr, int(row_bytes), _ptr(index, torch.int32, "index"), int(n), dst_ptr),
               "srl_gather_rows")


def ring_slots(refs: torch.Tensor, capacity: int, out: torch.Tensor, base: int = 0):
    """out[i] = (refs[i] - base) % capacity (``srl_ring_slots``): int64 stamps -> int32 storage slots, on the device."""
    _check(lib().srl_ring_slots(_stream(), _ptr(refs, torch.int64, "refs"), refs.numel(), int(capacity), int(base),
                                _ptr(out, torch.int32, "slots")), "srl_ring_slots")


def ring_stack_push(store: torch.Tensor, planes: torch.Tensor, prev: torch.Tensor, slot0: int, C: int, H: int, W: int,
                    mean: torch.Tensor, rstd: torch.Tensor):
    """Rows ``slot0 ...`` of the ring's space-to-depth storage = [channels 1.. of row prev[i], planes[i]] (``srl_ring_stack_push``)."""
    n = planes.shape[0]
    _check(lib().srl_ring_stack_push(_stream(), _ptr(store, torch.uint8, "store"), _ptr(planes, torch.uint8, "planes"),
                                     _ptr(prev, torch.int32, "prev"), int(slot0), int(n), int(C), int(H), int(W),
                                     _ptr(mean, torch.float32, "mean"), _ptr(rstd, torch.float32, "rstd")), "srl_ring_stack_push")


def conv2d_obs_bwd_workspace(d: ConvDesc) -> int:
    return int(lib().srl_conv2d_obs_bwd_workspace(ctypes.byref(d)))


def conv2d_obs_bwd(d: ConvDesc, obs_ptr, is_u8, mean_ptr, rstd_ptr, gamma_ptr, beta_ptr, w_ptr, dz_ptr, dw_ptr, db_ptr,
                   dgamma_ptr, dbeta_ptr, ws_ptr, channels_last=False, row_index: Optional[torch.Tensor] = None, phase: int = 3,
                   dz_absmax_ptr=None):
    """``phase``: 3 = a call of its own; bit 0 opens / bit 1 closes an accumulation of the position sums over several calls (the
    chunks of one update then share one finalisation; ``srl_hip.h``).  ``dz_absmax_ptr``: device float >= max |dz| (selects the
    block kernel of ``csrc/obs_h2.h`` on the Atari geometry)."""
    # with a bound of |dz| the Atari geometry runs obs_h2.h's weight gradient: bytes x two f16 pieces of dz'
    with _scope("conv_obs_bwd", _conv_flops(d), "obs2" if dz_absmax_ptr and os.environ.get("SRL_OBS_BWD_H2BLOCK", "")[:1] != "0" else "obs"):
        _check(
            lib().srl_conv2d_obs_bwd(_stream(), ctypes.byref(d), obs_ptr, int(is_u8), int(channels_last), mean_ptr,
                                     rstd_ptr, gamma_ptr, beta_ptr, w_ptr, dz_ptr, dw_ptr, db_ptr, dgamma_ptr, dbeta_ptr,
                                     ws_ptr, _ptr(row_index, torch.int32, "row_index"), int(phase),
                                     dz_absmax_ptr or None), "srl_conv2d_obs_bwd")


def _wrap_for_profile(names):
    import functools
    g = globals()
    for name in names:
        fn = g[name]

        def make(fn, name):

            @functools.wraps(fn)
            def wrapper(*a, **k):
                if _prof is None:
                    return fn(*a, **k)
                with _scope(name):
                    return fn(*a, **k)

            return wrapper

        g[name] = make(fn, name)


_wrap_for_profile(["masked_stats", "masked_normalize", "ppo_loss_fwd_bwd", "categorical_fwd", "categorical_bwd",
                   "categorical_sample", "layernorm_fwd", "layernorm_bwd", "obs_ln_stats", "im2col_obs_ln", "im2col_nhwc",
                   "col2im_nhwc", "obs_ln_affine_bwd", "colsum", "copy2d", "grad_sumsq", "adam_step", "sgd_step",
                   "rmsprop_step"])


# ------------------------------------------------------------------------------------------------ pre-split ("h2") kernels
H2_CONV2_FWD, H2_CONV3_FWD, H2_CONV3_DGRAD, H2_CONV2_DGRAD = 0, 1, 2, 3
H2_WGRAD_CONV2, H2_WGRAD_CONV3 = 0, 1


def _vp(v):
    """Device pointer (int) or None -> c_void_p."""
    return c_void_p(int(v)) if v else None


def h2_pack_rows(src_ptr, ld, rows, C, dst_ptr, absmax=None, scale_in=None, scale_out=None):
    _check(lib().srl_h2_pack_rows(_stream(), _vp(src_ptr), int(ld), int(rows), int(C), _vp(absmax), _vp(scale_in), _vp(scale_out),
                                  _vp(dst_ptr)), "srl_h2_pack_rows")


def h2_pack_rows_colsum_workspace(rows, C) -> int:
    return int(lib().srl_h2_pack_rows_colsum_workspace(int(rows), int(C)))


def h2_pack_rows_colsum(src, ld, rows, C, dst, workspace, colsum, absmax=None, scale_in=None, scale_out=None, accumulate=True):
    """``h2_pack_rows`` + colsum[C] (+)= column sums of src (the bias gradient of the layer whose output gradient is packed)."""
    _check(lib().srl_h2_pack_rows_colsum(_stream(), _vp(src), int(ld), int(rows), int(C), _vp(absmax), _vp(scale_in), _vp(scale_out), _vp(dst),
                                         _vp(workspace), _vp(colsum), int(bool(accumulate))), "srl_h2_pack_rows_colsum")


def h2_unpack_rows(src_ptr, rows, C, scale, dst_ptr, ld):
    _check(lib().srl_h2_unpack_rows(_stream(), _vp(src_ptr), int(rows), int(C), _vp(scale), _vp(dst_ptr), int(ld)), "srl_h2_unpack_rows")


def h2_pack_image(src_ptr, n, H, W, C, layout, dst_ptr, absmax=None, scale_in=None, scale_out=None):
    _check(lib().srl_h2_pack_image(_stream(), _vp(src_ptr), int(n), int(H), int(W), int(C), int(layout), _vp(absmax), _vp(scale_in),
                                   _vp(scale_out), _vp(dst_ptr)), "srl_h2_pack_image")


def h2_unpack_image(src_ptr, n, H, W, C, layout, scale, dst_ptr):
    _check(lib().srl_h2_unpack_image(_stream(), _vp(src_ptr), int(n), int(H), int(W), int(C), int(layout), _vp(scale), _vp(dst_ptr)),
           "srl_h2_unpack_image")


def h2_weights(w_ptr, rows, K, mode, absmax, scale_out, rownorm_out, dst_ptr, desc=None):
    _check(lib().srl_h2_weights(_stream(), _vp(w_ptr), int(rows), int(K), int(mode), ctypes.byref(desc) if desc is not None else None,
                                _vp(absmax), _vp(scale_out), _vp(rownorm_out), _vp(dst_ptr)), "srl_h2_weights")


def h2_conv(kind, x, w, sx, sw, n, out, out_absmax, bias=None, act=0, out_scale=None, bound_in=None, bound_w=None, bound_b=None,
            mask_out=None, mask_in=None):
    a = H2ConvArgs(_vp(x), _vp(w), _vp(sx), _vp(sw), int(n), _vp(bias), int(act), _vp(out), _vp(out_scale), _vp(bound_in), _vp(bound_w),
                   _vp(bound_b), _vp(out_absmax), _vp(mask_out), _vp(mask_in))
    # algorithmic flops of the convolution each kind stands for (data gradients: the forward layer's count)
    flops = 2.0 * int(n) * (81 * 64 * 512 if kind in (H2_CONV2_FWD, H2_CONV2_DGRAD) else 49 * 64 * 576)
    name = ("conv_fwd", "conv_fwd", "conv_dgrad", "conv_dgrad")[int(kind)]
    with _scope(name, flops, "2h"):
        _check(lib().srl_h2_conv(_stream(), int(kind), ctypes.byref(a)), "srl_h2_conv")


def h2_wgrad_workspace(kind) -> int:
    return int(lib().srl_h2_wgrad_workspace(int(kind)))


def h2_wgrad(kind, x, dz, sx, sz, n, workspace, gw, gb=None):
    flops = 2.0 * int(n) * (81 * 64 * 512 if kind == H2_WGRAD_CONV2 else 49 * 64 * 576)
    with _scope("conv_wgrad", flops, "2h"):
        _check(lib().srl_h2_wgrad(_stream(), int(kind), _vp(x), _vp(dz), _vp(sx), _vp(sz), int(n), _vp(workspace), _vp(gw), _vp(gb)),
               "srl_h2_wgrad")


def h2_wgrad_dense_workspace(M, NA, NB) -> int:
    return int(lib().srl_h2_wgrad_dense_workspace(int(M), int(NA), int(NB)))


def h2_wgrad_dense(a, b, sa, sb, M, NA, NB, workspace, gw, accumulate=True, a_row_bytes=None, b_row_bytes=None):
    """gw[NA][NB] (+)= a^T b over h2p rows: a [M][NA] (the output gradient), b [M][NB] (the layer's input)."""
    with _scope("gemm", 2.0 * int(M) * int(NA) * int(NB), "2h"):
        _check(lib().srl_h2_wgrad_dense(_stream(), _vp(a), _vp(b), _vp(sa), _vp(sb), int(M), int(NA), int(NB),
                                        int(a_row_bytes or 4 * NA), int(b_row_bytes or 4 * NB), _vp(workspace), _vp(gw), int(bool(accumulate))),
               "srl_h2_wgrad_dense")


def h2_gemm(x, w, sx, sw, M, NC, K, out, bias=None, act=0, out_h2=False, out_scale=None, bound_in=None, bound_w=None, bound_b=None,
            out_absmax=None, mask_out=None, mask_in=None, mask_in_h2order=False):
    d = H2GemmDesc(_vp(x), _vp(w), _vp(sx), _vp(sw), int(M), int(NC), int(K), _vp(bias), int(act), int(bool(out_h2)), _vp(out),
                   _vp(out_scale), _vp(bound_in), _vp(bound_w), _vp(bound_b), _vp(out_absmax), _vp(mask_out), _vp(mask_in),
                   int(bool(mask_in_h2order)))
    with _scope("gemm", 2.0 * int(M) * int(NC) * int(K), "2h"):
        _check(lib().srl_h2_gemm(_stream(), ctypes.byref(d)), "srl_h2_gemm")


def h2_gemm_splitk(x, w, sx, sw, M, NC, K, out, ksplits, wide=False):
    """``h2_gemm`` with its reduction split over ``ksplits`` workgroups per tile: ``out`` = [ksplits][M][NC] float32 raw partial sums.
    ``wide``: 256 channels per workgroup (training chunks)."""
    d = H2GemmDesc(_vp(x), _vp(w), _vp(sx), _vp(sw), int(M), int(NC), int(K), None, 0, 0, _vp(out), None, None, None, None, None, None, None, 0)
    with _scope("gemm", 2.0 * int(M) * int(NC) * int(K), "2h"):
        _check(lib().srl_h2_gemm_splitk(_stream(), ctypes.byref(d), int(ksplits), int(bool(wide))), "srl_h2_gemm_splitk")


def conv2d_obs_fold_h2(desc, gamma, beta, w, bias, ws_ptr) -> bool:
    """Only the folded first-layer weights ``conv2d_obs_fwd_h2(..., reuse_folded=True)`` reads.  False: not that kernel's layer."""
    rc = lib().srl_conv2d_obs_fold_h2(_stream(), ctypes.byref(desc), _vp(gamma), _vp(beta), _vp(w), _vp(bias), _vp(ws_ptr))
    if rc == 1:
        return False
    _check(rc, "srl_conv2d_obs_fold_h2")
    return True


def conv2d_obs_fwd_h2(desc, obs_ptr, mean, rstd, gamma, beta, w, bias, y_h2, y_scale, ws_ptr, row_index, y_absmax, y_mask,
                      reuse_folded=False, ent_order=2, records=None):
    ri = _ptr(row_index, torch.int32, "row_index") if isinstance(row_index, torch.Tensor) else row_index
    with _scope("conv_obs_fwd", _conv_flops(desc), "obs2"):
        _check(lib().srl_conv2d_obs_fwd_h2(_stream(), ctypes.byref(desc), _vp(obs_ptr), _vp(mean), _vp(rstd), _vp(gamma), _vp(beta), _vp(w),
                                           _vp(bias), _vp(y_h2), _vp(y_scale), _vp(ws_ptr), _vp(ri), _vp(y_absmax), _vp(y_mask),
                                           int(bool(reuse_folded)), int(ent_order), _vp(records)), "srl_conv2d_obs_fwd_h2")
