"""ctypes binding of libselfmask_hip.so (C ABI declared in include/selfmask_hip.h).

The product path has NO CPU fallback: if the HIP library is missing, ``load()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SM_HIP_LIB points the binding at another build of the same ABI - the tuning build with the timing-only ablation
# switches (build.py --tuning -> lib/libselfmask_hip_tuning.so), used by scripts/*_ablate.py only
LIB_PATH = os.environ.get("SM_HIP_LIB") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libselfmask_hip.so"))

EMBED, HEADS, HEAD_DIM, MLP, ENC_DEPTH, MAX_DEC_LAYERS = 384, 6, 64, 1536, 12, 8
EPI_BIAS, EPI_GELU, EPI_RELU, EPI_RESIDUAL, EPI_SIGMOID2, EPI_PATCH, EPI_RESIDUAL_LN = range(7)

fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class GemmArgs(C.Structure):
    _fields_ = [("A", fp), ("W", fp), ("bias", fp), ("C", fp), ("R", fp), ("C2", fp), ("A_alt", fp),
                ("strideA", C.c_int64), ("strideW", C.c_int64), ("strideC", C.c_int64), ("strideR", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldw", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
                ("batch", C.c_int32), ("epilogue", C.c_int32), ("alt_from_n", C.c_int32), ("split_k", C.c_int32),
                ("patch_n", C.c_int32), ("ln_gamma", fp), ("ln_beta", fp), ("ln_eps", C.c_float), ("w_scale", C.c_float),
                ("ln_stats", fp), ("ln_c", fp), ("ln_stats_out", fp), ("mfma_terms", C.c_int32)]


class RowMap(C.Structure):
    _fields_ = [("group", C.c_int32), ("stride", C.c_int32), ("offset", C.c_int32)]


class LnArgs(C.Structure):
    _fields_ = [("x", fp), ("ldx", C.c_int64), ("in_map", RowMap), ("gamma", fp), ("beta", fp),
                ("y", fp), ("ldy", C.c_int64), ("out_map", RowMap), ("y2", fp), ("ldy2", C.c_int64),
                ("add", fp), ("add_rows", C.c_int32), ("rows", C.c_int32), ("eps", C.c_float), ("n_partials", C.c_int32),
                ("partial_stride", C.c_int64), ("pre_bias", fp), ("residual", fp), ("ys", fp), ("y2_f16x2", C.c_int32),
                ("raw", fp), ("chain_gamma", fp), ("chain_beta", fp), ("chain_y", fp), ("chain_ys", fp), ("chain_ldy", C.c_int64),
                ("chain_map", RowMap), ("chain_eps", C.c_float)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", fp), ("K", fp), ("V", fp), ("O", fp),
                ("sQb", C.c_int64), ("sQr", C.c_int64), ("sKb", C.c_int64), ("sKr", C.c_int64),
                ("sVb", C.c_int64), ("sVr", C.c_int64), ("sOb", C.c_int64), ("sOr", C.c_int64),
                ("batch", C.c_int32), ("heads", C.c_int32), ("n_q", C.c_int32), ("n_k", C.c_int32),
                ("scale", C.c_float), ("out_f16x2", C.c_int32)]


class QkvAttnArgs(C.Structure):
    _fields_ = [("Xn", fp), ("Wqkv", fp), ("bias", fp), ("O", fp), ("ldx", C.c_int64), ("ldo", C.c_int64),
                ("B", C.c_int32), ("N", C.c_int32), ("w_scale", C.c_float), ("scale", C.c_float), ("out_f16x2", C.c_int32),
                ("mfma_terms", C.c_int32), ("ln_stats", fp), ("ln_c", fp), ("ln_eps", C.c_float)]


ENC_FIELDS = ["norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b",
              "fc2_w", "fc2_b"]
DEC_FIELDS = ["sa_in_w", "sa_in_b", "sa_out_w", "sa_out_b", "ca_in_w", "ca_in_b", "ca_out_w", "ca_out_b",
              "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b", "norm3_w",
              "norm3_b"]


class EncLayer(C.Structure):
    _fields_ = [(n, fp) for n in ENC_FIELDS] + [(n, fp) for n in ("qkv_fw", "qkv_fb", "qkv_c", "fc1_fw", "fc1_fb", "fc1_c")] + \
               [(n, C.c_float) for n in ("qkv_s", "proj_s", "fc1_s", "fc2_s", "qkv_fs", "fc1_fs")]


class DecLayer(C.Structure):
    _fields_ = [(n, fp) for n in DEC_FIELDS] + [(n, C.c_float) for n in ("sa_in_s", "sa_out_s", "ca_in_s", "ca_out_s",
                                                                          "lin1_s", "lin2_s")]


class Weights(C.Structure):
    _fields_ = [("query_embed", fp), ("cls_token", fp), ("pos_embed", fp), ("patch_w", fp), ("patch_b", fp),
                ("enc", EncLayer * ENC_DEPTH), ("enc_norm_w", fp), ("enc_norm_b", fp),
                ("dec", DecLayer * MAX_DEC_LAYERS), ("dec_norm_w", fp), ("dec_norm_b", fp),
                ("ffn0_w", fp), ("ffn0_b", fp), ("ffn1_w", fp), ("ffn1_b", fp), ("ffn2_w", fp), ("ffn2_b", fp),
                ("dec_kv_w", fp), ("dec_kv_b", fp), ("gemm_mode", C.c_int32), ("patch", C.c_int32), ("pos_grid", C.c_int32), ("n_queries", C.c_int32),
                ("n_dec_layers", C.c_int32), ("patch_s", C.c_float), ("ffn0_s", C.c_float), ("ffn1_s", C.c_float),
                ("dec_kv_s", C.c_float), ("ffn2_s", C.c_float), ("mask_head_ffn", C.c_int32), ("normalize_before", C.c_int32),
                ("ln_fold", C.c_int32), ("scale_factor", C.c_int32), ("no_objectness", C.c_int32)]


class ForwardIO(C.Structure):
    _fields_ = [("x", fp), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("mask_logits", fp), ("mask_pred", fp), ("objectness", fp), ("features", fp), ("queries", fp),
                ("patch_tokens", fp), ("encoder_only", C.c_int32), ("attn_path", C.c_int32),
                ("last_layer_only", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("launches", C.c_int32), ("total_us", C.c_double), ("overhead_us", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


class PreImage(C.Structure):
    _fields_ = [("off", C.c_int64), ("out_off", C.c_int64), ("H", C.c_int32), ("W", C.c_int32), ("coef_x", C.c_int32),
                ("coef_y", C.c_int32), ("ksx", C.c_int32), ("ksy", C.c_int32)]


class EvalImage(C.Structure):
    _fields_ = [("gt_off", C.c_int64), ("H", C.c_int32), ("W", C.c_int32)]


class EvalArgs(C.Structure):
    _fields_ = [("mask_pred", fp), ("mask_stride_b", C.c_int64), ("objectness", fp), ("obj_stride_b", C.c_int64),
                ("gt", fp), ("images", fp), ("thresholds", fp), ("rows", fp), ("ious", fp), ("workspace", fp),
                ("workspace_bytes", C.c_size_t), ("B", C.c_int32), ("nq", C.c_int32), ("mh", C.c_int32),
                ("mw", C.c_int32), ("max_pixels", C.c_int32), ("scale", C.c_float)]


class BilateralArgs(C.Structure):
    _fields_ = [("img", fp), ("target", fp), ("soft", fp), ("binary", fp), ("info", fp), ("workspace", fp),
                ("workspace_bytes", C.c_size_t), ("sigma_spatial", C.c_double), ("sigma_luma", C.c_double),
                ("sigma_chroma", C.c_double), ("lam", C.c_double), ("a_diag_min", C.c_double), ("cg_tol", C.c_double),
                ("confidence", C.c_double), ("cg_maxiter", C.c_int32), ("H", C.c_int32), ("W", C.c_int32)]


class SpectralArgs(C.Structure):
    _fields_ = [("features", fp), ("labels", fp), ("cluster_sizes", C.POINTER(C.c_int32)), ("knn", fp), ("eigenvalues", fp),
                ("embedding", fp), ("residuals", fp), ("info", fp), ("workspace", fp), ("workspace_bytes", C.c_size_t),
                ("tol", C.c_double), ("B", C.c_int32), ("n", C.c_int32), ("n_sizes", C.c_int32), ("n_neighbors", C.c_int32),
                ("degree", C.c_int32), ("max_outer", C.c_int32), ("kmeans_max_iter", C.c_int32)]


# every symbol include/selfmask_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sm_version": (C.c_int, []),
    "sm_last_error": (C.c_char_p, []),
    "sm_gemm_f32": (C.c_int, [C.POINTER(GemmArgs), fp]),
    "sm_gemm_f32_tile": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_int, fp]),
    "sm_gemm_f32_pick_tile": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sm_split_f16x2": (C.c_int, [fp, C.c_int64, fp, C.c_int64, C.c_int64, C.c_int32, fp]),
    "sm_gemm_f16x2_tile": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_int, C.c_int, fp]),
    "sm_gemm_f16x2": (C.c_int, [C.POINTER(GemmArgs), C.c_int, fp]),
    "sm_gemm_f16x2_pick_tile": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sm_split_w16": (C.c_int, [fp, C.c_int64, fp, C.c_int64, C.c_int64, C.c_int32, C.c_float, fp]),
    "sm_gemm_w16_tile": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_int, fp]),
    "sm_gemm_w16": (C.c_int, [C.POINTER(GemmArgs), C.c_int, fp]),
    "sm_gemm_w16_pick": (C.c_int, [C.POINTER(GemmArgs)]),
    "sm_gemm_w16_variant_name": (C.c_char_p, [C.c_int]),
    "sm_im2col_patches_f16x2": (C.c_int, [fp, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_upsample2x_tokens_f16x2": (C.c_int, [fp, C.c_int64, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_layernorm_f32": (C.c_int, [fp, C.c_int64, fp, fp, fp, C.c_int64, C.c_int32, C.c_int32, C.c_float, fp]),
    "sm_layernorm_rows_f32": (C.c_int, [C.POINTER(LnArgs), fp]),
    "sm_broadcast_rows_f32": (C.c_int, [fp, fp, C.c_int32, C.c_int32, fp]),
    "sm_attention_f32": (C.c_int, [C.POINTER(AttnArgs), fp]),
    "sm_attention_f16x2": (C.c_int, [C.POINTER(AttnArgs), fp]),
    "sm_qkv_attention_w16": (C.c_int, [C.POINTER(QkvAttnArgs), fp]),
    "sm_qkv_attention_max_tokens": (C.c_int, []),
    "sm_im2col_patches_f32": (C.c_int, [fp, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_cls_rows_f32": (C.c_int, [fp, fp, fp, C.c_int32, C.c_int32, fp]),
    "sm_pos_embed_bicubic_f32": (C.c_int, [fp, C.c_int32, fp, C.c_int32, C.c_int32, fp]),
    "sm_upsample2x_tokens_f32": (C.c_int, [fp, C.c_int64, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_upsample2x_logits_sigmoid_f32": (C.c_int, [fp, fp, fp, C.c_int64, C.c_int32, C.c_int32, fp]),
    "sm_upsample_tokens_f32": (C.c_int, [fp, C.c_int64, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_upsample_tokens_f16x2": (C.c_int, [fp, C.c_int64, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_upsample_logits_sigmoid_f32": (C.c_int, [fp, fp, fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_rowdot_sigmoid_f32": (C.c_int, [fp, fp, fp, fp, C.c_int32, fp]),
    "sm_query_mean_f32": (C.c_int, [fp, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_preprocess_resize_u8": (C.c_int, [fp, fp, fp, fp, fp, C.c_int64, fp, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_upsample_selected_f64": (C.c_int, [fp, C.c_int64, fp, C.c_int32, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_mask_u8_to_f32": (C.c_int, [fp, fp, C.c_int64, fp]),
    "sm_preprocess_normalize_u8": (C.c_int, [fp, fp, fp, fp, C.c_int32, C.c_int32, fp]),
    "sm_preprocess_normalize_pad_u8": (C.c_int, [fp, fp, fp, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_pick_mask_f32": (C.c_int, [fp, C.c_int64, fp, C.c_int64, fp, fp, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_vote_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "sm_vote_masks_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, fp, fp, C.c_size_t, fp]),
    "sm_rle_runs_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, fp, fp, C.c_int32, fp, fp]),
    "sm_vote_masks_sized_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, C.c_int32, C.c_int32, fp, fp, fp, fp, fp, C.c_size_t, fp]),
    "sm_vote_masks_batch_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, fp, fp, C.c_size_t, fp]),
    "sm_labels_to_masks_batch_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.c_int32, fp, fp]),
    "sm_upsample_tokens_aligned_f32": (C.c_int, [fp, C.c_int64, fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp]),
    "sm_kmeans_f32": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp, fp, fp]),
    "sm_labels_to_masks_u8": (C.c_int, [fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, fp, fp]),
    "sm_spectral_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sm_spectral_cluster_f32": (C.c_int, [C.POINTER(SpectralArgs), fp]),
    "sm_evaluate_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "sm_evaluate_masks_f32": (C.c_int, [C.POINTER(EvalArgs), fp]),
    "sm_bilateral_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double]),
    "sm_bilateral_solver_f64": (C.c_int, [C.POINTER(BilateralArgs), fp]),
    "sm_bilateral_solver_batch_f64": (C.c_int, [C.POINTER(BilateralArgs), C.c_int32, fp]),
    "sm_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Weights), C.c_int32, C.c_int32, C.c_int32]),
    "sm_maskformer_forward": (C.c_int, [C.POINTER(Weights), C.POINTER(ForwardIO), fp, C.c_size_t, fp]),
    "sm_forward_timing": (C.c_int, [C.c_int]),
    "sm_forward_timing_read": (C.c_int, [C.POINTER(KernelTime), C.c_int]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or fail loudly (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"selfmask_amd: {LIB_PATH} is missing - build it with `python salient-object-detection_amd/build.py` "
            f"(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the product path.")
    try:  # make sure PyTorch's HIP runtime (same soname, libamdhip64.so.7) is the one already mapped
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - the library itself only needs libamdhip64
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().sm_last_error()
        raise RuntimeError(f"selfmask_hip {what} failed (rc={rc}): {msg.decode() if msg else ''}")
