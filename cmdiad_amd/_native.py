"""ctypes binding of libcmdiad_hip.so (C ABI: include/cmdiad_hip.h).

There is NO CPU fallback: if the shared object is missing or a call fails, this raises.
`build()` compiles the library with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("CMDIAD_HIP_LIB") or os.path.join(_HERE, "libcmdiad_hip.so")  # override: A/B runs of two builds
_lib = None


class NativeError(RuntimeError):
    pass


class GemmArgs(Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("W", c_void_p), ("ldw", c_int),
                ("M", c_int), ("N", c_int), ("K", c_int),
                ("bias", c_void_p), ("group_bias", c_void_p), ("group_rows", c_int), ("act", c_int),
                ("residual", c_void_p), ("ldr", c_int),
                ("out_f32", c_void_p), ("ldo32", c_int), ("out_bf16", c_void_p), ("ldo16", c_int),
                ("out_pre_bf16", c_void_p), ("dact_of", c_void_p), ("split_k", c_int), ("m_count", c_void_p),
                ("row_scale", c_void_p), ("ln_xb", c_void_p), ("ld_xb", c_int), ("ln_part", c_void_p),
                ("add2", c_void_p), ("ld_add2", c_int)]


class ConvArgs(Structure):
    _fields_ = [("x", c_void_p), ("B", c_int), ("H", c_int), ("Wd", c_int), ("C", c_int),
                ("W", c_void_p), ("N", c_int), ("ksize", c_int), ("stride", c_int),
                ("bias", c_void_p), ("act", c_int), ("residual", c_void_p), ("ldr", c_int),
                ("out_f32", c_void_p), ("ldo32", c_int), ("out_bf16", c_void_p), ("ldo16", c_int)]


class BlockWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "qkv_wf", "qkv_bf", "fc1_wf", "fc1_bf")]


P, I, F, SZ, U32, D = c_void_p, c_int, c_float, c_size_t, c_uint32, c_double
# name -> argtypes (every entry point declared in include/cmdiad_hip.h; tests check the two agree)
SIGNATURES = {
    "cmdiad_fps": [P, P, I, I, I, P, P, P, SZ, P],
    "cmdiad_knn_group": [P, P, P, I, I, I, I, P, P, P],
    "cmdiad_knn_group_ws": [P, P, P, I, I, I, I, P, P, P, SZ, P],
    "cmdiad_unorganize": [P, I, I, I, P, P, P, P, P],
    "cmdiad_interp3nn": [P, P, P, I, I, I, P, P, P],
    "cmdiad_interp3nn_ws": [P, P, P, I, I, I, P, P, P, SZ, P],
    "cmdiad_interp_gather": [P, P, P, P, I, I, I, I, P, P],
    "cmdiad_xyz_patch_fused": [P, P, P, P, I, I, I, I, I, I, F, F, P, P, P],
    "cmdiad_gemm_bf16": [POINTER(GemmArgs), P],
    "cmdiad_gemm_streamk_bf16": [POINTER(GemmArgs), P, SZ, P],
    "cmdiad_gemm_streamk_eligible": [I, I, I],
    "cmdiad_gemm_qkv": [P, P, P, P, I, I, I, P, P, P, P],
    "cmdiad_ln_stats_finalize": [P, I, I, F, P, P, P],
    "cmdiad_gemm_tn_bf16": [P, I, P, I, I, I, I, I, P, I, P, P],
    "cmdiad_attention": [P, P, P, I, I, I, P, P],
    "cmdiad_encoder_tail": [P, P, P, P, P, I, I, P, P],
    "cmdiad_conv2d_nhwc_bf16": [POINTER(ConvArgs), P],
    "cmdiad_conv_stem": [P, P, P, I, I, I, I, I, I, P, P],
    "cmdiad_upsample_bicubic": [P, I, I, I, I, I, I, I, P, I, P, P],
    "cmdiad_transformer_block_fwd": [P, P, POINTER(BlockWeights), I, I, I, I, I, F, I, P, P, P, P, SZ, P],
    "cmdiad_layernorm": [P, P, P, P, F, I, I, P, P, I, P, P, P],
    "cmdiad_loss_head": [P, P, I, I, I, F, P, P, P, P],
    "cmdiad_bn_affine": [P, P, P, P, SZ, D, I, P, P, P, P, P, P, P],
    "cmdiad_bn_relu_fwd": [P, P, P, P, I, SZ, I, P, P, P],
    "cmdiad_bn_relu_bwd_reduce": [P, P, P, P, P, P, I, SZ, I, I, P, P, P],
    "cmdiad_bn_partials_sum": [P, P, I, I, P, P, P],
    "cmdiad_bn_relu_bwd_apply": [P, P, P, P, P, P, P, P, I, SZ, I, P, P],
    "cmdiad_pad_nhwc_bf16": [P, I, I, I, I, P, P],
    "cmdiad_relu_bwd_bf16": [P, P, SZ, P, P, P],
    "cmdiad_upsample_bicubic_bwd": [P, I, I, I, I, I, I, P, P, P],
    "cmdiad_reduce_slabs": [P, I, SZ, SZ, F, P, P],
    "cmdiad_sum_vector": [P, SZ, F, P, P],
    "cmdiad_colsum_bf16": [P, I, I, I, P, P],
    "cmdiad_ln_param_grad": [P, P, P, P, I, I, I, P, P, P],
    "cmdiad_adam_step": [P, P, P, P, SZ, F, F, F, F, I, F, P, P],
    "cmdiad_encoder_stage1": [P, P, P, P, I, I, P, P, P, P],
    "cmdiad_gemm_groupmax": [P, P, P, I, I, I, I, P, P, P],
    "cmdiad_l2_min_keys": [P, P, P, P, I, I, I, U32, P, P, I, P],
    "cmdiad_l2_min_keys_counted": [P, P, P, I, P, P, I, I, U32, P, P, I, P],
    "cmdiad_l2_min_keys_segments": [P, P, P, I, I, P, P, I, I, U32, P, P, I, P],
    "cmdiad_rows_dedup_plan": [P, P, I, I, P, P, P, P, P, P, P],
    "cmdiad_keys_expand": [P, P, I, P, P],
    "cmdiad_rows_expand_f32": [P, P, I, I, P, P],
    "cmdiad_l2_rescore": [P, P, P, I, I, I, U32, P, P, P],
    "cmdiad_l2_rescore2": [P, P, P, P, I, I, I, U32, P, P, P, P],
    "cmdiad_l2_choose": [P, P, P, I, P, P, P],
    "cmdiad_reweight_scan": [P, P, P, I, I, I, U32, P, P, SZ, P],
    "cmdiad_reweight_scan_pair": [P, P, P, I, I, U32, P, P, P, P, I, I, U32, P, I, P, SZ, P],
    "cmdiad_bank_block16": [P, I, I, P, P],
    "cmdiad_col_moments": [P, SZ, I, I, P, P, P],
    "cmdiad_moments3": [P, SZ, P, P],
    "cmdiad_l2_dist_matrix": [P, P, I, I, I, P, P],
    "cmdiad_score_head": [P, P, P, P, I, I, I, I, U32, P, P, P, P, P],
    "cmdiad_score_tail": [P, P, P, P, I, I, I, U32, P, P],
    "cmdiad_score_final": [P, P, I, I, P, P],
    "cmdiad_coreset_greedy": [P, I, I, I, I, P, P, SZ, P],
    "cmdiad_coreset_greedy_f32": [P, I, I, I, I, P, P, SZ, P],
    "cmdiad_coreset_prepare": [P, I, I, I, P, SZ, P],
    "cmdiad_coreset_round": [P, I, I, I, I, P, I, P, P],
    "cmdiad_coreset_decode": [P, I, I, P, P],
    "cmdiad_sparse_project_f32": [P, SZ, I, P, P, P, I, P, P],
    "cmdiad_normalize_cast": [P, SZ, I, F, F, P, P, P, I, P],
    "cmdiad_normalize_cast_rows": [P, SZ, I, I, I, F, F, P, P, P, I, P],
    "cmdiad_im2col_patch8": [P, I, I, P, P],
    "cmdiad_im2col3x3_bf16": [P, I, I, I, I, I, I, P, P],
    "cmdiad_vit_assemble": [P, P, P, I, I, I, P, P],
    "cmdiad_bilinear_up": [P, I, I, I, P, P],
    "cmdiad_ball_query": [P, P, P, I, I, I, F, I, P, P],
    "cmdiad_gather_points": [P, P, I, I, I, I, P, P],
    "cmdiad_blur8_maps": [P, I, I, I, F, P, P],
    "cmdiad_ocsvm_score_maps": [P, I, I, I, P, P, D, P, P],
    "cmdiad_ocsvm_fit": [P, I, I, D, I, D, I, U32, P, P, P, P, P, SZ, P],
    "cmdiad_linear3": [P, P, SZ, I, I, P, P],
    "cmdiad_cast_bf16": [P, SZ, P, P],
    "cmdiad_transpose_bf16": [P, I, I, P, P],
}
SIZE_QUERIES = {
    "cmdiad_gemm_streamk_workspace_bytes": [],
    "cmdiad_fps_workspace_bytes": [I, I],
    "cmdiad_reweight_workspace_bytes": [I, I],
    "cmdiad_reweight_pair_workspace_bytes": [I, I],
    "cmdiad_bank_block16_floats": [I, I],
    "cmdiad_coreset_workspace_bytes": [I, I, I],
    "cmdiad_coreset_f32_workspace_bytes": [I, I, I],
    "cmdiad_blur8_lds_bytes": [I, I],
    "cmdiad_ocsvm_fit_workspace_bytes": [I, I],
    "cmdiad_rows_dedup_workspace_bytes": [I],
    "cmdiad_knn_workspace_bytes": [I, I],
    "cmdiad_interp3nn_workspace_bytes": [I, I],
    "cmdiad_transformer_block_workspace_bytes": [I, I, I],
}


AB_SO_PATH = os.path.join(_HERE, "libcmdiad_hip_ab.so")   # test-only build with the superseded kernel formulations


def build(force=False, verbose=False, ab=False):
    """Compile cmdiad_amd/csrc into libcmdiad_hip.so with hipcc --offload-arch=gfx950 (ab=True: also the test-only
    libcmdiad_hip_ab.so, -DCMDIAD_AB_VARIANTS, that the variant parity tests and the A/B tools load via CMDIAD_HIP_LIB)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8"] + (["-B"] if force else []) + (["all", "ab"] if ab else [])
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:], res.stderr[-4000:])
    if res.returncode != 0:
        raise NativeError("building libcmdiad_hip.so failed")
    return SO_PATH


def lib():
    """Load the shared object (once).  Raises NativeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise NativeError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"(or `make -C cmdiad_amd/csrc`). cmdiad_amd has no CPU fallback.")
        L = ctypes.CDLL(SO_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        for name, args in SIZE_QUERIES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_size_t
        L.cmdiad_last_error.restype = c_char_p
        L.cmdiad_abi_version.restype = c_int
        L.cmdiad_has_ab_variants.restype = c_int
        _lib = L
    return _lib


def check(rc, name):
    if rc != 0:
        raise NativeError(f"{name} failed ({rc}): {lib().cmdiad_last_error().decode()}")
