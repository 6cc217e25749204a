"""ctypes binding of libmv2d_hip.so (the C-ABI declared in include/mv2d_hip.h).

The product path has NO fallback: if the library is missing or was built for a different arch, loading
raises (``Mv2dHipError``) instead of silently running something else.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libmv2d_hip.so')


class Mv2dHipError(RuntimeError):
    pass


P, I, LL, F, D = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
ABI_VERSION = 6                  # include/mv2d_hip.h: mv2d_abi_version()

class TdDims(C.Structure):
    """struct mv2d_td_dims (include/mv2d_hip.h): the scalar arguments of mv2d_train_decoder_fwd / _bwd"""
    _fields_ = [('T', I), ('S', I), ('L', I), ('F', I), ('sa_nnz', I), ('ca_nnz', I), ('p_sa_attn', F), ('p_sa_out', F), ('p_ca_attn', F),
                ('p_ca_out', F), ('p_ffn_act', F), ('p_ffn_out', F), ('seed', C.c_uint), ('eps', F), ('pad', I), ('nk', I)]


class ThDims(C.Structure):
    """struct mv2d_th_dims (include/mv2d_hip.h): the scalar arguments of mv2d_train_heads_fwd / _bwd"""
    _fields_ = [('T', I), ('L', I), ('NC', I), ('eps', F)]


# name -> (restype, argtypes) — mirrors include/mv2d_hip.h one to one
SIGNATURES = {
    'mv2d_last_error': (C.c_char_p, []),
    'mv2d_abi_version': (I, []),
    'mv2d_device_arch': (I, [C.c_char_p, I]),
    'mv2d_spin': (I, [I, P]),
    'mv2d_gemm_bf16': (I, [P, P, I, I, P, P, I, I, I, I, P, I, P, I, P, I, P, I, I, LL, I, P, P, I, I, P]),
    'mv2d_gemm_bf16_ex': (I, [P, P, I, I, P, P, I, I, I, I, P, I, P, I, P, I, P, I, I, LL, I, P, P, I, I, I, P, I, I, LL, P]),
    'mv2d_split3_rows': (I, [P, P, P, I, I, P, P]),
    'mv2d_pe_fused_tab': (I, [P, P, P, P, P, I] + [P] * 9 + [I, P, P, P]),
    'mv2d_pe_fused_tab2': (I, [P, P, P, P, P, I] + [P] * 9 + [I, P, P, I, P]),
    'mv2d_pe_fused_x3': (I, [P, P, P, P, I] + [P] * 13 + [I] + [P] * 5 + [I, I, P, P]),
    'mv2d_pe_fused_x3b': (I, [P, P, P, P, I] + [P] * 13 + [I] + [P] * 5 + [I, I, P, P]),
    'mv2d_key16_format': (I, []),
    'mv2d_f32_to_key16': (I, [P, P, P, LL, P]),
    'mv2d_split_rows_key16': (I, [P, P, P, P, I, I, P, P]),
    'mv2d_qg_conv_pool': (I, [P, P, P, P, I, I, P]),
    'mv2d_qg_conv_pool_x3': (I, [P, P, P, P, P, P, I, I, P]),
    'mv2d_pack_wfrag_bf16': (I, [P, P, I, I, P]),
    'mv2d_gemm_f32': (I, [P, P, I, P, P, I, I, I, I, I, I, I, F, F, P, I, I, LL, I, LL, LL, LL, LL, P]),
    'mv2d_attn_out_fused': (I, [P, P, P, P, P, P, P, P, P, P, F, P, I, F, P]),
    'mv2d_sa_block_fused_x3': (I, [P, P, P, P, P, P, P, P, P, P, P, P, F, P, I, F, P]),
    'mv2d_query_embed_fused_x3': (I, [P] * 17 + [I, P]),
    'mv2d_ffn_out_fused_x3': (I, [P, I, LL, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, F, P]),
    'mv2d_attn_out_fused_x3': (I, [P, P, P, P, P, P, P, P, P, P, P, P, F, P, I, F, P]),
    'mv2d_pack_wfrag_f32': (I, [P, P, I, I, I, P]),
    'mv2d_heads_fused': (I, [P, P, P, P, P, P, I, I, F, P, F, P, P]),
    'mv2d_linear_x3': (I, [P, P, I, I, P, P, P, P, I, I, I, I, I, F, I, LL, LL, LL, LL, P]),
    'mv2d_linear_x3_ex': (I, [P, P, I, I, P, P, P, P, I, I, I, I, I, F, I, LL, LL, LL, LL, P, I, P, P, I, P]),
    'mv2d_heads_fused_x3': (I, [P, P, P, P, P, P, I, I, F, P, F, P, P]),
    'mv2d_ffn_fused': (I, [P, P, P, P, P, I, I, P]),
    'mv2d_ffn_pack_weights': (I, [P, P, P, P, I, P]),
    'mv2d_ffn_fused_x3': (I, [P, P, P, P, P, P, P, I, I, I, P]),
    'mv2d_split_bf16x2': (I, [P, P, P, LL, P]),
    'mv2d_split_q16x2': (I, [P, P, P, LL, P]),
    'mv2d_q16_format': (I, []),
    'mv2d_row_ln': (I, [P, I, LL, P, P, P, P, I, P, P, P, P, P, P, I, F, I, P]),
    'mv2d_finalize_reg': (I, [P, P, I, I, P, F, P]),
    'mv2d_avgpool49': (I, [P, P, I, I, P]),
    'mv2d_f32_to_bf16': (I, [P, P, LL, P]),
    'mv2d_nchw_to_nhwc': (I, [P, P, I, I, I, P]),
    'mv2d_nchw_to_nhwc_masked': (I, [P, P, P, I, I, I, P]),
    'mv2d_nchw_to_nhwc_bf16': (I, [P, P, I, I, I, P]),
    'mv2d_map_conv3x3': (I, [P, P, P, P, I, I, I, P]),
    'mv2d_self_attn_fwd': (I, [P, P, I, P, I, P]),
    'mv2d_self_attn_dn_fwd': (I, [P, P, I, I, I, P]),
    'mv2d_self_attn_x3_fwd': (I, [P, P, I, P, I, I, I, I, P]),
    'mv2d_dn_queries': (I, [P, P, P, I, I, F, F, F, I, P, F, P, P, P, P]),
    'mv2d_sparse_xattn_fwd': (I, [P, P, P, P, P, P, P, LL, I, I, P]),
    'mv2d_sparse_xattn_bwd': (I, [P] * 14 + [I, I, P]),
    'mv2d_sparse_xattn_fwd_drop': (I, [P, P, P, P, P, P, P, LL, I, I, F, C.c_uint, P]),
    'mv2d_sparse_xattn_bwd_drop': (I, [P] * 14 + [I, I, F, C.c_uint, P]),
    'mv2d_sparse_xattn_bwd_ex': (I, [P] * 14 + [I, I, F, C.c_uint, I, F, P]),
    'mv2d_dgrad_relu_f32x3': (I, [P, P, P, F, P, I, I, I, P]),
    'mv2d_xattn_qmap': (I, [P, P, P, P, I, P]),
    'mv2d_attn_out_qmap_x3': (I, [P] * 12 + [F, P, P, P, I, F, P]),
    'mv2d_attn_out_zmap_x3': (I, [P, P, P, P, P, I, P, P, P, P, P, P, P, I, F, P]),
    'mv2d_xattn_tile_fwd': (I, [P, P, P, P, P, P, P, P, P, LL, I, I, I, P]),
    'mv2d_xattn_tile_fwd_ordered': (I, [P, P, P, P, P, P, P, P, P, LL, I, I, I, P, I, P]),
    'mv2d_xattn_fused_fwd': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, I, P]),
    'mv2d_xattn_group_max': (I, [I, I]),
    'mv2d_xattn_group_tables': (I, [P, P, P, P, I, I, I, P, P, P, P, P, P, I, P, P, P]),
    'mv2d_xattn_group_fwd': (I, [P] * 19 + [I, I, P]),
    'mv2d_xattn_query_order': (I, [P, P, P, I, I, P, P, I, P]),
    'mv2d_xattn_ctxmap': (I, [P, P, P, P, P, P, I, I, P]),
    'mv2d_box_params': (I, [P, P, P, P, P, I, P, I, F, F, F, P]),
    'mv2d_refpoint_posemb': (I, [P, I, P, P, P, P, P, I, P, P]),
    'mv2d_lidar2img_inverse': (I, [P, P, P, I, P]),
    'mv2d_posemb3d': (I, [P, P, P, I, P]),
    'mv2d_roi_align': (I, [P, P, P, P, P, P, P, I, I, I, I, F, I, P, I, P]),
    'mv2d_roi_align_ex': (I, [P, P, P, P, P, P, P, I, I, I, I, F, I, P, I, P, P, P, P, P, P]),
    'mv2d_box_correlation': (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, F, F, F, I, P]),
    'mv2d_csr_workspace_bytes': (LL, [I, I, I, I]),
    'mv2d_mask_compact': (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, F, F, I, P]),
    'mv2d_roi_positions': (I, [P, P, P, P, P, P, P, I, I, I, I, F, F, P]),
    'mv2d_csr_from_corr': (I, [P, P, P, P, I, I, I, P]),
    'mv2d_roi_positions_csr': (I, [P, P, P, P, P, P, P, I, I, I, I, F, F, P, P, P, P, I, I, P, I, P, P, P]),
    'mv2d_frame_geometry': (I, [P, P, P, P, P, I, P, F, F, F, P, P, P, P, P, I, I, I, I, I, I, I, F, F, F, I, P, LL, P]),
    'mv2d_pe_inputs': (I, [P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P, P]),
    'mv2d_pe_frustum_f32': (I, [P, P, I, P, P, P, P, P, I, I, I, I, P, P]),
    'mv2d_result_pack': (I, [P, P, P, P, F, I, P, P, P, P, I, I, P]),
    'mv2d_nms_bev': (I, [P, P, P, P, F, P, I, I, P]),
    'mv2d_pack_detections': (I, [P, P, P, P, P, I, I, I, P]),
    'mv2d_roi_align_bwd': (I, [P, P, P, P, I, I, I, I, F, I, P]),
    'mv2d_colsum_scratch_rows': (I, [I]),
    'mv2d_colsum': (I, [P, LL, I, I, P, P, P]),
    'mv2d_gemm_f32x3_ws_bytes': (LL, [I, I, I]),
    'mv2d_gemm_f32x3': (I, [P, LL, I, P, LL, I, P, I, P, LL, I, I, I, P, LL, P]),
    'mv2d_gemm_f32x3_ex': (I, [P, LL, I, P, LL, I, P, I, F, I, I, P, LL, I, I, I, P, LL, P]),
    'mv2d_colsum_add': (I, [P, LL, I, I, P, P, P, P]),
    'mv2d_softmax_bwd_rows': (I, [P, P, P, LL, I, I, F, P]),
    'mv2d_gemm_f32x3_batched_ws_bytes': (LL, [I, I, I, I]),
    'mv2d_gemm_f32x3_batched': (I, [P, LL, LL, I, P, LL, LL, I, P, LL, LL, I, I, I, I, F, P, LL, P]),
    'mv2d_wgrad_f32x3': (I, [P, P, P, P, I, I, I, P, LL, P, P]),
    'mv2d_train_decoder_act_bytes': (LL, [P]),
    'mv2d_train_decoder_ws_bytes': (LL, [P, I]),
    'mv2d_train_decoder_fwd': (I, [P] * 14),
    'mv2d_train_decoder_bwd': (I, [P] * 24),
    'mv2d_box_code_fwd': (I, [P, P, P, I, I, I, F, P, P]),
    'mv2d_box_code_bwd': (I, [P, P, P, P, P, I, I, I, F, P, P]),
    'mv2d_im2col3x3': (I, [P, P, I, P]),
    'mv2d_col2im3x3': (I, [P, P, I, P]),
    'mv2d_center2lidar_fwd': (I, [P, P, P, I, P, P]),
    'mv2d_center2lidar_bwd': (I, [P, P, P, P, I, P, P]),
    'mv2d_dense_attn_ws_bytes': (LL, [I, I, I]),
    'mv2d_dense_attn_fwd': (I, [P, P, P, I, I, F, C.c_uint, P, P, P, P]),
    'mv2d_dense_attn_bwd': (I, [P, P, P, P, P, P, I, I, F, C.c_uint, F, P, P, P, P, P]),
    'mv2d_dense_attn_bwd_parts': (I, [P, P, P, P, P, P, I, I, F, C.c_uint, F, P, P, P, P, I, P]),
    'mv2d_lsap_layers': (I, [P, I, I, I, P, I]),
    'mv2d_train_heads_act_bytes': (LL, [P]),
    'mv2d_train_heads_ws_bytes': (LL, [P, I]),
    'mv2d_train_heads_fwd': (I, [P] * 8),
    'mv2d_train_heads_bwd': (I, [P] * 10),
    'mv2d_linear_bwd_x3_ws_bytes': (LL, [I, I, I]),
    'mv2d_linear_bwd_x3': (I, [P, P, P, P, P, P, P, I, I, I, P, LL, P]),
    'mv2d_layer_norm_bwd_blocks': (I, [I]),
    'mv2d_layer_norm_bwd': (I, [P, P, P, P, P, P, P, P, I, F, P]),
    'mv2d_match_cost': (I, [P, P, P, P, P, I, I, I, I, F, F, F, F, P]),
    'mv2d_set_loss': (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, F, F, F, F, F, F, I, P]),
    'mv2d_decode_topk': (I, [P, P, I, I, I, P, P, P, P, P, P, P, P, I, I, P, P]),
}

_lib = None


def load(path=None):
    """Load (once) and return the ctypes library with typed entry points.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get('MV2D_HIP_LIB', LIB_PATH)
    if not os.path.exists(path):
        raise Mv2dHipError(
            f'{path} not found: the MV2D HIP extension is not built.  Run `python -m mv2d_amd.build` '
            '(or __graft_entry__.build()).  There is no CPU/PyTorch fallback for the product path.')
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise Mv2dHipError(f'{path} does not export {name}; rebuild with `python -m mv2d_amd.build --force`')
        fn.restype = res
        fn.argtypes = args
    if lib.mv2d_abi_version() != ABI_VERSION:
        raise Mv2dHipError(f'{path} implements ABI version {lib.mv2d_abi_version()}, this package needs {ABI_VERSION}: rebuild with `python -m mv2d_amd.build --force`')
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mv2d_last_error()
        raise Mv2dHipError(f'{what} failed (rc={rc}): {msg.decode() if msg else ""}')
