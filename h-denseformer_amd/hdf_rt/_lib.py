"""ctypes binding of libhdf_hip.so (C ABI declared in include/hdf.h).

The product path has NO fallback: if the shared library is missing or a call fails, this module
raises.  Build it with `python h-denseformer_amd/build.py` (or __graft_entry__.build())."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HDF_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libhdf_hip.so")

F32, BF16, F16 = 0, 1, 2

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

_PROTOS = {
    "hdf_version": (C.c_char_p, []),
    "hdf_last_error": (C.c_char_p, []),
    "hdf_set_cu_budget": (_i, [_i]),
    "hdf_plan_create": (_i, [_i] * 8 + [C.POINTER(_vp)]),
    "hdf_plan_create_2d": (_i, [_i] * 7 + [C.POINTER(_vp)]),
    "hdf_plan_create_2d_embedded": (_i, [_i] * 7 + [C.POINTER(_vp)]),
    "hdf_plan_destroy": (None, [_vp]),
    "hdf_plan_num_params": (_i64, [_vp]),
    "hdf_plan_param_floats": (_i64, [_vp]),
    "hdf_plan_param_info": (_i, [_vp, _i64, C.c_char_p, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i),
                                 C.POINTER(_i64)]),
    "hdf_plan_workspace_bytes": (_i64, [_vp, _i]),
    "hdf_plan_inference_workspace_bytes": (_i64, [_vp, _i]),
    "hdf_plan_buffer_info": (_i, [_vp, _i, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i),
                                  C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "hdf_plan_region_info": (_i, [_vp, _i, C.c_char_p, C.POINTER(_i64), C.POINTER(_i64)]),
    "hdf_forward": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _u64, _vp]),
    "hdf_backward": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "hdf_backward_stages": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hdf_backward_events": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.POINTER(_vp)]),
    "hdf_plan_grad_bucket": (_i, [_vp, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "hdf_stream_wait_event": (_i, [_vp, _vp]),
    "hdf_plan_set_probe": (_i, [_vp, _vp, _vp]),
    "hdf_plan_set_chain_timeout_us": (_i, [_vp, _i64]),
    "hdf_plan_chain_state": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "hdf_plan_force_persistent": (_i, [_vp, _i]),
    "hdf_op_occupy": (_i, [_i, _i, _i, _i, _vp]),
    "hdf_loss_workspace_bytes": (_i64, [_i]),
    "hdf_loss_forward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "hdf_loss_backward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp]),
    "hdf_loss_terms_forward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "hdf_loss_terms_backward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp]),
    "hdf_loss_weighted_forward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _vp, _vp,
                                       _vp]),
    "hdf_loss_weighted_backward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _vp]),
    "hdf_dice_counts": (_i, [_i, _vp, _vp, _i, _i, _i64, _vp, _vp]),
    "hdf_confusion_matrix": (_i, [_i, _vp, _vp, _i, _i, _i64, _vp, _i, _vp]),
    "hdf_confusion_matrix_labels": (_i, [_vp, _vp, _i, _i64, _vp, _i, _vp]),
    "hdf_normalize_workspace_bytes": (_i64, [_i]),
    "hdf_normalize_mr": (_i, [_vp, _i, _i64, _vp, _vp]),
    "hdf_normalize_petct": (_i, [_vp, _i, _i64, _f, _f, _vp, _vp]),
    "hdf_sw_accumulate": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "hdf_sw_finalize": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "hdf_onehot_from_labels": (_i, [_vp, _vp, _i, _i, _i64, _vp]),
    "hdf_adam_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "hdf_op_to_channels_last": (_i, [_i, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "hdf_op_pack_weights": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i, _vp]),
    "hdf_op_conv3d": (_i, [_i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp, _i,
                           _vp]),
    "hdf_op_conv3d_first": (_i, [_i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "hdf_op_conv3d_first_wgrad": (_i, [_i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp]),
    "hdf_op_conv3d_first_wgrad_in": (_i, [_i, _vp, _i64, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i,
                                          _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp]),
    "hdf_op_conv3d_bwd_stats": (_i, [_i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _vp, _vp, _vp,
                                     _vp, _vp]),
    "hdf_op_conv3d_wr": (_i, [_i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i64, _i, _vp, _i, _vp]),
    "hdf_op_conv3d_stat_tiles": (_i, [_i, _i, _i, _i, _i]),
    "hdf_op_conv3d_split": (_i, [_i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i, _vp]),
    "hdf_op_in_bwd_workspace_floats": (_i64, [_i, _i, _i64]),
    "hdf_op_in_bwd": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i64, _vp,
                           _vp]),
    "hdf_op_wgrad_workspace_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _i]),
    "hdf_op_in_bwd_wgrad": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i,
                                 _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp]),
    "hdf_op_conv3d_wgrad": (_i, [_i, _i, _vp, _i64, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i,
                                 _vp, _i, _i, _i, _vp, _i64, _vp]),
    "hdf_op_in_finalize": (_i, [_vp, _i, _i, _i, _i, _i64, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "hdf_op_norm_relu_add": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i64, _vp]),
    "hdf_op_maxpool_fwd": (_i, [_i, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_maxpool_bwd_in_rows": (_i, [_i, _i, _i, _i]),
    "hdf_op_maxpool_bwd_in": (_i, [_i, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_enc_tail": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_enc_tail_up": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i,
                                _vp]),
    "hdf_op_maxpool_bwd": (_i, [_i, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_upsample_fwd": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_upsample_bwd": (_i, [_i, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "hdf_op_attention_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "hdf_op_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hdf_op_attention_amp_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "hdf_op_patch_embed_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i64, _vp, _i, _u64, _vp]),
    "hdf_op_patch_embed_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _vp, _i, _u64, _vp]),
    "hdf_op_dense_layer_fwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _i, _u64, _vp]),
    "hdf_op_dense_layer_bwd": (_i, [_i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _u64, _vp]),
    "hdf_op_block_out_fwd": (_i, [_i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp, _i, _i, _u64, _vp]),
    "hdf_op_block_out_bwd": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _vp, _vp, _i, _vp, _i, _u64, _vp]),
    "hdf_op_head_fwd": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp]),
    "hdf_op_head_bwd": (_i, [_i, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _i64, _vp]),
}

EXPORTS = tuple(_PROTOS.keys())

_lib = None


class HdfError(RuntimeError):
    pass


def lib():
    """Load the shared library once.  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (it owns the
        # device memory and streams we are handed), so make sure it is loaded BEFORE libhdf_hip.so pulls in
        # a second copy from /opt/rocm -- two runtimes in one process cannot see each other's devices.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise HdfError(f"{LIB_PATH} not found: build the HIP extension first "
                           f"(python h-denseformer_amd/build.py); there is no CPU/eager fallback")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if os.environ.get("HDF_LIB_PATH"):      # A/B runs against an older build: calling it will raise
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise HdfError(f"{what} failed (rc={rc}): {lib().hdf_last_error().decode(errors='replace')}")


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL)"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
