"""ctypes binding of libgdl_hip.so (the C ABI declared in include/gdl_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this
raises.  PyTorch is used only as the owner of device memory and streams.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("GDL_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "build", "libgdl_hip.so")  # GDL_LIB: tuning builds

GDL_F32, GDL_BF16 = 0, 1
GDL_AUDIO, GDL_VISUAL = 0, 1
GATHER_FWD, GATHER_DGRAD = 0, 1
ENC_NPARAMS, ENC_NBN = 60, 20

_C = {"i": ctypes.c_int, "p": ctypes.c_void_p, "z": ctypes.c_size_t, "f": ctypes.c_float, "d": ctypes.c_double,
      "l": ctypes.c_int64, "s": ctypes.c_char_p}

# name -> (restype code, argtypes codes); mirrors include/gdl_hip.h one to one
SIGNATURES = {
    "gdl_last_error": ("s", ""),
    "gdl_version": ("i", ""),
    "gdl_device_info": ("i", "ppi"),
    "gdl_conv_bn_tiles": ("i", "iiiiiiiiii"),
    "gdl_conv_table_bytes": ("z", "iiiiiiii"),
    "gdl_conv_build_table": ("i", "ii" + "iiiiiiiii" + "pp"),
    "gdl_conv_fwd": ("i", "ippppp" + "iiiiiiiii" + "p"),
    "gdl_conv_dgrad": ("i", "ippppp" + "iiiiiiiii" + "p"),
    "gdl_bn_act_bits": ("i", "ippp" + "ppp" + "pp" + "zi" + "p"),
    "gdl_conv_dgrad_relu": ("i", "ipppppp" + "iiiiiiiii" + "p"),
    "gdl_conv_fwd_bias": ("i", "ippppppp" + "iiiiiiiii" + "p"),
    "gdl_conv_dgrad_ds": ("i", "ippppppp" + "iiiii" + "p"),
    "gdl_conv_dgrad_bn_tiles": ("i", "iiiiiiiiii"),
    "gdl_conv_dgrad_bn": ("i", "ipppppp" + "iiiiiiiii" + "pppppppp" + "p"),
    "gdl_conv_split_workspace_bytes": ("z", "iiiiiiiiiii"),
    "gdl_conv_fwd_split": ("i", "ippppp" + "iiiiiiiii" + "pz" + "p"),
    "gdl_conv_dgrad_bn_split": ("i", "ipppppp" + "iiiiiiiii" + "pppppppp" + "pz" + "p"),
    "gdl_conv_dgrad_gelu": ("i", "ippppp" + "d" + "p" + "iiiiiiiii" + "p"),
    "gdl_acc_to_float": ("i", "pidpp"),
    "gdl_comm_unique_id": ("i", "p"),
    "gdl_comm_init": ("i", "piip"),
    "gdl_comm_world": ("i", "p"),
    "gdl_comm_allreduce_bucket": ("i", "ppz" + "p"),
    "gdl_comm_destroy": ("i", "p"),
    "gdl_head_concat_xy_fwd": ("i", "ppppppp" + "iiii" + "p"),
    "gdl_head_concat_xy_bwd": ("i", "pppppp" + "ii" + "pppp" + "iiii" + "p"),
    "gdl_swin_patch_gather": ("i", "ipp" + "iiiii" + "p"),
    "gdl_swin_bias_act": ("i", "ipppp" + "zii" + "p"),
    "gdl_swin_drop_path": ("i", "ipppp" + "zii" + "p"),
    "gdl_linear_bwd_ok": ("i", "izii"),
    "gdl_linear_bwd_workspace_bytes": ("z", "zii"),
    "gdl_linear_bwd": ("i", "ipppppp" + "pz" + "ziii" + "p"),
    "gdl_swin_ln_fwd": ("i", "ippppp" + "zii" + "p"),
    "gdl_swin_partial_bytes": ("z", "i"),
    "gdl_swin_ln_bwd": ("i", "ipppppppp" + "zii" + "p"),
    "gdl_swin_ln_bwd_colsum": ("i", "ipppppppp" + "zii" + "p"),
    "gdl_swin_colsum": ("i", "ipppp" + "zi" + "p"),
    "gdl_swin_ln_bwd_rows": ("i", "izi"),
    "gdl_swin_colsum_rows": ("i", "izi"),
    "gdl_swin_partial_reduce_batched": ("i", "pii" + "p"),
    "gdl_swin_attn_fwd": ("i", "ippp" + "iiiiiii" + "p"),
    "gdl_swin_attn_bwd_workspace_bytes": ("z", "iiiii"),
    "gdl_swin_attn_bwd": ("i", "ipppppp" + "iiiiiii" + "p"),
    "gdl_swin_merge": ("i", "ipp" + "iiiiii" + "p"),
    "gdl_swin_token_mean": ("i", "ipp" + "iiii" + "p"),
    "gdl_swin_token_mean_bwd": ("i", "ipp" + "iiii" + "p"),
    "gdl_swin_pack_matrix": ("i", "ippp" + "iiiiii" + "p"),
    "gdl_swin_pack_batched": ("i", "piii" + "p"),
    "gdl_swin_unpack_matrix": ("i", "pp" + "iiiiii" + "p"),
    "gdl_conv_wgrad_workspace_bytes": ("z", "iiiiiiiiii"),
    "gdl_conv_wgrad": ("i", "ipppp" + "iiiiiiiii" + "pzp"),
    "gdl_pack_weight": ("i", "ippp" + "iiii" + "p"),
    "gdl_stem_pad_bytes": ("z", "iiii"),
    "gdl_stem_weight_bytes": ("z", "i"),
    "gdl_stem_table_bytes": ("z", "iii"),
    "gdl_stem_pad": ("i", "ippiiiii" + "p"),
    "gdl_pack_stem_rows": ("i", "ippi" + "p"),
    "gdl_stem_build_table": ("i", "iiiip" + "p"),
    "gdl_stem_conv_bn_tiles": ("i", "iiii"),
    "gdl_stem_conv_fwd": ("i", "ippppp" + "iiii" + "p"),
    "gdl_stem_conv_wgrad_workspace_bytes": ("z", "iii"),
    "gdl_stem_conv_wgrad": ("i", "ipppp" + "iiii" + "pz" + "p"),
    "gdl_nhwc_to_nchw_f32": ("i", "ipp" + "iiii" + "p"),
    "gdl_nchw_f32_to_nhwc": ("i", "ipp" + "iiii" + "p"),
    "gdl_bn_stats_tiles": ("i", "i"),
    "gdl_bn_stats": ("i", "ippiip"),
    "gdl_bn_finalize_train": ("i", "pii" + "d" + "pp" + "ff" + "ppp" + "pppp" + "p"),
    "gdl_bn_finalize_eval": ("i", "ippf" + "pppp" + "p"),
    "gdl_bn_act": ("i", "ippp" + "ppp" + "i" + "p" + "zi" + "p"),
    "gdl_bn_bwd_blocks": ("i", "zi"),
    "gdl_bn_bwd_reduce": ("i", "ipppppp" + "i" + "p" + "zi" + "p"),
    "gdl_bn_bwd_finalize": ("i", "pii" + "d" + "ppp" + "p"),
    "gdl_bn_bwd_apply": ("i", "ipppppppp" + "i" + "p" + "zi" + "p"),
    "gdl_relu_bwd": ("i", "ipppzp"),
    "gdl_bn_relu_maxpool_fwd": ("i", "ipppppp" + "iiii" + "p"),
    "gdl_maxpool_bn_bwd_apply": ("i", "ippppppppp" + "p" + "iiii" + "p"),
    "gdl_maxpool_bwd": ("i", "ippp" + "iiii" + "p"),
    "gdl_avgpool_fwd": ("i", "ipp" + "iiii" + "p"),
    "gdl_avgpool_bwd": ("i", "ipp" + "iiii" + "p"),
    "gdl_head_uni_dfeat": ("i", "pp" + "i" + "pp" + "f" + "p" + "ii" + "p"),
    "gdl_head_uni_dfeat_w": ("i", "pp" + "i" + "pp" + "f" + "p" + "iii" + "p"),
    "gdl_head_concat_fwd": ("i", "ppppppp" + "ii" + "p"),
    "gdl_head_concat_bwd": ("i", "pppppp" + "ii" + "pppp" + "ii" + "p"),
    "gdl_softmax_ce": ("i", "ppf" + "pp" + "ii" + "p"),
    "gdl_softmax_ce3": ("i", "pppp" + "fff" + "pppp" + "ii" + "p"),
    "gdl_head_sum_fwd": ("i", "ppppppppp" + "ii" + "p"),
    "gdl_head_sum_bwd": ("i", "ppppppp" + "ii" + "pppppp" + "ii" + "p"),
    "gdl_head_gated_fwd": ("i", "p" * 13 + "ii" + "p"),
    "gdl_head_gated_bwd": ("i", "p" * 10 + "i" + "p" * 9 + "ii" + "p"),
    "gdl_head_film_workspace_bytes": ("z", "i"),
    "gdl_head_film_fwd": ("i", "p" * 10 + "ii" + "pz" + "p"),
    "gdl_head_film_bwd": ("i", "p" * 8 + "i" + "p" * 6 + "ii" + "pz" + "p"),
    "gdl_logspec_frames": ("i", "ii"),
    "gdl_logspec": ("i", "p" + "iiiii" + "pp"),
    "gdl_frames_normalize": ("i", "p" + "lii" + "ppp" + "p"),
    "gdl_eval_count": ("i", "pppp" + "ii" + "pppp" + "p"),
    "gdl_optim_create": ("i", "pppi"),
    "gdl_optim_destroy": (None, "p"),
    "gdl_optim_workspace_bytes": ("z", "p"),
    "gdl_optim_stats_len": ("i", "p"),
    "gdl_optim_bind_workspace": ("i", "ppzp"),
    "gdl_optim_grad_stats": ("i", "ppffp" + "pzp"),
    "gdl_optim_sgd_step": ("i", "ppppp" + "ffff" + "p"),
    "gdl_encoder_create": ("i", "piiiiii"),
    "gdl_encoder_destroy": (None, "p"),
    "gdl_encoder_side_stream": ("i", "pi"),
    "gdl_encoder_borrow_side_stream": ("i", "pp"),
    "gdl_encoder_backward_phase": ("i", "pippp" + "p"),
    "gdl_encoder_workspace_bytes": ("z", "p"),
    "gdl_encoder_param_numel": ("i", "pp"),
    "gdl_encoder_out_shape": ("i", "pppp"),
    "gdl_encoder_bind": ("i", "ppz"),
    "gdl_encoder_set_params": ("i", "ppppp"),
    "gdl_encoder_forward": ("i", "ppippp"),
    "gdl_encoder_backward": ("i", "ppppp"),
    "gdl_encoder_forward_serial": ("l", "p"),
    "gdl_encoder_bn_overflow": ("i", "pp"),
    "gdl_prof_enable": ("i", "i"),
    "gdl_prof_enabled": ("i", ""),
    "gdl_prof_set_filter": ("i", "s"),
    "gdl_debug_timing_buffer": ("i", "p"),
    "gdl_prof_nslots": ("i", ""),
    "gdl_prof_slot_name": ("s", "i"),
    "gdl_prof_slot_bound": ("i", "i"),
    "gdl_prof_collect": ("i", "ppp"),
    "gdl_prof_set_peaks": ("i", "dd"),
    "gdl_prof_collect_floor": ("i", "pp"),
    "gdl_prof_timeline": ("i", "ipppppp"),
    "gdl_stem_bwd_fused_ok": ("i", "ii"),
    "gdl_stem_bwd_fused": ("i", "ipppppppppp" + "p" + "iiii" + "pz" + "p"),
}

_lib = None


class GdlError(RuntimeError):
    pass


def load():
    """Load libgdl_hip.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise GdlError(f"libgdl_hip.so not found at {SO_PATH}: build it with `make -C iccv2025-gdl_amd/csrc` "
                       "(or __graft_entry__.build()); there is no fallback path")
    # ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and the device memory / streams this binding passes
    # around are PyTorch's.  Imported first, the library's dependency resolves to that copy; loaded the other way round (the
    # library before torch, e.g. __graft_entry__.build() followed by smoke() in one process) the library's first HIP call
    # fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401

    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = _C[res] if res else None
        fn.argtypes = [_C[a] for a in args]
    _lib = lib
    return lib


def last_error():
    return load().gdl_last_error().decode()


def check(rc, what=""):
    if rc != 0:
        raise GdlError(f"{what}: {last_error()} (code {rc})")


def call(name, *args):
    """Call an int-returning entry point and raise GdlError on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise GdlError(f"{name}: {last_error()} (code {rc})")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "gdl: tensor must be contiguous"
    return t.data_ptr()


def cur_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream


def dtype_code(name):
    if name in ("bf16", "bfloat16", GDL_BF16):
        return GDL_BF16
    if name in ("f32", "fp32", "float32", GDL_F32):
        return GDL_F32
    raise ValueError(f"gdl: unknown dtype {name!r} (use 'bf16' or 'f32')")


def torch_dtype(code):
    import torch

    return torch.bfloat16 if code == GDL_BF16 else torch.float32
