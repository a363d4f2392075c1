"""ctypes binding of libgligen_amd.so (C ABI declared in include/gligen_amd.h).

The library is the only compute path: if it is missing or a call fails, this module raises —
there is no PyTorch/CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libgligen_amd.so"


class GligenAmdError(RuntimeError):
    pass


class UNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("model_channels", C.c_int),
        ("num_res_blocks", C.c_int), ("num_heads", C.c_int), ("context_dim", C.c_int),
        ("n_mult", C.c_int), ("channel_mult", C.c_int * 8),
        ("n_attn", C.c_int), ("attention_resolutions", C.c_int * 8),
        ("inpaint_mode", C.c_int), ("grounding_kind", C.c_int),
        ("gr_in_dim", C.c_int), ("gr_out_dim", C.c_int), ("max_persons", C.c_int), ("fuser_kind", C.c_int),
        ("extra_channels", C.c_int), ("tok_resize", C.c_int), ("tok_in_dim", C.c_int),
    ]


class VaeConfig(C.Structure):
    _fields_ = [
        ("ch", C.c_int), ("out_ch", C.c_int), ("z_channels", C.c_int), ("num_res_blocks", C.c_int),
        ("embed_dim", C.c_int), ("n_mult", C.c_int), ("ch_mult", C.c_int * 8), ("scale_factor", C.c_float),
    ]


class Grounding(C.Structure):
    _fields_ = [
        ("n", C.c_int),
        ("boxes", C.c_void_p), ("masks", C.c_void_p), ("text_masks", C.c_void_p), ("image_masks", C.c_void_p),
        ("text_embeddings", C.c_void_p), ("image_embeddings", C.c_void_p), ("points", C.c_void_p), ("tokens", C.c_void_p),
    ]


class PlmsArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint), ("B", C.c_int), ("h", C.c_int), ("w", C.c_int), ("n_steps", C.c_int),
        ("timesteps", C.POINTER(C.c_int64)), ("a_t", C.POINTER(C.c_float)), ("a_prev", C.POINTER(C.c_float)),
        ("fuser_scale", C.POINTER(C.c_float)), ("guidance_scale", C.c_float),
        ("x", C.c_void_p), ("inpaint_extra", C.c_void_p), ("mask", C.c_void_p), ("x0", C.c_void_p),
        ("noise", C.c_void_p), ("mask_B", C.c_int), ("x0_B", C.c_int), ("noise_B", C.c_int),
        ("sqrt_ac", C.POINTER(C.c_float)), ("sqrt_1mac", C.POINTER(C.c_float)),
        ("use_graph", C.c_int), ("sd_conv_w", C.c_void_p), ("sd_conv_b", C.c_void_p), ("sd_conv_step", C.c_int), ("ddim", C.c_int),
    ]


class TrainUNetIn(C.Structure):   # = gl_train_unet_in
    _fields_ = [(n, C.c_int) for n in ("B", "H", "W", "ctx_T", "Ng")] + \
               [(n, C.c_void_p) for n in ("x", "timesteps", "context", "boxes", "masks", "positive_embeddings", "target")] + \
               [("fuser_scale", C.c_float)] + [(n, C.c_void_p) for n in ("text_masks", "image_masks", "image_embeddings")] + [("checkpoint", C.c_int), ("use_weight_cache", C.c_int)]


class BoxCalibration(C.Structure):   # = gl_box_calibration
    _fields_ = [("hbm_copy_gbs", C.c_float), ("lds_dma_tbs", C.c_float), ("mfma_bf16_tflops", C.c_float)]


class MfmaCalibration(C.Structure):   # = gl_mfma_calibration
    _fields_ = [("tflops", C.c_float), ("sclk_mhz", C.c_float), ("ms", C.c_float), ("cycles_per_mfma", C.c_float)]


class ProfRec(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("calls", C.c_int), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/gligen_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I = C.c_int
SYMBOLS = {
    "gl_last_error": (C.c_char_p, []),
    "gl_ctx_create": (_I, [_I, C.c_size_t, C.POINTER(_P)]),
    "gl_ctx_fork": (_I, [_P, C.c_size_t, C.POINTER(_P)]),
    "gl_ctx_memory": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(_I)]),
    "gl_ctx_destroy": (_I, [_P]),
    "gl_unet_configure": (_I, [_P, C.POINTER(UNetConfig)]),
    "gl_vae_configure": (_I, [_P, C.POINTER(VaeConfig)]),
    "gl_weight_upload": (_I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64), _I]),
    "gl_finalize": (_I, [_P]),
    "gl_unet_set_cond": (_I, [_P, _I, _P, _I, C.POINTER(Grounding), _P]),
    "gl_unet_set_fuser_scale": (_I, [_P, C.c_float, _P]),
    "gl_unet_set_fuser_scales": (_I, [_P, C.POINTER(C.c_float), _I, _P]),
    "gl_unet_grounding_tokens": (_I, [_P, _P, _P]),
    "gl_op_spatial_tokens": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "gl_op_grounding_downsample": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
    "gl_unet_restore_first_conv": (_I, [_P, _P, _P, _P]),
    "gl_unet_forward": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P]),
    "gl_vae_decode": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "gl_vae_encode": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "gl_sample_plms": (_I, [_P, C.POINTER(PlmsArgs), _P]),
    "gl_sampler_timing": (_I, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_I)]),
    "gl_unet_profile": (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, C.POINTER(ProfRec), _I, C.POINTER(_I), _P]),
    "gl_to_uint8": (_I, [_P, _P, _I, _I, _I, _P]),
    "gl_arena_high_water": (_I, [_P, C.POINTER(C.c_size_t)]),
    "gl_launch_count": (_I, [_P, C.POINTER(C.c_int64)]),
    "gl_set_ff_rows_policy": (_I, [_I]),
    "gl_ff_rows_policy_report": (_I, [C.c_char_p, C.c_size_t]),
    "gl_box_calibrate": (_I, [_P, C.POINTER(BoxCalibration), _P]),
    "gl_mfma_calibrate": (_I, [_P, _I, _I, _I, _I, C.c_float, C.POINTER(MfmaCalibration), _P]),
    "gl_op_linear": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gl_op_geglu": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gl_op_ln_linear": (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, C.POINTER(_I), _P]),
    "gl_op_feedforward": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_I), _P]),
    "gl_train_block_param_names": (C.POINTER(C.c_char_p), []),
    "gl_op_block_train": (_I, [_P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_P), _P]),
    "gl_train_st_param_names": (C.POINTER(C.c_char_p), []),
    "gl_op_st_train": (_I, [_P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(_P), _P]),
    "gl_train_resblock_param_names": (C.POINTER(C.c_char_p), []),
    "gl_op_resblock_train": (_I, [_P, _P, C.POINTER(_P), _P, _P, _P, _P, _P, _P, _P]),
    "gl_op_resample_train": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gl_unet_train_step": (_I, [_P, C.POINTER(UNetConfig), C.POINTER(TrainUNetIn), _I, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(_P), _P, _P, _P]),
    "gl_train_wait_grads": (_I, [_P, _I, _P]),
    "gl_train_weight_cache": (_I, [_P, _I, C.POINTER(C.c_size_t)]),
    "gl_op_adamw_step": (_I, [_P, _P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _I, _P]),
    "gl_op_ff_chain": (_I, [_P, _P, _I, _I] + [_P] * 16),
    "gl_op_conv3x3": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "gl_op_groupnorm": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, C.c_float, _I, _P, _P]),
    "gl_op_gn_silu_conv3x3": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _P, _P, C.c_float, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    "gl_op_layernorm": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, C.c_float, _P, _P]),
    "gl_op_attention": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gl_op_ff_chain_q": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gl_op_proj_attention": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, C.POINTER(_I), _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise GligenAmdError(
            f"{LIB_PATH} not found: build it with `python -m gligen_amd.build` "
            "(gligen_amd has no fallback path without its HIP library)")
    # torch first: its wheel bundles its own libamdhip64.so. Loaded after ours (which resolves /opt/rocm's copy) the process ends up
    # with two HIP runtimes and the second one finds no device ("no HIP device available" from gl_context_create although
    # torch.cuda.is_available() is True) -- seen with build() called before the first `import torch` of the process.
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().gl_last_error()
        raise GligenAmdError(f"libgligen_amd error {rc}: {msg.decode() if msg else '?'}")
