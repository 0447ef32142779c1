"""ctypes binding of include/mvae_hip.h.  The HIP library is the only compute path: loading fails loudly."""
import ctypes as C
import os

from ._build import LIB_PATH

MVAE_ABI_VERSION = 4
MVAE_MAX_LEVELS = 16
MVAE_MAX_BLOCKS = 16
MVAE_NAME_CAP = 96
MVAE_COMM_ID_BYTES = 128
MVAE_OK, MVAE_E_INVALID, MVAE_E_STATE, MVAE_E_HIP, MVAE_E_NOMEM = 0, -1, -2, -3, -4
REG_NAMES = {0: None, 1: "l1", 2: "l2"}
ACT_DTYPES = {"f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}

_I32xL = C.c_int32 * MVAE_MAX_LEVELS
_I32xB = C.c_int32 * MVAE_MAX_BLOCKS


class MvaeConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("input_h", C.c_int32), ("input_w", C.c_int32), ("input_c", C.c_int32),
        ("levels", C.c_int32),
        ("z_dims", _I32xL),
        ("enc_n", C.c_int32), ("enc_filters", _I32xB),
        ("enc_kh", _I32xB), ("enc_kw", _I32xB), ("enc_sh", _I32xB), ("enc_sw", _I32xB),
        ("dec_n", C.c_int32), ("dec_filters", _I32xB),
        ("dec_kh", _I32xB), ("dec_kw", _I32xB), ("dec_sh", _I32xB), ("dec_sw", _I32xB),
        ("min_value", C.c_float), ("max_value", C.c_float), ("sample_std", C.c_float),
        ("max_batch", C.c_int32),
        ("act_dtype", C.c_int32),
    ]


class MvaeStepIO(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("batch", C.c_int32), ("training", C.c_int32),
        ("eps", C.c_void_p), ("noise", C.c_void_p), ("keep_mask", C.c_void_p),
        ("seed", C.c_uint64),
        ("recon", C.c_void_p), ("mu", C.c_void_p), ("log_var", C.c_void_p), ("z", C.c_void_p),
        ("losses", C.c_void_p),
    ]


# every symbol include/mvae_hip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "mvae_abi_version": (C.c_int, []),
    "mvae_debug_build": (C.c_int, []),
    "mvae_deterministic": (C.c_int, [_H]),
    "mvae_split_conv_status": (C.c_int, []),
    "mvae_split_conv_erratum": (C.c_int, []),
    "mvae_packed_f32_hazard": (C.c_int, [C.POINTER(C.c_int32)] * 2),
    "mvae_stamps": (C.c_int, [_H, C.POINTER(C.c_uint64), C.c_int32]),
    "mvae_fused_launch_stats": (C.c_int, [C.POINTER(C.c_int32)] * 3),
    "mvae_create": (C.c_int, [C.POINTER(MvaeConfig), C.POINTER(_H)]),
    "mvae_destroy": (None, [_H]),
    "mvae_last_error": (C.c_char_p, [_H]),
    "mvae_param_count": (C.c_int64, [_H]),
    "mvae_param_elems": (C.c_int64, [_H]),
    "mvae_param_info": (C.c_int, [_H, C.c_int64, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "mvae_state_count": (C.c_int64, [_H]),
    "mvae_state_elems": (C.c_int64, [_H]),
    "mvae_state_info": (C.c_int, [_H, C.c_int64, C.c_char_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mvae_latent_dim": (C.c_int64, [_H]),
    "mvae_reduce_elems": (C.c_int64, [_H]),
    "mvae_metrics_offset": (C.c_int64, [_H]),
    "mvae_workspace_bytes": (C.c_int64, [_H]),
    "mvae_bind": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "mvae_forward": (C.c_int, [_H, C.POINTER(MvaeStepIO), C.c_void_p]),
    "mvae_backward": (C.c_int, [_H, C.c_float, C.c_float, C.c_void_p]),
    "mvae_backward_phase": (C.c_int, [_H, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "mvae_reduce_split": (C.c_int64, [_H]),
    "mvae_graph_stats": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mvae_apply_adagrad": (C.c_int, [_H, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mvae_train_step": (C.c_int, [_H, C.POINTER(MvaeStepIO), C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mvae_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mvae_comm_init": (C.c_int, [_H, C.c_char_p, C.c_int32, C.c_int32]),
    "mvae_comm_destroy": (C.c_int, [_H]),
    "mvae_comm_size": (C.c_int, [_H]),
    "mvae_allreduce": (C.c_int, [_H, C.c_int64, C.c_int64, C.c_void_p]),
    "mvae_train_step_dp": (C.c_int, [_H, C.POINTER(MvaeStepIO), C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mvae_reg_loss": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "mvae_decode": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mvae_gather_rows": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "mvae_laplacian_split": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_void_p), C.c_void_p,
                                       C.c_void_p]),
    "mvae_laplacian_merge": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvae_laplacian_merge_mix": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_void_p), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mvae_mnv2_forward": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 10 + [C.c_void_p]),
    "mvae_mnv2_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 5 + [C.c_int32] * 5 + [C.c_void_p] * 11 + [C.c_void_p]),
    "mvae_resnet_forward": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p] * 9 + [C.c_void_p]),
    "mvae_resnet_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 8 + [C.c_void_p] * 11 + [C.c_void_p]),
    "mvae_conv2d_forward": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p] * 2 + [C.c_int32] * 6 + [C.c_void_p] * 2),
    "mvae_conv2d_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 4),
    "mvae_depthwise3x3_forward": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p] * 4),
    "mvae_depthwise3x3_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p] * 6),
    "mvae_activation_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "mvae_activation_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "mvae_eltwise": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mvae_scale_channels_forward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "mvae_scale_channels_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "mvae_global_maxpool_forward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "mvae_global_maxpool_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]),
    "mvae_maxpool_same_forward": (C.c_int, [C.c_int32, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p] * 3),
    "mvae_maxpool_same_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 8 + [C.c_void_p] * 2),
    "mvae_batchnorm_forward": (C.c_int, [C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32] + [C.c_void_p] * 7),
    "mvae_batchnorm_backward": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 5),
    "mvae_attention_core_forward": (C.c_int, [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 3),
    "mvae_attention_core_backward": (C.c_int, [C.c_int32] + [C.c_void_p] * 5 + [C.c_int32, C.c_int64, C.c_int32] + [C.c_void_p] * 5),
    "mvae_profile_enable": (C.c_int, [C.c_int32]),
    "mvae_profile_report": (C.c_int64, [C.c_char_p, C.c_int64]),
    "mvae_tensor_lookup": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "mvae_tensor_lookup2": (C.c_int, [_H, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "mvae_scale_dtype": (C.c_int, [_H, C.c_int32]),
}

_lib = None


def load_library(path=None):
    """dlopen libmvae_hip.so and type every entry point.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MVAE_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "libmvae_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the HIP path." % p)
    # The engine shares device memory, streams and RCCL with PyTorch, so both must run on ONE HIP runtime: torch is
    # imported first and libmvae_hip.so then binds to the libamdhip64 torch has already loaded.  Loaded the other way
    # round the process ends up with two runtimes and the second one finds no device (hipSetDevice: "no ROCm-capable
    # device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.mvae_abi_version() != MVAE_ABI_VERSION:
        raise RuntimeError("libmvae_hip.so ABI version %d != %d" % (lib.mvae_abi_version(), MVAE_ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


def make_config(input_dims, z_dims, encoder, decoder, min_value, max_value, sample_std, max_batch, act_dtype="f32"):
    cfg = MvaeConfig()
    if act_dtype not in ACT_DTYPES:
        raise ValueError("act_dtype must be one of %s" % sorted(ACT_DTYPES))
    cfg.act_dtype = ACT_DTYPES[act_dtype]
    cfg.abi_version = MVAE_ABI_VERSION
    cfg.input_h, cfg.input_w, cfg.input_c = [int(v) for v in input_dims]
    if len(z_dims) > MVAE_MAX_LEVELS:
        raise ValueError("at most %d levels are supported" % MVAE_MAX_LEVELS)
    cfg.levels = len(z_dims)
    for i, z in enumerate(z_dims):
        cfg.z_dims[i] = int(z)
    for pre, d in (("enc", encoder), ("dec", decoder)):
        n = len(d["filters"])
        if n > MVAE_MAX_BLOCKS:
            raise ValueError("at most %d encoder/decoder entries are supported" % MVAE_MAX_BLOCKS)
        setattr(cfg, pre + "_n", n)
        for i in range(n):
            getattr(cfg, pre + "_filters")[i] = int(d["filters"][i])
            getattr(cfg, pre + "_kh")[i] = int(d["kernel_size"][i][0])
            getattr(cfg, pre + "_kw")[i] = int(d["kernel_size"][i][1])
            getattr(cfg, pre + "_sh")[i] = int(d["strides"][i][0])
            getattr(cfg, pre + "_sw")[i] = int(d["strides"][i][1])
    cfg.min_value, cfg.max_value, cfg.sample_std = float(min_value), float(max_value), float(sample_std)
    cfg.max_batch = int(max_batch)
    return cfg
