"""ctypes binding of libbirefnet_hip.so (include/birefnet_hip.h).

This is the Python stand-in for the Rust shim of INTEGRATION.md: same entry points, same ownership rules.  There is no
CPU fallback: if the shared library is missing or cannot be loaded the import of this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BRN_LIB_PATH") or os.path.join(_HERE, "libbirefnet_hip.so")   # override: A/B runs of two builds

BRN_OK = 0
BRN_MEM_HOST, BRN_MEM_DEVICE = 0, 1
BRN_F32, BRN_F32_SPLIT3, BRN_F32_SPLIT2, BRN_BF16_OPERANDS, BRN_BF16, BRN_BF16_DEC_SPLIT2, BRN_F32_HALF2, BRN_F16 = 0, 1, 2, 3, 4, 5, 6, 7
BRN_DEFORM_REFERENCE_CPU, BRN_DEFORM_DEFORMABLE = 0, 1
BRN_ACT_NONE, BRN_ACT_RELU, BRN_ACT_GELU_ERF = 0, 1, 2

BRN_ERR_INVALID_ARG, BRN_ERR_MISSING_TENSOR, BRN_ERR_SHAPE, BRN_ERR_NO_DEVICE, BRN_ERR_HIP, BRN_ERR_OOM = 1, 2, 3, 4, 5, 6
ERR_NAMES = {1: "INVALID_ARG", 2: "MISSING_TENSOR", 3: "SHAPE", 4: "NO_DEVICE", 5: "HIP", 6: "OOM"}


class BrnError(RuntimeError):
    """The Python face of candle_core::Error::Msg coming out of the library."""

    def __init__(self, status, msg):
        super().__init__(f"[{ERR_NAMES.get(status, status)}] {msg}")
        self.status = status


class brn_config(C.Structure):
    _fields_ = [
        ("size_w", C.c_int), ("size_h", C.c_int),
        ("backbone", C.c_char * 32),
        ("backbone_channels", C.c_int * 4),
        ("mul_scl_ipt", C.c_int), ("ms_supervision", C.c_int), ("dec_ipt", C.c_int), ("use_aspp_deformable", C.c_int),
        ("cxt", C.c_int * 3), ("n_cxt", C.c_int),
        ("embed_dim", C.c_int), ("depths", C.c_int * 4), ("num_heads", C.c_int * 4), ("window_size", C.c_int),
        ("mlp_ratio", C.c_float), ("patch_size", C.c_int), ("in_channels", C.c_int), ("drop_path_rate", C.c_float),
        ("deform_mode", C.c_int),
    ]


class brn_named_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("shape", C.POINTER(C.c_int64)), ("ndim", C.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this package.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("brn_abi_version", C.c_int)
_sig("brn_last_error", C.c_char_p)
_sig("brn_build_info", C.c_char_p)
_sig("brn_device_count", C.c_int, C.POINTER(C.c_int))
_sig("brn_config_default_swin_l", None, C.POINTER(brn_config))
_sig("brn_config_lateral_channels", None, C.POINTER(brn_config), C.POINTER(C.c_int * 4))
_sig("brn_config_x4_channels", C.c_int, C.POINTER(brn_config))
_sig("brn_model_create", C.c_int, C.POINTER(brn_config), C.POINTER(brn_named_tensor), C.c_size_t, C.c_int, C.c_int,
     C.c_int, C.c_int, C.c_int, C.POINTER(_vp))
_sig("brn_model_create_from_safetensors", C.c_int, C.POINTER(brn_config), C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
     C.c_int, C.c_int, C.POINTER(_vp))
_sig("brn_decoder_create", C.c_int, C.POINTER(brn_config), C.POINTER(brn_named_tensor), C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.POINTER(_vp))
_sig("brn_model_destroy", None, _vp)
_sig("brn_forward_logits", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp)
_sig("brn_forward", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp)
_sig("brn_model_backbone_forward", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.c_int, _vp)
_sig("brn_model_squeeze_forward", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp)
_sig("brn_model_decoder_forward", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp)
_sig("brn_model_set_streams", C.c_int, _vp, C.c_int, C.c_int)
_sig("brn_model_set_profiling", C.c_int, _vp, C.c_int)
_sig("brn_model_last_timings", C.c_int, _vp, C.POINTER(C.c_float * 5))
_sig("brn_model_last_kernel_stats", C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double),
     C.POINTER(C.c_double), C.POINTER(C.c_int))
_sig("brn_kernel_family_name", C.c_char_p, C.c_int)
_sig("brn_swin_create", C.c_int, C.POINTER(brn_config), C.POINTER(brn_named_tensor), C.c_size_t, C.c_char_p, C.c_int,
     C.POINTER(_vp))
_sig("brn_swin_destroy", None, _vp)
_sig("brn_swin_forward", C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp), C.c_int, _vp)
_sig("brn_linear_forward", C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp)
_sig("brn_linear_residual_layer_norm_forward", C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_float, _vp, _vp, C.c_int, C.c_int, _vp)
_sig("brn_layer_norm_forward", C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, C.c_float, _vp, C.c_int, C.c_int, _vp)
_sig("brn_conv2d_forward", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
     C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_float, C.c_int, _vp, C.c_int, C.c_int, _vp)
_sig("brn_upsample_bilinear2d", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp)
_sig("brn_window_attention_forward", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
     _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp)
_sig("brn_patch_merging_forward", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp)
_sig("brn_deform_conv2d_forward", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int,
     C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp)

_sig("brn_aspp_deformable_forward", C.c_int, _vp, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp)

_sig("brn_decblk_forward", C.c_int, _vp, C.c_size_t, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int,
     C.c_int, _vp)

_sig("brn_preprocess_image", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp)
_sig("brn_postprocess_mask", C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp)
_sig("brn_infer_images_u8", C.c_int, _vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(_vp), _vp)
_sig("brn_set_op_compute", C.c_int, C.c_int)
if hasattr(lib, "brn_gemm_microbench"):   # only in libbirefnet_hip_diag.so (make diag; include/birefnet_hip_diag.h), reached through BRN_LIB_PATH
    _sig("brn_gemm_microbench", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float))

# every symbol include/birefnet_hip.h declares (checked by tests/test_abi.py against the header text)
DECLARED = [
    "brn_abi_version", "brn_last_error", "brn_build_info", "brn_device_count", "brn_config_default_swin_l",
    "brn_config_lateral_channels", "brn_config_x4_channels", "brn_model_create", "brn_model_create_from_safetensors", "brn_decoder_create", "brn_model_destroy", "brn_forward_logits",
    "brn_forward", "brn_model_backbone_forward", "brn_model_squeeze_forward", "brn_model_decoder_forward",
    "brn_model_set_streams", "brn_model_set_profiling", "brn_model_last_timings", "brn_model_last_kernel_stats", "brn_kernel_family_name",
    "brn_swin_create", "brn_swin_destroy", "brn_swin_forward", "brn_linear_forward", "brn_linear_residual_layer_norm_forward", "brn_layer_norm_forward",
    "brn_conv2d_forward", "brn_upsample_bilinear2d", "brn_window_attention_forward", "brn_patch_merging_forward",
    "brn_deform_conv2d_forward", "brn_aspp_deformable_forward", "brn_decblk_forward", "brn_set_op_compute", "brn_preprocess_image", "brn_postprocess_mask", "brn_infer_images_u8",
]


def check(status):
    if status != BRN_OK:
        raise BrnError(status, (lib.brn_last_error() or b"").decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    check(lib.brn_device_count(C.byref(n)))
    return n.value
