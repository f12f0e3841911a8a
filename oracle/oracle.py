"""ctypes loader of oracle/liboracle.so — the CPU ORACLE (test infrastructure; see the header of brn_oracle.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product never does."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "brn_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


class _Cfg(C.Structure):   # same layout as brn_config (include/birefnet_hip.h)
    _fields_ = [
        ("size_w", C.c_int), ("size_h", C.c_int), ("backbone", C.c_char * 32), ("backbone_channels", C.c_int * 4),
        ("mul_scl_ipt", C.c_int), ("ms_supervision", C.c_int), ("dec_ipt", C.c_int), ("use_aspp_deformable", C.c_int),
        ("cxt", C.c_int * 3), ("n_cxt", C.c_int), ("embed_dim", C.c_int), ("depths", C.c_int * 4), ("num_heads", C.c_int * 4),
        ("window_size", C.c_int), ("mlp_ratio", C.c_float), ("patch_size", C.c_int), ("in_channels", C.c_int),
        ("drop_path_rate", C.c_float), ("deform_mode", C.c_int),
    ]


class _NT(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("shape", C.POINTER(C.c_int64)), ("ndim", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_last_error.restype = C.c_char_p
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("oracle: " + lib().orc_last_error().decode())


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_cfg(depths=(2, 2, 18, 2), deform_mode=0, embed_dim=192, num_heads=(6, 12, 24, 48), window_size=12, patch_size=4, in_channels=3):
    c = _Cfg()
    c.size_w = c.size_h = 1024
    c.backbone = b"swin_v1_l"
    for i, v in enumerate((192, 384, 768, 1536)):
        c.backbone_channels[i] = v
    c.mul_scl_ipt = c.ms_supervision = c.dec_ipt = c.use_aspp_deformable = 1
    for i, v in enumerate((192, 384, 768)):
        c.cxt[i] = v
    c.n_cxt = 3
    c.embed_dim = embed_dim
    for i in range(4):
        c.depths[i] = int(depths[i])
        c.num_heads[i] = int(num_heads[i])
    c.window_size, c.mlp_ratio, c.patch_size, c.in_channels, c.drop_path_rate = window_size, 4.0, patch_size, in_channels, 0.2
    c.deform_mode = int(deform_mode)
    return c


def cfg_from(birefnet_config):
    """from a candle_birefnet_amd.BiRefNetConfig-shaped object (duck-typed: no import of the product)."""
    s = birefnet_config.swin
    return make_cfg(s.depths, {"reference_cpu": 0, "deformable": 1}[birefnet_config.deform_mode], s.embed_dim, s.num_heads,
                    s.window_size, s.patch_size, s.in_channels)


def named(tensors):
    n = len(tensors)
    arr = (_NT * n)()
    keep = []
    for i, (name, a) in enumerate(tensors.items()):
        a = _f(a)
        shp = (C.c_int64 * a.ndim)(*a.shape)
        nm = name.encode()
        keep += [a, shp, nm]
        arr[i].name, arr[i].data, arr[i].shape, arr[i].ndim = nm, a.ctypes.data_as(C.POINTER(C.c_float)), shp, a.ndim
    return arr, n, keep


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def forward_logits(cfg, weights, x):
    x = _f(x)
    B, _, H, W = x.shape
    arr, n, keep = named(weights)
    out = np.empty((B, 1, H, W), np.float32)
    _chk(lib().orc_forward_logits(C.byref(cfg), arr, C.c_size_t(n), _p(x), B, H, W, _p(out)))
    return out


def forward_parts(cfg, weights, x):
    x = _f(x)
    B, _, H, W = x.shape
    arr, n, keep = named(weights)
    out = np.empty((B, 1, H, W), np.float32)
    hs = [-(-H // 4)]
    ws = [-(-W // 4)]
    for _ in range(3):
        hs.append((hs[-1] + 1) // 2); ws.append((ws[-1] + 1) // 2)
    f = [np.empty((B, 192 << i, hs[i], ws[i]), np.float32) for i in range(4)]
    fp = (C.c_void_p * 4)(*[a.ctypes.data for a in f])
    x1 = np.empty((B, 384, H // 4, W // 4), np.float32); x2 = np.empty((B, 768, H // 8, W // 8), np.float32)
    x3 = np.empty((B, 1536, H // 16, W // 16), np.float32); x4 = np.empty((B, 5760, H // 32, W // 32), np.float32)
    x4s = np.empty((B, 3072, H // 32, W // 32), np.float32)
    _chk(lib().orc_forward_parts(C.byref(cfg), arr, C.c_size_t(n), _p(x), B, H, W, _p(out), fp, _p(x1), _p(x2), _p(x3), _p(x4), _p(x4s)))
    return out, dict(f=f, x1=x1, x2=x2, x3=x3, x4=x4, x4s=x4s)


def swin_forward(cfg, weights, x, prefix=""):
    x = _f(x)
    B, _, H, W = x.shape
    arr, n, keep = named(weights)
    P = cfg.patch_size
    hs, ws = [-(-H // P)], [-(-W // P)]
    for _ in range(3):
        hs.append((hs[-1] + 1) // 2); ws.append((ws[-1] + 1) // 2)
    outs = [np.empty((B, cfg.embed_dim << i, hs[i], ws[i]), np.float32) for i in range(4)]
    op = (C.c_void_p * 4)(*[a.ctypes.data for a in outs])
    _chk(lib().orc_swin_forward(C.byref(cfg), arr, C.c_size_t(n), prefix.encode(), _p(x), B, H, W, op))
    return outs


_ACT = {None: 0, "none": 0, "relu": 1, "gelu_erf": 2}


def linear(x, w, bias=None, act=None, residual=None):
    x, w = _f(x), _f(w)
    M, K = x.shape
    N = w.shape[0]
    b = _f(bias) if bias is not None else None
    r = _f(residual) if residual is not None else None
    y = np.empty((M, N), np.float32)
    _chk(lib().orc_linear(_p(x), M, K, _p(w), _p(b), N, _ACT[act], _p(r), _p(y)))
    return y


def layer_norm(x, g, b, eps=1e-5):
    x, g, b = _f(x), _f(g), _f(b)
    Cc = x.shape[-1]
    y = np.empty_like(x)
    _chk(lib().orc_layer_norm(_p(x), x.size // Cc, Cc, _p(g), _p(b), C.c_float(eps), _p(y)))
    return y


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, bn=None, bn_eps=1e-5, act=None):
    x, w = _f(x), _f(w)
    B, Cc, H, W = x.shape
    O, _, kh, kw = w.shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    b = _f(bias) if bias is not None else None
    bnp = [_f(a) for a in bn] if bn is not None else [None] * 4
    y = np.empty((B, O, Ho, Wo), np.float32)
    _chk(lib().orc_conv2d(_p(x), B, Cc, H, W, _p(w), _p(b), O, kh, kw, stride, padding, dilation, _p(bnp[0]), _p(bnp[1]), _p(bnp[2]),
                          _p(bnp[3]), C.c_float(bn_eps), _ACT[act], _p(y)))
    return y


def upsample_bilinear2d(x, oh, ow):
    x = _f(x)
    B, Cc, H, W = x.shape
    y = np.empty((B, Cc, oh, ow), np.float32)
    _chk(lib().orc_upsample_bilinear2d(_p(x), B, Cc, H, W, oh, ow, _p(y)))
    return y


def window_attention(x, heads, shift, weights, prefix="", window_size=12):
    x = _f(x)
    B, H, W, Cc = x.shape
    arr, n, keep = named(weights)
    y = np.empty_like(x)
    _chk(lib().orc_window_attention(_p(x), B, H, W, Cc, heads, window_size, shift, arr, C.c_size_t(n), prefix.encode(), _p(y)))
    return y


def patch_merging(x, H, W, weights, prefix=""):
    x = _f(x)
    B, L, Cc = x.shape
    arr, n, keep = named(weights)
    y = np.empty((B, ((H + 1) // 2) * ((W + 1) // 2), 2 * Cc), np.float32)
    _chk(lib().orc_patch_merging(_p(x), B, H, W, Cc, arr, C.c_size_t(n), prefix.encode(), _p(y)))
    return y


def deform_conv2d(x, offset_w, offset_b, mod_w, mod_b, w, bias, k, stride, pad, mode):
    x = _f(x)
    B, Cc, H, W = x.shape
    O = w.shape[0]
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    arrs = [_f(a) if a is not None else None for a in (offset_w, offset_b, mod_w, mod_b, w, bias)]
    y = np.empty((B, O, Ho, Wo), np.float32)
    _chk(lib().orc_deform_conv2d(_p(x), B, Cc, H, W, *[_p(a) for a in arrs], O, k, stride, pad, int(mode), _p(y)))
    return y


def aspp_deformable(x, weights, mode=0, prefix="", out_channels=None):
    """ASPPDeformable::forward (aspp.rs:303-333): x [B,in_channels,H,W]; mode 0 = reference_cpu, 1 = deformable; out_channels None = in_channels"""
    x = _f(x)
    B, ic, H, W = x.shape
    oc = int(out_channels) if out_channels else ic
    arr, n, keep = named(weights)
    y = np.empty((B, oc, H, W), np.float32)
    _chk(lib().orc_aspp(arr, C.c_size_t(n), prefix.encode(), ic, oc, int(mode), _p(x), B, H, W, _p(y)))
    return y


def decblk(x, weights, out_channels, mode=0, prefix="", use_aspp=True, inter_channels_adaptive=False):
    """BasicDecBlk::forward (decoder.rs:126-141): x [B,Cin,H,W] -> [B,out_channels,H,W]; weights under `prefix`"""
    x = _f(x)
    B, cin, H, W = x.shape
    arr, n, keep = named(weights)
    y = np.empty((B, out_channels, H, W), np.float32)
    inter = cin // 4 if inter_channels_adaptive else 64
    _chk(lib().orc_decblk(arr, C.c_size_t(n), prefix.encode(), cin, int(out_channels), inter, int(bool(use_aspp)), int(mode), _p(x), B, H, W, _p(y)))
    return y


def squeeze(cfg, weights, x4):
    x4 = _f(x4)
    B, _, h, w = x4.shape
    arr, n, keep = named(weights)
    y = np.empty((B, 3072, h, w), np.float32)
    _chk(lib().orc_squeeze(C.byref(cfg), arr, C.c_size_t(n), _p(x4), B, h, w, _p(y)))
    return y


def decoder(cfg, weights, x, x1, x2, x3, x4):
    x, x1, x2, x3, x4 = (_f(a) for a in (x, x1, x2, x3, x4))
    B, _, H, W = x.shape
    arr, n, keep = named(weights)
    y = np.empty((B, 1, H, W), np.float32)
    _chk(lib().orc_decoder(C.byref(cfg), arr, C.c_size_t(n), _p(x), _p(x1), _p(x2), _p(x3), _p(x4), B, H, W, _p(y)))
    return y
