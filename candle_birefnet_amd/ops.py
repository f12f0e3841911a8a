"""Op-level entry points of the C ABI (the candle ops the reference's hot path calls), used by the parity tests.
Weights are host arrays; activations may be numpy (host) or torch-on-GPU (device) tensors."""
from . import _ffi
from . import tensors as T

_COMPUTE = {"f32": _ffi.BRN_F32, "f32_split3": _ffi.BRN_F32_SPLIT3, "f32_split2": _ffi.BRN_F32_SPLIT2, "f32_half2": _ffi.BRN_F32_HALF2, "bf16": _ffi.BRN_BF16, "f16": _ffi.BRN_F16}


def set_compute(mode: str):
    """contraction arithmetic of ops.linear / ops.conv2d on this thread (include/birefnet_hip.h brn_dtype)"""
    _ffi.check(_ffi.lib.brn_set_op_compute(_COMPUTE[mode]))


_ACT = {None: _ffi.BRN_ACT_NONE, "none": _ffi.BRN_ACT_NONE, "relu": _ffi.BRN_ACT_RELU, "gelu_erf": _ffi.BRN_ACT_GELU_ERF}


def linear(x, w, bias=None, act=None, residual=None, device=0):
    """candle_nn::linear(+gelu_erf)(+residual): x [M,K], w [N,K] -> [M,N]  (swin.rs:98-107,130-131)."""
    M, K = (int(v) for v in x.shape)
    N = int(w.shape[0])
    px, loc, keep, _ = T.as_arg(x)
    pr, rloc, rkeep, _ = T.as_arg(residual, (M, N)) if residual is not None else (None, loc, None, None)
    if residual is not None and rloc != loc:
        raise ValueError("x and residual must live on the same side")
    pw, kw = T.host_ptr(w)
    pb, kb = T.host_ptr(bias)
    y = T.alloc_like(keep, (M, N))
    _ffi.check(_ffi.lib.brn_linear_forward(px, M, K, pw, pb, N, _ACT[act], pr, T.ptr_of(y), loc, T.device_of(keep, device),
                                           T.stream_of(keep)))
    return y


def layer_norm(x, gamma, beta, eps=1e-5, device=0):
    """candle_nn::layer_norm forward over the last dim (swin.rs:333)."""
    C = int(x.shape[-1])
    rows = 1
    for v in x.shape[:-1]:
        rows *= int(v)
    px, loc, keep, _ = T.as_arg(x)
    pg, kg = T.host_ptr(gamma)
    pb, kb = T.host_ptr(beta)
    y = T.alloc_like(keep, tuple(x.shape))
    _ffi.check(_ffi.lib.brn_layer_norm_forward(px, rows, C, pg, pb, float(eps), T.ptr_of(y), loc, T.device_of(keep, device),
                                               T.stream_of(keep)))
    return y


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, bn=None, bn_eps=1e-5, act=None, device=0):
    """candle_nn::conv2d (NCHW) with optional eval-mode batch_norm (gamma, beta, running_mean, running_var) + activation."""
    B, Cc, H, W = (int(v) for v in x.shape)
    O, Ci, kh, kw = (int(v) for v in w.shape)
    if Ci != Cc:
        raise ValueError("channel mismatch")
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    px, loc, keep, _ = T.as_arg(x)
    pw, k1 = T.host_ptr(w)
    pb, k2 = T.host_ptr(bias)
    bnp = [T.host_ptr(a) for a in bn] if bn is not None else [(None, None)] * 4
    y = T.alloc_like(keep, (B, O, Ho, Wo))
    _ffi.check(_ffi.lib.brn_conv2d_forward(px, B, Cc, H, W, pw, pb, O, kh, kw, stride, padding, dilation, bnp[0][0], bnp[1][0],
                                           bnp[2][0], bnp[3][0], float(bn_eps), _ACT[act], T.ptr_of(y), loc,
                                           T.device_of(keep, device), T.stream_of(keep)))
    return y


def upsample_bilinear2d(x, out_h, out_w, device=0):
    """Tensor::upsample_bilinear2d(h, w, align_corners=true), NCHW (birefnet.rs:332)."""
    B, Cc, H, W = (int(v) for v in x.shape)
    px, loc, keep, _ = T.as_arg(x)
    y = T.alloc_like(keep, (B, Cc, out_h, out_w))
    _ffi.check(_ffi.lib.brn_upsample_bilinear2d(px, B, Cc, H, W, out_h, out_w, T.ptr_of(y), loc, T.device_of(keep, device),
                                                T.stream_of(keep)))
    return y


def window_attention(x, heads, shift, qkv_w, qkv_b, proj_w, proj_b, rel_table, window_size=12, device=0):
    """SwinTransformerBlock::forward between norm1 and the residual (swin.rs:356-403): x [B,H,W,C] -> [B,H,W,C]."""
    B, H, W, Cc = (int(v) for v in x.shape)
    px, loc, keep, _ = T.as_arg(x)
    hp = [T.host_ptr(a) for a in (qkv_w, qkv_b, proj_w, proj_b, rel_table)]
    y = T.alloc_like(keep, (B, H, W, Cc))
    _ffi.check(_ffi.lib.brn_window_attention_forward(px, B, H, W, Cc, heads, window_size, shift, *[h[0] for h in hp],
                                                     T.ptr_of(y), loc, T.device_of(keep, device), T.stream_of(keep)))
    return y


def patch_merging(x, H, W, norm_g, norm_b, reduction_w, device=0):
    """PatchMerging::forward (swin.rs:491-527): x [B,H*W,C] -> [B, ceil(H/2)*ceil(W/2), 2C]."""
    B, L, Cc = (int(v) for v in x.shape)
    if L != H * W:
        raise ValueError("Input feature has wrong size")
    px, loc, keep, _ = T.as_arg(x)
    hp = [T.host_ptr(a) for a in (norm_g, norm_b, reduction_w)]
    y = T.alloc_like(keep, (B, ((H + 1) // 2) * ((W + 1) // 2), 2 * Cc))
    _ffi.check(_ffi.lib.brn_patch_merging_forward(px, B, H, W, Cc, *[h[0] for h in hp], T.ptr_of(y), loc,
                                                  T.device_of(keep, device), T.stream_of(keep)))
    return y


def aspp_deformable(x, tensors, mode="reference_cpu", prefix="", out_channels=None, device=0):
    """ASPPDeformable::new(in_channels, out_channels, vb.pp(prefix)) + forward (aspp.rs:236-333) on an NCHW map; `tensors`: name -> host
    array of the module's weights under `prefix` (SURVEY.md App. A <ASPP>; BasicDecBlk builds it with 64 -> 64).  mode: "reference_cpu"
    (aspp.rs:183-185) or "deformable" (aspp.rs:58-165).  out_channels None = in_channels (aspp.rs:242)."""
    from .birefnet import _named_array
    B, Cc, H, W = (int(v) for v in x.shape)
    oc = int(out_channels) if out_channels else Cc
    arr, keep_w = _named_array(tensors)
    px, loc, keep, _ = T.as_arg(x)
    y = T.alloc_like(keep, (B, oc, H, W))
    m = {"reference_cpu": _ffi.BRN_DEFORM_REFERENCE_CPU, "deformable": _ffi.BRN_DEFORM_DEFORMABLE}[mode]
    _ffi.check(_ffi.lib.brn_aspp_deformable_forward(arr, len(arr), prefix.encode(), Cc, oc, m, px, B, H, W, T.ptr_of(y), loc, T.device_of(keep, device),
                                                    T.stream_of(keep)))
    del keep_w
    return y


def decblk(x, tensors, out_channels, mode="reference_cpu", prefix="", use_aspp=True, inter_channels_adaptive=False, device=0):
    """BasicDecBlk::new(in_channels, out_channels, &DecoderConfig { use_aspp_deformable: use_aspp, inter_channels_adaptive }, vb.pp(prefix))
    + forward (decoder.rs:78-141) on an NCHW map [B,in_channels,H,W] -> [B,out_channels,H,W]; `tensors`: name -> host array of the
    block's weights under `prefix` (SURVEY.md App. A <DecBlk>).  inter_channels = 64, or in_channels // 4 when adaptive (decoder.rs:94-98)."""
    from .birefnet import _named_array
    B, Cc, H, W = (int(v) for v in x.shape)
    arr, keep_w = _named_array(tensors)
    px, loc, keep, _ = T.as_arg(x)
    y = T.alloc_like(keep, (B, int(out_channels), H, W))
    m = {"reference_cpu": _ffi.BRN_DEFORM_REFERENCE_CPU, "deformable": _ffi.BRN_DEFORM_DEFORMABLE}[mode]
    inter = Cc // 4 if inter_channels_adaptive else 64
    _ffi.check(_ffi.lib.brn_decblk_forward(arr, len(arr), prefix.encode(), Cc, int(out_channels), inter, int(bool(use_aspp)), m, px, B, H, W, T.ptr_of(y), loc,
                                           T.device_of(keep, device), T.stream_of(keep)))
    del keep_w
    return y


def linear_residual_layer_norm(x, w, bias, residual, gamma, beta, eps=1e-5, device=0):
    """x_out = x W^T + bias + residual; y_out = LayerNorm(x_out) gamma + beta — swin.rs:310 + :406 + :407 (and :106-107 + the next block's
    norm1) as one call; returns (x_out, y_out), both [M,N]."""
    M, K = (int(v) for v in x.shape)
    N = int(w.shape[0])
    px, loc, keep, _ = T.as_arg(x)
    pr, rloc, rkeep, _ = T.as_arg(residual, (M, N))
    if rloc != loc:
        raise ValueError("x and residual must live on the same side")
    (pw, kw), (pb, kb), (pg, kg), (pbt, kbt) = T.host_ptr(w), T.host_ptr(bias), T.host_ptr(gamma), T.host_ptr(beta)
    xo, yo = T.alloc_like(keep, (M, N)), T.alloc_like(keep, (M, N))
    _ffi.check(_ffi.lib.brn_linear_residual_layer_norm_forward(px, M, K, pw, pb, N, pr, pg, pbt, float(eps), T.ptr_of(xo), T.ptr_of(yo), loc,
                                                               T.device_of(keep, device), T.stream_of(keep)))
    return xo, yo
