"""BiRefNet / SwinTransformer / DeformableConv2d — the reference crate's public surface (lib.rs:12-14,
birefnet.rs:380-476, swin.rs:718-797, deform_conv.rs:17-222) on top of the C ABI.  Method names, argument meaning and
error behaviour follow the Rust API; arithmetic happens only inside libbirefnet_hip.so."""
import ctypes as C

import numpy as np

from . import _ffi
from . import tensors as T
from .config import BiRefNetConfig, SwinConfig, swin_to_c
from .weights import VarBuilder


def _named_array(tensors):
    """dict name -> np.ndarray  ->  (brn_named_tensor[], keepalive)"""
    n = len(tensors)
    arr = (_ffi.brn_named_tensor * n)()
    keep = []
    for i, (name, a) in enumerate(tensors.items()):
        a = np.ascontiguousarray(a, dtype=np.float32)
        shp = (C.c_int64 * a.ndim)(*a.shape)
        nm = name.encode()
        keep += [a, shp, nm]
        arr[i].name = nm
        arr[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].shape = shp
        arr[i].ndim = a.ndim
    return arr, keep


class _Piece:
    """A pub field of BiRefNet that bench_inference.rs calls on its own (model.backbone / squeeze_module / decoder)."""

    def __init__(self, model, kind):
        self._m, self._kind = model, kind

    def forward(self, *args):
        return getattr(self._m, "_" + self._kind + "_forward")(*args)


class BiRefNet:
    """birefnet.rs:380-385.  `BiRefNet.new(config, vb)` == BiRefNet::new (birefnet.rs:389)."""

    COMPUTE = {"f32": _ffi.BRN_F32, "f32_split3": _ffi.BRN_F32_SPLIT3, "f32_split2": _ffi.BRN_F32_SPLIT2, "bf16": _ffi.BRN_BF16,
               "f32_half2": _ffi.BRN_F32_HALF2, "f16": _ffi.BRN_F16, "bf16_dec_split2": _ffi.BRN_BF16_DEC_SPLIT2}   # (BRN_BF16_OPERANDS: diag build only; bf16_dec_split2: whole models only)

    def __init__(self, config: BiRefNetConfig, vb, device: int = 0, max_batch: int = 0, max_size=(0, 0), compute: str = "f32"):
        """vb: a VarBuilder, or the path of a .safetensors checkpoint (read natively by the library: the
        VarBuilder::from_mmaped_safetensors + BiRefNet::new pair of infer_image.rs:35-40)."""
        self.config = config
        self.compute = compute
        self._h = C.c_void_p()
        self._device = device
        cfg = config.to_c()
        if isinstance(vb, (str, bytes)) or hasattr(vb, "__fspath__"):
            import os
            path = os.fsencode(vb)
            _ffi.check(_ffi.lib.brn_model_create_from_safetensors(C.byref(cfg), path, b"", device, self.COMPUTE[compute], int(max_batch),
                                                                  int(max_size[0]), int(max_size[1]), C.byref(self._h)))
        else:
            arr, keep = _named_array(vb.tensors_under_prefix())
            _ffi.check(_ffi.lib.brn_model_create(C.byref(cfg), arr, len(arr), device, self.COMPUTE[compute], int(max_batch),
                                                 int(max_size[0]), int(max_size[1]), C.byref(self._h)))
            del keep
        self.backbone = _Piece(self, "backbone")
        self.squeeze_module = _Piece(self, "squeeze")
        self.decoder = _Piece(self, "decoder")

    @staticmethod
    def new(config: BiRefNetConfig, vb, **kw):
        return BiRefNet(config, vb, **kw)

    @staticmethod
    def from_safetensors(config: BiRefNetConfig, path, **kw):
        return BiRefNet(config, path, **kw)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _ffi.lib.brn_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- forward_logits / forward (birefnet.rs:412-469) --
    def _run(self, fn, x):
        if len(x.shape) != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [B,3,H,W], got {tuple(x.shape)}")
        B, _, H, W = (int(v) for v in x.shape)
        px, loc, keep, _ = T.as_arg(x)
        out = T.alloc_like(keep, (B, 1, H, W))
        _ffi.check(fn(self._h, px, B, H, W, loc, T.ptr_of(out), loc, T.stream_of(keep)))
        return out

    def forward_logits(self, x):
        """x [B,3,H,W] f32 -> logits [B,1,H,W] (pre-sigmoid)."""
        return self._run(_ffi.lib.brn_forward_logits, x)

    def forward(self, x):
        """sigmoid(forward_logits(x)) (birefnet.rs:466-469)."""
        return self._run(_ffi.lib.brn_forward, x)

    __call__ = forward  # impl Module for BiRefNet (birefnet.rs:472-476)

    # -- pieces --
    def _backbone_forward(self, x):
        B, _, H, W = (int(v) for v in x.shape)
        px, loc, keep, _ = T.as_arg(x)
        hs, ws = _stage_dims(H, W, self.config.swin.patch_size)
        outs = [T.alloc_like(keep, (B, self.config.swin.embed_dim << i, hs[i], ws[i])) for i in range(4)]
        ptrs = (C.c_void_p * 4)(*[T.ptr_of(o) for o in outs])
        _ffi.check(_ffi.lib.brn_model_backbone_forward(self._h, px, B, H, W, loc, ptrs, loc, T.stream_of(keep)))
        return outs

    def _squeeze_forward(self, x4):
        B, Cc, h, w = (int(v) for v in x4.shape)
        if Cc != self.config.x4_channels():
            raise ValueError(f"squeeze_module expects {self.config.x4_channels()} channels, got {Cc}")
        px, loc, keep, _ = T.as_arg(x4)
        out = T.alloc_like(keep, (B, self.config.lateral_channels()[3], h, w))
        _ffi.check(_ffi.lib.brn_model_squeeze_forward(self._h, px, B, h, w, loc, T.ptr_of(out), loc, T.stream_of(keep)))
        return out

    def _decoder_forward(self, x, x1, x2, x3, x4):
        B, _, H, W = (int(v) for v in x.shape)
        lat = self.config.lateral_channels()
        args = [T.as_arg(x, (B, 3, H, W)), T.as_arg(x1, (B, lat[0], H // 4, W // 4)), T.as_arg(x2, (B, lat[1], H // 8, W // 8)),
                T.as_arg(x3, (B, lat[2], H // 16, W // 16)), T.as_arg(x4, (B, lat[3], H // 32, W // 32))]
        loc = args[0][1]
        if any(a[1] != loc for a in args):
            raise ValueError("decoder inputs must all live on the same side (host or device)")
        out = T.alloc_like(args[0][2], (B, 1, H, W))
        _ffi.check(_ffi.lib.brn_model_decoder_forward(self._h, *[a[0] for a in args], B, H, W, loc, T.ptr_of(out), loc,
                                                      T.stream_of(args[0][2])))
        return out

    def set_streams(self, sub_batch_streams: int = 0, branch_stream_mask: int = -1):
        """how a forward is spread over HIP streams (brn_model_set_streams); 0 / -1 = the library defaults"""
        _ffi.check(_ffi.lib.brn_model_set_streams(self._h, int(sub_batch_streams), int(branch_stream_mask)))

    # -- timers mirroring bench_inference.rs:37-92 --
    def set_profiling(self, on: bool):
        _ffi.check(_ffi.lib.brn_model_set_profiling(self._h, int(bool(on))))

    def last_timings(self):
        ms = (C.c_float * 5)()
        _ffi.check(_ffi.lib.brn_model_last_timings(self._h, C.byref(ms)))
        return dict(zip(("backbone_full", "backbone_half_fusion", "squeeze", "decoder", "total"), [float(v) for v in ms]))

    def last_kernel_stats(self):
        n = 16
        la, ms, fl, by, no = (C.c_int * n)(), (C.c_float * n)(), (C.c_double * n)(), (C.c_double * n)(), C.c_int(0)
        _ffi.check(_ffi.lib.brn_model_last_kernel_stats(self._h, n, la, ms, fl, by, C.byref(no)))
        return {_ffi.lib.brn_kernel_family_name(i).decode(): {"launches": la[i], "ms": float(ms[i]), "flop": fl[i], "bytes": by[i]}
                for i in range(no.value)}


class BiRefNetDecoder:
    """birefnet.rs:121-377 on its own: BiRefNetDecoder::new(config, vb) / forward(x, x1, x2, x3, x4) — a handle that holds only the
    decoder's weights (brn_decoder_create); `vb` is positioned at the decoder's prefix (vb.pp("decoder") in birefnet.rs:401)."""

    def __init__(self, config: BiRefNetConfig, vb, device: int = 0, compute: str = "f32"):
        self.config = config
        self._h = C.c_void_p()
        cfg = config.to_c()
        arr, keep = _named_array(vb.tensors_under_prefix())
        _ffi.check(_ffi.lib.brn_decoder_create(C.byref(cfg), arr, len(arr), b"", device, BiRefNet.COMPUTE[compute], C.byref(self._h)))
        del keep

    @staticmethod
    def new(config, vb, **kw):
        return BiRefNetDecoder(config, vb, **kw)

    forward = BiRefNet._decoder_forward

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _ffi.lib.brn_model_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SqueezeModule:
    """birefnet.rs:70-94 on its own: SqueezeModule::new(in_channels, out_channels, vb) = one BasicDecBlk under vb.pp("0"), with the
    ASPP always on (birefnet.rs:76-79)."""

    def __init__(self, in_channels, out_channels, vb, mode="reference_cpu", device=0):
        self.in_channels, self.out_channels, self.mode, self.device = in_channels, out_channels, mode, device
        self._t = vb.pp("0").tensors_under_prefix()

    @staticmethod
    def new(*a, **kw):
        return SqueezeModule(*a, **kw)

    def forward(self, x):
        from . import ops
        if int(x.shape[1]) != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {int(x.shape[1])}")
        return ops.decblk(x, self._t, self.out_channels, mode=self.mode, device=self.device)

    __call__ = forward


def _stage_dims(H, W, patch):
    h, w = -(-H // patch), -(-W // patch)
    hs, ws = [], []
    for _ in range(4):
        hs.append(h); ws.append(w)
        h, w = (h + 1) // 2, (w + 1) // 2
    return hs, ws


class SwinTransformer:
    """swin.rs:718-797: SwinTransformer::new(config, vb) / forward(x) -> [x1, x2, x3, x4] (NCHW)."""

    def __init__(self, config: SwinConfig, vb: VarBuilder, device: int = 0, compute: str = "f32"):
        self.config = config
        self.compute = compute
        self._h = C.c_void_p()
        arr, keep = _named_array(vb.tensors_under_prefix())
        cfg = swin_to_c(config)
        _ffi.check(_ffi.lib.brn_set_op_compute(BiRefNet.COMPUTE[compute]))     # the handle is built for the calling thread's op arithmetic
        try:
            _ffi.check(_ffi.lib.brn_swin_create(C.byref(cfg), arr, len(arr), b"", device, C.byref(self._h)))
        finally:
            _ffi.check(_ffi.lib.brn_set_op_compute(_ffi.BRN_F32))
        del keep

    @staticmethod
    def new(config, vb, **kw):
        return SwinTransformer(config, vb, **kw)

    def forward(self, x):
        B, _, H, W = (int(v) for v in x.shape)
        px, loc, keep, _ = T.as_arg(x)
        hs, ws = _stage_dims(H, W, self.config.patch_size)
        outs = [T.alloc_like(keep, (B, self.config.embed_dim << i, hs[i], ws[i])) for i in range(4)]
        ptrs = (C.c_void_p * 4)(*[T.ptr_of(o) for o in outs])
        _ffi.check(_ffi.lib.brn_swin_forward(self._h, px, B, H, W, loc, ptrs, loc, T.stream_of(keep)))
        return outs

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _ffi.lib.brn_swin_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeformableConv2d:
    """deform_conv.rs:17-222: DeformableConv2d::new(in, out, k, stride, pad, vb) / forward(x).
    mode "reference_cpu" = the CPU fallback (deform_conv.rs:95-98), "deformable" = the Metal path (:101-215)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, vb: VarBuilder, mode="reference_cpu", device=0):
        k = kernel_size
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = in_channels, out_channels, k, stride, padding
        self.offset_w = vb.get((2 * k * k, in_channels, k, k), "offset_conv.weight")
        self.offset_b = vb.get((2 * k * k,), "offset_conv.bias")
        self.mod_w = vb.get((k * k, in_channels, k, k), "modulator_conv.weight")
        self.mod_b = vb.get((k * k,), "modulator_conv.bias")
        self.w = vb.get((out_channels, in_channels, k, k), "regular_conv.weight")
        self.b = vb.get((out_channels,), "regular_conv.bias")
        self.mode = {"reference_cpu": _ffi.BRN_DEFORM_REFERENCE_CPU, "deformable": _ffi.BRN_DEFORM_DEFORMABLE}[mode]
        self.device = device

    @staticmethod
    def new(*a, **kw):
        return DeformableConv2d(*a, **kw)

    def forward(self, x):
        B, Cc, H, W = (int(v) for v in x.shape)
        if Cc != self.in_channels:
            raise ValueError(f"expected {self.in_channels} input channels, got {Cc}")
        k, s, p = self.kernel_size, self.stride, self.padding
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        px, loc, keep, _ = T.as_arg(x)
        out = T.alloc_like(keep, (B, self.out_channels, Ho, Wo))
        hp = [T.host_ptr(a) for a in (self.offset_w, self.offset_b, self.mod_w, self.mod_b, self.w, self.b)]
        _ffi.check(_ffi.lib.brn_deform_conv2d_forward(px, B, Cc, H, W, *[h[0] for h in hp], self.out_channels, k, s, p, self.mode,
                                                      T.ptr_of(out), loc, T.device_of(keep, self.device), T.stream_of(keep)))
        return out

    __call__ = forward
