"""Inputs of the golden fixtures (regenerated from seeds — tests/golden/make_golden.py stores outputs only) expressed
against a generic backend `be` exposing window_attention / patch_merging / upsample_bilinear2d / deform_conv2d."""
import numpy as np

import candle_birefnet_amd as cb   # config + synthetic-weight recipe (pure python; no GPU work at import)
from candle_birefnet_amd.weights import synth_tensor


def rnd(*shape, seed=0, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * std).astype(np.float32)


def attn_weights(C, heads, seed):
    return {"attn.qkv.weight": rnd(3 * C, C, seed=seed, std=C ** -0.5), "attn.qkv.bias": rnd(3 * C, seed=seed + 1, std=0.2),
            "attn.proj.weight": rnd(C, C, seed=seed + 2, std=C ** -0.5), "attn.proj.bias": rnd(C, seed=seed + 3, std=0.02),
            "attn.relative_position_bias_table": rnd(529, heads, seed=seed + 4, std=0.5)}


PM_W = {"norm.weight": 1 + rnd(128, seed=1, std=0.1), "norm.bias": rnd(128, seed=2, std=0.1), "reduction.weight": rnd(64, 128, seed=3, std=128 ** -0.5)}


def deform_tensors(k, C=64, O=32):
    return {"offset_conv.weight": rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": rnd(2 * k * k, seed=2, std=0.3),
            "modulator_conv.weight": rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), "modulator_conv.bias": rnd(k * k, seed=4, std=0.1),
            "regular_conv.weight": rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), "regular_conv.bias": rnd(O, seed=6, std=0.1)}


def _attn(H, W, shift):
    def run(be):
        w = attn_weights(64, 2, 10)
        return be.window_attention(rnd(1, H, W, 64, seed=99), 2, shift, w)
    return run


def _pm(H, W, seed):
    return lambda be: be.patch_merging(rnd(1, H * W, 32, seed=seed), H, W, PM_W)


def _up(a, b):
    return lambda be: be.upsample_bilinear2d(rnd(1, 3, a, a, seed=6), b, b)


def _deform(k, mode):
    def run(be):
        t = deform_tensors(k)
        return be.deform_conv2d(rnd(1, 64, 8, 8, seed=9), t, k, 1, k // 2, mode)
    return run


KAT_CASES = {
    "attn_24_s0": _attn(24, 24, 0), "attn_24_s6": _attn(24, 24, 6), "attn_16_s6": _attn(16, 16, 6), "attn_16_s0": _attn(16, 16, 0),
    "pm_6x6": _pm(6, 6, 4), "pm_7x5": _pm(7, 5, 5),
    "up_5_9": _up(5, 9), "up_9_5": _up(9, 5), "up_4_4": _up(4, 4),
    "deform_k1": _deform(1, 1), "deform_k3": _deform(3, 1), "deform_k7": _deform(7, 1),
    "regular_k1": _deform(1, 0), "regular_k3": _deform(3, 0), "regular_k7": _deform(7, 0),
}

MODEL_CASES = {"m64_d2222_ref": ([2, 2, 2, 2], 64, 1, "reference_cpu"), "m64_d2222_def": ([2, 2, 2, 2], 64, 1, "deformable"),
               "m96_d2222_ref_b2": ([2, 2, 2, 2], 96, 2, "reference_cpu"), "m128_full_ref": ([2, 2, 18, 2], 128, 1, "reference_cpu")}


def model_case(tag):
    depths, S, B, mode = MODEL_CASES[tag]
    cfg = cb.BiRefNetConfig(deform_mode=mode)
    cfg.swin.depths = list(depths)
    return cfg, cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42), cb.synth_input(B, S, S)


def weights_checksum():
    names = [("bb.layers.0.blocks.0.attn.qkv.weight", (576, 192), "lin_w"), ("decoder.conv_out1.0.weight", (1, 240, 1, 1), "conv_w"),
             ("bb.layers.3.blocks.1.attn.relative_position_bias_table", (529, 48), "rel_bias"),
             ("squeeze_module.0.bn_in.running_var", (64,), "bn_var")]
    return np.array([float(np.asarray(synth_tensor(n, s, k, 42), np.float64).sum()) for n, s, k in names] +
                    [float(np.asarray(cb.synth_input(1, 8, 8), np.float64).sum())], np.float64)


class OracleBackend:
    """adapts oracle/oracle.py to the KAT case signatures"""

    def __init__(self, O):
        self.O = O

    def window_attention(self, x, heads, shift, w):
        return self.O.window_attention(x, heads, shift, w)

    def patch_merging(self, x, H, W, w):
        return self.O.patch_merging(x, H, W, w)

    def upsample_bilinear2d(self, x, oh, ow):
        return self.O.upsample_bilinear2d(x, oh, ow)

    def deform_conv2d(self, x, t, k, stride, pad, mode):
        return self.O.deform_conv2d(x, t["offset_conv.weight"], t["offset_conv.bias"], t["modulator_conv.weight"], t["modulator_conv.bias"],
                                    t["regular_conv.weight"], t["regular_conv.bias"], k, stride, pad, mode)


class HipBackend:
    """adapts the product's C-ABI wrappers (candle_birefnet_amd.ops / DeformableConv2d) to the KAT case signatures"""

    def window_attention(self, x, heads, shift, w):
        return cb.ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                       w["attn.relative_position_bias_table"])

    def patch_merging(self, x, H, W, w):
        return cb.ops.patch_merging(x, H, W, w["norm.weight"], w["norm.bias"], w["reduction.weight"])

    def upsample_bilinear2d(self, x, oh, ow):
        return cb.ops.upsample_bilinear2d(x, oh, ow)

    def deform_conv2d(self, x, t, k, stride, pad, mode):
        layer = cb.DeformableConv2d.new(x.shape[1], t["regular_conv.weight"].shape[0], k, stride, pad, cb.VarBuilder.from_tensors(t),
                                        mode="deformable" if mode else "reference_cpu")
        return layer.forward(x)
