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


# ---- SURVEY.md §8(c) KATs added in round 3: ASPPDeformable on 12x12 (both deform modes), SimpleConvs, GdtConvs, and the
# roll + partition + mask INDEX MAP for R = 16 / ws = 12 (trivial arithmetic: every output is a mean of position codes) ----
def aspp_tensors():
    from candle_birefnet_amd.weights import _decblk
    return {n[len("b.dec_att."):]: synth_tensor("kat." + n, s, k, 42) for n, s, k in _decblk("b.", 64, 64) if n.startswith("b.dec_att.")}


def _aspp(mode):
    return lambda be: be.aspp(rnd(1, 64, 12, 12, seed=21), aspp_tensors(), mode)


SC_W = {"conv1.weight": rnd(64, 48, 3, 3, seed=31, std=(48 * 9) ** -0.5), "conv1.bias": rnd(64, seed=32, std=0.1),
        "conv_out.weight": rnd(96, 64, 3, 3, seed=33, std=(64 * 9) ** -0.5), "conv_out.bias": rnd(96, seed=34, std=0.1)}
GDT_W = {"0.weight": rnd(16, 96, 3, 3, seed=41, std=(96 * 9) ** -0.5), "0.bias": rnd(16, seed=42, std=0.1),
         "1.weight": 1 + rnd(16, seed=43, std=0.1), "1.bias": rnd(16, seed=44, std=0.1), "1.running_mean": rnd(16, seed=45, std=0.1),
         "1.running_var": (0.5 + np.random.default_rng(46).random(16)).astype(np.float32)}


def indexmap_weights():
    """q = k = 0 (all scores 0 + mask), v = x, proj = identity, zero bias table: an output token is the MEAN of the input tokens of its
    window that the SW-MSA mask lets it see (pad tokens count with value 0), so the result shows nothing but the index map of
    swin.rs:359-401 (pad -> roll -> partition -> mask -> reverse -> roll -> crop)"""
    C, heads = 64, 2
    qkv = np.zeros((3 * C, C), np.float32)
    qkv[2 * C:] = np.eye(C, dtype=np.float32)
    return {"attn.qkv.weight": qkv, "attn.qkv.bias": np.zeros(3 * C, np.float32), "attn.proj.weight": np.eye(C, dtype=np.float32),
            "attn.proj.bias": np.zeros(C, np.float32), "attn.relative_position_bias_table": np.zeros((529, heads), np.float32)}


def _indexmap(shift):
    def run(be):
        i, j, c = np.meshgrid(np.arange(16), np.arange(16), np.arange(64), indexing="ij")
        x = ((((i * 16 + j) * 37 + c * 11) % 64) / 64.0).astype(np.float32)[None]      # a position code per token and channel
        return be.window_attention(x, 2, shift, indexmap_weights())
    return run


KAT_CASES = {
    "aspp_12_ref": _aspp(0), "aspp_12_def": _aspp(1), "simpleconvs_12": lambda be: be.simple_convs(rnd(1, 48, 12, 12, seed=22), SC_W),
    "gdtconvs_12": lambda be: be.gdt_convs(rnd(1, 96, 12, 12, seed=23), GDT_W),
    "indexmap_16_s0": _indexmap(0), "indexmap_16_s6": _indexmap(6),
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


    def aspp(self, x, t, mode):
        return self.O.aspp_deformable(x, t, mode)

    def simple_convs(self, x, t):        # decoder.rs:50-56: conv3x3 -> conv3x3, no activation in between
        return self.O.conv2d(self.O.conv2d(x, t["conv1.weight"], t["conv1.bias"], padding=1), t["conv_out.weight"], t["conv_out.bias"], padding=1)

    def gdt_convs(self, x, t):           # birefnet.rs:111-117: conv3x3 -> BN -> ReLU
        return self.O.conv2d(x, t["0.weight"], t["0.bias"], padding=1, bn=(t["1.weight"], t["1.bias"], t["1.running_mean"], t["1.running_var"]), act="relu")


class HipBackend:
    """adapts the product's C-ABI wrappers (candle_birefnet_amd.ops / DeformableConv2d) to the KAT case signatures"""

    def aspp(self, x, t, mode):
        return cb.ops.aspp_deformable(x, t, "deformable" if mode else "reference_cpu")

    def simple_convs(self, x, t):
        return cb.ops.conv2d(cb.ops.conv2d(x, t["conv1.weight"], t["conv1.bias"], padding=1), t["conv_out.weight"], t["conv_out.bias"], padding=1)

    def gdt_convs(self, x, t):
        return cb.ops.conv2d(x, t["0.weight"], t["0.bias"], padding=1, bn=(t["1.weight"], t["1.bias"], t["1.running_mean"], t["1.running_var"]), act="relu")

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
