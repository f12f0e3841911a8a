"""Weight-name contract of the path (the VarBuilder paths BiRefNet::new walks, birefnet.rs:389-409),
a VarBuilder mirror over host arrays, a safetensors loader (infer_image.rs:35-40) and the synthetic-weight recipe the
benchmarks use — real ZhengPeng7/BiRefNet weights cannot be downloaded in this environment."""
import zlib
from typing import Dict, List, Tuple

import numpy as np

from .config import BiRefNetConfig, SwinConfig


# ---- name / shape inventory ---------------------------------------------------------------------------------------------
def swin_weight_spec(cfg: SwinConfig, prefix: str = "") -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for SwinTransformer::new (swin.rs:726-764)."""
    s = []
    E, P, IC, ws = cfg.embed_dim, cfg.patch_size, cfg.in_channels, cfg.window_size
    s.append((prefix + "patch_embed.proj.weight", (E, IC, P, P), "conv_w"))
    s.append((prefix + "patch_embed.proj.bias", (E,), "bias"))
    s += _ln(prefix + "patch_embed.norm", E)
    for i, depth in enumerate(cfg.depths):
        C, h = E << i, cfg.num_heads[i]
        hidden = int(C * cfg.mlp_ratio)
        for j in range(depth):
            bp = f"{prefix}layers.{i}.blocks.{j}."
            s += _ln(bp + "norm1", C)
            s += _lin(bp + "attn.qkv", 3 * C, C)
            s += _lin(bp + "attn.proj", C, C)
            s.append((bp + "attn.relative_position_bias_table", ((2 * ws - 1) ** 2, h), "rel_bias"))
            s += _ln(bp + "norm2", C)
            s += _lin(bp + "mlp.fc1", hidden, C)
            s += _lin(bp + "mlp.fc2", C, hidden)
        if i < len(cfg.depths) - 1:
            s += _ln(f"{prefix}layers.{i}.downsample.norm", 4 * C)
            s.append((f"{prefix}layers.{i}.downsample.reduction.weight", (2 * C, 4 * C), "lin_w"))
        s += _ln(f"{prefix}norm{i}", C)
    return s


def _ln(p, C):
    return [(p + ".weight", (C,), "norm_g"), (p + ".bias", (C,), "norm_b")]


def _bn(p, C):
    return [(p + ".weight", (C,), "norm_g"), (p + ".bias", (C,), "norm_b"), (p + ".running_mean", (C,), "bn_mean"),
            (p + ".running_var", (C,), "bn_var")]


def _lin(p, N, K, bias=True):
    return [(p + ".weight", (N, K), "lin_w")] + ([(p + ".bias", (N,), "bias")] if bias else [])


def _conv(p, O, Cin, k, bias=True, kind="conv_w"):
    return [(p + ".weight", (O, Cin, k, k), kind)] + ([(p + ".bias", (O,), "bias")] if bias else [])


def _aspp(ap, ic=64, oc=None):
    """ASPPDeformable::new(ic, oc, vb.pp(ap)) (aspp.rs:247-290)"""
    oc = oc or ic
    s = []
    for mod, k in (("aspp1", 1), ("aspp_deforms.0", 1), ("aspp_deforms.1", 3), ("aspp_deforms.2", 7)):
        cp = f"{ap}{mod}.atrous_conv."
        s += _conv(cp + "offset_conv", 2 * k * k, ic, k, kind="offset_w")
        s += _conv(cp + "modulator_conv", k * k, ic, k, kind="mod_w")
        s += _conv(cp + "regular_conv", 256, ic, k, bias=False)
        s += _bn(f"{ap}{mod}.bn", 256)
    s += _conv(ap + "global_avg_pool.1", 256, ic, 1, bias=False) + _bn(ap + "global_avg_pool.2", 256)
    s += _conv(ap + "conv1", oc, 1280, 1, bias=False) + _bn(ap + "bn1", oc)
    return s


def _decblk(p, cin, cout, inter=64, use_aspp=True):
    """BasicDecBlk::new + ASPPDeformable::new (decoder.rs:104-114, aspp.rs:247-290)."""
    s = _conv(p + "conv_in", inter, cin, 3) + _bn(p + "bn_in", inter)
    if use_aspp:
        s += _aspp(p + "dec_att.", inter)
    s += _conv(p + "conv_out", cout, inter, 3) + _bn(p + "bn_out", cout)
    return s


def birefnet_weight_spec(cfg: BiRefNetConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Every tensor BiRefNet::new asks its VarBuilder for (birefnet.rs:389-409, 170-273)."""
    s = swin_weight_spec(cfg.swin, "bb.")
    lat = cfg.lateral_channels()
    s += _decblk("squeeze_module.0.", cfg.x4_channels(), lat[3])
    ipt_out = [48, 96, 192, 384, 384]
    ipt_in = [3, ipt_out[0], lat[0] // 2, lat[2] // 2, lat[3]]
    for i in range(5):
        p = f"decoder.ipt_blk{i + 1}."
        s += _conv(p + "conv1", 64, ipt_in[i], 3) + _conv(p + "conv_out", ipt_out[i], 64, 3)
    dec_out = [lat[2], lat[1], lat[0], lat[0] // 2]
    dec_in = [lat[3] + ipt_out[4], dec_out[0] + ipt_out[3], dec_out[1] + ipt_out[2], dec_out[2] + ipt_out[1]]
    for n, ci, co in zip((4, 3, 2, 1), dec_in, dec_out):
        s += _decblk(f"decoder.decoder_block{n}.", ci, co)
    for n, c in zip((4, 3, 2), (lat[2], lat[1], lat[0])):
        s += _conv(f"decoder.lateral_block{n}.conv", c, c, 1)
    for n, c in zip((4, 3, 2), dec_out[:3]):
        s += _conv(f"decoder.gdt_convs_{n}.0", 16, c, 3) + _bn(f"decoder.gdt_convs_{n}.1", 16)
        s += _conv(f"decoder.gdt_convs_attn_{n}.0", 1, 16, 1)
        s += _conv(f"decoder.gdt_convs_pred_{n}.0", 1, 16, 1)      # loaded, unused (birefnet.rs:230-232)
        s += _conv(f"decoder.conv_ms_spvn_{n}", 1, c, 1)           # loaded, unused (birefnet.rs:241-243)
    s += _conv("decoder.conv_out1.0", 1, dec_out[3] + ipt_out[0], 1)
    return s


# ---- synthetic weights (BASELINE.md §3 recipe) ----------------------------------------------------------------------------
def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def synth_tensor(name, shape, kind, seed=42):
    g = _rng(seed, name)
    n = int(np.prod(shape))
    if kind in ("lin_w", "conv_w", "offset_w", "mod_w"):
        fan_in = int(np.prod(shape[1:]))
        std = 1.0 / np.sqrt(fan_in)
        if kind == "offset_w":
            std *= 1.5          # offsets of ~1.5 px on O(1) activations: exercises the bilinear gather
        a = g.standard_normal(n, dtype=np.float32) * np.float32(std)
    elif kind == "bias":
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.02)
    elif kind == "norm_g":
        a = np.float32(1.0) + g.standard_normal(n, dtype=np.float32) * np.float32(0.05)
    elif kind == "norm_b":
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.02)
    elif kind == "bn_mean":
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.1)
    elif kind == "bn_var":
        a = g.random(n, dtype=np.float32) + np.float32(0.5)
    elif kind == "rel_bias":
        a = g.standard_normal(n, dtype=np.float32) * np.float32(0.5)   # non-zero, std as test_flash_bias.rs:25
    else:
        raise ValueError(kind)
    return a.reshape(shape)


def synth_weights(spec, seed=42) -> Dict[str, np.ndarray]:
    return {name: synth_tensor(name, shape, kind, seed) for name, shape, kind in spec}


def synth_input(B, H, W, seed0=1000):
    """x[b] ~ N(0,1), seed 1000+b (reference benches: Tensor::randn(0,1,(1,3,1024,1024)), bench_inference.rs:30)."""
    out = np.empty((B, 3, H, W), dtype=np.float32)
    for b in range(B):
        out[b] = _rng(seed0 + b, "input").standard_normal((3, H, W), dtype=np.float32)
    return out


# ---- VarBuilder mirror -------------------------------------------------------------------------------------------------------
class VarBuilder:
    """candle_nn::VarBuilder::from_tensors + pp + get over host arrays (infer_image.rs:38-40)."""

    def __init__(self, tensors: Dict[str, np.ndarray], prefix: str = ""):
        self._t = tensors
        self._prefix = prefix

    @staticmethod
    def from_tensors(tensors: Dict[str, np.ndarray]):
        return VarBuilder(tensors)

    @staticmethod
    def from_safetensors(path: str):
        from safetensors.numpy import load_file
        return VarBuilder({k: np.ascontiguousarray(v, dtype=np.float32) for k, v in load_file(path).items()})

    def pp(self, s):
        return VarBuilder(self._t, f"{self._prefix}{s}.")

    def get(self, shape, name):
        full = self._prefix + name
        if full not in self._t:
            raise KeyError(f"cannot find tensor {full}")
        a = self._t[full]
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"shape mismatch for {full}: expected {tuple(shape)} got {tuple(a.shape)}")
        return a

    def tensors_under_prefix(self):
        p = self._prefix
        return {k[len(p):]: v for k, v in self._t.items() if k.startswith(p)}
