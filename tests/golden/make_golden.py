#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/ from the fp64 torch restatement (tests/torch_ref.py).

PARITY UNPINNED: the reference ships no golden vectors and cannot be run here, so these vectors pin the
oracle and the HIP path to an independent restatement of the same Rust sources, not to candle itself.
Weights and inputs are never stored: both sides regenerate them from seeds (candle_birefnet_amd.weights, numpy Philox);
`weights_checksum` guards against PRNG drift.

  python tests/golden/make_golden.py [--full1024 | --full2048] [--deformable]   (the 1024x1024 fp64 run takes several minutes and ~20 GB;
                                                                  --full2048 runs the restatement in fp32)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import torch_ref as R  # noqa: E402
from candle_birefnet_amd.config import BiRefNetConfig  # noqa: E402  (pure-python config/weight recipe; no GPU code runs)
from candle_birefnet_amd.weights import birefnet_weight_spec, synth_input, synth_tensor, synth_weights  # noqa: E402

D = torch.float64


def rnd(*shape, seed=0, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * std).astype(np.float32)


def attn_weights(C, heads, seed):
    return {"attn.qkv.weight": rnd(3 * C, C, seed=seed, std=C ** -0.5), "attn.qkv.bias": rnd(3 * C, seed=seed + 1, std=0.2),
            "attn.proj.weight": rnd(C, C, seed=seed + 2, std=C ** -0.5), "attn.proj.bias": rnd(C, seed=seed + 3, std=0.02),
            "attn.relative_position_bias_table": rnd(529, heads, seed=seed + 4, std=0.5)}


def kats():
    """per-op known-answer tests at tiny shapes; inputs regenerated from the seeds listed in tests/test_golden.py"""
    out = {}
    # window attention: 4 windows, 2 heads, with and without the SW-MSA mask; and a padded map (16 -> 24)
    w = attn_weights(64, 2, 10)
    for name, (H, W, shift) in {"attn_24_s0": (24, 24, 0), "attn_24_s6": (24, 24, 6), "attn_16_s6": (16, 16, 6), "attn_16_s0": (16, 16, 0)}.items():
        x = torch.from_numpy(rnd(1, H, W, 64, seed=99)).to(D)
        out[name] = R.window_attention_block(x, w, "", 2, 12, shift, D).numpy().astype(np.float32)
    # SURVEY.md §8(c) KATs of round 3 (ASPP 12x12 in both deform modes, SimpleConvs, GdtConvs, the roll + partition + mask index map for
    # R = 16 / ws = 12): inputs and weights are defined once, in tests/golden_cases.py; here they run through the fp64 restatement
    import golden_cases as G

    class TorchBackend:
        def window_attention(self, x, heads, shift, w):
            return R.window_attention_block(torch.from_numpy(x).to(D), w, "", heads, 12, shift, D).numpy()

        def aspp(self, x, t, mode):
            return R.aspp_deformable(torch.from_numpy(x).to(D), t, "", D, "deformable" if mode else "reference_cpu").numpy()

        def simple_convs(self, x, t):
            return R.simple_convs(torch.from_numpy(x).to(D), t, "", D).numpy()

        def gdt_convs(self, x, t):
            return torch.nn.functional.relu(R.bn(R.conv(torch.from_numpy(x).to(D), t, "0", D, pad=1), t, "1", D)).numpy()

    for name in ("aspp_12_ref", "aspp_12_def", "simpleconvs_12", "gdtconvs_12", "indexmap_16_s0", "indexmap_16_s6"):
        out[name] = np.asarray(G.KAT_CASES[name](TorchBackend())).astype(np.float32)
    # patch merging 6x6x32 and an odd 7x5 map
    pw = {"norm.weight": 1 + rnd(128, seed=1, std=0.1), "norm.bias": rnd(128, seed=2, std=0.1), "reduction.weight": rnd(64, 128, seed=3, std=128 ** -0.5)}
    out["pm_6x6"] = R.patch_merging(torch.from_numpy(rnd(1, 36, 32, seed=4)).to(D), 6, 6, pw, "", D).numpy().astype(np.float32)
    out["pm_7x5"] = R.patch_merging(torch.from_numpy(rnd(1, 35, 32, seed=5)).to(D), 7, 5, pw, "", D).numpy().astype(np.float32)
    # bilinear align_corners: 5->9, 9->5, 4->4
    for name, (a, b) in {"up_5_9": (5, 9), "up_9_5": (9, 5), "up_4_4": (4, 4)}.items():
        out[name] = R.up(torch.from_numpy(rnd(1, 3, a, a, seed=6)).to(D), b, b).numpy().astype(np.float32)
    # deformable conv k1 / k3 / k7 on 8x8, both modes
    for k in (1, 3, 7):
        C, O = 64, 32
        t = {"offset_conv.weight": rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": rnd(2 * k * k, seed=2, std=0.3),
             "modulator_conv.weight": rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), "modulator_conv.bias": rnd(k * k, seed=4, std=0.1),
             "regular_conv.weight": rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), "regular_conv.bias": rnd(O, seed=6, std=0.1)}
        td = {n: torch.from_numpy(a).to(D) for n, a in t.items()}
        x = torch.from_numpy(rnd(1, C, 8, 8, seed=9)).to(D)
        off = torch.nn.functional.conv2d(x, td["offset_conv.weight"], td["offset_conv.bias"], padding=k // 2)
        msk = 1.0 / (torch.exp(-torch.nn.functional.conv2d(x, td["modulator_conv.weight"], td["modulator_conv.bias"], padding=k // 2)) + 1.0) * 2.0
        out[f"deform_k{k}"] = R.deform_conv2d(x, off, msk, td["regular_conv.weight"], td["regular_conv.bias"], 1, k // 2).numpy().astype(np.float32)
        out[f"regular_k{k}"] = torch.nn.functional.conv2d(x, td["regular_conv.weight"], td["regular_conv.bias"], padding=k // 2).numpy().astype(np.float32)
    return out


def model_goldens():
    out = {}
    for tag, depths, S, B, mode in (("m64_d2222_ref", [2, 2, 2, 2], 64, 1, "reference_cpu"), ("m64_d2222_def", [2, 2, 2, 2], 64, 1, "deformable"),
                                    ("m96_d2222_ref_b2", [2, 2, 2, 2], 96, 2, "reference_cpu"), ("m128_full_ref", [2, 2, 18, 2], 128, 1, "reference_cpu")):
        cfg = BiRefNetConfig(deform_mode=mode)
        cfg.swin.depths = depths
        w = synth_weights(birefnet_weight_spec(cfg), seed=42)
        x = synth_input(B, S, S)
        t = time.time()
        y, parts = R.forward_logits(x, w, cfg, D, return_parts=True)
        out[tag] = y.numpy().astype(np.float32)
        if tag == "m64_d2222_ref":   # stage-level goldens for the pieces bench_inference.rs drives
            for i, f in enumerate(parts["f"]):
                out[f"{tag}_f{i}"] = f.numpy().astype(np.float32)
            out[f"{tag}_x4s"] = parts["x4s"].numpy().astype(np.float32)
        print(tag, tuple(y.shape), f"{time.time() - t:.1f}s", flush=True)
    return out


def full1024(deform_mode="reference_cpu"):
    """deform_mode="deformable": the Metal / upstream semantics (aspp.rs:58-165) that BASELINE configs[2..4] time;
    stored under m1024_full_def_* in model_1024_def.npz"""
    cfg = BiRefNetConfig(deform_mode=deform_mode)
    tag = "ref" if deform_mode == "reference_cpu" else "def"
    w = synth_weights(birefnet_weight_spec(cfg), seed=42)
    x = synth_input(1, 1024, 1024)
    t = time.time()
    y, parts = R.forward_logits(x, w, cfg, D, return_parts=True)
    print("1024 fp64 restatement", f"{time.time() - t:.1f}s", flush=True)
    yn = y.numpy()
    out = {f"m1024_full_{tag}_s16": yn[:, :, ::16, ::16].astype(np.float32),
           f"m1024_full_{tag}_stats": np.array([yn.sum(), np.abs(yn).sum(), yn.min(), yn.max()], np.float64)}
    for i, f in enumerate(parts["f"]):
        fn = f.numpy()
        out[f"m1024_full_{tag}_f{i}_stats"] = np.array([fn.sum(), np.abs(fn).sum(), fn.min(), fn.max()], np.float64)
    return out


def full2048(deform_mode="reference_cpu"):
    """BASELINE configs[4] geometry (2048x2048, B=1) through the torch restatement in fp32 (fp64 does not fit this container's
    62 GiB): every 32nd pixel + global statistics.  fp32-vs-fp32: the GPU test's tolerance is the north-star gate."""
    cfg = BiRefNetConfig(deform_mode=deform_mode)
    tag = "ref" if deform_mode == "reference_cpu" else "def"
    w = synth_weights(birefnet_weight_spec(cfg), seed=42)
    x = synth_input(1, 2048, 2048)
    t = time.time()
    with torch.no_grad():
        y = R.forward_logits(x, w, cfg, torch.float32)
    print("2048 fp32 restatement", f"{time.time() - t:.1f}s", flush=True)
    yn = y.numpy().astype(np.float64)
    return {f"m2048_full_{tag}_s32": yn[:, :, ::32, ::32].astype(np.float32),
            f"m2048_full_{tag}_stats": np.array([yn.sum(), np.abs(yn).sum(), yn.min(), yn.max()], np.float64)}


def weights_checksum():
    names = [("bb.layers.0.blocks.0.attn.qkv.weight", (576, 192), "lin_w"), ("decoder.conv_out1.0.weight", (1, 240, 1, 1), "conv_w"),
             ("bb.layers.3.blocks.1.attn.relative_position_bias_table", (529, 48), "rel_bias"),
             ("squeeze_module.0.bn_in.running_var", (64,), "bn_var")]
    return np.array([float(np.asarray(synth_tensor(n, s, k, 42), np.float64).sum()) for n, s, k in names] +
                    [float(np.asarray(synth_input(1, 8, 8), np.float64).sum())], np.float64)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full1024", action="store_true")
    ap.add_argument("--full2048", action="store_true")
    ap.add_argument("--deformable", action="store_true", help="with --full1024 / --full2048: deform_mode=deformable -> model_<S>_def.npz")
    a = ap.parse_args()
    dm, sfx = ("deformable", "_def") if a.deformable else ("reference_cpu", "")
    torch.set_num_threads(os.cpu_count() or 1)
    if a.full2048:
        np.savez_compressed(os.path.join(HERE, f"model_2048{sfx}.npz"), **full2048(dm))
    elif a.full1024:
        np.savez_compressed(os.path.join(HERE, f"model_1024{sfx}.npz"), **full1024(dm))
    else:
        np.savez_compressed(os.path.join(HERE, "kats.npz"), **kats())
        g = model_goldens()
        g["weights_checksum"] = weights_checksum()
        np.savez_compressed(os.path.join(HERE, "models_small.npz"), **g)
    print("done")
