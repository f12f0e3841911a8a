"""Op-level parity on the GPU: every C-ABI op entry point against a CPU restatement of the candle op it replaces.
Tolerances are fp32 reorder noise (the MFMA f32 path is an exact fmaf chain): 2e-5 relative to the output scale."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import torch_ref as R

pytestmark = pytest.mark.gpu


def _close(a, b, tol=3e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e} (scale {scale:.3g})"


def rnd(*shape, seed=0, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * std).astype(np.float32)


@pytest.mark.parametrize("M,K,N", [(300, 192, 576), (1000, 64, 512), (129, 768, 200), (64, 32, 16), (5000, 384, 1536), (7, 96, 130)])
@pytest.mark.parametrize("act", [None, "gelu_erf", "relu"])
def test_linear(gpu, M, K, N, act):
    from candle_birefnet_amd import ops
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    y = ops.linear(x, w, b, act=act, residual=r)
    ref = torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()
    if act == "gelu_erf":
        ref = F.gelu(ref)
    elif act == "relu":
        ref = F.relu(ref)
    ref = ref + torch.from_numpy(r).double()
    _close(y, ref.numpy())


def test_linear_device_tensors(gpu):
    from candle_birefnet_amd import ops
    x, w = rnd(512, 192, seed=5), rnd(384, 192, seed=6, std=0.07)
    xd = torch.from_numpy(x).cuda()
    y = ops.linear(xd, w)
    assert y.is_cuda
    _close(y.cpu().numpy(), x.astype(np.float64) @ w.astype(np.float64).T)


@pytest.mark.parametrize("rows,C", [(10, 192), (1000, 384), (33, 768), (5, 1536), (7, 3072), (3, 64)])
def test_layer_norm(gpu, rows, C):
    from candle_birefnet_amd import ops
    x, g, b = rnd(rows, C, seed=1, std=3.0) + 0.5, 1 + rnd(C, seed=2, std=0.1), rnd(C, seed=3, std=0.1)
    y = ops.layer_norm(x, g, b)
    ref = F.layer_norm(torch.from_numpy(x).double(), (C,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), 1e-5)
    _close(y, ref.numpy())


@pytest.mark.parametrize("B,C,H,W,O,k,s,p,d", [
    (2, 64, 16, 16, 256, 3, 1, 1, 1),    # ASPP k3
    (1, 64, 20, 12, 256, 7, 1, 3, 1),    # ASPP k7
    (1, 96, 9, 11, 64, 3, 1, 1, 1),      # conv_in-like, ragged map
    (2, 3, 32, 32, 192, 4, 4, 0, 1),     # PatchEmbed.proj (gather from NCHW)
    (1, 3, 30, 27, 64, 3, 1, 1, 1),      # ipt_blk1.conv1 (gather), ragged
    (1, 64, 14, 14, 128, 3, 2, 1, 1),    # strided (DeformableConv2d surface)
    (1, 64, 24, 24, 256, 3, 1, 6, 6),    # dilated (dead-code ASPP, D2)
    (2, 128, 8, 8, 48, 1, 1, 0, 1),      # 1x1
    (1, 48, 8, 8, 64, 3, 1, 1, 1),       # Cin not a multiple of 32 -> gather path
])
def test_conv2d(gpu, B, C, H, W, O, k, s, p, d):
    from candle_birefnet_amd import ops
    x, w, b = rnd(B, C, H, W, seed=1), rnd(O, C, k, k, seed=2, std=(C * k * k) ** -0.5), rnd(O, seed=3)
    y = ops.conv2d(x, w, b, stride=s, padding=p, dilation=d)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=p, dilation=d)
    _close(y, ref.numpy())


def test_conv2d_bn_relu(gpu):
    from candle_birefnet_amd import ops
    B, C, H, W, O = 2, 64, 12, 12, 64
    x, w, b = rnd(B, C, H, W, seed=1), rnd(O, C, 3, 3, seed=2, std=0.04), rnd(O, seed=3)
    g, be, m, v = 1 + rnd(O, seed=4, std=0.1), rnd(O, seed=5, std=0.1), rnd(O, seed=6, std=0.1), np.random.default_rng(7).random(O).astype(np.float32) + 0.5
    y = ops.conv2d(x, w, b, padding=1, bn=(g, be, m, v), act="relu")
    t = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)
    ref = F.relu(F.batch_norm(t, torch.from_numpy(m).double(), torch.from_numpy(v).double(), torch.from_numpy(g).double(),
                              torch.from_numpy(be).double(), False, 0.0, 1e-5))
    _close(y, ref.numpy())


@pytest.mark.parametrize("H,W,oh,ow", [(5, 5, 9, 9), (9, 9, 5, 5), (4, 4, 4, 4), (16, 16, 32, 32), (64, 48, 2, 3), (7, 3, 1, 1)])
def test_upsample_bilinear(gpu, H, W, oh, ow):
    from candle_birefnet_amd import ops
    x = rnd(2, 5, H, W, seed=1)
    y = ops.upsample_bilinear2d(x, oh, ow)
    ref = F.interpolate(torch.from_numpy(x).double(), size=(oh, ow), mode="bilinear", align_corners=True)
    _close(y, ref.numpy(), tol=1e-5)


def _attn_weights(C, heads, seed):
    return {"attn.qkv.weight": rnd(3 * C, C, seed=seed, std=C ** -0.5), "attn.qkv.bias": rnd(3 * C, seed=seed + 1, std=0.2),
            "attn.proj.weight": rnd(C, C, seed=seed + 2, std=C ** -0.5), "attn.proj.bias": rnd(C, seed=seed + 3, std=0.02),
            "attn.relative_position_bias_table": rnd(529, heads, seed=seed + 4, std=0.5)}


@pytest.mark.parametrize("B,H,W,heads,shift", [
    (1, 12, 12, 2, 0), (1, 12, 12, 2, 6),        # one window
    (2, 24, 24, 3, 0), (2, 24, 24, 3, 6),        # 4 windows, no padding
    (1, 16, 16, 2, 0), (1, 16, 16, 2, 6),        # R=16 -> padded to 24 (125 % pad tokens)
    (1, 32, 20, 6, 6),                           # ragged, Swin-L stage-0 head count
    (1, 4, 4, 1, 6), (1, 1, 1, 1, 0),            # tiny maps (half-scale stage 3 at small inputs)
])
def test_window_attention(gpu, B, H, W, heads, shift):
    from candle_birefnet_amd import ops
    C = heads * 32
    w = _attn_weights(C, heads, seed=10)
    x = rnd(B, H, W, C, seed=99)
    y = ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                             w["attn.relative_position_bias_table"])
    ref = R.window_attention_block(torch.from_numpy(x).double(), w, "", heads, 12, shift, torch.float64)
    _close(y, ref.numpy())


def test_window_attention_swinl_stage_shapes(gpu):
    """the Swin-L shapes of test_flash_bias.rs:155-158,218-224: stage 0, 484 windows x 6 heads, shifted and not; tolerance
    there was 0.1 between two GPU paths, here fp32 reorder noise against fp64."""
    from candle_birefnet_amd import ops
    heads, C = 6, 192
    w = _attn_weights(C, heads, seed=20)
    x = rnd(1, 256, 256, C, seed=7)
    for shift in (0, 6):
        y = ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"],
                                 w["attn.proj.bias"], w["attn.relative_position_bias_table"])
        ref = R.window_attention_block(torch.from_numpy(x), w, "", heads, 12, shift, torch.float32)
        _close(y, ref.numpy(), tol=1e-4)
        assert np.abs(y).sum() > 1.0   # "not all zeros" check of test_flash_bias.rs:60-61


@pytest.mark.parametrize("B,H,W,C", [(1, 6, 6, 32), (2, 8, 8, 192), (1, 7, 5, 64), (1, 1, 1, 384)])
def test_patch_merging(gpu, B, H, W, C):
    from candle_birefnet_amd import ops
    w = {"norm.weight": 1 + rnd(4 * C, seed=1, std=0.1), "norm.bias": rnd(4 * C, seed=2, std=0.1),
         "reduction.weight": rnd(2 * C, 4 * C, seed=3, std=(4 * C) ** -0.5)}
    x = rnd(B, H * W, C, seed=4)
    y = ops.patch_merging(x, H, W, w["norm.weight"], w["norm.bias"], w["reduction.weight"])
    ref = R.patch_merging(torch.from_numpy(x).double(), H, W, w, "", torch.float64)
    _close(y, ref.numpy())


@pytest.mark.parametrize("k,stride,pad,O,H", [(3, 1, 1, 128, 32), (1, 1, 0, 256, 8), (7, 1, 3, 256, 12), (3, 2, 1, 64, 16)])
@pytest.mark.parametrize("mode", ["reference_cpu", "deformable"])
def test_deform_conv2d(gpu, k, stride, pad, O, H, mode):
    """test_deform_conv.rs:24-82 (64->128, k3, s1, p1 on [1,64,32,32], shape assertion) + numeric parity in both modes."""
    import candle_birefnet_amd as cb
    C = 64
    t = {"offset_conv.weight": rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": rnd(2 * k * k, seed=2, std=0.3),
         "modulator_conv.weight": rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), "modulator_conv.bias": rnd(k * k, seed=4, std=0.1),
         "regular_conv.weight": rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), "regular_conv.bias": rnd(O, seed=6, std=0.1)}
    layer = cb.DeformableConv2d.new(C, O, k, stride, pad, cb.VarBuilder.from_tensors(t), mode=mode)
    x = rnd(1, C, H, H, seed=9)
    y = layer.forward(x)
    Ho = (H + 2 * pad - k) // stride + 1
    assert y.shape == (1, O, Ho, Ho)
    xt = torch.from_numpy(x).double()
    td = {n: torch.from_numpy(a).double() for n, a in t.items()}
    if mode == "reference_cpu":
        ref = F.conv2d(xt, td["regular_conv.weight"], td["regular_conv.bias"], stride=stride, padding=pad)
    else:
        off = F.conv2d(xt, td["offset_conv.weight"], td["offset_conv.bias"], stride=stride, padding=pad)
        msk = 1.0 / (torch.exp(-F.conv2d(xt, td["modulator_conv.weight"], td["modulator_conv.bias"], stride=stride, padding=pad)) + 1.0) * 2.0
        ref = R.deform_conv2d(xt, off, msk, td["regular_conv.weight"], td["regular_conv.bias"], stride, pad)
    _close(y, ref.numpy(), tol=1e-4 if mode == "deformable" else 3e-5)


@pytest.mark.parametrize("mode,tol", [("f32_split3", 3e-5), ("f32_split2", 1e-4), ("f32_half2", 3e-5)])
@pytest.mark.parametrize("M,K,N", [(300, 192, 576), (129, 768, 200), (5000, 384, 1536), (7, 96, 130), (1024, 3072, 768), (64, 51840, 64)])
def test_linear_split_modes(gpu, mode, tol, M, K, N):
    """the split-bf16 contraction kernels (all tile configs incl. the warp-specialised one and split-K) against fp64"""
    from candle_birefnet_amd import ops
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ops.set_compute(mode)
    try:
        y = ops.linear(x, w, b, act="gelu_erf", residual=r)
    finally:
        ops.set_compute("f32")
    ref = F.gelu(torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()) + torch.from_numpy(r).double()
    _close(y, ref.numpy(), tol=tol)


@pytest.mark.parametrize("mode,tol", [("f32_split3", 3e-5), ("f32_split2", 1e-4), ("f32_half2", 3e-5)])
@pytest.mark.parametrize("B,C,H,W,O,k,p", [(2, 64, 16, 16, 256, 3, 1), (1, 64, 20, 12, 256, 7, 3), (1, 96, 33, 31, 64, 3, 1), (1, 480, 64, 64, 64, 3, 1)])
def test_conv2d_split_modes(gpu, mode, tol, B, C, H, W, O, k, p):
    from candle_birefnet_amd import ops
    x, w, b = rnd(B, C, H, W, seed=1), rnd(O, C, k, k, seed=2, std=(C * k * k) ** -0.5), rnd(O, seed=3)
    ops.set_compute(mode)
    try:
        y = ops.conv2d(x, w, b, padding=p)
    finally:
        ops.set_compute("f32")
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=p)
    _close(y, ref.numpy(), tol=tol)


@pytest.mark.parametrize("astd", [1e-3, 0.03, 1.0, 300.0])
@pytest.mark.parametrize("M,K,N", [(512, 768, 384), (256, 3072, 256), (200, 96, 64)])
def test_half2_is_fp32_class_at_every_magnitude(gpu, astd, M, K, N):
    """Mode f32_half2 (two fp16 planes of the power-of-two-scaled operands, 3 MFMAs per product): the error of a Linear against fp64, relative
    to the largest output, is that of an fp32 GEMM (a few 1e-7: accumulation noise) for activations of magnitude 0.03 ... 300 and weights
    of the usual 0.02 scale with one large outlier (the per-tensor weight scale is chosen by the largest weight) — never worse than twice
    mode f32_split3's on the same data plus 2e-7.  Below |8 x| = 2^-3 the low plane is a subnormal fp16 (absolute step 2^-24 / 8 per
    element): activations of magnitude 1e-3 measure 1.8e-6, twice f32_split3 and a quarter of f32_split2 — bounded at 4e-6.  This is also
    where flushed fp16 subnormals in the matrix instruction would show: they would cost 1e-4 here (the MFMA does not flush them)."""
    from candle_birefnet_amd import ops
    x, w, b = rnd(M, K, seed=21, std=astd), rnd(N, K, seed=22, std=0.02), rnd(N, seed=23, std=0.1 * astd)
    w[3, 5] = 0.9
    ref = (torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()).numpy()
    err = {}
    for mode in ("f32_split3", "f32_half2", "f32_split2"):
        ops.set_compute(mode)
        try:
            y = ops.linear(x, w, b)
        finally:
            ops.set_compute("f32")
        err[mode] = float(np.abs(y.astype(np.float64) - ref).max() / np.abs(ref).max())
    print(f"|A| ~ {astd:g}, K {K}: relative-to-max error  f32_split3 {err['f32_split3']:.2e}  f32_half2 {err['f32_half2']:.2e}  f32_split2 {err['f32_split2']:.2e}")
    if astd >= 0.03:
        assert err["f32_half2"] <= 2.0 * err["f32_split3"] + 2e-7, err
        assert err["f32_half2"] < 1.5e-6, err
    else:
        assert err["f32_half2"] < 4e-6 and err["f32_half2"] < 0.5 * err["f32_split2"], err


def test_half2_out_of_range_is_loud(gpu):
    """an activation beyond the fp16 range of mode f32_half2 (|8 x| >= 65520) gives a non-finite output row, never a wrong finite one; the
    other rows are untouched"""
    from candle_birefnet_amd import ops
    x, w, b = rnd(64, 96, seed=31), rnd(32, 96, seed=32, std=0.1), rnd(32, seed=33)
    ops.set_compute("f32_half2")
    try:
        y0 = ops.linear(x, w, b)
        x2 = x.copy()
        x2[7, 11] = 9000.0
        y1 = ops.linear(x2, w, b)
    finally:
        ops.set_compute("f32")
    assert np.isfinite(y0).all()
    assert not np.isfinite(y1[7]).any()
    np.testing.assert_array_equal(np.delete(y1, 7, axis=0), np.delete(y0, 7, axis=0))


@pytest.mark.parametrize("mode,tol", [("f32_split2", 1e-4), ("f32_half2", 2e-5)])
@pytest.mark.parametrize("B,H,W,heads,shift", [(1, 12, 12, 2, 0), (2, 24, 24, 3, 6), (1, 16, 16, 2, 6), (1, 32, 20, 6, 6), (1, 4, 4, 1, 6), (1, 64, 64, 24, 6)])
def test_window_attention_split_modes(gpu, mode, tol, B, H, W, heads, shift):
    """the bf16-split attention kernel (window_attention_split_kernel) incl. pad tokens, shift mask, odd geometries"""
    from candle_birefnet_amd import ops
    C = heads * 32
    w = _attn_weights(C, heads, seed=10)
    x = rnd(B, H, W, C, seed=99)
    ops.set_compute(mode)
    try:
        y = ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                 w["attn.relative_position_bias_table"])
    finally:
        ops.set_compute("f32")
    ref = R.window_attention_block(torch.from_numpy(x).double(), w, "", heads, 12, shift, torch.float64)
    _close(y, ref.numpy(), tol=tol)


@pytest.mark.parametrize("mode", ["f32_split2", "f32_split3", "f32_half2"])
def test_split_conv_exact_and_repeatable(gpu, mode):
    """Small-integer data is exact in every bf16 plane, so the split conv must reproduce the integer result bit for bit,
    every time.  This is the reproducer of a rare wrong-rows fault (packed-fp32 instructions in the staging waves while
    MFMA waves share their SIMD: 2 rows x 128 columns wrong a few times per 10^5 K tiles); 98 K tiles x 1024 workgroups
    per run made it show in every run before the fix."""
    from candle_birefnet_amd import ops
    B, Cin, H, W, Cout, k, pad = 1, 64, 256, 256, 256, 7, 3
    c = np.arange(Cin)[:, None, None]
    x = np.broadcast_to((c % 32 + 1).astype(np.float32), (Cin, H, W))[None].copy()
    w = np.ones((Cout, Cin, k, k), np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), padding=pad).numpy()
    ops.set_compute(mode)
    try:
        xd = torch.from_numpy(x).cuda()
        for _ in range(4):
            y = ops.conv2d(xd, w, np.zeros(Cout, np.float32), stride=1, padding=pad).cpu().numpy()
            wrong = int((y.astype(np.float64) != ref).sum())
            assert wrong == 0, f"{wrong} wrong elements (max |err| {np.abs(y - ref).max()})"
    finally:
        ops.set_compute("f32")


# ---- compute mode BRN_BF16 (kernels/gemm_bf16.hip): bf16 operands in HBM, fp32 accumulation ------------------------------
ATT_BF16_TOL = 1.0e-2      # relative to max |ref|: 3x the largest measured on MI355X over _ATT_BF16_CASES (3.0e-3 .. 3.3e-3 of the scale)


def _bf16_round(a, s16="bf16"):
    """round-to-nearest-even fp32 -> bf16 (compute mode bf16) or fp16 (compute mode f16) -> fp32, as the library rounds operands (weights
    at load, activations at the edge)"""
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.float16 if s16 == "f16" else torch.bfloat16).to(torch.float64).numpy()


S16_ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}      # one unit in the last place of the mode's stored values, relative


@pytest.mark.parametrize("M,K,N", [(300, 192, 576), (1000, 64, 512), (129, 768, 200), (64, 32, 16), (5000, 384, 1536), (7, 96, 130),
                                   (4096, 3072, 768), (513, 160, 64), (33000, 192, 192)])
@pytest.mark.parametrize("act", [None, "gelu_erf"])
@pytest.mark.parametrize("s16", ["bf16", "f16"])
def test_linear_bf16_mode(gpu, M, K, N, act, s16):
    """the linear entry keeps y and the residual fp32 in this mode, so the reference is EXACT up to fp32 accumulation order:
    operands rounded to bf16, products and sums in fp64.  K % 64 == 32 (the zero-page K tail), ragged M / N, split-K free."""
    from candle_birefnet_amd import ops
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ops.set_compute(s16)
    try:
        y = ops.linear(x, w, b, act=act, residual=r)
    finally:
        ops.set_compute("f32")
    ref = torch.from_numpy(_bf16_round(x, s16) @ _bf16_round(w, s16).T) + torch.from_numpy(b).double()
    if act == "gelu_erf":
        ref = F.gelu(ref)
    ref = ref + torch.from_numpy(r).double()
    _close(y, ref.numpy(), tol=2e-5)


@pytest.mark.parametrize("B,C,H,W,O,k,p", [(2, 64, 16, 16, 256, 3, 1), (1, 64, 20, 12, 256, 7, 3), (1, 96, 9, 11, 64, 3, 1), (1, 480, 12, 12, 64, 3, 1),
                                           (1, 64, 33, 17, 192, 3, 1), (2, 3456, 8, 8, 64, 3, 1), (1, 64, 24, 24, 16, 3, 1),
                                           (1, 128, 20, 20, 64, 3, 1), (2, 192, 15, 17, 192, 3, 1), (1, 960, 12, 12, 64, 3, 1), (1, 128, 13, 13, 256, 7, 3),
                                           (1, 1920, 16, 16, 64, 3, 1)])
@pytest.mark.parametrize("s16", ["bf16", "f16"])
def test_conv2d_bf16_mode(gpu, B, C, H, W, O, k, p, s16):
    """implicit-GEMM conv on bf16 maps: zero padding through out-of-range buffer offsets (answered with zeros), Cin = 480 (K steps that straddle taps, K % 64 = 32),
    ragged maps, the tall-K split-K plan (3456 x 9), N = 16 (scalar stores); Cin = 128 / 192 / 960 / 1920 / 3456 run the chunk-major K
    order (64-channel chunk, tap, channel), 3x3 and 7x7, with and without split-K.  The output map is bf16: tolerance = one bf16 ulp."""
    from candle_birefnet_amd import ops
    x, w, b = rnd(B, C, H, W, seed=1), rnd(O, C, k, k, seed=2, std=(C * k * k) ** -0.5), rnd(O, seed=3, std=0.1)
    ops.set_compute(s16)
    try:
        y = ops.conv2d(x, w, b, padding=p, act="relu")
    finally:
        ops.set_compute("f32")
    ref = F.relu(F.conv2d(torch.from_numpy(_bf16_round(x, s16)), torch.from_numpy(_bf16_round(w, s16)), torch.from_numpy(b).double(), padding=p)).numpy()
    err = np.abs(np.asarray(y, np.float64) - ref)
    assert (err <= S16_ULP[s16] * np.abs(ref) + 1e-6).all(), f"max abs err {err.max():.3e}"


@pytest.mark.parametrize("k,stride,pad,O,H,C", [(3, 1, 1, 128, 32, 64), (1, 1, 0, 256, 8, 64), (7, 1, 3, 256, 12, 64), (3, 2, 1, 64, 16, 64), (3, 1, 1, 256, 19, 128),
                                                (1, 1, 0, 256, 40, 64), (7, 1, 3, 32, 9, 64)])
@pytest.mark.parametrize("s16", ["bf16", "f16"])
def test_deform_conv2d_bf16_mode(gpu, k, stride, pad, O, H, C, s16):
    """kernels/deform_bf16.hip (the bf16-MFMA gather of compute mode bf16) against an EXACT-operand reference: x and all three
    weights rounded to bf16, offsets / modulator / bilinear sampling / contraction in fp64 (the torchvision semantics of
    tests/torch_ref.py, aspp.rs:77-164).  What the kernel rounds on top: the sampled column (mask x bilinear) to bf16 before the MFMA,
    the offset conv's fp32 accumulation, the bf16 output map.  Covers stride 2, ragged maps (19, 9: pixel tiles with rows >= M),
    Cin = 128 (two K steps per tap), N = 32 / 64 / 128 (partly filled 256-column tiles), samples outside the image.
    Tolerance: 2^-8 |ref| + 4e-3 max|ref|.  Measured on MI355X (round 4, gemm_deform_bf16_v2_kernel; bit-identical to v1): max abs err
    6.1e-3 ... 2.1e-2 on outputs of max magnitude 2.5 ... 6.6, i.e. 2.4e-3 ... 3.7e-3 of the scale — what ONE bf16 rounding of each of the
    K = 64 k^2 sampled-column elements (2^-9 relative each, random sign) leaves on a sum of that many terms; the bound is 1.3 - 2 x that."""
    import candle_birefnet_amd as cb
    from candle_birefnet_amd import ops
    t = {"offset_conv.weight": rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": rnd(2 * k * k, seed=2, std=0.3),
         "modulator_conv.weight": rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), "modulator_conv.bias": rnd(k * k, seed=4, std=0.1),
         "regular_conv.weight": rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), "regular_conv.bias": rnd(O, seed=6, std=0.1)}
    layer = cb.DeformableConv2d.new(C, O, k, stride, pad, cb.VarBuilder.from_tensors(t), mode="deformable")
    x = rnd(2, C, H, H, seed=9)
    ops.set_compute(s16)
    try:
        y = layer.forward(x)
        y2 = layer.forward(x)
    finally:
        ops.set_compute("f32")
    np.testing.assert_array_equal(y, y2)
    xt = torch.from_numpy(_bf16_round(x, s16))
    td = {n: torch.from_numpy(_bf16_round(a, s16) if n.endswith("weight") else np.asarray(a, np.float64)) for n, a in t.items()}
    off = F.conv2d(xt, td["offset_conv.weight"], td["offset_conv.bias"], stride=stride, padding=pad)
    msk = 1.0 / (torch.exp(-F.conv2d(xt, td["modulator_conv.weight"], td["modulator_conv.bias"], stride=stride, padding=pad)) + 1.0) * 2.0
    ref = R.deform_conv2d(xt, off, msk, td["regular_conv.weight"], td["regular_conv.bias"], stride, pad).numpy()
    err = np.abs(np.asarray(y, np.float64) - ref)
    # sampled columns carry <= 2^-9 relative rounding each (independent over K = C k^2 terms), the output one bf16 rounding
    scale = np.abs(ref).max()
    # (compute mode f16, fp16 storage: the same kernel in namespace brn::hf, every bound 8 x tighter)
    assert (err <= S16_ULP[s16] * np.abs(ref) + (4e-3 if s16 == "bf16" else 5e-4) * scale).all(), f"max abs err {err.max():.3e} (|ref| max {scale:.2f})"
    print(f"deform {s16} k{k} s{stride} O{O} H{H} C{C}: max abs err {err.max():.2e}, |ref| max {scale:.2f}")


_ATT_BF16_CASES = [(1, 12, 12, 2, 0), (2, 24, 24, 3, 6), (1, 16, 16, 2, 6), (1, 16, 16, 2, 0), (1, 32, 20, 6, 6), (1, 4, 4, 1, 6), (1, 64, 64, 24, 6), (2, 36, 36, 12, 6)]


def _att_bf16_reference(B, H, W, heads, shift, qk_gain=1.0, s16="bf16"):
    """exact-operand reference of window_attention_bf16_kernel between its two GEMMs: x and the weights rounded to bf16, the qkv
    matrix and the attention output rounded to bf16 where the mode stores them in HBM, everything else (scores, bias, mask, softmax,
    PV, proj) in fp64.  What the kernel adds: fp32 accumulation, softmax numerators rounded to bf16 for the PV MFMA, exp2 on
    log2-scaled scores."""
    C = heads * 32
    w = _attn_weights(C, heads, seed=10)
    if qk_gain != 1.0:       # q and k rows of the qkv Linear scaled: the scores grow by qk_gain^2
        w["attn.qkv.weight"] = w["attn.qkv.weight"].copy(); w["attn.qkv.bias"] = w["attn.qkv.bias"].copy()
        w["attn.qkv.weight"][:2 * C] *= np.float32(qk_gain); w["attn.qkv.bias"][:2 * C] *= np.float32(qk_gain)
    x = rnd(B, H, W, C, seed=99)
    wr = {n: (_bf16_round(a, s16) if n in ("attn.qkv.weight", "attn.proj.weight") else np.asarray(a, np.float64)) for n, a in w.items()}
    store = lambda t: t.to(torch.float16 if s16 == "f16" else torch.bfloat16).to(torch.float64)
    ref = R.window_attention_block(torch.from_numpy(_bf16_round(x, s16)), wr, "", heads, 12, shift, torch.float64, store=store).numpy()
    return x, w, ref


@pytest.mark.parametrize("B,H,W,heads,shift", _ATT_BF16_CASES)
@pytest.mark.parametrize("s16", ["bf16", "f16"])
def test_window_attention_bf16_mode(gpu, B, H, W, heads, shift, s16):
    """op-level parity of window_attention_bf16_kernel (compute mode bf16; swin.rs:266-312 on bf16 qkv): one window, 4 windows with the
    shift mask, R = 16 padded to 24 (pad tokens synthesised from the qkv bias, shifted and not), a ragged map, tiny maps, the stage-2
    head count (24) and the stage-1 geometry (36 = 3 windows a side)."""
    from candle_birefnet_amd import ops
    x, w, ref = _att_bf16_reference(B, H, W, heads, shift, s16=s16)
    ops.set_compute(s16)
    try:
        y = ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                 w["attn.relative_position_bias_table"])
        y2 = ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                  w["attn.relative_position_bias_table"])
    finally:
        ops.set_compute("f32")
    np.testing.assert_array_equal(y, y2)
    err = np.abs(np.asarray(y, np.float64) - ref)
    scale = np.abs(ref).max()
    print(f"attention {s16} B{B} {H}x{W} h{heads} s{shift}: max abs err {err.max():.2e}, |ref| max {scale:.2f}")
    assert err.max() <= ATT_BF16_TOL * scale * (1.0 if s16 == "bf16" else 0.125)


@pytest.mark.parametrize("B,H,W,heads", [(2, 24, 24, 2), (1, 32, 20, 6), (1, 36, 36, 4)])
def test_window_attention_bf16_mode_shift_mask_is_exactly_minus_100(gpu, B, H, W, heads):
    """the shift mask adds the reference's constant -100 (swin.rs:283-296, 651) whatever the two regions are (the kernel
    once added -100 x a region distance of 1 .. 15).  That only shows when a masked key's score exceeds the unmasked ones by about 100:
    q and k are scaled so that scores have a standard deviation of ~60 and a few per cent of the query rows of the edge windows are in
    that regime; same exact-operand reference and tolerance as test_window_attention_bf16_mode."""
    from candle_birefnet_amd import ops
    x, w, ref = _att_bf16_reference(B, H, W, heads, 6, qk_gain=7.75)
    # the case must actually discriminate: the same reference with the mask at -200 differs by far more than the tolerance
    C = heads * 32
    wr = {n: (_bf16_round(a) if n in ("attn.qkv.weight", "attn.proj.weight") else np.asarray(a, np.float64)) for n, a in w.items()}
    store = lambda t: t.to(torch.bfloat16).to(torch.float64)
    Hp, Wp = -(-H // 12) * 12, -(-W // 12) * 12
    m200 = R.attn_mask(Hp, Wp, 12, 6, torch.float64) * 2.0
    ref200 = R.window_attention_block(torch.from_numpy(_bf16_round(x)), wr, "", heads, 12, 6, torch.float64, mask=m200, store=store).numpy()
    scale = np.abs(ref).max()
    assert np.abs(ref200 - ref).max() > 10 * ATT_BF16_TOL * scale, "the scores of this case never reach the masked regime"
    ops.set_compute("bf16")
    try:
        y = ops.window_attention(x, heads, 6, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                 w["attn.relative_position_bias_table"])
    finally:
        ops.set_compute("f32")
    err = np.abs(np.asarray(y, np.float64) - ref)
    print(f"attention bf16, large scores, B{B} {H}x{W} h{heads}: max abs err {err.max():.2e}, |ref| max {scale:.2f}; -200 mask would differ by {np.abs(ref200 - ref).max():.2e}")
    assert err.max() <= ATT_BF16_TOL * scale


def test_window_attention_bf16_two_heads_per_workgroup(gpu, tmp_path):
    """both instantiations of window_attention_bf16_kernel — two heads per workgroup (the default for an even head count) and one
    (BRN_ATT_HPW=1; the switch is read once per process, hence the child processes): same cases, same reference, bit-equal results"""
    import subprocess, sys, os
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from candle_birefnet_amd import ops\nimport test_ops_gpu as T\n"
        "ops.set_compute('bf16')\nouts = {}\n"
        "for i, (B, H, W, heads, shift) in enumerate(T._ATT_BF16_CASES):\n"
        "    if heads & 1: continue\n"
        "    x, w, ref = T._att_bf16_reference(B, H, W, heads, shift)\n"
        "    y = ops.window_attention(x, heads, shift, w['attn.qkv.weight'], w['attn.qkv.bias'], w['attn.proj.weight'], w['attn.proj.bias'], w['attn.relative_position_bias_table'])\n"
        "    assert np.abs(np.asarray(y, np.float64) - ref).max() <= T.ATT_BF16_TOL * np.abs(ref).max(), (B, H, W, heads, shift)\n"
        "    outs[str(i)] = np.asarray(y)\n"
        "np.savez(sys.argv[1], **outs)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for hpw in ("1", "2"):
        out = str(tmp_path / f"hpw{hpw}.npz")
        pr = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, BRN_ATT_HPW=hpw), capture_output=True, text=True, timeout=600)
        assert pr.returncode == 0, pr.stderr[-2000:]
        res[hpw] = np.load(out)
    assert sorted(res["1"].files) == sorted(res["2"].files) and len(res["1"].files) >= 5
    for k in res["1"].files:
        np.testing.assert_array_equal(res["1"][k], res["2"][k])


@pytest.mark.parametrize("C,H,W,O,act", [(192, 192, 192, 576, None), (192, 181, 183, 768, "gelu_erf"), (192, 192, 171, 192, "relu"), (192, 256, 256, 1152, "gelu_erf"),
                                         (384, 192, 192, 1152, None), (384, 181, 183, 1536, "gelu_erf"), (384, 192, 171, 384, "relu"), (384, 200, 203, 192, None)])
def test_short_k_weight_stationary_gemm_bf16_mode(gpu, C, H, W, O, act):
    """gemm_wstat_bf16_kernel (K = 192 with 64-row tiles, K = 384 with 32-row tiles; N % 192 == 0, M >= 32768: the stage-0 / stage-1
    qkv and fc1 GEMMs at batch >= 4) through a 1x1 conv on a bf16 map: full and ragged row tiles (M % 64, M % 32 != 0), 1 ... 8 column
    groups, bias, GELU / ReLU.  Exact-operand reference; the output map is bf16: one bf16 ulp + 1e-5 (fp32 accumulation; the bf16-output
    GELU, 2^P(|x|) with P of degree 5, is within 3e-6 of erf-GELU: the erfc series it replaced needed 4e-5 here)."""
    from candle_birefnet_amd import ops
    x, w, b = rnd(1, C, H, W, seed=1), rnd(O, C, 1, 1, seed=2, std=C ** -0.5), rnd(O, seed=3, std=0.1)
    ops.set_compute("bf16")
    try:
        y = ops.conv2d(x, w, b, act=act)
        y2 = ops.conv2d(x, w, b, act=act)
    finally:
        ops.set_compute("f32")
    np.testing.assert_array_equal(y, y2)
    ref = F.conv2d(torch.from_numpy(_bf16_round(x)), torch.from_numpy(_bf16_round(w)), torch.from_numpy(b).double())
    ref = (F.gelu(ref) if act == "gelu_erf" else F.relu(ref) if act == "relu" else ref).numpy()
    err = np.abs(np.asarray(y, np.float64) - ref)
    assert (err <= 2.0 ** -8 * np.abs(ref) + 1e-5).all(), f"max abs err {err.max():.3e}"


def test_conv2d_nan_stays_local(gpu):
    """masked (zero-padded) taps are zeroed by a bit mask / zero page, not by 0 * x: a non-finite input pixel spreads
    only over its receptive field, like candle's conv2d; before, Inf at pixel (0,0) turned every border output into NaN."""
    from candle_birefnet_amd import ops
    x, w = rnd(1, 64, 12, 12, seed=1), rnd(32, 64, 3, 3, seed=2, std=0.05)
    x[0, 3, 0, 0] = np.inf
    for mode in ("f32", "f32_split3", "f32_split2", "f32_half2", "bf16", "f16"):
        ops.set_compute(mode)
        try:
            y = np.asarray(ops.conv2d(x, w, None, padding=1))
        finally:
            ops.set_compute("f32")
        bad = ~np.isfinite(y)
        assert bad[:, :, :2, :2].any() and not bad[:, :, 2:, :].any() and not bad[:, :, :, 2:].any(), mode


def test_op_level_shape_fuzz(gpu):
    """tools/op_fuzz.py: 50 random conv2d / linear shapes (odd channel counts, 1 ... 7 taps, stride, dilation, ragged tiles, bias /
    activation / residual) in each of the four compute modes against torch fp64 on the operands the mode multiplies: no refused
    launch, no NaN, error within the mode's bound.  (620 cases over three seeds passed when the tool was written.)"""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, os.path.join(root, "tools", "op_fuzz.py"), "50", "11"], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert "50 cases, 0 problems" in pr.stdout, pr.stdout[-3000:]


def test_window_attention_geometry_fuzz(gpu):
    """tools/att_fuzz.py: 30 random (batch, H, W, heads, shift) geometries through the op-level window attention in each compute mode
    (maps from 1 x 1 to 49 x 49: pad tokens, ragged window grids, shifted and not) against the fp64 torch restatement."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, os.path.join(root, "tools", "att_fuzz.py"), "30", "5"], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert "30 cases, 0 problems" in pr.stdout, pr.stdout[-3000:]


def test_deform_conv2d_geometry_fuzz(gpu):
    """tools/deform_fuzz.py: 25 random DeformableConv2d geometries (32 ... 128 input channels, 1 / 3 / 7 taps, stride 1 / 2, padded or not,
    ragged maps, batch 1 ... 3) in both deform modes and each compute mode against the fp64 torch restatement."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pr = subprocess.run([sys.executable, os.path.join(root, "tools", "deform_fuzz.py"), "25", "3"], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert "25 cases, 0 problems" in pr.stdout, pr.stdout[-3000:]


@pytest.mark.parametrize("cin,cout,H,W,mode,use_aspp,compute", [
    (480, 192, 24, 20, "reference_cpu", True, "f32"),        # decoder_block1's widths (birefnet.rs:212) on a small ragged map
    (480, 192, 16, 16, "deformable", True, "f32_split3"),
    (100, 48, 12, 16, "deformable", True, "f32"),            # in_channels off the 32-channel granule (the reference takes any)
    (96, 40, 16, 12, "reference_cpu", False, "f32_split2"),  # DecoderConfig { use_aspp_deformable: false }: dec_att is None (decoder.rs:107-111)
    (1920, 768, 8, 8, "reference_cpu", True, "f32"),         # decoder_block3's widths
])
def test_decblk_forward_vs_oracle(gpu, cin, cout, H, W, mode, use_aspp, compute):
    """brn_decblk_forward = BasicDecBlk::new + forward (decoder.rs:78-141) for blocks other than the squeeze instance, against the
    oracle's restatement of the same lines (oracle/brn_oracle.cpp dec_blk) at the north-star gate."""
    from candle_birefnet_amd import ops
    from candle_birefnet_amd.weights import _decblk, synth_weights
    from oracle import oracle as O
    w = synth_weights(_decblk("blk.", cin, cout), seed=7)
    x = rnd(2, cin, H, W, seed=3)
    ops.set_compute(compute)
    try:
        y = ops.decblk(x, w, cout, mode=mode, prefix="blk.", use_aspp=use_aspp)
    finally:
        ops.set_compute("f32")
    ref = O.decblk(x, w, cout, mode=1 if mode == "deformable" else 0, prefix="blk.", use_aspp=use_aspp).astype(np.float64)
    assert y.shape == (2, cout, H, W)
    err = np.abs(np.asarray(y, np.float64) - ref)
    assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"max abs err {err.max():.3e}"
    assert err.max() <= 2e-4 * max(1.0, np.abs(ref).max())
    if not use_aspp:     # the block without dec_att must not need the ASPP tensors at all
        w2 = {k: v for k, v in w.items() if ".dec_att." not in k}
        np.testing.assert_array_equal(ops.decblk(x, w2, cout, mode=mode, prefix="blk.", use_aspp=False), ops.decblk(x, w, cout, mode=mode, prefix="blk.", use_aspp=False))


def test_decblk_forward_bf16_mode(gpu):
    """the same entry in compute mode bf16 (bf16 maps inside, in_channels padded to the 64-channel chunk): bounded against the oracle"""
    from candle_birefnet_amd import ops
    from candle_birefnet_amd.weights import _decblk, synth_weights
    from oracle import oracle as O
    for cin, cout, mode in ((480, 192, "reference_cpu"), (100, 64, "deformable")):
        w = synth_weights(_decblk("", cin, cout), seed=9)
        x = rnd(1, cin, 24, 24, seed=4)
        ops.set_compute("bf16")
        try:
            y = ops.decblk(x, w, cout, mode=mode)
        finally:
            ops.set_compute("f32")
        ref = O.decblk(x, w, cout, mode=1 if mode == "deformable" else 0).astype(np.float64)
        err = np.abs(np.asarray(y, np.float64) - ref).max()
        print(f"decblk bf16 {cin}->{cout} {mode}: max abs err {err:.2e}, |ref| max {np.abs(ref).max():.2f}")
        assert err <= 3e-2 * max(1.0, np.abs(ref).max())


def test_deform_conv2d_bf16_kernel_versions_bit_equal(gpu, tmp_path):
    """gemm_deform_bf16_v2_kernel (one lane per (pixel, tap) computes the sampling parameters for the wave, corners fetched two K steps
    ahead; the default) against the round-3 kernel (BRN_DEFORM_V=1: every thread computes them itself): the same products in the same
    order, so the same bits — every geometry of test_deform_conv2d_bf16_mode plus taps that do not fill the last block of four, three
    K steps per tap, a tile that spans two images, tiny maps.  Child processes: the switch is read once per process."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import candle_birefnet_amd as cb\nfrom candle_birefnet_amd import ops\nfrom test_ops_gpu import rnd\n"
        "outs = {}\n"
        "cases = [(3, 1, 1, 128, 32, 64, 2), (1, 1, 0, 256, 8, 64, 2), (7, 1, 3, 256, 12, 64, 2), (3, 2, 1, 64, 16, 64, 2), (3, 1, 1, 256, 19, 128, 2),\n"
        "         (1, 1, 0, 256, 40, 64, 1), (7, 1, 3, 32, 9, 64, 3), (5, 1, 2, 64, 11, 192, 1), (3, 1, 1, 256, 6, 64, 5), (7, 1, 3, 256, 33, 64, 1)]\n"
        "for i, (k, stride, pad, O, H, C, B) in enumerate(cases):\n"
        "    t = {'offset_conv.weight': rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), 'offset_conv.bias': rnd(2 * k * k, seed=2, std=0.3),\n"
        "         'modulator_conv.weight': rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), 'modulator_conv.bias': rnd(k * k, seed=4, std=0.1),\n"
        "         'regular_conv.weight': rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), 'regular_conv.bias': rnd(O, seed=6, std=0.1)}\n"
        "    layer = cb.DeformableConv2d.new(C, O, k, stride, pad, cb.VarBuilder.from_tensors(t), mode='deformable')\n"
        "    ops.set_compute('bf16')\n"
        "    outs[str(i)] = np.asarray(layer.forward(rnd(B, C, H, H + (i %% 3), seed=9)))\n"
        "    ops.set_compute('f32')\n"
        "np.savez(sys.argv[1], **outs)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for v in ("1", "2"):
        out = str(tmp_path / f"v{v}.npz")
        pr = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, BRN_DEFORM_V=v), capture_output=True, text=True, timeout=600)
        assert pr.returncode == 0, pr.stderr[-2000:]
        res[v] = np.load(out)
    assert sorted(res["1"].files) == sorted(res["2"].files) and len(res["1"].files) == 10
    for k_ in res["1"].files:
        assert np.isfinite(res["2"][k_]).all()
        np.testing.assert_array_equal(res["1"][k_], res["2"][k_])


@pytest.mark.parametrize("M,K,N", [(8192, 768, 768), (8192 + 37, 768, 768), (20480, 768, 768), (9000, 3072, 768), (16384, 384, 384), (8200, 1536, 384),
                                   (33000, 192, 192), (4096, 768, 768), (8192, 768, 1536)])
def test_linear_residual_layer_norm_bf16_mode(gpu, M, K, N):
    """x = A W^T + b + x; y = LayerNorm(x) (swin.rs:310,406,407) in compute mode bf16: gemm_rowln_bf16_kernel for N = 768 / 384 (64 whole
    rows per workgroup, W streamed through LDS; K = 768 projection, K = 3072 / 1536 = the fc2 shape, ragged last tile, tile counts below
    and above one round of workgroups), gemm_wstat_ln_bf16_kernel for N = 192, GEMM + LayerNorm launches otherwise (M below the fused
    kernel's threshold, N = 1536).  Exact-operand reference: A and W rounded to bf16, everything else fp64; x is an fp32 matrix
    (accumulation noise), y a bf16-rounded one (one ulp + the LayerNorm arithmetic in fp32)."""
    from candle_birefnet_amd import ops
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3, std=0.1)
    r = rnd(M, N, seed=4, std=2.0)
    g, bt = (1.0 + rnd(N, seed=5, std=0.05)).astype(np.float32), rnd(N, seed=6, std=0.02)
    ops.set_compute("bf16")
    try:
        xo, yo = ops.linear_residual_layer_norm(a, w, b, r, g, bt)
        xo2, yo2 = ops.linear_residual_layer_norm(a, w, b, r, g, bt)
    finally:
        ops.set_compute("f32")
    np.testing.assert_array_equal(xo, xo2); np.testing.assert_array_equal(yo, yo2)
    xr = _bf16_round(a) @ _bf16_round(w).T + b.astype(np.float64) + r.astype(np.float64)
    mu = xr.mean(-1, keepdims=True)
    var = ((xr - mu) ** 2).mean(-1, keepdims=True)
    yr = (xr - mu) / np.sqrt(var + 1e-5) * g.astype(np.float64) + bt.astype(np.float64)
    ex = np.abs(np.asarray(xo, np.float64) - xr).max()
    ey = np.abs(np.asarray(yo, np.float64) - yr)
    print(f"linear+residual+LN bf16 M{M} K{K} N{N}: x max abs err {ex:.2e} (|x| max {np.abs(xr).max():.1f}), y max abs err {ey.max():.2e}")
    assert ex <= 2e-5 * max(1.0, np.abs(xr).max())
    assert (ey <= 2.0 ** -8 * np.abs(yr) + 1e-4).all()


def test_linear_residual_layer_norm_fp32_modes(gpu):
    """the same entry in the fp32-class modes (GEMM + LayerNorm launches) against fp64"""
    from candle_birefnet_amd import ops
    M, K, N = 1000, 384, 768
    a, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3, std=0.1), rnd(M, N, seed=4)
    g, bt = (1.0 + rnd(N, seed=5, std=0.05)).astype(np.float32), rnd(N, seed=6, std=0.02)
    xr = a.astype(np.float64) @ w.astype(np.float64).T + b + r
    mu = xr.mean(-1, keepdims=True)
    yr = (xr - mu) / np.sqrt(((xr - mu) ** 2).mean(-1, keepdims=True) + 1e-5) * g + bt
    for mode in ("f32", "f32_split3", "f32_split2", "f32_half2"):
        ops.set_compute(mode)
        try:
            xo, yo = ops.linear_residual_layer_norm(a, w, b, r, g, bt)
        finally:
            ops.set_compute("f32")
        _close(xo, xr, tol=3e-5 if mode != "f32_split2" else 2e-4)
        _close(yo, yr, tol=3e-5 if mode != "f32_split2" else 2e-4)


@pytest.mark.parametrize("ic,oc,H,W,mode,compute", [
    (64, None, 12, 12, "deformable", "f32"),                 # what BasicDecBlk builds
    (128, None, 10, 14, "reference_cpu", "f32_split3"),       # wider, out_channels None = in_channels (aspp.rs:242)
    (48, 80, 12, 9, "deformable", "f32"),                    # in_channels off the channel granule, out_channels != in_channels
    (96, 32, 16, 16, "deformable", "f32_split2"),
    (20, 24, 8, 8, "reference_cpu", "f32"),
])
def test_aspp_deformable_any_width_vs_oracle(gpu, ic, oc, H, W, mode, compute):
    """ASPPDeformable::new(in_channels, out_channels, vb) for widths other than the 64 -> 64 of BasicDecBlk (aspp.rs:236-246 takes any
    in_channels / out_channels): brn_aspp_deformable_forward against the oracle's restatement, at the north-star gate."""
    from candle_birefnet_amd import ops
    from candle_birefnet_amd.weights import _aspp, synth_weights
    from oracle import oracle as O
    w = synth_weights(_aspp("", ic, oc), seed=11)
    x = rnd(2, ic, H, W, seed=5)
    ops.set_compute(compute)
    try:
        y = ops.aspp_deformable(x, w, mode, out_channels=oc)
    finally:
        ops.set_compute("f32")
    ref = O.aspp_deformable(x, w, 1 if mode == "deformable" else 0, out_channels=oc).astype(np.float64)
    assert y.shape == ref.shape == (2, oc or ic, H, W)
    err = np.abs(np.asarray(y, np.float64) - ref)
    assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"max abs err {err.max():.3e}"
    assert err.max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("mode", ["reference_cpu", "deformable"])
def test_aspp_deformable_any_width_bf16_mode(gpu, mode):
    """the same in compute mode bf16 (maps padded to 64-channel chunks inside): bounded against the oracle"""
    from candle_birefnet_amd import ops
    from candle_birefnet_amd.weights import _aspp, synth_weights
    from oracle import oracle as O
    for ic, oc in ((128, None), (48, 80)):
        w = synth_weights(_aspp("", ic, oc), seed=12)
        x = rnd(1, ic, 16, 16, seed=6)
        ops.set_compute("bf16")
        try:
            y = ops.aspp_deformable(x, w, mode, out_channels=oc)
        finally:
            ops.set_compute("f32")
        ref = O.aspp_deformable(x, w, 1 if mode == "deformable" else 0, out_channels=oc).astype(np.float64)
        err = np.abs(np.asarray(y, np.float64) - ref).max()
        print(f"ASPP bf16 {ic}->{oc or ic} {mode}: max abs err {err:.2e}, |ref| max {np.abs(ref).max():.2f}")
        assert err <= 3e-2 * max(1.0, np.abs(ref).max())


def test_decblk_inter_channels_adaptive_vs_oracle(gpu):
    """DecoderConfig { inter_channels_adaptive: true }: inter_channels = in_channels / 4 (decoder.rs:94-98) — 96 and 40 here, the second
    off the channel granule (the maps between the convs are padded) — with and without the ASPP"""
    from candle_birefnet_amd import ops
    from candle_birefnet_amd.weights import _decblk, synth_weights
    from oracle import oracle as O
    for cin, cout, use_aspp, mode in ((384, 96, True, "deformable"), (160, 64, True, "reference_cpu"), (160, 72, False, "reference_cpu")):
        w = synth_weights(_decblk("", cin, cout, inter=cin // 4, use_aspp=use_aspp), seed=13)
        x = rnd(1, cin, 12, 16, seed=7)
        y = ops.decblk(x, w, cout, mode=mode, use_aspp=use_aspp, inter_channels_adaptive=True)
        ref = O.decblk(x, w, cout, mode=1 if mode == "deformable" else 0, use_aspp=use_aspp, inter_channels_adaptive=True).astype(np.float64)
        err = np.abs(np.asarray(y, np.float64) - ref)
        assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"{cin}->{cout}: max abs err {err.max():.3e}"


@pytest.mark.parametrize("C,O,k,stride,compute", [(20, 24, 3, 1, "f32"), (100, 64, 3, 2, "f32_split3"), (48, 40, 1, 1, "bf16"), (72, 32, 7, 1, "bf16")])
def test_deformable_conv2d_any_in_channels(gpu, C, O, k, stride, compute):
    """DeformableConv2d::new(in_channels, ..) takes any in_channels in the reference (deform_conv.rs:29-36); the deformable mode used to
    need a multiple of 32: the map is now padded with zero channels inside.  Against the oracle (bf16: bounded)."""
    import candle_birefnet_amd as cb
    from candle_birefnet_amd import ops
    from oracle import oracle as ORC
    pad = k // 2
    t = {"offset_conv.weight": rnd(2 * k * k, C, k, k, seed=1, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": rnd(2 * k * k, seed=2, std=0.3),
         "modulator_conv.weight": rnd(k * k, C, k, k, seed=3, std=(C * k * k) ** -0.5), "modulator_conv.bias": rnd(k * k, seed=4, std=0.1),
         "regular_conv.weight": rnd(O, C, k, k, seed=5, std=(C * k * k) ** -0.5), "regular_conv.bias": rnd(O, seed=6, std=0.1)}
    layer = cb.DeformableConv2d.new(C, O, k, stride, pad, cb.VarBuilder.from_tensors(t), mode="deformable")
    x = rnd(2, C, 13, 11, seed=9)
    ops.set_compute(compute)
    try:
        y = layer.forward(x)
    finally:
        ops.set_compute("f32")
    ref = ORC.deform_conv2d(x, t["offset_conv.weight"], t["offset_conv.bias"], t["modulator_conv.weight"], t["modulator_conv.bias"],
                          t["regular_conv.weight"], t["regular_conv.bias"], k, stride, pad, 1).astype(np.float64)
    err = np.abs(np.asarray(y, np.float64) - ref)
    if compute == "bf16":
        assert err.max() <= 3e-2 * max(1.0, np.abs(ref).max())
    else:
        assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"max abs err {err.max():.3e}"
