"""BASELINE.json configs[2..4] under -m gpu: B=8 1024^2 in the bf16 modes, 2048^2 against a committed golden
(fp32 torch restatement, tests/golden/make_golden.py --full2048) and through the batch-independence property, and the N>1
product path (two processes, two handles, one device) against one process."""
import os
import sys

import numpy as np
import pytest

import golden_cases as G

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
# bf16 modes are throughput modes: the fp32 gate (1e-3 abs | 1e-2 rel) does not apply; their error against the fp64 / fp32
# restatement is REPORTED and bounded here.  |logit| max of the 1024^2 synthetic run is ~2.2.
BF16_ABS_BOUND = {"bf16": 2.4e-2,   # 3x the largest error measured on MI355X: 6.4e-3 (1024^2), 7.9e-3 (2048^2, deformable)
                  "f16": 2.7e-3}    # compute mode f16 (fp16 storage), 3x the largest measured: 5.1e-4 (c3), 8.8e-4 (c3 deformable), 8.0e-4 (c5 deformable)
# the same in mask space (forward() = sigmoid(logits), birefnet.rs:466-469; north_star: "masks within 1e-3 of reference" — an fp32
# criterion): sigmoid' <= 1/4, so a logit error e is a mask error <= e / 4
BF16_MASK_BOUND = 0.25 * BF16_ABS_BOUND["bf16"]
S16_MASK_BOUND = {"bf16": BF16_MASK_BOUND, "f16": 1e-3}       # f16 (fp16 storage): under the north star's "masks within 1e-3"


def _mask_err(m, x, gold_logits, stride):
    """max |forward(x)[0] - sigmoid(golden logits)| on the strided golden of image 0"""
    p = m.forward(x[:1]).cpu().numpy().astype(np.float64)[0, :, ::stride, ::stride]
    return float(np.abs(p - 1.0 / (1.0 + np.exp(-gold_logits.astype(np.float64)))).max())


def _full_model(mode, max_batch=0, size=0):
    import candle_birefnet_amd as cb
    cfg = cb.BiRefNetConfig()
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    return cb, cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode, max_batch=max_batch, max_size=(size, size))


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_c3_batch8_1024_bf16(gpu, mode):
    """configs[2] (and one rank of configs[3]): B=8, 1024x1024.  Finite; the same call twice gives the same bits; an image alone
    equals the same image inside the batch up to the mode's rounding; error vs the strided fp64 golden of image 0 is bounded."""
    import torch
    cb, m = _full_model(mode, 8, 1024)
    x = torch.from_numpy(cb.synth_input(8, 1024, 1024)).cuda()
    y = m.forward_logits(x)
    y2 = m.forward_logits(x)
    assert torch.isfinite(y).all()
    assert torch.equal(y, y2)                                        # no float atomics anywhere on the path
    k = np.load(os.path.join(GOLD, "model_1024.npz"))
    yn = y.cpu().numpy().astype(np.float64)
    e0 = float(np.abs(yn[0, :, ::16, ::16] - k["m1024_full_ref_s16"][0]).max())
    alone = [m.forward_logits(x[b:b + 1]).cpu().numpy().astype(np.float64)[0] for b in (0, 5)]
    d = max(float(np.abs(alone[0] - yn[0]).max()), float(np.abs(alone[1] - yn[5]).max()))
    em = _mask_err(m, x, k["m1024_full_ref_s16"][0], 16)
    print(f"c3 [{mode}] B=8 1024^2: max abs err of image 0 vs the fp64 golden {e0:.3e} (mask space {em:.3e}); image alone vs in batch {d:.3e}")
    assert e0 < BF16_ABS_BOUND[mode] and d < BF16_ABS_BOUND[mode] and em < S16_MASK_BOUND[mode]
    # the images of the batch are different images (seeds 1000..1007): a stuck batch index would show here
    assert float(np.abs(yn[1] - yn[0]).max()) > 0.1
    m.close()


def _deform_model(mode, max_batch, size):
    import candle_birefnet_amd as cb
    cfg = cb.BiRefNetConfig(deform_mode="deformable")
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    return cb, cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode, max_batch=max_batch, max_size=(size, size))


def test_full_1024_deformable_fp32_equivalent_against_golden(gpu):
    """deform_mode=deformable (aspp.rs:58-165, the mode SURVEY D1 assigns to configs 3-5) at the full 1024^2 Swin-L geometry in the
    parity-graded arithmetic, against the fp64 restatement (tests/golden/model_1024_def.npz: every 16th pixel + global statistics)."""
    import torch
    cb, m = _deform_model("f32_split3", 1, 1024)
    k = np.load(os.path.join(GOLD, "model_1024_def.npz"))
    x = torch.from_numpy(cb.synth_input(1, 1024, 1024)).cuda()
    y = m.forward_logits(x).cpu().numpy().astype(np.float64)
    ref = k["m1024_full_def_s16"].astype(np.float64)
    err = np.abs(y[:, :, ::16, ::16] - ref)
    assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"max abs err {err.max():.3e}"
    st = k["m1024_full_def_stats"]
    assert abs(y.sum() - st[0]) <= 1e-4 * st[1] and abs(np.abs(y).sum() - st[1]) <= 1e-4 * st[1]
    assert abs(y.min() - st[2]) <= 1e-3 and abs(y.max() - st[3]) <= 1e-3
    print(f"1024x1024 Swin-L deformable [f32_split3]: max abs err on the strided fp64 golden {err.max():.2e}")
    assert err.max() < 2e-4
    m.close()


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_c3_batch8_1024_bf16_deformable(gpu, mode):
    """configs[2] / one rank of configs[3] in deform_mode=deformable (the bf16-MFMA gather kernel): B=8, 1024x1024, finite, repeatable,
    image 0 bounded against the fp64 golden, an image alone equals the image inside the batch up to the mode's rounding."""
    import torch
    cb, m = _deform_model(mode, 8, 1024)
    x = torch.from_numpy(cb.synth_input(8, 1024, 1024)).cuda()
    y = m.forward_logits(x)
    assert torch.isfinite(y).all() and torch.equal(y, m.forward_logits(x))
    k = np.load(os.path.join(GOLD, "model_1024_def.npz"))
    yn = y.cpu().numpy().astype(np.float64)
    e0 = float(np.abs(yn[0, :, ::16, ::16] - k["m1024_full_def_s16"][0]).max())
    d = float(np.abs(m.forward_logits(x[5:6]).cpu().numpy().astype(np.float64)[0] - yn[5]).max())
    em = _mask_err(m, x, k["m1024_full_def_s16"][0], 16)
    print(f"c3 deformable [{mode}] B=8 1024^2: max abs err of image 0 vs the fp64 golden {e0:.3e} (mask space {em:.3e}); image alone vs in batch {d:.3e}")
    assert e0 < BF16_ABS_BOUND[mode] and d < BF16_ABS_BOUND[mode] and em < S16_MASK_BOUND[mode]
    m.close()


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_c5_batch4_2048_bf16_deformable(gpu, mode):
    """configs[4] in deform_mode=deformable: B=4, 2048x2048 (dec1's ASPP gathers on 512^2 maps), against the fp32 restatement."""
    import torch
    cb, m = _deform_model(mode, 4, 2048)
    k = np.load(os.path.join(GOLD, "model_2048_def.npz"))
    x = torch.from_numpy(cb.synth_input(4, 2048, 2048)).cuda()
    y = m.forward_logits(x)
    assert torch.isfinite(y).all() and torch.equal(y, m.forward_logits(x))
    e0 = float(np.abs(y.cpu().numpy().astype(np.float64)[0, :, ::32, ::32] - k["m2048_full_def_s32"][0]).max())
    em = _mask_err(m, x, k["m2048_full_def_s32"][0], 32)
    print(f"c5 deformable [{mode}] B=4 2048^2: max abs err of image 0 vs the fp32 golden {e0:.3e} (mask space {em:.3e})")
    assert e0 < BF16_ABS_BOUND[mode] and em < S16_MASK_BOUND[mode]
    m.close()


def test_c5_2048_fp32_equivalent_against_golden(gpu):
    """configs[4] geometry, B=1, in the parity-graded arithmetic: every 32nd pixel + global statistics of the fp32 restatement."""
    import torch
    cb, m = _full_model("f32_split3", 1, 2048)
    k = np.load(os.path.join(GOLD, "model_2048.npz"))
    x = torch.from_numpy(cb.synth_input(1, 2048, 2048)).cuda()
    y = m.forward_logits(x).cpu().numpy().astype(np.float64)
    ref = k["m2048_full_ref_s32"].astype(np.float64)
    err = np.abs(y[:, :, ::32, ::32] - ref)
    assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(ref))).all(), f"max abs err {err.max():.3e}"
    st = k["m2048_full_ref_stats"]
    assert abs(y.sum() - st[0]) <= 2e-4 * st[1] and abs(np.abs(y).sum() - st[1]) <= 2e-4 * st[1]
    assert abs(y.min() - st[2]) <= 1e-3 and abs(y.max() - st[3]) <= 1e-3
    print(f"2048x2048 Swin-L [f32_split3]: max abs err on the strided fp32 golden {err.max():.2e}")
    m.close()


@pytest.mark.parametrize("mode", ["bf16"])
def test_c5_batch4_2048_bf16(gpu, mode):
    """configs[4]: B=4, 2048x2048 in the bf16 modes: finite, repeatable, image 0 bounded against the golden, batch independence."""
    import torch
    cb, m = _full_model(mode, 4, 2048)
    k = np.load(os.path.join(GOLD, "model_2048.npz"))
    x = torch.from_numpy(cb.synth_input(4, 2048, 2048)).cuda()
    y = m.forward_logits(x)
    assert torch.isfinite(y).all() and torch.equal(y, m.forward_logits(x))
    yn = y.cpu().numpy().astype(np.float64)
    e0 = float(np.abs(yn[0, :, ::32, ::32] - k["m2048_full_ref_s32"][0]).max())
    d = float(np.abs(m.forward_logits(x[3:4]).cpu().numpy().astype(np.float64)[0] - yn[3]).max())
    em = _mask_err(m, x, k["m2048_full_ref_s32"][0], 32)
    print(f"c5 [{mode}] B=4 2048^2: max abs err of image 0 vs the fp32 golden {e0:.3e} (mask space {em:.3e}); image alone vs in batch {d:.3e}")
    assert e0 < BF16_ABS_BOUND[mode] and d < BF16_ABS_BOUND[mode] and em < BF16_MASK_BOUND
    m.close()


@pytest.mark.parametrize("compute", ["f32_split3", "bf16"])
def test_two_processes_two_handles_bit_equal(gpu, tmp_path, compute):
    """The N>1 path of the PRODUCT: bench.py's launcher starts 2 ranks (own process, own handle, device LOCAL_RANK % ndev = 0
    here), each runs its shard of a 3-image global batch through libbirefnet_hip; the concatenation equals ONE process running
    the three images one by one, bit for bit (images are independent units; no float atomics; same plan for the same shape)."""
    import bench
    import candle_birefnet_amd as cb
    tag, gb = "m96_d2222_ref_b2", 3
    rc = bench.launch_ranks(2, [sys.executable, os.path.join(HERE, "_shard_worker.py"), str(tmp_path), tag, str(gb), compute], timeout=900)
    assert rc == 0
    ys = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(2)], 0)
    depths, S, _, mode = G.MODEL_CASES[tag]
    cfg = cb.BiRefNetConfig(deform_mode=mode)
    cfg.swin.depths = list(depths)
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=compute)
    # rank 0 ran images [0,2) as one batch of 2, rank 1 image 2 alone: reproduce exactly those calls in this process
    x = cb.synth_input(gb, S, S)
    y1 = np.concatenate([m.forward_logits(x[0:2]), m.forward_logits(x[2:3])], 0)
    np.testing.assert_array_equal(ys, y1)
    m.close()


def test_bench_two_ranks_reports_c4_blocks(gpu):
    """configs[3] (B = 64 over 8 GPUs = 8 images per rank) through bench.py's own N > 1 path, rehearsed with 2 ranks that share this
    box's one card (gloo for the control plane: RCCL refuses two ranks on one device; the rates of such a run mean nothing): the
    self-launcher, the rendezvous, shard_range, per-rank times, and the c4 / c4_deformable blocks of the default line."""
    import json
    import subprocess
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off"],
                        env=dict(os.environ, BRN_BENCH_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert d["outputs_finite"] and d["max_abs_err_image0_vs_strided_golden"] < 2e-4
    assert sorted(d["other_configs"]) == ["c4", "c4_deformable"]
    for k, blk in d["other_configs"].items():
        assert blk["n_gpus"] == 2 and blk["batch_per_gpu"] == 8 and blk["compute"] == "bf16" and len(blk["per_rank_ms_per_step"]) == 2
        assert blk["outputs_finite"] and blk["max_abs_err_image0_vs_strided_golden"] < BF16_ABS_BOUND["bf16"]
        assert blk["deform_mode"] == ("deformable" if k.endswith("deformable") else "reference_cpu")
        assert blk["roofline"]["families"]["gemm_deform_nhwc"]["launches"] == (20 if k.endswith("deformable") else 0)


def test_bf16_ragged_token_count_against_oracle(gpu):
    """1056 x 1056 (33 x 33 patches of 32): the stage-0 token count, 264^2 + 132^2 = 87120, is not a multiple of the 64-row tiles of the
    weight-stationary kernels (gemm_wstat_bf16_kernel for qkv / fc1, gemm_wstat_ln_bf16_kernel = projection + norm2 in one launch),
    the maps are padded to the window size at every stage, and the window grid has a ragged border.  Mode bf16 against the fp32 CPU
    oracle on the same input (test infrastructure, ~10 s): bounded like the other bf16 runs; the call repeated is bit-identical."""
    import torch
    from oracle import oracle as O
    cb, m = _full_model("bf16", 1, 1056)
    x = cb.synth_input(1, 1056, 1056)
    xd = torch.from_numpy(x).cuda()
    y = m.forward_logits(xd)
    assert torch.equal(y, m.forward_logits(xd))
    cfg = cb.BiRefNetConfig()
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    ref = np.asarray(O.forward_logits(O.cfg_from(cfg), w, x), np.float64)
    err = float(np.abs(y.cpu().numpy().astype(np.float64) - ref).max())
    print(f"1056x1056 Swin-L [bf16]: max abs err vs the fp32 oracle {err:.3e} (|logit| max {np.abs(ref).max():.2f})")
    assert np.isfinite(err) and err < BF16_ABS_BOUND["bf16"]
    m.close()


@pytest.mark.parametrize("deform", ["reference_cpu", "deformable"])
def test_odd_sizes_and_batches_bf16_tracks_fp32_equivalent(gpu, deform):
    """Sizes and batches the tile planners never saw while being tuned (non-square, odd patch-grid sides, batches 1 ... 5; token
    counts that are not multiples of any tile): every launch must be accepted — a 1056 x 1056 input once made the bf16 planner pick
    a 192-wide tile grid that reached past the padding of a 512-row weight matrix — and mode bf16 must stay within its bound of the
    parity-graded f32_split3 on the same input.  (tools/size_sweep.py is the longer list.)"""
    import torch
    import candle_birefnet_amd as cb
    cfg = cb.BiRefNetConfig(deform_mode=deform)
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    ms = {mode: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode) for mode in ("f32_split3", "bf16")}
    for H, W, B in [(608, 608, 1), (736, 864, 2), (1184, 1056, 1), (672, 672, 5), (416, 1760, 3)]:
        x = torch.from_numpy(cb.synth_input(B, H, W)).cuda()
        ys = {k: m.forward_logits(x).float().cpu().numpy() for k, m in ms.items()}
        assert all(np.isfinite(v).all() for v in ys.values()), (H, W, B)
        d = float(np.abs(ys["bf16"] - ys["f32_split3"]).max())
        assert d < BF16_ABS_BOUND["bf16"], (H, W, B, d)
    for m in ms.values():
        m.close()


def test_sub_batch_workspace_is_planned_for_the_part_and_falls_back(gpu, tmp_path):
    """a device-resident batch runs as two sub-batches with a workspace each; every workspace is sized by a dry run of the PART
    (not of the whole batch), and when the second workspace cannot be allocated (BRN_FAULT_SIDE_ARENA: the test hook that makes that
    hipMalloc fail) the batch runs as one part on one stream instead of failing.  Child processes: the switches are read per process."""
    import subprocess
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "import candle_birefnet_amd as cb\n"
        "cfg = cb.BiRefNetConfig(); cfg.swin.depths = [2, 2, 2, 2]\n"
        "w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)\n"
        "m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute='bf16', max_batch=6, max_size=(256, 256))\n"
        "x = torch.from_numpy(cb.synth_input(6, 256, 256)).cuda()\n"
        "free0 = torch.cuda.mem_get_info()[0]\n"
        "y = m.forward_logits(x); torch.cuda.synchronize()\n"
        "assert torch.equal(y, m.forward_logits(x))\n"
        "y5 = m.forward_logits(x[:5])            # parts of 3 + 2 images: the larger part was planned\n"
        "np.savez(sys.argv[1], y=y.cpu().numpy(), y5=y5.cpu().numpy(), grew=np.int64(free0 - torch.cuda.mem_get_info()[0]))\n") % ROOT
    outs = {}
    for tag, env in (("split", {}), ("fallback", {"BRN_FAULT_SIDE_ARENA": "1"}), ("one", {"BRN_SPLIT_STREAMS": "1"})):
        out = str(tmp_path / f"{tag}.npz")
        pr = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert pr.returncode == 0, pr.stderr[-2000:]
        outs[tag] = np.load(out)
    # the fallback IS the one-stream path: same plans, same bits
    np.testing.assert_array_equal(outs["fallback"]["y"], outs["one"]["y"])
    np.testing.assert_array_equal(outs["fallback"]["y5"], outs["one"]["y5"])
    # two parts compute the same images with the plans of a smaller M: equal up to the mode's rounding
    assert np.abs(outs["split"]["y"] - outs["one"]["y"]).max() < BF16_ABS_BOUND["bf16"]
    assert np.abs(outs["split"]["y5"] - outs["one"]["y5"]).max() < BF16_ABS_BOUND["bf16"]
    # the second workspace is allocated at the first split forward and is no larger than a part needs: the device memory taken by the
    # first forward (side workspace) is at most what the handle reserved at creation for the main one (+ allocator slack)
    print("device memory taken by the first forward: split %d MB, one stream %d MB" % (outs["split"]["grew"] >> 20, outs["one"]["grew"] >> 20))


def test_mixed_mode_bf16_backbone_split2_decoder(gpu):
    """BRN_BF16_DEC_SPLIT2 (compute "bf16_dec_split2"): the backbone as mode bf16, fusion / squeeze / decoder as mode f32_split2 on fp32
    maps — the decoder's chained bf16 roundings are most of mode bf16's error.  Small full-depth goldens in both
    deform modes (finite, repeatable, between the two pure modes), the piece-wise entries (backbone = the bf16 one bit for bit, decoder =
    the f32_split2 one bit for bit), and c3 (B = 8, 1024^2, reference_cpu): the mask-space error of image 0 under the north star's 1e-3."""
    import torch
    import candle_birefnet_amd as cb
    k = np.load(os.path.join(GOLD, "models_small.npz"))
    for tag in sorted(G.MODEL_CASES):
        cfg, w, x = G.model_case(tag)
        vb = cb.VarBuilder.from_tensors(w)
        ms = {c: cb.BiRefNet.new(cfg, vb, compute=c) for c in ("bf16_dec_split2", "bf16", "f32_split2")}
        y = {c: np.asarray(m.forward_logits(x)) for c, m in ms.items()}
        np.testing.assert_array_equal(np.asarray(ms["bf16_dec_split2"].forward_logits(x)), y["bf16_dec_split2"])
        e = {c: float(np.abs(v.astype(np.float64) - k[tag]).max()) for c, v in y.items()}
        print(f"{tag}: max abs err vs the fp64 golden: bf16 {e['bf16']:.2e}, mixed {e['bf16_dec_split2']:.2e}, f32_split2 {e['f32_split2']:.2e}")
        assert np.isfinite(y["bf16_dec_split2"]).all() and e["bf16_dec_split2"] < BF16_ABS_BOUND["bf16"]
        feats_mixed = ms["bf16_dec_split2"].backbone.forward(x)
        for a, b in zip(feats_mixed, ms["bf16"].backbone.forward(x)):
            np.testing.assert_array_equal(a, b)
        f32f = ms["f32_split2"].backbone.forward(x)          # any features will do for the decoder-side comparison
        S = x.shape[-1]
        fz = [np.concatenate([f, f], 1) for f in f32f[:3]] + [np.random.default_rng(0).standard_normal((x.shape[0], 3072, S // 32, S // 32)).astype(np.float32)]
        np.testing.assert_array_equal(ms["bf16_dec_split2"].decoder.forward(x, *fz), ms["f32_split2"].decoder.forward(x, *fz))
        for m in ms.values():
            m.close()
    cb2, m = _full_model("bf16_dec_split2", 8, 1024)
    kk = np.load(os.path.join(GOLD, "model_1024.npz"))
    xb = torch.from_numpy(cb2.synth_input(8, 1024, 1024)).cuda()
    yb = m.forward_logits(xb)
    assert torch.isfinite(yb).all() and torch.equal(yb, m.forward_logits(xb))
    e0 = float(np.abs(yb.cpu().numpy().astype(np.float64)[0, :, ::16, ::16] - kk["m1024_full_ref_s16"][0]).max())
    em = _mask_err(m, xb, kk["m1024_full_ref_s16"][0], 16)
    print(f"c3 [bf16_dec_split2] B=8 1024^2: max abs err of image 0 vs the fp64 golden {e0:.3e} (mask space {em:.3e})")
    assert em < 1e-3 and e0 < 4e-3
    m.close()
