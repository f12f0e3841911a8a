"""Image pre/post-processing on the GPU (brn_preprocess_image / brn_postprocess_mask) against the numpy restatement of
image 0.25.9's resampler: same weight tables, same f32 accumulation order, no fma — bit-exact."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import image_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,c,S", [(37, 53, 3, 64), (200, 120, 4, 96), (64, 64, 3, 64), (500, 333, 3, 128), (9, 1000, 3, 32)])
def test_preprocess_matches_oracle_bitwise(gpu, h, w, c, S):
    from candle_birefnet_amd.imageproc import preprocess_image
    img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, c), dtype=np.uint8)
    ref = O.preprocess(img, S)
    x_dev = preprocess_image(img, S).cpu().numpy()
    x_host = preprocess_image(img, S, to_device=False)
    np.testing.assert_array_equal(x_dev, ref)
    np.testing.assert_array_equal(x_host, ref)


@pytest.mark.parametrize("S,oh,ow", [(64, 37, 90), (64, 64, 64), (96, 128, 128), (64, 500, 21), (128, 33, 33)])
def test_postprocess_matches_oracle(gpu, S, oh, ow):
    import torch
    from candle_birefnet_amd.imageproc import postprocess_mask
    lg = (np.random.default_rng(S + oh).standard_normal((1, 1, S, S)) * 4).astype(np.float32)
    ref = O.postprocess(lg[0, 0], oh, ow)
    m_host = postprocess_mask(lg, (oh, ow))
    m_dev = postprocess_mask(torch.from_numpy(lg).cuda(), (oh, ow))
    np.testing.assert_array_equal(m_host, m_dev)
    # the sigmoid's expf is libm on one side and the device library on the other: a pixel whose v*255 sits on an integer may land
    # one level apart before the Lanczos pass; everything else is bit-exact
    d = np.abs(m_host.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, f"max diff {d.max()}, {(d != 0).mean():.2e} of pixels differ"
    # with the sigmoid taken out of the comparison (probabilities in, apply_sigmoid=0) it is exact
    p = (1.0 / (1.0 + np.exp(-lg.astype(np.float64)))).astype(np.float32)
    np.testing.assert_array_equal(postprocess_mask(p, (oh, ow), apply_sigmoid=False), O.postprocess(p[0, 0], oh, ow, apply_sigmoid=False))


def test_image_entry_points_reject_bad_arguments(gpu):
    import candle_birefnet_amd as cb
    from candle_birefnet_amd.imageproc import preprocess_image, postprocess_mask
    with pytest.raises(ValueError):
        preprocess_image(np.zeros((4, 4, 2), np.uint8), 32)
    with pytest.raises(ValueError):
        postprocess_mask(np.zeros((1, 1, 8, 9), np.float32), (4, 4))
    with pytest.raises(cb.BrnError, match="INVALID_ARG"):
        postprocess_mask(np.zeros((1, 1, 8, 8), np.float32), (0, 4))


def test_end_to_end_png_to_mask(gpu, tmp_path):
    """examples/infer_image.rs as a whole on a synthetic picture: PNG -> preprocess -> forward -> postprocess -> PNG"""
    import candle_birefnet_amd as cb
    from candle_birefnet_amd.imageproc import preprocess_image, postprocess_mask, read_png, write_png_gray
    cfg = cb.BiRefNetConfig()
    cfg.swin.depths = [1, 1, 1, 1]
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=3)
    m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w))
    img = np.random.default_rng(5).integers(0, 256, (75, 50, 3), dtype=np.uint8)
    x = preprocess_image(img, 64)
    logits = m.forward_logits(x)
    mask = postprocess_mask(logits, img.shape[:2])
    assert mask.shape == (75, 50) and mask.dtype == np.uint8
    ref = O.postprocess(logits.cpu().numpy()[0, 0], 75, 50)
    assert np.abs(mask.astype(np.int32) - ref.astype(np.int32)).max() <= 1
    p = str(tmp_path / "mask.png")
    write_png_gray(p, mask)
    np.testing.assert_array_equal(read_png(p)[:, :, 0], mask)


def test_imageproc_size_fuzz(gpu):
    """tests/fuzz/imageproc_fuzz.py: 40 random (image size, channel count, model size, output size) combinations, 1 x 1 up to 700 x 700,
    through both entry points against the resampler restatement: pre-processing bit-exact, masks within one grey level."""
    import subprocess
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "imageproc_fuzz.py"), "40", "7"], capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    assert "40 cases, 0 problems" in pr.stdout, pr.stdout[-3000:]


def test_infer_images_batch_equals_image_by_image(gpu):
    """brn_infer_images_u8 (examples/infer_image.rs:44-110 for a batch: uploads, resize + normalise, ONE forward of the batch, u8, resize
    back, downloads) against the three single-image entry points chained by hand with the same batched forward: bit-identical masks;
    mixed image sizes, RGB and RGBA, a second call on the cached staging / tables, a larger batch that regrows the staging."""
    import torch
    import candle_birefnet_amd as cb
    from candle_birefnet_amd.imageproc import infer_images, postprocess_mask, preprocess_image
    cfg = cb.BiRefNetConfig()
    cfg.swin.depths = [2, 2, 2, 2]
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    S = 128
    for compute in ("f32_split3", "bf16"):
        m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=compute)
        rng = np.random.default_rng(5)
        for ch, sizes in ((3, [(150, 97), (64, 64), (33, 201)]), (4, [(90, 90), (128, 128), (77, 140), (200, 30), (55, 66)])):
            imgs = [rng.integers(0, 256, (h, wd, ch), dtype=np.uint8) for h, wd in sizes]
            masks = infer_images(m, imgs, S)
            masks2 = infer_images(m, imgs, S)
            x = torch.cat([preprocess_image(im, S) for im in imgs], 0)
            p = m.forward(x)
            for im, mk, mk2, pi in zip(imgs, masks, masks2, p):
                assert mk.shape == im.shape[:2] and mk.dtype == np.uint8
                np.testing.assert_array_equal(mk, mk2)
                np.testing.assert_array_equal(mk, postprocess_mask(pi[None], im.shape[:2], apply_sigmoid=False))
        with pytest.raises(cb.BrnError):
            infer_images(m, [np.zeros((10, 10, 3), np.uint8)], 100)       # S not a multiple of 32
        m.close()
