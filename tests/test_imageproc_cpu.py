"""Image pre/post-processing: properties of the numpy restatement of image 0.25.9's resampler, and the
PNG codec the example uses.  No GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import image_oracle as O  # noqa: E402


def test_weight_tables_are_normalised_and_cover_the_support():
    for (i, o, name) in [(37, 64, "triangle"), (1000, 64, "triangle"), (64, 37, "lanczos3"), (64, 301, "lanczos3"), (8, 8, "lanczos3")]:
        tabs = O.axis_table(i, o, name)
        assert len(tabs) == o
        for left, ws in tabs:
            assert 0 <= left and left + len(ws) <= i and len(ws) >= 1
            assert abs(float(ws.sum()) - 1.0) < 1e-5
        ratio = max(i / o, 1.0)
        assert max(len(ws) for _, ws in tabs) <= int(np.ceil(2 * (1.0 if name == "triangle" else 3.0) * ratio)) + 2


def test_constant_images_stay_constant_and_same_size_triangle_is_identity():
    img = np.full((13, 29, 3), 77, np.uint8)
    assert (O.resize_u8(img, 64, 48, "triangle") == 77).all()
    assert (O.resize_u8(img, 5, 7, "lanczos3") == 77).all()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (16, 20, 4), dtype=np.uint8)
    np.testing.assert_array_equal(O.resize_u8(img, 20, 16, "triangle"), img)       # ratio 1: taps (0, 1, 0)


def test_upscale_by_two_triangle_known_answer():
    # 1-D ramp [0, 100] -> 4 samples: centres at 0.25, 0.75, 1.25, 1.75 input pixels; triangle weights -> 0, 25, 75, 100
    img = np.array([[[0], [100]]], np.uint8)
    out = O.resize_u8(img, 4, 1, "triangle")[0, :, 0]
    np.testing.assert_array_equal(out, [0, 25, 75, 100])


def test_preprocess_shape_and_normalisation():
    img = np.zeros((10, 12, 3), np.uint8)
    img[..., 0] = 255
    x = O.preprocess(img, 32)
    assert x.shape == (1, 3, 32, 32) and x.dtype == np.float32
    np.testing.assert_allclose(x[0, 0], (1.0 - 0.485) / 0.229, rtol=1e-6)
    np.testing.assert_allclose(x[0, 1], (0.0 - 0.456) / 0.224, rtol=1e-6)


def test_postprocess_saturates_and_truncates():
    lg = np.array([[-100.0, 100.0], [0.0, 0.0]], np.float32)
    m = O.postprocess(lg, 2, 2)
    assert m[0, 0] == 0 and m[0, 1] == 255 and m[1, 0] == 127 and m[1, 1] == 127    # 0.5 * 255 = 127.5 -> `as u8` = 127


def test_png_roundtrip_and_reference_asset(tmp_path):
    from candle_birefnet_amd.imageproc import read_png, write_png_gray
    rng = np.random.default_rng(1)
    m = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    p = str(tmp_path / "m.png")
    write_png_gray(p, m)
    back = read_png(p)
    assert back.shape == (33, 47, 3)
    np.testing.assert_array_equal(back[:, :, 0], m)
