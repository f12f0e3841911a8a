"""random image sizes / channel counts / model sizes through brn_preprocess_image and brn_postprocess_mask against the numpy restatement
of the image crate's resampler (oracle/image_oracle.py): pre-processing bit-exact, masks within one grey level.  tests/fuzz/imageproc_fuzz.py [n] [seed]   (lives under tests/: it checks against oracle/, which only tests may use)"""
import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from oracle import image_oracle as O
from candle_birefnet_amd.imageproc import preprocess_image, postprocess_mask
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n):
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700)); c = int(rng.choice([3, 4])); S = int(rng.choice([32, 64, 96, 160, 256]))
    img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    try:
        if not np.array_equal(preprocess_image(img, S, to_device=False), O.preprocess(img, S)):
            bad += 1; print("BAD  preprocess", h, w, c, S, flush=True)
        oh, ow = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        lg = (rng.standard_normal((1, 1, S, S)) * 4).astype(np.float32)
        d = np.abs(postprocess_mask(lg, (oh, ow)).astype(np.int32) - O.postprocess(lg[0, 0], oh, ow).astype(np.int32))
        if d.max() > 1 or (d != 0).mean() > 5e-3:
            bad += 1; print("BAD  postprocess", S, oh, ow, int(d.max()), float((d != 0).mean()), flush=True)
    except Exception as e:
        bad += 1; print("FAIL", h, w, c, S, str(e)[:200], flush=True)
print(f"{n} cases, {bad} problems")
