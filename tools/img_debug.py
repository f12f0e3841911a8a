import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import image_oracle as O
from candle_birefnet_amd.imageproc import preprocess_image
h, w, c, S = 200, 120, 4, 96
img = np.random.default_rng(h * 1000 + w).integers(0, 256, (h, w, c), dtype=np.uint8)
ref = O.preprocess(img, S)
x = preprocess_image(img, S, to_device=False)
idx = np.argwhere(x != ref)
print(len(idx), "mismatches")
tmp = O._sample(img.astype(np.float32), O.axis_table(h, S, "triangle"), 0)
t = O._sample(tmp, O.axis_table(w, S, "triangle"), 1)
for (_, ch, y, xx) in idx[:6]:
    print("c", ch, "y", y, "x", xx, "oracle t =", repr(float(t[y, xx, ch])), "gpu", x[0, ch, y, xx], "ref", ref[0, ch, y, xx])
