"""Unprofiled per-piece times (backbone full / squeeze / decoder) the way bench_inference.rs:37-92 splits them."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
cfg = cb.BiRefNetConfig()
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), max_batch=1, max_size=(1024, 1024), compute=sys.argv[1] if len(sys.argv) > 1 else "f32_split2")
x = torch.from_numpy(cb.synth_input(1, 1024, 1024)).cuda()
lat = cfg.lateral_channels()
rng = np.random.default_rng(1)
feats = [torch.from_numpy(rng.standard_normal((1, lat[i], 256 >> i, 256 >> i)).astype(np.float32)).cuda() for i in range(4)]
x4 = torch.from_numpy(rng.standard_normal((1, cfg.x4_channels(), 32, 32)).astype(np.float32)).cuda()
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"forward_logits {t(lambda: m.forward_logits(x)):.3f} ms | backbone(full only) {t(lambda: m.backbone.forward(x)):.3f} | squeeze {t(lambda: m.squeeze_module.forward(x4)):.3f} | decoder {t(lambda: m.decoder.forward(x, *feats)):.3f}")
