import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
from candle_birefnet_amd import ops
cfg = cb.BiRefNetConfig()
mode = sys.argv[1] if len(sys.argv) > 1 else "f32_split2"
rng = np.random.default_rng(1)
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), max_batch=1, max_size=(1024, 1024), compute=mode)
x = torch.from_numpy(cb.synth_input(1, 1024, 1024)).cuda()
lat = cfg.lateral_channels()
feats = [torch.from_numpy(rng.standard_normal((1, lat[i], 256 >> i, 256 >> i)).astype(np.float32)).cuda() for i in range(4)]
x4 = torch.from_numpy(rng.standard_normal((1, cfg.x4_channels(), 32, 32)).astype(np.float32)).cuda()
def rep(name, fn, n=5):
    o0 = fn().cpu().numpy()
    for r in range(n):
        o = fn().cpu().numpy()
        print(f"{mode} {name} run {r}: max|diff| {np.abs(o - o0).max():.3e} #diff {(o != o0).sum()} of {o.size}", flush=True)
m.close()
ops.set_compute(mode)
for (M, N, K) in [(1024, 512, 64), (4096, 1536, 1536), (1024, 64, 1024), (16384, 768, 768), (65536, 512, 64), (65536, 64, 1024), (65536, 384, 384)]:
    xx = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    ww = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bb = rng.standard_normal(N).astype(np.float32)
    rep(f"linear {M}x{N}x{K}", lambda: ops.linear(xx, ww, bb), 3)
for (B, Cin, H, W, Cout, k, pad) in [(1, 64, 32, 32, 256, 7, 3), (1, 64, 256, 256, 256, 7, 3), (1, 1536, 32, 32, 16, 3, 1), (1, 384, 128, 128, 16, 3, 1), (1, 5760, 32, 32, 64, 3, 1), (1, 3072, 32, 32, 64, 3, 1), (1, 64, 32, 32, 3072, 3, 1), (1, 480, 256, 256, 64, 3, 1), (1, 960, 128, 128, 64, 3, 1)]:
    xx = torch.from_numpy(rng.standard_normal((B, Cin, H, W)).astype(np.float32)).cuda()
    ww = (rng.standard_normal((Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bb = rng.standard_normal(Cout).astype(np.float32)
    rep(f"conv {Cin}->{Cout} k{k} {H}x{W}", lambda: ops.conv2d(xx, ww, bb, stride=1, padding=pad), 3)
ops.set_compute("f32")
