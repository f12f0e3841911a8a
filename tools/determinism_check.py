"""Run the same forward several times per compute mode and report bitwise differences (race hunting)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
cfg = cb.BiRefNetConfig()
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
x = torch.from_numpy(cb.synth_input(1, 1024, 1024)).cuda()
for mode in sys.argv[1:] or ["f32_split2"]:
    m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), max_batch=1, max_size=(1024, 1024), compute=mode)
    outs = []
    for r in range(6):
        feats = [f.cpu().numpy() for f in m.backbone.forward(x)]
        y = m.forward_logits(x).cpu().numpy()
        outs.append(feats + [y])
    for r in range(1, 6):
        d = [float(np.abs(a - b).max()) for a, b in zip(outs[0], outs[r])]
        n = [int((a != b).sum()) for a, b in zip(outs[0], outs[r])]
        print(mode, "run", r, "max |diff| x1..x4,logits:", " ".join(f"{v:.2e}" for v in d), "| #diff", n, flush=True)
    m.close()
