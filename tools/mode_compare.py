#!/usr/bin/env python3
"""speed + error of every compute mode at BASELINE configs[1] against the committed fp64 strided golden"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
k = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_1024.npz"))
ref = k["m1024_full_ref_s16"].astype(np.float64)
cfg = cb.BiRefNetConfig()
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
x = torch.from_numpy(cb.synth_input(1, 1024, 1024)).cuda()
for mode in sys.argv[1:] or ["f32", "f32_split3", "f32_split2", "bf16"]:
    m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), max_batch=1, max_size=(1024, 1024), compute=mode)
    for _ in range(3): y = m.forward_logits(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): y = m.forward_logits(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    e = np.abs(y.cpu().numpy()[:, :, ::16, ::16].astype(np.float64) - ref)
    rel = e / np.maximum(np.abs(ref), 1e-12)
    gate = ((e <= 1e-3) | (rel <= 1e-2)).all()
    print(f"{mode:14s} {dt*1e3:7.2f} ms  {1/dt:6.1f} img/s  max abs err {e.max():.3e}  mean abs {e.mean():.3e}  gate(1e-3|1e-2) {gate}", flush=True)
    m.close()
