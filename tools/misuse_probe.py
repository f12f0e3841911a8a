import sys, numpy as np, torch
sys.path.insert(0, ".")
import candle_birefnet_amd as cb
cfg = cb.BiRefNetConfig(); cfg.swin.depths = [2, 2, 2, 2]
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute="bf16")
def t(name, f):
    try:
        r = f(); print(name, "-> OK", getattr(r, "shape", r))
    except Exception as e:
        print(name, "->", type(e).__name__, str(e)[:140])
t("100x100", lambda: m.forward_logits(np.zeros((1, 3, 100, 100), np.float32)))
t("4 channels", lambda: m.forward_logits(np.zeros((1, 4, 64, 64), np.float32)))
t("B=0", lambda: m.forward_logits(np.zeros((0, 3, 64, 64), np.float32)))
t("0x0", lambda: m.forward_logits(np.zeros((1, 3, 0, 0), np.float32)))
t("after errors 64x64", lambda: m.forward_logits(np.zeros((1, 3, 64, 64), np.float32)))
t("non-contiguous cuda", lambda: m.forward_logits(torch.zeros(1, 64, 64, 3).cuda().permute(0, 3, 1, 2)))
t("fp16 cuda", lambda: m.forward_logits(torch.zeros(1, 3, 64, 64, dtype=torch.float16).cuda()))
t("fp64 numpy", lambda: m.forward_logits(np.zeros((1, 3, 64, 64), np.float64)))
m.close()
t("after close", lambda: m.forward_logits(np.zeros((1, 3, 64, 64), np.float32)))
t("bad compute", lambda: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute="fp8"))
w2 = dict(w); k0 = next(iter(w2)); w2.pop(k0)
t("missing tensor", lambda: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w2)))
w3 = dict(w); w3[k0] = np.zeros((3, 3), np.float32)
t("wrong shape", lambda: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w3)))
