"""Characterise the rare wrong rows of the split conv: exact small-integer data, so every wrong element is visible and its
error says what the kernel multiplied instead (zeros, a neighbour's rows, another K tile)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
from candle_birefnet_amd import ops
mode = sys.argv[1] if len(sys.argv) > 1 else "f32_split2"
variant = sys.argv[2] if len(sys.argv) > 2 else "ones"
ops.set_compute(mode)
B, Cin, H, W, Cout, k, pad = 1, 64, 256, 256, 256, 7, 3
c = np.arange(Cin)[:, None, None]; h = np.arange(H)[None, :, None]; w = np.arange(W)[None, None, :]
if variant == "ones":
    x = np.ones((B, Cin, H, W), np.float32)
elif variant == "rows":      # value identifies the pixel row/col coarsely, exact in bf16
    x = np.broadcast_to(((h // 2) % 64 + 1).astype(np.float32), (Cin, H, W))[None].copy()
else:                        # value identifies the channel (= position inside the K tile)
    x = np.broadcast_to((c % 32 + 1).astype(np.float32), (Cin, H, W))[None].copy()
ww = np.ones((Cout, Cin, k, k), np.float32)
bb = np.zeros(Cout, np.float32)
ref = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(ww).double(), padding=pad).numpy()
xx = torch.from_numpy(x).cuda()
for r in range(10):
    o = ops.conv2d(xx, ww, bb, stride=1, padding=pad).cpu().numpy().astype(np.float64)
    d = o - ref
    idx = np.argwhere(d[0] != 0)
    if len(idx) == 0:
        print("run", r, "clean", flush=True); continue
    m = idx[:, 1] * W + idx[:, 2]
    vals, cnt = np.unique(d[0][d[0] != 0], return_counts=True)
    rows = sorted(set(m.tolist()))
    print("run", r, "#wrong", len(idx), "rows(m)", rows[:8], "err values:", dict(zip(vals.tolist()[:10], cnt.tolist()[:10])), flush=True)
    for mm in rows[:4]:
        hh, wc = mm // W, mm % W
        print("    m", mm, "(h,w)=", (hh, wc), "ref", ref[0, 0, hh, wc], "got", sorted(set(o[0, :, hh, wc].tolist()))[:6])
