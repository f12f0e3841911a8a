import sys, time, numpy as np
sys.path.insert(0, ".")
import torch, candle_birefnet_amd as cb
cfg = cb.BiRefNetConfig()
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute="f32_split3", max_batch=1, max_size=(1024, 1024))
x = cb.synth_input(1, 1024, 1024)
for _ in range(3): y = m.forward_logits(x)
t = time.time()
for _ in range(20): y = m.forward_logits(x)
host_ms = (time.time() - t) / 20 * 1e3
xd = torch.from_numpy(np.asarray(x)).cuda()
for _ in range(3): yd = m.forward_logits(xd)
torch.cuda.synchronize(); t = time.time()
for _ in range(20): yd = m.forward_logits(xd)
torch.cuda.synchronize(); dev_ms = (time.time() - t) / 20 * 1e3
print(f"host buffers {host_ms:.2f} ms/image ({1e3/host_ms:.1f} img/s), device buffers {dev_ms:.2f} ms/image ({1e3/dev_ms:.1f} img/s)")
