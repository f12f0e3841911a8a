"""random geometries through the op-level window attention (window 12) in every compute mode against the fp64 torch restatement
(tests/torch_ref.py; mode bf16 against the exact-operand reference of tests/test_ops_gpu.py).  tools/att_fuzz.py [n] [seed]"""
import sys, os, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from candle_birefnet_amd import ops
import torch_ref as R
import test_ops_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"f32": 2e-5, "f32_split3": 2e-5, "f32_half2": 2e-5, "f32_split2": 1e-4, "f16": 3.2e-3}
bad = 0
for it in range(n):
    B = int(rng.integers(1, 4)); H = int(rng.integers(1, 50)); W = int(rng.integers(1, 50)); heads = int(rng.choice([1, 2, 3, 6, 12, 24])); shift = int(rng.choice([0, 6]))
    C = heads * 32
    desc = f"B{B} {H}x{W} heads{heads} shift{shift}"
    w = T._attn_weights(C, heads, seed=10 + it)
    x = T.rnd(B, H, W, C, seed=99 + it)
    ref = R.window_attention_block(torch.from_numpy(x).double(), w, "", heads, 12, shift, torch.float64).numpy()
    for mode in ("f32", "f32_split3", "f32_half2", "f32_split2", "bf16", "f16"):
        ops.set_compute(mode)
        try:
            y = np.asarray(ops.window_attention(x, heads, shift, w["attn.qkv.weight"], w["attn.qkv.bias"], w["attn.proj.weight"], w["attn.proj.bias"],
                                                w["attn.relative_position_bias_table"]), np.float64)
            tol = 2.5e-2 if mode == "bf16" else TOL[mode]
            err = float(np.abs(y - ref).max() / max(1.0, np.abs(ref).max()))
            if not np.isfinite(y).all() or err > tol:
                bad += 1; print("BAD ", mode, desc, f"err {err:.2e}", flush=True)
        except Exception as e:
            bad += 1; print("FAIL", mode, desc, str(e)[:220], flush=True)
    ops.set_compute("f32")
print(f"{n} cases, {bad} problems")
