"""The CPU oracle (oracle/brn_oracle.cpp) against the committed golden fixtures (fp64 torch restatement) and against
torch functional ops on random shapes.  PARITY UNPINNED by the reference itself (no vectors there): see oracle header."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_cases as G
from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(a, b, tol=3e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, float(np.abs(b).max()))
    assert a.shape == b.shape
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e}"


def test_weights_checksum_guard():
    k = np.load(os.path.join(GOLD, "models_small.npz"))
    np.testing.assert_allclose(G.weights_checksum(), k["weights_checksum"], rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", sorted(G.KAT_CASES))
def test_oracle_kats(name):
    k = np.load(os.path.join(GOLD, "kats.npz"))
    _close(G.KAT_CASES[name](G.OracleBackend(O)), k[name])


@pytest.mark.parametrize("tag", ["m64_d2222_ref", "m64_d2222_def", "m96_d2222_ref_b2", "m128_full_ref"])
def test_oracle_model_goldens(tag):
    k = np.load(os.path.join(GOLD, "models_small.npz"))
    cfg, w, x = G.model_case(tag)
    y = O.forward_logits(O.cfg_from(cfg), w, x)
    err = np.abs(y.astype(np.float64) - k[tag])
    assert ((err <= 1e-3) | (err <= 1e-2 * np.abs(k[tag]))).all()
    assert err.max() < 5e-5, err.max()       # fp32 restatement vs fp64 restatement: reorder noise only


def test_oracle_pieces_goldens():
    k = np.load(os.path.join(GOLD, "models_small.npz"))
    cfg, w, x = G.model_case("m64_d2222_ref")
    y, parts = O.forward_parts(O.cfg_from(cfg), w, x)
    for i in range(4):
        _close(parts["f"][i], k[f"m64_d2222_ref_f{i}"], tol=1e-4)
    _close(parts["x4s"], k["m64_d2222_ref_x4s"], tol=1e-4)
    # pieces driven one by one reproduce the whole (bench_inference.rs:34-90)
    sq = O.squeeze(O.cfg_from(cfg), w, parts["x4"])
    np.testing.assert_array_equal(sq, parts["x4s"])
    out = O.decoder(O.cfg_from(cfg), w, x, parts["x1"], parts["x2"], parts["x3"], parts["x4s"])
    np.testing.assert_array_equal(out, y)


def rnd(*shape, seed=0, std=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * std).astype(np.float32)


@pytest.mark.parametrize("M,K,N", [(37, 192, 576), (6, 64, 17), (1, 32, 1), (200, 400, 33)])
def test_oracle_linear(M, K, N):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = F.gelu(torch.from_numpy(x).double() @ torch.from_numpy(w).double().T + torch.from_numpy(b).double()) + torch.from_numpy(r).double()
    _close(O.linear(x, w, b, act="gelu_erf", residual=r), ref.numpy())


@pytest.mark.parametrize("B,C,H,W,O_,k,s,p,d", [(2, 5, 9, 11, 7, 3, 1, 1, 1), (1, 3, 16, 16, 8, 4, 4, 0, 1), (1, 8, 12, 12, 4, 3, 1, 6, 6), (1, 4, 9, 9, 6, 7, 2, 3, 1)])
def test_oracle_conv2d(B, C, H, W, O_, k, s, p, d):
    x, w, b = rnd(B, C, H, W, seed=1), rnd(O_, C, k, k, seed=2, std=0.2), rnd(O_, seed=3)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=s, padding=p, dilation=d)
    _close(O.conv2d(x, w, b, stride=s, padding=p, dilation=d), ref.numpy())


def test_oracle_layer_norm_and_empty_edge():
    x, g, b = rnd(11, 192, seed=1, std=3.0) + 1.0, 1 + rnd(192, seed=2, std=0.1), rnd(192, seed=3, std=0.1)
    ref = F.layer_norm(torch.from_numpy(x).double(), (192,), torch.from_numpy(g).double(), torch.from_numpy(b).double(), 1e-5)
    _close(O.layer_norm(x, g, b), ref.numpy())


def test_oracle_errors_are_reported():
    cfg, w, x = G.model_case("m64_d2222_ref")
    w = dict(w)
    del w["decoder.gdt_convs_pred_4.0.weight"]      # loaded-but-unused head must exist (birefnet.rs:230-232)... in the GPU lib;
    y = O.forward_logits(O.cfg_from(cfg), w, x)      # the oracle only touches what forward reads, so this still runs
    assert np.isfinite(y).all()
    del w["bb.norm0.weight"]
    with pytest.raises(RuntimeError, match="cannot find tensor bb.norm0.weight"):
        O.forward_logits(O.cfg_from(cfg), w, x)
