"""The C-ABI library loads on a CPU-only host and exports exactly the symbols include/birefnet_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "birefnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(brn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    import candle_birefnet_amd as cb
    lib = ctypes.CDLL(cb.LIB_PATH)
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in birefnet_hip.h but not exported"
    assert sorted(cb._ffi.DECLARED) == names


def test_abi_version_and_defaults():
    import candle_birefnet_amd as cb
    assert cb._ffi.lib.brn_abi_version() == 1
    assert b"gfx950" in cb._ffi.lib.brn_build_info()
    c = cb._ffi.brn_config()
    cb._ffi.lib.brn_config_default_swin_l(c)
    # BiRefNetConfig::default (birefnet.rs:32-46) and SwinConfig::swin_l (swin.rs:69-80)
    assert (c.size_w, c.size_h) == (1024, 1024) and c.backbone == b"swin_v1_l"
    assert list(c.backbone_channels) == [192, 384, 768, 1536] and list(c.cxt) == [192, 384, 768]
    assert (c.mul_scl_ipt, c.ms_supervision, c.dec_ipt, c.use_aspp_deformable) == (1, 1, 1, 1)
    assert (c.embed_dim, list(c.depths), list(c.num_heads), c.window_size, c.patch_size, c.in_channels) == (192, [2, 2, 18, 2], [6, 12, 24, 48], 12, 4, 3)
    lat = (ctypes.c_int * 4)()
    cb._ffi.lib.brn_config_lateral_channels(c, lat)
    assert list(lat) == [384, 768, 1536, 3072]            # birefnet.rs:50-53
    assert cb._ffi.lib.brn_config_x4_channels(c) == 5760   # birefnet.rs:56-61


def test_no_cpu_fallback_errors_are_loud():
    """Without a HIP device every compute entry returns BRN_ERR_NO_DEVICE with a message; nothing silently runs on the CPU."""
    import numpy as np
    import candle_birefnet_amd as cb
    if cb.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cb.BrnError) as e:
        cb.ops.linear(np.zeros((4, 32), np.float32), np.zeros((8, 32), np.float32))
    assert e.value.status == 4 and "no CPU fallback" in str(e.value)
    cfg = cb.BiRefNetConfig()
    with pytest.raises(cb.BrnError):
        cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors({}))


def test_product_never_imports_oracle():
    """the product package must not reference oracle/ (a product path through the oracle would void parity)"""
    pkg = os.path.join(ROOT, "candle_birefnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                s = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in s and "from oracle" not in s and "import oracle" not in s and "brn_oracle" not in s, f
