"""safetensors -> VarBuilder (infer_image.rs:35-40).  Round trip of a checkpoint-shaped file on the CPU."""
import numpy as np


def test_varbuilder_from_safetensors_roundtrip(tmp_path):
    from safetensors.numpy import save_file
    import candle_birefnet_amd as cb
    cfg = cb.BiRefNetConfig()
    cfg.swin.depths = [1, 1, 1, 1]
    spec = cb.birefnet_weight_spec(cfg)
    small = [(n, s, k) for n, s, k in spec if int(np.prod(s)) <= 20000][:40]
    w = cb.synth_weights(small, seed=7)
    p = str(tmp_path / "model.safetensors")
    save_file(w, p)
    vb = cb.VarBuilder.from_safetensors(p)
    for n, s, _ in small:
        head, _, leaf = n.rpartition(".")
        b = vb
        for part in head.split("."):
            b = b.pp(part)
        np.testing.assert_array_equal(b.get(s, leaf), w[n])


def _create_from_file(path, cfg=None):
    import ctypes as C
    import candle_birefnet_amd as cb
    from candle_birefnet_amd import _ffi
    cfg = cfg or cb.BiRefNetConfig()
    h = C.c_void_p()
    c = cfg.to_c()
    st = _ffi.lib.brn_model_create_from_safetensors(C.byref(c), str(path).encode(), b"", 0, _ffi.BRN_F32, 0, 0, 0, C.byref(h))
    return st, (_ffi.lib.brn_last_error() or b"").decode()


def test_native_loader_rejects_missing_and_malformed_files(tmp_path):
    """brn_model_create_from_safetensors: file errors surface as BRN_ERR_INVALID_ARG with a message, before any device use."""
    from candle_birefnet_amd import _ffi
    st, msg = _create_from_file(tmp_path / "nope.safetensors")
    assert st == _ffi.BRN_ERR_INVALID_ARG and "cannot open" in msg
    p = tmp_path / "short.safetensors"
    p.write_bytes(b"\x10\x00\x00")
    st, msg = _create_from_file(p)
    assert st == _ffi.BRN_ERR_INVALID_ARG and "not a safetensors file" in msg
    p = tmp_path / "liar.safetensors"
    p.write_bytes((1 << 40).to_bytes(8, "little") + b"{}")
    st, msg = _create_from_file(p)
    assert st == _ffi.BRN_ERR_INVALID_ARG and "header length" in msg
    p = tmp_path / "offsets.safetensors"
    hdr = b'{"a":{"dtype":"F32","shape":[4],"data_offsets":[0,64]}}'
    p.write_bytes(len(hdr).to_bytes(8, "little") + hdr + b"\0" * 16)
    st, msg = _create_from_file(p)
    assert st == _ffi.BRN_ERR_INVALID_ARG and "outside the file" in msg


def test_native_loader_parses_then_needs_a_device_or_a_tensor(tmp_path):
    """A well-formed file gets past the parser: without the checkpoint's tensors the error names the missing one (candle: Err
    from vb.get); on a box without a GPU the call stops at BRN_ERR_NO_DEVICE — there is no CPU fallback behind this entry."""
    import torch
    from safetensors.numpy import save_file
    from candle_birefnet_amd import _ffi
    p = str(tmp_path / "tiny.safetensors")
    save_file({"bb.patch_embed.proj.bias": np.zeros(192, np.float32), "unrelated": np.ones((2, 3), np.float16)}, p)
    st, msg = _create_from_file(p)
    if torch.cuda.is_available():
        assert st == _ffi.BRN_ERR_MISSING_TENSOR and "bb." in msg
    else:
        assert st == _ffi.BRN_ERR_NO_DEVICE


def _raw_file(path, hdr: bytes, data: bytes):
    path.write_bytes(len(hdr).to_bytes(8, "little") + hdr + data)
    return path


def test_native_loader_rejects_crafted_headers(tmp_path):
    """The header is untrusted input: wrapped shapes, duplicate names, overlapping tensors, unbalanced or trailing JSON are refused
    before any size derived from them reaches a memcpy (the safetensors crate rejects the same files)."""
    from candle_birefnet_amd import _ffi
    # 2^62 x 4 elements of 4 bytes wraps to 0 modulo 2^64 == (b1 - b0): must not pass as an empty tensor
    st, msg = _create_from_file(_raw_file(tmp_path / "wrap.safetensors", b'{"a":{"dtype":"F32","shape":[4611686018427387904,4],"data_offsets":[0,0]}}', b""))
    assert st in (_ffi.BRN_ERR_SHAPE, _ffi.BRN_ERR_INVALID_ARG) and ("impossible shape" in msg or "out of range" in msg), msg
    st, msg = _create_from_file(_raw_file(tmp_path / "wrap2.safetensors", b'{"a":{"dtype":"F32","shape":[1073741824,1073741824,4],"data_offsets":[0,0]}}', b""))
    assert st == _ffi.BRN_ERR_SHAPE and "impossible shape" in msg, msg
    st, msg = _create_from_file(_raw_file(tmp_path / "dup.safetensors",
                                          b'{"a":{"dtype":"F32","shape":[1],"data_offsets":[0,4]},"a":{"dtype":"F32","shape":[1],"data_offsets":[4,8]}}', b"\0" * 8))
    assert st == _ffi.BRN_ERR_INVALID_ARG and "appears twice" in msg, msg
    st, msg = _create_from_file(_raw_file(tmp_path / "overlap.safetensors",
                                          b'{"a":{"dtype":"F32","shape":[2],"data_offsets":[0,8]},"b":{"dtype":"F32","shape":[2],"data_offsets":[4,12]}}', b"\0" * 12))
    assert st == _ffi.BRN_ERR_INVALID_ARG and "overlap" in msg, msg
    st, msg = _create_from_file(_raw_file(tmp_path / "open.safetensors", b'{"a":{"dtype":"F32","shape":[1],"data_offsets":[0,4]}', b"\0" * 4))
    assert st == _ffi.BRN_ERR_INVALID_ARG, msg
    st, msg = _create_from_file(_raw_file(tmp_path / "trail.safetensors", b'{"a":{"dtype":"F32","shape":[1],"data_offsets":[0,4]}} x', b"\0" * 4))
    assert st == _ffi.BRN_ERR_INVALID_ARG and "after the header" in msg, msg
    st, msg = _create_from_file(_raw_file(tmp_path / "meta.safetensors", b'{"__metadata__":{"k":["}",{"q":"]"}]},"a":{"dtype":"F32","shape":[1],"data_offsets":[0,4]}}   ', b"\0" * 4))
    assert st in (_ffi.BRN_ERR_MISSING_TENSOR, _ffi.BRN_ERR_NO_DEVICE), msg      # parsed; stops later for want of tensors / a device
