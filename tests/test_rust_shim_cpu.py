"""the Rust shim (rust_shim/, uncompiled here: no rustc in the image) stays consistent with the C header, and its
weight-name contract equals the Python mirror's (which the GPU tests exercise through the library)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_matches_header():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_shim.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_shim_weight_names_cover_the_python_spec():
    """every leaf pattern of the Python weight spec (candle_birefnet_amd.weights, which the library is tested with) appears in the
    Rust source that builds the same list: module prefixes and leaf names, including the loaded-but-unused heads"""
    import candle_birefnet_amd as cb
    names = [n for n, _, _ in cb.birefnet_weight_spec(cb.BiRefNetConfig())]
    src = open(os.path.join(ROOT, "rust_shim", "src", "birefnet.rs")).read() + open(os.path.join(ROOT, "rust_shim", "src", "swin.rs")).read()
    def pat(n):                       # "decoder.decoder_block3.dec_att.aspp_deforms.1.bn.running_var" -> its distinctive components
        return [c for c in re.split(r"[.\d]+", n) if c]
    vocab = set()
    for n in names:
        vocab.update(pat(n))
    missing = [v for v in sorted(vocab) if v not in src]
    assert not missing, missing
    assert len(names) == len(set(names)) and len(names) > 600
