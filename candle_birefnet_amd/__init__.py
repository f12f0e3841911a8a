"""candle_birefnet_amd — MI355X (gfx950) native BiRefNet inference path behind the candle-birefnet API.

Public surface mirrors the reference crate root (lib.rs:6-14): BiRefNet, DeformableConv2d, SwinTransformer, SwinConfig,
plus birefnet::{BiRefNetConfig} and the VarBuilder stand-in.  Importing this package loads libbirefnet_hip.so and fails
loudly if it is missing: there is no CPU path in the product."""
from ._ffi import BrnError, LIB_PATH, device_count  # noqa: F401  (loads the shared library)
from .config import BiRefNetConfig, DecoderConfig, SwinConfig  # noqa: F401
from .weights import VarBuilder, birefnet_weight_spec, swin_weight_spec, synth_input, synth_weights  # noqa: F401
from .birefnet import BiRefNet, BiRefNetDecoder, DeformableConv2d, SqueezeModule, SwinTransformer  # noqa: F401
from . import ops  # noqa: F401
from . import imageproc  # noqa: F401
