"""BiRefNetConfig / SwinConfig / DecoderConfig — field-for-field mirrors of the reference's config structs
(birefnet.rs:13-67, swin.rs:14-88, decoder.rs:12-24), decorative fields included."""
from dataclasses import dataclass, field
from typing import List, Tuple

from . import _ffi


@dataclass
class SwinConfig:
    """swin.rs:14-23"""
    embed_dim: int = 192
    depths: List[int] = field(default_factory=lambda: [2, 2, 18, 2])
    num_heads: List[int] = field(default_factory=lambda: [6, 12, 24, 48])
    window_size: int = 12
    mlp_ratio: float = 4.0
    patch_size: int = 4
    in_channels: int = 3
    drop_path_rate: float = 0.2

    @staticmethod
    def swin_t():  # swin.rs:27-38
        return SwinConfig(96, [2, 2, 6, 2], [3, 6, 12, 24], 7, 4.0, 4, 3, 0.2)

    @staticmethod
    def swin_s():  # swin.rs:41-52
        return SwinConfig(96, [2, 2, 18, 2], [3, 6, 12, 24], 7, 4.0, 4, 3, 0.2)

    @staticmethod
    def swin_b():  # swin.rs:55-66
        return SwinConfig(128, [2, 2, 18, 2], [4, 8, 16, 32], 12, 4.0, 4, 3, 0.2)

    @staticmethod
    def swin_l():  # swin.rs:69-80
        return SwinConfig(192, [2, 2, 18, 2], [6, 12, 24, 48], 12, 4.0, 4, 3, 0.2)

    def stage_channels(self):  # swin.rs:83-87
        return [self.embed_dim * (1 << i) for i in range(len(self.depths))]


@dataclass
class DecoderConfig:
    """decoder.rs:12-24"""
    use_aspp_deformable: bool = True
    inter_channels_adaptive: bool = False


@dataclass
class BiRefNetConfig:
    """birefnet.rs:13-46.  `deform_mode` is this library's switch for decision D1 (not a reference field):
    "reference_cpu" reproduces the CPU path the parity target runs (aspp.rs:183-185), "deformable" the Metal path."""
    size: Tuple[int, int] = (1024, 1024)
    backbone: str = "swin_v1_l"
    backbone_channels: List[int] = field(default_factory=lambda: [192, 384, 768, 1536])
    mul_scl_ipt: bool = True
    ms_supervision: bool = True
    dec_ipt: bool = True
    use_aspp_deformable: bool = True
    cxt: List[int] = field(default_factory=lambda: [192, 384, 768])
    deform_mode: str = "reference_cpu"
    # BiRefNet::new always builds SwinConfig::swin_l() (birefnet.rs:390-391); tests may shrink `depths` through this
    swin: SwinConfig = field(default_factory=SwinConfig.swin_l)

    @staticmethod
    def swin_l():  # birefnet.rs:64-66
        return BiRefNetConfig()

    def lateral_channels(self):  # birefnet.rs:50-53
        mult = 2 if self.mul_scl_ipt else 1
        return [c * mult for c in self.backbone_channels]

    def x4_channels(self):  # birefnet.rs:56-61
        mult = 2 if self.mul_scl_ipt else 1
        return self.backbone_channels[3] * mult + sum(c * mult for c in self.cxt)

    def to_c(self):
        c = _ffi.brn_config()
        _ffi.lib.brn_config_default_swin_l(c)
        c.size_w, c.size_h = int(self.size[0]), int(self.size[1])
        c.backbone = self.backbone.encode()[:31]
        for i in range(4):
            c.backbone_channels[i] = int(self.backbone_channels[i])
            c.depths[i] = int(self.swin.depths[i])
            c.num_heads[i] = int(self.swin.num_heads[i])
        c.mul_scl_ipt = int(self.mul_scl_ipt)
        c.ms_supervision = int(self.ms_supervision)
        c.dec_ipt = int(self.dec_ipt)
        c.use_aspp_deformable = int(self.use_aspp_deformable)
        c.n_cxt = min(3, len(self.cxt))
        for i in range(c.n_cxt):
            c.cxt[i] = int(self.cxt[i])
        c.embed_dim = int(self.swin.embed_dim)
        c.window_size = int(self.swin.window_size)
        c.mlp_ratio = float(self.swin.mlp_ratio)
        c.patch_size = int(self.swin.patch_size)
        c.in_channels = int(self.swin.in_channels)
        c.drop_path_rate = float(self.swin.drop_path_rate)
        modes = {"reference_cpu": _ffi.BRN_DEFORM_REFERENCE_CPU, "deformable": _ffi.BRN_DEFORM_DEFORMABLE}
        if self.deform_mode not in modes:
            raise ValueError(f"deform_mode must be one of {sorted(modes)}")
        c.deform_mode = modes[self.deform_mode]
        return c


def swin_to_c(cfg: SwinConfig):
    b = BiRefNetConfig(swin=cfg)
    return b.to_c()
