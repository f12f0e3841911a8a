//! `ASPPDeformable` of the reference (src/aspp.rs:227-333) over `brn_aspp_deformable_forward`: the module `BasicDecBlk` builds
//! (decoder.rs:107-111) — five branches on a 64-channel map (aspp1 and aspp_deforms.{0,1,2} = DeformConvASPP k 1,1,3,7 -> 256, BN,
//! ReLU; global average pool -> 1x1 -> BN -> ReLU -> broadcast), concat 1280, conv1 1x1 + bn1 + ReLU.
use candle_core::{Module, Result, Tensor, D};
use candle_nn::VarBuilder;

use crate::decoder::{BnW, ConvW};
use crate::deform_conv::DeformableConv2d;
use crate::hip_ffi as ffi;

/// Deformable Conv wrapper matching the pretrained weight structure: offset_conv, modulator_conv, regular_conv (aspp.rs:13-187).
/// Stride 1, `regular_conv` without bias (aspp.rs:45).  `mode` as for `DeformableConv2d`: `BRN_DEFORM_REFERENCE_CPU` = the reference's
/// CPU branch (regular conv, aspp.rs:183-185), `BRN_DEFORM_DEFORMABLE` = forward_metal (aspp.rs:58-165).
pub struct DeformConvASPP {
    inner: DeformableConv2d,
}

impl DeformConvASPP {
    /// aspp.rs:24-56 — same signature
    pub fn new(in_channels: usize, out_channels: usize, kernel_size: usize, padding: usize, vb: VarBuilder) -> Result<Self> {
        let mut inner = DeformableConv2d::new_no_bias(in_channels, out_channels, kernel_size, 1, padding, vb)?;
        inner.mode = ffi::BRN_DEFORM_REFERENCE_CPU;
        Ok(Self { inner })
    }
    pub fn set_mode(&mut self, mode: i32) {
        self.inner.mode = mode;
    }
}

impl Module for DeformConvASPP {
    /// aspp.rs:168-187
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.inner.forward(x)
    }
}

/// ASPP module with deformable conv and BatchNorm (aspp.rs:190-223): atrous_conv -> bn -> relu
pub struct ASPPModuleDeformable {
    pub atrous_conv: DeformConvASPP,
    bn: BnW,
}

impl ASPPModuleDeformable {
    /// aspp.rs:196-214 — same signature
    pub fn new(in_channels: usize, planes: usize, kernel_size: usize, padding: usize, vb: VarBuilder) -> Result<Self> {
        let atrous_conv = DeformConvASPP::new(in_channels, planes, kernel_size, padding, vb.pp("atrous_conv"))?;
        Ok(Self { atrous_conv, bn: BnW::load(planes, vb.pp("bn"))? })
    }
}

impl Module for ASPPModuleDeformable {
    /// aspp.rs:217-223
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.bn.apply(&self.atrous_conv.forward(x)?, true)
    }
}

/// Regular ASPP module (non-deformable; aspp.rs:337-374, dead code in the reference but a pub type): dilated conv + ReLU
pub struct ASPPModule {
    atrous_conv: ConvW,
}

impl ASPPModule {
    /// aspp.rs:343-366 — same signature
    pub fn new(in_channels: usize, planes: usize, kernel_size: usize, padding: usize, dilation: usize, vb: VarBuilder) -> Result<Self> {
        Ok(Self { atrous_conv: ConvW::load_cfg(in_channels, planes, kernel_size, 1, padding, dilation, true, vb.pp("atrous_conv"))? })
    }
}

impl Module for ASPPModule {
    /// aspp.rs:369-373
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.atrous_conv.forward(x, None, ffi::BRN_ACT_RELU)
    }
}

/// Regular ASPP with dilation rates 1, 6, 12, 18 (aspp.rs:377-447, dead code in the reference but a pub type)
pub struct ASPP {
    branches: Vec<ASPPModule>,
    global_avg_pool_conv: ConvW,
    conv1: ConvW,
}

impl ASPP {
    /// aspp.rs:388-426 — same signature
    pub fn new(in_channels: usize, out_channels: Option<usize>, vb: VarBuilder) -> Result<Self> {
        let out_channels = out_channels.unwrap_or(in_channels);
        let inter = 256usize;
        let mut branches = Vec::new();
        for (i, (k, d)) in [(1usize, 1usize), (3, 6), (3, 12), (3, 18)].iter().enumerate() {
            let pad = if *k == 1 { 0 } else { *d };
            branches.push(ASPPModule::new(in_channels, inter, *k, pad, *d, vb.pp(format!("aspp{}", i + 1)))?);
        }
        let global_avg_pool_conv = ConvW::load(in_channels, inter, 1, 0, vb.pp("global_avg_pool").pp("1"))?;
        let conv1 = ConvW::load(inter * 5, out_channels, 1, 0, vb.pp("conv1"))?;
        Ok(Self { branches, global_avg_pool_conv, conv1 })
    }
}

impl Module for ASPP {
    /// aspp.rs:429-447
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (_, _, h, w) = x.dims4()?;
        let mut parts = Vec::new();
        for m in &self.branches {
            parts.push(m.forward(x)?);
        }
        let pooled = x.mean_keepdim(D::Minus2)?.mean_keepdim(D::Minus1)?;
        let x5 = self.global_avg_pool_conv.forward(&pooled, None, ffi::BRN_ACT_RELU)?.upsample_nearest2d(h, w)?;
        parts.push(x5);
        let refs: Vec<&Tensor> = parts.iter().collect();
        self.conv1.forward(&Tensor::cat(&refs, 1)?, None, ffi::BRN_ACT_RELU)
    }
}

/// every tensor `ASPPDeformable::new(ic, Some(oc), vb)` asks its VarBuilder for (aspp.rs:39-45, 247-290), names relative to `vb`
pub fn aspp_weight_spec_for(ic: usize, oc: usize) -> Vec<(String, Vec<usize>)> {
    let mut s: Vec<(String, Vec<usize>)> = Vec::new();
    let pl = 256usize;
    let mut bn = |s: &mut Vec<(String, Vec<usize>)>, p: &str, c: usize| {
        for leaf in ["weight", "bias", "running_mean", "running_var"] {
            s.push((format!("{p}.{leaf}"), vec![c]));
        }
    };
    for (module, k) in [("aspp1", 1usize), ("aspp_deforms.0", 1), ("aspp_deforms.1", 3), ("aspp_deforms.2", 7)] {
        let cp = format!("{module}.atrous_conv.");
        s.push((format!("{cp}offset_conv.weight"), vec![2 * k * k, ic, k, k]));
        s.push((format!("{cp}offset_conv.bias"), vec![2 * k * k]));
        s.push((format!("{cp}modulator_conv.weight"), vec![k * k, ic, k, k]));
        s.push((format!("{cp}modulator_conv.bias"), vec![k * k]));
        s.push((format!("{cp}regular_conv.weight"), vec![pl, ic, k, k]));      // no bias (aspp.rs:45)
        bn(&mut s, &format!("{module}.bn"), pl);
    }
    s.push(("global_avg_pool.1.weight".to_string(), vec![pl, ic, 1, 1]));
    bn(&mut s, "global_avg_pool.2", pl);
    s.push(("conv1.weight".to_string(), vec![oc, 5 * pl, 1, 1]));
    bn(&mut s, "bn1", oc);
    s
}
/// the module `BasicDecBlk` builds: `ASPPDeformable::new(64, None, vb)` (decoder.rs:107-111)
pub fn aspp_weight_spec() -> Vec<(String, Vec<usize>)> {
    aspp_weight_spec_for(64, 64)
}

/// ASPP with deformable convolutions (aspp.rs:227).  `mode`: `BRN_DEFORM_REFERENCE_CPU` = what the reference's CPU path computes
/// (aspp.rs:183-185), `BRN_DEFORM_DEFORMABLE` = the Metal path (aspp.rs:58-165).
pub struct ASPPDeformable {
    named: ffi::NamedTensors,
    in_channels: usize,
    out_channels: usize,
    pub mode: i32,
}

impl ASPPDeformable {
    /// aspp.rs:237 — same signature, any widths (`out_channels` None = `in_channels`, aspp.rs:242)
    pub fn new(in_channels: usize, out_channels: Option<usize>, vb: VarBuilder) -> Result<Self> {
        let out_channels = out_channels.unwrap_or(in_channels);
        let named = ffi::NamedTensors::from_varbuilder(&vb, &aspp_weight_spec_for(in_channels, out_channels))?;
        Ok(Self { named, in_channels, out_channels, mode: ffi::BRN_DEFORM_REFERENCE_CPU })
    }
}

impl Module for ASPPDeformable {
    /// aspp.rs:303 — x [B,in_channels,H,W] -> [B,out_channels,H,W]
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != self.in_channels {
            candle_core::bail!("expected {} input channels, got {c}", self.in_channels)
        }
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * self.out_channels * h * w];
        let prefix = std::ffi::CString::new("").unwrap();
        ffi::check(unsafe {
            ffi::brn_aspp_deformable_forward(self.named.views.as_ptr(), self.named.views.len(), prefix.as_ptr(), self.in_channels as i32,
                                             self.out_channels as i32, self.mode, xin.as_ptr(), b as i32, h as i32, w as i32, out.as_mut_ptr(),
                                             ffi::BRN_MEM_HOST, 0, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, self.out_channels, h, w), x.device())
    }
}
