//! Decoder blocks of the reference (src/decoder.rs:1-141) over the C ABI of libbirefnet_hip.so: `DecoderConfig`, `SimpleConvs`,
//! `BasicLatBlk` (plain convolutions: `brn_conv2d_forward`) and `BasicDecBlk` (`brn_decblk_forward`).  Same constructors, same
//! `Module::forward`, same VarBuilder names; `ResBlk` (decoder.rs:143-217, dead code in the reference but a pub type) as the
//! BasicDecBlk body + a 1x1 skip convolution.
use candle_core::{Module, Result, Tensor};
use candle_nn::VarBuilder;

use crate::hip_ffi as ffi;

/// Configuration for decoder blocks (decoder.rs:12-24)
#[derive(Clone)]
pub struct DecoderConfig {
    pub use_aspp_deformable: bool,
    pub inter_channels_adaptive: bool,
}

impl Default for DecoderConfig {
    /// decoder.rs:17-24
    fn default() -> Self {
        Self { use_aspp_deformable: true, inter_channels_adaptive: false }
    }
}

/// one Conv2d kept on the host in candle's [O, Cin, k, k] layout (bias optional: `conv2d_no_bias`)
pub(crate) struct ConvW {
    w: Vec<f32>,
    b: Option<Vec<f32>>,
    pub(crate) o: usize,
    pub(crate) cin: usize,
    k: usize,
    stride: usize,
    pad: usize,
    dil: usize,
}

impl ConvW {
    /// candle_nn::conv2d(cin, o, k, Conv2dConfig { padding, stride, dilation, .. }, vb): "weight" [o,cin,k,k] (+ "bias" [o])
    pub(crate) fn load_cfg(cin: usize, o: usize, k: usize, stride: usize, pad: usize, dil: usize, bias: bool, vb: VarBuilder) -> Result<Self> {
        let w = ffi::to_host(&vb.get((o, cin, k, k), "weight")?)?;
        let b = if bias { Some(ffi::to_host(&vb.get(o, "bias")?)?) } else { None };
        Ok(Self { w, b, o, cin, k, stride, pad, dil })
    }
    pub(crate) fn load(cin: usize, o: usize, k: usize, pad: usize, vb: VarBuilder) -> Result<Self> {
        Self::load_cfg(cin, o, k, 1, pad, 1, true, vb)
    }
    pub(crate) fn forward(&self, x: &Tensor, bn: Option<&BnW>, act: i32) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != self.cin {
            candle_core::bail!("conv expects {} input channels, got {c}", self.cin)
        }
        let span = self.dil * (self.k - 1) + 1;
        if h + 2 * self.pad < span || w + 2 * self.pad < span {
            candle_core::bail!("conv {}x{} (dilation {}) does not fit a {h}x{w} map padded by {}", self.k, self.k, self.dil, self.pad)
        }
        let (ho, wo) = ((h + 2 * self.pad - span) / self.stride + 1, (w + 2 * self.pad - span) / self.stride + 1);
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * self.o * ho * wo];
        let null = std::ptr::null::<f32>();
        let (g, be, m, v, eps) = match bn {
            Some(n) => (n.g.as_ptr(), n.b.as_ptr(), n.m.as_ptr(), n.v.as_ptr(), 1e-5f32),
            None => (null, null, null, null, 0f32),
        };
        let bias = self.b.as_ref().map_or(null, |v| v.as_ptr());
        ffi::check(unsafe {
            ffi::brn_conv2d_forward(xin.as_ptr(), b as i32, c as i32, h as i32, w as i32, self.w.as_ptr(), bias, self.o as i32,
                                    self.k as i32, self.k as i32, self.stride as i32, self.pad as i32, self.dil as i32, g, be, m, v, eps, act,
                                    out.as_mut_ptr(), ffi::BRN_MEM_HOST, 0, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, self.o, ho, wo), x.device())
    }
}

/// candle_nn::batch_norm(c, 1e-5, vb) in eval mode: weight / bias / running_mean / running_var
pub(crate) struct BnW {
    pub(crate) g: Vec<f32>,
    pub(crate) b: Vec<f32>,
    pub(crate) m: Vec<f32>,
    pub(crate) v: Vec<f32>,
}

impl BnW {
    pub(crate) fn load(c: usize, vb: VarBuilder) -> Result<Self> {
        Ok(Self {
            g: ffi::to_host(&vb.get(c, "weight")?)?,
            b: ffi::to_host(&vb.get(c, "bias")?)?,
            m: ffi::to_host(&vb.get(c, "running_mean")?)?,
            v: ffi::to_host(&vb.get(c, "running_var")?)?,
        })
    }
    /// `BatchNorm::forward_t(x, false)` (+ ReLU) on an NCHW tensor, on the host: the few places of the public surface where a batch
    /// norm follows an op that has no fused form at the C ABI (ASPPModuleDeformable::forward, aspp.rs:217-223)
    pub(crate) fn apply(&self, x: &Tensor, relu: bool) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        let mut v = ffi::to_host(x)?;
        let hw = h * w;
        for bi in 0..b {
            for ci in 0..c {
                let scale = self.g[ci] / (self.v[ci] + 1e-5f32).sqrt();
                let shift = self.b[ci] - self.m[ci] * scale;
                for e in v[(bi * c + ci) * hw..(bi * c + ci + 1) * hw].iter_mut() {
                    let t = *e * scale + shift;
                    *e = if relu && t < 0.0 { 0.0 } else { t };
                }
            }
        }
        Tensor::from_vec(v, (b, c, h, w), x.device())
    }
}

/// conv 3x3 (pad 1) + BatchNorm(eval) + ReLU in one library call: what `birefnet::GdtConvs` is (birefnet.rs:97-118)
pub(crate) struct ConvBnRelu {
    conv: ConvW,
    bn: BnW,
}

impl ConvBnRelu {
    pub(crate) fn load(cin: usize, o: usize, conv_vb: VarBuilder, bn_vb: VarBuilder) -> Result<Self> {
        Ok(Self { conv: ConvW::load(cin, o, 3, 1, conv_vb)?, bn: BnW::load(o, bn_vb)? })
    }
    pub(crate) fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.conv.forward(x, Some(&self.bn), ffi::BRN_ACT_RELU)
    }
}

/// Simple convolution block (used for ipt_blk): conv1 -> conv_out, NO activation between (decoder.rs:28-56)
pub struct SimpleConvs {
    conv1: ConvW,
    conv_out: ConvW,
}

impl SimpleConvs {
    /// decoder.rs:34-47 — same signature
    pub fn new(in_channels: usize, out_channels: usize, inter_channels: usize, vb: VarBuilder) -> Result<Self> {
        let conv1 = ConvW::load(in_channels, inter_channels, 3, 1, vb.pp("conv1"))?;
        let conv_out = ConvW::load(inter_channels, out_channels, 3, 1, vb.pp("conv_out"))?;
        Ok(Self { conv1, conv_out })
    }
}

impl Module for SimpleConvs {
    /// decoder.rs:50-56
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let x = self.conv1.forward(x, None, ffi::BRN_ACT_NONE)?;
        self.conv_out.forward(&x, None, ffi::BRN_ACT_NONE)
    }
}

/// Basic lateral block — 1x1 conv for channel projection (decoder.rs:59-74)
pub struct BasicLatBlk {
    conv: ConvW,
}

impl BasicLatBlk {
    /// decoder.rs:64-67 — same signature
    pub fn new(in_channels: usize, out_channels: usize, vb: VarBuilder) -> Result<Self> {
        Ok(Self { conv: ConvW::load(in_channels, out_channels, 1, 0, vb.pp("conv"))? })
    }
}

impl Module for BasicLatBlk {
    /// decoder.rs:70-74
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.conv.forward(x, None, ffi::BRN_ACT_NONE)
    }
}

/// every tensor `BasicDecBlk::new(in_channels, out_channels, config, vb)` asks its VarBuilder for (decoder.rs:104-114), names
/// relative to `vb`
pub fn decblk_weight_spec(in_channels: usize, out_channels: usize, config: &DecoderConfig) -> Vec<(String, Vec<usize>)> {
    let ic = if config.inter_channels_adaptive { in_channels / 4 } else { 64usize };   // decoder.rs:94-98
    let mut s: Vec<(String, Vec<usize>)> = Vec::new();
    let bn = |s: &mut Vec<(String, Vec<usize>)>, p: &str, c: usize| {
        for leaf in ["weight", "bias", "running_mean", "running_var"] {
            s.push((format!("{p}.{leaf}"), vec![c]));
        }
    };
    s.push(("conv_in.weight".to_string(), vec![ic, in_channels, 3, 3]));
    s.push(("conv_in.bias".to_string(), vec![ic]));
    bn(&mut s, "bn_in", ic);
    if config.use_aspp_deformable {
        for (name, shape) in crate::aspp::aspp_weight_spec_for(ic, ic) {
            s.push((format!("dec_att.{name}"), shape));
        }
    }
    s.push(("conv_out.weight".to_string(), vec![out_channels, ic, 3, 3]));
    s.push(("conv_out.bias".to_string(), vec![out_channels]));
    bn(&mut s, "bn_out", out_channels);
    s
}

/// Basic decoder block with BatchNorm and optional ASPP: conv_in -> bn_in -> relu -> [aspp] -> conv_out -> bn_out (decoder.rs:78-141).
/// The reference's pub fields are candle layers; here the block is one library call (`brn_decblk_forward`), so the fields are the
/// block's description instead.
pub struct BasicDecBlk {
    named: ffi::NamedTensors,
    pub in_channels: usize,
    pub out_channels: usize,
    pub inter_channels: usize,
    pub use_aspp_deformable: bool,
    /// `BRN_DEFORM_REFERENCE_CPU` (what the reference's CPU path computes, aspp.rs:183-185) or `BRN_DEFORM_DEFORMABLE` (aspp.rs:58-165)
    pub mode: i32,
}

impl BasicDecBlk {
    /// decoder.rs:87-123 — same signature (inter_channels = 64, or in_channels / 4 with `inter_channels_adaptive`, decoder.rs:94-98)
    pub fn new(in_channels: usize, out_channels: usize, config: &DecoderConfig, vb: VarBuilder) -> Result<Self> {
        let inter_channels = if config.inter_channels_adaptive { in_channels / 4 } else { 64 };
        let named = ffi::NamedTensors::from_varbuilder(&vb, &decblk_weight_spec(in_channels, out_channels, config))?;
        Ok(Self { named, in_channels, out_channels, inter_channels, use_aspp_deformable: config.use_aspp_deformable, mode: ffi::BRN_DEFORM_REFERENCE_CPU })
    }
}

impl Module for BasicDecBlk {
    /// decoder.rs:126-141 — x [B,in_channels,H,W] -> [B,out_channels,H,W]
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != self.in_channels {
            candle_core::bail!("expected {} input channels, got {c}", self.in_channels)
        }
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * self.out_channels * h * w];
        let prefix = std::ffi::CString::new("").unwrap();
        ffi::check(unsafe {
            ffi::brn_decblk_forward(self.named.views.as_ptr(), self.named.views.len(), prefix.as_ptr(), self.in_channels as i32,
                                    self.out_channels as i32, self.inter_channels as i32, self.use_aspp_deformable as i32, self.mode, xin.as_ptr(), b as i32, h as i32,
                                    w as i32, out.as_mut_ptr(), ffi::BRN_MEM_HOST, 0, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, self.out_channels, h, w), x.device())
    }
}

/// Residual block with skip connection (decoder.rs:143-217; `#[allow(dead_code)]` in the reference, built by nothing, but a pub type):
/// the BasicDecBlk body + a 1x1 `conv_resi` of the input, added
pub struct ResBlk {
    body: BasicDecBlk,
    conv_resi: ConvW,
}

impl ResBlk {
    /// decoder.rs:156-195 — same signature, same names ("conv_in", "bn_in", "dec_att", "conv_out", "bn_out", "conv_resi")
    pub fn new(in_channels: usize, out_channels: usize, config: &DecoderConfig, vb: VarBuilder) -> Result<Self> {
        let conv_resi = ConvW::load(in_channels, out_channels, 1, 0, vb.pp("conv_resi"))?;
        Ok(Self { body: BasicDecBlk::new(in_channels, out_channels, config, vb)?, conv_resi })
    }
}

impl Module for ResBlk {
    /// decoder.rs:198-216
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let resi = self.conv_resi.forward(x, None, ffi::BRN_ACT_NONE)?;
        self.body.forward(x)? + resi
    }
}
