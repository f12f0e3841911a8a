//! `DeformableConv2d` of the reference (src/deform_conv.rs:17-222) over `brn_deform_conv2d_forward`.
use candle_core::{Module, Result, Tensor};
use candle_nn::VarBuilder;

use crate::hip_ffi as ffi;

/// Deformable convolution v2 (offset + modulator + regular convs; `impl Module` below, deform_conv.rs:218-222).  `mode` is this backend's switch for the two behaviours the
/// reference has: `BRN_DEFORM_REFERENCE_CPU` = the CPU fallback (offsets and modulator computed and discarded,
/// deform_conv.rs:95-98), `BRN_DEFORM_DEFORMABLE` = the Metal path (modulated deformable im2col + matmul, :101-215).
pub struct DeformableConv2d {
    offset_w: Vec<f32>,
    offset_b: Vec<f32>,
    modulator_w: Vec<f32>,
    modulator_b: Vec<f32>,
    regular_w: Vec<f32>,
    regular_b: Option<Vec<f32>>,
    kernel_size: usize,
    padding: usize,
    stride: usize,
    in_channels: usize,
    out_channels: usize,
    pub mode: i32,
}

impl DeformableConv2d {
    /// deform_conv.rs:29-36 — same signature; the three convs are read under the same names ("offset_conv", "modulator_conv",
    /// "regular_conv", each with weight + bias)
    pub fn new(in_channels: usize, out_channels: usize, kernel_size: usize, stride: usize, padding: usize, vb: VarBuilder) -> Result<Self> {
        Self::load(in_channels, out_channels, kernel_size, stride, padding, true, vb)
    }
    /// the same layer with `regular_conv` built by `conv2d_no_bias` (what `aspp::DeformConvASPP` holds, aspp.rs:45)
    pub(crate) fn new_no_bias(in_channels: usize, out_channels: usize, kernel_size: usize, stride: usize, padding: usize, vb: VarBuilder) -> Result<Self> {
        Self::load(in_channels, out_channels, kernel_size, stride, padding, false, vb)
    }
    fn load(in_channels: usize, out_channels: usize, kernel_size: usize, stride: usize, padding: usize, bias: bool, vb: VarBuilder) -> Result<Self> {
        let k = kernel_size;
        let get = |name: &str, shape: &[usize]| -> Result<Vec<f32>> { ffi::to_host(&vb.get(shape, name)?) };
        Ok(Self {
            offset_w: get("offset_conv.weight", &[2 * k * k, in_channels, k, k])?,
            offset_b: get("offset_conv.bias", &[2 * k * k])?,
            modulator_w: get("modulator_conv.weight", &[k * k, in_channels, k, k])?,
            modulator_b: get("modulator_conv.bias", &[k * k])?,
            regular_w: get("regular_conv.weight", &[out_channels, in_channels, k, k])?,
            regular_b: if bias { Some(get("regular_conv.bias", &[out_channels])?) } else { None },
            kernel_size,
            padding,
            stride,
            in_channels,
            out_channels,
            mode: ffi::BRN_DEFORM_DEFORMABLE,
        })
    }

    /// deform_conv.rs:82 — x [B, in_channels, H, W] -> [B, out_channels, H', W']
    pub fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != self.in_channels {
            candle_core::bail!("expected {} input channels, got {c}", self.in_channels)
        }
        let (k, s, p) = (self.kernel_size, self.stride, self.padding);
        let (ho, wo) = ((h + 2 * p - k) / s + 1, (w + 2 * p - k) / s + 1);
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * self.out_channels * ho * wo];
        ffi::check(unsafe {
            ffi::brn_deform_conv2d_forward(xin.as_ptr(), b as i32, c as i32, h as i32, w as i32, self.offset_w.as_ptr(), self.offset_b.as_ptr(),
                                           self.modulator_w.as_ptr(), self.modulator_b.as_ptr(), self.regular_w.as_ptr(), self.regular_b.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()),
                                           self.out_channels as i32, k as i32, s as i32, p as i32, self.mode, out.as_mut_ptr(), ffi::BRN_MEM_HOST, 0,
                                           std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, self.out_channels, ho, wo), x.device())
    }
}

/// deform_conv.rs:218-222
impl Module for DeformableConv2d {
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        DeformableConv2d::forward(self, x)
    }
}
