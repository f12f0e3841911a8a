//! `ASPPDeformable` of the reference (src/aspp.rs:227-333) over `brn_aspp_deformable_forward`: the module `BasicDecBlk` builds
//! (decoder.rs:107-111) — five branches on a 64-channel map (aspp1 and aspp_deforms.{0,1,2} = DeformConvASPP k 1,1,3,7 -> 256, BN,
//! ReLU; global average pool -> 1x1 -> BN -> ReLU -> broadcast), concat 1280, conv1 1x1 + bn1 + ReLU.
use candle_core::{Module, Result, Tensor};
use candle_nn::VarBuilder;

use crate::hip_ffi as ffi;

/// every tensor `ASPPDeformable::new(64, None, vb)` asks its VarBuilder for (aspp.rs:39-45, 247-290), names relative to `vb`
pub fn aspp_weight_spec() -> Vec<(String, Vec<usize>)> {
    let mut s: Vec<(String, Vec<usize>)> = Vec::new();
    let (ic, pl) = (64usize, 256usize);
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
    s.push(("conv1.weight".to_string(), vec![ic, 5 * pl, 1, 1]));
    bn(&mut s, "bn1", ic);
    s
}

/// ASPP with deformable convolutions (aspp.rs:227).  `mode`: `BRN_DEFORM_REFERENCE_CPU` = what the reference's CPU path computes
/// (aspp.rs:183-185), `BRN_DEFORM_DEFORMABLE` = the Metal path (aspp.rs:58-165).
pub struct ASPPDeformable {
    named: ffi::NamedTensors,
    pub mode: i32,
}

impl ASPPDeformable {
    /// aspp.rs:237 — same signature; the HIP backend covers the configuration the model uses: in_channels 64, out_channels None / 64
    pub fn new(in_channels: usize, out_channels: Option<usize>, vb: VarBuilder) -> Result<Self> {
        if in_channels != 64 || out_channels.unwrap_or(in_channels) != 64 {
            candle_core::bail!("ASPPDeformable (hip): 64 -> 64 channels only (the module BasicDecBlk builds, decoder.rs:107-111)")
        }
        Ok(Self { named: ffi::NamedTensors::from_varbuilder(&vb, &aspp_weight_spec())?, mode: ffi::BRN_DEFORM_REFERENCE_CPU })
    }
}

impl Module for ASPPDeformable {
    /// aspp.rs:303 — x [B,64,H,W] -> [B,64,H,W]
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != 64 {
            candle_core::bail!("expected 64 input channels, got {c}")
        }
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * 64 * h * w];
        let prefix = std::ffi::CString::new("").unwrap();
        ffi::check(unsafe {
            ffi::brn_aspp_deformable_forward(self.named.views.as_ptr(), self.named.views.len(), prefix.as_ptr(), self.mode, xin.as_ptr(), b as i32,
                                             h as i32, w as i32, out.as_mut_ptr(), ffi::BRN_MEM_HOST, 0, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, 64, h, w), x.device())
    }
}
