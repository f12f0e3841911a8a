//! `BiRefNetConfig` / `BiRefNet` / `SqueezeModule` / `BiRefNetDecoder` of the reference (src/birefnet.rs:13-67, 70-94, 121-377,
//! 380-476) over `brn_model_*` / `brn_forward*`.  One HBM-resident model handle is shared by the three pub fields the reference
//! exposes (`backbone`: a `SwinTransformer`, as in the reference; `squeeze_module`; `decoder`), which examples/bench_inference.rs:34,77,83 drive
//! one by one.
use std::sync::Arc;

use candle_core::{Module, Result, Tensor};
use candle_nn::VarBuilder;

use crate::hip_ffi as ffi;
use crate::swin::{swin_weight_spec, SwinConfig, SwinTransformer};

/// BiRefNet configuration — the reference's struct, field for field (birefnet.rs:13-30)
#[derive(Clone)]
pub struct BiRefNetConfig {
    pub size: (usize, usize),
    pub backbone: String,
    pub backbone_channels: Vec<usize>,
    pub mul_scl_ipt: bool,
    pub ms_supervision: bool,
    pub dec_ipt: bool,
    pub use_aspp_deformable: bool,
    pub cxt: Vec<usize>,
}

impl Default for BiRefNetConfig {
    /// birefnet.rs:32-46
    fn default() -> Self {
        Self {
            size: (1024, 1024),
            backbone: "swin_v1_l".to_string(),
            backbone_channels: vec![192, 384, 768, 1536],
            mul_scl_ipt: true,
            ms_supervision: true,
            dec_ipt: true,
            use_aspp_deformable: true,
            cxt: vec![192, 384, 768],
        }
    }
}

impl BiRefNetConfig {
    /// birefnet.rs:50-53
    pub fn lateral_channels(&self) -> Vec<usize> {
        let m = if self.mul_scl_ipt { 2 } else { 1 };
        self.backbone_channels.iter().map(|c| c * m).collect()
    }
    /// birefnet.rs:56-61
    pub fn x4_channels(&self) -> usize {
        let m = if self.mul_scl_ipt { 2 } else { 1 };
        self.backbone_channels[3] * m + self.cxt.iter().map(|c| c * m).sum::<usize>()
    }
    /// birefnet.rs:64-66
    pub fn swin_l() -> Self {
        Self::default()
    }

    pub(crate) fn to_c(&self, deform_mode: i32) -> Result<ffi::BrnConfig> {
        let mut c: ffi::BrnConfig = unsafe { std::mem::zeroed() };
        unsafe { ffi::brn_config_default_swin_l(&mut c) };       // BiRefNet::new always builds swin_l (birefnet.rs:390-391)
        if self.backbone_channels.len() != 4 || self.cxt.len() > 3 {
            candle_core::bail!("backbone_channels must have 4 entries and cxt at most 3")
        }
        c.size_w = self.size.0 as i32;
        c.size_h = self.size.1 as i32;
        for (d, s) in c.backbone.iter_mut().zip(self.backbone.bytes().take(31)) {
            *d = s as libc::c_char;
        }
        for i in 0..4 {
            c.backbone_channels[i] = self.backbone_channels[i] as i32;
        }
        c.mul_scl_ipt = self.mul_scl_ipt as i32;
        c.ms_supervision = self.ms_supervision as i32;
        c.dec_ipt = self.dec_ipt as i32;
        c.use_aspp_deformable = self.use_aspp_deformable as i32;
        c.cxt = [0; 3];
        for (i, v) in self.cxt.iter().enumerate() {
            c.cxt[i] = *v as i32;
        }
        c.n_cxt = self.cxt.len() as i32;
        c.deform_mode = deform_mode;
        Ok(c)
    }
}

// ---- the weight-name contract (birefnet.rs:170-273, decoder.rs:104-114, aspp.rs:39-45,247-290) ----
type Spec = Vec<(String, Vec<usize>)>;

fn conv(s: &mut Spec, p: &str, o: usize, cin: usize, k: usize, bias: bool) {
    s.push((format!("{p}.weight"), vec![o, cin, k, k]));
    if bias {
        s.push((format!("{p}.bias"), vec![o]));
    }
}
fn bn(s: &mut Spec, p: &str, c: usize) {
    for leaf in ["weight", "bias", "running_mean", "running_var"] {
        s.push((format!("{p}.{leaf}"), vec![c]));
    }
}
/// BasicDecBlk::new + ASPPDeformable::new
fn decblk(s: &mut Spec, p: &str, cin: usize, cout: usize) {
    conv(s, &format!("{p}conv_in"), 64, cin, 3, true);
    bn(s, &format!("{p}bn_in"), 64);
    let ap = format!("{p}dec_att.");
    for (module, k) in [("aspp1", 1usize), ("aspp_deforms.0", 1), ("aspp_deforms.1", 3), ("aspp_deforms.2", 7)] {
        let cp = format!("{ap}{module}.atrous_conv.");
        conv(s, &format!("{cp}offset_conv"), 2 * k * k, 64, k, true);
        conv(s, &format!("{cp}modulator_conv"), k * k, 64, k, true);
        conv(s, &format!("{cp}regular_conv"), 256, 64, k, false);
        bn(s, &format!("{ap}{module}.bn"), 256);
    }
    conv(s, &format!("{ap}global_avg_pool.1"), 256, 64, 1, false);
    bn(s, &format!("{ap}global_avg_pool.2"), 256);
    conv(s, &format!("{ap}conv1"), 64, 1280, 1, false);
    bn(s, &format!("{ap}bn1"), 64);
    conv(s, &format!("{p}conv_out"), cout, 64, 3, true);
    bn(s, &format!("{p}bn_out"), cout);
}

/// every tensor `BiRefNet::new` asks its VarBuilder for, including the loaded-but-unused heads (birefnet.rs:150-166, 229-243)
pub fn weight_spec(config: &BiRefNetConfig) -> Spec {
    let mut s = swin_weight_spec(&SwinConfig::swin_l(), "bb.");
    let lat = config.lateral_channels();
    decblk(&mut s, "squeeze_module.0.", config.x4_channels(), lat[3]);
    let ipt_out = [48usize, 96, 192, 384, 384];
    let ipt_in = [3usize, ipt_out[0], lat[0] / 2, lat[2] / 2, lat[3]];
    for i in 0..5 {
        let p = format!("decoder.ipt_blk{}.", i + 1);
        conv(&mut s, &format!("{p}conv1"), 64, ipt_in[i], 3, true);
        conv(&mut s, &format!("{p}conv_out"), ipt_out[i], 64, 3, true);
    }
    let dec_out = [lat[2], lat[1], lat[0], lat[0] / 2];
    let dec_in = [lat[3] + ipt_out[4], dec_out[0] + ipt_out[3], dec_out[1] + ipt_out[2], dec_out[2] + ipt_out[1]];
    for (i, n) in [4, 3, 2, 1].iter().enumerate() {
        decblk(&mut s, &format!("decoder.decoder_block{n}."), dec_in[i], dec_out[i]);
    }
    for (n, c) in [(4, lat[2]), (3, lat[1]), (2, lat[0])] {
        conv(&mut s, &format!("decoder.lateral_block{n}.conv"), c, c, 1, true);
    }
    for (i, n) in [4, 3, 2].iter().enumerate() {
        let c = dec_out[i];
        conv(&mut s, &format!("decoder.gdt_convs_{n}.0"), 16, c, 3, true);
        bn(&mut s, &format!("decoder.gdt_convs_{n}.1"), 16);
        conv(&mut s, &format!("decoder.gdt_convs_attn_{n}.0"), 1, 16, 1, true);
        conv(&mut s, &format!("decoder.gdt_convs_pred_{n}.0"), 1, 16, 1, true);
        conv(&mut s, &format!("decoder.conv_ms_spvn_{n}"), 1, c, 1, true);
    }
    conv(&mut s, "decoder.conv_out1.0", 1, dec_out[3] + ipt_out[0], 1, true);
    s
}

/// GDT (Gradient Detail) convolutions — conv 3x3 -> 16 + BatchNorm + ReLU (birefnet.rs:97-118): one `brn_conv2d_forward` call with the
/// batch norm folded and the ReLU in the epilogue
pub struct GdtConvs {
    inner: crate::decoder::ConvBnRelu,
}

impl GdtConvs {
    /// birefnet.rs:103-108 — same signature: conv under `vb.pp("0")`, batch norm under `vb.pp("1")`
    pub fn new(in_channels: usize, vb: VarBuilder) -> Result<Self> {
        Ok(Self { inner: crate::decoder::ConvBnRelu::load(in_channels, 16, vb.pp("0"), vb.pp("1"))? })
    }
}

impl Module for GdtConvs {
    /// birefnet.rs:111-117
    fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.inner.forward(x)
    }
}

// ---- the shared model handle: hip_ffi::ModelHandle ----
type Handle = ffi::ModelHandle;

/// Squeeze module (birefnet.rs:70-94): inside a `BiRefNet` it is a view of the shared model handle; built on its own
/// (`SqueezeModule::new`) it is one `BasicDecBlk` under `vb.pp("0")`, as in the reference
pub struct SqueezeModule {
    inner: SqueezeImpl,
}
enum SqueezeImpl {
    Shared { h: Arc<Handle>, out_channels: usize },
    Own(crate::decoder::BasicDecBlk),
}
impl SqueezeModule {
    /// birefnet.rs:75-83 — same signature (the ASPP is always on: birefnet.rs:76-79)
    pub fn new(in_channels: usize, out_channels: usize, vb: VarBuilder) -> Result<Self> {
        let config = crate::decoder::DecoderConfig { use_aspp_deformable: true, inter_channels_adaptive: false };
        let mut blk = crate::decoder::BasicDecBlk::new(in_channels, out_channels, &config, vb.pp("0"))?;
        blk.mode = deform_from_env();
        Ok(Self { inner: SqueezeImpl::Own(blk) })
    }
}
impl Module for SqueezeModule {
    fn forward(&self, x4: &Tensor) -> Result<Tensor> {
        let (h_, out_channels) = match &self.inner {
            SqueezeImpl::Own(blk) => return blk.forward(x4),
            SqueezeImpl::Shared { h, out_channels } => (h, *out_channels),
        };
        let (b, _c, h, w) = x4.dims4()?;
        let xin = ffi::to_host(x4)?;
        let mut out = vec![0f32; b * out_channels * h * w];
        ffi::check(unsafe {
            ffi::brn_model_squeeze_forward(h_.0, xin.as_ptr(), b as i32, h as i32, w as i32, ffi::BRN_MEM_HOST, out.as_mut_ptr(),
                                           ffi::BRN_MEM_HOST, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, out_channels, h, w), x4.device())
    }
}

/// the decoder's share of `weight_spec` (names relative to the decoder's own VarBuilder: birefnet.rs:170-273)
pub fn decoder_weight_spec(config: &BiRefNetConfig) -> Spec {
    weight_spec(config).into_iter().filter_map(|(n, shp)| n.strip_prefix("decoder.").map(|r| (r.to_string(), shp))).collect()
}

/// BiRefNet decoder (birefnet.rs:121-377)
pub struct BiRefNetDecoder {
    h: Arc<Handle>,
}
impl BiRefNetDecoder {
    /// birefnet.rs:170 — same signature: a decoder on its own (`vb` at the decoder's prefix, `vb.pp("decoder")` in birefnet.rs:401),
    /// behind a handle that holds only the decoder's weights (`brn_decoder_create`)
    pub fn new(config: BiRefNetConfig, vb: VarBuilder) -> Result<Self> {
        let named = ffi::NamedTensors::from_varbuilder(&vb, &decoder_weight_spec(&config))?;
        let c = config.to_c(deform_from_env())?;
        let mut raw = std::ptr::null_mut();
        let prefix = std::ffi::CString::new("").unwrap();
        ffi::check(unsafe {
            ffi::brn_decoder_create(&c, named.views.as_ptr(), named.views.len(), prefix.as_ptr(), device_from_env(), compute_from_env(), &mut raw)
        })?;
        Ok(Self { h: Arc::new(ffi::ModelHandle(raw)) })
    }
    /// birefnet.rs:278 — same signature
    pub fn forward(&self, x: &Tensor, x1: &Tensor, x2: &Tensor, x3: &Tensor, x4: &Tensor) -> Result<Tensor> {
        let (b, _c, h, w) = x.dims4()?;
        let (xi, a1, a2, a3, a4) = (ffi::to_host(x)?, ffi::to_host(x1)?, ffi::to_host(x2)?, ffi::to_host(x3)?, ffi::to_host(x4)?);
        let mut out = vec![0f32; b * h * w];
        ffi::check(unsafe {
            ffi::brn_model_decoder_forward(self.h.0, xi.as_ptr(), a1.as_ptr(), a2.as_ptr(), a3.as_ptr(), a4.as_ptr(), b as i32, h as i32, w as i32,
                                           ffi::BRN_MEM_HOST, out.as_mut_ptr(), ffi::BRN_MEM_HOST, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, 1, h, w), x.device())
    }
}

/// BiRefNet model with Swin Transformer backbone (birefnet.rs:380-385): the same four pub fields
pub struct BiRefNet {
    pub config: BiRefNetConfig,
    pub backbone: SwinTransformer,
    pub squeeze_module: SqueezeModule,
    pub decoder: BiRefNetDecoder,
    h: Arc<Handle>,
    device: i32,
}

impl BiRefNet {
    /// birefnet.rs:389 — same signature.  Compute mode and deform mode of the HIP backend come from the environment
    /// (`BIREFNET_HIP_COMPUTE` = f32 | f32_split3 (default) | f32_split2 | f32_half2 | bf16 | f16 | bf16_dec_split2; `BIREFNET_HIP_DEFORM` =
    /// reference_cpu (default) | deformable) so that the reference's call sites compile unchanged.
    /// The HIP device and the largest batch the workspace is planned for come from `BIREFNET_HIP_DEVICE` (default 0) and
    /// `BIREFNET_HIP_MAX_BATCH` (default 1; larger batches re-plan on first use); `new_on` takes them as arguments.
    pub fn new(config: BiRefNetConfig, vb: VarBuilder) -> Result<Self> {
        Self::new_on(config, vb, device_from_env(), max_batch_from_env())
    }

    /// `new` on an explicit HIP device ordinal with the workspace planned for `max_batch` images: what a data-parallel caller uses,
    /// one model (= one weight replica, one handle) per GPU of the node and a contiguous shard of the batch each — images are
    /// independent units, there is no collective on the path (bench.py does the same with one process per GPU).
    pub fn new_on(config: BiRefNetConfig, vb: VarBuilder, device: i32, max_batch: usize) -> Result<Self> {
        let named = ffi::NamedTensors::from_varbuilder(&vb, &weight_spec(&config))?;
        let c = config.to_c(deform_from_env())?;
        let mut raw = std::ptr::null_mut();
        ffi::check(unsafe {
            ffi::brn_model_create(&c, named.views.as_ptr(), named.views.len(), device, compute_from_env(), max_batch.max(1) as i32,
                                  config.size.1 as i32, config.size.0 as i32, &mut raw)
        })?;
        Ok(Self::wrap(config, raw, device))
    }

    /// `VarBuilder::from_mmaped_safetensors(&[path], DType::F32, &device)` + `BiRefNet::new` in one call (infer_image.rs:35-40):
    /// the library memory-maps and parses the checkpoint itself, no tensor crosses the FFI
    pub fn from_safetensors(config: BiRefNetConfig, path: &std::path::Path) -> Result<Self> {
        Self::from_safetensors_on(config, path, device_from_env(), max_batch_from_env())
    }

    /// `from_safetensors` on an explicit device / planned batch (see `new_on`)
    pub fn from_safetensors_on(config: BiRefNetConfig, path: &std::path::Path, device: i32, max_batch: usize) -> Result<Self> {
        let p = std::ffi::CString::new(path.to_string_lossy().as_bytes()).map_err(|e| candle_core::Error::Msg(e.to_string()))?;
        let c = config.to_c(deform_from_env())?;
        let mut raw = std::ptr::null_mut();
        ffi::check(unsafe {
            ffi::brn_model_create_from_safetensors(&c, p.as_ptr(), std::ptr::null(), device, compute_from_env(), max_batch.max(1) as i32,
                                                   config.size.1 as i32, config.size.0 as i32, &mut raw)
        })?;
        Ok(Self::wrap(config, raw, device))
    }

    fn wrap(config: BiRefNetConfig, raw: *mut ffi::BrnModel, device: i32) -> Self {
        let h = Arc::new(ffi::ModelHandle(raw));
        let out_channels = config.lateral_channels()[3];
        Self { config, backbone: SwinTransformer::shared(SwinConfig::swin_l(), h.clone()), squeeze_module: SqueezeModule { inner: SqueezeImpl::Shared { h: h.clone(), out_channels } }, decoder: BiRefNetDecoder { h: h.clone() }, h, device }
    }

    /// HIP device ordinal this model's weights and workspace live on
    pub fn device_ordinal(&self) -> i32 {
        self.device
    }

    /// `forward_logits` (or `forward` with `sigmoid`) on buffers that already live in this model's device memory: `x_dev` =
    /// [b,3,h,w] f32 NCHW, `logits_dev` = [b,1,h,w] f32, both HIP device pointers; the launches are enqueued on `stream` (a
    /// `hipStream_t`, null = the default stream) and NOT waited for.  No host staging: the `Tensor` entry points copy 12.6 MB in
    /// and 4.2 MB out per 1024x1024 image through host `Vec<f32>`s, this one copies nothing (what bench.py times).
    ///
    /// # Safety
    /// the pointers must be valid device allocations of at least those sizes on device `device_ordinal()` and stay alive until
    /// the stream has executed the forward.
    pub unsafe fn forward_logits_device(&self, x_dev: *const f32, b: usize, h: usize, w: usize, logits_dev: *mut f32, sigmoid: bool,
                                        stream: *mut libc::c_void) -> Result<()> {
        let f = if sigmoid { ffi::brn_forward } else { ffi::brn_forward_logits };
        ffi::check(f(self.h.0, x_dev, b as i32, h as i32, w as i32, ffi::BRN_MEM_DEVICE, logits_dev, ffi::BRN_MEM_DEVICE, stream))
    }

    fn run(&self, x: &Tensor, sigmoid: bool) -> Result<Tensor> {
        let (b, c, h, w) = x.dims4()?;
        if c != 3 {
            candle_core::bail!("expected [B,3,H,W], got {c} channels")
        }
        let xin = ffi::to_host(x)?;
        let mut out = vec![0f32; b * h * w];
        let f = if sigmoid { ffi::brn_forward } else { ffi::brn_forward_logits };
        ffi::check(unsafe { f(self.h.0, xin.as_ptr(), b as i32, h as i32, w as i32, ffi::BRN_MEM_HOST, out.as_mut_ptr(), ffi::BRN_MEM_HOST, std::ptr::null_mut()) })?;
        Tensor::from_vec(out, (b, 1, h, w), x.device())
    }

    /// birefnet.rs:412 — raw logits [B,1,H,W]
    pub fn forward_logits(&self, x: &Tensor) -> Result<Tensor> {
        self.run(x, false)
    }

    /// birefnet.rs:466 — sigmoid(forward_logits), fused in the library's final kernel
    pub fn forward(&self, x: &Tensor) -> Result<Tensor> {
        self.run(x, true)
    }
}

/// birefnet.rs:472-476
impl Module for BiRefNet {
    fn forward(&self, xs: &Tensor) -> Result<Tensor> {
        BiRefNet::forward(self, xs)
    }
}

fn compute_from_env() -> i32 {
    match std::env::var("BIREFNET_HIP_COMPUTE").as_deref() {
        Ok("f32") => ffi::BRN_F32,
        Ok("f32_split2") => ffi::BRN_F32_SPLIT2,
        Ok("bf16") => ffi::BRN_BF16,
        Ok("bf16_dec_split2") => ffi::BRN_BF16_DEC_SPLIT2,
        Ok("f32_half2") => ffi::BRN_F32_HALF2,
        Ok("f16") => ffi::BRN_F16,
        _ => ffi::BRN_F32_SPLIT3,
    }
}
fn device_from_env() -> i32 {
    std::env::var("BIREFNET_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}
fn max_batch_from_env() -> usize {
    std::env::var("BIREFNET_HIP_MAX_BATCH").ok().and_then(|v| v.parse().ok()).unwrap_or(1)
}
fn deform_from_env() -> i32 {
    match std::env::var("BIREFNET_HIP_DEFORM").as_deref() {
        Ok("deformable") => ffi::BRN_DEFORM_DEFORMABLE,
        _ => ffi::BRN_DEFORM_REFERENCE_CPU,
    }
}
