//! `SwinConfig` / `SwinTransformer` of the reference (src/swin.rs:14-88, 718-797) over `brn_swin_*`.
use candle_core::{Result, Tensor};
use candle_nn::VarBuilder;

use crate::hip_ffi as ffi;

/// Configuration for Swin Transformer — the reference's struct, field for field (swin.rs:14-23)
#[derive(Clone, Debug)]
pub struct SwinConfig {
    pub embed_dim: usize,
    pub depths: Vec<usize>,
    pub num_heads: Vec<usize>,
    pub window_size: usize,
    pub mlp_ratio: f64,
    pub patch_size: usize,
    pub in_channels: usize,
    pub drop_path_rate: f64,
}

impl SwinConfig {
    fn of(embed_dim: usize, depths: [usize; 4], num_heads: [usize; 4], window_size: usize) -> Self {
        Self { embed_dim, depths: depths.to_vec(), num_heads: num_heads.to_vec(), window_size, mlp_ratio: 4.0, patch_size: 4, in_channels: 3, drop_path_rate: 0.2 }
    }
    /// swin.rs:27-38
    pub fn swin_t() -> Self { Self::of(96, [2, 2, 6, 2], [3, 6, 12, 24], 7) }
    /// swin.rs:41-52
    pub fn swin_s() -> Self { Self::of(96, [2, 2, 18, 2], [3, 6, 12, 24], 7) }
    /// swin.rs:55-66
    pub fn swin_b() -> Self { Self::of(128, [2, 2, 18, 2], [4, 8, 16, 32], 12) }
    /// swin.rs:69-80 (used by BiRefNet)
    pub fn swin_l() -> Self { Self::of(192, [2, 2, 18, 2], [6, 12, 24, 48], 12) }
    /// swin.rs:83-87
    pub fn stage_channels(&self) -> Vec<usize> { (0..self.depths.len()).map(|i| self.embed_dim * (1 << i)).collect() }

    /// the swin_* fields of `brn_config`
    pub(crate) fn fill(&self, c: &mut ffi::BrnConfig) -> Result<()> {
        if self.depths.len() != 4 || self.num_heads.len() != 4 {
            candle_core::bail!("the HIP backend builds 4-stage Swin transformers (got {} stages)", self.depths.len())
        }
        c.embed_dim = self.embed_dim as i32;
        for i in 0..4 {
            c.depths[i] = self.depths[i] as i32;
            c.num_heads[i] = self.num_heads[i] as i32;
        }
        c.window_size = self.window_size as i32;
        c.mlp_ratio = self.mlp_ratio as f32;
        c.patch_size = self.patch_size as i32;
        c.in_channels = self.in_channels as i32;
        c.drop_path_rate = self.drop_path_rate as f32;
        Ok(())
    }
}

/// (name, shape) of every tensor `SwinTransformer::new` asks its VarBuilder for (swin.rs:726-764)
pub(crate) fn swin_weight_spec(cfg: &SwinConfig, prefix: &str) -> Vec<(String, Vec<usize>)> {
    let mut s: Vec<(String, Vec<usize>)> = Vec::new();
    let mut ln = |s: &mut Vec<(String, Vec<usize>)>, p: String, c: usize| {
        s.push((format!("{p}.weight"), vec![c]));
        s.push((format!("{p}.bias"), vec![c]));
    };
    let (e, p, ic, ws) = (cfg.embed_dim, cfg.patch_size, cfg.in_channels, cfg.window_size);
    s.push((format!("{prefix}patch_embed.proj.weight"), vec![e, ic, p, p]));
    s.push((format!("{prefix}patch_embed.proj.bias"), vec![e]));
    ln(&mut s, format!("{prefix}patch_embed.norm"), e);
    for (i, &depth) in cfg.depths.iter().enumerate() {
        let c = e << i;
        let heads = cfg.num_heads[i];
        let hidden = (c as f64 * cfg.mlp_ratio) as usize;
        for j in 0..depth {
            let bp = format!("{prefix}layers.{i}.blocks.{j}.");
            ln(&mut s, format!("{bp}norm1"), c);
            s.push((format!("{bp}attn.qkv.weight"), vec![3 * c, c]));
            s.push((format!("{bp}attn.qkv.bias"), vec![3 * c]));
            s.push((format!("{bp}attn.proj.weight"), vec![c, c]));
            s.push((format!("{bp}attn.proj.bias"), vec![c]));
            s.push((format!("{bp}attn.relative_position_bias_table"), vec![(2 * ws - 1) * (2 * ws - 1), heads]));
            ln(&mut s, format!("{bp}norm2"), c);
            s.push((format!("{bp}mlp.fc1.weight"), vec![hidden, c]));
            s.push((format!("{bp}mlp.fc1.bias"), vec![hidden]));
            s.push((format!("{bp}mlp.fc2.weight"), vec![c, hidden]));
            s.push((format!("{bp}mlp.fc2.bias"), vec![c]));
        }
        if i + 1 < cfg.depths.len() {
            ln(&mut s, format!("{prefix}layers.{i}.downsample.norm"), 4 * c);
            s.push((format!("{prefix}layers.{i}.downsample.reduction.weight"), vec![2 * c, 4 * c]));
        }
        ln(&mut s, format!("{prefix}norm{i}"), c);
    }
    s
}

/// feature-map sizes of the four stages for an H x W input: ceil(H / patch), then ceil-halved per stage (swin.rs:595, 696-702)
pub(crate) fn stage_dims(h: usize, w: usize, patch: usize) -> [(usize, usize); 4] {
    let (mut a, mut b) = ((h + patch - 1) / patch, (w + patch - 1) / patch);
    let mut out = [(0, 0); 4];
    for o in out.iter_mut() {
        *o = (a, b);
        a = (a + 1) / 2;
        b = (b + 1) / 2;
    }
    out
}

/// Patch Embedding layer (swin.rs:659-715): zero-pad to a multiple of the patch, conv k = stride = patch_size, LayerNorm over the
/// channels of every token.  Over `brn_conv2d_forward` (stride) and `brn_layer_norm_forward`; the pad and the NCHW <-> token-major
/// transposes are candle tensor ops, as in the reference.
pub struct PatchEmbed {
    proj: crate::decoder::ConvW,
    norm: Option<(Vec<f32>, Vec<f32>)>,
    patch_size: usize,
}

impl PatchEmbed {
    /// swin.rs:666-690 — same signature
    pub fn new(patch_size: usize, in_channels: usize, embed_dim: usize, norm: bool, vb: VarBuilder) -> Result<Self> {
        let proj = crate::decoder::ConvW::load_cfg(in_channels, embed_dim, patch_size, patch_size, 0, 1, true, vb.pp("proj"))?;
        let norm = if norm {
            let nb = vb.pp("norm");
            Some((ffi::to_host(&nb.get(embed_dim, "weight")?)?, ffi::to_host(&nb.get(embed_dim, "bias")?)?))
        } else {
            None
        };
        Ok(Self { proj, norm, patch_size })
    }

    /// swin.rs:692-714 — x [B, in_channels, H, W] -> [B, embed_dim, ceil(H / p), ceil(W / p)]
    pub fn forward(&self, x: &Tensor) -> Result<Tensor> {
        let (_, _, h, w) = x.dims4()?;
        let p = self.patch_size;
        let x = if w % p != 0 || h % p != 0 { x.pad_with_zeros(3, 0, (p - w % p) % p)?.pad_with_zeros(2, 0, (p - h % p) % p)? } else { x.clone() };
        let y = self.proj.forward(&x, None, ffi::BRN_ACT_NONE)?;
        let Some((g, bta)) = &self.norm else { return Ok(y) };
        let (b, c, wh, ww) = y.dims4()?;
        let tokens = ffi::to_host(&y.flatten_from(2)?.transpose(1, 2)?.contiguous()?)?;
        let mut out = vec![0f32; tokens.len()];
        ffi::check(unsafe {
            ffi::brn_layer_norm_forward(tokens.as_ptr(), (b * wh * ww) as i32, c as i32, g.as_ptr(), bta.as_ptr(), 1e-5, out.as_mut_ptr(),
                                        ffi::BRN_MEM_HOST, 0, std::ptr::null_mut())
        })?;
        Tensor::from_vec(out, (b, wh * ww, c), y.device())?.transpose(1, 2)?.reshape((b, c, wh, ww))
    }
}

/// Swin Transformer backbone (swin.rs:718-723).  The weights live in HBM: behind a handle of its own when built by
/// `SwinTransformer::new`, behind the owning model's handle when it is the `backbone` field of a `BiRefNet` (birefnet.rs:381).
pub struct SwinTransformer {
    config: SwinConfig,
    inner: SwinImpl,
}
enum SwinImpl {
    Own(*mut ffi::BrnSwin),
    Model(std::sync::Arc<ffi::ModelHandle>),
}
// the library serialises calls on one handle with an internal mutex
unsafe impl Send for SwinTransformer {}
unsafe impl Sync for SwinTransformer {}

impl SwinTransformer {
    /// swin.rs:726 — same signature
    pub fn new(config: SwinConfig, vb: VarBuilder) -> Result<Self> {
        let named = ffi::NamedTensors::from_varbuilder(&vb, &swin_weight_spec(&config, ""))?;
        let mut c: ffi::BrnConfig = unsafe { std::mem::zeroed() };
        unsafe { ffi::brn_config_default_swin_l(&mut c) };
        config.fill(&mut c)?;
        let mut handle = std::ptr::null_mut();
        let empty = std::ffi::CString::new("").unwrap();
        ffi::check(unsafe { ffi::brn_swin_create(&c, named.views.as_ptr(), named.views.len(), empty.as_ptr(), 0, &mut handle) })?;
        Ok(Self { config, inner: SwinImpl::Own(handle) })
    }

    /// the backbone of a whole model (the `backbone` pub field of `BiRefNet`): no weights of its own
    pub(crate) fn shared(config: SwinConfig, model: std::sync::Arc<ffi::ModelHandle>) -> Self {
        Self { config, inner: SwinImpl::Model(model) }
    }

    /// swin.rs:768 — x [B, in_channels, H, W] -> [x1, x2, x3, x4], each [B, C_i, H_i, W_i]
    pub fn forward(&self, x: &Tensor) -> Result<Vec<Tensor>> {
        let (b, c, h, w) = x.dims4()?;
        if c != self.config.in_channels {
            candle_core::bail!("expected {} input channels, got {c}", self.config.in_channels)
        }
        let xin = ffi::to_host(x)?;
        let dims = stage_dims(h, w, self.config.patch_size);
        let chans = self.config.stage_channels();
        let mut bufs: Vec<Vec<f32>> = (0..4).map(|i| vec![0f32; b * chans[i] * dims[i].0 * dims[i].1]).collect();
        let ptrs: Vec<*mut f32> = bufs.iter_mut().map(|v| v.as_mut_ptr()).collect();
        ffi::check(unsafe {
            match &self.inner {
                SwinImpl::Own(handle) => ffi::brn_swin_forward(*handle, xin.as_ptr(), b as i32, h as i32, w as i32, ffi::BRN_MEM_HOST, ptrs.as_ptr(),
                                                               ffi::BRN_MEM_HOST, std::ptr::null_mut()),
                SwinImpl::Model(m) => ffi::brn_model_backbone_forward(m.0, xin.as_ptr(), b as i32, h as i32, w as i32, ffi::BRN_MEM_HOST, ptrs.as_ptr(),
                                                                      ffi::BRN_MEM_HOST, std::ptr::null_mut()),
            }
        })?;
        bufs.into_iter().enumerate().map(|(i, v)| Tensor::from_vec(v, (b, chans[i], dims[i].0, dims[i].1), x.device())).collect()
    }
}

impl Drop for SwinTransformer {
    fn drop(&mut self) {
        if let SwinImpl::Own(handle) = self.inner {
            unsafe { ffi::brn_swin_destroy(handle) }
        }
    }
}
