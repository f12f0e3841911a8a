//! Raw bindings of include/birefnet_hip.h — every `extern "C"` entry of the header, declared once, in header order.
//! tools/check_rust_shim.py compares the `#[repr(C)]` structs and this list with the header.
#![allow(non_camel_case_types)]
use libc::{c_char, c_double, c_float, c_int, c_uchar, c_void, size_t};

pub const BRN_ABI_VERSION: c_int = 1;
pub const BRN_OK: c_int = 0;
/// brn_mem
pub const BRN_MEM_HOST: c_int = 0;
pub const BRN_MEM_DEVICE: c_int = 1;
/// brn_dtype
pub const BRN_F32: c_int = 0;
pub const BRN_F32_SPLIT3: c_int = 1;
pub const BRN_F32_SPLIT2: c_int = 2;
pub const BRN_BF16_OPERANDS: c_int = 3;
pub const BRN_BF16: c_int = 4;
pub const BRN_BF16_DEC_SPLIT2: c_int = 5;
pub const BRN_F32_HALF2: c_int = 6;
pub const BRN_F16: c_int = 7;
/// brn_deform_mode
pub const BRN_DEFORM_REFERENCE_CPU: c_int = 0;
pub const BRN_DEFORM_DEFORMABLE: c_int = 1;
/// brn_act
pub const BRN_ACT_NONE: c_int = 0;
pub const BRN_ACT_RELU: c_int = 1;
pub const BRN_ACT_GELU_ERF: c_int = 2;

/// == `brn_config` (field for field, same order)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct BrnConfig {
    pub size_w: c_int,
    pub size_h: c_int,
    pub backbone: [c_char; 32],
    pub backbone_channels: [c_int; 4],
    pub mul_scl_ipt: c_int,
    pub ms_supervision: c_int,
    pub dec_ipt: c_int,
    pub use_aspp_deformable: c_int,
    pub cxt: [c_int; 3],
    pub n_cxt: c_int,
    pub embed_dim: c_int,
    pub depths: [c_int; 4],
    pub num_heads: [c_int; 4],
    pub window_size: c_int,
    pub mlp_ratio: c_float,
    pub patch_size: c_int,
    pub in_channels: c_int,
    pub drop_path_rate: c_float,
    pub deform_mode: c_int,
}

/// == `brn_named_tensor`
#[repr(C)]
pub struct BrnNamedTensor {
    pub name: *const c_char,
    pub data: *const c_float,
    pub shape: *const i64,
    pub ndim: c_int,
}

pub enum BrnModel {}
pub enum BrnSwin {}

extern "C" {
    pub fn brn_abi_version() -> c_int;
    pub fn brn_last_error() -> *const c_char;
    pub fn brn_device_count(n: *mut c_int) -> c_int;
    pub fn brn_build_info() -> *const c_char;
    pub fn brn_config_default_swin_l(cfg: *mut BrnConfig);
    pub fn brn_config_lateral_channels(cfg: *const BrnConfig, out: *mut c_int);
    pub fn brn_config_x4_channels(cfg: *const BrnConfig) -> c_int;
    pub fn brn_model_create(cfg: *const BrnConfig, weights: *const BrnNamedTensor, n_weights: size_t, device_ordinal: c_int,
                            compute_dtype: c_int, max_batch: c_int, max_h: c_int, max_w: c_int, out: *mut *mut BrnModel) -> c_int;
    pub fn brn_model_create_from_safetensors(cfg: *const BrnConfig, path: *const c_char, prefix: *const c_char, device_ordinal: c_int,
                                             compute_dtype: c_int, max_batch: c_int, max_h: c_int, max_w: c_int,
                                             out: *mut *mut BrnModel) -> c_int;
    pub fn brn_decoder_create(cfg: *const BrnConfig, weights: *const BrnNamedTensor, n_weights: size_t, prefix: *const c_char,
                              device_ordinal: c_int, compute_dtype: c_int, out: *mut *mut BrnModel) -> c_int;
    pub fn brn_model_destroy(m: *mut BrnModel);
    pub fn brn_forward_logits(m: *mut BrnModel, x_nchw: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                              logits_out: *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_forward(m: *mut BrnModel, x_nchw: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                       mask_out: *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_model_backbone_forward(m: *mut BrnModel, x_nchw: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                                      outs: *const *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_model_squeeze_forward(m: *mut BrnModel, x4_nchw: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                                     out: *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_model_decoder_forward(m: *mut BrnModel, x_nchw: *const c_float, x1: *const c_float, x2: *const c_float,
                                     x3: *const c_float, x4: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                                     logits_out: *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_model_set_streams(m: *mut BrnModel, sub_batch_streams: c_int, branch_stream_mask: c_int) -> c_int;
    pub fn brn_model_set_profiling(m: *mut BrnModel, enable: c_int) -> c_int;
    pub fn brn_model_last_timings(m: *mut BrnModel, ms: *mut c_float) -> c_int;
    pub fn brn_model_last_kernel_stats(m: *mut BrnModel, n: c_int, launches: *mut c_int, ms: *mut c_float, flop: *mut c_double,
                                       bytes: *mut c_double, n_out: *mut c_int) -> c_int;
    pub fn brn_kernel_family_name(f: c_int) -> *const c_char;
    pub fn brn_swin_create(cfg: *const BrnConfig, weights: *const BrnNamedTensor, n_weights: size_t, prefix: *const c_char,
                           device_ordinal: c_int, out: *mut *mut BrnSwin) -> c_int;
    pub fn brn_swin_destroy(s: *mut BrnSwin);
    pub fn brn_swin_forward(s: *mut BrnSwin, x_nchw: *const c_float, b: c_int, h: c_int, w: c_int, in_loc: c_int,
                            outs: *const *mut c_float, out_loc: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_set_op_compute(dtype: c_int) -> c_int;
    pub fn brn_linear_forward(x: *const c_float, m: c_int, k: c_int, w: *const c_float, bias: *const c_float, n: c_int, act: c_int,
                              residual: *const c_float, y: *mut c_float, loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_linear_residual_layer_norm_forward(x: *const c_float, m: c_int, k: c_int, w: *const c_float, bias: *const c_float, n: c_int,
                                                  residual: *const c_float, gamma: *const c_float, beta: *const c_float, eps: c_float,
                                                  x_out: *mut c_float, y_out: *mut c_float, loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_layer_norm_forward(x: *const c_float, rows: c_int, c: c_int, gamma: *const c_float, beta: *const c_float, eps: c_float,
                                  y: *mut c_float, loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_conv2d_forward(x: *const c_float, b: c_int, c: c_int, h: c_int, w: c_int, wgt: *const c_float, bias: *const c_float,
                              o: c_int, kh: c_int, kw: c_int, stride: c_int, pad: c_int, dil: c_int, bn_gamma: *const c_float,
                              bn_beta: *const c_float, bn_mean: *const c_float, bn_var: *const c_float, bn_eps: c_float, act: c_int,
                              y: *mut c_float, loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_upsample_bilinear2d(x: *const c_float, b: c_int, c: c_int, h: c_int, w: c_int, out_h: c_int, out_w: c_int,
                                   y: *mut c_float, loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_window_attention_forward(x: *const c_float, b: c_int, h: c_int, w: c_int, c: c_int, heads: c_int, window_size: c_int,
                                        shift: c_int, qkv_w: *const c_float, qkv_b: *const c_float, proj_w: *const c_float,
                                        proj_b: *const c_float, rel_table: *const c_float, y: *mut c_float, loc: c_int,
                                        device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_patch_merging_forward(x: *const c_float, b: c_int, h: c_int, w: c_int, c: c_int, norm_g: *const c_float,
                                     norm_b: *const c_float, reduction_w: *const c_float, y: *mut c_float, loc: c_int,
                                     device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_deform_conv2d_forward(x: *const c_float, b: c_int, c: c_int, h: c_int, w: c_int, offset_w: *const c_float,
                                     offset_b: *const c_float, mod_w: *const c_float, mod_b: *const c_float, wgt: *const c_float,
                                     bias: *const c_float, o: c_int, k: c_int, stride: c_int, pad: c_int, mode: c_int, y: *mut c_float,
                                     loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_aspp_deformable_forward(weights: *const BrnNamedTensor, n_weights: usize, prefix: *const c_char, in_channels: c_int,
                                       out_channels: c_int, mode: c_int, x: *const c_float, b: c_int, h: c_int, w: c_int, y: *mut c_float,
                                       loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_decblk_forward(weights: *const BrnNamedTensor, n_weights: usize, prefix: *const c_char, in_channels: c_int, out_channels: c_int,
                              inter_channels: c_int, use_aspp: c_int, mode: c_int, x: *const c_float, b: c_int, h: c_int, w: c_int, y: *mut c_float, loc: c_int,
                              device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_preprocess_image(pixels: *const c_uchar, h: c_int, w: c_int, channels: c_int, s: c_int, x_nchw_out: *mut c_float,
                                out_loc: c_int, device_ordinal: c_int, stream: *mut c_void) -> c_int;
    pub fn brn_infer_images_u8(m: *mut BrnModel, n: c_int, pixels: *const *const c_uchar, heights: *const c_int, widths: *const c_int, channels: c_int,
                               s: c_int, masks: *const *mut c_uchar, stream: *mut c_void) -> c_int;
    pub fn brn_postprocess_mask(logits: *const c_float, s: c_int, in_loc: c_int, apply_sigmoid: c_int, out_h: c_int, out_w: c_int,
                                mask_out: *mut c_uchar, device_ordinal: c_int, stream: *mut c_void) -> c_int;
}

/// an HBM-resident model (`brn_model*`) shared by the pieces of a `BiRefNet` (`backbone`, `squeeze_module`, `decoder`); destroyed with
/// its last owner.  Forwards on one handle are serialised inside the library (mutex + stream event): safe to share between threads.
pub(crate) struct ModelHandle(pub(crate) *mut BrnModel);
unsafe impl Send for ModelHandle {}
unsafe impl Sync for ModelHandle {}
impl Drop for ModelHandle {
    fn drop(&mut self) {
        unsafe { brn_model_destroy(self.0) }
    }
}

/// status -> `candle_core::Result`: the message of `brn_last_error()` (thread-local) becomes `candle_core::Error::Msg`, the error
/// type the reference already returns for its own failures (aspp.rs:101,148).
pub fn check(status: c_int) -> candle_core::Result<()> {
    if status == BRN_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(brn_last_error()) }.to_string_lossy().into_owned();
    Err(candle_core::Error::Msg(msg))
}

/// `Tensor` [.., f32 on any candle device] -> contiguous host `Vec<f32>` (the form infer_image.rs:67 builds its input in)
pub fn to_host(t: &candle_core::Tensor) -> candle_core::Result<Vec<f32>> {
    t.to_dtype(candle_core::DType::F32)?.flatten_all()?.to_vec1::<f32>()
}

/// (name, shape) list -> owned host copies + the `brn_named_tensor` views over them
pub struct NamedTensors {
    _names: Vec<std::ffi::CString>,
    _data: Vec<Vec<f32>>,
    _shapes: Vec<Vec<i64>>,
    pub views: Vec<BrnNamedTensor>,
}

impl NamedTensors {
    /// walks `vb` with the reference's own names: a missing name or a wrong shape is the `Err` `vb.get` returns today
    pub fn from_varbuilder(vb: &candle_nn::VarBuilder, spec: &[(String, Vec<usize>)]) -> candle_core::Result<Self> {
        let mut names = Vec::with_capacity(spec.len());
        let mut data = Vec::with_capacity(spec.len());
        let mut shapes = Vec::with_capacity(spec.len());
        for (name, shape) in spec {
            let t = vb.get(shape.as_slice(), name)?;
            names.push(std::ffi::CString::new(name.as_str()).map_err(|e| candle_core::Error::Msg(e.to_string()))?);
            data.push(to_host(&t)?);
            shapes.push(shape.iter().map(|&d| d as i64).collect::<Vec<i64>>());
        }
        let views = (0..names.len())
            .map(|i| BrnNamedTensor { name: names[i].as_ptr(), data: data[i].as_ptr(), shape: shapes[i].as_ptr(), ndim: shapes[i].len() as c_int })
            .collect();
        Ok(Self { _names: names, _data: data, _shapes: shapes, views })
    }
}
