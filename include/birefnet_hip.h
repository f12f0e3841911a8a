/*
 * birefnet_hip.h — C ABI of libbirefnet_hip.so (MI355X / gfx950 native BiRefNet inference path).
 *
 * Every entry point is what the reference crate's FFI for the hot path would bind.  Citations are into
 * /root/reference (imperatormk/candle-birefnet); the Rust shim a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.  All tensors are contiguous fp32.
 *   - brn_status: 0 = ok, non-zero = error; the message is available from brn_last_error() (thread-local).
 *     No exception or abort crosses the boundary (the one reference panic, swin.rs:352, is an error here).
 *   - brn_mem says where a caller buffer lives.  BRN_MEM_DEVICE pointers are HIP device pointers of the
 *     model's device; BRN_MEM_HOST buffers are staged through HBM by the library (H2D/D2H on `stream`).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are asynchronous for
 *     BRN_MEM_DEVICE buffers and synchronous (stream-synchronised before return) for BRN_MEM_HOST output.
 *     A model handle owns ONE workspace: forwards on it are serialised on the host by a mutex and, when consecutive calls
 *     use different streams, on the GPU by an event (the later call's stream waits for the earlier forward's last kernel),
 *     so a handle may be driven from several threads / streams; for concurrent forwards use one handle per stream.
 *   - image tensors at this boundary are NCHW like candle's (infer_image.rs:67); the library's internal
 *     layout (NHWC) is not visible.
 *   - there is NO CPU fallback: every call needs a HIP device and fails with BRN_ERR_NO_DEVICE without one.
 */
#ifndef BIREFNET_HIP_H
#define BIREFNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRN_ABI_VERSION 1

typedef int brn_status;
enum {
    BRN_OK = 0,
    BRN_ERR_INVALID_ARG = 1,   /* bad shape / null pointer / unsupported config          */
    BRN_ERR_MISSING_TENSOR = 2,/* weight name absent (candle: Err from vb.get)            */
    BRN_ERR_SHAPE = 3,         /* weight present with the wrong shape                     */
    BRN_ERR_NO_DEVICE = 4,     /* no usable HIP device                                    */
    BRN_ERR_HIP = 5,           /* a HIP runtime call failed                               */
    BRN_ERR_OOM = 6            /* device workspace allocation failed                      */
};

typedef enum { BRN_MEM_HOST = 0, BRN_MEM_DEVICE = 1 } brn_mem;
/* Arithmetic of the contraction kernels.  In the BRN_F32* modes (and BRN_BF16_OPERANDS) everything else — LayerNorm, softmax, epilogues,
 * storage — is fp32 (the reference runs DType::F32, infer_image.rs:26):
 *   BRN_F32            fp32 operands on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32), exact fmaf chain
 *   BRN_F32_SPLIT3     fp32 operands split error-free into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate:
 *                      fp32-class accuracy (dropped terms < 2^-24 of a product) at 2.67x the fp32-MFMA rate
 *   BRN_F32_SPLIT2     2 planes, 3 MFMAs: ~2^-16 relative per product
 *   BRN_F32_HALF2      fp32 operands as TWO fp16 planes of the operand scaled by a power of two (weights: per tensor, so that the largest
 *                      lands in (2^13, 2^14]; activations: by 8), 3 fp16 MFMAs per product (hh, hl, lh), fp32 accumulate, the
 *                      accumulator un-scaled exactly: ~2^-22 relative per product — fp32-class accuracy at twice BRN_F32_SPLIT3's
 *                      matrix rate.  Range: a GEMM input activation of magnitude >= 8190 overflows fp16 and the logits come back
 *                      NaN (never silently wrong); BRN_F32_SPLIT3 has fp32's full range.
 *   BRN_BF16_OPERANDS  operands rounded to bf16, fp32 accumulate, fp32 storage: superseded by BRN_BF16; the value is reserved, the
 *                      product library answers BRN_ERR_INVALID_ARG (only libbirefnet_hip_diag.so, `make diag`, still builds it)
 *   BRN_BF16           the bf16 throughput mode of BASELINE configs[2..4]: activations AND weights live in HBM as bf16,
 *                      bf16 MFMA with fp32 accumulation, fp32 statistics inside LayerNorm / softmax / GAP; x and the logits
 *                      stay fp32 at this boundary.  Parity is informational (error vs the fp32 oracle is reported).
 *   BRN_F16            the same graph and kernels as BRN_BF16 with fp16 as the 16-bit type: activations and weights live in HBM as fp16, fp16 MFMA
 *                      (the bf16 rate), fp32 accumulate / statistics / residual stream.  Same bytes and speed, 3 more mantissa bits: the error
 *                      against the fp32 oracle is ~8x smaller than BRN_BF16's (masks within 1e-3 of the reference's).
 *                      Range: a stored activation of magnitude >= 65520 becomes Inf (the logits NaN); BRN_BF16 has fp32's exponent range.
 *   BRN_BF16_DEC_SPLIT2  mixed: the Swin backbone (79 % of the FLOPs) as BRN_BF16, everything after it — multi-scale / context fusion,
 *                      squeeze module, decoder — as BRN_F32_SPLIT2 on fp32 maps.  The decoder's ~20 chained bf16 roundings are most of mode
 *                      BRN_BF16's error; this mode pays for removing them where they are cheapest. */
typedef enum { BRN_F32 = 0, BRN_F32_SPLIT3 = 1, BRN_F32_SPLIT2 = 2, BRN_BF16_OPERANDS = 3, BRN_BF16 = 4, BRN_BF16_DEC_SPLIT2 = 5, BRN_F32_HALF2 = 6, BRN_F16 = 7 } brn_dtype;

/* D1 of SURVEY.md §8: what DeformConvASPP::forward computes.
 * REFERENCE_CPU = aspp.rs:183-185 (offset/modulator discarded, regular_conv(x)) — the graded parity target.
 * DEFORMABLE    = aspp.rs:58-165 (Metal path: modulated deformable im2col + matmul).                     */
typedef enum { BRN_DEFORM_REFERENCE_CPU = 0, BRN_DEFORM_DEFORMABLE = 1 } brn_deform_mode;

/* activation selector of the op-level entry points */
typedef enum { BRN_ACT_NONE = 0, BRN_ACT_RELU = 1, BRN_ACT_GELU_ERF = 2 } brn_act;

/* Mirrors BiRefNetConfig (birefnet.rs:13-67) + SwinConfig (swin.rs:14-88), field for field,
 * decorative fields included.  BiRefNet::new ignores `backbone` and always builds swin_l (birefnet.rs:390-391);
 * here the swin_* fields are honoured so that reduced-depth models can be built for tests. */
typedef struct brn_config {
    int size_w, size_h;            /* BiRefNetConfig.size (never read in forward)                  */
    char backbone[32];             /* "swin_v1_l" (ignored, as in the reference)                   */
    int backbone_channels[4];      /* [192,384,768,1536]                                           */
    int mul_scl_ipt;               /* 1                                                            */
    int ms_supervision;            /* 1 (decorative)                                               */
    int dec_ipt;                   /* 1 (decorative: decoder always uses ipt blocks)               */
    int use_aspp_deformable;       /* 1                                                            */
    int cxt[3]; int n_cxt;         /* [192,384,768], 3                                             */
    /* SwinConfig */
    int embed_dim;                 /* 192 */
    int depths[4];                 /* [2,2,18,2] */
    int num_heads[4];              /* [6,12,24,48] */
    int window_size;               /* 12 */
    float mlp_ratio;               /* 4.0 */
    int patch_size;                /* 4 */
    int in_channels;               /* 3 */
    float drop_path_rate;          /* 0.2 (unused) */
    /* library-side switch (not in the reference config) */
    int deform_mode;               /* brn_deform_mode */
} brn_config;

/* One weight tensor as VarBuilder would hand it over (names: SURVEY.md App. A; shapes as in the safetensors). */
typedef struct brn_named_tensor {
    const char* name;
    const float* data;   /* host pointer, fp32, contiguous; the library copies, the caller keeps ownership */
    const int64_t* shape;
    int ndim;
} brn_named_tensor;

typedef struct brn_model brn_model;     /* BiRefNet (birefnet.rs:380-385) */
typedef struct brn_swin brn_swin;       /* stand-alone SwinTransformer (swin.rs:718-723) */

/* ---- library ------------------------------------------------------------------------------------------ */
int brn_abi_version(void);
const char* brn_last_error(void);               /* thread-local; wrapped into candle_core::Error::Msg by the shim */
brn_status brn_device_count(int* n);
const char* brn_build_info(void);               /* "gfx950 hipcc <ver> ..." */

/* ---- config ------------------------------------------------------------------------------------------- */
/* BiRefNetConfig::swin_l() / Default (birefnet.rs:32-46, 64-66) + SwinConfig::swin_l() (swin.rs:69-80). */
void brn_config_default_swin_l(brn_config* cfg);
/* BiRefNetConfig::lateral_channels (birefnet.rs:50-53) and x4_channels (birefnet.rs:56-61). */
void brn_config_lateral_channels(const brn_config* cfg, int out[4]);
int brn_config_x4_channels(const brn_config* cfg);

/* ---- model lifecycle: BiRefNet::new (birefnet.rs:389-409) --------------------------------------------- */
/* Weights are looked up by the names of SURVEY.md App. A below `prefix` ("" for a full checkpoint).
 * Missing name -> BRN_ERR_MISSING_TENSOR naming it; extra names are ignored.  The loaded-but-unused heads
 * (gdt_convs_pred_*, conv_ms_spvn_*; birefnet.rs:150-166) must be present, as in the reference.
 * max_batch/max_h/max_w size the HBM workspace (forward with larger inputs re-plans it). */
brn_status brn_model_create(const brn_config* cfg, const brn_named_tensor* weights, size_t n_weights,
                            int device_ordinal, brn_dtype compute_dtype,
                            int max_batch, int max_h, int max_w, brn_model** out);
/* VarBuilder::from_mmaped_safetensors(&[path], DType::F32, &device) + BiRefNet::new(config, vb) in one call
 * (infer_image.rs:35-40).  The file is memory-mapped and parsed natively (8-byte LE header length, JSON index, raw LE
 * tensors); F32 is read in place, F16 / BF16 are widened to fp32 as candle's VarBuilder does for DType::F32.  Tensor names
 * are looked up below `prefix` ("" or NULL for a full checkpoint).  Errors as brn_model_create, plus BRN_ERR_INVALID_ARG for
 * an unreadable or malformed file. */
brn_status brn_model_create_from_safetensors(const brn_config* cfg, const char* path, const char* prefix,
                                            int device_ordinal, brn_dtype compute_dtype,
                                            int max_batch, int max_h, int max_w, brn_model** out);
void brn_model_destroy(brn_model* m);

/* BiRefNet::forward_logits (birefnet.rs:412-461): x [B,3,H,W] -> logits [B,1,H,W] (pre-sigmoid).
 * H and W must be multiples of 32 (the decoder's image2patches, birefnet.rs:288-300, needs it). */
brn_status brn_forward_logits(brn_model* m, const float* x_nchw, int B, int H, int W, brn_mem in_loc,
                              float* logits_out, brn_mem out_loc, void* stream);
/* BiRefNet::forward (birefnet.rs:466-469): sigmoid(forward_logits). */
brn_status brn_forward(brn_model* m, const float* x_nchw, int B, int H, int W, brn_mem in_loc,
                       float* mask_out, brn_mem out_loc, void* stream);

/* Pub fields used individually by bench_inference.rs:34,77,83:
 * model.backbone.forward(x) -> 4 NCHW feature maps (swin.rs:768-797).  outs[i] is [B, C_i, ceil(H/4)/2^i, ...]. */
brn_status brn_model_backbone_forward(brn_model* m, const float* x_nchw, int B, int H, int W, brn_mem in_loc,
                                      float* const outs[4], brn_mem out_loc, void* stream);
/* model.squeeze_module.forward(x4) (birefnet.rs:86-94): [B,5760,h,w] -> [B,3072,h,w]. */
brn_status brn_model_squeeze_forward(brn_model* m, const float* x4_nchw, int B, int h, int w, brn_mem in_loc,
                                     float* out, brn_mem out_loc, void* stream);
/* model.decoder.forward(x, x1, x2, x3, x4) (birefnet.rs:278-376). x [B,3,H,W]; x1..x4 at H/4..H/32 with
 * 384/768/1536/3072 channels. */
brn_status brn_model_decoder_forward(brn_model* m, const float* x_nchw, const float* x1, const float* x2,
                                     const float* x3, const float* x4, int B, int H, int W, brn_mem in_loc,
                                     float* logits_out, brn_mem out_loc, void* stream);

/* BiRefNetDecoder::new(config, vb) on its own (birefnet.rs:170-273): a handle that holds ONLY the decoder's weights (names relative
 * to `prefix`, e.g. "decoder."; the loaded-but-unused heads of birefnet.rs:229-243 must be present, as in the reference).  Such a handle
 * serves brn_model_decoder_forward (and brn_model_destroy); every entry that needs the backbone or the squeeze module answers
 * BRN_ERR_INVALID_ARG on it. */
brn_status brn_decoder_create(const brn_config* cfg, const brn_named_tensor* weights, size_t n_weights, const char* prefix,
                              int device_ordinal, brn_dtype compute_dtype, brn_model** out);

/* How a forward is spread over HIP streams (it never changes what is computed per image; results for a given setting are repeatable bit
 * for bit).  sub_batch_streams: a device-resident batch of B >= 4 images runs as that many sub-batches (>= 2 images each) on as many
 * streams, forked from and joined to the caller's stream, each with its own workspace (0 = library default: 2, or BRN_SPLIT_STREAMS;
 * 1 = everything on the caller's stream).  branch_stream_mask: independent branches of one forward on auxiliary streams — bits 0-2 the
 * ASPP branches, 3 the image-patch convolutions, 4 the lateral convolutions; -1 = automatic (on when the batch runs as one part), 0 = off. */
brn_status brn_model_set_streams(brn_model* m, int sub_batch_streams, int branch_stream_mask);

/* Per-stage wall time of the last forward on this handle, measured with HIP events on the call's stream:
 * [0]=backbone full, [1]=backbone half+fusion, [2]=squeeze, [3]=decoder, [4]=total (ms).  Mirrors the timers of
 * bench_inference.rs:37-92.  Enabled by brn_model_set_profiling(m, 1) (adds event records + one sync). */
brn_status brn_model_set_profiling(brn_model* m, int enable);
brn_status brn_model_last_timings(brn_model* m, float ms[5]);
/* Per-kernel-family accounting of the last profiled forward: for family f (see brn_kernel_family_name)
 * launches[f], ms[f] (HIP-event time around each launch, summed), flop[f] (2*M*N*K of the launches) and bytes[f] (their
 * algorithmic operand + result bytes).  After the families come coarse graph regions counted a second time under their own
 * rows ("region_aspp" = every launch of the ASPPDeformable modules, aspp.rs:303-333; "region_none" stays zero).
 * n is the array capacity; returns the number of rows (families + regions) through *n_out. */
brn_status brn_model_last_kernel_stats(brn_model* m, int n, int* launches, float* ms, double* flop,
                                       double* bytes, int* n_out);
const char* brn_kernel_family_name(int f);

/* ---- stand-alone SwinTransformer: SwinTransformer::new / forward (swin.rs:725-797) -------------------- */
/* Any SwinConfig (swin_t / swin_s / swin_b / swin_l: window 7 or 12) and any input size.  The handle is built for the arithmetic that
 * brn_set_op_compute selected on the creating thread (default BRN_F32): window 12 uses the mode's own attention kernel, window 7 the
 * fp32-MFMA attention kernel in every mode (on bf16 matrices in mode BRN_BF16). */
brn_status brn_swin_create(const brn_config* cfg /* swin_* fields */, const brn_named_tensor* weights,
                           size_t n_weights, const char* prefix, int device_ordinal, brn_swin** out);
void brn_swin_destroy(brn_swin* s);
brn_status brn_swin_forward(brn_swin* s, const float* x_nchw, int B, int H, int W, brn_mem in_loc,
                            float* const outs[4], brn_mem out_loc, void* stream);

/* ---- op-level entry points (the candle ops the reference calls; used by the parity tests) -------------- */
/* Selects the contraction arithmetic (brn_dtype) used by brn_linear_forward / brn_conv2d_forward on the calling thread
 * (default BRN_F32); models carry their own setting from brn_model_create. */
brn_status brn_set_op_compute(int dtype);
/* candle_nn::linear / linear_no_bias + optional gelu_erf + optional residual (swin.rs:98-107,130-131,406-407):
 * y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual[M,N]).  bias/residual may be NULL. */
brn_status brn_linear_forward(const float* x, int M, int K, const float* w, const float* bias, int N,
                              int act, const float* residual, float* y, brn_mem loc, int device_ordinal,
                              void* stream);
/* The attention half's tail of a Swin block as the reference chains it (swin.rs:310 proj, :406 shortcut + attn, :407 norm2 — and the
 * same shape at the end of the MLP, :106-107 + the next block's norm1): x_out = x W^T + bias + residual, y_out = LayerNorm(x_out) gamma +
 * beta (eps inside the sqrt, biased variance).  x [M,K], w [N,K], residual / x_out / y_out [M,N], all fp32 at this boundary.  In compute
 * mode bf16 (brn_set_op_compute) x is rounded to bf16 at the edge and the pair runs as ONE row-owning kernel where the shape allows (N = 192,
 * 384, 768: gemm_wstat_ln_bf16_kernel / gemm_rowln_bf16_kernel; y_out is then a bf16-rounded matrix), otherwise as a GEMM and a LayerNorm launch. */
brn_status brn_linear_residual_layer_norm_forward(const float* x, int M, int K, const float* w, const float* bias, int N,
                                                  const float* residual, const float* gamma, const float* beta, float eps,
                                                  float* x_out, float* y_out, brn_mem loc, int device_ordinal, void* stream);

/* candle_nn::layer_norm(dim, eps) forward (swin.rs:333,335,486,680,754): rows of length C. */
brn_status brn_layer_norm_forward(const float* x, int rows, int C, const float* gamma, const float* beta,
                                  float eps, float* y, brn_mem loc, int device_ordinal, void* stream);
/* candle_nn::conv2d / conv2d_no_bias forward (decoder.rs:44-45,104,113; aspp.rs:39-45; swin.rs:677), NCHW,
 * optional eval-mode batch_norm (decoder.rs:105,129: (x-mean)/sqrt(var+eps)*gamma+beta) and activation fused.
 * bn = {gamma,beta,running_mean,running_var} each [O] or all NULL. */
brn_status brn_conv2d_forward(const float* x, int B, int C, int H, int W, const float* w, const float* bias,
                              int O, int kh, int kw, int stride, int pad, int dil,
                              const float* bn_gamma, const float* bn_beta, const float* bn_mean,
                              const float* bn_var, float bn_eps, int act,
                              float* y, brn_mem loc, int device_ordinal, void* stream);
/* Tensor::upsample_bilinear2d(h, w, align_corners=true) (birefnet.rs:332,425,435-438,450-452), NCHW. */
brn_status brn_upsample_bilinear2d(const float* x, int B, int C, int H, int W, int out_h, int out_w,
                                   float* y, brn_mem loc, int device_ordinal, void* stream);
/* The attention half of SwinTransformerBlock::forward between norm1 and the residual (swin.rs:356-403):
 * pad -> roll(-shift) -> window_partition -> WindowAttention::forward (qkv, q*scale, q@k^T + rel-pos bias
 * (+ SW-MSA mask), softmax, @v, proj) -> window_reverse -> roll(+shift) -> crop.
 * x [B,H,W,C] is the norm1 output; y [B,H,W,C].  rel_table is relative_position_bias_table [(2ws-1)^2, heads].
 * head_dim must be 32; window_size 12 (Swin-B / L, swin.rs:55-80; every compute mode) or 7 (Swin-T / S, swin.rs:27-52); shift 0 or
 * window_size / 2. */
brn_status brn_window_attention_forward(const float* x, int B, int H, int W, int C, int heads, int window_size,
                                        int shift, const float* qkv_w, const float* qkv_b,
                                        const float* proj_w, const float* proj_b, const float* rel_table,
                                        float* y, brn_mem loc, int device_ordinal, void* stream);
/* PatchMerging::forward (swin.rs:491-527): x [B,H*W,C] -> [B, ceil(H/2)*ceil(W/2), 2C]. */
brn_status brn_patch_merging_forward(const float* x, int B, int H, int W, int C, const float* norm_g,
                                     const float* norm_b, const float* reduction_w, float* y, brn_mem loc,
                                     int device_ordinal, void* stream);
/* DeformableConv2d::forward (deform_conv.rs:82-99 / :101-215), NCHW.  mode = brn_deform_mode.
 * offset_w [2k^2,C,k,k]+offset_b, mod_w [k^2,C,k,k]+mod_b, w [O,C,k,k], bias [O] or NULL. */
brn_status brn_deform_conv2d_forward(const float* x, int B, int C, int H, int W,
                                     const float* offset_w, const float* offset_b,
                                     const float* mod_w, const float* mod_b,
                                     const float* w, const float* bias, int O, int k, int stride, int pad,
                                     int mode, float* y, brn_mem loc, int device_ordinal, void* stream);

/* ASPPDeformable::new(in_channels, out_channels, vb.pp(prefix)) + forward (aspp.rs:236-333; BasicDecBlk builds it with
 * (64, None), decoder.rs:107-111): five branches on the map (aspp1 and aspp_deforms.{0,1,2} = DeformConvASPP k 1,1,3,7 -> 256, BN, ReLU;
 * global average pool -> 1x1 -> BN -> ReLU -> broadcast), concat 1280, conv1 1x1 (no bias) + bn1 + ReLU.  weights: the module's tensors
 * under `prefix` ("aspp1.atrous_conv.offset_conv.weight", ..., "conv1.weight", "bn1.*"; SURVEY.md App. A <ASPP>); mode =
 * brn_deform_mode; out_channels 0 = None = in_channels (aspp.rs:242).  x [B,in_channels,H,W] -> y [B,out_channels,H,W], NCHW; any widths. */
brn_status brn_aspp_deformable_forward(const brn_named_tensor* weights, size_t n_weights, const char* prefix, int in_channels,
                                       int out_channels, int mode, const float* x, int B, int H, int W, float* y, brn_mem loc,
                                       int device_ordinal, void* stream);

/* BasicDecBlk::new(in_channels, out_channels, &DecoderConfig, vb.pp(prefix)) + forward (decoder.rs:78-141), the block behind
 * SqueezeModule and decoder_block{4,3,2,1}: conv_in 3x3 (in_channels -> 64, bias) + bn_in + ReLU -> ASPPDeformable(64) (use_aspp != 0:
 * DecoderConfig::use_aspp_deformable, decoder.rs:107-111; 0 = dec_att is None) -> conv_out 3x3 (64 -> out_channels, bias) + bn_out (no
 * ReLU).  inter_channels: 64 (DecoderConfig::default(), what BiRefNet builds) or in_channels / 4 (inter_channels_adaptive,
 * decoder.rs:94-98); 0 = 64.
 * weights: "conv_in.weight|bias", "bn_in.*", "dec_att.<ASPP>", "conv_out.weight|bias", "bn_out.*" under `prefix` (SURVEY.md App. A
 * <DecBlk>); mode = brn_deform_mode.  x [B,in_channels,H,W] -> y [B,out_channels,H,W], NCHW; any widths (maps are padded to the
 * kernels' channel granule inside). */
brn_status brn_decblk_forward(const brn_named_tensor* weights, size_t n_weights, const char* prefix, int in_channels,
                              int out_channels, int inter_channels, int use_aspp, int mode, const float* x, int B, int H, int W,
                              float* y, brn_mem loc, int device_ordinal, void* stream);

/* ---- image pre/post-processing: the steps either side of forward_logits in examples/infer_image.rs ------ */
/* infer_image.rs:44-67.  `img.resize_exact(S, S, FilterType::Triangle)` -> `to_rgb8()` -> (v/255 - mean) / std with the
 * ImageNet constants of :53-54 -> x [3,S,S] fp32 (NCHW, one image).  pixels: host, interleaved RGB8 or RGBA8 (channels 3|4;
 * alpha is resampled like the crate does and then dropped by to_rgb8), row-major [h][w][channels].  The resampler restates
 * image 0.25.9 (Cargo.lock:1102; imageops/sample.rs: vertical pass into f32, horizontal pass with clamp + round-to-nearest
 * into u8, per-output weight tables normalised by their sum) — that crate is not vendored in the reference tree. */
brn_status brn_preprocess_image(const unsigned char* pixels, int h, int w, int channels, int S,
                                float* x_nchw_out, brn_mem out_loc, int device_ordinal, void* stream);
/* infer_image.rs:84-110.  logits [S,S] -> sigmoid (apply_sigmoid != 0; pass 0 for brn_forward's output) ->
 * `(v * 255.0).clamp(0.0, 255.0) as u8` -> `imageops::resize(&mask, out_w, out_h, FilterType::Lanczos3)` -> mask
 * [out_h][out_w] u8 on the host. */
brn_status brn_postprocess_mask(const float* logits, int S, brn_mem in_loc, int apply_sigmoid, int out_h, int out_w,
                                unsigned char* mask_out, int device_ordinal, void* stream);

/* examples/infer_image.rs:44-110 for a BATCH, end to end on the device: n images (host, RGB8 / RGBA8, each its own size) ->
 * resize_exact(S, S, Triangle) + ImageNet normalisation -> forward() (sigmoid fused) of the whole batch -> `as u8` -> resize back to each
 * image's own size (Lanczos3) -> n masks on the host (masks[i]: heights[i] x widths[i] bytes).  One call: the uploads, 4 small kernels
 * per image, ONE forward of batch n, the downloads; no allocation after the first call of a shape (the staging buffers and the
 * resampling tables live with the model handle and are reused), one stream synchronisation at the end.  Results are bit-identical to
 * brn_preprocess_image -> brn_forward -> brn_postprocess_mask image by image when the forward runs the same batch. */
brn_status brn_infer_images_u8(brn_model* m, int n, const unsigned char* const* pixels, const int* heights, const int* widths,
                               int channels, int S, unsigned char* const* masks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIREFNET_HIP_H */
