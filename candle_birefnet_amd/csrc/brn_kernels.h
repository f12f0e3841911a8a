// brn_kernels.h — internal launch interface of the gfx950 kernels (not part of the C ABI).
// All activations are fp32, channels-last: a "map" is [B, H, W, ld] with the logical channels living in a
// column window [coff, coff+C) of the ld-wide rows, so concatenations are written in place (no cat copies).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace brn {

enum GemmMode {
    GEMM_DENSE = 0,       // A[m][k] = A + m*lda + k                         (Linear layers, 1x1 convs)
    GEMM_CONV_NHWC = 1,   // implicit im2col over a channels-last map, K = (ky,kx,ci), Cin % 32 == 0
    GEMM_GATHER_NCHW = 2, // implicit im2col over an NCHW image, K = (ci,ky,kx) (candle weight order), any Cin
    GEMM_DEFORM_NHWC = 3  // modulated deformable im2col (bilinear gather * mask), K = (ky,kx,ci), Cin % 32 == 0
};
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU_ERF = 2 };

struct GemmParams {
    const float* A;
    const float* W;       // [Npad][K] row-major, K contiguous, rows >= roundup(N, 128), K % 32 == 0, zero padded
    float* C;
    int M, N, K;
    int mode;
    int lda;              // dense: row stride; conv/deform: floats per pixel of the input map
    int a_coff;           // conv/deform: first channel of the input window
    // conv geometry (modes 1..3)
    int Hin, Win, Cin, kh, kw, stride, pad, dil, Hout, Wout;
    int Kreal;            // mode 2: Cin*kh*kw before padding to 32
    // deformable (mode 3): om = [B,Hout,Wout,om_ld] holding 2*kh*kw offsets (dy,dx interleaved, torchvision
    // order) then kh*kw modulator logits (already 2*sigmoid applied) starting at column om_mask_off
    const float* om; int om_ld; int om_mask_off;
    int om_sigmoid;       // deform_bf16 kernel only: the modulator columns hold raw logits, 2 * sigmoid (aspp.rs:173-174) is applied in the gather
    // epilogue:  v = acc (+ bias[n]) (+ bbias[m / bbias_rows][n]);  v = v*scale[n] + shift[n];  v = act(v);
    //            v += R[m*ldr + r_coff + n];  C[m*ldc + c_coff + n] = v
    const float* bias;
    const float* bbias; int bbias_rows;
    const float* scale; const float* shift;
    int act;
    const float* R; int ldr; int r_coff;
    int ldc; int c_coff;
    // split-bf16 path (planes = 0: fp32 MFMA path): Wp = W pre-split into `planes` bf16 planes [planes][wp_rows][K]
    const void* Wp; int planes; int wp_rows;
    // split-K (filled by launch_gemm from the plan): slices write raw partials to part[slice][M][N]
    int splitk; float* part;
    int a_planes, c_planes;   // 2: A is read / C is written in the P2 layout (kernels/split_planes.h); split-bf16 dense ws kernel only
    // mode f32_half2 (planes == 2 with fp16 planes, kernels/split_planes.h: split4h): Wp holds the fp16 planes of w_scale * W (a power of two per
    // tensor, GemmW::w_scale), A is split as the fp16 planes of a_scale * A while it is staged, and the accumulators are multiplied by
    // out_scale = 1 / (a_scale * w_scale) — exact — before the epilogue
    int h2; float a_scale, out_scale;
    // bf16-storage mode (kernels/gemm_bf16.hip): A, C, R are bf16 unless flagged; Wp is the plain [wp_rows][wp_ld] bf16 matrix
    int wp_ld;                // elements per W row (K rounded up to 64, zero padded)
    int k_chunk_major;        // bf16 implicit GEMM: W's K order is (64-channel chunk, tap, channel in chunk) instead of (tap, channel); Cin % 64 == 0
    int c_f32, r_f32;         // 1: C is written / R is read as fp32 (offset maps of the deformable mode, fp32 side outputs)
    int a_bf16, c_bf16;       // gemm_f32_kernel only (deformable gather in bf16 mode): A map read / C written as bf16
    unsigned long long* trace;   // diagnostics only: per-workgroup cycle stamps (gemm_split_ws_kernel), null in production
    int abl;   // diagnostics only (brn_gemm_microbench): 1 = no global loads in the K loop, 2 = no LDS staging, 4 = no fragment reads / MFMA
};

struct GemmPlan { int cfg; int splitk; size_t ws_floats; };   // cfg: 0 = 128x128, 1 = 128x64, 2 = 64x64 block tile
GemmPlan plan_gemm(int M, int N, int K, int planes = 0);   // planes: bf16 planes of the split modes (0 = fp32 kernels)
// ws: plan.ws_floats floats of scratch when plan.splitk > 1.  Returns the hipError_t of the launch(es).
hipError_t launch_gemm(const GemmParams& p, const GemmPlan& plan, float* ws, hipStream_t s);

// CUs the persistent grids (gemm_bf16_kernel, gemm_wstat_*) are sized for on the calling host thread: 256 unless the launch stream carries a CU mask
int launch_cus();
void set_launch_cus(int n);
// bf16-storage mode: plan + launch (kernels/gemm_bf16.hip).  cfg: 0 = 128x128, 1 = 128x64, 2 = 256x256, 3 = 256x192 block tile
GemmPlan plan_gemm_bf16(int M, int N, int K, bool f32_residual = false, bool gelu = false);   // f32_residual: fp32 C with an fp32 residual (proj / fc2)
hipError_t launch_gemm_bf16(const GemmParams& p, const GemmPlan& plan, float* ws, hipStream_t s);

// bf16-storage mode, short-K dense GEMM with the weights resident in registers (kernels/gemm_bf16.hip, gemm_wstat_bf16_kernel): Wp = the weights in
// MFMA fragment order (GemmW::wf, attach_dense_frags), K = 192 (64-row A tiles) or 384 (32-row tiles), N % 192 == 0, bf16 out, bias + activation only
bool gemm_wstat_eligible(const GemmParams& p);
hipError_t launch_gemm_wstat(const GemmParams& p, hipStream_t s);
// the same with N = K = 192, fp32 C + fp32 residual (in place or not) AND y = LayerNorm(C) gamma + beta written as a bf16 matrix
// (gemm_wstat_ln_bf16_kernel: the attention projection of a C = 192 stage with the block's norm2 in its epilogue)
// PatchEmbed for the Swin-L geometry as one kernel (kernels/patch_embed.hip): 4 x 4 stride-4 conv of the NCHW image (3 -> 192) + bias +
// LayerNorm, fp32, written into the residual stream x [B * H/4 * W/4][ldx]
bool patch_embed_ln_eligible(int Cin, int N, int k, int stride, int H, int W, int ldw, int ldx);
hipError_t launch_patch_embed_ln(const float* img, int B, int H, int W, const float* wgt, int ldw, const float* bias, const float* gamma,
                                 const float* beta, float eps, float* x, int ldx, hipStream_t s, const float* gamma1 = nullptr,
                                 const float* beta1 = nullptr, void* xn_bf16 = nullptr, int ldxn = 0, int xn_f16 = 0);   // (gamma1 / beta1 / xn: also LayerNorm(x) gamma1 + beta1 as a bf16 matrix)
bool gemm_wstat_ln_eligible(const GemmParams& p);
// the same fusion for N = 768 / 384 (any K % 32 == 0): gemm_rowln_bf16_kernel, a workgroup owns 64 whole rows, Wp = the plain [wp_rows][wp_ld] bf16 matrix
bool gemm_rowln_eligible(const GemmParams& p);
hipError_t launch_gemm_rowln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s);
hipError_t launch_gemm_wstat_ln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s);

// bf16-storage mode, modulated deformable conv (kernels/deform_bf16.hip): A = bf16 channels-last map, om = fp32 offsets | modulator,
// Wp = the weights in MFMA fragment order (GemmW::wf), C = bf16 window.  eligible(): shapes the kernel covers (else gemm_f32_kernel)
bool deform_bf16_eligible(const GemmParams& p);
hipError_t launch_deform_bf16(const GemmParams& p, hipStream_t s);

// split modes with A already in the P layout (kernels/gemm_planes.hip): LDS-DMA staged, persistent, no splitting wave
bool gemm_planes_eligible(const GemmParams& p);
GemmPlan plan_gemm_planes(int M, int N, int K, int planes, bool c_planes);
hipError_t launch_gemm_planes(const GemmParams& p, const GemmPlan& plan, float* ws, hipStream_t s);

// compute mode BRN_F16: the same kernels with fp16 as the 16-bit storage / MFMA operand type (kernels/gemm_bf16.hip and kernels/deform_bf16.hip
// compiled with -DBRN_S16_F16=1).  Same contracts as the functions of the same name above.
namespace hf {
GemmPlan plan_gemm_bf16(int M, int N, int K, bool f32_residual = false, bool gelu = false);
hipError_t launch_gemm_bf16(const GemmParams& p, const GemmPlan& plan, float* ws, hipStream_t s);
bool gemm_wstat_eligible(const GemmParams& p);
hipError_t launch_gemm_wstat(const GemmParams& p, hipStream_t s);
bool gemm_wstat_ln_eligible(const GemmParams& p);
hipError_t launch_gemm_wstat_ln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s);
bool gemm_rowln_eligible(const GemmParams& p);
hipError_t launch_gemm_rowln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s);
bool deform_bf16_eligible(const GemmParams& p);
hipError_t launch_deform_bf16(const GemmParams& p, hipStream_t s);
}  // namespace hf

#ifdef BRN_DIAG_BUILD
hipError_t launch_mfma_valu_probe(int blocks, int iters, int mode, float* sink, hipStream_t s);
hipError_t launch_lds_mfma_probe(int blocks, int iters, int np, int variant, float* sink, hipStream_t s);
hipError_t launch_mfma_peak_bf16(int blocks, int iters, float* sink, unsigned long long* clk, int nacc, hipStream_t s);
hipError_t launch_mfma_peak(int blocks, int iters, float* sink, unsigned long long* clk, hipStream_t s);
#endif

struct LayerNormParams {
    const float* x; float* y;
    int rows, C;
    const float* gamma; const float* beta; float eps;
    int ldx;              // input row stride (mode 0)
    int ldy, y_coff;      // output row stride / column offset
    // mode 1: PatchMerging gather (swin.rs:505-522): row (b,i,j) of the merged map gathers the four tokens
    // (2i,2j),(2i+1,2j),(2i,2j+1),(2i+1,2j+1) of x [B,H,W,Cin], zero outside (odd H/W padding), C == 4*Cin
    int mode; int H, W, Cin;
    int y_planes;         // 2: write y in the P2 layout (kernels/split_planes.h) for a split-bf16 GEMM; 0: fp32
    float y_h2;           // > 0 (with y_planes == 2): the planes are the fp16 planes of y_h2 * y (mode f32_half2)
    int y_bf16;           // 1: y is a bf16 matrix (compute mode BRN_BF16; x stays fp32: the residual stream)
};
hipError_t launch_layernorm(const LayerNormParams& p, hipStream_t s);

struct WindowAttnParams {
    const float* qkv;     // [B, H, W, 3C] natural token order (q | k | v, heads-major inside each, swin.rs:218)
    const float* qkv_bias;// [3C]  (q/k/v of a zero pad token)
    const float* rel_table;// relative_position_bias_table (swin.rs:138-141) transposed to [heads][(2*12-1)^2]; cached_bias (swin.rs:147-152) is never built
    float* out;           // [B, H, W, C]
    int B, H, W, C, heads;
    int Hp, Wp;           // padded canvas (multiples of 12)
    int shift;            // 0 or 6
    float scale;          // head_dim^-0.5
    int planes;           // 0: fp32 MFMA kernel (modes f32, f32_split3); 2 / 1: bf16-split kernel (f32_split2 / bf16_operands)
    int out_planes;       // 2: write `out` in the P2 layout (kernels/split_planes.h) for the proj GEMM; 0: fp32
    int h2;               // 1 (with planes == 2): the split kernel on fp16 planes (mode f32_half2)
    float out_h2;         // > 0 (with out_planes == 2, planes == 0 or h2): fp16 planes of out_h2 * out (mode f32_half2, written by the fp32-MFMA kernel)
    int ws;               // window side: 12 (0 = 12) or 7 (Swin-T / S: fp32 kernel only)
    int io_bf16;          // 1: qkv and out are bf16 matrices (compute mode BRN_BF16; qkv_bias / rel_table stay fp32)
};
hipError_t launch_window_attention(const WindowAttnParams& p, hipStream_t s);
// two maps of the same stage in one launch (p2 may be null)
hipError_t launch_window_attention2(const WindowAttnParams& p, const WindowAttnParams* p2, hipStream_t s);

// ---- data movement / elementwise (HBM-bound) ------------------------------------------------------------
// bilinear, align_corners=true, channels-last window -> window; optional accumulate (y += )
hipError_t launch_resize_nhwc(const float* x, int B, int Hin, int Win, int C, int ldx, int x_coff,
                              float* y, int Hout, int Wout, int ldy, int y_coff, hipStream_t s, int bf16 = 0, int accumulate = 0);
// NCHW planar bilinear (the 3-channel image -> half scale), align_corners=true
hipError_t launch_resize_nchw(const float* x, int BC, int Hin, int Win, float* y, int Hout, int Wout, hipStream_t s);
// NCHW -> NHWC window and back
// (bf16 = 1: the channels-last side is a bf16 map, ld / coff in elements; the NCHW side is always fp32)
hipError_t launch_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, int ldy, int y_coff, hipStream_t s, int bf16 = 0);
hipError_t launch_nhwc_to_nchw(const float* x, int B, int C, int H, int W, int ldx, int x_coff, float* y, hipStream_t s, int bf16 = 0);
// image2patches (birefnet.rs:288-300): x NCHW [B,Cimg,H,W] -> y[b, th, tw, (c,gh,gw)] channels-last, ld = ldy
hipError_t launch_image2patches(const float* x, int B, int Cimg, int H, int W, int th, int tw,
                                float* y, int ldy, int cpad, hipStream_t s, int bf16 = 0);
// per-(b,c) mean over H*W of a channels-last window: out[b][c]   (aspp.rs:314)
size_t gap_scratch_floats(int B, int HW, int C);
hipError_t launch_gap_nhwc(const float* x, int B, int HW, int C, int ldx, int x_coff, float* scratch, float* out, hipStream_t s, int bf16 = 0);
// tiny dense layers on [B,Cin] vectors: y[b][n] = act((sum_k x[b][k] w[n*ldw + w_off + k]) * scale[n] + shift[n])
hipError_t launch_small_fc(const float* x, int B, int Cin, const float* w, int ldw, int w_off, int N,
                           const float* scale, const float* shift, int act, float* y, hipStream_t s);
// GDT gate (birefnet.rs:327-329): a = sigmoid(dot(g[pix][0:16], w) + b); p[pix][0:C] *= a
hipError_t launch_gdt_gate(float* p, int npix, int C, int ldp, int p_coff, const float* g, int ldg,
                           const float* w, float bias, hipStream_t s, int bf16 = 0);
// per-pixel dot: y[pix] = dot(x[pix][0:C], w) (+ bias)
hipError_t launch_pixel_dot(const float* x, int npix, int C, int ldx, int x_coff, const float* w, float bias,
                            float* y, hipStream_t s, int bf16 = 0);
// contiguous fp32 -> bf16 (round to nearest even)
hipError_t launch_f32_to_bf16(const float* x, size_t n, float* y_bf16, hipStream_t s, int f16 = 0);   // f16: fp16 instead (compute mode BRN_F16)
hipError_t launch_bf16_to_f32(const float* x_bf16, size_t n, float* y, hipStream_t s, int f16 = 0);
// final head (birefnet.rs:372-375 with conv_out1 commuted through the bilinear upsample):
// out[b][oy][ox] = bilinear(q [B,h,w] -> H,W) + t[b][oy][ox] (+bias); optional sigmoid
hipError_t launch_final_head(const float* q, int B, int h, int w, const float* t, float bias, int H, int W,
                             int apply_sigmoid, float* out, hipStream_t s);
// t = stencil5x5(x): the composed ipt_blk1 head on the NCHW image [B,3,H,W]; k = [9 border cases][5][5][3], bias = [9]
hipError_t launch_head_stencil5x5(const float* img, int B, int H, int W, const float* k, const float* bias, float* y, hipStream_t s);
// ---- image pre/post-processing (kernels/imageproc.hip; infer_image.rs:44-67,84-110) ----
hipError_t launch_resample_v_u8(const unsigned char* in, int h, int w, int C, int nh, const int* left, const int* count,
                                const float* wts, int max_taps, float* out, hipStream_t s);
// out_f32 != null: channels 0..2 as ((v/255) - mean[c]) / std[c] into NCHW [3][nh][nw]; else u8 [nh][nw][C]
hipError_t launch_resample_h(const float* in, int nh, int w, int C, int nw, const int* left, const int* count, const float* wts,
                             int max_taps, unsigned char* out_u8, float* out_f32, const float* mean, const float* stdv, hipStream_t s);
hipError_t launch_mask_u8(const float* logits, long n, int apply_sigmoid, unsigned char* out, hipStream_t s);
// y = sigmoid(x)
hipError_t launch_sigmoid(const float* x, size_t n, float* y, hipStream_t s);
// modulator epilogue for deformable mode: columns [c0,c1) of rows get 2/(1+exp(-x))   (aspp.rs:173-174)
hipError_t launch_mod_sigmoid2(float* x, size_t rows, int ld, int c0, int c1, hipStream_t s);

} // namespace brn
