// imageproc.hip — the steps either side of forward_logits in the reference's inference example (infer_image.rs:44-67,84-110):
//   pre : image 0.25.9 `DynamicImage::resize_exact(S, S, Triangle)` -> to_rgb8 -> (v/255 - mean) / std -> [3,S,S] fp32
//   post: sigmoid -> (v*255).clamp(0,255) as u8 -> image 0.25.9 `imageops::resize(.., w, h, Lanczos3)` -> u8 mask
// `image` is a crates.io dependency that is not vendored under /root/reference; its resampler (src/imageops/sample.rs:
// vertical_sample into an f32 image, then horizontal_sample with clamp + round-to-nearest into the pixel type; the
// per-output-pixel weight table left/right/ws normalised by its sum) is restated here from the published algorithm.  The
// weight tables are built on the host exactly as the crate builds them; these kernels only apply them, in the crate's
// accumulation order and WITHOUT fused multiply-add (Rust does not contract `t += p * w`).  HBM-bound byte work.
#include "../brn_kernels.h"

namespace brn {

// hipcc contracts `t + p * w` into an fma by default (-ffp-contract=fast), and HIP's __fmul_rn / __fadd_rn are plain operators:
// the crate's arithmetic is a separate multiply and add, so this file is built with -ffp-contract=off (Makefile).

// vertical pass: in u8 [h][w][C] -> out f32 [nh][w][C];  left/count/wts index the OUTPUT row
__global__ void __launch_bounds__(256) resample_v_u8_kernel(const unsigned char* __restrict__ in, int h, int w, int C, int nh,
                                                            const int* __restrict__ left, const int* __restrict__ count,
                                                            const float* __restrict__ wts, int max_taps, float* __restrict__ out) {
    const long n = (long)nh * w * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / ((long)w * C));
        const long xc = i - (long)oy * w * C;
        const int l = left[oy], k = count[oy];
        const float* ww = wts + (long)oy * max_taps;
        float t = 0.f;
        for (int j = 0; j < k; ++j) t = __fadd_rn(t, __fmul_rn((float)in[(long)(l + j) * w * C + xc], ww[j]));
        out[i] = t;
    }
}

// horizontal pass: in f32 [nh][w][C] -> u8 value v = round(clamp(t, 0, 255)) (FloatNearest: half away from zero).
// MODE 0: store u8 [nh][nw][C].   MODE 1 (preprocess): channels 0..2 -> ((v / 255) - mean[c]) / std[c] into NCHW fp32 [3][nh][nw]
template <int MODE>
__global__ void __launch_bounds__(256) resample_h_kernel(const float* __restrict__ in, int nh, int w, int C, int nw,
                                                         const int* __restrict__ left, const int* __restrict__ count,
                                                         const float* __restrict__ wts, int max_taps, unsigned char* __restrict__ out_u8,
                                                         float* __restrict__ out_f32, float m0, float m1, float m2, float s0, float s1,
                                                         float s2) {
    const long n = (long)nh * nw * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int ox = (int)(p % nw);
        const int y = (int)(p / nw);
        const int l = left[ox], k = count[ox];
        const float* ww = wts + (long)ox * max_taps;
        float t = 0.f;
        for (int j = 0; j < k; ++j) t = __fadd_rn(t, __fmul_rn(in[((long)y * w + l + j) * C + c], ww[j]));
        const float v = roundf(fminf(fmaxf(t, 0.f), 255.f));
        if (MODE == 0) {
            out_u8[i] = (unsigned char)v;
        } else if (c < 3) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            out_f32[((long)c * nh + y) * nw + ox] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.0f), mean), sd);
        }
    }
}

// logits -> sigmoid -> (v * 255).clamp(0, 255) as u8   (infer_image.rs:84-99; `as u8` truncates)
__global__ void __launch_bounds__(256) mask_u8_kernel(const float* __restrict__ logits, long n, int apply_sigmoid,
                                                      unsigned char* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = logits[i];
        if (apply_sigmoid) v = __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-v)));
        v = fminf(fmaxf(__fmul_rn(v, 255.0f), 0.f), 255.f);
        out[i] = (unsigned char)v;
    }
}

static unsigned grid_for(long n) {
    long b = (n + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

hipError_t launch_resample_v_u8(const unsigned char* in, int h, int w, int C, int nh, const int* left, const int* count,
                                const float* wts, int max_taps, float* out, hipStream_t s) {
    if (h < 1 || w < 1 || C < 1 || nh < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(resample_v_u8_kernel, dim3(grid_for((long)nh * w * C)), dim3(256), 0, s, in, h, w, C, nh, left, count, wts, max_taps, out);
    return hipGetLastError();
}
hipError_t launch_resample_h(const float* in, int nh, int w, int C, int nw, const int* left, const int* count, const float* wts,
                             int max_taps, unsigned char* out_u8, float* out_f32, const float* mean, const float* stdv, hipStream_t s) {
    if (nh < 1 || w < 1 || C < 1 || nw < 1 || (!out_u8 && !out_f32)) return hipErrorInvalidValue;
    const dim3 g(grid_for((long)nh * nw * C)), b(256);
    if (out_f32) hipLaunchKernelGGL(resample_h_kernel<1>, g, b, 0, s, in, nh, w, C, nw, left, count, wts, max_taps, out_u8, out_f32,
                                    mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    else hipLaunchKernelGGL(resample_h_kernel<0>, g, b, 0, s, in, nh, w, C, nw, left, count, wts, max_taps, out_u8, out_f32,
                            0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
    return hipGetLastError();
}
hipError_t launch_mask_u8(const float* logits, long n, int apply_sigmoid, unsigned char* out, hipStream_t s) {
    if (n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mask_u8_kernel, dim3(grid_for(n)), dim3(256), 0, s, logits, n, apply_sigmoid, out);
    return hipGetLastError();
}

}  // namespace brn
