// elementwise.hip — the HBM-bound data-movement kernels of the path (channels-last, float4 per lane).
// Each one names the reference tensor op(s) it stands for.
#include "../brn_kernels.h"

namespace brn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4_e __attribute__((ext_vector_type(4)));

// activation element access for both storage types: T = float (fp32 modes) or __bf16 (compute mode BRN_BF16); arithmetic is fp32
template <class T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
    const bf16x4_e h = *reinterpret_cast<const bf16x4_e*>(p);
    f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
}
template <class T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, f32x4 v) {
    bf16x4_e h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x4_e*>(p) = h;
}

typedef _Float16 f16x4_e __attribute__((ext_vector_type(4)));        // compute mode BRN_F16: the same kernels on fp16 maps (launcher flag bf16 == 2)
template <> __device__ __forceinline__ f32x4 ld4<_Float16>(const _Float16* p) {
    const f16x4_e h = *reinterpret_cast<const f16x4_e*>(p);
    f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    return r;
}
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, f32x4 v) {
    f16x4_e h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[e];
    *reinterpret_cast<f16x4_e*>(p) = h;
}

static inline dim3 grid1d(size_t n, int block) {
    size_t g = (n + block - 1) / block;
    if (g > 65535u * 32u) g = 65535u * 32u;   // kernels grid-stride
    if (g == 0) g = 1;
    return dim3((unsigned)g);
}

// PyTorch/candle align_corners=true source coordinate: src = dst * (in-1)/(out-1) (0 when out == 1)
__device__ __forceinline__ void ac_coord(int dst, int in, int out, int& i0, int& i1, float& l) {
    const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    const float src = scale * (float)dst;
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = src - (float)i0;
}

// Tensor::upsample_bilinear2d(h, w, true) on a channels-last window (birefnet.rs:332,347,362,435-438,450-452)
// ACC: y += resize(x) (the decoder's lateral sums, birefnet.rs:333,348,363, when the lateral conv wrote y first)
template <class T, bool ACC = false>
__global__ void resize_nhwc_kernel(const T* __restrict__ x, int B, int Hin, int Win, int C4, int ldx, int x_coff,
                                   T* __restrict__ y, int Hout, int Wout, int ldy, int y_coff) {
    const size_t total = (size_t)B * Hout * Wout * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t pix = idx / C4;
        const int ox = (int)(pix % Wout); pix /= Wout;
        const int oy = (int)(pix % Hout);
        const int b = (int)(pix / Hout);
        int y0, y1, x0, x1; float ly, lx;
        ac_coord(oy, Hin, Hout, y0, y1, ly);
        ac_coord(ox, Win, Wout, x0, x1, lx);
        const T* base = x + (size_t)b * Hin * Win * ldx + x_coff + c4 * 4;
        const f32x4 v00 = ld4<T>(base + ((size_t)y0 * Win + x0) * ldx);
        const f32x4 v01 = ld4<T>(base + ((size_t)y0 * Win + x1) * ldx);
        const f32x4 v10 = ld4<T>(base + ((size_t)y1 * Win + x0) * ldx);
        const f32x4 v11 = ld4<T>(base + ((size_t)y1 * Win + x1) * ldx);
        const f32x4 top = v00 + (v01 - v00) * lx;
        const f32x4 bot = v10 + (v11 - v10) * lx;
        f32x4 r = top + (bot - top) * ly;
        T* yp = y + (((size_t)b * Hout + oy) * Wout + ox) * ldy + y_coff + c4 * 4;
        if (ACC) r = ld4<T>(yp) + r;
        st4<T>(yp, r);
    }
}

// bf16 maps, 8 channels (16 bytes) per lane: same arithmetic per element as the 4-wide form, half the memory instructions
template <class E = __bf16>
__global__ void resize_nhwc_bf16x8_kernel(const E* __restrict__ x, int B, int Hin, int Win, int C8, int ldx, int x_coff,
                                          E* __restrict__ y, int Hout, int Wout, int ldy, int y_coff) {
    typedef E bf16x8_e __attribute__((ext_vector_type(8)));
    const size_t total = (size_t)B * Hout * Wout * C8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(idx % C8);
        size_t pix = idx / C8;
        const int ox = (int)(pix % Wout); pix /= Wout;
        const int oy = (int)(pix % Hout);
        const int b = (int)(pix / Hout);
        int y0, y1, x0, x1; float ly, lx;
        ac_coord(oy, Hin, Hout, y0, y1, ly);
        ac_coord(ox, Win, Wout, x0, x1, lx);
        const E* base = x + (size_t)b * Hin * Win * ldx + x_coff + c8 * 8;
        const bf16x8_e v00 = *reinterpret_cast<const bf16x8_e*>(base + ((size_t)y0 * Win + x0) * ldx);
        const bf16x8_e v01 = *reinterpret_cast<const bf16x8_e*>(base + ((size_t)y0 * Win + x1) * ldx);
        const bf16x8_e v10 = *reinterpret_cast<const bf16x8_e*>(base + ((size_t)y1 * Win + x0) * ldx);
        const bf16x8_e v11 = *reinterpret_cast<const bf16x8_e*>(base + ((size_t)y1 * Win + x1) * ldx);
        bf16x8_e r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a = (float)v00[e], bq = (float)v01[e], c = (float)v10[e], d = (float)v11[e];
            const float top = a + (bq - a) * lx, bot = c + (d - c) * lx;
            r[e] = (E)(top + (bot - top) * ly);
        }
        *reinterpret_cast<bf16x8_e*>(y + (((size_t)b * Hout + oy) * Wout + ox) * ldy + y_coff + c8 * 8) = r;
    }
}

hipError_t launch_resize_nhwc(const float* x, int B, int Hin, int Win, int C, int ldx, int x_coff,
                              float* y, int Hout, int Wout, int ldy, int y_coff, hipStream_t s, int bf16, int accumulate) {
    if (C % 4 || ldx % 4 || ldy % 4 || x_coff % 4 || y_coff % 4) return hipErrorInvalidValue;
    if (accumulate) {                 // fp32 maps only: a bf16 sum of two rounded terms is not what the fused epilogue computes
        if (bf16) return hipErrorInvalidValue;
        hipLaunchKernelGGL((resize_nhwc_kernel<float, true>), grid1d((size_t)B * Hout * Wout * (C / 4), 256), dim3(256), 0, s, x, B, Hin, Win, C / 4, ldx, x_coff,
                           y, Hout, Wout, ldy, y_coff);
        return hipGetLastError();
    }
    const size_t total = (size_t)B * Hout * Wout * (C / 4);
    if (bf16 == 2 && ((C | ldx | ldy | x_coff | y_coff) & 7) == 0)
        hipLaunchKernelGGL(resize_nhwc_bf16x8_kernel<_Float16>, grid1d(total / 2, 256), dim3(256), 0, s, reinterpret_cast<const _Float16*>(x), B, Hin, Win, C / 8,
                           ldx, x_coff, reinterpret_cast<_Float16*>(y), Hout, Wout, ldy, y_coff);
    else if (bf16 == 2) hipLaunchKernelGGL(resize_nhwc_kernel<_Float16>, grid1d(total, 256), dim3(256), 0, s, reinterpret_cast<const _Float16*>(x), B, Hin, Win, C / 4,
                                 ldx, x_coff, reinterpret_cast<_Float16*>(y), Hout, Wout, ldy, y_coff);
    else if (bf16 && ((C | ldx | ldy | x_coff | y_coff) & 7) == 0)
        hipLaunchKernelGGL(resize_nhwc_bf16x8_kernel<__bf16>, grid1d(total / 2, 256), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), B, Hin, Win, C / 8,
                           ldx, x_coff, reinterpret_cast<__bf16*>(y), Hout, Wout, ldy, y_coff);
    else if (bf16) hipLaunchKernelGGL(resize_nhwc_kernel<__bf16>, grid1d(total, 256), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), B, Hin, Win, C / 4,
                                 ldx, x_coff, reinterpret_cast<__bf16*>(y), Hout, Wout, ldy, y_coff);
    else hipLaunchKernelGGL(resize_nhwc_kernel<float>, grid1d(total, 256), dim3(256), 0, s, x, B, Hin, Win, C / 4, ldx, x_coff,
                            y, Hout, Wout, ldy, y_coff);
    return hipGetLastError();
}

// planar (NCHW) variant for the 3-channel image -> half scale (birefnet.rs:425)
__global__ void resize_nchw_kernel(const float* __restrict__ x, int BC, int Hin, int Win, float* __restrict__ y,
                                   int Hout, int Wout) {
    const size_t total = (size_t)BC * Hout * Wout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % Wout);
        size_t t = idx / Wout;
        const int oy = (int)(t % Hout);
        const size_t bc = t / Hout;
        int y0, y1, x0, x1; float ly, lx;
        ac_coord(oy, Hin, Hout, y0, y1, ly);
        ac_coord(ox, Win, Wout, x0, x1, lx);
        const float* base = x + bc * Hin * Win;
        const float v00 = base[(size_t)y0 * Win + x0], v01 = base[(size_t)y0 * Win + x1];
        const float v10 = base[(size_t)y1 * Win + x0], v11 = base[(size_t)y1 * Win + x1];
        const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
        y[idx] = top + (bot - top) * ly;
    }
}
hipError_t launch_resize_nchw(const float* x, int BC, int Hin, int Win, float* y, int Hout, int Wout, hipStream_t s) {
    const size_t total = (size_t)BC * Hout * Wout;
    hipLaunchKernelGGL(resize_nchw_kernel, grid1d(total, 256), dim3(256), 0, s, x, BC, Hin, Win, y, Hout, Wout);
    return hipGetLastError();
}

// NCHW <-> channels-last window, 32x32 tile transpose through LDS (coalesced on both sides)
template <class T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int C, int HW, T* __restrict__ y, int ldy, int y_coff) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, pq = p0 + tx;
        tile[i][tx] = (c < C && pq < HW) ? x[((size_t)b * C + c) * HW + pq] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int pq = p0 + i, c = c0 + tx;
        if (pq < HW && c < C) y[((size_t)b * HW + pq) * ldy + y_coff + c] = (T)tile[tx][i];
    }
}
hipError_t launch_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, int ldy, int y_coff, hipStream_t s, int bf16) {
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
    if (bf16 == 2) hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, grid, dim3(256), 0, s, x, C, HW, reinterpret_cast<_Float16*>(y), ldy, y_coff);
    else if (bf16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, grid, dim3(256), 0, s, x, C, HW, reinterpret_cast<__bf16*>(y), ldy, y_coff);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, x, C, HW, y, ldy, y_coff);
    return hipGetLastError();
}
template <class T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, int C, int HW, int ldx, int x_coff, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int pq = p0 + i, c = c0 + tx;
        tile[i][tx] = (c < C && pq < HW) ? (float)x[((size_t)b * HW + pq) * ldx + x_coff + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, pq = p0 + tx;
        if (c < C && pq < HW) y[((size_t)b * C + c) * HW + pq] = tile[tx][i];
    }
}
hipError_t launch_nhwc_to_nchw(const float* x, int B, int C, int H, int W, int ldx, int x_coff, float* y, hipStream_t s, int bf16) {
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
    if (bf16 == 2) hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, grid, dim3(256), 0, s, reinterpret_cast<const _Float16*>(x), C, HW, ldx, x_coff, y);
    else if (bf16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, grid, dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), C, HW, ldx, x_coff, y);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, x, C, HW, ldx, x_coff, y);
    return hipGetLastError();
}

// image2patches 'b c (hg h) (wg w) -> b (c hg wg) h w' (birefnet.rs:288-300), written channels-last:
// y[b][ty][tx][(c*gh + hg)*gw + wg] = x[b][c][hg*th + ty][wg*tw + tx]; channels [Cimg*gh*gw, cpad) are zeroed.
// One workgroup = (b, ty, 64 consecutive tx, 64 consecutive output channels): it reads, for each of its channels (c, hg, wg), 64
// consecutive floats of an image row (coalesced 256-byte segments), turns the [channel][tx] tile through LDS and writes 64-channel
// vectors per pixel (the one-thread-per-output-element form read 4 bytes per lane from addresses tw or th*W floats apart: 1 TB/s).
constexpr int I2P_T = 64;
template <class T>
__global__ void __launch_bounds__(256) image2patches_kernel(const float* __restrict__ x, int B, int Cimg, int H, int W, int th, int tw,
                                                            T* __restrict__ y, int ldy, int cpad) {
    __shared__ float tile[I2P_T][I2P_T + 1];             // [channel][tx]
    const int gh = H / th, gw = W / tw;
    const int cout = Cimg * gh * gw;
    const int txb = (tw + I2P_T - 1) / I2P_T, chb = (cpad + I2P_T - 1) / I2P_T;
    int blk = blockIdx.x;
    const int ch0 = (blk % chb) * I2P_T; blk /= chb;
    const int tx0 = (blk % txb) * I2P_T; blk /= txb;
    const int ty = blk % th;
    const int b = blk / th;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < I2P_T; i += 4) {
        const int ch = ch0 + i, tx = tx0 + lane;
        float v = 0.f;
        if (ch < cout && tx < tw) {
            const int wg = ch % gw, t = ch / gw;
            const int hg = t % gh, c = t / gh;
            v = x[(((size_t)b * Cimg + c) * H + hg * th + ty) * W + wg * tw + tx];
        }
        tile[i][lane] = v;
    }
    __syncthreads();
    for (int i = wv; i < I2P_T; i += 4) {                 // pixel tx0 + i, channel ch0 + lane
        const int tx = tx0 + i, ch = ch0 + lane;
        if (tx < tw && ch < cpad) y[(((size_t)b * th + ty) * tw + tx) * ldy + ch] = (T)tile[lane][i];
    }
}
hipError_t launch_image2patches(const float* x, int B, int Cimg, int H, int W, int th, int tw,
                                float* y, int ldy, int cpad, hipStream_t s, int bf16) {
    if (H % th || W % tw) return hipErrorInvalidValue;
    const size_t blocks = (size_t)B * th * ((tw + I2P_T - 1) / I2P_T) * ((cpad + I2P_T - 1) / I2P_T);
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
    if (bf16 == 2) hipLaunchKernelGGL(image2patches_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, x, B, Cimg, H, W, th, tw, reinterpret_cast<_Float16*>(y), ldy, cpad);
    else if (bf16) hipLaunchKernelGGL(image2patches_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, x, B, Cimg, H, W, th, tw, reinterpret_cast<__bf16*>(y), ldy, cpad);
    else hipLaunchKernelGGL(image2patches_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, B, Cimg, H, W, th, tw, y, ldy, cpad);
    return hipGetLastError();
}

// x.mean_keepdim(H).mean_keepdim(W) (aspp.rs:314) on a channels-last window.  Two deterministic passes (no float
// atomics: replicas on different GPUs must agree bit for bit): per-chunk partial sums, then a fixed-order final sum.
// Pass 1: a block of 256 threads owns GAP_CHUNK pixels x 64 channels — 16 channel quads (one ld4 per lane) x 16 pixel lanes, 8 loads per thread in
// a fixed order, then a fixed-order LDS sum over the 16 pixel lanes.  Pass 2: a block per 64 channels, 4 lanes of chunks each summing its chunks in
// order, then (s0 + s1) + (s2 + s3).  (Round 4: the round-1 form read one scalar per thread 128 times in a row and summed up to 512 partials in
// one thread: 33 + 11 us on a 16 MB map.)
constexpr int GAP_CHUNK = 128;   // pixels per block of pass 1
template <class T>
__global__ void __launch_bounds__(256) gap_partial_kernel(const T* __restrict__ x, int HW, int C, int ldx, int x_coff, float* __restrict__ part) {
    __shared__ f32x4 red[16][17];
    const int b = blockIdx.z, chunk = blockIdx.y, nchunks = gridDim.y;
    const int cq = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int p0 = chunk * GAP_CHUNK, p1 = min(HW, p0 + GAP_CHUNK);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < C)                                    // (C, ldx, x_coff are multiples of 4: launcher)
        for (int pq = p0 + sub; pq < p1; pq += 16) acc = acc + ld4<T>(x + ((size_t)b * HW + pq) * ldx + x_coff + c);
    red[sub][cq] = acc;
    __syncthreads();
    if (sub == 0 && c < C) {
        f32x4 t = red[0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) t = t + red[k][cq];
        *reinterpret_cast<f32x4*>(part + ((size_t)b * nchunks + chunk) * C + c) = t;
    }
}
__global__ void __launch_bounds__(256) gap_final_kernel(const float* __restrict__ part, int nchunks, int C, int HW, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, sub = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C)
        for (int k = sub; k < nchunks; k += 4) acc += part[((size_t)b * nchunks + k) * C + c];
    red[sub][cl] = acc;
    __syncthreads();
    if (sub == 0 && c < C) out[(size_t)b * C + c] = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) / (float)HW;
}
size_t gap_scratch_floats(int B, int HW, int C) { return (size_t)B * ((HW + GAP_CHUNK - 1) / GAP_CHUNK) * C; }
hipError_t launch_gap_nhwc(const float* x, int B, int HW, int C, int ldx, int x_coff, float* scratch, float* out, hipStream_t s, int bf16) {
    const int nchunks = (HW + GAP_CHUNK - 1) / GAP_CHUNK;
    if ((C | ldx | x_coff) & 3) return hipErrorInvalidValue;
    if (bf16 == 2) hipLaunchKernelGGL(gap_partial_kernel<_Float16>, dim3((C + 63) / 64, nchunks, B), dim3(256), 0, s, reinterpret_cast<const _Float16*>(x), HW, C, ldx, x_coff, scratch);
    else if (bf16) hipLaunchKernelGGL(gap_partial_kernel<__bf16>, dim3((C + 63) / 64, nchunks, B), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), HW, C, ldx, x_coff, scratch);
    else hipLaunchKernelGGL(gap_partial_kernel<float>, dim3((C + 63) / 64, nchunks, B), dim3(256), 0, s, x, HW, C, ldx, x_coff, scratch);
    hipLaunchKernelGGL(gap_final_kernel, dim3((C + 63) / 64, B), dim3(256), 0, s, scratch, nchunks, C, HW, out);
    return hipGetLastError();
}

// tiny dense layer on pooled vectors (global_avg_pool.1 + BN + ReLU, and its projection through conv1; aspp.rs:315-317,329)
__global__ void small_fc_kernel(const float* __restrict__ x, int Cin, const float* __restrict__ w, int ldw, int w_off, int N,
                                const float* __restrict__ scale, const float* __restrict__ shift, int act, float* __restrict__ y) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < Cin; k += 64) acc += x[(size_t)b * Cin + k] * w[(size_t)n * ldw + w_off + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        float v = acc;
        if (scale) v = v * scale[n] + shift[n];
        if (act == ACT_RELU) v = fmaxf(v, 0.f);
        y[(size_t)b * N + n] = v;
    }
}
hipError_t launch_small_fc(const float* x, int B, int Cin, const float* w, int ldw, int w_off, int N,
                           const float* scale, const float* shift, int act, float* y, hipStream_t s) {
    dim3 grid((N + 3) / 4, B);
    hipLaunchKernelGGL(small_fc_kernel, grid, dim3(256), 0, s, x, Cin, w, ldw, w_off, N, scale, shift, act, y);
    return hipGetLastError();
}

// GDT gate (birefnet.rs:327-329): attn = sigmoid(conv1x1_16->1(g)); p *= attn (broadcast over channels).
// One wave per pixel: lanes 0..15 read g, wave-reduce, then the whole wave scales the C channels.
template <class T>
__global__ void gdt_gate_kernel(T* __restrict__ p, size_t npix, int C4, int ldp, int p_coff, const T* __restrict__ g,
                                int ldg, const float* __restrict__ w, float bias) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t pix = wave; pix < npix; pix += nwaves) {
        float d = lane < 16 ? (float)g[pix * ldg + lane] * w[lane] : 0.f;
        d += __shfl_xor(d, 8); d += __shfl_xor(d, 4); d += __shfl_xor(d, 2); d += __shfl_xor(d, 1);
        d = __shfl(d, 0);
        const float a = 1.0f / (1.0f + expf(-(d + bias)));
        T* row = p + pix * ldp + p_coff;
        for (int c4 = lane; c4 < C4; c4 += 64) st4<T>(row + c4 * 4, ld4<T>(row + c4 * 4) * a);
    }
}
hipError_t launch_gdt_gate(float* p, int npix, int C, int ldp, int p_coff, const float* g, int ldg,
                           const float* w, float bias, hipStream_t s, int bf16) {
    if (C % 4) return hipErrorInvalidValue;
    size_t blocks = ((size_t)npix + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (bf16 == 2) hipLaunchKernelGGL(gdt_gate_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<_Float16*>(p), (size_t)npix, C / 4, ldp, p_coff,
                                 reinterpret_cast<const _Float16*>(g), ldg, w, bias);
    else if (bf16) hipLaunchKernelGGL(gdt_gate_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<__bf16*>(p), (size_t)npix, C / 4, ldp, p_coff,
                                 reinterpret_cast<const __bf16*>(g), ldg, w, bias);
    else hipLaunchKernelGGL(gdt_gate_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, p, (size_t)npix, C / 4, ldp, p_coff, g, ldg, w, bias);
    return hipGetLastError();
}

// per-pixel dot product over channels: y[pix] = <x[pix][0:C], w> + bias.  16 lanes per pixel.
template <class T>
__global__ void pixel_dot_kernel(const T* __restrict__ x, size_t npix, int C4, int ldx, int x_coff,
                                 const float* __restrict__ w, float bias, float* __restrict__ y) {
    const int sub = threadIdx.x & 15;
    const size_t grp = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const size_t ngrp = ((size_t)gridDim.x * blockDim.x) >> 4;
    for (size_t pix = grp; pix < npix; pix += ngrp) {
        const T* row = x + pix * ldx + x_coff;
        float acc = 0.f;
        for (int c4 = sub; c4 < C4; c4 += 16) {
            const f32x4 v = ld4<T>(row + c4 * 4);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c4 * 4);
            acc += (v[0] * ww[0] + v[1] * ww[1]) + (v[2] * ww[2] + v[3] * ww[3]);
        }
        acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 1);
        if (sub == 0) y[pix] = acc + bias;
    }
}
hipError_t launch_pixel_dot(const float* x, int npix, int C, int ldx, int x_coff, const float* w, float bias,
                            float* y, hipStream_t s, int bf16) {
    if (C % 4) return hipErrorInvalidValue;
    size_t blocks = ((size_t)npix * 16 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (bf16 == 2) hipLaunchKernelGGL(pixel_dot_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const _Float16*>(x), (size_t)npix, C / 4, ldx, x_coff, w, bias, y);
    else if (bf16) hipLaunchKernelGGL(pixel_dot_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x), (size_t)npix, C / 4, ldx, x_coff, w, bias, y);
    else hipLaunchKernelGGL(pixel_dot_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, x, (size_t)npix, C / 4, ldx, x_coff, w, bias, y);
    return hipGetLastError();
}

// Final head.  The reference upsamples p1 x4 (192 ch), concatenates ipt1 (48 ch) and applies conv_out1 1x1 240->1
// (birefnet.rs:372-375).  Both the bilinear upsample (weights sum to 1) and the 1x1 conv are linear, so
// conv(up(p1)) == up(conv(p1)): q = <p1, w[0:192]> is computed at 1/4 resolution and upsampled here, t = <ipt1, w[192:240]>.
__global__ void final_head_kernel(const float* __restrict__ q, int B, int h, int w, const float* __restrict__ t, float bias,
                                  int H, int W, int apply_sigmoid, float* __restrict__ out) {
    const size_t total = (size_t)B * H * W;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % W);
        size_t tt = idx / W;
        const int oy = (int)(tt % H);
        const size_t b = tt / H;
        int y0, y1, x0, x1; float ly, lx;
        ac_coord(oy, h, H, y0, y1, ly);
        ac_coord(ox, w, W, x0, x1, lx);
        const float* base = q + b * h * w;
        const float v00 = base[(size_t)y0 * w + x0], v01 = base[(size_t)y0 * w + x1];
        const float v10 = base[(size_t)y1 * w + x0], v11 = base[(size_t)y1 * w + x1];
        const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
        float v = (top + (bot - top) * ly) + t[idx] + bias;
        if (apply_sigmoid) v = 1.0f / (1.0f + expf(-v));
        out[idx] = v;
    }
}
hipError_t launch_final_head(const float* q, int B, int h, int w, const float* t, float bias, int H, int W,
                             int apply_sigmoid, float* out, hipStream_t s) {
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(final_head_kernel, grid1d(total, 256), dim3(256), 0, s, q, B, h, w, t, bias, H, W, apply_sigmoid, out);
    return hipGetLastError();
}

// fp32 -> bf16 (RNE) of a contiguous buffer: the op-level entry points in compute mode BRN_BF16 convert their operands at the edge
template <class E>
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, size_t n, E* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (E)x[i];
}
hipError_t launch_f32_to_bf16(const float* x, size_t n, float* y_bf16, hipStream_t s, int f16) {
    if (f16) hipLaunchKernelGGL(f32_to_bf16_kernel<_Float16>, grid1d(n, 256), dim3(256), 0, s, x, n, reinterpret_cast<_Float16*>(y_bf16));
    else hipLaunchKernelGGL(f32_to_bf16_kernel<__bf16>, grid1d(n, 256), dim3(256), 0, s, x, n, reinterpret_cast<__bf16*>(y_bf16));
    return hipGetLastError();
}

// bf16 -> fp32 of a contiguous buffer (op-level entry points that hand a bf16 result back across the fp32 boundary)
template <class E>
__global__ void bf16_to_f32_kernel(const E* __restrict__ x, size_t n, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
hipError_t launch_bf16_to_f32(const float* x_bf16, size_t n, float* y, hipStream_t s, int f16) {
    if (f16) hipLaunchKernelGGL(bf16_to_f32_kernel<_Float16>, grid1d(n, 256), dim3(256), 0, s, reinterpret_cast<const _Float16*>(x_bf16), n, y);
    else hipLaunchKernelGGL(bf16_to_f32_kernel<__bf16>, grid1d(n, 256), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x_bf16), n, y);
    return hipGetLastError();
}

__global__ void sigmoid_kernel(const float* __restrict__ x, size_t n, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = 1.0f / (1.0f + expf(-x[i]));
}
hipError_t launch_sigmoid(const float* x, size_t n, float* y, hipStream_t s) {
    hipLaunchKernelGGL(sigmoid_kernel, grid1d(n, 256), dim3(256), 0, s, x, n, y);
    return hipGetLastError();
}

// modulator = 2 / (1 + exp(-x))  (aspp.rs:173-174) applied in place to columns [c0, c1)
__global__ void mod_sigmoid2_kernel(float* __restrict__ x, size_t rows, int ld, int c0, int c1) {
    const int nc = c1 - c0;
    const size_t total = rows * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / nc; const int c = (int)(i - r * nc);
        float* ptr = x + r * ld + c0 + c;
        *ptr = (1.0f / (1.0f + expf(-*ptr))) * 2.0f;
    }
}
hipError_t launch_mod_sigmoid2(float* x, size_t rows, int ld, int c0, int c1, hipStream_t s) {
    hipLaunchKernelGGL(mod_sigmoid2_kernel, grid1d(rows * (size_t)(c1 - c0), 256), dim3(256), 0, s, x, rows, ld, c0, c1);
    return hipGetLastError();
}

// composed ipt_blk1 head (brn_weights.cpp): one thread per output pixel, 75 taps on the 3 image planes (zero outside), the
// kernel of the pixel's border case read through the scalar cache (uniform per wave except in the 1-pixel frame)
__global__ void __launch_bounds__(256) head_stencil5x5_kernel(const float* __restrict__ img, int B, int H, int W,
                                                              const float* __restrict__ k, const float* __restrict__ bias,
                                                              float* __restrict__ y) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= W || oy >= H) return;
    const int cs = (oy == 0 ? 0 : (oy == H - 1 ? 2 : 1)) * 3 + (ox == 0 ? 0 : (ox == W - 1 ? 2 : 1));
    const float* kk = k + cs * 75;
    const float* xb = img + (size_t)b * 3 * H * W;
    float acc = bias[cs];
    // all 75 loads go out unconditionally from clamped coordinates and a tap outside the image contributes a selected 0.0f
    // (fmaf(0, k, acc) == acc: the same value as skipping it).  Skipping by `continue` put every load behind a lane-dependent
    // branch; hipcc waits vmcnt(0) at each join, so a wave paid up to 25 memory round trips in a row (27 us per 1024^2 image).
#pragma unroll
    for (int fy = 0; fy < 5; ++fy) {
        const int iy = oy + fy - 2;
        const bool vy = (unsigned)iy < (unsigned)H;
        const int iyc = min(max(iy, 0), H - 1);
#pragma unroll
        for (int fx = 0; fx < 5; ++fx) {
            const int ix = ox + fx - 2;
            const bool ok = vy && (unsigned)ix < (unsigned)W;
            const size_t o = (size_t)iyc * W + min(max(ix, 0), W - 1);
            const float* kf = kk + (fy * 5 + fx) * 3;
            const float x0 = xb[o], x1 = xb[(size_t)H * W + o], x2 = xb[2 * (size_t)H * W + o];
            acc = fmaf(ok ? x0 : 0.0f, kf[0], acc);
            acc = fmaf(ok ? x1 : 0.0f, kf[1], acc);
            acc = fmaf(ok ? x2 : 0.0f, kf[2], acc);
        }
    }
    y[((size_t)b * H + oy) * W + ox] = acc;
}
hipError_t launch_head_stencil5x5(const float* img, int B, int H, int W, const float* k, const float* bias, float* y, hipStream_t s) {
    if (B <= 0 || H < 2 || W < 2) return hipErrorInvalidValue;
    dim3 grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)B), block(256);
    hipLaunchKernelGGL(head_stencil5x5_kernel, grid, block, 0, s, img, B, H, W, k, bias, y);
    return hipGetLastError();
}

}  // namespace brn
