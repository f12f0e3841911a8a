// patch_embed.hip — Swin PatchEmbed (swin.rs:692-714) as ONE kernel for the Swin-L geometry: the 4 x 4 stride-4 convolution of the
// NCHW image (3 -> 192 channels, K = 48), its bias, and the LayerNorm that follows, written straight into the fp32 residual
// stream.  A workgroup owns whole rows (4 waves x 48 output channels), so the row statistics are local and neither the conv output
// nor a second pass over it touches HBM: image in (12 B per pixel), x out (768 B per token).  fp32 operands on
// v_mfma_f32_16x16x4_f32 (the flops are nothing: 18 KFLOP per token), fp32 accumulation, the arithmetic of layernorm_kernel
// (two-pass mean / biased variance) on the row; optionally the first block's norm1 of that row too (bf16, the qkv operand).
// Used in compute mode BRN_BF16, where the two launches it replaces (gather GEMM +
// LayerNorm) cost 0.5 ms per 8-image step; the fp32 modes keep their own kernels.
#include "../brn_kernels.h"

namespace brn {

typedef float f32x4_p __attribute__((ext_vector_type(4)));

constexpr int PE_N = 192, PE_K = 48, PE_TM = 64, PE_LDA = 52;            // channels, 3 x 4 x 4 taps, tokens per tile, LDS row stride (floats)

__global__ void __launch_bounds__(256, 2) patch_embed_ln_kernel(const float* __restrict__ img, const int B, const int H, const int W,
                                                                const float* __restrict__ wgt, const int ldw, const float* __restrict__ bias,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, const float eps,
                                                                float* __restrict__ x, const int ldx, const int img_aligned16,
                                                                const float* __restrict__ gamma1, const float* __restrict__ beta1, void* __restrict__ xn, const int ldxn, const int xn_f16) {
    __shared__ __attribute__((aligned(16))) float As[PE_TM * PE_LDA];    // [token][k = (c, ky, kx)]
    __shared__ __attribute__((aligned(16))) float Cs[32 * PE_N];         // a 32-row half of the C tile, 16-byte chunks XOR-ed with the row
    __shared__ __attribute__((aligned(16))) float gb_s[4 * PE_N];        // gamma | beta of PatchEmbed's norm, then of the first block's norm1 (optional)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int Hp = H >> 2, Wp = W >> 2;
    const long M = (long)B * Hp * Wp;
    const int T = (int)((M + PE_TM - 1) / PE_TM);
    if (tid < PE_N) {
        gb_s[tid] = gamma[tid]; gb_s[PE_N + tid] = beta[tid];
        if (xn) { gb_s[2 * PE_N + tid] = gamma1[tid]; gb_s[3 * PE_N + tid] = beta1[tid]; }
    }
    // this wave's 48 columns of W as MFMA A operands: lane (n = 16 j + li, k = 4 ks + g)
    float wfr[3][12];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) wfr[j][ks] = wgt[(long)(wave * 48 + 16 * j + li) * ldw + 4 * ks + g];
    f32x4_p bv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) bv[j] = bias ? *reinterpret_cast<const f32x4_p*>(bias + wave * 48 + 16 * j + 4 * g) : f32x4_p{0.f, 0.f, 0.f, 0.f};
    const int er = tid >> 3, ec = tid & 7;                               // read-back: row of the 32-row half, first chunk (then + 8 k)

    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        // ---- gather: lane = token of the tile, wave w takes the (c, ky) rows q = w, w + 4, w + 8: 64 x 16 B contiguous per instruction ----
        {
            const long m = min((long)t * PE_TM + lane, M - 1);
            const int b = (int)(m / ((long)Hp * Wp));
            const int rem = (int)(m - (long)b * Hp * Wp);
            const int py = rem / Wp, px = rem - py * Wp;
            f32x4_p v[3];
            const float* src[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int q = wave + 4 * r, c = q >> 2, ky = q & 3;
                src[r] = img + (((long)b * 3 + c) * H + (py * 4 + ky)) * W + px * 4;
            }
            if (img_aligned16) {                                         // (uniform: a caller's image view may start anywhere)
#pragma unroll
                for (int r = 0; r < 3; ++r) v[r] = *reinterpret_cast<const f32x4_p*>(src[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) v[r] = f32x4_p{src[r][0], src[r][1], src[r][2], src[r][3]};
            }
            __syncthreads();                                             // (the previous tile's fragment reads are done)
#pragma unroll
            for (int r = 0; r < 3; ++r) *reinterpret_cast<f32x4_p*>(As + lane * PE_LDA + (wave + 4 * r) * 4) = v[r];
        }
        __syncthreads();
        f32x4_p acc[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = f32x4_p{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            float af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = As[(16 * i + li) * PE_LDA + 4 * ks + g];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wfr[j][ks], af[i], acc[i][j], 0, 0, 0);
        }
        // ---- epilogue: per 32-row half through LDS, 8 lanes per row: bias is already in, LayerNorm, store ----
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();                                   // the first half has been read back
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * half + ii, row = 16 * ii + li;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ch = (wave * 48 + 16 * j + 4 * g) >> 2;
                    *reinterpret_cast<f32x4_p*>(Cs + row * PE_N + ((ch ^ (row & 7)) << 2)) = acc[i][j] + bv[j];
                }
            }
            __syncthreads();
            const long m = (long)t * PE_TM + half * 32 + er;
            f32x4_p xv[6];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int ch = ec + 8 * k;
                xv[k] = *reinterpret_cast<const f32x4_p*>(Cs + er * PE_N + ((ch ^ (er & 7)) << 2));
                sum += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            }
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
            const float mean = sum / (float)PE_N;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                xv[k] = xv[k] - mean;
                sq += (xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1]) + (xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3]);
            }
            sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
            const float rstd = 1.0f / sqrtf(sq / (float)PE_N + eps);
            float sum1 = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int c0 = (ec + 8 * k) * 4;
                const f32x4_p gm = *reinterpret_cast<const f32x4_p*>(gb_s + c0), bt = *reinterpret_cast<const f32x4_p*>(gb_s + PE_N + c0);
                xv[k] = xv[k] * rstd * gm + bt;
                if (m < M) *reinterpret_cast<f32x4_p*>(x + m * ldx + c0) = xv[k];
                sum1 += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            }
            if (xn) {
                // the first block's norm1 (swin.rs:355) of the row just written, as the bf16 operand of its qkv GEMM: the row is still in registers
                sum1 += __shfl_xor(sum1, 1); sum1 += __shfl_xor(sum1, 2); sum1 += __shfl_xor(sum1, 4);
                const float mean1 = sum1 / (float)PE_N;
                float sq1 = 0.f;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    xv[k] = xv[k] - mean1;
                    sq1 += (xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1]) + (xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3]);
                }
                sq1 += __shfl_xor(sq1, 1); sq1 += __shfl_xor(sq1, 2); sq1 += __shfl_xor(sq1, 4);
                const float rstd1 = 1.0f / sqrtf(sq1 / (float)PE_N + eps);
                if (m < M) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int c0 = (ec + 8 * k) * 4;
                        const f32x4_p gm = *reinterpret_cast<const f32x4_p*>(gb_s + 2 * PE_N + c0), bt = *reinterpret_cast<const f32x4_p*>(gb_s + 3 * PE_N + c0);
                        const f32x4_p o = xv[k] * rstd1 * gm + bt;
                        if (xn_f16) {                 // compute mode BRN_F16
                            typedef _Float16 f16x4_p __attribute__((ext_vector_type(4)));
                            f16x4_p h;
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = (_Float16)o[e];
                            *reinterpret_cast<f16x4_p*>(reinterpret_cast<_Float16*>(xn) + m * ldxn + c0) = h;
                        } else {
                        typedef __bf16 bf16x4_p __attribute__((ext_vector_type(4)));
                        bf16x4_p h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h[e] = (__bf16)o[e];
                        *reinterpret_cast<bf16x4_p*>(reinterpret_cast<__bf16*>(xn) + m * ldxn + c0) = h;
                        }
                    }
                }
            }
        }
    }
}

bool patch_embed_ln_eligible(int Cin, int N, int k, int stride, int H, int W, int ldw, int ldx) {
    return Cin == 3 && N == PE_N && k == 4 && stride == 4 && H >= 4 && W >= 4 && (H & 3) == 0 && (W & 3) == 0 && ldw >= PE_K && (ldx & 3) == 0;
}
hipError_t launch_patch_embed_ln(const float* img, int B, int H, int W, const float* wgt, int ldw, const float* bias, const float* gamma,
                                 const float* beta, float eps, float* x, int ldx, hipStream_t s, const float* gamma1, const float* beta1,
                                 void* xn_bf16, int ldxn, int xn_f16) {
    if (!patch_embed_ln_eligible(3, PE_N, 4, 4, H, W, ldw, ldx) || B <= 0 || !img || !wgt || !gamma || !beta || !x) return hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(x) & 15) return hipErrorInvalidValue;
    if (xn_bf16 && (!gamma1 || !beta1 || (ldxn & 3) || (reinterpret_cast<uintptr_t>(xn_bf16) & 7))) return hipErrorInvalidValue;
    const int img_aligned16 = (reinterpret_cast<uintptr_t>(img) & 15) == 0;
    const long M = (long)B * (H >> 2) * (W >> 2);
    long tiles = (M + PE_TM - 1) / PE_TM;
    const int grid = (int)(tiles < 1024 ? tiles : 1024);
    hipLaunchKernelGGL(patch_embed_ln_kernel, dim3(grid), dim3(256), 0, s, img, B, H, W, wgt, ldw, bias, gamma, beta, eps, x, ldx, img_aligned16, gamma1, beta1, xn_bf16, ldxn, xn_f16);
    return hipGetLastError();
}

}  // namespace brn
