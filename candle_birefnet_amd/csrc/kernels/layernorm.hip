// layernorm.hip — candle_nn::layer_norm forward (swin.rs:333,335,486,680,754), one wave64 per row, the row kept
// in registers (two-pass mean / biased variance, eps inside the sqrt), float4 loads and stores.  HBM-bound.
// mode 1 fuses PatchMerging's 2x2 strided gather + concat (swin.rs:505-522) into the load.
#include "../brn_kernels.h"
#include "split_planes.h"

namespace brn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LN_MAX_V = 12;   // float4 per lane: C <= 64*4*12 = 3072
constexpr int LN_WAVES = 4;

template <int MODE, int NV>
__global__ void __launch_bounds__(LN_WAVES * 64) layernorm_kernel(const LayerNormParams p) {
    const int lane = threadIdx.x & 63;
    const int nv = p.C >> 2;   // float4 per row
    const long nwaves = (long)gridDim.x * LN_WAVES;
    // each wave walks rows with a grid stride (few, fat workgroups: the kernel is launch/latency bound otherwise)
    for (long row = (long)blockIdx.x * LN_WAVES + (threadIdx.x >> 6); row < p.rows; row += nwaves) {
    f32x4 v[NV];
    int h2 = 0, w2 = 0, bi = 0, Ho = 0, Wo = 0;
    if (MODE == 1) {
        Ho = (p.H + 1) >> 1; Wo = (p.W + 1) >> 1;
        const int hw = Ho * Wo;
        bi = (int)(row / hw);
        const int rem = (int)(row - (long)bi * hw);
        h2 = rem / Wo; w2 = rem - h2 * Wo;
    }
    const int cin4 = p.Cin >> 2;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (c4 < nv) {
            if (MODE == 0) {
                t = *reinterpret_cast<const f32x4*>(p.x + row * p.ldx + c4 * 4);
            } else {
                // concat order (even,even),(odd,even),(even,odd),(odd,odd) in (row,col) — swin.rs:509-519
                const int part = c4 / cin4, cc = c4 - part * cin4;
                const int y = 2 * h2 + (part & 1), x = 2 * w2 + (part >> 1);
                if (y < p.H && x < p.W)
                    t = *reinterpret_cast<const f32x4*>(p.x + (((long)bi * p.H + y) * p.W + x) * p.Cin + cc * 4);
            }
            sum += (t[0] + t[1]) + (t[2] + t[3]);
        }
        v[i] = t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + i * 64 < nv) {
            const f32x4 d = v[i] - mean;
            sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)p.C + p.eps);
    float* yrow = p.y + row * p.ldy + p.y_coff;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c4 = lane + i * 64;
        if (c4 < nv) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c4 * 4);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(p.beta + c4 * 4);
            const f32x4 o = (v[i] - mean) * rstd * gm + bt;
            if (p.y_planes && p.y_h2 > 0.f) store_planes_h(p.y + row * p.ldy, p.y_coff + c4 * 4, o, p.y_h2);   // consumer = an fp16-pair GEMM (mode f32_half2)
            else if (p.y_planes) store_planes_n(p.y_planes, p.y + row * p.ldy, p.y_coff + c4 * 4, o);   // consumer = a split-bf16 GEMM (P layout)
            else if (p.y_bf16) {                 // compute mode BRN_BF16: the consumer GEMM reads bf16 (ldy / y_coff in bf16 elements)
                if (p.y_bf16 == 2) {             // compute mode BRN_F16: fp16
                    typedef _Float16 f16x4_ln __attribute__((ext_vector_type(4)));
                    f16x4_ln h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (_Float16)o[e];
                    *reinterpret_cast<f16x4_ln*>(reinterpret_cast<_Float16*>(p.y) + row * p.ldy + p.y_coff + c4 * 4) = h;
                } else {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)o[e];
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.y) + row * p.ldy + p.y_coff + c4 * 4) = h;
                }
            } else *reinterpret_cast<f32x4*>(yrow + c4 * 4) = o;
        }
    }
    }
}

hipError_t launch_layernorm(const LayerNormParams& p, hipStream_t s) {
    if (p.C % 4 || p.C > 64 * 4 * LN_MAX_V || p.rows <= 0) return hipErrorInvalidValue;
    if (p.mode == 1 && (p.C != 4 * p.Cin || p.Cin % 4)) return hipErrorInvalidValue;
    if (p.y_bf16 && (p.y_planes || p.ldy % 4 || p.y_coff % 4)) return hipErrorInvalidValue;
    if (p.y_planes && (!(p.y_planes == 2 || p.y_planes == 3) || p.C % 32 || p.ldy % 16 || p.y_coff % 32)) return hipErrorInvalidValue;
    if (p.y_h2 > 0.f && p.y_planes != 2) return hipErrorInvalidValue;
    long blocks = (p.rows + LN_WAVES - 1) / LN_WAVES;
    if (blocks > 2048) blocks = 2048;            // 8 workgroups per CU, grid-stride over rows
    dim3 grid((unsigned)blocks), block(LN_WAVES * 64);
    const int nvl = (p.C / 4 + 63) / 64;         // float4 per lane
#define BRN_LN(MODE_, NV_) hipLaunchKernelGGL((layernorm_kernel<MODE_, NV_>), grid, block, 0, s, p)
#define BRN_LN_NV(MODE_)                                                                        \
    if (nvl <= 1) BRN_LN(MODE_, 1); else if (nvl <= 2) BRN_LN(MODE_, 2); else if (nvl <= 3) BRN_LN(MODE_, 3); \
    else if (nvl <= 6) BRN_LN(MODE_, 6); else BRN_LN(MODE_, 12)
    if (p.mode == 0) { BRN_LN_NV(0); } else { BRN_LN_NV(1); }
#undef BRN_LN_NV
#undef BRN_LN
    return hipGetLastError();
}

}  // namespace brn
