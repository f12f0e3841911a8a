// split_planes.h — device helpers shared by the kernels that produce or consume bf16 operand planes.
#pragma once
#include <hip/hip_runtime.h>

namespace brn {

typedef float f32x4_sp __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Error-free split of four fp32 values (AND-ed with a 0 / ~0 row mask: a masked element is +0 whatever was loaded from the clamped
// address, NaN and Inf included — candle's conv zero-pads) into NP bf16 planes: plane 0 = RNE bf16(x), plane p+1 = RNE
// bf16 of what is left.  Written with one-instruction asm pieces on purpose: left to the compiler, the multiplies and
// subtractions become packed-fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 with op_sel), and with those this kernel's
// producer waves stored wrong A rows a few times per 10^5 K tiles while MFMA waves shared their SIMD (always the last 16
// lanes, always the op_sel'd operand; tools/race_ints.py is the reproducer).  Plain VALU forms are also cheaper beside MFMAs.
__device__ __forceinline__ float valu_and(float a, unsigned keep) { float r; asm("v_and_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(keep)); return r; }
__device__ __forceinline__ float valu_sub(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned valu_cvt_pk_bf16(float a, float b) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int NP, bool MASKED = true>   // MASKED = false: the caller's rows are all valid (or were zero-filled by the load): no AND
__device__ __forceinline__ void split4(const f32x4_sp v, const unsigned keep, bf16x4 (&out)[NP]) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = MASKED ? valu_and(v[e], keep) : v[e];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        u32x2 h;
        h[0] = valu_cvt_pk_bf16(r[0], r[1]);
        h[1] = valu_cvt_pk_bf16(r[2], r[3]);
        out[pl] = __builtin_bit_cast(bf16x4, h);
        if (pl + 1 < NP) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                r[2 * q] = valu_sub(r[2 * q], __builtin_bit_cast(float, h[q] << 16));
                r[2 * q + 1] = valu_sub(r[2 * q + 1], __builtin_bit_cast(float, h[q] & 0xffff0000u));
            }
        }
    }
}


// "P" activation layout (modes f32_split2 / f32_split3): an fp32 matrix [M][K], K % 32 == 0, stored by its PRODUCER as the NP
// bf16 planes the GEMM would otherwise split out while staging it.  K tile kt (32 elements) of a row occupies NP x 64 bytes:
// plane p at bytes [64 NP kt + 64 p, + 64).  NP = 2 is byte-for-byte the size of the fp32 row (ld unchanged); NP = 3 rows are
// 1.5x as long (ld = 3K/2 floats).  `row` points at the row's first byte, `col` (a multiple of 4) is the logical column of v[0].
template <int NP>
__device__ __forceinline__ void store_planes(float* row, int col, const f32x4_sp v) {
    bf16x4 sp[NP];
    split4<NP, false>(v, 0xffffffffu, sp);
    char* base = reinterpret_cast<char*>(row) + (col >> 5) * (64 * NP) + (col & 31) * 2;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<bf16x4*>(base + 64 * pl) = sp[pl];
}
__device__ __forceinline__ void store_planes_n(int np, float* row, int col, const f32x4_sp v) {
    if (np == 2) store_planes<2>(row, col, v); else store_planes<3>(row, col, v);
}

}  // namespace brn
