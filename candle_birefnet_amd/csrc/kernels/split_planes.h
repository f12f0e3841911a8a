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


// Mode f32_half2: the two fp16 planes of s x (s a power of two, so s x is exact): hi = RN_f16(s x), lo = RN_f16(s x - hi).  The difference is
// exact in fp32 (one fma, the fp16 operand read straight out of the packed register by v_fma_mix), so hi + lo = s x up to 2^-22 |s x| while
// lo is a normal fp16 (|s x| >= 2^-3) and up to 2^-25 absolute below that.  5 VALU per pair of elements, the scaling included (the bf16 split
// is 6).  |s x| >= 65520 rounds to Inf and the products to NaN: out of the mode's range, never a silently wrong finite result.
__device__ __forceinline__ void split_pair_h(const float x0, const float x1, const float s, unsigned& hi, unsigned& lo) {
    float t0, t1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(t0) : "v"(x0), "v"(s), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(t1) : "v"(x1), "v"(s), "v"(hi));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(t0), "v"(t1));
}
template <bool MASKED = true>
__device__ __forceinline__ void split4h(const f32x4_sp v, const unsigned keep, const float s, bf16x4 (&out)[2]) {   // (bf16x4 = the 8-byte container)
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = MASKED ? valu_and(v[e], keep) : v[e];
    unsigned h0, h1, l0, l1;
    split_pair_h(r[0], r[1], s, h0, l0);
    split_pair_h(r[2], r[3], s, h1, l1);
    const u32x2 h = {h0, h1}, l = {l0, l1};
    out[0] = __builtin_bit_cast(bf16x4, h);
    out[1] = __builtin_bit_cast(bf16x4, l);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


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
// the P2 layout of mode f32_half2: the same bytes, the two planes are the fp16 planes of s * v (split4h)
__device__ __forceinline__ void store_planes_h(float* row, int col, const f32x4_sp v, const float s) {
    bf16x4 sp[2];
    split4h<false>(v, 0xffffffffu, s, sp);
    char* base = reinterpret_cast<char*>(row) + (col >> 5) * 128 + (col & 31) * 2;
    *reinterpret_cast<bf16x4*>(base) = sp[0];
    *reinterpret_cast<bf16x4*>(base + 64) = sp[1];
}
__device__ __forceinline__ void store_planes_n(int np, float* row, int col, const f32x4_sp v) {
    if (np == 2) store_planes<2>(row, col, v); else store_planes<3>(row, col, v);
}

}  // namespace brn
