// window_attention.hip — fused (shifted-)window attention for the Swin-L geometry (window 12x12 = 144 tokens,
// head_dim 32), exact fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup = one (window, head).  It replaces, for that window/head, everything the reference does between
// the qkv Linear and the proj Linear (swin.rs:359-401 + 266-307):
//   pad_with_zeros (swin.rs:359-366)      -> a pad token's q/k/v is the qkv bias vector (LN output row == 0)
//   roll_2d(-shift) (swin.rs:371-377)     -> source position = (window position + shift) mod (Hp, Wp)
//   window_partition (swin.rs:446-459)    -> index math on load
//   q*scale, q@k^T, +bias, +mask, softmax_last_dim, @v (swin.rs:278-303)
//   transpose/reshape, window_reverse, roll_2d(+shift), narrow (swin.rs:306-307, 387-401) -> index math on store
// so none of the reference's ~8 full-tensor copies per block and no [B_,h,144,144] score tensor ever touch HBM.
//
// MFMA formulation (all tiles 16x16, k-step 4; C/D map: col = lane&15, row = 4*(lane>>4) + reg):
//   S^T[key][query] = K . Q^T   A = K fragment (row = key),  B = Q^T (col = query); lane group g = lane>>4
//                               contracts d = 8g + s at step s (both operands use the same permutation), so a
//                               lane's 8 q (or k) values are 8 consecutive floats of one row.
//   bias                        = this head's 529-entry table column in LDS, indexed arithmetically (no [h,144,144] tensor)
//   softmax over keys           = over the rows of S^T for a fixed column -> in-lane over 36 registers, then
//                               across the 4 lane groups (shfl_xor 16, 32).
//   O^T[d][query] = V^T . P^T   B = P^T taken straight from the S^T accumulator registers (lane group g holds keys
//                               4g..4g+3 of a 16-key tile; step r contracts key 4g + r), A = V^T read from LDS with
//                               the same key permutation.  No transpose, no LDS round trip for P.
#include "../brn_kernels.h"
#include "split_planes.h"
#include <cstdlib>
#include <type_traits>

namespace brn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WS = 12;
constexpr int NTOK = 144;
constexpr int HD = 32;
constexpr int KV_LD = 36;          // floats per LDS row of K and V
constexpr int ATT_THREADS = 192;   // 3 waves, 3 query tiles each

// Both kernels take TWO geometries (the full- and the half-scale map of a co-batched backbone pass, same C / heads / shift):
// blocks [0, nblk0) work on pa, the rest on pb — one launch, one ramp and one tail instead of two (the half-scale launch alone
// fills a fifth of the CU slots).  nblk0 = all blocks when there is only one map.
// WSZ = window side: 12 (Swin-B / L, swin.rs:60,74) or 7 (Swin-T / S, swin.rs:32,46); head_dim is 32 in all four configurations.
// A window holds WSZ^2 tokens = NT16 key / query tiles of 16; the tail of the last tile (49 -> 64) is dummy: zero K / V rows, scores
// forced to -3e38 so that they leave the softmax, queries never stored.
// NWV = waves per workgroup: 3 (three query tiles each at window 12) or 9 (one query tile each: the workgroup lives a third as long on
// the same 41.5 KB of LDS = 3 workgroups per CU; at batch 1 a stage-2 launch is 1080 workgroups on 768 slots = two rounds whatever
// the wave count, so the round time is what counts)
// IOB = the qkv matrix and the output are bf16 (compute mode BRN_BF16 with a window the bf16 kernel below is not built for: window 7 of
// Swin-T / S): operands are widened on load — a pad token's q / k / v is the qkv bias rounded to bf16, what the qkv GEMM of that mode would
// have stored — the arithmetic stays the exact fp32 MFMA chain, the result is rounded once on store.
template <int IOB>
__device__ __forceinline__ f32x4 att_load4(const float* base, long off) {
    if constexpr (IOB != 0) {                   // 1: bf16 matrices, 2: fp16 matrices (compute mode BRN_F16)
        using E = std::conditional_t<IOB == 2, _Float16, __bf16>;
        typedef E ex4_ld __attribute__((ext_vector_type(4)));
        const ex4_ld h = *reinterpret_cast<const ex4_ld*>(reinterpret_cast<const E*>(base) + off);
        return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else return *reinterpret_cast<const f32x4*>(base + off);
}
template <int IOB>
__device__ __forceinline__ f32x4 att_bias4(const float* bias) {
    f32x4 v = *reinterpret_cast<const f32x4*>(bias);
    if constexpr (IOB != 0) {
        using E = std::conditional_t<IOB == 2, _Float16, __bf16>;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)(E)v[e];
    }
    return v;
}
template <int WSZ, int NWV = 3, int IOB = 0>
__global__ void __launch_bounds__(NWV * 64) window_attention_f32_kernel(const WindowAttnParams pa, const WindowAttnParams pb, const int nblk0) {
    constexpr int WS = WSZ, NTOK = WSZ * WSZ, NT16 = (NTOK + 15) / 16, NPADTOK = NT16 * 16;
    constexpr int NTHR = NWV * 64;
    const bool second = (int)blockIdx.x >= nblk0;
    const WindowAttnParams& p = second ? pb : pa;
    __shared__ __attribute__((aligned(16))) float Ks[NPADTOK * KV_LD];
    __shared__ __attribute__((aligned(16))) float Vs[NPADTOK * KV_LD];
    __shared__ int src_s[NPADTOK];   // source token offset (pixel index) or -1 for a pad token
    __shared__ int rid_s[NPADTOK];   // SW-MSA region id of the token (swin.rs:608-629)
    __shared__ float tab_s[(2 * WS - 1) * (2 * WS - 1)];   // this head's column of relative_position_bias_table

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y;
    const int nWw = p.Wp / WS, nW = (p.Hp / WS) * nWw;
    const int bw = second ? (int)blockIdx.x - nblk0 : (int)blockIdx.x;
    const int b = bw / nW, w = bw - b * nW;      // window batch index is b*nW + w (swin.rs:456-458)
    const int wr = w / nWw, wc = w - wr * nWw;
    const int C = p.C, C3 = 3 * C;

    if (tid < NTOK) {
        const int ti = tid / WS, tj = tid - ti * WS;
        const int ph = wr * WS + ti, pw = wc * WS + tj;       // position on the shifted, padded canvas
        int sh = ph + p.shift, sw = pw + p.shift;              // shifted[i] = x[(i + shift) mod Hp] (swin.rs:412-444)
        if (sh >= p.Hp) sh -= p.Hp;
        if (sw >= p.Wp) sw -= p.Wp;
        src_s[tid] = (sh < p.H && sw < p.W) ? (b * p.H + sh) * p.W + sw : -1;
        const int fh = ph < p.Hp - WS ? 0 : (ph < p.Hp - p.shift ? 1 : 2);   // h_slices / w_slices, swin.rs:608-617
        const int fw = pw < p.Wp - WS ? 0 : (pw < p.Wp - p.shift ? 1 : 2);
        rid_s[tid] = fh * 3 + fw;
    } else if (tid < NPADTOK) { src_s[tid] = -1; rid_s[tid] = 0; }
    // window 12: the table is kept REVERSED and in log2 units (tab_s[j] = log2(e) * table[528 - j]): the 4 keys 16 kt + 4 g + {0..3} of
    // a lane lie in one window row (12 % 4 == 0), so their entries are 4 consecutive words — one index per key tile instead of
    // one per score — and the softmax runs on exp2.  Window 7 keeps the plain table and the per-score index.
    constexpr int TABN = (2 * WS - 1) * (2 * WS - 1);
    constexpr bool REV = WS == 12;
    constexpr float LOG2E = 1.4426950408889634f;
    for (int i = tid; i < TABN; i += NTHR) tab_s[i] = REV ? LOG2E * p.rel_table[head * TABN + (TABN - 1 - i)] : p.rel_table[head * TABN + i];
    // the shift mask (swin.rs:283-296) is non-zero only in the last row / column of windows
    const bool has_mask = p.shift > 0 && (wr == p.Hp / WS - 1 || wc == nWw - 1);
    __syncthreads();

    // ---- stage K and V of this (window, head) into LDS: 144 rows x 8 float4 each ----
    for (int idx = tid; idx < NPADTOK * 8; idx += NTHR) {
        const int t = idx >> 3, c4 = (idx & 7) * 4;
        const int src = src_s[t];
        f32x4 kv, vv;
        if (src >= 0) { kv = att_load4<IOB>(p.qkv, (long)src * C3 + C + head * HD + c4); vv = att_load4<IOB>(p.qkv, (long)src * C3 + 2 * C + head * HD + c4); }
        else { kv = att_bias4<IOB>(p.qkv_bias + C + head * HD + c4); vv = att_bias4<IOB>(p.qkv_bias + 2 * C + head * HD + c4); }
        if (t >= NTOK) { kv = f32x4{0.f, 0.f, 0.f, 0.f}; vv = kv; }      // dummy tail of the last tile
        *reinterpret_cast<f32x4*>(Ks + t * KV_LD + c4) = kv;
        *reinterpret_cast<f32x4*>(Vs + t * KV_LD + c4) = vv;
    }
    __syncthreads();

    const int li = lane & 15, g = lane >> 4;

    // the Q fragment of the NEXT query tile is requested while this one is multiplied (8 registers; fetching all three up front
    // cost occupancy and was slower)
    auto load_q = [&](int qt, f32x4& q0, f32x4& q1) {
        const int qs = src_s[qt * 16 + li];
        if (qs >= 0) { q0 = att_load4<IOB>(p.qkv, (long)qs * C3 + head * HD + g * 8); q1 = att_load4<IOB>(p.qkv, (long)qs * C3 + head * HD + g * 8 + 4); }
        else { q0 = att_bias4<IOB>(p.qkv_bias + head * HD + g * 8); q1 = att_bias4<IOB>(p.qkv_bias + head * HD + g * 8 + 4); }
    };
    f32x4 qn0 = {0.f, 0.f, 0.f, 0.f}, qn1 = qn0;
    if (wave < NT16) load_q(wave, qn0, qn1);
    for (int qt = wave; qt < NT16; qt += NWV) {
        const int qtok = qt * 16 + li;
        const int qsrc = src_s[qtok];
        const int qrid = rid_s[qtok];
        // relative position index (swin.rs:182-184): (qi-ki+11)*23 + (qj-kj+11) = qbase - (key + 11*(key/12))
        const int qt_ = qtok < NTOK ? qtok : NTOK - 1;            // (dummy queries of the last tile compute on a valid row, never stored)
        const int qbase = (qt_ / WS + WS - 1) * (2 * WS - 1) + (qt_ % WS) + WS - 1;
        // Q fragment: d = 8g .. 8g+7 of query row qtok, scaled before the product (swin.rs:278)
        float qf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { qf[e] = qn0[e] * p.scale; qf[4 + e] = qn1[e] * p.scale; }
        if (qt + NWV < NT16) load_q(qt + NWV, qn0, qn1);
        // S^T = K . Q^T : 9 key tiles
        f32x4 st[NT16];
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt) {
            const float* kp = Ks + (kt * 16 + li) * KV_LD + g * 8;
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(kp);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(kp + 4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[e], qf[e], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[e], qf[4 + e], acc, 0, 0, 0);
            st[kt] = acc;
            if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // bound the scheduler's hoisting (register pressure)
        }
        // + relative position bias (swin.rs:284-285), + SW-MSA mask (swin.rs:288-297, value -100 swin.rs:651)
        float mx = -3.0e38f;
        if constexpr (REV) {
            const int qrev = (TABN - 1) - qbase;
#pragma unroll
            for (int kt = 0; kt < NT16; ++kt) {
                const int key0 = kt * 16 + g * 4;
                const float* tb = tab_s + qrev + key0 + (WS - 1) * (key0 / WS);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = fmaf(st[kt][r], LOG2E, tb[r]);
            }
            if (has_mask) {
#pragma unroll
                for (int kt = 0; kt < NT16; ++kt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kt][r] += (rid_s[kt * 16 + g * 4 + r] != qrid) ? -100.0f * LOG2E : 0.0f;
                }
            }
#pragma unroll
            for (int kt = 0; kt < NT16; ++kt) mx = fmaxf(fmaxf(mx, fmaxf(st[kt][0], st[kt][1])), fmaxf(st[kt][2], st[kt][3]));
        } else {
#pragma unroll
            for (int kt = 0; kt < NT16; ++kt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + g * 4 + r;
                    const int kk = key < NTOK ? key : 0;
                    float s = st[kt][r] + tab_s[max(0, qbase - kk - (WS - 1) * (kk / WS))];
                    if (p.shift > 0) s += (rid_s[key] != qrid) ? -100.0f : 0.0f;
                    if (NTOK % 16 != 0 && key >= NTOK) s = -3.0e38f;
                    st[kt][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = REV ? __builtin_amdgcn_exp2f(st[kt][r] - mx) : __expf(st[kt][r] - mx);
                st[kt][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // O^T = V^T . P^T : two d tiles
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT16; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = Vs + (kt * 16 + g * 4 + r) * KV_LD + li;
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], st[kt][r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16], st[kt][r], o1, 0, 0, 0);
            }
            if (kt % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // lane holds O^T[d = 16*dt + 4g + r][query = li]; softmax denominator applied here (swin.rs:300,303)
        if (qsrc >= 0 && qtok < NTOK) {
            const float inv = 1.0f / sum;
            if (p.out_planes == 3) {        // mode f32_split3: the proj GEMM reads the 3-plane P layout (rows 1.5x as long)
                float* orow = p.out + (long)qsrc * (C + C / 2);
                store_planes<3>(orow, head * HD + g * 4, o0 * inv);
                store_planes<3>(orow, head * HD + 16 + g * 4, o1 * inv);
            } else if (p.out_planes == 2) { // mode f32_half2: the proj GEMM reads the two fp16 planes (rows as long as fp32 rows)
                float* orow = p.out + (long)qsrc * C;
                store_planes_h(orow, head * HD + g * 4, o0 * inv, p.out_h2);
                store_planes_h(orow, head * HD + 16 + g * 4, o1 * inv, p.out_h2);
            } else if constexpr (IOB != 0) {
                using E = std::conditional_t<IOB == 2, _Float16, __bf16>;
                typedef E ex4_st __attribute__((ext_vector_type(4)));
                E* op = reinterpret_cast<E*>(p.out) + (long)qsrc * C + head * HD + g * 4;
                ex4_st h0, h1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { h0[e] = (E)(o0[e] * inv); h1[e] = (E)(o1[e] * inv); }
                *reinterpret_cast<ex4_st*>(op) = h0;
                *reinterpret_cast<ex4_st*>(op + 16) = h1;
            } else {
                float* op = p.out + (long)qsrc * C + head * HD + g * 4;
                *reinterpret_cast<f32x4*>(op) = o0 * inv;
                *reinterpret_cast<f32x4*>(op + 16) = o1 * inv;
            }
        }
    }
}

// =====================================================================================================================
// window_attention_split_kernel — the same fused window attention on the bf16 matrix cores (compute modes f32_split2 /
// bf16_operands).  q*scale, K, V and the softmax numerators P are split into NP bf16 terms (x ~ x_h + x_l) in registers /
// while staging, every product is the sum of the NP*(NP+1)/2 leading plane products on v_mfma_f32_16x16x32_bf16 (head_dim
// 32 = ONE MFMA k), accumulation, bias, mask and softmax stay fp32.  MFMA cycles per 16-query tile: 57 x 16 vs 144 x 32.
// LDS (40.2 KB -> 4 workgroups per CU instead of 3: the 864 workgroups of a Swin-L stage-2 launch fit one round):
//   Kp[pl][key][32 d]   bf16, 64-byte rows, 16-byte chunk c stored at c ^ perm[(key >> 2) & 3], perm = {0,2,3,1}
//                       (conflict-free ds_read_b128 of the A operand: row = key, k = d = 8g + j)
//   Vt[pl][d][148 keys] bf16 (V transposed): the A operand of O^T = V^T P^T needs 8 keys per lane for a fixed d; the k slots
//                       of an MFMA over key tiles (2t, 2t+1) are ordered so that lane group g contracts keys 16*(2t)+4g+0..3
//                       and 16*(2t+1)+4g+0..3 — exactly the keys whose P^T values the lane already holds in its S^T
//                       accumulators (C/D map row = 4g + reg) — so P never leaves registers.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int VT_LD = 148;

__device__ __forceinline__ int kswz(int key) { return (0x78 >> (2 * ((key >> 2) & 3))) & 3; }

// H (mode f32_half2): the two planes are fp16 planes of the scaled operands (kernels/split_planes.h, split_pair_h) — q (with its head_dim^-0.5), k
// and v scaled by 8, the un-normalised probabilities (in (0, 1]) by 2048 — on v_mfma_f32_16x16x32_f16; the scores and the output are
// un-scaled exactly.  ~2^-22 relative per product instead of 2^-16.
constexpr float ATT_H_QKV = 8.0f, ATT_H_P = 2048.0f;
template <int NP, bool H = false>
__global__ void __launch_bounds__(ATT_THREADS) window_attention_split_kernel(const WindowAttnParams pa, const WindowAttnParams pb, const int nblk0) {
    static_assert(!H || NP == 2, "fp16 planes come in pairs");
    const bool second = (int)blockIdx.x >= nblk0;
    const WindowAttnParams& p = second ? pb : pa;
    __shared__ __attribute__((aligned(16))) __bf16 Kp[NP * NTOK * HD];
    __shared__ __attribute__((aligned(16))) __bf16 Vt[NP * HD * VT_LD];
    __shared__ float tab_s[(2 * WS - 1) * (2 * WS - 1)];
    __shared__ int src_s[NTOK];
    __shared__ unsigned char rid_s[NTOK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y;
    const int nWw = p.Wp / WS, nW = (p.Hp / WS) * nWw;
    const int bw = second ? (int)blockIdx.x - nblk0 : (int)blockIdx.x;
    const int b = bw / nW, w = bw - b * nW;
    const int wr = w / nWw, wc = w - wr * nWw;
    const int C = p.C, C3 = 3 * C;

    if (tid < NTOK) {
        const int ti = tid / WS, tj = tid - ti * WS;
        const int ph = wr * WS + ti, pw = wc * WS + tj;
        int sh = ph + p.shift, sw = pw + p.shift;
        if (sh >= p.Hp) sh -= p.Hp;
        if (sw >= p.Wp) sw -= p.Wp;
        src_s[tid] = (sh < p.H && sw < p.W) ? (b * p.H + sh) * p.W + sw : -1;
        const int fh = ph < p.Hp - WS ? 0 : (ph < p.Hp - p.shift ? 1 : 2);
        const int fw = pw < p.Wp - WS ? 0 : (pw < p.Wp - p.shift ? 1 : 2);
        rid_s[tid] = (unsigned char)(fh * 3 + fw);
    }
    for (int i = tid; i < (2 * WS - 1) * (2 * WS - 1); i += ATT_THREADS) tab_s[i] = p.rel_table[head * ((2 * WS - 1) * (2 * WS - 1)) + i];
    __syncthreads();

    // ---- stage K (row-major, swizzled) and V (transposed), split into bf16 planes: items = (token pair, 4-wide d chunk) ----
    for (int idx = tid; idx < (NTOK / 2) * 8; idx += ATT_THREADS) {
        const int tp = idx >> 3, c4 = (idx & 7) * 4;
        f32x4 kv[2], vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tp * 2 + u, src = src_s[t];
            const float* kp = src >= 0 ? p.qkv + (long)src * C3 + C + head * HD + c4 : p.qkv_bias + C + head * HD + c4;
            kv[u] = *reinterpret_cast<const f32x4*>(kp);
            vv[u] = *reinterpret_cast<const f32x4*>(kp + C);
        }
        if constexpr (H) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tp * 2 + u;
                bf16x4 sp[2];
                split4h<false>(kv[u], 0xffffffffu, ATT_H_QKV, sp);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    *reinterpret_cast<bf16x4*>(Kp + (pl * NTOK + t) * HD + (((c4 >> 3) ^ kswz(t)) << 3) + (c4 & 4)) = sp[pl];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned hi, lo;
                split_pair_h(vv[0][e], vv[1][e], ATT_H_QKV, hi, lo);                          // keys (2tp, 2tp+1) of row d = c4+e
                *reinterpret_cast<unsigned*>(Vt + (c4 + e) * VT_LD + tp * 2) = hi;
                *reinterpret_cast<unsigned*>(Vt + (HD + c4 + e) * VT_LD + tp * 2) = lo;
            }
        } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = tp * 2 + u;
            f32x4 r = kv[u];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                bf16x4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (__bf16)r[e];
                *reinterpret_cast<bf16x4*>(Kp + (pl * NTOK + t) * HD + (((c4 >> 3) ^ kswz(t)) << 3) + (c4 & 4)) = h;
                if (pl + 1 < NP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] -= (float)h[e];
                }
            }
        }
        f32x4 r0 = vv[0], r1 = vv[1];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16x2 h;
                h[0] = (__bf16)r0[e]; h[1] = (__bf16)r1[e];
                *reinterpret_cast<bf16x2*>(Vt + (pl * HD + c4 + e) * VT_LD + tp * 2) = h;   // keys (2tp, 2tp+1) of row d = c4+e
                if (pl + 1 < NP) { r0[e] -= (float)h[0]; r1[e] -= (float)h[1]; }
            }
        }
        }
    }
    __syncthreads();

    const int li = lane & 15, g = lane >> 4;
    for (int qt = wave; qt < 9; qt += 3) {
        const int qtok = qt * 16 + li;
        const int qsrc = src_s[qtok];
        const int qrid = rid_s[qtok];
        const int qbase = (qtok / WS + WS - 1) * (2 * WS - 1) + (qtok % WS) + WS - 1;
        // Q fragment (B operand of S^T = K Q^T): d = 8g .. 8g+7 of this lane's query, scaled (swin.rs:278), split
        bf16x8 qf[NP];
        {
            const float* qp = qsrc >= 0 ? p.qkv + (long)qsrc * C3 + head * HD + g * 8 : p.qkv_bias + head * HD + g * 8;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qp + 4);
            float r[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { r[e] = q0[e] * p.scale; r[4 + e] = q1[e] * p.scale; }
            if constexpr (H) {
                typedef unsigned u32x4_q __attribute__((ext_vector_type(4)));
                u32x4_q hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) { unsigned a, b2; split_pair_h(r[2 * e], r[2 * e + 1], ATT_H_QKV, a, b2); hi[e] = a; lo[e] = b2; }
                qf[0] = __builtin_bit_cast(bf16x8, hi);
                qf[1] = __builtin_bit_cast(bf16x8, lo);
            } else {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const __bf16 h = (__bf16)r[e];
                    qf[pl][e] = h;
                    r[e] -= (float)h;
                }
            }
            }
        }
        f32x4 st[9];
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            const int key = kt * 16 + li;
            bf16x8 kf[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                kf[pl] = *reinterpret_cast<const bf16x8*>(Kp + (pl * NTOK + key) * HD + ((g ^ kswz(key)) << 3));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (H) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[1]), __builtin_bit_cast(f16x8, qf[0]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[0]), __builtin_bit_cast(f16x8, qf[1]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kf[0]), __builtin_bit_cast(f16x8, qf[0]), acc, 0, 0, 0);
            } else {
            if (NP == 2) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[1], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[0], acc, 0, 0, 0);
            }
            st[kt] = acc;
            if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + g * 4 + r;
                float sv = H ? fmaf(st[kt][r], 1.0f / (ATT_H_QKV * ATT_H_QKV), tab_s[qbase - key - 11 * (key / WS)]) : st[kt][r] + tab_s[qbase - key - 11 * (key / WS)];
                if (p.shift > 0) sv += ((int)rid_s[key] != qrid) ? -100.0f : 0.0f;
                st[kt][r] = sv;
                mx = fmaxf(mx, sv);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(st[kt][r] - mx);
                st[kt][r] = e;
                sum += e;
            }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        // O^T = V^T P^T: 5 k-steps of 32 keys (key tiles 2t, 2t+1; the 10th tile does not exist: zero operands)
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            bf16x8 pf[NP];
            {
                float r[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { r[e] = st[2 * t][e]; r[4 + e] = (2 * t + 1 < 9) ? st[(2 * t + 1 < 9) ? 2 * t + 1 : 0][e] : 0.f; }
                if constexpr (H) {
                    typedef unsigned u32x4_p __attribute__((ext_vector_type(4)));
                    u32x4_p hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { unsigned a, b2; split_pair_h(r[2 * e], r[2 * e + 1], ATT_H_P, a, b2); hi[e] = a; lo[e] = b2; }
                    pf[0] = __builtin_bit_cast(bf16x8, hi);
                    pf[1] = __builtin_bit_cast(bf16x8, lo);
                } else {
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 h = (__bf16)r[e];
                        pf[pl][e] = h;
                        r[e] -= (float)h;
                    }
                }
                }
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                bf16x8 vf[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    const __bf16* vrow = Vt + (pl * HD + dt * 16 + li) * VT_LD + g * 4;
                    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow + (2 * t) * 16);
                    bf16x4 hi = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
                    if (2 * t + 1 < 9) hi = *reinterpret_cast<const bf16x4*>(vrow + (2 * t + 1) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[pl][e] = lo[e]; vf[pl][4 + e] = hi[e]; }
                }
                f32x4 o = dt == 0 ? o0 : o1;
                if constexpr (H) {
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[1]), __builtin_bit_cast(f16x8, pf[0]), o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[0]), __builtin_bit_cast(f16x8, pf[1]), o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vf[0]), __builtin_bit_cast(f16x8, pf[0]), o, 0, 0, 0);
                } else {
                if (NP == 2) {
                    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1], pf[0], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pf[1], o, 0, 0, 0);
                }
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0], pf[0], o, 0, 0, 0);
                }
                if (dt == 0) o0 = o; else o1 = o;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qsrc >= 0) {
            const float inv = (H ? 1.0f / (ATT_H_P * ATT_H_QKV) : 1.0f) / sum;
            if (H && p.out_planes == 2) {
                float* orow = p.out + (long)qsrc * C;
                store_planes_h(orow, head * HD + g * 4, o0 * inv, p.out_h2);
                store_planes_h(orow, head * HD + 16 + g * 4, o1 * inv, p.out_h2);
            } else if (p.out_planes == 2) {        // the proj GEMM reads the P layout: a head's 32 outputs are one K tile of the row
                float* orow = p.out + (long)qsrc * C;
                store_planes<2>(orow, head * HD + g * 4, o0 * inv);
                store_planes<2>(orow, head * HD + 16 + g * 4, o1 * inv);
            } else {
                float* op = p.out + (long)qsrc * C + head * HD + g * 4;
                *reinterpret_cast<f32x4*>(op) = o0 * inv;
                *reinterpret_cast<f32x4*>(op + 16) = o1 * inv;
            }
        }
    }
}

// =====================================================================================================================
// window_attention_bf16_kernel — compute mode BRN_BF16: qkv arrives as bf16 (the qkv GEMM's output), `out` leaves as bf16
// (the proj GEMM's input).  Same structure and LDS images as window_attention_split_kernel<1>; what changes is the staging
// (16-byte loads of 8 bf16, no split) and that q's scale is applied to the fp32 scores instead of to q (q is already
// rounded to bf16: scaling the accumulator avoids a second rounding; identical in real arithmetic, swin.rs:278).
// =====================================================================================================================
// v_max3_f32 without the canonicalising v_max x, x that fmaxf puts in front of it (the scores are MFMA / fma results: never signalling)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <typename E>
__device__ __forceinline__ auto bias8_s16(const float* bp) {
    typedef E ex8_b __attribute__((ext_vector_type(8)));
    ex8_b r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (E)bp[e];
    return r;
}

// Two heads per workgroup (6 waves: waves 0-2 head 2y, waves 3-5 head 2y + 1): a token's K (or V) rows of two adjacent heads are one
// 128-byte line, so the staging loads fetch whole lines (one head per workgroup used half of every line it touched).
// F16: qkv / out are fp16 matrices (compute mode BRN_F16): the same kernel on v_mfma_f32_16x16x32_f16
template <int ATT_BF16_HPW, bool F16 = false>
__global__ void __launch_bounds__(ATT_BF16_HPW * ATT_THREADS) __attribute__((amdgpu_waves_per_eu(5, 8))) window_attention_bf16_kernel(const WindowAttnParams pa, const WindowAttnParams pb, const int nblk0) {
    using E = std::conditional_t<F16, _Float16, __bf16>;
    typedef E ex8 __attribute__((ext_vector_type(8)));
    typedef E ex4 __attribute__((ext_vector_type(4)));
    typedef E ex2 __attribute__((ext_vector_type(2)));
    auto mfma = [](const ex8 a, const ex8 b, const f32x4 c) -> f32x4 {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    const bool second = (int)blockIdx.x >= nblk0;
    const WindowAttnParams& p = second ? pb : pa;
    constexpr int TABN = (2 * WS - 1) * (2 * WS - 1);                 // 529
    constexpr float LOG2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) E Kp_[ATT_BF16_HPW][NTOK * HD];
    __shared__ __attribute__((aligned(16))) E Vt_[ATT_BF16_HPW][HD * VT_LD];
    // a head's bias table REVERSED and in log2 units: rev[j] = log2(e) * table[528 - j].  The 4 keys 16 kt + 4 g + {0..3} of a lane
    // lie in one window row (12 % 4 == 0), so their table entries are 4 consecutive words of rev: one index per key tile
    // instead of one per score (the index arithmetic was most of the softmax's VALU work).
    __shared__ float rev_[ATT_BF16_HPW][TABN + 3];
    __shared__ __attribute__((aligned(4))) unsigned char rid_s[NTOK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hp = wave / 3, wv = wave - 3 * hp;                      // head of the pair, wave within the head
    const int head = blockIdx.y * ATT_BF16_HPW + hp;
    const int nWh = p.Hp / WS, nWw = p.Wp / WS, nW = nWh * nWw;
    const int bw = second ? (int)blockIdx.x - nblk0 : (int)blockIdx.x;
    const int b = bw / nW, w = bw - b * nW;
    const int wr = w / nWw, wc = w - wr * nWw;
    const int C = p.C, C3 = 3 * C;
    const E* qkv = reinterpret_cast<const E*>(p.qkv);
    const int li = lane & 15, g = lane >> 4;
    // the shift mask (swin.rs:283-296) is non-zero only in the last row / column of windows
    const bool has_mask = p.shift > 0 && (wr == nWh - 1 || wc == nWw - 1);
    E* Kp = Kp_[hp];
    E* Vt = Vt_[hp];
    const float* rev_s = rev_[hp];

    auto tok_src = [&](int t) {                                       // source row of window token t (roll + partition), -1 = pad token
        const int ti = t / WS, tj = t - ti * WS;
        int sh = wr * WS + ti + p.shift, sw = wc * WS + tj + p.shift;
        if (sh >= p.Hp) sh -= p.Hp;
        if (sw >= p.Wp) sw -= p.Wp;
        return (sh < p.H && sw < p.W) ? (b * p.H + sh) * p.W + sw : -1;
    };
    // ---- every global load of the workgroup goes out FIRST, unconditionally, from clamped addresses (a pad token reads row 0 and is
    // fixed up afterwards): the three Q fragments, the head's bias table, the K / V rows.  The first version loaded under lane-dependent
    // branches and in two dependent loops — hipcc waits vmcnt(0) at every join — and paid five HBM round trips one after the other
    // (3 table iterations, 2 staging iterations) in a kernel whose lifetime is little more than its memory latency. ----
    constexpr int NTHR = ATT_BF16_HPW * ATT_THREADS;
    constexpr int NITEM = (NTOK / 2) * 4 * ATT_BF16_HPW;                // (token pair, head of the pair, 8-wide d chunk)
    constexpr int NTAB = ATT_BF16_HPW * TABN;
    static_assert(2 * NTHR >= NITEM && 3 * NTHR >= NTAB, "two staging items and three table words per thread cover a head group");
    ex8 qf[3];
    int qsrc_[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int qs = tok_src((wv + 3 * u) * 16 + li);
        qsrc_[u] = qs;
        qf[u] = *reinterpret_cast<const ex8*>(qkv + (long)max(qs, 0) * C3 + head * HD + g * 8);
    }
    float tbw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = min(tid + j * NTHR, NTAB - 1), hh = i / TABN, jj = i - hh * TABN;
        tbw[j] = p.rel_table[(blockIdx.y * ATT_BF16_HPW + hh) * TABN + (TABN - 1 - jj)];
    }
    ex8 kv[2][2], vv[2][2];
    int ssrc[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = min(tid + it * NTHR, NITEM - 1);
        const int c8 = (idx & 3) * 8, hh = (idx >> 2) & (ATT_BF16_HPW - 1), tp = idx / (4 * ATT_BF16_HPW);
        const int hd = (blockIdx.y * ATT_BF16_HPW + hh) * HD;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int src = tok_src(tp * 2 + u);
            ssrc[it][u] = src;
            const E* kp = qkv + (long)max(src, 0) * C3 + C + hd + c8;
            kv[it][u] = *reinterpret_cast<const ex8*>(kp);
            vv[it][u] = *reinterpret_cast<const ex8*>(kp + C);
        }
    }
    if (tid < NTOK) {
        const int ti = tid / WS, tj = tid - ti * WS;
        const int ph = wr * WS + ti, pw = wc * WS + tj;
        const int fh = ph < p.Hp - WS ? 0 : (ph < p.Hp - p.shift ? 1 : 2);
        const int fw = pw < p.Wp - WS ? 0 : (pw < p.Wp - p.shift ? 1 : 2);
        rid_s[tid] = (unsigned char)(fh * 3 + fw);
    }
    // ---- consume: table, K (row-major, chunk-swizzled), V (transposed); pad tokens take the qkv bias (swin.rs:359-366: their LayerNorm
    // output row is zero) in a second, divergent step that only the windows on the padded border execute ----
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = tid + j * NTHR;
        if (i < NTAB) { const int hh = i / TABN; rev_[hh][i - hh * TABN] = LOG2E * tbw[j]; }
    }
#pragma unroll
    for (int u = 0; u < 3; ++u)
        if (qsrc_[u] < 0) qf[u] = bias8_s16<E>(p.qkv_bias + head * HD + g * 8);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * NTHR;
        if (idx < NITEM) {
            const int c8 = (idx & 3) * 8, hh = (idx >> 2) & (ATT_BF16_HPW - 1), tp = idx / (4 * ATT_BF16_HPW);
            const int hd = (blockIdx.y * ATT_BF16_HPW + hh) * HD;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (ssrc[it][u] < 0) {
                    kv[it][u] = bias8_s16<E>(p.qkv_bias + C + hd + c8);
                    vv[it][u] = bias8_s16<E>(p.qkv_bias + 2 * C + hd + c8);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = tp * 2 + u;
                *reinterpret_cast<ex8*>(Kp_[hh] + t * HD + (((c8 >> 3) ^ kswz(t)) << 3)) = kv[it][u];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ex2 h;
                h[0] = vv[it][0][e]; h[1] = vv[it][1][e];
                *reinterpret_cast<ex2*>(Vt_[hh] + (c8 + e) * VT_LD + tp * 2) = h;
            }
        }
    }
    __syncthreads();

    const float scale2 = p.scale * LOG2E;
    ex8 ones8;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones8[e] = (E)1.0f;
    // per key tile: koff = key0 + 11 (key0 / 12) for key0 = 16 kt + 4 g; the table word of (query, key0 + r) is rev[qrev + koff + r]
    // (byte offsets, one register per key tile, added to an LDS address the compiler cannot take apart: with `rev_s + qrev + koff[kt]`
    // it re-associated the array's constant base out of the sum and spent four VALU instructions per key tile on addresses: 94 of the
    // ~650 vector instructions of a query-tile triple)
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    unsigned koffb[9];
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) { const int key0 = kt * 16 + g * 4; koffb[kt] = (unsigned)(key0 + 11 * (key0 / WS)) * 4u; }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int qt = wv + 3 * u;
        const int qtok = qt * 16 + li;
        const int qsrc = qsrc_[u];
        const unsigned qrid4 = 0x01010101u * (unsigned)rid_s[qtok];
        // table index of (query, key) = qbase - key - 11 (key / 12), qbase = (qi + 11) 23 + qj + 11 (swin.rs:143-152 arithmetically)
        const int qrev = (TABN - 1) - ((qtok / WS + WS - 1) * (2 * WS - 1) + (qtok % WS) + WS - 1);
        f32x4 st[9];
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            const int key = kt * 16 + li;
            const ex8 kf = *reinterpret_cast<const ex8*>(Kp + key * HD + ((g ^ kswz(key)) << 3));
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            st[kt] = mfma(kf, qf[u], acc);
            if (kt % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        float mx = -3.0e38f;
        unsigned tbq = (unsigned)(unsigned long)(lds_cfloat*)(rev_s + qrev);
        asm volatile("" : "+v"(tbq));
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
            lds_cfloat* tb = (lds_cfloat*)(unsigned long)(tbq + koffb[kt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] = fmaf(st[kt][r], scale2, tb[r]);
        }
        if (has_mask) {
#pragma unroll
            for (int kt = 0; kt < 9; ++kt) {
                // byte r of x is 1 where key r lies in another region than the query, else 0: one byte-to-float conversion and one fma per
                // score, and the term added is exactly the reference's -100 (swin.rs:283-296, 651).  (region ids are 0 .. 8, so a byte of
                // the XOR is 0 .. 15: + 0x7f carries into bit 7 exactly when it is non-zero, and never into the next byte)
                const unsigned xr = *reinterpret_cast<const unsigned*>(rid_s + kt * 16 + g * 4) ^ qrid4;
                const unsigned x = ((xr + 0x7f7f7f7fu) >> 7) & 0x01010101u;
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kt][r] = fmaf((float)((x >> (8 * r)) & 0xffu), -100.0f * LOG2E, st[kt][r]);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) mx = max3f(max3f(mx, st[kt][0], st[kt][1]), st[kt][2], st[kt][3]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] - mx);
        }
        // the row sums come out of the matrix pipe: a third "V" tile of ones makes every row of osum the sum over the keys of the
        // SAME bf16 weights the P V product uses (36 adds and two cross-lane steps per query tile less on the vector pipe)
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f}, osum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            ex8 pf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                pf[e] = (E)st[2 * t][e];
                pf[4 + e] = (2 * t + 1 < 9) ? (E)st[(2 * t + 1 < 9) ? 2 * t + 1 : 0][e] : (E)0.f;
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const E* vrow = Vt + (dt * 16 + li) * VT_LD + g * 4;
                const ex4 lo = *reinterpret_cast<const ex4*>(vrow + (2 * t) * 16);
                ex4 hi = {(E)0.f, (E)0.f, (E)0.f, (E)0.f};
                if (2 * t + 1 < 9) hi = *reinterpret_cast<const ex4*>(vrow + (2 * t + 1) * 16);
                ex8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
                if (dt == 0) o0 = mfma(vf, pf, o0);
                else o1 = mfma(vf, pf, o1);
            }
            osum = mfma(ones8, pf, osum);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (qsrc >= 0) {
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            E* op = reinterpret_cast<E*>(p.out) + (long)qsrc * C + head * HD + g * 4;
            ex4 h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (E)(o0[e] * inv); h1[e] = (E)(o1[e] * inv); }
            *reinterpret_cast<ex4*>(op) = h0;
            *reinterpret_cast<ex4*>(op + 16) = h1;
        }
    }
}

static hipError_t check_attention(const WindowAttnParams& p) {
    const int ws = p.ws ? p.ws : WS;
    if (!(ws == 12 || ws == 7)) return hipErrorInvalidValue;
    if (p.C != p.heads * HD || p.Hp % ws || p.Wp % ws || p.Hp < p.H || p.Wp < p.W) return hipErrorInvalidValue;
    if (!(p.shift == 0 || p.shift == ws / 2)) return hipErrorInvalidValue;
    if (ws != WS && (p.planes || p.out_planes)) return hipErrorInvalidValue;   // window 7 (Swin-T / S): the fp32-MFMA kernel (fp32 or bf16 matrices in / out)
    return hipSuccess;
}
hipError_t launch_window_attention2(const WindowAttnParams& p, const WindowAttnParams* p2, hipStream_t s) {
    if (check_attention(p) != hipSuccess) return hipErrorInvalidValue;
    const int ws = p.ws ? p.ws : WS;
    const int n0 = p.B * (p.Hp / ws) * (p.Wp / ws);
    int n1 = 0;
    if (p2) {
        if ((p2->ws ? p2->ws : WS) != ws) return hipErrorInvalidValue;
        if (check_attention(*p2) != hipSuccess || p2->C != p.C || p2->heads != p.heads || p2->planes != p.planes || p2->out_planes != p.out_planes || p2->io_bf16 != p.io_bf16) return hipErrorInvalidValue;
        n1 = p2->B * (p2->Hp / ws) * (p2->Wp / ws);
    }
    if (p.out_planes && !((p.out_planes == 2 && p.planes == 2 && (p.out_h2 > 0.f) == (p.h2 != 0)) || (p.out_planes == 3 && p.planes == 0) || (p.out_planes == 2 && p.planes == 0 && p.out_h2 > 0.f && !p.io_bf16)))
        return hipErrorInvalidValue;
    if (p.h2 && (p.planes != 2 || p.io_bf16)) return hipErrorInvalidValue;
    if (p2 && (p2->out_h2 != p.out_h2 || p2->h2 != p.h2)) return hipErrorInvalidValue;
    const WindowAttnParams& q = p2 ? *p2 : p;
    dim3 grid(n0 + n1, p.heads), block(ATT_THREADS);
    if (p.io_bf16 && ws == 7) {
        if (p.out_planes || (p.C & 3)) return hipErrorInvalidValue;
        if (p.io_bf16 == 2) hipLaunchKernelGGL((window_attention_f32_kernel<7, 3, 2>), grid, block, 0, s, p, q, n0);
        else hipLaunchKernelGGL((window_attention_f32_kernel<7, 3, 1>), grid, block, 0, s, p, q, n0);
    } else if (p.io_bf16) {
        if (p.out_planes || (p.C & 7)) return hipErrorInvalidValue;
        // (round 2, one stream: two heads per workgroup measured 1 % slower end to end, 224.5 vs 226.8 img/s at batch 8)
        // round 3: two heads per workgroup is the default — alone the launch is 6 % slower (3.25 against 3.05 ms per 8-image step), but it reads
        // whole 128-byte lines (one head per workgroup touches half of every line it fetches: 8.5 GB read against 5.9 algorithmic), and with
        // two sub-batch streams sharing the HBM the step is 1.0 % faster (tools/ab_env.sh BRN_ATT_HPW "1 2"); BRN_ATT_HPW=1 selects one head
        static const bool two_heads = !(getenv("BRN_ATT_HPW") && atoi(getenv("BRN_ATT_HPW")) == 1);
        if (p.io_bf16 == 2) {                  // fp16 matrices (compute mode BRN_F16)
            if (two_heads && !(p.heads & 1)) hipLaunchKernelGGL((window_attention_bf16_kernel<2, true>), dim3(n0 + n1, p.heads / 2), dim3(2 * ATT_THREADS), 0, s, p, q, n0);
            else hipLaunchKernelGGL((window_attention_bf16_kernel<1, true>), grid, block, 0, s, p, q, n0);
        } else if (two_heads && !(p.heads & 1)) hipLaunchKernelGGL(window_attention_bf16_kernel<2>, dim3(n0 + n1, p.heads / 2), dim3(2 * ATT_THREADS), 0, s, p, q, n0);
        else hipLaunchKernelGGL(window_attention_bf16_kernel<1>, grid, block, 0, s, p, q, n0);
    } else if (p.planes == 2 && p.h2) hipLaunchKernelGGL((window_attention_split_kernel<2, true>), grid, block, 0, s, p, q, n0);
    else if (p.planes == 2) hipLaunchKernelGGL(window_attention_split_kernel<2>, grid, block, 0, s, p, q, n0);
#ifdef BRN_DIAG_BUILD
    else if (p.planes == 1) hipLaunchKernelGGL(window_attention_split_kernel<1>, grid, block, 0, s, p, q, n0);
#else
    else if (p.planes == 1) return hipErrorInvalidValue;      // mode bf16_operands: diag build only
#endif
    else if (ws == 7) hipLaunchKernelGGL(window_attention_f32_kernel<7>, grid, block, 0, s, p, q, n0);
    else {
        static const int f32_waves = getenv("BRN_ATT_F32_WAVES") ? atoi(getenv("BRN_ATT_F32_WAVES")) : 9;
        if (f32_waves == 9) hipLaunchKernelGGL((window_attention_f32_kernel<12, 9>), grid, dim3(9 * 64), 0, s, p, q, n0);
        else hipLaunchKernelGGL((window_attention_f32_kernel<12, 3>), grid, block, 0, s, p, q, n0);
    }
    return hipGetLastError();
}
hipError_t launch_window_attention(const WindowAttnParams& p, hipStream_t s) { return launch_window_attention2(p, nullptr, s); }

}  // namespace brn
