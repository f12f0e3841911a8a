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

namespace brn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WS = 12;
constexpr int NTOK = 144;
constexpr int HD = 32;
constexpr int KV_LD = 36;          // floats per LDS row of K and V
constexpr int ATT_THREADS = 192;   // 3 waves, 3 query tiles each

__global__ void __launch_bounds__(ATT_THREADS) window_attention_f32_kernel(const WindowAttnParams p) {
    __shared__ __attribute__((aligned(16))) float Ks[NTOK * KV_LD];
    __shared__ __attribute__((aligned(16))) float Vs[NTOK * KV_LD];
    __shared__ int src_s[NTOK];   // source token offset (pixel index) or -1 for a pad token
    __shared__ int rid_s[NTOK];   // SW-MSA region id of the token (swin.rs:608-629)
    __shared__ float tab_s[(2 * WS - 1) * (2 * WS - 1)];   // this head's column of relative_position_bias_table

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y;
    const int nWw = p.Wp / WS, nW = (p.Hp / WS) * nWw;
    const int bw = blockIdx.x;
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
    }
    for (int i = tid; i < (2 * WS - 1) * (2 * WS - 1); i += ATT_THREADS) tab_s[i] = p.rel_table[head * ((2 * WS - 1) * (2 * WS - 1)) + i];
    __syncthreads();

    // ---- stage K and V of this (window, head) into LDS: 144 rows x 8 float4 each ----
    for (int idx = tid; idx < NTOK * 8; idx += ATT_THREADS) {
        const int t = idx >> 3, c4 = (idx & 7) * 4;
        const int src = src_s[t];
        const float* kp = src >= 0 ? p.qkv + (long)src * C3 + C + head * HD + c4 : p.qkv_bias + C + head * HD + c4;
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kp);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(kp + C);
        *reinterpret_cast<f32x4*>(Ks + t * KV_LD + c4) = kv;
        *reinterpret_cast<f32x4*>(Vs + t * KV_LD + c4) = vv;
    }
    __syncthreads();

    const int li = lane & 15, g = lane >> 4;

    for (int qt = wave; qt < 9; qt += 3) {
        const int qtok = qt * 16 + li;
        const int qsrc = src_s[qtok];
        const int qrid = rid_s[qtok];
        // relative position index (swin.rs:182-184): (qi-ki+11)*23 + (qj-kj+11) = qbase - (key + 11*(key/12))
        const int qbase = (qtok / WS + WS - 1) * (2 * WS - 1) + (qtok % WS) + WS - 1;
        // Q fragment: d = 8g .. 8g+7 of query row qtok, scaled before the product (swin.rs:278)
        float qf[8];
        {
            const float* qp = qsrc >= 0 ? p.qkv + (long)qsrc * C3 + head * HD + g * 8 : p.qkv_bias + head * HD + g * 8;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(qp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qf[e] = q0[e] * p.scale; qf[4 + e] = q1[e] * p.scale; }
        }
        // S^T = K . Q^T : 9 key tiles
        f32x4 st[9];
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
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
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 16 + g * 4 + r;
                float s = st[kt][r] + tab_s[qbase - key - 11 * (key / WS)];
                if (p.shift > 0) s += (rid_s[key] != qrid) ? -100.0f : 0.0f;
                st[kt][r] = s;
                mx = fmaxf(mx, s);
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
        // O^T = V^T . P^T : two d tiles
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 9; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vp = Vs + (kt * 16 + g * 4 + r) * KV_LD + li;
                o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[0], st[kt][r], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16], st[kt][r], o1, 0, 0, 0);
            }
            if (kt % 2 == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // lane holds O^T[d = 16*dt + 4g + r][query = li]; softmax denominator applied here (swin.rs:300,303)
        if (qsrc >= 0) {
            const float inv = 1.0f / sum;
            float* op = p.out + (long)qsrc * C + head * HD + g * 4;
            *reinterpret_cast<f32x4*>(op) = o0 * inv;
            *reinterpret_cast<f32x4*>(op + 16) = o1 * inv;
        }
    }
}

hipError_t launch_window_attention(const WindowAttnParams& p, hipStream_t s) {
    if (p.C != p.heads * HD || p.Hp % WS || p.Wp % WS || p.Hp < p.H || p.Wp < p.W) return hipErrorInvalidValue;
    if (!(p.shift == 0 || p.shift == WS / 2)) return hipErrorInvalidValue;
    const int nW = (p.Hp / WS) * (p.Wp / WS);
    dim3 grid(p.B * nW, p.heads), block(ATT_THREADS);
    hipLaunchKernelGGL(window_attention_f32_kernel, grid, block, 0, s, p);
    return hipGetLastError();
}

}  // namespace brn
