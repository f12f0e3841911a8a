// gemm_bf16.hip — the contraction kernel of compute mode BRN_BF16 (BASELINE configs[2..4]): activations and weights live in
// HBM as bf16, C = epilogue(A_gather[M,K] x W[N,K]^T) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bf16 (or fp32) out.
// Same role as gemm_f32.hip (every candle Linear / Conv2d of the path: swin.rs:98-99,130-131,487; decoder.rs:44-45,65,104,113;
// aspp.rs:39-45,271,282; birefnet.rs:105) and the same fused epilogue (bias / per-image bias / folded eval-BN / ReLU / erf-GELU /
// residual / concat-slice write).
//
// Structure (all waves load AND multiply; no register staging at all):
//   * operands go HBM/L2 -> LDS directly (LDS-DMA, 16 B per lane, 1 KiB per wave instruction), K step 64 (128-byte tile rows), two
//     ring slots, ONE raw s_barrier per K step placed before the step's last sub-step, so that the next step's first fragments
//     and the step-after-next's loads ride on MFMAs (see the K loop);
//   * dense operands and the implicit-GEMM form are buffer-addressed (`buffer_load_dwordx4 v_off, s[rsrc], s_koff offen lds`):
//     a resource per tile, a 32-bit lane offset fixed for the tile, the K advance in an SGPR — no vector arithmetic per piece;
//     conv taps outside the image get an offset past num_records (the buffer unit answers with zeros); only a dense K tail
//     (K % 64 == 32) keeps 64-bit per-lane addresses with a select on the ADDRESS (zero page), never on the data;
//   * LDS image of a tile: rows 2p, 2p+1 share one 256-byte bank row, whose sixteen 16-byte slots are XOR-permuted by
//     (p & 15): slot(r, c) = ((r & 1) << 3 | c) ^ ((r >> 1) & 15).  A ds_read_b128 lane group (16 lanes = 16 different rows,
//     same logical chunk c) then touches 16 different slots: conflict-free.  LDS-DMA writes lane-linear, so the
//     permutation is applied to the per-lane SOURCE offset (which row / chunk a lane fetches), never to the destination;
//   * a wave's instructions are i = w + NW j, so a lane's (row parity, chunk) is the same for all of them: the implicit-GEMM
//     mode tracks ONE (tap, channel) per lane per K step, by increments (no division in the loop).
#include "../brn_kernels.h"
#include "split_planes.h"
#include <type_traits>

// The file is compiled twice (Makefile): as is for compute mode BRN_BF16, and with -DBRN_S16_F16=1 for BRN_F16 — the same kernels with fp16
// as the 16-bit storage / MFMA operand type (3 more mantissa bits at the same bytes and the same matrix rate), in namespace brn::hf.
#ifndef BRN_S16_F16
#define BRN_S16_F16 0
#endif
namespace brn {
#if BRN_S16_F16
namespace hf {
typedef _Float16 s16_t;
#define BRN_MFMA_32X32X16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define BRN_MFMA_16X16X32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#else
typedef __bf16 s16_t;
#define BRN_MFMA_32X32X16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
#define BRN_MFMA_16X16X32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#endif
typedef s16_t s16x8 __attribute__((ext_vector_type(8)));
#if BRN_S16_F16      // (an unqualified call would also find the brn:: function of the same name through its brn::GemmParams argument)
#define BRN_S16_SELF(FN) hf::FN
#else
#define BRN_S16_SELF(FN) FN
#endif

typedef float f32x16_b __attribute__((ext_vector_type(16)));
typedef float f32x4_b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_b __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {      // v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (round to nearest even): lo in bits 0-15
    typedef s16_t s16x2_b __attribute__((ext_vector_type(2)));
    const s16x2_b t = {(s16_t)lo, (s16_t)hi};
    return __builtin_bit_cast(unsigned, t);
}
// the two 16-bit storage values packed in a dword, as fp32
__device__ __forceinline__ float s16_lo_f32(unsigned r) {
#if BRN_S16_F16
    typedef _Float16 h2_b __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2_b, r)[0];
#else
    return __builtin_bit_cast(float, r << 16);
#endif
}
__device__ __forceinline__ float s16_hi_f32(unsigned r) {
#if BRN_S16_F16
    typedef _Float16 h2_b __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2_b, r)[1];
#else
    return __builtin_bit_cast(float, r & 0xffff0000u);
#endif
}

__device__ __attribute__((aligned(16))) unsigned g_zero_page[64];   // 256 zero bytes (code-object global: zero-initialised)

constexpr int BBK_PAD = 64;          // W rows and the K tail are zero-padded to this (the larger of the two K steps built)

__device__ __forceinline__ float gelu_erf_b(float x) {   // same fit as gemm_f32.hip (|error| < 2e-7)
    const float s = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, s, 1.0f));
    float q = -0.29582387555232f;
    q = fmaf(q, t, 1.4920114662361241f);
    q = fmaf(q, t, -2.0596673810742456f);
    q = fmaf(q, t, 2.012361787754068f);
    q = fmaf(q, t, -0.7324354234987704f);
    q = fmaf(q, t, 0.42581723346182204f);
    q = fmaf(q, t, 0.15773620453694617f);
    q = q * t * __expf(-s * s);
    const float one_plus_erf = x < 0.f ? q : 2.0f - q;
    return 0.5f * x * one_plus_erf;
}

// gelu_erf for a result that is rounded to bf16 right away (flavour 0; the weight-stationary kernels).  fc1's epilogue is VALU time the
// persistent workgroup cannot hide behind MFMAs (20 % of an fc1 launch with the round-2 form: erfc by Abramowitz-Stegun 7.1.25, 3 terms =
// 9 VALU + v_rcp + v_exp, |error| < 2.6e-5), so the form is chosen by issue slots.  Round 4:
//     gelu(x) = relu(x) - |x| h(|x|),   h(u) = erfc(u / sqrt 2) / 2 = 2^P(u)
// with P a degree-5 polynomial: log2 of the Gaussian tail is almost a parabola (P(u) ~ -1 - 1.15 u - 0.46 u^2 ...), which one v_exp_f32
// undoes — ONE transcendental instead of two, 1 clamp + 5 fma + v_exp + max + fma = 12 issue slots instead of 17, and a better fit:
// weighted least squares on [0, 8] (weights = the tolerance budget below; coefficients rounded to fp32 and the whole form re-evaluated in
// emulated fp32 over 5e6 points of [-40, 40]): |gelu error| < 3.0e-6 absolute and < 5.2e-5 relative for |gelu| >= 1e-2 — a hundredth of
// half a bf16 ulp (a fifth of half an fp16 ulp: the fp16 build, compute mode BRN_F16, keeps the form).  Beyond u = 8 the clamp holds h at
// 2^-51.9: |x| h is below 1e-7 for every |x| < 1e9.
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
    const float ax = fabsf(x);
    float u, r;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(u) : "v"(x), "v"(8.0f));         // (plain v_min / v_max: fminf / fmaxf put a canonicalising
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(0.0f));           // v_max x, x in front of each; x is an MFMA / fma result, never signalling)
    float p = -0.00040414920658804476f;
    p = fmaf(p, u, 0.006561775226145983f);
    p = fmaf(p, u, -0.050444595515728f);
    p = fmaf(p, u, -0.46150773763656616f);
    p = fmaf(p, u, -1.150171160697937f);
    p = fmaf(p, u, -1.0000925064086914f);
    return fmaf(-ax, __builtin_amdgcn_exp2f(p), r);
}

__device__ __forceinline__ void bf16_tile_coords(int tile, int tilesM, int tilesN, int& tm, int& tn) {
    constexpr int GN = 8;            // N walked in groups of 8 tile columns, M fastest-but-one inside a group (L2 reuse of the W panels)
    const int per_group = tilesM * GN;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gw = min(GN, tilesN - g * GN);
    tm = r / gw;
    tn = g * GN + (r - tm * gw);
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

#ifndef BRN_BF16_SCHED
#define BRN_BF16_SCHED 1
#endif
constexpr bool SCHED = BRN_BF16_SCHED != 0;
#ifndef BRN_BF16_MFMA16
#define BRN_BF16_MFMA16 1
#endif
#ifndef BRN_BF16_BSINGLE
#define BRN_BF16_BSINGLE 1
#endif
#ifndef BRN_BF16_CFG2_M16          // the 256 x 256 tile keeps 32x32x16 with both fragment sets double-buffered: with 16x16x32 it has no
#define BRN_BF16_CFG2_M16 0        // registers left (spills, rematerialised addresses) and measured 7 % slower; the smaller tiles gain 5-13 %
#endif

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, void* lds_dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ f32x4_b zero4b() { f32x4_b z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return s16_lo_f32((unsigned)h); }

// 8 consecutive outputs of one row: everything of the epilogue after the accumulator
template <bool VEC>
__device__ __forceinline__ void store_row8(const GemmParams& p, int m, int n, float (&v)[8], const float (&bias)[8], const float (&sc)[8],
                                           const float (&sh)[8]) {
    if (p.bbias) {
        const float* bp = p.bbias + (long)(m / p.bbias_rows) * p.N + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) if (VEC || n + e < p.N) v[e] += bp[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = v[e] + bias[e];
        if (p.scale) t = t * sc[e] + sh[e];
        if (p.act == ACT_RELU) t = fmaxf(t, 0.f);
        else if (p.act == ACT_GELU_ERF) t = gelu_erf_b(t);
        v[e] = t;
    }
    if (p.R) {
        if (p.r_f32) {
            const float* rp = p.R + (long)m * p.ldr + p.r_coff + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) if (VEC || n + e < p.N) v[e] += rp[e];
        } else {
            const unsigned short* rp = reinterpret_cast<const unsigned short*>(p.R) + (long)m * p.ldr + p.r_coff + n;
            if (VEC) {
                const u32x4_b r = *reinterpret_cast<const u32x4_b*>(rp);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += s16_lo_f32(r[e]);
                    v[2 * e + 1] += s16_hi_f32(r[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += bf16_bits_to_f32(rp[e]);
            }
        }
    }
    if (p.c_f32) {
        float* dst = p.C + (long)m * p.ldc + p.c_coff + n;
        if (VEC) {
            f32x4_b a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4_b*>(dst) = a;
            *reinterpret_cast<f32x4_b*>(dst + 4) = b;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = v[e];
        }
    } else {
        s16_t* dst = reinterpret_cast<s16_t*>(p.C) + (long)m * p.ldc + p.c_coff + n;
        if (VEC) {
            s16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (s16_t)v[e];
            *reinterpret_cast<s16x8*>(dst) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = (s16_t)v[e];
        }
    }
}

// BBK = K step (bf16 elements): 64 (128-byte tile rows, two per 256-byte LDS bank row) or 32 (64-byte rows, four per bank row:
// half the LDS per stage, so more stages and / or more workgroups per CU).  Slot of (row r, 16-byte chunk c) inside its bank row
// p = r / RPB:  ((r % RPB) * CPR + c) ^ (BBK == 64 ? p & 15 : p & 3).
// EPI = epilogue flavour, chosen by the launcher (three small straight-line epilogues instead of one that branches on every flag
// per pass: the all-flags version was 100+ KB of code and cost 9 us per tile):
//   0  bf16 C, bias / per-image bias / folded BN / activation (/ bf16 residual), vector stores (N, ldc, c_coff % 8 == 0): qkv, fc1, the convs
//   1  fp32 C (+ fp32 residual, in place or not): proj, fc2, PatchMerging reduction, the op-level entry points
//   2  everything else (per-image bias, bf16 residual, ragged N, split-K partial sums): per-element, correct, not fast
// PERSISTENT WORKGROUPS: the grid is (CUs x workgroups per CU); a workgroup walks work items id, id + g, ... of its XCD's
// contiguous share.  Between two items the LDS ring is free except for its last slot, where each wave transposes its
// accumulators: the first NSTAGE-1 K steps of the NEXT item are already in flight (into the other slots) while the finished
// item's C rows are stored, and those stores are not waited for before the next K loop starts (counted vmcnt: loads, stores
// and LDS-DMA retire in issue order).  Measured before (one workgroup per tile, tools/gemm_bf16_ablate.py, 40960 x 2304 x 768):
// 49 us of 264 were workgroup launch + prologue, 68 the epilogue, 64 exposed load latency, 83 the MFMA loop itself.
template <int BM, int BN, int WM, int WN, int NSTAGE, int MODE, int BBK, int EPI, bool M16 = BRN_BF16_MFMA16 != 0, bool BSINGLE = BRN_BF16_BSINGLE != 0, bool KT = false>
__global__ void __launch_bounds__(WM* WN * 64) gemm_bf16_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;   // TM x TN: 32 x 32 blocks of a wave tile (epilogue rounds)
    // MFMA shape: 16x16x32 by default — the same cycles per flop as 32x32x16, but the chip holds a higher clock under it
    // (MI355X_MICROARCH.md, DVFS item 7: x 1.12-1.14 in LDS-fed bf16 loops on random data); fragments are FR rows x (BBK / KSUB) k
    constexpr int FR = M16 ? 16 : 32, FM = WTM / FR, FN = WTN / FR;
    constexpr int ROWB = BBK * 2;                               // bytes per tile row
    constexpr int RPB = 256 / ROWB, CPR = BBK / 8;              // tile rows per bank row, 16-byte chunks per tile row
    constexpr int SWZ_MASK = BBK == 64 ? 15 : 3;
    constexpr int RPI = 1024 / ROWB;                            // tile rows per 1-KiB load instruction
    constexpr int LA = BM / RPI / NW, LB = BN / RPI / NW;       // load instructions per wave and stage (A, W)
    static_assert(BBK == 64 || BBK == 32, "K step");
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0 && TM >= 1 && TN >= 1, "tile does not divide over the waves");
    static_assert(MODE == GEMM_DENSE || MODE == GEMM_CONV_NHWC, "register-staged loaders live in gemm_f32.hip");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    constexpr int LPS = LA + LB;                                // vmcnt units per stage and wave
    constexpr int A_BYTES = BM * ROWB, STAGE_BYTES = (BM + BN) * ROWB;
    // Epilogue: the product is formed TRANSPOSED (the W fragment is the MFMA's first operand), so per MFMA block a lane holds ONE row
    // m of C and 4 consecutive columns per 4 registers (16x16: columns 4 (lane >> 4) + {0..3}; 32x32: 8g + 4 (lane >> 5) + {0..3} in
    // registers 4g .. 4g+3) — 16-byte row pieces for ds_write_b128.  One patch per wave (32 rows x EWN fp32, 16-byte chunks XOR-ed
    // with the row: conflict-free both ways) in the LAST ring slot; a wave tile goes through it in TM x NJB rounds.
    constexpr int EWN = (WTN % 64 == 0 && NW * 32 * 64 * 4 <= STAGE_BYTES) ? 64 : 32;
    constexpr int NJB = WTN / EWN;                              // column blocks per wave tile
    static_assert(NW * 32 * EWN * 4 <= STAGE_BYTES, "epilogue patches must fit one ring slot");
    constexpr int LPR = EWN / 8, RPP = 64 / LPR;                // read-back: lanes per row (8 outputs each), rows per pass
    constexpr int PASSES = 32 / RPP;
    constexpr int STORES = TM * NJB * PASSES * (EPI == 1 ? 2 : 1);   // store instructions per wave and FULL tile (flavours 0 / 1)
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef BRN_DIAG_BUILD
    const int abl = p.abl;       // 1: no A loads, 2: no W loads, 4: no fragment reads / MFMA, 8: no epilogue, 16: epilogue without its (bf16) stores
#else
    constexpr int abl = 0;
#endif

    // ---- work distribution: XCD x (= blockIdx % 8 labels the blocks that share an L2) owns a contiguous run of work ids ----
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    const int ntiles = tilesM * tilesN, total = ntiles * p.splitk;
    int id, id_end, id_step;
    {
        const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
        const int cs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        id_end = cs + q + (xcd < r ? 1 : 0);
        id_step = ((int)gridDim.x - xcd + 7) >> 3;               // workgroups of this launch on the same XCD
        id = cs + ((int)blockIdx.x >> 3);
    }
    const int nk_all = (p.K + BBK - 1) / BBK;
    const int kts = (nk_all + p.splitk - 1) / p.splitk;

    // ---- this lane's share of every stage: bank row pr = 4 (w + NW j) + (lane >> 4), slot q' = lane & 15 ----
    const int pr0 = 4 * wave + (lane >> 4);
    const int qs = (lane & 15) ^ (pr0 & SWZ_MASK);              // logical slot: row within the bank row (qs / CPR), 16-byte chunk (qs % CPR)
    const int lrow = RPB * pr0 + qs / CPR;                      // tile row of instruction j = 0; + RPI NW per further instruction
    const int kch = (qs % CPR) * 8;                             // first k (within the K step) of this lane's chunk
    const char* zero = reinterpret_cast<const char*>(g_zero_page);
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    const s16_t* Wb = reinterpret_cast<const s16_t*>(p.Wp);

    // fragment addressing: 32x32x16: lane reads row (lane & 31) of a 32-row block, logical chunk 2 s + (lane >> 5) at k16 step s;
    // 16x16x32: row (lane & 15) of a 16-row block, chunk 4 s + (lane >> 4) at k32 step s.  With 128-byte tile rows a 16-row block is 8
    // bank rows, so odd blocks see the XOR key (p & 15) with bit 3 flipped: their offsets are foff ^ 128.
    constexpr int KSUB = BBK / (M16 ? 32 : 16);
    constexpr int ODD_FLIP = (M16 && BBK == 64) ? 128 : 0;      // (BBK 32: 4 rows per bank row, key p & 3: a 16-row block is 4 bank rows, no flip)
    const int a_base = wm * WTM * ROWB, b_base = A_BYTES + wn * WTN * ROWB;

    // ---- state of the work item whose K steps are being staged ----
    int m0 = 0, n0 = 0, slice = 0, kt0 = 0, nt = 0;
    // Dense operands without a K tail (KT = false: K % 64 == 0, every dense GEMM of the model): an LDS-DMA piece's address is a UNIFORM
    // 64-bit base (tile origin + K step, SGPRs, advanced by scalar adds) + a 32-bit per-lane byte offset fixed for the whole tile
    // (a_voff / w_voff) — no vector arithmetic per piece (the 64-bit per-lane address + zero-page select of the general form was
    // ~8 VALU per piece: with the MFMA issue slots it saturated the SIMD's issue).  Rows >= M read row M - 1 instead of zeros: they
    // only feed accumulator rows that are never stored.  KT = true (K % 64 == 32) and the conv form keep the general address + select.
    constexpr bool FAST = MODE == GEMM_DENSE && !KT;
    long a_off[LA];        // dense: element offset of (row, k = kch); conv: element offset of image b of the row's pixel (+ a_coff)
    unsigned a_voff[LA], w_voff[LB];
    int a_pix[LA];         // conv: byte offset of (image - first image of the tile, iy0, ix0, channel 0) from the tile's buffer base (may be negative in the padding)
    int tap_off = 0;       // conv: byte offset of this lane's (tap, channel) inside a pixel neighbourhood: ((ky dil) Win + kx dil) lda 2 + 2 ci
    int a_iy[LA], a_ix[LA];
    bool a_ok[LA];
    int c_ci = 0, c_ky = 0, c_kx = 0;      // conv: (tap, channel) of this lane's chunk, advanced by BBK channels per K step
    long w_off0 = 0;
    // FAST: buffer resources based at (row m0, k = kt0 * BBK) of A / (row n0, same k) of W: `buffer_load_dwordx4 v_off, s[rsrc], s_koff offen lds`
    __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Wb), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int j = 0; j < LB; ++j) w_voff[j] = (unsigned)(((lrow + RPI * NW * j) * p.wp_ld + kch) * 2);
    auto setup = [&](int work) {
        slice = work / ntiles;
        const int tile = work - slice * ntiles;
        int tile_m, tile_n;
        bf16_tile_coords(tile, tilesM, tilesN, tile_m, tile_n);
        m0 = tile_m * BM; n0 = tile_n * BN;
        kt0 = slice * kts;
        const int nk = min(nk_all, kt0 + kts);
        nt = nk > kt0 ? nk - kt0 : 0;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int m = m0 + lrow + RPI * NW * j;
            a_ok[j] = m < p.M;
            a_iy[j] = 0; a_ix[j] = 0;
            a_voff[j] = 0;
            if (MODE == GEMM_DENSE) {
                a_off[j] = (long)m * p.lda + p.a_coff + kch;
                a_voff[j] = (unsigned)(((min(m, p.M - 1) - m0) * p.lda + kch) * 2);
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                a_iy[j] = oy * p.stride - p.pad;
                a_ix[j] = ox * p.stride - p.pad;
                a_off[j] = (long)b * p.Hin * p.Win * p.lda + p.a_coff;
                const int b0 = min(m0, p.M - 1) / hw;            // first image of the tile: the buffer base (a tile spans at most two images' worth of offsets)
                a_pix[j] = (((a_ok[j] ? b - b0 : 0) * p.Hin + a_iy[j]) * p.Win + a_ix[j]) * p.lda * 2;
            }
        }
        if (MODE == GEMM_CONV_NHWC) {
            const int k = kt0 * BBK + kch;
            int tap;
            if (p.k_chunk_major) {         // K order (64-channel chunk, tap, channel in chunk): the taps of a chunk re-read the same pixels back to back
                const int per = p.kh * p.kw * 64, ch = k / per, rem = k - ch * per;
                tap = rem >> 6;
                c_ci = ch * 64 + (rem & 63);
            } else {
                tap = k / p.Cin;
                c_ci = k - tap * p.Cin;
            }
            c_ky = tap / p.kw; c_kx = tap - c_ky * p.kw;
            tap_off = ((c_ky * p.dil) * p.Win + c_kx * p.dil) * p.lda * 2 + c_ci * 2;
            const int b0 = min(m0, p.M - 1) / (p.Hout * p.Wout);
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)b0 * p.Hin * p.Win * p.lda + p.a_coff), 0, 0x7fffffff, 0x00020000);
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Wb + (long)n0 * p.wp_ld + (long)kt0 * BBK), 0, 0x7fffffff, 0x00020000);
        }
        // W rows n0 + lrow + RPI NW j of the padded [rows][wp_ld] bf16 matrix (rows and K zero-padded to the tile: always in bounds)
        w_off0 = (long)(n0 + lrow) * p.wp_ld + kch;
        if (FAST) {
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)m0 * p.lda + p.a_coff + (long)kt0 * BBK), 0, 0x7fffffff, 0x00020000);
            w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Wb + (long)n0 * p.wp_ld + (long)kt0 * BBK), 0, 0x7fffffff, 0x00020000);
        }
    };
    // piece j of K step t (local index) into ring slot t % NSTAGE: j < LA = A rows, else W rows; stage_advance() after the last piece
    auto stage_piece = [&](int t, int j) {
        char* sbase = smem + (t % NSTAGE) * STAGE_BYTES + wave * 1024;
        const int kbase = (kt0 + t) * BBK;
        if (FAST) {
            const int koff = t * (BBK * 2);                      // uniform: the instruction's SGPR offset
            if (j < LA) { if (!(abl & 1)) blds16(a_rsrc, a_voff[j], koff, sbase + j * (NW * 1024)); }
            else if (!(abl & 2)) blds16(w_rsrc, w_voff[j - LA], koff, sbase + A_BYTES + (j - LA) * (NW * 1024));
        } else if (j < LA) {
            if (MODE == GEMM_DENSE) {
                const bool kin = kbase + kch < p.K;              // K tail (K % 64 == 32): the upper chunks read zeros
                const char* src = (a_ok[j] && kin) ? reinterpret_cast<const char*>(Ab + a_off[j] + kbase) : zero;
                if (!(abl & 1)) glds16(src, sbase + j * (NW * 1024));
            } else {
                // implicit GEMM: buffer-addressed too — the lane's pixel offset is fixed for the tile, its (tap, channel) offset is kept
                // incrementally, and a tap outside the image (or beyond the last tap: K tail) gets an offset past num_records, which
                // the buffer unit answers with zeros (no zero page, no 64-bit address arithmetic)
                const bool kin = p.k_chunk_major ? c_ci < p.Cin : c_ky < p.kh;
                const int iy = a_iy[j] + c_ky * p.dil, ix = a_ix[j] + c_kx * p.dil;
                const bool ok = a_ok[j] && kin && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const unsigned voff = ok ? (unsigned)(a_pix[j] + tap_off) : 0x80000000u;
                if (!(abl & 1)) blds16(a_rsrc, voff, 0, sbase + j * (NW * 1024));
            }
        } else {
            const int jw = j - LA;
            if (MODE == GEMM_CONV_NHWC) { if (!(abl & 2)) blds16(w_rsrc, w_voff[jw], t * (BBK * 2), sbase + A_BYTES + jw * (NW * 1024)); }
            else if (!(abl & 2)) glds16(Wb + w_off0 + (long)(RPI * NW * jw) * p.wp_ld + kbase, sbase + A_BYTES + jw * (NW * 1024));
        }
    };
    auto stage_advance = [&]() {
        if (MODE == GEMM_CONV_NHWC && p.k_chunk_major) {        // next tap of the same 64-channel chunk; after the last tap the next chunk
            if (++c_kx == p.kw) { c_kx = 0; if (++c_ky == p.kh) { c_ky = 0; c_ci += 64; } }
            tap_off = ((c_ky * p.dil) * p.Win + c_kx * p.dil) * p.lda * 2 + c_ci * 2;
        } else if (MODE == GEMM_CONV_NHWC) {
            c_ci += BBK;
            tap_off += BBK * 2;
            if (c_ci >= p.Cin) {
                c_ci -= p.Cin;
                if (++c_kx == p.kw) { c_kx = 0; ++c_ky; }
                tap_off = ((c_ky * p.dil) * p.Win + c_kx * p.dil) * p.lda * 2 + c_ci * 2;
            }
        }
    };
    auto stage = [&](int t) {                                   // all pieces of K step t at once (prologue of a work item)
#pragma unroll
        for (int j = 0; j < LPS; ++j) stage_piece(t, j);
        stage_advance();
    };

    // ---- epilogue constants ----
    const bool has_scale = p.scale != nullptr;
    const int act = p.act;

    bool have = id < id_end;
    if (have) {
        setup(id);
        if (nt > 0) stage(0);
    }
    bool counted = false;        // the previous item's epilogue issued exactly STORES stores per wave after this item's first loads
    while (have) {
        typename std::conditional<M16, f32x4_b, f32x16_b>::type acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

        // ---- K loop.  Two ring slots; the workgroup barrier of a K step sits BEFORE its last sub-step:
        //   sub-steps 0 .. KSUB-2 of step t : MFMAs, each followed by one fragment read of the next sub-step (second register set);
        //   mid-step                        : s_waitcnt (step t+1 has landed: its LDS-DMA pieces went out one whole K step ago), s_barrier
        //                                     — every wave has now READ all its fragments of step t, so slot t % 2 is free;
        //   last sub-step of step t         : MFMAs from registers, each followed by one fragment read of step t+1's FIRST sub-step
        //                                     (other slot) and one LDS-DMA piece of step t+2 (into the slot just freed).
        // So no MFMA ever waits for an LDS read right behind a barrier (the first version read its first fragments after the barrier:
        // ~300 idle cycles of 2300 per step), and a K step's loads have a full K step to land (they had between a quarter and one).
        // The issue order of every sub-step is pinned with sched_group_barrier: left to itself the scheduler issued all LDS-DMA
        // pieces in one burst (60-180 issue cycles each, every wave of the CU at once) and kept ONE fragment register set.  Source
        // order fixes the order of the LDS traffic (the compiler keeps ds_read / LDS-DMA order: it cannot prove them disjoint).
        // Fragment offsets are recomputed per K step from an opaque copy of the lane id: live across the persistent loop they were
        // what the register allocator spilled.
        static_assert(NSTAGE == 2 && KSUB >= 2 && KSUB % 2 == 0, "two ring slots, an even number of sub-steps per K step");
        constexpr int NF = FM + FN, NM = FM * FN;                  // fragment reads and MFMAs per sub-step (k16 / k32)
        static_assert(NM >= LPS && NM >= NF, "a sub-step has fewer MFMAs than LDS operations to place between them");
        s16x8 af[2][FM], bf[2][FN];
        // fragment f of sub-step s from ring slot `sb`: A blocks 0 .. FM-1, W blocks FM .. NF-1
        auto frag_offsets = [&](int (&foff)[KSUB]) {
            int flane = lane;
            asm volatile("" : "+v"(flane));
            const int frow = flane & (FR - 1), fh = flane / FR;
            const int fswz = (frow / RPB) & SWZ_MASK;
#pragma unroll
            for (int q = 0; q < KSUB; ++q) foff[q] = (frow / RPB) * 256 + ((((frow % RPB) * CPR + ((M16 ? 4 : 2) * q + fh)) ^ fswz) << 4);
        };
        auto read_a = [&](const char* sb, const int (&foff)[KSUB], int sub, int f) {
            af[sub & 1][f] = *reinterpret_cast<const s16x8*>(sb + a_base + f * (FR * ROWB) + (foff[sub] ^ ((f & 1) ? ODD_FLIP : 0)));
        };
        auto read_b = [&](const char* sb, const int (&foff)[KSUB], int sub, int f) {
            bf[sub & 1][f] = *reinterpret_cast<const s16x8*>(sb + b_base + f * (FR * ROWB) + (foff[sub] ^ ((f & 1) ? ODD_FLIP : 0)));
        };
        // one sub-step: NM MFMAs on register set S & 1; after MFMA m, in this order: the A fragment m (m < FM) and the W fragment of the
        // next sub-step (BSINGLE: W block m / FM right after its last use, MFMA order W block outer; else W fragment m - FM), then
        // LDS-DMA piece m of K step `tstage` (STAGE only)
        auto substep = [&](auto sc, auto next_c, auto stage_c, const char* sbn, const int (&foff)[KSUB], int nsub, int tstage) {
            constexpr int S = decltype(sc)::value;
            constexpr bool NEXT = decltype(next_c)::value, STAGE = decltype(stage_c)::value;
            static_for<0, NM>([&](auto mc) {
                constexpr int Mi = decltype(mc)::value;
                if (NEXT && Mi < FM) read_a(sbn, foff, nsub, Mi);
                if (NEXT && BSINGLE && Mi % FM == FM - 1) read_b(sbn, foff, nsub, Mi / FM);
                if (NEXT && !BSINGLE && Mi >= FM && Mi < NF) read_b(sbn, foff, nsub, Mi - FM);
                if (STAGE && Mi < LPS) stage_piece(tstage, Mi);
            });
            static_for<0, NM>([&](auto mc) {                      // transposed product: W fragment first
                constexpr int Mi = decltype(mc)::value;
                constexpr int i = BSINGLE ? Mi % FM : Mi / FN, j = BSINGLE ? Mi / FM : Mi % FN;
                if constexpr (M16) acc[i][j] = BRN_MFMA_16X16X32(bf[S & 1][j], af[S & 1][i], acc[i][j]);
                else acc[i][j] = BRN_MFMA_32X32X16(bf[S & 1][j], af[S & 1][i], acc[i][j]);
            });
            if (STAGE) stage_advance();
            if (SCHED) {
                static_for<0, NM>([&](auto mc) {
                    constexpr int Mi = decltype(mc)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (NEXT && Mi < FM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (NEXT && BSINGLE && Mi % FM == FM - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (NEXT && !BSINGLE && Mi >= FM && Mi < NF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (STAGE && Mi < LPS) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                });
            }
        };
        // K step t: has_next = step t+1 exists (mid-step barrier + its first fragments), has_next2 = step t+2 exists (its loads go out)
        auto kstep = [&](int t, auto has_next, auto has_next2) {
            constexpr bool HN = decltype(has_next)::value, HN2 = decltype(has_next2)::value;
            const char* sb = smem + (t & 1) * STAGE_BYTES;
            const char* sbn = smem + ((t + 1) & 1) * STAGE_BYTES;
            int foff[KSUB];
            frag_offsets(foff);
            static_for<0, KSUB - 1>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                substep(sc, std::true_type{}, std::false_type{}, sb, foff, S + 1, 0);
            });
            if (HN) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // step t+1 landed (this wave's pieces); this wave's reads of slot t & 1 done
                __builtin_amdgcn_s_barrier();
            }
            substep(std::integral_constant<int, KSUB - 1>{}, has_next, has_next2, sbn, foff, 0, t + 2);
        };
        if (abl & 4) {                                                // diag: loads only
            for (int t = 0; t < nt; ++t) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (t + 1 < nt) stage(t + 1);
            }
        } else if (nt > 0) {
            // tile prologue: step 0 has landed (at most the previous item's un-waited stores are younger), every wave is out of the previous
            // item's epilogue (its patches live in slot 1); step 1 goes out, the first fragments come in — the one exposed LDS read per tile
            constexpr int VM0 = STORES > 63 ? 63 : STORES;           // (the counter has 6 bits; a smaller N only waits longer)
            if (counted) wait_vmcnt<VM0>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (nt > 1) stage(1);
            {
                int foff[KSUB];
                frag_offsets(foff);
#pragma unroll
                for (int f = 0; f < FM; ++f) read_a(smem, foff, 0, f);
#pragma unroll
                for (int f = 0; f < FN; ++f) read_b(smem, foff, 0, f);
            }
            int t = 0;
            for (; t + 2 < nt; ++t) kstep(t, std::true_type{}, std::true_type{});
            if (t + 1 < nt) { kstep(t, std::true_type{}, std::false_type{}); ++t; }
            kstep(t, std::false_type{}, std::false_type{});
        }

        // ---- hand over: the finished item keeps (e_m0, e_n0, e_slice); the staging state moves on to the next item ----
        const int e_m0 = m0, e_n0 = n0, e_slice = slice;
        id += id_step;
        have = id < id_end;
        __builtin_amdgcn_s_barrier();                            // all fragment reads of the finished item are done: the ring is free
        if (have) {
            setup(id);
            if (nt > 0) stage(0);                                // next item's first K step: in flight during the epilogue below
        }
        if (abl & 8) { if (acc[0][0][0] == 123.456f) p.C[0] = 1.f; counted = false; continue; }

        {
            // ---- epilogue, flavours 0 / 1.  The product was formed transposed: per 32 x 32 block a lane holds row (lane & 31), columns
            // 8g + 4h + {0..3} in registers 4g .. 4g+3 (h = lane >> 5) = whole 16-byte pieces of a row.  A (row block i, column block jb)
            // round puts the wave's 32 x EWN fp32 block into its LDS patch with ds_write_b128 (16-byte chunks XOR-ed with the row:
            // conflict-free both ways) and reads it back as row segments of 8 columns per lane: a lane's columns are then the same for
            // the whole column block (bias / BN constants in registers), rows are stored as whole 128/256-byte segments.  The code is
            // specialised (activation, BN, full tile) so that a pass is ~20 instructions: the first version of this epilogue tested
            // every flag per pass and recomputed 64-bit addresses (~1900 instructions per wave and tile = 7 us on a 256 x 256 tile). ----
            constexpr int CM = EWN / 4 - 1;                            // chunk mask of a patch row
            // the lane id goes through an opaque asm once per tile: everything the epilogue derives from it (patch addresses, row /
            // column offsets) is recomputed here instead of being hoisted out of the persistent loop into registers that would
            // then be live across the K loop (which has none to spare: the hoisted version spilled)
            int elane = lane;
            asm volatile("" : "+v"(elane));
            const int wrow = elane & (FR - 1), wt = (elane / FR) ^ (wrow & CM);   // write map: patch row (+ 16 ii), chunk key
            const int er = elane / LPR, ec = (elane % LPR) * 8;       // read-back map: row er of a pass, columns ec .. ec + 7
            char* wpatch = smem + (NSTAGE - 1) * STAGE_BYTES + wave * (32 * EWN * 4);
            char* wbase = wpatch + wrow * (EWN * 4);
            const int esz = EPI == 0 ? 2 : 4;
            const long row0 = e_m0 + wm * WTM;                         // first row of the wave tile
            const bool full = e_m0 + BM <= p.M && e_n0 + BN <= p.N;
            auto run = [&](auto act_c, auto scale_c, auto full_c) {
                constexpr int ACT = decltype(act_c)::value;
                constexpr bool SCALE = decltype(scale_c)::value, FULL = decltype(full_c)::value;
                // rounds r = jb * TM + i; flavour 1 keeps ONE round of fp32 residual rows in registers: the rows of pass ps of round r + 1
                // are requested as soon as pass ps of round r has used its registers (the K loop has no registers to spare for more)
                f32x4_b rres[PASSES][2];
                auto load_res = [&](int r, int ps) {
                    const int jb = r / TM, i = r % TM;
                    const int n = e_n0 + wn * WTN + jb * EWN + ec;
                    const long mb = row0 + i * 32;
                    const bool ok = FULL || (mb + ps * RPP + er < p.M && n < p.N);
                    const char* rp = reinterpret_cast<const char*>(p.R) + (mb * p.ldr + p.r_coff + (n - ec)) * 4 + (long)(ps * RPP) * p.ldr * 4 +
                                     (unsigned)(er * p.ldr + ec) * 4;
                    if (!ok) rp = reinterpret_cast<const char*>(g_zero_page);
                    rres[ps][0] = *reinterpret_cast<const f32x4_b*>(rp);
                    rres[ps][1] = *reinterpret_cast<const f32x4_b*>(rp + 16);
                };
                const bool has_res1 = EPI == 1 && p.R != nullptr;
                if (has_res1) {
#pragma unroll
                    for (int ps = 0; ps < PASSES; ++ps) load_res(0, ps);
                }
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) {
                    const int n = e_n0 + wn * WTN + jb * EWN + ec;   // this lane's 8 columns after the read-back
                    const bool nin = FULL || n < p.N;
                    const int nl = nin ? n : 0;
                    f32x4_b bias0 = zero4b(), bias1 = zero4b(), sc0, sc1, sh0, sh1;
                    if (EPI != 2 && p.bias) { bias0 = *reinterpret_cast<const f32x4_b*>(p.bias + nl); bias1 = *reinterpret_cast<const f32x4_b*>(p.bias + nl + 4); }
                    if (EPI != 2 && SCALE) {
                        sc0 = *reinterpret_cast<const f32x4_b*>(p.scale + nl); sc1 = *reinterpret_cast<const f32x4_b*>(p.scale + nl + 4);
                        sh0 = *reinterpret_cast<const f32x4_b*>(p.shift + nl); sh1 = *reinterpret_cast<const f32x4_b*>(p.shift + nl + 4);
                    }
                    // per-lane byte offsets inside a pass (32-bit) on top of uniform row pointers
                    const unsigned coff = (unsigned)(er * p.ldc + ec) * esz, roff = (unsigned)(er * p.ldr + ec) * 2;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int r = jb * TM + i;
                        const long mb = row0 + i * 32;                // first row of this round
                        char* cu = reinterpret_cast<char*>(p.C) + (mb * p.ldc + p.c_coff + (n - ec)) * esz;            // uniform
                        const char* ru = reinterpret_cast<const char*>(p.R) + (mb * p.ldr + p.r_coff + (n - ec)) * 2;   // flavour 0: bf16 residual
                        if constexpr (M16) {
#pragma unroll
                            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                                for (int jj = 0; jj < EWN / 16; ++jj)
                                    *reinterpret_cast<f32x4_b*>(wbase + ii * (16 * EWN * 4) + (((jj * 4) ^ wt) << 4)) = acc[2 * i + ii][jb * (EWN / 16) + jj];
                        } else {
#pragma unroll
                            for (int jj = 0; jj < EWN / 32; ++jj)
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const auto& c = acc[i][jb * (EWN / 32) + jj];
                                    const f32x4_b v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
                                    *reinterpret_cast<f32x4_b*>(wbase + (((jj * 8 + 2 * g) ^ wt) << 4)) = v;
                                }
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int ps = 0; ps < PASSES; ++ps) {
                            const int row = ps * RPP + er;
                            const int sw = row & CM;
                            const char* rb = wpatch + row * (EWN * 4);
                            const f32x4_b v0 = *reinterpret_cast<const f32x4_b*>(rb + ((((ec >> 2)) ^ sw) << 4));
                            const f32x4_b v1 = *reinterpret_cast<const f32x4_b*>(rb + ((((ec >> 2) + 1) ^ sw) << 4));
                            const bool ok = FULL || (mb + row < p.M && nin);
                            if constexpr (EPI == 2) {                // everything else: per element, correct, not fast
                                if (ok) {
                                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                                    if (p.splitk > 1) {
                                        float* dst = p.part + (long)e_slice * p.M * p.N + (mb + row) * p.N + n;
#pragma unroll
                                        for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = v[e];
                                    } else {
                                        float bias[8], sc[8], sh[8];
#pragma unroll
                                        for (int e = 0; e < 8; ++e) {
                                            bias[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f;
                                            if (n + e < p.N) {
                                                if (p.bias) bias[e] = p.bias[n + e];
                                                if (has_scale) { sc[e] = p.scale[n + e]; sh[e] = p.shift[n + e]; }
                                            }
                                        }
                                        store_row8<false>(p, (int)(mb + row), n, v, bias, sc, sh);
                                    }
                                }
                                continue;
                            }
                            f32x4_b a = v0 + bias0, b = v1 + bias1;
                            if (EPI == 0 && p.bbias) {               // per-image bias (the pooled ASPP branch folded into conv1)
                                const long m = ok ? mb + row : 0;
                                const float* bp = p.bbias + (m / p.bbias_rows) * p.N + nl;
                                a = a + *reinterpret_cast<const f32x4_b*>(bp); b = b + *reinterpret_cast<const f32x4_b*>(bp + 4);
                            }
                            if (SCALE) { a = a * sc0 + sh0; b = b * sc1 + sh1; }
                            if (ACT == ACT_RELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { a[e] = fmaxf(a[e], 0.f); b[e] = fmaxf(b[e], 0.f); }
                            } else if (ACT == ACT_GELU_ERF) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    a[e] = EPI == 0 ? gelu_erf_bf16out(a[e]) : gelu_erf_b(a[e]);
                                    b[e] = EPI == 0 ? gelu_erf_bf16out(b[e]) : gelu_erf_b(b[e]);
                                }
                            }
                            char* cp = cu + (long)(ps * RPP) * p.ldc * esz + coff;
                            if (EPI == 0) {
                                if (p.R) {                           // bf16 residual (the decoder's lateral adds, in place)
                                    const char* rp = ru + (long)(ps * RPP) * p.ldr * 2 + roff;
                                    const u32x4_b rr = *reinterpret_cast<const u32x4_b*>(ok ? rp : reinterpret_cast<const char*>(g_zero_page));
                                    a[0] += s16_lo_f32(rr[0]); a[1] += s16_hi_f32(rr[0]);
                                    a[2] += s16_lo_f32(rr[1]); a[3] += s16_hi_f32(rr[1]);
                                    b[0] += s16_lo_f32(rr[2]); b[1] += s16_hi_f32(rr[2]);
                                    b[2] += s16_lo_f32(rr[3]); b[3] += s16_hi_f32(rr[3]);
                                }
                                const u32x4_b o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
                                if (!(abl & 16) || a[0] == 123.456f)  // diag bit 16: the epilogue without its stores
                                if (ok) *reinterpret_cast<u32x4_b*>(cp) = o;
                            } else {
                                if (has_res1) { a = a + rres[ps][0]; b = b + rres[ps][1]; }
                                if (ok) { *reinterpret_cast<f32x4_b*>(cp) = a; *reinterpret_cast<f32x4_b*>(cp + 16) = b; }
                                if (has_res1 && r + 1 < NJB * TM) load_res(r + 1, ps);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();              // the patch is rewritten by the next round
                    }
                }
            };
            auto run_full = [&](auto act_c, auto scale_c) {
                if (full) run(act_c, scale_c, std::true_type{}); else run(act_c, scale_c, std::false_type{});
            };
            using I0 = std::integral_constant<int, ACT_NONE>; using I1 = std::integral_constant<int, ACT_RELU>; using I2 = std::integral_constant<int, ACT_GELU_ERF>;
            if constexpr (EPI == 2) run(I0{}, std::false_type{}, std::false_type{});
            else if (has_scale) {
                if (act == ACT_RELU) run_full(I1{}, std::true_type{});
                else if (act == ACT_GELU_ERF) run_full(I2{}, std::true_type{});
                else run_full(I0{}, std::true_type{});
            } else {
                if (act == ACT_RELU) run_full(I1{}, std::false_type{});
                else if (act == ACT_GELU_ERF) run_full(I2{}, std::false_type{});
                else run_full(I0{}, std::false_type{});
            }
        }
        // every wave of a FULL tile issued exactly STORES store instructions after the next item's first loads
        counted = EPI < 2 && e_m0 + BM <= p.M && e_n0 + BN <= p.N;
    }
}


// =====================================================================================================================
// gemm_wstat_bf16_kernel — dense C = act(A W^T + b) for SHORT K (K = 64 KS: 192, or 384 with 32-row tiles) and wide A (M >> N): the stage-0 /
// stage-1 qkv and fc1 GEMMs of the Swin backbone at batch >= 4 (swin.rs:98,130; M = 655 360, K = 192, N = 576 / 768 at batch 8).  With three K steps per tile the
// persistent 256 x 256 kernel above is prologue / epilogue all the way (2.0 - 2.6 TB/s of algorithmic traffic); here the weights never
// move: a wave keeps its 48 columns of W (K x 48 bf16 = 72 VGPRs, loaded once, stored in MFMA fragment order by attach_dense_frags) for
// the whole launch and the workgroup streams 64-row tiles of A through a two-buffer LDS ring (LDS-DMA, XOR-swizzled 128-byte rows as
// above); what is left per tile is 24 KB in, 96 MFMAs per wave, 24 KB out.  A workgroup = 4 waves = 192 columns; the N / 192 column
// groups of a row tile are neighbouring workgroups of one XCD walking the same tiles (the A tile is read from HBM once, then from L2).
// The C tile goes back through the A buffer it came from (two 32-row halves, 512-byte rows, XOR-ed chunks) and leaves as 384-byte
// row segments.
// =====================================================================================================================
constexpr int WSTAT_WG_PER_CU = 2;      // 2: <= 256 VGPRs, no spill (forced to 168 for three per CU the kernel spills 30-40 registers)
// CUs the persistent grids of this file are sized for: 256, or the size of the CU mask of the stream being launched on (a sub-batch
// stream confined to a part of the chip, brn_api.cpp run_model); thread-local: set by the host thread that enqueues the forward
#if !BRN_S16_F16     // (one state for both builds: brn::launch_cus)
static thread_local int g_launch_cus = 256;
int launch_cus() { return g_launch_cus; }
void set_launch_cus(int n) { g_launch_cus = n < 8 ? 8 : (n > 256 ? 256 : n) / 8 * 8; }
#endif

__device__ __forceinline__ int ws_slot(int r, int c) { return (r >> 1) * 256 + (((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) << 4); }

// WBM = rows per A tile: 64 at K = 192 (two 24-KB buffers), 32 at K = 384 (the W slice is 144 VGPRs there; two 24-KB buffers again,
// so two workgroups still share a CU and one's epilogue overlaps the other's MFMAs)
template <int KS, int ACT, int WBM = 64>
__global__ void __launch_bounds__(256, 2) gemm_wstat_bf16_kernel(const GemmParams p) {
    constexpr int WBN = 192, SUB = WBM * 128;                            // columns per workgroup, bytes of one K step's sub-tile
    constexpr int K32 = KS * 2, ABUF = KS * SUB;
    constexpr int NJ = WBM / 32, RF = WBM / 16;                          // LDS-DMA instructions per wave and K step, 16-row fragments per tile
    static_assert(WBM == 64 || WBM == 32, "tile rows");
    static_assert(ABUF >= 32 * 512, "a 32-row half of the C tile (512-byte rows) must fit the A buffer it replaces");
    __shared__ __attribute__((aligned(1024))) char smem[2 * ABUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = p.N / WBN, T = (p.M + WBM - 1) / WBM;
    // workgroups of one XCD (same blockIdx % 8): local index li -> column group li % G, walker li / G of nw; XCD x owns a contiguous run of row tiles
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, nw = ((int)gridDim.x >> 3) / G;
    const int grp = li % G, wk = li / G;
    const int t_lo = (int)((long)T * xcd / 8), t_hi = (int)((long)T * (xcd + 1) / 8);
    if (wk >= nw) return;
    const int n0 = grp * WBN + wave * 48;                               // this wave's first column
    // ---- W fragments, resident for the whole launch ----
    s16x8 wfr[3][K32];
    {
        const char* wf = reinterpret_cast<const char*>(p.Wp) + ((long)(n0 >> 4) * K32 * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int ks = 0; ks < K32; ++ks) wfr[j][ks] = *reinterpret_cast<const s16x8*>(wf + (long)(j * K32 + ks) * 1024);
    }
    f32x4_b bias[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) bias[j] = p.bias ? *reinterpret_cast<const f32x4_b*>(p.bias + n0 + 16 * j + 4 * (lane >> 4)) : zero4b();
    // ---- this lane's share of an A tile: LDS-DMA instruction ii = wave + 4 j (j < NJ) of every sub-tile fills bank rows 4 ii .. 4 ii + 3 ----
    unsigned a_voff[NJ];
    int a_row[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int pr = 4 * (wave + 4 * j) + (lane >> 4), qs = (lane & 15) ^ (pr & 15);
        a_row[j] = 2 * pr + (qs >> 3);
        a_voff[j] = (unsigned)((qs & 7) * 16);                          // byte offset inside the 128-byte K-step piece of the row
    }
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    auto issue = [&](int t, char* buf) {
        const int m0 = t * WBM;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)m0 * p.lda + p.a_coff), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned ro = (unsigned)((min(m0 + a_row[j], p.M - 1) - m0) * p.lda * 2) + a_voff[j];   // rows >= M re-read row M - 1 (never stored)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) blds16(rs, ro, ks * 128, buf + ks * SUB + (wave + 4 * j) * 1024);
        }
    };
    int a_foff[RF][2];
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) a_foff[i][s2] = ws_slot(16 * i + (lane & 15), 4 * s2 + (lane >> 4));

    int t = t_lo + wk;
    if (t < t_hi) issue(t, smem);
    for (int it = 0; t < t_hi; ++it, t += nw) {
        char* buf = smem + (it & 1) * ABUF;
        const bool more = t + nw < t_hi;
        if (more) { issue(t + nw, smem + ((it + 1) & 1) * ABUF); wait_vmcnt<NJ * KS>(); }   // this tile (and the previous tile's stores) landed; the next one may be in flight
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        f32x4_b acc[RF][3];
#pragma unroll
        for (int i = 0; i < RF; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = zero4b();
#pragma unroll
        for (int ks = 0; ks < K32; ++ks) {
            s16x8 af[RF];
#pragma unroll
            for (int i = 0; i < RF; ++i) af[i] = *reinterpret_cast<const s16x8*>(buf + (ks >> 1) * SUB + a_foff[i][ks & 1]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < RF; ++i) acc[i][j] = BRN_MFMA_16X16X32(wfr[j][ks], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave has read its fragments: the buffer now takes the C tile
        const int m0 = t * WBM;
        s16_t* Cb = reinterpret_cast<s16_t*>(p.C);
#pragma unroll
        for (int half = 0; half < WBM / 32; ++half) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * half + ii, row = 16 * ii + (lane & 15);          // row within the 32-row half
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    f32x4_b v = acc[i][j] + bias[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ACT == ACT_GELU_ERF) v[e] = gelu_erf_bf16out(v[e]);
                        else if (ACT == ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                    }
                    const int col = wave * 48 + 16 * j + 4 * (lane >> 4);           // column within the workgroup's 192
                    const u32x2_b o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    *reinterpret_cast<u32x2_b*>(buf + row * 512 + (((col >> 3) ^ (row & 31)) << 4) + (col & 4) * 2) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < 3; ++q) {                                // 32 rows x 24 chunks of 16 bytes over 256 threads
                const int idx = tid + 256 * q, row = idx / 24, ch = idx - row * 24;
                const u32x4_b v = *reinterpret_cast<const u32x4_b*>(buf + row * 512 + ((ch ^ (row & 31)) << 4));
                const int m = m0 + half * 32 + row;
                if (m < p.M) *reinterpret_cast<u32x4_b*>(Cb + (long)m * p.ldc + p.c_coff + grp * WBN + ch * 8) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // the half (and, after the second one, the buffer) is free again
        }
    }
}

// =====================================================================================================================
// gemm_wstat_ln_bf16_kernel — the attention output projection of a C = 192 stage with the block's second LayerNorm in its epilogue
// (swin.rs:310,406,407): x = A Wᵀ + bias + x (fp32 residual stream, in place) and y = LayerNorm(x) γ + β (bf16, the fc1 operand).
// N = K = 192: a workgroup of the weight-stationary kernel above owns WHOLE rows (4 waves x 48 columns), so the row statistics are
// local.  The C tile goes through the A buffer as fp32 (32 rows x 768 B = the 24-KB buffer, 16-byte chunks XOR-ed with the row); on the
// way back 8 lanes share a row (6 chunks each): residual add, store of x, two-pass mean / biased variance (the arithmetic of
// layernorm_kernel) over the 8 lanes by DPP shuffles, store of y.  What it saves is the stand-alone LayerNorm's read of x.
// =====================================================================================================================
__global__ void __launch_bounds__(256, 2) gemm_wstat_ln_bf16_kernel(const GemmParams p, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     const float eps, s16_t* __restrict__ Y, const int ldy) {
    constexpr int KS = 3, WBM = 64, WBN = 192, SUB = WBM * 128;
    constexpr int K32 = KS * 2, ABUF = KS * SUB;
    static_assert(ABUF == 32 * WBN * 4, "a 32-row half of the fp32 C tile is exactly one A buffer");
    __shared__ __attribute__((aligned(1024))) char smem[2 * ABUF];
    __shared__ __attribute__((aligned(16))) float gb_s[2 * WBN];         // gamma | beta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (p.M + WBM - 1) / WBM;
    const int xcd = blockIdx.x & 7, wk = blockIdx.x >> 3, nw = (int)gridDim.x >> 3;
    const int t_lo = (int)((long)T * xcd / 8), t_hi = (int)((long)T * (xcd + 1) / 8);
    if (tid < WBN) { gb_s[tid] = gamma[tid]; gb_s[WBN + tid] = beta[tid]; }
    const int n0 = wave * 48;
    s16x8 wfr[3][K32];
    {
        const char* wf = reinterpret_cast<const char*>(p.Wp) + ((long)(n0 >> 4) * K32 * 64 + lane) * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int ks = 0; ks < K32; ++ks) wfr[j][ks] = *reinterpret_cast<const s16x8*>(wf + (long)(j * K32 + ks) * 1024);
    }
    f32x4_b bias[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) bias[j] = p.bias ? *reinterpret_cast<const f32x4_b*>(p.bias + n0 + 16 * j + 4 * (lane >> 4)) : zero4b();
    unsigned a_voff[2];
    int a_row[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pr = 4 * (wave + 4 * j) + (lane >> 4), qs = (lane & 15) ^ (pr & 15);
        a_row[j] = 2 * pr + (qs >> 3);
        a_voff[j] = (unsigned)((qs & 7) * 16);
    }
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    auto issue = [&](int t, char* buf) {
        const int m0 = t * WBM;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)m0 * p.lda + p.a_coff), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned ro = (unsigned)((min(m0 + a_row[j], p.M - 1) - m0) * p.lda * 2) + a_voff[j];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) blds16(rs, ro, ks * 128, buf + ks * SUB + (wave + 4 * j) * 1024);
        }
    };
    int a_foff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) a_foff[i][s2] = ws_slot(16 * i + (lane & 15), 4 * s2 + (lane >> 4));
    const int er = tid >> 3, ec = tid & 7;                               // read-back: row of the 32-row half, first chunk (then + 8 k)
    float* Cf = reinterpret_cast<float*>(p.C);
    const float* Rf = reinterpret_cast<const float*>(p.R);

    int t = t_lo + wk;
    if (t < t_hi) issue(t, smem);
    for (int it = 0; t < t_hi; ++it, t += nw) {
        char* buf = smem + (it & 1) * ABUF;
        const bool more = t + nw < t_hi;
        if (more) { issue(t + nw, smem + ((it + 1) & 1) * ABUF); wait_vmcnt<2 * KS>(); }
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        f32x4_b acc[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = zero4b();
#pragma unroll
        for (int ks = 0; ks < K32; ++ks) {
            s16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const s16x8*>(buf + (ks >> 1) * SUB + a_foff[i][ks & 1]);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = BRN_MFMA_16X16X32(wfr[j][ks], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                   // every wave has read its fragments: the buffer now takes the C tile
        const int m0 = t * WBM;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m = m0 + half * 32 + er, mc = min(m, p.M - 1);
            f32x4_b rr[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) rr[k] = *reinterpret_cast<const f32x4_b*>(Rf + (long)mc * p.ldr + p.r_coff + (ec + 8 * k) * 4);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * half + ii, row = 16 * ii + (lane & 15);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int ch = (wave * 48 + 16 * j + 4 * (lane >> 4)) >> 2;      // 16-byte chunk of the 768-byte row
                    *reinterpret_cast<f32x4_b*>(buf + row * 768 + ((ch ^ (row & 7)) << 4)) = acc[i][j] + bias[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4_b xv[6];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int ch = ec + 8 * k;
                xv[k] = *reinterpret_cast<const f32x4_b*>(buf + er * 768 + ((ch ^ (er & 7)) << 4)) + rr[k];
                sum += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            }
            if (m < p.M) {
#pragma unroll
                for (int k = 0; k < 6; ++k) *reinterpret_cast<f32x4_b*>(Cf + (long)m * p.ldc + p.c_coff + (ec + 8 * k) * 4) = xv[k];
            }
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
            const float mean = sum / (float)WBN;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                xv[k] = xv[k] - mean;
                sq += (xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1]) + (xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3]);
            }
            sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4);
            const float rstd = 1.0f / sqrtf(sq / (float)WBN + eps);
            if (m < p.M) {
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int c0 = (ec + 8 * k) * 4;
                    const f32x4_b gm = *reinterpret_cast<const f32x4_b*>(gb_s + c0), bt = *reinterpret_cast<const f32x4_b*>(gb_s + WBN + c0);
                    const f32x4_b o = xv[k] * rstd * gm + bt;
                    const u32x2_b ob = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *reinterpret_cast<u32x2_b*>(Y + (long)m * ldy + c0) = ob;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // the half (and, after the second one, the buffer) is free again
        }
    }
}
// =====================================================================================================================
// gemm_rowln_bf16_kernel<NC> (round 4) — the row-owning projection for the WIDE stages (NC = 768: stage 2, 18 of the 24 blocks; NC = 384:
// stage 1): x = A W^T + bias + x on the fp32 residual stream (in place) AND y = LayerNorm(x) gamma + beta as the bf16 operand of the next
// GEMM, in one launch (swin.rs:310,406,407) — the stand-alone LayerNorm launch and its fp32 re-read of x are gone.
// A workgroup of 8 waves owns 64 WHOLE rows (64 x NC fp32 accumulators = 96 / 48 VGPRs per lane; wave w = columns [w NC/8, (w+1) NC/8)),
// so the row statistics are local.  W does not fit anything but L2, so it streams through LDS: K step 32 (64-byte tile rows, four per
// 256-byte bank row, 16-byte chunks XOR-ed with the bank row as in gemm_bf16_kernel<BBK = 32>), three ring slots of (64 + NC) x 64 B, the
// pieces of K step t + 2 issued right after the barrier of step t (counted vmcnt; waves 0-3 carry the A pieces).  Per 64 rows the
// workgroup takes in the whole W (1.18 MB at NC = 768): the K loop is bound by the CU's L2 -> LDS intake, not by the matrix pipe, and the
// launch as a whole by HBM (A once, x read + written once, y once) — which is the point: 378 MB per stage-2 launch instead of the 504 MB of
// projection + LayerNorm.  Epilogue: the C tile leaves through the ring as fp32 in two 32-row halves (16-byte chunks XOR-ed with the row),
// comes back with 16 lanes per row (whole 256-byte segments), + residual, store x, two-pass mean / biased variance (the arithmetic of
// layernorm_kernel), store y.  The next tile's first K step is already in flight (slot 2) during the epilogue.
// =====================================================================================================================
template <int NC>
__global__ void __launch_bounds__(512) gemm_rowln_bf16_kernel(const GemmParams p, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float eps, s16_t* __restrict__ Y, const int ldy) {
    constexpr int RBM = 64, RBK = 32, ROWB = RBK * 2;                    // 64-byte tile rows
    constexpr int WCOL = NC / 8;                                         // columns per wave: 96 / 48
    constexpr int FN = WCOL / 16, FM = RBM / 16;                         // 16 x 16 blocks of a wave tile
    constexpr int A_BYTES = RBM * ROWB, SLOT = (RBM + NC) * ROWB;        // 4 KB + 48 / 24 KB
    constexpr int LW = NC / 16 / 8;                                      // W pieces (1 KiB = 16 tile rows) per wave and K step: 6 / 3
    constexpr int HALF_BYTES = 32 * NC * 4;                              // a 32-row half of the fp32 C tile
    constexpr int CH_ROW = NC / 4, CPL = CH_ROW / 16;                    // 16-byte chunks per C row, chunks per lane in the read-back (16 lanes per row)
    static_assert(3 * SLOT <= 160 * 1024 && HALF_BYTES <= 2 * SLOT - 0 && NC % 128 == 0, "ring / epilogue geometry");
    __shared__ __attribute__((aligned(1024))) char smem[3 * SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (p.M + RBM - 1) / RBM;
    const int xcd = blockIdx.x & 7, wk = blockIdx.x >> 3, nw = (int)gridDim.x >> 3;
    const int t_lo = (int)((long)T * xcd / 8), t_hi = (int)((long)T * (xcd + 1) / 8);
    const int nk = p.K / RBK;
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    const s16_t* Wb = reinterpret_cast<const s16_t*>(p.Wp);

    // ---- LDS-DMA: instruction ii fills bank rows 4 ii .. 4 ii + 3 (16 tile rows); lane -> (tile row, k chunk) through the swizzle ----
    const int pr_l = lane >> 4;
    // (bank row pr = 4 ii + pr_l, key pr & 3 = pr_l: the lane's logical slot and hence its (row in the 16-row piece, chunk) are the same for every piece)
    const int qs = (lane & 15) ^ pr_l;
    const int row16 = 4 * pr_l + (qs >> 2);                              // row within a 16-row piece
    const unsigned kch_b = (unsigned)((qs & 3) * 16);                    // byte offset of the lane's k chunk within the 64-byte K step
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Wb), 0, 0x7fffffff, 0x00020000);
    unsigned w_voff[LW];
#pragma unroll
    for (int j = 0; j < LW; ++j) w_voff[j] = (unsigned)((16 * (wave * LW + j) + row16) * p.wp_ld * 2) + kch_b;
    auto issue = [&](int m0, int kt, int slot) {                         // K step kt of the row tile at m0 into ring slot `slot`
        char* sb = smem + slot * SLOT;
        if (wave < 4) {
            const int r = 16 * wave + row16;
            const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)m0 * p.lda + p.a_coff), 0, 0x7fffffff, 0x00020000);
            const unsigned vo = (unsigned)((min(m0 + r, p.M - 1) - m0) * p.lda * 2) + kch_b;     // rows >= M re-read row M - 1 (never stored)
            blds16(a_rsrc, vo, kt * ROWB, sb + wave * 1024);
        }
#pragma unroll
        for (int j = 0; j < LW; ++j) blds16(w_rsrc, w_voff[j], kt * ROWB, sb + A_BYTES + (wave * LW + j) * 1024);
    };
    // fragment offsets within a slot: 16x16x32: lane reads row (lane & 15) of a 16-row block (4 bank rows), chunk lane >> 4 (all of a K step)
    const int f_off = ((lane & 15) >> 2) * 256 + (((((lane & 15) & 3) << 2) | (lane >> 4)) ^ (((lane & 15) >> 2) & 3)) * 16;

    int t = t_lo + wk;
    if (t < t_hi) issue(t * RBM, 0, 2);                                  // K step kt lives in slot (kt + 2) % 3
    for (; t < t_hi; t += nw) {
        const int m0 = t * RBM;
        f32x4_b acc[FM][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = zero4b();
        if (nk > 1) issue(m0, 1, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // this wave's pieces of step kt have landed when at most those of step kt + 1 are outstanding
            if (kt + 1 < nk) { if (wave < 4) wait_vmcnt<LW + 1>(); else wait_vmcnt<LW>(); }
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                // everybody's pieces landed; everybody is done reading step kt - 1
            if (kt + 2 < nk) issue(m0, kt + 2, (kt + 4) % 3);            // into the slot step kt - 1 just left
            const char* sb = smem + ((kt + 2) % 3) * SLOT;
            s16x8 af[FM], bfr[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const s16x8*>(sb + i * 1024 + f_off);
#pragma unroll
            for (int j = 0; j < FN; ++j) bfr[j] = *reinterpret_cast<const s16x8*>(sb + A_BYTES + (wave * FN + j) * 1024 + f_off);
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j] = BRN_MFMA_16X16X32(bfr[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                    // every fragment read of the tile is done: the ring is free
        const bool more = t + nw < t_hi;
        if (more) issue((t + nw) * RBM, 0, 2);                           // the next tile's first K step flies during the epilogue (slot 2 is not touched by it)
        // ---- epilogue: two 32-row halves through smem[0, HALF_BYTES) ----
        const int er = tid >> 4, ec = tid & 15;                          // read-back: row of the half, first chunk (then + 16 k)
        float* Cf = reinterpret_cast<float*>(p.C);
        const float* Rf = reinterpret_cast<const float*>(p.R);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m = m0 + half * 32 + er, mc = min(m, p.M - 1);
            f32x4_b rr[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) rr[k] = *reinterpret_cast<const f32x4_b*>(Rf + (long)mc * p.ldr + p.r_coff + (ec + 16 * k) * 4);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * half + ii, row = 16 * ii + (lane & 15);
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int col = wave * WCOL + 16 * j + 4 * (lane >> 4);
                    f32x4_b bias4 = p.bias ? *reinterpret_cast<const f32x4_b*>(p.bias + col) : zero4b();
                    const int ch = col >> 2;
                    *reinterpret_cast<f32x4_b*>(smem + row * (NC * 4) + (((ch & ~15) | ((ch ^ row) & 15)) << 4)) = acc[i][j] + bias4;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4_b xv[CPL];
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int ch = ec + 16 * k;
                xv[k] = *reinterpret_cast<const f32x4_b*>(smem + er * (NC * 4) + (((ch & ~15) | ((ch ^ er) & 15)) << 4)) + rr[k];
                sum += (xv[k][0] + xv[k][1]) + (xv[k][2] + xv[k][3]);
            }
            if (m < p.M) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) *reinterpret_cast<f32x4_b*>(Cf + (long)m * p.ldc + p.c_coff + (ec + 16 * k) * 4) = xv[k];
            }
            sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
            const float mean = sum / (float)NC;
            float sq = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                xv[k] = xv[k] - mean;
                sq += (xv[k][0] * xv[k][0] + xv[k][1] * xv[k][1]) + (xv[k][2] * xv[k][2] + xv[k][3] * xv[k][3]);
            }
            sq += __shfl_xor(sq, 1); sq += __shfl_xor(sq, 2); sq += __shfl_xor(sq, 4); sq += __shfl_xor(sq, 8);
            const float rstd = 1.0f / sqrtf(sq / (float)NC + eps);
            if (m < p.M) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    const int c0 = (ec + 16 * k) * 4;
                    const f32x4_b gm = *reinterpret_cast<const f32x4_b*>(gamma + c0), bt = *reinterpret_cast<const f32x4_b*>(beta + c0);
                    const f32x4_b o = xv[k] * rstd * gm + bt;
                    const u32x2_b ob = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    *reinterpret_cast<u32x2_b*>(Y + (long)m * ldy + c0) = ob;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // the half is free again (second half; after it: the K loop's slots 0 / 1)
        }
    }
}
bool gemm_rowln_eligible(const GemmParams& p) {
    return p.mode == GEMM_DENSE && p.Wp && (p.N == 768 || p.N == 384) && p.K >= 64 && (p.K % 32) == 0 && p.c_f32 && p.R && p.r_f32 && !p.scale && !p.bbias &&
           p.act == ACT_NONE && ((p.lda | p.a_coff) & 7) == 0 && ((p.ldc | p.c_coff | p.ldr | p.r_coff) & 3) == 0 && p.M >= 8192 && p.wp_rows >= p.N &&
           p.wp_ld >= p.K && (p.wp_ld & 7) == 0 && (double)p.wp_ld * 2.0 * p.N < 2147483648.0 && (double)p.lda * 2.0 * 64 < 2147483648.0;
}
hipError_t launch_gemm_rowln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s) {
    if (!BRN_S16_SELF(gemm_rowln_eligible)(p) || !gamma || !beta || !y_bf16 || (ldy & 3)) return hipErrorInvalidValue;
    const int tiles = (p.M + 63) / 64;
    int g = launch_cus();
    if (g > tiles) g = (tiles + 7) / 8 * 8;
    dim3 grid(g), block(512);
    if (p.N == 768) hipLaunchKernelGGL(gemm_rowln_bf16_kernel<768>, grid, block, 0, s, p, gamma, beta, eps, reinterpret_cast<s16_t*>(y_bf16), ldy);
    else hipLaunchKernelGGL(gemm_rowln_bf16_kernel<384>, grid, block, 0, s, p, gamma, beta, eps, reinterpret_cast<s16_t*>(y_bf16), ldy);
    return hipGetLastError();
}

bool gemm_wstat_ln_eligible(const GemmParams& p) {
    return p.mode == GEMM_DENSE && p.Wp && p.K == 192 && p.N == 192 && p.c_f32 && p.R && p.r_f32 && !p.scale && !p.bbias && p.act == ACT_NONE &&
           ((p.lda | p.a_coff) & 7) == 0 && ((p.ldc | p.c_coff | p.ldr | p.r_coff) & 3) == 0 && p.M >= 32768 && (long)p.lda * 2 * 64 < 0x7fffffffL;
}
hipError_t launch_gemm_wstat_ln(const GemmParams& p, const float* gamma, const float* beta, float eps, void* y_bf16, int ldy, hipStream_t s) {
    if (!BRN_S16_SELF(gemm_wstat_ln_eligible)(p) || !gamma || !beta || !y_bf16 || (ldy & 3)) return hipErrorInvalidValue;
    dim3 grid(8 * (launch_cus() * 2 / 8)), block(256);
    hipLaunchKernelGGL(gemm_wstat_ln_bf16_kernel, grid, block, 0, s, p, gamma, beta, eps, reinterpret_cast<s16_t*>(y_bf16), ldy);
    return hipGetLastError();
}

bool gemm_wstat_eligible(const GemmParams& p) {
    // (N / 192 column groups share the chip's workgroup slots: beyond that many groups there is no walker left per group, and the
    // tiled kernel serves the shape)
    return p.mode == GEMM_DENSE && p.Wp && (p.K == 192 || p.K == 384) && p.N >= 192 && (p.N % 192) == 0 && p.N / 192 <= launch_cus() * WSTAT_WG_PER_CU / 8 &&
           !p.c_f32 && !p.R && !p.scale && !p.bbias &&
           ((p.lda | p.a_coff | p.ldc | p.c_coff) & 7) == 0 && p.M >= 32768 && (long)p.lda * 2 * 64 < 0x7fffffffL;
}
hipError_t launch_gemm_wstat(const GemmParams& p, hipStream_t s) {
    if (!BRN_S16_SELF(gemm_wstat_eligible)(p)) return hipErrorInvalidValue;
    const int G = p.N / 192;
    const int nw = (launch_cus() * WSTAT_WG_PER_CU / 8) / G;
    if (nw < 1) return hipErrorInvalidValue;
    dim3 grid(8 * G * nw), block(256);
    if (p.K == 384) {
        if (p.act == ACT_GELU_ERF) hipLaunchKernelGGL((gemm_wstat_bf16_kernel<6, ACT_GELU_ERF, 32>), grid, block, 0, s, p);
        else if (p.act == ACT_RELU) hipLaunchKernelGGL((gemm_wstat_bf16_kernel<6, ACT_RELU, 32>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_wstat_bf16_kernel<6, ACT_NONE, 32>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    if (p.act == ACT_GELU_ERF) hipLaunchKernelGGL((gemm_wstat_bf16_kernel<3, ACT_GELU_ERF>), grid, block, 0, s, p);
    else if (p.act == ACT_RELU) hipLaunchKernelGGL((gemm_wstat_bf16_kernel<3, ACT_RELU>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_wstat_bf16_kernel<3, ACT_NONE>), grid, block, 0, s, p);
    return hipGetLastError();
}

// split-K second pass for the bf16 mode: fixed-order sum of the fp32 slices + the epilogue, bf16 (or fp32) out
__global__ void splitk_reduce_bf16_kernel(const GemmParams p) {
    const long total = (long)p.M * p.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.part[(long)s * total + idx];
        if (p.bbias) v += p.bbias[(long)(m / p.bbias_rows) * p.N + n];
        if (p.bias) v += p.bias[n];
        if (p.scale) v = v * p.scale[n] + p.shift[n];
        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == ACT_GELU_ERF) v = gelu_erf_b(v);
        if (p.R) v += p.r_f32 ? p.R[(long)m * p.ldr + p.r_coff + n]
                              : bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.R)[(long)m * p.ldr + p.r_coff + n]);
        if (p.c_f32) p.C[(long)m * p.ldc + p.c_coff + n] = v;
        else reinterpret_cast<s16_t*>(p.C)[(long)m * p.ldc + p.c_coff + n] = (s16_t)v;
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int BBK, bool M16 = BRN_BF16_MFMA16 != 0, bool BSINGLE = BRN_BF16_BSINGLE != 0, bool KT = false>
static hipError_t launch_bf16_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
    // persistent workgroups: as many as fit the chip at once (LDS-limited: NSTAGE ring slots each), never more than work items
    constexpr int LDS_BYTES = NSTAGE * (BM + BN) * BBK * 2;
    constexpr int WG_PER_CU = (160 * 1024 / LDS_BYTES) < (2048 / (WM * WN * 64)) ? (160 * 1024 / LDS_BYTES) : (2048 / (WM * WN * 64));
    const int slots = launch_cus() * (WG_PER_CU < 1 ? 1 : WG_PER_CU);
    dim3 grid(tiles < slots ? tiles : slots), block(WM * WN * 64);
    // epilogue flavour (see the kernel): 0 = bf16 out, 1 = fp32 out (+ fp32 residual), 2 = generic
    int epi = 2;
    const bool plain = p.splitk == 1 && (p.N & 7) == 0;
    if (plain && !p.c_f32 && ((p.ldc | p.c_coff) & 7) == 0 && (!p.R || (!p.r_f32 && ((p.ldr | p.r_coff) & 7) == 0))) epi = 0;
    else if (plain && !p.bbias && p.c_f32 && ((p.ldc | p.c_coff) & 3) == 0 && (!p.R || (p.r_f32 && ((p.ldr | p.r_coff) & 3) == 0))) epi = 1;
#define BRN_BF16_LAUNCH(MODE_, EPI_) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, MODE_, BBK, EPI_, M16, BSINGLE, KT && MODE_ == GEMM_DENSE>), grid, block, 0, s, p)
    if (p.mode == GEMM_DENSE) {
        if (epi == 0) BRN_BF16_LAUNCH(GEMM_DENSE, 0); else if (epi == 1) BRN_BF16_LAUNCH(GEMM_DENSE, 1); else BRN_BF16_LAUNCH(GEMM_DENSE, 2);
    } else if (p.mode == GEMM_CONV_NHWC) {
        // (flavour 1 = fp32 out: decoder_block1's conv_out, whose result p1 stays fp32 in mode bf16 — brn_graph.cpp, decoder_forward)
        if (epi == 0) BRN_BF16_LAUNCH(GEMM_CONV_NHWC, 0); else if (epi == 1 && !p.R) BRN_BF16_LAUNCH(GEMM_CONV_NHWC, 1); else BRN_BF16_LAUNCH(GEMM_CONV_NHWC, 2);
    } else return hipErrorInvalidValue;
#undef BRN_BF16_LAUNCH
    return hipGetLastError();
}

// Tile choice, from sweeps on MI355X (tools/gemm_bf16_sweep.py).  What bounds these kernels is the L2 -> LDS intake of a CU (~68 GB/s
// measured with every CU streaming) and the clock the chip holds under MFMA load, so the largest tile that still fills the chip wins
// (bytes per flop halve from 128x128 to 256x256); small grids keep two or three independent 4-wave workgroups per CU.  All K step
// 64, 2 ring slots, persistent workgroups:
//   cfg 0: 128x128, 4 waves (64 KB: 2 workgroups / CU)       cfg 1: 128x64, 4 waves (48 KB: 3 / CU)
//   cfg 2: 256x256, 8 waves (128 KB: 1 / CU), 32x32x16       cfg 3: 256x192, 8 waves (112 KB: 1 / CU)
// eff = measured throughput of a full-chip launch relative to cfg 2 (same box, end of round 2: after the buffer-addressed LDS-DMA
// and the mid-step barrier, which helped the 8-wave 256 x 256 tile most); cfg 3 wins where it divides N or the grid better.
struct Bf16Cfg { int cfg, bm, bn, slots; double eff, eff_long_k; };
static const Bf16Cfg kBf16Cfgs[] = {{0, 128, 128, 512, 0.86, 0.80}, {1, 128, 64, 768, 0.72, 0.68}, {2, 256, 256, 256, 1.00, 1.00}, {3, 256, 192, 256, 0.93, 0.85}};
GemmPlan plan_gemm_bf16(int M, int N, int K, bool f32_residual, bool gelu) {
    GemmPlan pl{0, 1, 0};
    double best = 1e300;
    long best_tiles = 1;
    for (const Bf16Cfg& c : kBf16Cfgs) {
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        // W rows are padded to multiples of 256 (brn_weights.cpp): a 192-wide tile grid must not reach past that padding (N = 512
        // would need 576 rows; found at 1056 x 1056, where M = 17424 made 256 x 192 the cheapest tile for the ASPP 1 x 1 pair)
        if ((N + c.bn - 1) / c.bn * c.bn > (N + 255) / 256 * 256) continue;
        // a launch lasts ~ rounds x (tile area x workgroups sharing a CU) / relative CU throughput of the config
        double eff = K > 1536 ? c.eff_long_k : c.eff;
        // (f32_residual — proj / fc2 — is a hint without effect for now: two workgroups per CU measured 4 % faster in isolation on
        // the proj shape and 8 % slower inside the model, where A comes straight out of the attention kernel)
        (void)f32_residual;
        // a GELU epilogue (fc1) is VALU time no tile hides: 256 x 192 measured 582-596 TF/s there against 647-669 for 256 x 256
        if (gelu && c.cfg == 3) eff *= 0.9;
        const double cost = (double)((tiles + c.slots - 1) / c.slots) * (c.slots / 256) * c.bm * c.bn / eff;
        if (cost < best) { best = cost; pl.cfg = c.cfg; best_tiles = tiles; }
    }
    const int nk = (K + 63) / 64;
    if (best_tiles < 200 && nk >= 16) {   // tall-K convs on small maps: cut K so that ~512 work items exist (>= 8 K steps per slice)
        int s = (int)(512 / best_tiles);
        if (s > nk / 8) s = nk / 8;
        if (s > 32) s = 32;
        if (s > 1) { pl.splitk = s; pl.ws_floats = (size_t)s * M * N; }
    }
    return pl;
}

hipError_t launch_gemm_bf16(const GemmParams& p_in, const GemmPlan& pl, float* ws, hipStream_t s) {
    if (p_in.M <= 0 || p_in.N <= 0 || p_in.K <= 0 || (p_in.K % 32) != 0 || !p_in.Wp) return hipErrorInvalidValue;
    if ((p_in.lda | p_in.a_coff) & 7) return hipErrorInvalidValue;                       // 16-byte chunks of 8 bf16
    if (p_in.wp_ld < (p_in.K + 63) / 64 * 64 || (p_in.wp_ld & 7)) return hipErrorInvalidValue;
    if (p_in.mode == GEMM_CONV_NHWC && ((p_in.Cin & 31) || p_in.Cin < 64 || p_in.K != p_in.kh * p_in.kw * p_in.Cin)) return hipErrorInvalidValue;
    if (p_in.k_chunk_major && (p_in.mode != GEMM_CONV_NHWC || (p_in.Cin & 63))) return hipErrorInvalidValue;
    if (p_in.mode != GEMM_DENSE && p_in.mode != GEMM_CONV_NHWC) return hipErrorInvalidValue;
    if (p_in.mode == GEMM_CONV_NHWC && (double)p_in.Hin * p_in.Win * p_in.lda * 2.0 >= 1073741824.0) return hipErrorInvalidValue;   // 32-bit buffer offsets span two images
    const int bn_need = pl.cfg == 1 ? 64 : ((pl.cfg == 2 || pl.cfg == 19) ? 256 : (pl.cfg == 3 ? 192 : 128));
    if (p_in.wp_rows < (p_in.N + bn_need - 1) / bn_need * bn_need) return hipErrorInvalidValue;   // W rows padded to the tile
    GemmParams p = p_in;
    p.splitk = pl.splitk < 1 ? 1 : pl.splitk;
    p.part = ws;
    if (p.splitk > 1 && !ws) return hipErrorInvalidValue;
    hipError_t e;
    // a dense K tail (K % 64 == 32: none in the model, reachable through the op-level entry points) runs on the 128 x 128 tile's general-address form
    if (p.mode == GEMM_DENSE && (p.K & 63)) e = launch_bf16_cfg<128, 128, 2, 2, 2, 64, BRN_BF16_MFMA16 != 0, BRN_BF16_BSINGLE != 0, true>(p, s);
    else if (pl.cfg == 1) e = launch_bf16_cfg<128, 64, 2, 2, 2, 64>(p, s);
    else if (pl.cfg == 2) e = launch_bf16_cfg<256, 256, 4, 2, 2, 64, BRN_BF16_CFG2_M16 != 0, BRN_BF16_CFG2_M16 != 0>(p, s);
    else if (pl.cfg == 3) e = launch_bf16_cfg<256, 192, 4, 2, 2, 64>(p, s);
#ifdef BRN_DIAG_BUILD          // candidates kept for sweeps (tools/gemm_bf16_sweep.py)
    else if (pl.cfg == 12) e = launch_bf16_cfg<256, 128, 4, 2, 2, 64>(p, s);
    else if (pl.cfg == 19) e = launch_bf16_cfg<256, 256, 2, 2, 2, 64, false, false>(p, s);   // 4 waves of 128 x 128, accumulators in AGPRs
#endif
    else e = launch_bf16_cfg<128, 128, 2, 2, 2, 64>(p, s);
    if (e != hipSuccess || p.splitk == 1) return e;
    long total = (long)p.M * p.N;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

#if BRN_S16_F16
}  // namespace hf
#endif
}  // namespace brn
