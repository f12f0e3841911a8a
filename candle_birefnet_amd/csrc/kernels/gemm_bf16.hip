// gemm_bf16.hip — the contraction kernel of compute mode BRN_BF16 (BASELINE configs[2..4]): activations and weights live in
// HBM as bf16, C = epilogue(A_gather[M,K] x W[N,K]^T) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, bf16 (or fp32) out.
// Same role as gemm_f32.hip (every candle Linear / Conv2d of the path: swin.rs:98-99,130-131,487; decoder.rs:44-45,65,104,113;
// aspp.rs:39-45,271,282; birefnet.rs:105) and the same fused epilogue (bias / per-image bias / folded eval-BN / ReLU / erf-GELU /
// residual / concat-slice write).
//
// Structure (all waves load AND multiply; no register staging at all):
//   * operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave instruction), K step 64
//     (128-byte tile rows), NSTAGE-deep LDS ring, ONE raw s_barrier per K step, counted s_waitcnt vmcnt(N): the tiles of
//     the next NSTAGE-2 K steps stay in flight across the barrier (cdna_hip_programming.md, "Pipelining across barriers");
//   * LDS image of a tile: rows 2p, 2p+1 share one 256-byte bank row, whose sixteen 16-byte slots are XOR-permuted by
//     (p & 15): slot(r, c) = ((r & 1) << 3 | c) ^ ((r >> 1) & 15).  A ds_read_b128 lane group (16 lanes = 16 different rows,
//     same logical chunk c) then touches 16 different slots: conflict-free.  global_load_lds writes lane-linear, so the
//     permutation is applied to the per-lane SOURCE address (which row / chunk a lane fetches), never to the destination;
//   * a wave's instructions are i = w + NW j, so a lane's (row parity, chunk) is the same for all of them: the implicit-GEMM
//     modes compute ONE (tap, channel) per lane per K step, by increments (no division in the loop);
//   * masked elements (conv zero padding, rows >= M, the K tail) are fetched from a 16-byte zero page: a select on the
//     ADDRESS, never on the data.
#include "../brn_kernels.h"
#include "split_planes.h"

namespace brn {

typedef float f32x16_b __attribute__((ext_vector_type(16)));
typedef float f32x4_b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));

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

// gelu_erf for a result that is rounded to bf16 right away (flavour 0): erfc by Abramowitz-Stegun 7.1.25 (3 terms), |gelu error|
// < 2.6e-5 absolute and < 0.23 % relative for |gelu| >= 1e-2 — under half a bf16 ulp (0.39 %) — at half the VALU work of the
// degree-7 fit, which the fp32 outputs keep.  (fc1's epilogue is VALU time the persistent workgroup cannot hide behind MFMAs.)
__device__ __forceinline__ float gelu_erf_bf16out(float x) {
    const float s = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.47047f, s, 1.0f));
    float q = fmaf(0.7478556f, t, -0.0958798f);
    q = fmaf(q, t, 0.3480242f);
    q = q * t * __expf(-s * s);
    const float one_plus_erf = x < 0.f ? q : 2.0f - q;
    return 0.5f * x * one_plus_erf;
}

__device__ __forceinline__ void bf16_tile_coords(int tile, int tilesM, int tilesN, int& tm, int& tn) {
    constexpr int GN = 8;            // N walked in groups of 8 tile columns, M fastest-but-one inside a group (L2 reuse of the W panels)
    const int per_group = tilesM * GN;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gw = min(GN, tilesN - g * GN);
    tm = r / gw;
    tn = g * GN + (r - tm * gw);
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ f32x4_b zero4b() { f32x4_b z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

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
                    v[2 * e] += __builtin_bit_cast(float, r[e] << 16);
                    v[2 * e + 1] += __builtin_bit_cast(float, r[e] & 0xffff0000u);
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
        __bf16* dst = reinterpret_cast<__bf16*>(p.C) + (long)m * p.ldc + p.c_coff + n;
        if (VEC) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *reinterpret_cast<bf16x8*>(dst) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = (__bf16)v[e];
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
template <int BM, int BN, int WM, int WN, int NSTAGE, int MODE, int BBK, int EPI>
__global__ void __launch_bounds__(WM* WN * 64) gemm_bf16_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
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
    // epilogue patches (one per wave, PR rows x EWN floats, unpadded, 16-byte chunks XOR-ed with the row parity) live in the LAST ring
    // slot; a wave's WTM x WTN accumulators go through it in blocks of PR rows x EWN columns
    constexpr int EWN = (WTN % 64 == 0) ? 64 : 32;
    constexpr int NJB = WTN / EWN, JPB = EWN / 32;              // column blocks per wave tile, 32-wide MFMA tiles per column block
    constexpr int PR = (NW * 32 * EWN * 4 <= STAGE_BYTES) ? 32 : (NW * 16 * EWN * 4 <= STAGE_BYTES) ? 16 : 8;
    static_assert(NW * PR * EWN * 4 <= STAGE_BYTES, "epilogue patches must fit one ring slot");
    constexpr int NB = 32 / PR;                                 // patch blocks per 32-row accumulator block
    constexpr int LPR = EWN / 8, RPP = 64 / LPR;                // lanes per row (8 outputs each), rows per pass
    constexpr int PASSES = PR / RPP;
    static_assert(PASSES >= 1, "patch smaller than one pass");
    constexpr int STORES = TM * NJB * NB * PASSES * (EPI == 1 ? 2 : 1);   // store instructions per wave and FULL tile (flavours 0 / 1)
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef BRN_DIAG_BUILD
    const int abl = p.abl;       // 1: no A loads, 2: no W loads, 4: no fragment reads / MFMA, 8: no epilogue
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
    const __bf16* Ab = reinterpret_cast<const __bf16*>(p.A);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(p.Wp);

    // fragment addressing: lane reads row (lane & 31) of a 32-row block, logical chunk 2 s + (lane >> 5) at k16 step s
    constexpr int KS16 = BBK / 16;
    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow / RPB) & SWZ_MASK;                    // same for every 32-row block (blocks are 32 / RPB bank rows apart)
    int foff[KS16];
#pragma unroll
    for (int s = 0; s < KS16; ++s) foff[s] = (frow / RPB) * 256 + ((((frow % RPB) * CPR + (2 * s + fh)) ^ fswz) << 4);
    const int a_base = wm * WTM * ROWB, b_base = A_BYTES + wn * WTN * ROWB;

    // ---- state of the work item whose K steps are being staged ----
    int m0 = 0, n0 = 0, slice = 0, kt0 = 0, nt = 0;
    long a_off[LA];        // dense: element offset of (row, k = kch); conv: element offset of image b of the row's pixel (+ a_coff)
    int a_iy[LA], a_ix[LA];
    bool a_ok[LA];
    int c_ci = 0, c_ky = 0, c_kx = 0;      // conv: (tap, channel) of this lane's chunk, advanced by BBK channels per K step
    long w_off0 = 0;
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
            if (MODE == GEMM_DENSE) {
                a_off[j] = (long)m * p.lda + p.a_coff + kch;
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                a_iy[j] = oy * p.stride - p.pad;
                a_ix[j] = ox * p.stride - p.pad;
                a_off[j] = (long)b * p.Hin * p.Win * p.lda + p.a_coff;
            }
        }
        if (MODE == GEMM_CONV_NHWC) {
            const int k = kt0 * BBK + kch;
            const int tap = k / p.Cin;
            c_ci = k - tap * p.Cin;
            c_ky = tap / p.kw; c_kx = tap - c_ky * p.kw;
        }
        // W rows n0 + lrow + RPI NW j of the padded [rows][wp_ld] bf16 matrix (rows and K zero-padded to the tile: always in bounds)
        w_off0 = (long)(n0 + lrow) * p.wp_ld + kch;
    };
    auto stage = [&](int t) {                                   // issue the loads of K step t (local index) into ring slot t % NSTAGE
        char* sbase = smem + (t % NSTAGE) * STAGE_BYTES + wave * 1024;
        const int kbase = (kt0 + t) * BBK;
        if (MODE == GEMM_DENSE) {
            const bool kin = kbase + kch < p.K;                  // K tail (K % 64 == 32): the upper chunks read zeros
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const char* src = (a_ok[j] && kin) ? reinterpret_cast<const char*>(Ab + a_off[j] + kbase) : zero;
                if (!(abl & 1)) glds16(src, sbase + j * (NW * 1024));
            }
        } else {
            const bool kin = c_ky < p.kh;                        // beyond the last tap: K tail
            const int dy = c_ky * p.dil, dx = c_kx * p.dil;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int iy = a_iy[j] + dy, ix = a_ix[j] + dx;
                const bool ok = a_ok[j] && kin && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const char* src = ok ? reinterpret_cast<const char*>(Ab + a_off[j] + ((long)iy * p.Win + ix) * p.lda + c_ci) : zero;
                if (!(abl & 1)) glds16(src, sbase + j * (NW * 1024));
            }
            c_ci += BBK;
            if (c_ci >= p.Cin) { c_ci -= p.Cin; if (++c_kx == p.kw) { c_kx = 0; ++c_ky; } }
        }
#pragma unroll
        for (int j = 0; j < LB; ++j)
            if (!(abl & 2)) glds16(Wb + w_off0 + (long)(RPI * NW * j) * p.wp_ld + kbase, sbase + A_BYTES + j * (NW * 1024));
    };

    // ---- epilogue constants ----
    float* patch = reinterpret_cast<float*>(smem + (NSTAGE - 1) * STAGE_BYTES) + wave * (PR * EWN);
    const int col = lane & 31, rhalf = (lane >> 5) * 4;           // C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int er = lane / LPR, ec = (lane % LPR) * 8;             // read-back map: row er of a pass, columns ec .. ec + 7
    const bool has_scale = p.scale != nullptr;
    const int act = p.act;

    bool have = id < id_end;
    if (have) {
        setup(id);
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nt) stage(s);
    }
    bool counted = false;        // the previous item's epilogue issued exactly STORES stores per wave after this item's first loads
    while (have) {
        f32x16_b acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        for (int t = 0; t < nt; ++t) {
            // K step t has landed once at most `after` younger steps (and, at t = 0, the previous item's stores) are outstanding
            const int after = min(nt - 1, t + NSTAGE - 2) - t;
            if (t == 0 && counted && after == NSTAGE - 2) wait_vmcnt<(NSTAGE - 2) * LPS + STORES>();
            else if (NSTAGE >= 4 && after >= 2) wait_vmcnt<2 * LPS>();
            else if (NSTAGE >= 3 && after >= 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                        // every wave's part of step t is in LDS; slot (t-1) % NSTAGE is free
            if (t + NSTAGE - 1 < nt) stage(t + NSTAGE - 1);
            if (abl & 4) continue;
            const char* sb = smem + (t % NSTAGE) * STAGE_BYTES;
            // fragments of k16 step s+1 are read while the MFMAs of step s run (two register sets)
            bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * (32 * ROWB) + foff[0]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const bf16x8*>(sb + b_base + j * (32 * ROWB) + foff[0]);
#pragma unroll
            for (int s = 0; s < KS16; ++s) {
                if (s + 1 < KS16) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * (32 * ROWB) + foff[(s + 1) % KS16]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(sb + b_base + j * (32 * ROWB) + foff[(s + 1) % KS16]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
            }
        }

        // ---- hand over: the finished item keeps (e_m0, e_n0, e_slice); the staging state moves on to the next item ----
        const int e_m0 = m0, e_n0 = n0, e_slice = slice;
        id += id_step;
        have = id < id_end;
        __builtin_amdgcn_s_barrier();                            // all fragment reads of the finished item are done: the ring is free
        if (have) {
            setup(id);
#pragma unroll
            for (int s = 0; s < NSTAGE - 1; ++s)
                if (s < nt) stage(s);                            // next item's first K steps: in flight during the epilogue below
        }
        if (abl & 8) { if (acc[0][0][0] == 123.456f) p.C[0] = 1.f; counted = false; continue; }

        // ---- epilogue: PR x WTN row blocks through the wave's LDS patch (last ring slot), whole row segments to HBM ----
        const bool split = EPI == 2 && p.splitk > 1;
        float* part = split ? p.part + (long)e_slice * p.M * p.N : nullptr;
        const long rowbase = (long)(e_m0 + wm * WTM);
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            const int n = e_n0 + wn * WTN + jb * EWN + ec;
            float bias[8], sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bias[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f;
                if (!split && n + e < p.N) {
                    if (p.bias) bias[e] = p.bias[n + e];
                    if (has_scale) { sc[e] = p.scale[n + e]; sh[e] = p.shift[n + e]; }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int hb = 0; hb < NB; ++hb) {
                    const long mb = rowbase + i * 32 + hb * PR;   // first row of this patch block
                    f32x4_b rres[PASSES][2];
                    if (EPI == 1) {                              // residual rows of the block, fetched before the LDS round trip
#pragma unroll
                        for (int ps = 0; ps < PASSES; ++ps) {
                            const long m = mb + ps * RPP + er;
                            rres[ps][0] = zero4b(); rres[ps][1] = zero4b();
                            if (p.R && m < p.M && n < p.N) {
                                const float* rp = p.R + m * p.ldr + p.r_coff + n;
                                rres[ps][0] = *reinterpret_cast<const f32x4_b*>(rp);
                                rres[ps][1] = *reinterpret_cast<const f32x4_b*>(rp + 4);
                            }
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < JPB; ++jj)
#pragma unroll
                        for (int r = hb * (16 / NB); r < (hb + 1) * (16 / NB); ++r) {
                            const int prow = (r & 3) + 8 * ((r >> 2) % (PR >= 8 ? (PR / 8) : 1)) + rhalf;
                            const int pcol = jj * 32 + col;
                            patch[prow * EWN + ((((pcol >> 2) ^ (prow & 1)) << 2) | (pcol & 3))] = acc[i][jb * JPB + jj][r];
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int ps = 0; ps < PASSES; ++ps) {
                        const int row = ps * RPP + er;
                        const long m = mb + row;
                        const int par = row & 1;
                        const f32x4_b v0 = *reinterpret_cast<const f32x4_b*>(patch + row * EWN + (((ec >> 2) ^ par) << 2));
                        const f32x4_b v1 = *reinterpret_cast<const f32x4_b*>(patch + row * EWN + ((((ec >> 2) + 1) ^ par) << 2));
                        if (m >= p.M || n >= p.N) continue;
                        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        if (EPI == 2) {
                            if (split) {
                                float* dst = part + m * p.N + n;
#pragma unroll
                                for (int e = 0; e < 8; ++e) if (n + e < p.N) dst[e] = v[e];
                            } else {
                                store_row8<false>(p, (int)m, n, v, bias, sc, sh);
                            }
                            continue;
                        }
                        // flavours 0 / 1: whole 8-column groups (the launcher checked N % 8 == 0 and the alignments)
                        if (EPI == 0 && p.bbias) {               // per-image bias (the pooled ASPP branch folded into conv1)
                            const float* bp = p.bbias + (m / p.bbias_rows) * p.N + n;
                            const f32x4_b b0 = *reinterpret_cast<const f32x4_b*>(bp), b1 = *reinterpret_cast<const f32x4_b*>(bp + 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float tt = v[e] + bias[e];
                            if (has_scale) tt = tt * sc[e] + sh[e];
                            v[e] = tt;
                        }
                        if (act == ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                        } else if (act == ACT_GELU_ERF) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = EPI == 0 ? gelu_erf_bf16out(v[e]) : gelu_erf_b(v[e]);
                        }
                        if (EPI == 0) {
                            if (p.R) {                           // bf16 residual (the decoder's lateral adds, in place)
                                const u32x4_b rr = *reinterpret_cast<const u32x4_b*>(reinterpret_cast<const unsigned short*>(p.R) + m * p.ldr + p.r_coff + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[2 * e] += __builtin_bit_cast(float, rr[e] << 16);
                                    v[2 * e + 1] += __builtin_bit_cast(float, rr[e] & 0xffff0000u);
                                }
                            }
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.C) + m * p.ldc + p.c_coff + n) = o;
                        } else {
                            f32x4_b a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
                            a = a + rres[ps][0];
                            b = b + rres[ps][1];
                            float* dst = p.C + m * p.ldc + p.c_coff + n;
                            *reinterpret_cast<f32x4_b*>(dst) = a;
                            *reinterpret_cast<f32x4_b*>(dst + 4) = b;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();              // the patch is rewritten by the next block
                }
            }
        }
        // every wave of a FULL tile issued exactly STORES store instructions after the next item's first loads
        counted = EPI < 2 && e_m0 + BM <= p.M && e_n0 + BN <= p.N;
    }
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
        else reinterpret_cast<__bf16*>(p.C)[(long)m * p.ldc + p.c_coff + n] = (__bf16)v;
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int BBK = 64>
static hipError_t launch_bf16_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
    // persistent workgroups: as many as fit the chip at once (LDS-limited: NSTAGE ring slots each), never more than work items
    constexpr int LDS_BYTES = NSTAGE * (BM + BN) * BBK * 2;
    constexpr int WG_PER_CU = (160 * 1024 / LDS_BYTES) < (2048 / (WM * WN * 64)) ? (160 * 1024 / LDS_BYTES) : (2048 / (WM * WN * 64));
    const int slots = 256 * (WG_PER_CU < 1 ? 1 : WG_PER_CU);
    dim3 grid(tiles < slots ? tiles : slots), block(WM * WN * 64);
    // epilogue flavour (see the kernel): 0 = bf16 out, 1 = fp32 out (+ fp32 residual), 2 = generic
    int epi = 2;
    const bool plain = p.splitk == 1 && (p.N & 7) == 0;
    if (plain && !p.c_f32 && ((p.ldc | p.c_coff) & 7) == 0 && (!p.R || (!p.r_f32 && ((p.ldr | p.r_coff) & 7) == 0))) epi = 0;
    else if (plain && !p.bbias && p.c_f32 && ((p.ldc | p.c_coff) & 3) == 0 && (!p.R || (p.r_f32 && ((p.ldr | p.r_coff) & 3) == 0))) epi = 1;
#define BRN_BF16_LAUNCH(MODE_, EPI_) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, NSTAGE, MODE_, BBK, EPI_>), grid, block, 0, s, p)
    if (p.mode == GEMM_DENSE) {
        if (epi == 0) BRN_BF16_LAUNCH(GEMM_DENSE, 0); else if (epi == 1) BRN_BF16_LAUNCH(GEMM_DENSE, 1); else BRN_BF16_LAUNCH(GEMM_DENSE, 2);
    } else if (p.mode == GEMM_CONV_NHWC) {
        if (epi == 0) BRN_BF16_LAUNCH(GEMM_CONV_NHWC, 0); else BRN_BF16_LAUNCH(GEMM_CONV_NHWC, 2);
    } else return hipErrorInvalidValue;
#undef BRN_BF16_LAUNCH
    return hipGetLastError();
}

// Tile choice, from sweeps on MI355X (tools/gemm_bf16_sweep.py, profiles/r02_bf16_gemm_sweep.txt).  What bounds these kernels is the
// L2 -> LDS intake of a CU (~68 GB/s measured with every CU streaming), not the matrix pipe, so the largest tile that still
// fills the chip wins (bytes per flop halve from 128x128 to 256x256); small grids keep two or three independent 4-wave
// workgroups per CU.  All K step 64, 2 ring slots, persistent workgroups:
//   cfg 0: 128x128, 4 waves (64 KB: 2 workgroups / CU)       cfg 1: 128x64, 4 waves (48 KB: 3 / CU)
//   cfg 2: 256x256, 8 waves (128 KB: 1 / CU)                 cfg 3: 256x192, 8 waves (112 KB: 1 / CU; N = 192, 384, 576, 1152)
struct Bf16Cfg { int cfg, bm, bn, slots; double eff; };
static const Bf16Cfg kBf16Cfgs[] = {{0, 128, 128, 512, 0.88}, {1, 128, 64, 768, 0.72}, {2, 256, 256, 256, 1.00}, {3, 256, 192, 256, 0.92}};
GemmPlan plan_gemm_bf16(int M, int N, int K) {
    GemmPlan pl{0, 1, 0};
    double best = 1e300;
    long best_tiles = 1;
    for (const Bf16Cfg& c : kBf16Cfgs) {
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        // a launch lasts ~ rounds x (tile area x workgroups sharing a CU) / relative CU throughput of the config
        const double cost = (double)((tiles + c.slots - 1) / c.slots) * (c.slots / 256) * c.bm * c.bn / c.eff;
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
    if (p_in.mode != GEMM_DENSE && p_in.mode != GEMM_CONV_NHWC) return hipErrorInvalidValue;
    const int bn_need = (pl.cfg == 1 || pl.cfg == 15) ? 64 : ((pl.cfg == 2 || pl.cfg == 13) ? 256 : (pl.cfg == 3 ? 192 : 128));
    if (p_in.wp_rows < (p_in.N + bn_need - 1) / bn_need * bn_need) return hipErrorInvalidValue;   // W rows padded to the tile
    GemmParams p = p_in;
    p.splitk = pl.splitk < 1 ? 1 : pl.splitk;
    p.part = ws;
    if (p.splitk > 1 && !ws) return hipErrorInvalidValue;
    hipError_t e;
    if (pl.cfg == 1) e = launch_bf16_cfg<128, 64, 2, 2, 2, 64>(p, s);
    else if (pl.cfg == 2) e = launch_bf16_cfg<256, 256, 4, 2, 2, 64>(p, s);
    else if (pl.cfg == 3) e = launch_bf16_cfg<256, 192, 4, 2, 2, 64>(p, s);
#ifdef BRN_DIAG_BUILD          // candidates kept for sweeps (tools/gemm_bf16_sweep.py)
    else if (pl.cfg == 10) e = launch_bf16_cfg<128, 128, 2, 2, 3, 32>(p, s);
    else if (pl.cfg == 11) e = launch_bf16_cfg<256, 128, 4, 2, 3, 64>(p, s);
    else if (pl.cfg == 12) e = launch_bf16_cfg<256, 128, 4, 2, 2, 64>(p, s);
    else if (pl.cfg == 13) e = launch_bf16_cfg<256, 256, 4, 2, 3, 32>(p, s);
    else if (pl.cfg == 14) e = launch_bf16_cfg<128, 128, 2, 2, 4, 32>(p, s);
    else if (pl.cfg == 15) e = launch_bf16_cfg<256, 64, 4, 2, 3, 64>(p, s);
#endif
    else e = launch_bf16_cfg<128, 128, 2, 2, 2, 64>(p, s);
    if (e != hipSuccess || p.splitk == 1) return e;
    long total = (long)p.M * p.N;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace brn
