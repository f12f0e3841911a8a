// gemm_planes.hip — the split-bf16 contraction (compute modes f32_split3 / f32_split2) for operands that ALREADY exist as bf16
// planes in HBM: A in the "P" activation layout its producer wrote (kernels/split_planes.h: LayerNorm, window attention, the fc1
// epilogue), W in the interleaved plane layout built at load time.  Same arithmetic as gemm_split_ws_kernel (gemm_f32.hip): every
// fp32 operand is NP bf16 planes (error-free for NP = 3), a product is the sum of the plane products down to 2^-16 of the leading
// one (NP = 3: hh, hm, mh, hl, lh, mm; NP = 2: hh, hl, lh), smallest first, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// What is different is how operands reach the matrix cores: with the planes already in memory NO wave has to split anything, so
// the structure of kernels/gemm_bf16.hip applies unchanged — global_load_lds straight into an XOR-swizzled LDS ring (K step 32:
// one 64-byte row per plane), counted vmcnt, one barrier per K step, every wave loads and multiplies, persistent workgroups
// with the next item's first K steps in flight during the epilogue.  The warp-specialised kernel is bound by its producer waves'
// split + LDS-store chain; here a 128 x 128 x 32 step moves 48 KB (NP = 3) for 48 MFMAs per wave: 32 B / clock
// per CU, under the measured ~68 GB/s L2 -> LDS intake, i.e. the loop is MFMA-bound by construction.
// Replaces candle_nn::linear of the Swin blocks (swin.rs:98-99,130-131) in the split modes; dense A only.
#include "../brn_kernels.h"
#include "split_planes.h"

namespace brn {

typedef float f32x16_p __attribute__((ext_vector_type(16)));
typedef float f32x4_p __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_p __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) unsigned g_zero_page_p[64];

__device__ __forceinline__ float gelu_erf_p(float x) {   // same fit as gemm_f32.hip (|error| < 2e-7)
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
__device__ __forceinline__ void glds16p(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt_p() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ f32x4_p zero4p() { f32x4_p z = {0.f, 0.f, 0.f, 0.f}; return z; }

// EPI: 1 = fp32 C (+ bias, + fp32 residual)        3 = C in the P layout with NP planes (+ bias, + activation): fc1 -> fc2
template <int BM, int BN, int WM, int WN, int NSTAGE, int NP, int EPI>
__global__ void __launch_bounds__(WM* WN * 64) gemm_planes_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int BBK = 32, ROWB = 64, RPB = 4, CPR = 4, RPI = 16;       // one K step = one 32-deep K tile: 64-byte rows per plane
    constexpr int LA = BM / RPI / NW, LB = BN / RPI / NW;       // load instructions per wave, stage and plane
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0 && TM >= 1 && TN >= 1, "tile does not divide over the waves");
    static_assert(NP == 2 || NP == 3, "planes");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4, "ring depth");
    constexpr int LPS = NP * (LA + LB);                         // vmcnt units per stage and wave
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_BYTES = NP * A_PLANE, STAGE_BYTES = NP * (A_PLANE + B_PLANE);
    constexpr int EWN = (WTN % 64 == 0) ? 64 : 32;
    constexpr int NJB = WTN / EWN, JPB = EWN / 32;
    constexpr int PR = (NW * 32 * EWN * 4 <= STAGE_BYTES) ? 32 : (NW * 16 * EWN * 4 <= STAGE_BYTES) ? 16 : 8;
    static_assert(NW * PR * EWN * 4 <= STAGE_BYTES, "epilogue patches must fit one ring slot");
    constexpr int NB = 32 / PR;
    constexpr int LPR = EWN / 8, RPP = 64 / LPR;
    constexpr int PASSES = PR / RPP;
    static_assert(PASSES >= 1, "patch smaller than one pass");
    constexpr int STORES = TM * NJB * NB * PASSES * (EPI == 1 ? 2 : NP);   // store instructions per wave and FULL tile
    __shared__ __attribute__((aligned(1024))) char smem[NSTAGE * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- work distribution: XCD x (= blockIdx % 8) owns a contiguous run of work ids (tile x K slice) ----
    const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.N + BN - 1) / BN;
    const int ntiles = tilesM * tilesN, total = ntiles * p.splitk;
    int id, id_end, id_step;
    {
        const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
        const int cs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        id_end = cs + q + (xcd < r ? 1 : 0);
        id_step = ((int)gridDim.x - xcd + 7) >> 3;
        id = cs + ((int)blockIdx.x >> 3);
    }
    const int nk_all = p.K / BBK;
    const int kts = (nk_all + p.splitk - 1) / p.splitk;

    // ---- this lane's share of a plane image: bank row pr = 4 (w + NW j) + (lane >> 4), slot q' = lane & 15 ----
    const int pr0 = 4 * wave + (lane >> 4);
    const int qs = (lane & 15) ^ (pr0 & 3);
    const int lrow = RPB * pr0 + qs / CPR;
    const int kchb = (qs % CPR) * 16;                           // byte offset of this lane's 16-byte chunk inside a plane's 64-byte K tile
    const char* zero = reinterpret_cast<const char*>(g_zero_page_p);
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const char* Wb = reinterpret_cast<const char*>(p.Wp);
    const long lda_b = (long)p.lda * 4;                         // bytes per A row (P layout: NP * 2 K)
    const long ldw_b = (long)p.K * NP * 2;                      // bytes per W row ([row][K/32][plane][32] bf16)

    constexpr int KS16 = 2;
    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow / RPB) & 3;
    int foff[KS16];
#pragma unroll
    for (int s = 0; s < KS16; ++s) foff[s] = (frow / RPB) * 256 + ((((frow % RPB) * CPR + (2 * s + fh)) ^ fswz) << 4);
    const int a_base = wm * WTM * ROWB, b_base = A_BYTES + wn * WTN * ROWB;

    int m0 = 0, n0 = 0, slice = 0, kt0 = 0, nt = 0;
    long a_off[LA];
    bool a_ok[LA];
    long w_off0 = 0;
    auto setup = [&](int work) {
        slice = work / ntiles;
        const int tile = work - slice * ntiles;
        // N walked in groups of 8 tile columns, M fastest-but-one inside a group (L2 reuse of the W panels)
        constexpr int GN = 8;
        const int per_group = tilesM * GN;
        const int g = tile / per_group, r = tile - g * per_group;
        const int gw = min(GN, tilesN - g * GN);
        const int tile_m = r / gw, tile_n = g * GN + (r - tile_m * gw);
        m0 = tile_m * BM; n0 = tile_n * BN;
        kt0 = slice * kts;
        const int nk = min(nk_all, kt0 + kts);
        nt = nk > kt0 ? nk - kt0 : 0;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int m = m0 + lrow + RPI * NW * j;
            a_ok[j] = m < p.M;
            a_off[j] = (long)m * lda_b + kchb;
        }
        w_off0 = (long)(n0 + lrow) * ldw_b + kchb;              // W rows are padded to 128: always in bounds
    };
    auto stage = [&](int t) {
        char* sbase = smem + (t % NSTAGE) * STAGE_BYTES + wave * 1024;
        const long kb = (long)(kt0 + t) * (64 * NP);            // byte offset of K tile kt inside a row (A and W alike)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const char* src = a_ok[j] ? Ab + a_off[j] + kb + pl * 64 : zero;
                glds16p(src, sbase + pl * A_PLANE + j * (NW * 1024));
            }
#pragma unroll
            for (int j = 0; j < LB; ++j)
                glds16p(Wb + w_off0 + (long)(RPI * NW * j) * ldw_b + kb + pl * 64, sbase + A_BYTES + pl * B_PLANE + j * (NW * 1024));
        }
    };

    float* patch = reinterpret_cast<float*>(smem + (NSTAGE - 1) * STAGE_BYTES) + wave * (PR * EWN);
    const int col = lane & 31, rhalf = (lane >> 5) * 4;
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int act = p.act;

    bool have = id < id_end;
    if (have) {
        setup(id);
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nt) stage(s);
    }
    bool counted = false;
    while (have) {
        f32x16_p acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // K loop, software-pipelined around ONE barrier per K step: the fragments of half-step (t, 1) are read before the MFMAs of
        // (t, 0) and those of (t+1, 0) before the MFMAs of (t, 1); the wait + barrier that makes step t+1 visible (and frees the
        // slot of step t: every wave's reads of it have returned) sits between the two MFMA groups, where it costs nothing.
        bf16x8 fa0[NP][TM], fb0[NP][TN], fa1[NP][TM], fb1[NP][TN];
        auto read_frags = [&](int t, int s, bf16x8 (&fa)[NP][TM], bf16x8 (&fb)[NP][TN]) {
            const char* sb = smem + (t % NSTAGE) * STAGE_BYTES;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(sb + pl * A_PLANE + a_base + i * (32 * ROWB) + foff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(sb + pl * B_PLANE + b_base + j * (32 * ROWB) + foff[s]);
            }
        };
        auto mfma_all = [&](const bf16x8 (&fa)[NP][TM], const bf16x8 (&fb)[NP][TN]) {
            // smallest plane products first
#pragma unroll
            for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
                for (int pa = 0; pa < NP; ++pa) {
                    const int pb = sum - pa;
                    if (pb < 0 || pb >= NP) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
                }
        };
        if (nt > 0) {
            // step 0 has landed once at most min(nt, NSTAGE-1) - 1 younger steps (and the previous item's stores) are outstanding
            const int after0 = min(nt, NSTAGE - 1) - 1;
            if (counted && after0 == NSTAGE - 2) wait_vmcnt_p<(NSTAGE - 2) * LPS + STORES>();
            else if (NSTAGE >= 4 && after0 >= 2) wait_vmcnt_p<2 * LPS>();
            else if (NSTAGE >= 3 && after0 >= 1) wait_vmcnt_p<LPS>();
            else wait_vmcnt_p<0>();
            __builtin_amdgcn_s_barrier();
            read_frags(0, 0, fa0, fb0);
        }
        for (int t = 0; t < nt; ++t) {
            if (t + NSTAGE - 1 < nt) stage(t + NSTAGE - 1);      // into the slot of step t-1 (free since the previous mid-step barrier)
            read_frags(t, 1, fa1, fb1);
            mfma_all(fa0, fb0);
            if (t + 1 < nt) {
                const int after = min(nt - 1, t + NSTAGE - 1) - (t + 1);
                if (NSTAGE >= 4 && after >= 2) wait_vmcnt_p<2 * LPS>();
                else if (NSTAGE >= 3 && after >= 1) wait_vmcnt_p<LPS>();
                else wait_vmcnt_p<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of step t have returned
                __builtin_amdgcn_s_barrier();                    // step t+1 is visible to every wave; the slot of step t is free
                read_frags(t + 1, 0, fa0, fb0);
            }
            mfma_all(fa1, fb1);
        }

        const int e_m0 = m0, e_n0 = n0, e_slice = slice;
        id += id_step;
        have = id < id_end;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // every wave's fragment reads of the finished item have returned: the ring is free
        if (have) {
            setup(id);
#pragma unroll
            for (int s = 0; s < NSTAGE - 1; ++s)
                if (s < nt) stage(s);
        }

        // ---- epilogue ----
        const bool split = p.splitk > 1;                        // raw partial sums to part[slice][M][N] (fixed-order reduce pass follows)
        float* part = split ? p.part + (long)e_slice * p.M * p.N : nullptr;
        const long rowbase = (long)(e_m0 + wm * WTM);
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            const int n = e_n0 + wn * WTN + jb * EWN + ec;
            float bias[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bias[e] = (!split && p.bias && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int hb = 0; hb < NB; ++hb) {
                    const long mb = rowbase + i * 32 + hb * PR;
                    f32x4_p rres[PASSES][2];
                    if (EPI == 1) {
#pragma unroll
                        for (int ps = 0; ps < PASSES; ++ps) {
                            const long m = mb + ps * RPP + er;
                            rres[ps][0] = zero4p(); rres[ps][1] = zero4p();
                            if (!split && p.R && m < p.M && n < p.N) {
                                const float* rp = p.R + m * p.ldr + p.r_coff + n;
                                rres[ps][0] = *reinterpret_cast<const f32x4_p*>(rp);
                                rres[ps][1] = *reinterpret_cast<const f32x4_p*>(rp + 4);
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
                        f32x4_p v0 = *reinterpret_cast<const f32x4_p*>(patch + row * EWN + (((ec >> 2) ^ par) << 2));
                        f32x4_p v1 = *reinterpret_cast<const f32x4_p*>(patch + row * EWN + ((((ec >> 2) + 1) ^ par) << 2));
                        if (m >= p.M || n >= p.N) continue;
                        if (EPI == 1) {
                            if (split) {
                                float* dst = part + m * p.N + n;
                                *reinterpret_cast<f32x4_p*>(dst) = v0;
                                *reinterpret_cast<f32x4_p*>(dst + 4) = v1;
                                continue;
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] += bias[e]; v1[e] += bias[4 + e]; }
                            if (act == ACT_GELU_ERF) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = gelu_erf_p(v0[e]); v1[e] = gelu_erf_p(v1[e]); }
                            } else if (act == ACT_RELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                            }
                            v0 = v0 + rres[ps][0];
                            v1 = v1 + rres[ps][1];
                            float* dst = p.C + m * p.ldc + p.c_coff + n;
                            *reinterpret_cast<f32x4_p*>(dst) = v0;
                            *reinterpret_cast<f32x4_p*>(dst + 4) = v1;
                        } else {
                            // P layout out: the 8 columns are one 16-byte chunk per plane of K tile (c_coff + n) / 32
#pragma unroll
                            for (int e = 0; e < 4; ++e) { v0[e] += bias[e]; v1[e] += bias[4 + e]; }
                            if (act == ACT_GELU_ERF) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = gelu_erf_p(v0[e]); v1[e] = gelu_erf_p(v1[e]); }
                            } else if (act == ACT_RELU) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                            }
                            bf16x4 s0[NP], s1[NP];
                            split4<NP>(v0, 0xffffffffu, s0);
                            split4<NP>(v1, 0xffffffffu, s1);
                            const int cc = p.c_coff + n;
                            char* base = reinterpret_cast<char*>(p.C + m * p.ldc) + (cc >> 5) * (64 * NP) + (cc & 31) * 2;
#pragma unroll
                            for (int pl = 0; pl < NP; ++pl) {
                                typedef unsigned u32x2_p __attribute__((ext_vector_type(2)));
                                const u32x2_p lo = __builtin_bit_cast(u32x2_p, s0[pl]), hi = __builtin_bit_cast(u32x2_p, s1[pl]);
                                u32x4_p o = {lo[0], lo[1], hi[0], hi[1]};
                                *reinterpret_cast<u32x4_p*>(base + 64 * pl) = o;
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        counted = e_m0 + BM <= p.M && e_n0 + BN <= p.N;
    }
}

// split-K second pass (fixed-order sum of the slices + bias / activation / residual), fp32 out
__global__ void splitk_reduce_planes_kernel(const GemmParams p) {
    const long total = (long)p.M * p.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.part[(long)s * total + idx];
        if (p.bias) v += p.bias[n];
        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == ACT_GELU_ERF) v = gelu_erf_p(v);
        if (p.R) v += p.R[(long)m * p.ldr + p.r_coff + n];
        p.C[(long)m * p.ldc + p.c_coff + n] = v;
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int NP>
static hipError_t launch_planes_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
    constexpr int LDS_BYTES = NSTAGE * NP * (BM + BN) * 64;
    constexpr int WG_PER_CU = (160 * 1024 / LDS_BYTES) < 1 ? 1 : (160 * 1024 / LDS_BYTES);
    const int slots = 256 * WG_PER_CU;
    dim3 grid(tiles < slots ? tiles : slots), block(WM * WN * 64);   // persistent workgroups
    if (p.c_planes) hipLaunchKernelGGL((gemm_planes_kernel<BM, BN, WM, WN, NSTAGE, NP, 3>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_planes_kernel<BM, BN, WM, WN, NSTAGE, NP, 1>), grid, block, 0, s, p);
    return hipGetLastError();
}

// can this GEMM run on the plane kernel?  (dense, A in the P layout of the weights' plane count, vector-friendly C)
bool gemm_planes_eligible(const GemmParams& p) {
    if (p.mode != GEMM_DENSE || !p.Wp || !(p.planes == 2 || p.planes == 3) || p.a_planes != p.planes) return false;
    if (p.scale || p.bbias || p.a_coff || (p.K % 32) || (p.N % 8) || (p.lda % 16)) return false;
    if (p.c_planes) return p.c_planes == p.planes && !p.R && (p.N % 32) == 0 && (p.ldc % 16) == 0 && (p.c_coff % 32) == 0;
    return ((p.ldc | p.c_coff) & 3) == 0 && (!p.R || ((p.ldr | p.r_coff) & 3) == 0);
}

// cfg 0: 128x128, 8 waves, 3 ring slots (NP = 3: 144 KB).  Split-K when the grid would leave most CUs idle or a K loop is long.
GemmPlan plan_gemm_planes(int M, int N, int K, int planes, bool c_planes) {
    GemmPlan pl{0, 1, 0};
    const long tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
    const int nk = K / 32;
    if (!c_planes && tiles <= 256 && nk >= 48) {           // e.g. fc2 at batch 1: 240 tiles x 96 K steps -> 480 items of 48 steps
        int s = (int)(512 / tiles);
        if (s > nk / 24) s = nk / 24;
        if (s > 4) s = 4;
        if (s > 1) { pl.splitk = s; pl.ws_floats = (size_t)s * M * N; }
    }
    (void)planes;
    return pl;
}

hipError_t launch_gemm_planes(const GemmParams& p_in, const GemmPlan& pl, float* ws, hipStream_t s) {
    if (!gemm_planes_eligible(p_in) || p_in.M <= 0) return hipErrorInvalidValue;
    if (p_in.wp_rows < (p_in.N + 127) / 128 * 128) return hipErrorInvalidValue;
    GemmParams p = p_in;
    p.splitk = pl.splitk < 1 ? 1 : pl.splitk;
    p.part = ws;
    if (p.splitk > 1 && (!ws || p.c_planes)) return hipErrorInvalidValue;
    hipError_t e;
    static const int cfg_env = getenv("BRN_PLANES_CFG") ? atoi(getenv("BRN_PLANES_CFG")) : 0;      // tuning override
    if (p.planes == 3) {
        if (cfg_env == 1) e = launch_planes_cfg<128, 128, 2, 2, 3, 3>(p, s);
        else if (cfg_env == 2) e = launch_planes_cfg<128, 128, 4, 2, 2, 3>(p, s);
        else if (cfg_env == 3) e = launch_planes_cfg<128, 64, 2, 2, 2, 3>(p, s);       // 72 KB: two workgroups per CU
        else if (cfg_env == 4) e = launch_planes_cfg<64, 128, 2, 2, 2, 3>(p, s);
        else e = launch_planes_cfg<128, 128, 4, 2, 3, 3>(p, s);
    } else {
        if (cfg_env == 1) e = launch_planes_cfg<128, 128, 2, 2, 3, 2>(p, s);
        else e = launch_planes_cfg<128, 128, 4, 2, 3, 2>(p, s);
    }
    if (e != hipSuccess || p.splitk == 1) return e;
    long total = (long)p.M * p.N;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace brn
