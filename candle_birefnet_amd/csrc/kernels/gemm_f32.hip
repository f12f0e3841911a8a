// gemm_f32.hip — the one contraction kernel of the path: C = epilogue(A_gather[M,K] x W[N,K]^T) in exact fp32
// on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, 64 FLOP/clk/SIMD).
//
// It replaces every candle Linear / Conv2d the reference's hot path executes:
//   GEMM_DENSE        candle_nn::linear (swin.rs:98-99,130-131,487) and all 1x1 convs (decoder.rs:65, aspp.rs:271,282)
//   GEMM_CONV_NHWC    candle_nn::conv2d 3x3 / 7x7 (decoder.rs:44-45,104,113; aspp.rs:39-45; birefnet.rs:105)
//   GEMM_GATHER_NCHW  the two convs that read the NCHW image directly: PatchEmbed.proj 4x4 s4 (swin.rs:677) and
//                     ipt_blk1.conv1 3x3 (birefnet.rs:189)
//   GEMM_DEFORM_NHWC  the Metal path's deformable_im2col + matmul (aspp.rs:58-165) with the column matrix never
//                     materialised: the bilinear gather * modulator is the A-tile loader
// with bias / folded eval-BatchNorm / ReLU / erf-GELU / residual / concat-slice writes fused into the epilogue.
//
// Tiling (wave64): block tile BMxBN, BK = 32; WMxWN waves, each owning (BM/WM)x(BN/WN) as 32x32 MFMA tiles.
// The k index inside a BK tile is permuted: lane-half h of an MFMA step s contracts k = 16h + s, so a lane's
// sixteen A (or B) values of a tile are 16 consecutive floats of one LDS row = four ds_read_b128 (row stride
// 36 floats: conflict-free for the b128 lane groups).  Global->LDS goes through registers (one float4 per
// thread per 32 rows) with the next tile's loads in flight during the current tile's 64-cycle MFMAs.
#include <cstdlib>
#include "../brn_kernels.h"
#include "split_planes.h"

namespace brn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
// a load the optimiser may not sink under the predicate that later selects its value: hipcc turns `ok ? *p : 0` (and
// `t = *p; ok ? t : 0`) into an exec-branch around the load, and then waits vmcnt(0) before the first use of ANY staged
// register in the loop, draining the prefetched K tiles every iteration.
// So: load from a clamped (always valid) address, remember a ~0 / 0 bit mask, and AND the value with it when the staged
// registers are consumed (at the LDS store) — never a select at the load, never an operation on the data at the load (that
// would wait for it right there).  An integer AND, not a multiply by 0/1: the clamped address holds real data (pixel (0,0) of
// the window, row 0), and Inf * 0 = NaN would leak a non-finite input into every zero-padded border output.
__device__ __forceinline__ f32x4 load4_masked(const float* ptr, bool ok, unsigned& keep) {
    keep = ok ? 0xffffffffu : 0u;
    return *reinterpret_cast<const f32x4*>(ptr);
}
__device__ __forceinline__ f32x4 and4(const f32x4 v, const unsigned keep) {
    typedef unsigned u32x4_m __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(u32x4_m, v) & keep);
}

constexpr int BK = 32;
constexpr int LDS_LD = 36;

// x.gelu_erf() (candle: 0.5 x (1 + erf(x / sqrt 2)), swin.rs:103), branch-free and with ONE transcendental (round 4):
//     gelu(x) = relu(x) - |x| h(|x|),   h(u) = erfc(u / sqrt 2) / 2 = 2^P(u),
// P a degree-7 polynomial — log2 of the Gaussian tail is nearly a parabola, and one v_exp_f32 undoes it.  1 clamp + 7 fma + v_exp + max
// + fma = 14 issue slots; the Abramowitz-Stegun form this replaces (erfc(s) = t (c1 + t (...)) exp(-s^2), t = 1 / (1 + p s): v_rcp + v_exp + 7
// fma + a select, ~26 slots) cost the fc1 epilogues 6 us per 5120 x 3072 GEMM, libm's erff 18.  Neither side of zero cancels: for x >= 0 the
// result is x minus a term <= 0.17.  Coefficients: weighted least squares on [0, 8] against scipy's erfc, rounded to fp32, and the whole
// form re-evaluated in emulated fp32 on 3e6 points of [-60, 60]: |gelu error| < 4.9e-7 for |x| <= 6 (the fp32 rounding of the result
// itself is 2.4e-7 there; the old form measured 6.1e-7 the same way), half an ulp of x beyond.  u is clamped at 8: |x| 2^P(8) < 1e-6 |x| 2^-29.
__device__ __forceinline__ float gelu_erf(float x) {
    float u, r;
    asm("v_min_f32 %0, |%1|, %2" : "=v"(u) : "v"(x), "v"(8.0f));         // (plain v_min / v_max: fminf / fmaxf put a canonicalising
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(0.0f));           // v_max x, x in front of each; x is an MFMA / fma result, never signalling)
    float p = 2.0329723611212103e-06f;
    p = fmaf(p, u, 1.31221850097063e-05f);
    p = fmaf(p, u, -0.0006936025456525385f);
    p = fmaf(p, u, 0.007940512150526047f);
    p = fmaf(p, u, -0.05327853187918663f);
    p = fmaf(p, u, -0.45883336663246155f);
    p = fmaf(p, u, -1.1511898040771484f);
    p = fmaf(p, u, -0.9999935030937195f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), r);
}

// tile id -> (m tile, n tile): N is walked in groups of GN tile columns, M fastest-but-one inside a group, so that while an XCD
// marches down M the GN weight panels of the group stay in its 4 MiB L2 and every A panel is fetched once per group
// (the split-bf16 kernels are bound by L2-miss traffic, not by the matrix pipe).
__device__ __forceinline__ void tile_coords(int tile, int tilesM, int tilesN, int& tm, int& tn) {
    constexpr int GN = 8;
    const int per_group = tilesM * GN;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gw = min(GN, tilesN - g * GN);
    tm = r / gw;
    tn = g * GN + (r - tm * gw);
}

// ---- epilogue shared by the fp32-MFMA and the split-bf16 kernels (same 32x32 C/D register map) ----
// Each 32x32 accumulator tile is transposed through a wave-private LDS patch (rows of 36 floats) so that the global side
// is row-major float4: 8 lanes cover one 128-byte row segment, residual / per-image-bias loads and the stores are 16 B
// per lane, and only one float4 of temporaries is live per lane.  The uniform switches are taken outside the element loops.
constexpr int EPI_LD = 36;
constexpr int EPI_WAVE_FLOATS = 32 * EPI_LD;

__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    if (act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (act == ACT_GELU_ERF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
    }
    return v;
}

template <int TM, int TN, int WTM, int WTN>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane,
                                              int slice, float* patch /* EPI_WAVE_FLOATS floats private to this wave */) {
    const int col = lane & 31, rhalf = (lane >> 5) * 4;      // C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int rrow = lane >> 3, c4 = (lane & 7) * 4;         // read-back map: 8 rows x 8 float4 per pass
    const bool split = p.splitk > 1;
    float* part = split ? p.part + (long)slice * p.M * p.N : nullptr;
    const bool vec = split ? ((p.N & 3) == 0)
                           : (((p.N | p.ldc | p.c_coff) & 3) == 0 && (!p.R || ((p.ldr | p.r_coff) & 3) == 0));
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 32 + c4;
        f32x4 bias = zero4(), sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4();
        if (!split) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e < p.N) {
                    if (p.bias) bias[e] = p.bias[n + e];
                    if (p.scale) { sc[e] = p.scale[n + e]; sh[e] = p.shift[n + e]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + rhalf) * EPI_LD + col] = acc[i][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // same-wave LDS ops complete in order; make it explicit
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int row = ps * 8 + rrow;
                const int m = m0 + wm * WTM + i * 32 + row;
                f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * EPI_LD + c4);
                if (m >= p.M || n >= p.N) continue;
                if (p.h2) v = v * p.out_scale;           // mode f32_half2: the operands were scaled by powers of two (exact)
                if (split) {
                    float* dst = part + (long)m * p.N + n;
                    if (vec) *reinterpret_cast<f32x4*>(dst) = v;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) dst[e] = v[e];
                    }
                    continue;
                }
                v = v + bias;
                if (p.bbias) {
                    const float* bp = p.bbias + (long)(m / p.bbias_rows) * p.N + n;
                    if (vec) v = v + *reinterpret_cast<const f32x4*>(bp);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bp[e];
                    }
                }
                if (p.scale) v = v * sc + sh;
                v = act4(v, p.act);
                float* dst = p.C + (long)m * p.ldc + p.c_coff + n;
                if (p.c_bf16 == 2) {            // compute mode BRN_F16: fp16 map out
                    _Float16* db = reinterpret_cast<_Float16*>(p.C) + (long)m * p.ldc + p.c_coff + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) db[e] = (_Float16)v[e];
                } else if (p.c_bf16) {          // compute mode BRN_BF16 (deformable gather convs): bf16 map out, no residual on this path
                    __bf16* db = reinterpret_cast<__bf16*>(p.C) + (long)m * p.ldc + p.c_coff + n;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.N) db[e] = (__bf16)v[e];
                } else if (vec) {
                    if (p.R) v = v + *reinterpret_cast<const f32x4*>(p.R + (long)m * p.ldr + p.r_coff + n);
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N) {
                            float t = v[e];
                            if (p.R) t += p.R[(long)m * p.ldr + p.r_coff + n + e];
                            dst[e] = t;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the patch is rewritten by the next tile
        }
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ void __launch_bounds__(WM* WN * 64) gemm_f32_kernel(const GemmParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;  // tile rows covered by one pass of float4 loads (8 float4 = one 32-float row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile too small for the thread count");

    constexpr int SMEM_MAIN = (BM + BN) * LDS_LD, SMEM_EPI = WM * WN * EPI_WAVE_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
    float* As = smem;
    float* Bs = smem + BM * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: blocks that share an XCD (same blockIdx % 8) walk a contiguous run of tiles,
    // n fastest, so an A panel is fetched into that XCD's L2 once for all its N tiles (bijective remap).
    const int tilesN = (p.N + BN - 1) / BN;
    int swz;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    // split-K: grid = tiles x splitk; slice s of a tile contracts k-tiles [s*kts, (s+1)*kts) and writes raw partial sums
    const int ntiles = tilesN * ((p.M + BM - 1) / BM);
    const int slice = swz / ntiles, tile = swz - slice * ntiles;
    int tile_m, tile_n;
    tile_coords(tile, (p.M + BM - 1) / BM, tilesN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kq = tid & 7, lrow = tid >> 3;

    // ---- per-row gather state (fixed over the K loop) ----
    long a_base[PA];
    int a_iy[PA], a_ix[PA];
    bool a_ok[PA];
    long om_base[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + i * RPP;
        a_ok[i] = m < p.M;
        a_iy[i] = 0; a_ix[i] = 0; om_base[i] = 0;
        if (MODE == GEMM_DENSE) {
            a_base[i] = (long)m * p.lda;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            if (MODE == GEMM_GATHER_NCHW) a_base[i] = (long)b * p.Cin * p.Hin * p.Win;
            else a_base[i] = (long)b * p.Hin * p.Win * p.lda + p.a_coff;
            om_base[i] = (long)m * p.om_ld;
        }
    }
    const float* wrow = p.W + (long)(n0 + lrow) * p.K + kq * 4;

    f32x4 ra[PA], rb[PB];
    unsigned am[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) am[i] = 0xffffffffu;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(wrow + (long)i * RPP * p.K + k0);
        if (MODE == GEMM_DENSE) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                {   // unconditional load from a clamped address + select: no exec branch, so the compiler keeps counted vmcnt waits
                    ra[i] = load4_masked(p.A + (a_ok[i] ? a_base[i] : 0) + k0 + kq * 4, a_ok[i], am[i]);
                }
        } else if (MODE == GEMM_CONV_NHWC) {
            const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const long off = ok ? a_base[i] + ((long)iy * p.Win + ix) * p.lda + ci0 : (long)p.a_coff;
                ra[i] = load4_masked(p.A + off + kq * 4, ok, am[i]);
            }
        } else if (MODE == GEMM_GATHER_NCHW && p.kw == 4 && p.stride == 4 && p.pad == 0 && p.dil == 1 && (p.Win & 3) == 0 && (reinterpret_cast<unsigned long long>(p.A) & 15) == 0) {
            // PatchEmbed (swin.rs:677: k 4, s 4, no padding) on an image whose width is a multiple of 4: a lane's four k indices
            // k4 .. k4 + 3 are the four kx of one (channel, ky) = 16 contiguous bytes of the image: one unpredicated float4 load from a
            // clamped address + the bit mask (the per-element form below costs 4 predicated scalar loads and 4 index divisions)
            const int k4 = k0 + kq * 4, khw = p.kh * 4;
            const int c = k4 / khw, ky = (k4 - c * khw) >> 2;
            const bool kin = k4 < p.Kreal;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int iy = a_iy[i] + ky;
                const bool ok = a_ok[i] && kin && iy < p.Hin;              // (bottom rows of a height that is not a multiple of 4 are zero padding)
                const long off = ok ? a_base[i] + ((long)c * p.Hin + iy) * p.Win + a_ix[i] : 0;
                ra[i] = load4_masked(p.A + off, ok, am[i]);
            }
        } else if (MODE == GEMM_GATHER_NCHW) {
            const int khw = p.kh * p.kw;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + kq * 4 + e;
                    const int c = k / khw, rem = k - c * khw;
                    const int ky = rem / p.kw, kx = rem - ky * p.kw;
                    const int iy = a_iy[i] + ky * p.dil, ix = a_ix[i] + kx * p.dil;
                    const bool ok = a_ok[i] && k < p.Kreal && (unsigned)iy < (unsigned)p.Hin &&
                                    (unsigned)ix < (unsigned)p.Win;
                    v[e] = ok ? p.A[a_base[i] + ((long)c * p.Hin + iy) * p.Win + ix] : 0.f;
                }
                { f32x4 t = {v[0], v[1], v[2], v[3]}; ra[i] = t; }
            }
        } else {  // GEMM_DEFORM_NHWC: torchvision deform_conv2d sampling, 1 offset group, modulated
            const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                f32x4 r = zero4();
                if (a_ok[i]) {
                    const float* omr = p.om + om_base[i];
                    const float offy = omr[2 * tap], offx = omr[2 * tap + 1];
                    const float mk = omr[p.om_mask_off + tap];
                    const float y = (float)(a_iy[i] + ky * p.dil) + offy;
                    const float x = (float)(a_ix[i] + kx * p.dil) + offx;
                    if (y > -1.f && y < (float)p.Hin && x > -1.f && x < (float)p.Win) {
                        const int yl = (int)floorf(y), xl = (int)floorf(x);
                        const int yh = yl + 1, xh = xl + 1;
                        const float ly = y - (float)yl, lx = x - (float)xl;
                        const float hy = 1.f - ly, hx = 1.f - lx;
                        const long boff = a_base[i] + ci0 + kq * 4;
                        // compute mode BRN_BF16: the sampled map is bf16 (lda / a_coff in elements either way)
                        auto tap4 = [&](int yy, int xx) -> f32x4 {
                            const long o = boff + ((long)yy * p.Win + xx) * p.lda;
                            if (p.a_bf16 == 2) {               // compute mode BRN_F16: an fp16 map
                                typedef _Float16 f16x4_g __attribute__((ext_vector_type(4)));
                                const f16x4_g h = *reinterpret_cast<const f16x4_g*>(reinterpret_cast<const _Float16*>(p.A) + o);
                                f32x4 r4 = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                                return r4;
                            }
                            if (p.a_bf16) {
                                const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.A) + o);
                                f32x4 r4 = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
                                return r4;
                            }
                            return *reinterpret_cast<const f32x4*>(p.A + o);
                        };
                        f32x4 v1 = zero4(), v2 = v1, v3 = v1, v4 = v1;
                        if (yl >= 0 && xl >= 0) v1 = tap4(yl, xl);
                        if (yl >= 0 && xh <= p.Win - 1) v2 = tap4(yl, xh);
                        if (yh <= p.Hin - 1 && xl >= 0) v3 = tap4(yh, xl);
                        if (yh <= p.Hin - 1 && xh <= p.Win - 1) v4 = tap4(yh, xh);
                        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                        r = mk * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    }
                }
                ra[i] = r;
            }
        }
    };
    auto lds_store = [&]() {
#pragma unroll
        for (int i = 0; i < PA; ++i)
            *reinterpret_cast<f32x4*>(As + (lrow + i * RPP) * LDS_LD + kq * 4) = and4(ra[i], am[i]);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            *reinterpret_cast<f32x4*>(Bs + (lrow + i * RPP) * LDS_LD + kq * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K / BK;
    const int kts = (nk_all + p.splitk - 1) / p.splitk;
    const int kt0 = slice * kts, nk = min(nk_all, kt0 + kts);
    const float* a_frag = As + (wm * WTM + (lane & 31)) * LDS_LD + (lane >> 5) * 16;
    const float* b_frag = Bs + (wn * WTN + (lane & 31)) * LDS_LD + (lane >> 5) * 16;

    if (kt0 < nk) {
        gload(kt0);
        lds_store();
    }
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int hs = 0; hs < 2; ++hs) {
            float af[TM][8], bf[TN][8];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDS_LD + hs * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDS_LD + hs * 8 + 4);
                af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
                af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDS_LD + hs * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDS_LD + hs * 8 + 4);
                bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
                bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) {
            lds_store();
            __syncthreads();
        }
    }

    gemm_epilogue<TM, TN, WTM, WTN>(p, acc, m0, n0, wm, wn, lane, slice, smem + wave * EPI_WAVE_FLOATS);
}

// =====================================================================================================================
// gemm_split_kernel — the same contraction on the bf16 matrix cores with fp32-class accuracy.
// Every fp32 operand x is split error-free into NP bf16 planes (x = x_h + x_m + x_l up to 2^-25 |x|: each plane is the
// round-to-nearest bf16 of what the previous planes left over), the product is the sum of the plane products whose
// magnitude is >= 2^-24 of the full product (NP = 3: hh, hm, mh, hl, lh, mm — 6 x v_mfma_f32_32x32x16_bf16), each exact in
// the MFMA's fp32 accumulator.  6 bf16 MFMAs replace 8 fp32 MFMAs (k = 16 vs 2) at 16x the per-instruction rate: 2.67x the
// fp32-MFMA peak.  A is split while it is staged (fp32 in HBM, no second copy); W is pre-split at load time ([NP][Npad][K]).
// NP = 2 keeps hh, hm, mh (~2^-16 relative); NP = 1 is plain bf16 x bf16 -> fp32 (the bf16 throughput mode).
// LDS: per plane [rows][40 bf16] (80-byte rows: conflict-free for the ds_read_b128 lane groups).
// =====================================================================================================================
constexpr int SLD = 40;   // bf16 elements per LDS row

template <int BM, int BN, int WM, int WN, int MODE, int NP, bool H = false>   // H: the two planes are fp16 (mode f32_half2)
__global__ void __launch_bounds__(WM* WN * 64) gemm_split_kernel(const GemmParams p) {
    static_assert(!H || NP == 2, "fp16 planes come in pairs");
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;           // A rows per pass (8 float4 per 32-float row)
    constexpr int PA = BM / RPP;
    constexpr int WRPP = NT / 4;          // W rows per pass (4 x 16-byte chunks per 32-bf16 row)
    constexpr int PB = BN / WRPP;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile too small for the thread count");

    constexpr int SMEM_MAIN = NP * (BM + BN) * SLD * 2, SMEM_EPI = WM * WN * EPI_WAVE_FLOATS * 4;   // bytes
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);
    __bf16* As = smem;                       // [NP][BM][SLD]
    __bf16* Bs = smem + NP * BM * SLD;       // [NP][BN][SLD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tilesN = (p.N + BN - 1) / BN;
    int swz;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ntiles = tilesN * ((p.M + BM - 1) / BM);
    const int slice = swz / ntiles, tile = swz - slice * ntiles;
    int tile_m, tile_n;
    tile_coords(tile, (p.M + BM - 1) / BM, tilesN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kq = tid & 7, lrow = tid >> 3;
    long a_base[PA];
    int a_iy[PA], a_ix[PA];
    bool a_ok[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + i * RPP;
        a_ok[i] = m < p.M;
        a_iy[i] = 0; a_ix[i] = 0;
        if (MODE == GEMM_DENSE) {
            a_base[i] = (long)m * p.lda;
        } else {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
            a_iy[i] = oy * p.stride - p.pad;
            a_ix[i] = ox * p.stride - p.pad;
            a_base[i] = (long)b * p.Hin * p.Win * p.lda + p.a_coff;
        }
    }
    const int wc = tid & 3, wrow = tid >> 2;
    // W planes are interleaved per 32-deep K tile: [row][K/32][plane][32] bf16, so the NP x 64 bytes a (row, K tile) needs
    // are contiguous (NP = 2: exactly one 128-byte line; separate planes fetched every line twice: measured 2x L2->L1 traffic)
    const long wrow_stride = (long)p.K * NP;
    const __bf16* wsrc = reinterpret_cast<const __bf16*>(p.Wp) + (long)(n0 + wrow) * wrow_stride + wc * 8;

    // two staging register sets: tile kt+2 is already in flight while tile kt is multiplied (bytes in flight per CU, not
    // bandwidth, bound this kernel: a bf16-rate K tile lasts a few hundred cycles, an L2/HBM round trip ~1-2 thousand)
    f32x4 ra[2][PA];
    bf16x8 rb[2][NP][PB];
    unsigned am[2][PA];

    auto gload = [&](int kt, f32x4 (&qa)[PA], bf16x8 (&qb)[NP][PB], unsigned (&qm)[PA]) {
        const int k0 = kt * BK;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                qb[pl][i] = *reinterpret_cast<const bf16x8*>(wsrc + (long)i * WRPP * wrow_stride + (long)kt * (NP * 32) + pl * 32);
        if (MODE == GEMM_DENSE) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                {
                    qa[i] = load4_masked(p.A + (a_ok[i] ? a_base[i] : 0) + k0 + kq * 4, a_ok[i], qm[i]);
                }
        } else {
            const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
            const int ky = tap / p.kw, kx = tap - ky * p.kw;
            const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const long off = ok ? a_base[i] + ((long)iy * p.Win + ix) * p.lda + ci0 : (long)p.a_coff;
                qa[i] = load4_masked(p.A + off + kq * 4, ok, qm[i]);
            }
        }
    };
    auto lds_store = [&](const f32x4 (&qa)[PA], const bf16x8 (&qb)[NP][PB], const unsigned (&qm)[PA]) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 sp[NP];
            if constexpr (H) split4h<true>(qa[i], qm[i], p.a_scale, sp); else split4<NP>(qa[i], qm[i], sp);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *reinterpret_cast<bf16x4*>(As + (pl * BM + lrow + i * RPP) * SLD + kq * 4) = sp[pl];
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                *reinterpret_cast<bf16x8*>(Bs + (pl * BN + wrow + i * WRPP) * SLD + wc * 8) = qb[pl][i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K / BK;
    const int kts = (nk_all + p.splitk - 1) / p.splitk;
    const int kt0 = slice * kts, nk = min(nk_all, kt0 + kts);
    // MFMA 32x32x16 bf16 operand map: lane l holds row (l & 31), k = 8 * (l >> 5) + j, j = 0..7
    const __bf16* a_frag = As + (wm * WTM + (lane & 31)) * SLD + (lane >> 5) * 8;
    const __bf16* b_frag = Bs + (wn * WTN + (lane & 31)) * SLD + (lane >> 5) * 8;

    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(a_frag + (pl * BM + i * 32) * SLD + ks * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(b_frag + (pl * BN + j * 32) * SLD + ks * 16);
            }
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
                            if constexpr (H) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[pa][i]), __builtin_bit_cast(f16x8, bf[pb][j]), acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], acc[i][j], 0, 0, 0);
                }
        }
    };

    if (kt0 < nk) {
        gload(kt0, ra[0], rb[0], am[0]);
        if (kt0 + 1 < nk) gload(kt0 + 1, ra[1], rb[1], am[1]);
        lds_store(ra[0], rb[0], am[0]);
    }
    __syncthreads();
    // body for one K tile whose successor sits in register set NXT; the set just consumed (CUR) is refilled 2 tiles ahead
#define BRN_SPLIT_STEP(KT, CUR, NXT)                                  \
    {                                                                 \
        if ((KT) + 2 < nk) gload((KT) + 2, ra[CUR], rb[CUR], am[CUR]);         \
        compute();                                                    \
        __syncthreads();                                              \
        if ((KT) + 1 < nk) {                                          \
            lds_store(ra[NXT], rb[NXT], am[NXT]);            \
            __syncthreads();                                          \
        }                                                             \
    }
    for (int kt = kt0; kt < nk; kt += 2) {
        BRN_SPLIT_STEP(kt, 0, 1)
        if (kt + 1 < nk) BRN_SPLIT_STEP(kt + 1, 1, 0)
    }
#undef BRN_SPLIT_STEP
    gemm_epilogue<TM, TN, WTM, WTN>(p, acc, m0, n0, wm, wn, lane, slice, reinterpret_cast<float*>(smem_raw) + wave * EPI_WAVE_FLOATS);
}

// Workgroup-wide epilogue of the warp-specialised kernels: the C tile sits row-major in LDS (ld floats per row); 512 threads,
// thread t owns columns 4*(t&31).. of rows (t>>5) + 16*pass.  Same arithmetic, in the same order, as gemm_epilogue.
template <int BM, int BN, int LD>
__device__ __forceinline__ void gemm_epilogue_tile(const GemmParams& p, const float* ctile, int m0, int n0, int tid, int slice) {
    static_assert(BN == 128, "32 float4 columns per row");
    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
    const int n = n0 + c4;
    if (n >= p.N) return;
    const bool split = p.splitk > 1;
    float* part = split ? p.part + (long)slice * p.M * p.N : nullptr;
    const bool vec = split ? ((p.N & 3) == 0)
                           : (((p.N | p.ldc | p.c_coff) & 3) == 0 && (!p.R || ((p.ldr | p.r_coff) & 3) == 0));
    f32x4 bias = zero4(), sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4();
    if (!split) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e < p.N) {
                if (p.bias) bias[e] = p.bias[n + e];
                if (p.scale) { sc[e] = p.scale[n + e]; sh[e] = p.shift[n + e]; }
            }
        }
    }
#pragma unroll 2
    for (int ps = 0; ps < BM / 16; ++ps) {
        const int row = ps * 16 + r0;
        const int m = m0 + row;
        if (m >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(ctile + row * LD + c4);
        if (p.h2) v = v * p.out_scale;
        if (split) {
            float* dst = part + (long)m * p.N + n;
            if (vec) *reinterpret_cast<f32x4*>(dst) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < p.N) dst[e] = v[e];
            }
            continue;
        }
        v = v + bias;
        if (p.bbias) {
            const float* bp = p.bbias + (long)(m / p.bbias_rows) * p.N + n;
            if (vec) v = v + *reinterpret_cast<const f32x4*>(bp);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < p.N) v[e] += bp[e];
            }
        }
        if (p.scale) v = v * sc + sh;
        v = act4(v, p.act);
        float* dst = p.C + (long)m * p.ldc + p.c_coff + n;
        if (p.c_planes) {                   // the next GEMM reads the P layout (launch_gemm checked N, c_coff % 32 == 0, no R)
            if (p.h2) store_planes_h(p.C + (long)m * p.ldc, p.c_coff + n, v, p.a_scale);     // (the next GEMM's A scale is this one's: one scale per model)
            else store_planes_n(p.c_planes, p.C + (long)m * p.ldc, p.c_coff + n, v);
        } else if (vec) {
            if (p.R) v = v + *reinterpret_cast<const f32x4*>(p.R + (long)m * p.ldr + p.r_coff + n);
            *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e < p.N) {
                    float t = v[e];
                    if (p.R) t += p.R[(long)m * p.ldr + p.r_coff + n + e];
                    dst[e] = t;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_split_ws_kernel — warp-specialised form of gemm_split_kernel for the 128x128 tile: 8 waves, two per SIMD.
// Waves 0-3 (consumers) own the 2x2 grid of 64x64 sub-tiles: fragment reads + MFMA only.  Waves 4-7 (producers) stage:
// global loads four K tiles ahead (two register sets), operand split, LDS writes into a ring of NBUF = 3 K-tile buffers,
// two tiles ahead of the consumers.  ONE workgroup barrier per K tile.  Because tile kt+1 is already complete while tile kt
// is multiplied, a consumer prefetches the next tile's first fragments before the barrier and double-buffers fragments in
// registers: at bf16 MFMA rates an exposed LDS read (~250 cycles) per 32-deep K tile (768 MFMA cycles) was a third of the
// loop (measured by ablation: staging and MFMA phases added up, then the fragment-read stall did).
// ---------------------------------------------------------------------------------------------------------------------
template <int MODE, int NP, int KS, bool DIAG, bool APL, bool H = false>   // APL: A arrives in the P2 layout (its producer already split it): the staging waves only copy
// H: the two planes are fp16 planes of the scaled operands (mode f32_half2; split4h / v_mfma_f32_32x32x16_f16), same bytes and layouts
// DIAG: ablation switches + per-K-tile cycle stamps (brn_gemm_microbench only; costs registers)
// KS = k elements per LDS stage (32, or 16 to halve the stage when 3 planes must fit twice per CU)
#ifndef BRN_WS_M16
#define BRN_WS_M16 1          // 2-plane, 32-deep stages: the consumers issue 16 x 16 x 32 MFMAs (0: 32 x 32 x 16, same-box A/B builds)
#endif
__global__ void __launch_bounds__(512) gemm_split_ws_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 128, WTM = 64, WTN = 64, TM = 2, TN = 2;
    static_assert(!APL || ((NP == 2 || NP == 3) && MODE == GEMM_DENSE), "the P input layout is the NP-plane split of a dense A");
    static_assert(!H || NP == 2, "fp16 planes come in pairs");
    // LDS rows (round 4): UNPADDED KS-element rows with the 16-byte chunks of a row XOR-permuted by the row — key (row >> 3) & 1 for the
    // 32-byte rows of a 16-deep stage (8 rows per 256-byte bank row), (row >> 2) & 3 for 64-byte rows (4 per bank row).  A ds_read_b128
    // lane group (16 consecutive rows, one logical chunk) then touches 16 different 16-byte slots, AND a producer store instruction
    // (8-byte pieces: 4 or 8 lanes per row, 32 lanes = 8 or 4 whole rows) covers one bank row exactly once.  The padded rows this
    // replaces (KS + 8 elements: 48 / 80 bytes) were conflict-free for the reads only: the producers' ds_write_b64 halves wrapped onto
    // banks of the first rows (rows 0 / 5, 1 / 6, 2 / 7 of a 32-lane half at 48 bytes) — the 4 % SQ_LDS_BANK_CONFLICT of
    // profiles/r03_pmc_sq_c2_f32_split3.csv, paid by the staging waves, which are this kernel's critical path.  BRN_WS_SWZ=0 builds the
    // padded layout (same-box A/B of two libraries: tools/ab_lib.sh).
#ifndef BRN_WS_SWZ
#define BRN_WS_SWZ 1
#endif
    constexpr bool SWZ = BRN_WS_SWZ != 0;
    constexpr int SLD = SWZ ? KS : KS + 8;          // bf16 per LDS row
    auto swz_key = [](int row) { return SWZ ? (KS == 16 ? (row >> 3) & 1 : (row >> 2) & 3) : 0; };
    constexpr int KSTEPS = KS / 16;                 // MFMA k-steps per stage
    constexpr int NBUF = 2;                         // 2 x NP x 20 KB: two workgroups per CU at NP <= 2 (a 3-deep ring was slower: 1 WG/CU exposes each tile's prologue + epilogue)
    constexpr int AQ = KS / 4, RPP = 256 / AQ;                  // producers: 256 threads, AQ float4 per KS-float row
    constexpr int PA = APL ? 2 * NP : BM / RPP;                 // P-layout input: 4 NP 16-byte chunks per (row, K tile), 128 rows / 256 threads
    constexpr int WQ = KS / 8, WRPP = 256 / WQ, PB = BN / WRPP; // WQ 16-byte chunks per KS-bf16 row
    // P-layout input: ONE ds_write_b128 instruction covers both planes of a row (lanes c = 0..3 plane 0, 4..7 plane 1), and BM x SLD x 2 bytes is a
    // multiple of the 256-byte bank row: the two planes of a row would sit on the same banks (2-way conflict on every staging write: 2.5 % of
    // wave cycles in profiles/r04_pmc_sq_c2_f32_half2.csv).  Plane p of A is therefore shifted by p x 128 bytes: rows r, r + 1 of both planes then
    // cover the four 64-byte quarters of a bank row.  (The fragment reads stay conflict-free: a constant shift per plane.)
#ifndef BRN_APL_PAD
#define BRN_APL_PAD 1
#endif
    constexpr int APAD = (BRN_APL_PAD && APL && KS == 32) ? 64 : 0;            // elements (BRN_APL_PAD=0 builds the unshifted layout: tools/ab_lib.sh)
    constexpr int AREG = NP * BM * SLD + (NP - 1) * APAD;       // A region of a buffer
    constexpr int BUF = AREG + NP * BN * SLD;       // bf16 elements per LDS buffer
    constexpr int EP_LD = BN + 4;                   // floats per row of the epilogue's LDS image of the C tile
    constexpr int SMEM_MAIN = NBUF * BUF * 2, SMEM_EPI = BM * EP_LD * 4;   // bytes
    __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI];
    __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int abl = DIAG ? p.abl : 0;
    const bool producer = wave >= 4;
    const int tilesN = (p.N + BN - 1) / BN;
    int swz;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int ntiles = tilesN * ((p.M + BM - 1) / BM);
    const int slice = swz / ntiles, tile = swz - slice * ntiles;
    int tile_m, tile_n;
    tile_coords(tile, (p.M + BM - 1) / BM, tilesN, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nk_all = p.K / KS;
    const int kts = (nk_all + p.splitk - 1) / p.splitk;
    const int kt0 = slice * kts, nk = min(nk_all, kt0 + kts);
    const int nt = nk > kt0 ? nk - kt0 : 0;         // K tiles of this slice; local tile index t = kt - kt0

    unsigned long long* trc = (DIAG && p.trace) ? p.trace + (long)blockIdx.x * 256 : nullptr;
    if (trc && (tid == 0 || tid == 256)) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trc[(tid >> 8) * 8 + 0] = clock64();
        trc[(tid >> 8) * 8 + 1] = wall_clock64();
        trc[(tid >> 8) * 8 + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
    f32x16 acc[TM][TN];
    if (producer) {
        const int pt = tid - 256;
        const int kq = pt % AQ, lrow = pt / AQ;
        long a_base[PA];
        int a_iy[PA], a_ix[PA];
        bool a_ok[PA];
        int p_lds[PA];          // P-layout input: LDS element offset of this thread's i-th chunk (plane, row, 8-element column)
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int m = m0 + lrow + i * RPP;
            p_lds[i] = 0;
            if (APL) {
                const int q = i * 256 + pt, row = q / (4 * NP), c = q - row * (4 * NP);   // chunk c of the row: plane c / 4, k = 8 (c % 4)
                m = m0 + row;
                p_lds[i] = ((c >> 2) * BM + row) * SLD + (c >> 2) * APAD + ((c & 3) ^ swz_key(row)) * 8;
            }
            a_ok[i] = m < p.M;
            a_iy[i] = 0; a_ix[i] = 0;
            if (APL) {
                const int q = i * 256 + pt, row = q / (4 * NP), c = q - row * (4 * NP);
                a_base[i] = (long)m * p.lda + c * 4;          // + 16 NP kt floats per K tile (gload)
            } else if (MODE == GEMM_DENSE) {
                a_base[i] = (long)m * p.lda;
            } else {
                const int hw = p.Hout * p.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                a_iy[i] = oy * p.stride - p.pad;
                a_ix[i] = ox * p.stride - p.pad;
                a_base[i] = (long)b * p.Hin * p.Win * p.lda + p.a_coff;
            }
        }
        const int wc = pt % WQ, wrow = pt / WQ;
        // element offsets inside an LDS row of this thread's pieces (RPP and WRPP are multiples of 32 rows: the key is that of lrow / wrow)
        const int a_sw = ((kq >> 1) ^ swz_key(lrow)) * 8 + (kq & 1) * 4;
        const int w_sw = (wc ^ swz_key(wrow)) * 8;
        static_assert(KS == 32 || KS == 16, "the interleaved W plane layout is per 32-deep K tile; a 16-deep stage takes one half of it");
        static_assert(!APL || KS == 32, "P-layout input is staged in whole 32-deep K tiles");
        const long wrow_stride = (long)p.K * NP;         // W planes interleaved per K tile: [row][K/32][plane][32] bf16
        const __bf16* wsrc = reinterpret_cast<const __bf16*>(p.Wp) + (long)(n0 + wrow) * wrow_stride + wc * 8;
        // Dense operands are buffer-addressed: a resource per tile (A: based at row m0, num_records = the tile's valid rows, so rows
        // >= M come back as zeros without a mask; W planes: based at row n0), a 32-bit lane offset fixed for the tile, the K tile in the
        // instruction's SGPR offset.  The producers share their SIMDs with the MFMA waves: the 64-bit per-lane address arithmetic and
        // the row mask were ~5 VALU per load, a quarter of the producers' vector work per K tile.
        // The implicit-GEMM form: the resource is based at the first image of the tile, a lane's pixel offset is fixed for the tile, the
        // K tile's (tap, channel) offset is uniform (Cin % KS == 0: a K tile lies inside one tap) and is added to it; a tap outside the
        // image gets an offset past num_records, which the buffer unit answers with zeros.  The P-layout form (APL) is the dense one
        // with 16 NP floats per K tile.
        constexpr bool BUFA = MODE == GEMM_DENSE;                       // dense (plain or P-layout rows): valid-row num_records, no mask
        constexpr bool BUFC = MODE == GEMM_CONV_NHWC;
        typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
        const int conv_b0 = BUFC ? min(m0, p.M - 1) / (p.Hout * p.Wout) : 0;
        // (32-bit byte offsets span the two images a tile can touch: larger maps keep the general 64-bit addresses below)
        const bool bufc_ok = BUFC && (double)p.Hin * p.Win * p.lda * 8.0 < 2147483648.0;
        __amdgpu_buffer_rsrc_t rsrc_a = BUFC
            ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + (long)conv_b0 * p.Hin * p.Win * p.lda + p.a_coff), 0, 0x7fffffff, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A + (long)m0 * p.lda), 0, (int)min((long)(p.M - m0) * p.lda * 4, 0x7fffffffL), 0x00020000);
        int conv_pix[PA];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            conv_pix[i] = 0;
            if (BUFC) {
                const int m = m0 + lrow + i * RPP, hw = p.Hout * p.Wout;
                const int bq = a_ok[i] ? m / hw - conv_b0 : 0;
                conv_pix[i] = (((bq * p.Hin + a_iy[i]) * p.Win + a_ix[i]) * p.lda + kq * 4) * 4;   // bytes; negative inside the padding
            }
        }
        __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(reinterpret_cast<const __bf16*>(p.Wp) + (long)n0 * wrow_stride), 0,
                                                                          0x7fffffff, 0x00020000);
        unsigned voff_a[PA], voff_w[PB];
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (APL) { const int q = i * 256 + pt, row = q / (4 * NP), c = q - row * (4 * NP); voff_a[i] = (unsigned)((row * p.lda + c * 4) * 4); }
            else voff_a[i] = (unsigned)(((lrow + i * RPP) * p.lda + kq * 4) * 4);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) voff_w[i] = (unsigned)(((long)(wrow + i * WRPP) * wrow_stride + wc * 8) * 2);
        f32x4 ra[2][PA];
        bf16x8 rb[2][NP][PB];
        unsigned am[2][PA];
        auto gload = [&](int t, f32x4 (&qa)[PA], bf16x8 (&qb)[NP][PB], unsigned (&qm)[PA]) {
            const int k0 = (kt0 + t) * KS;
            if (BUFA) {
                const int wk = (KS == 32 ? (kt0 + t) * (NP * 32) : ((kt0 + t) >> 1) * (NP * 32) + ((kt0 + t) & 1) * 16) * 2;   // bytes, uniform
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int i = 0; i < PB; ++i)
                        qb[pl][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w[i], wk + pl * 64, 0));
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    qa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[i], APL ? (kt0 + t) * (64 * NP) : k0 * 4, 0));
                    qm[i] = 0xffffffffu;
                }
                return;
            }
            if (BUFC && bufc_ok) {
                const int wk = (KS == 32 ? (kt0 + t) * (NP * 32) : ((kt0 + t) >> 1) * (NP * 32) + ((kt0 + t) & 1) * 16) * 2;   // bytes, uniform
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                    for (int i = 0; i < PB; ++i)
                        qb[pl][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff_w[i], wk + pl * 64, 0));
                const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                const int dy = ky * p.dil, dx = kx * p.dil;
                const int tap_off = ((dy * p.Win + dx) * p.lda + ci0) * 4;        // uniform
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                    qa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (unsigned)(conv_pix[i] + tap_off) : 0x80000000u, 0, 0));
                    qm[i] = 0xffffffffu;
                }
                return;
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    qb[pl][i] = *reinterpret_cast<const bf16x8*>(wsrc + (long)i * WRPP * wrow_stride +
                                                                 (KS == 32 ? (long)(kt0 + t) * (NP * 32) + pl * 32
                                                                           : (long)((kt0 + t) >> 1) * (NP * 32) + pl * 32 + ((kt0 + t) & 1) * 16));
            if (APL) {
                // row m's K tile = 64 NP bytes at float offset 16 NP kt: 4 NP 16-byte chunks, chunk c = plane c / 4, k = 8 (c % 4)
#pragma unroll
                for (int i = 0; i < PA; ++i)
                    qa[i] = load4_masked(p.A + (a_ok[i] ? a_base[i] : 0) + (long)(kt0 + t) * (16 * NP), a_ok[i], qm[i]);
            } else if (MODE == GEMM_DENSE) {
#pragma unroll
                for (int i = 0; i < PA; ++i)
                    {
                    qa[i] = load4_masked(p.A + (a_ok[i] ? a_base[i] : 0) + k0 + kq * 4, a_ok[i], qm[i]);
                }
            } else {
                const int tap = k0 / p.Cin, ci0 = k0 - tap * p.Cin;
                const int ky = tap / p.kw, kx = tap - ky * p.kw;
                const int dy = ky * p.dil, dx = kx * p.dil;
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                    const long off = ok ? a_base[i] + ((long)iy * p.Win + ix) * p.lda + ci0 : (long)p.a_coff;
                    qa[i] = load4_masked(p.A + off + kq * 4, ok, qm[i]);
                }
            }
        };
        auto lds_store = [&](int t, const f32x4 (&qa)[PA], const bf16x8 (&qb)[NP][PB], const unsigned (&qm)[PA]) {
            __bf16* As = smem + (t % NBUF) * BUF;
            __bf16* Bs = As + AREG;
            if (APL) {
                // 16 bytes = 8 bf16 of plane kq >> 2 at k = 8 (kq & 3): one ds_write_b128, no arithmetic (rows beyond M were
                // loaded from row 0 and are zeroed by an integer AND with the row's 0 / ~0 mask)
#pragma unroll
                for (int i = 0; i < PA; ++i) {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 bits = __builtin_bit_cast(u32x4, qa[i]) & qm[i];
                    *reinterpret_cast<u32x4*>(As + p_lds[i]) = bits;
                }
            } else {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                bf16x4 sp[NP];
                if constexpr (H) { if (BUFA || (BUFC && bufc_ok)) split4h<false>(qa[i], qm[i], p.a_scale, sp); else split4h<true>(qa[i], qm[i], p.a_scale, sp); }
                else if (BUFA || (BUFC && bufc_ok)) split4<NP, false>(qa[i], qm[i], sp); else split4<NP>(qa[i], qm[i], sp);
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    *reinterpret_cast<bf16x4*>(As + (pl * BM + lrow + i * RPP) * SLD + a_sw) = sp[pl];
            }
            }
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int i = 0; i < PB; ++i)
                    *reinterpret_cast<bf16x8*>(Bs + (pl * BN + wrow + i * WRPP) * SLD + w_sw) = qb[pl][i];
        };
        constexpr int AHEAD = NBUF - 1;     // LDS tiles the producers run ahead of the consumers
        // prologue: LDS tiles 0 .. AHEAD-1 stored, register sets hold the next two tiles
        if (nt > 0) gload(0, ra[0], rb[0], am[0]);
        if (nt > 1) gload(1, ra[1], rb[1], am[1]);
        if (nt > 0 && !(abl & 2)) lds_store(0, ra[0], rb[0], am[0]);
        if (nt > 2) gload(2, ra[0], rb[0], am[0]);
        if (AHEAD > 1) {
            if (nt > 1 && !(abl & 2)) lds_store(1, ra[1], rb[1], am[1]);
            if (nt > 3) gload(3, ra[1], rb[1], am[1]);
        }
        if (trc && tid == 256) trc[8 + 3] = clock64();
        __syncthreads();
        // step t: store tile t+AHEAD (register set (t+AHEAD)&1), refill that set with tile t+AHEAD+2
#define BRN_PROD_STEP(T, SET)                                                              \
        {                                                                                  \
            if (trc && tid == 256 && (T) < 24) trc[16 + (T) * 4 + 0] = clock64();          \
            if (trc) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 256 && (T) < 24) trc[16 + (T) * 4 + 3] = clock64(); } \
            if ((T) + AHEAD < nt) {                                                        \
                if (!(abl & 2)) lds_store((T) + AHEAD, ra[SET], rb[SET], am[SET]);                \
                if (trc && tid == 256 && (T) < 24) trc[16 + (T) * 4 + 1] = clock64();      \
                if ((T) + AHEAD + 2 < nt && !(abl & 1)) gload((T) + AHEAD + 2, ra[SET], rb[SET], am[SET]); \
            }                                                                              \
            if (trc && tid == 256 && (T) < 24) trc[16 + (T) * 4 + 2] = clock64();          \
            if (!(abl & 16)) __syncthreads();                                            \
        }
        for (int t = 0; t < nt; t += 2) {
            BRN_PROD_STEP(t, AHEAD & 1)
            if (t + 1 < nt) BRN_PROD_STEP(t + 1, (AHEAD + 1) & 1)
        }
#undef BRN_PROD_STEP
        if (trc && tid == 256) { trc[8 + 4] = clock64(); trc[8 + 5] = wall_clock64(); }
    } else if constexpr (BRN_WS_M16 != 0 && KS == 32 && NP == 2 && !DIAG) {
    // ---- consumers, 16 x 16 x 32 MFMAs (round 4): the same cycles per flop as 32 x 32 x 16, but the chip holds a higher clock under the smaller
    // shape (MI355X_MICROARCH.md, DVFS; gemm_bf16.hip measured + 5 ... 13 % on its LDS-fed tiles).  One MFMA k = the whole 32-deep stage; a wave's
    // 64 x 64 is 4 x 4 blocks, multiplied as four 2 x 2 quadrants.  Fragment of a 16-row block: lane l reads row l & 15, 16-byte k chunk l >> 4 (XOR the row's key).
    f32x4 acc4[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[i][j] = zero4();
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kc = lane >> 4;
    const int fch = (kc ^ swz_key(r16)) * 8;           // (the key of a row depends on its bits 2, 3: the same in every 16-row block)
    const int a_row = (wm * WTM + r16) * SLD + fch, b_row = AREG + (wn * WTN + r16) * SLD + fch;
    auto read_a = [&](int t, int half, bf16x8 (&af)[NP][2]) {
        const __bf16* buf = smem + (t % NBUF) * BUF;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(buf + a_row + (pl * BM + (half * 2 + i) * 16) * SLD + pl * APAD);
    };
    auto read_b = [&](int t, int half, bf16x8 (&bf)[NP][2]) {
        const __bf16* buf = smem + (t % NBUF) * BUF;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(buf + b_row + (pl * BN + (half * 2 + j) * 16) * SLD);
    };
    auto mfma_quad = [&](const bf16x8 (&af)[NP][2], const bf16x8 (&bf)[NP][2], const int ih, const int jh) {
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)        // smallest plane products first
#pragma unroll
            for (int pa = 0; pa < NP; ++pa) {
                const int pb = sum - pa;
                if (pb < 0 || pb >= NP) continue;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {       // transposed product (W fragment first): a lane holds ONE row of a block and 4 consecutive columns
                        f32x4& d = acc4[ih * 2 + i][jh * 2 + j];
                        if constexpr (H) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bf[pb][j]), __builtin_bit_cast(f16x8, af[pa][i]), d, 0, 0, 0);
                        else d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[pb][j], af[pa][i], d, 0, 0, 0);
                    }
            }
    };
    // per stage: the first quadrant's fragments (A blocks 0, 1; W blocks 0, 1: 8 reads) are fetched right after the barrier, the other 8 reads ride
    // under the first quadrant's 12 MFMAs
    bf16x8 fa0[NP][2], fa1[NP][2], fb0[NP][2], fb1[NP][2];
    __syncthreads();   // prologue barrier: LDS tiles 0 .. AHEAD-1 are complete
    if (nt > 0) { read_a(0, 0, fa0); read_b(0, 0, fb0); }
    for (int t = 0; t < nt; ++t) {
        read_a(t, 1, fa1);
        read_b(t, 1, fb1);
        mfma_quad(fa0, fb0, 0, 0);
        mfma_quad(fa1, fb0, 1, 0);
        mfma_quad(fa0, fb1, 0, 1);
        mfma_quad(fa1, fb1, 1, 1);
        __syncthreads();
        if (t + 1 < nt) { read_a(t + 1, 0, fa0); read_b(t + 1, 0, fb0); }
    }
    // the C tile image (see the 32 x 32 form below): row = the lane's row of the block, columns 4 (lane >> 4) .. + 3
    float* ctile = reinterpret_cast<float*>(smem_raw);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(ctile + (wm * WTM + i * 16 + r16) * EP_LD + wn * WTN + j * 16 + 4 * kc) = acc4[i][j];
    } else {
    // ---- consumers ----
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    // (the swizzle key of a fragment row depends on its low five bits only: every block offset below is a multiple of 32 rows)
    const int fkey = swz_key(lane & 31);
    const int a_row = (wm * WTM + (lane & 31)) * SLD, b_row = AREG + (wn * WTN + (lane & 31)) * SLD;
    int f_chunk[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) f_chunk[ks] = (((lane >> 5) + 2 * ks) ^ fkey) * 8;
    auto read_frags = [&](int t, int ks, bf16x8 (&af)[NP][TM], bf16x8 (&bf)[NP][TN]) {
        const __bf16* buf = smem + (t % NBUF) * BUF;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(buf + a_row + (pl * BM + i * 32) * SLD + pl * APAD + f_chunk[ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(buf + b_row + (pl * BN + j * 32) * SLD + f_chunk[ks]);
        }
    };
    auto mfma_all = [&](const bf16x8 (&af)[NP][TM], const bf16x8 (&bf)[NP][TN]) {
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
                        if constexpr (H) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[pb][j]), __builtin_bit_cast(f16x8, af[pa][i]), acc[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[pb][j], af[pa][i], acc[i][j], 0, 0, 0);   // transposed product: see the C tile image below
            }
    };
    bf16x8 fa0[NP][TM], fb0[NP][TN], fa1[NP][TM], fb1[NP][TN];
    __syncthreads();   // prologue barrier: LDS tiles 0 .. AHEAD-1 are complete
    if (trc && tid == 0) trc[3] = clock64();
    constexpr bool XPREFETCH = NBUF >= 3;   // tile t+1 is complete during step t only with a 3-deep ring
    if (nt > 0 && !(abl & 4)) read_frags(0, 0, fa0, fb0);
    for (int t = 0; t < nt; ++t) {
        if (trc && tid == 0 && t < 24) trc[128 + t * 4 + 0] = clock64();
        if (!(abl & 4)) {
            if (KSTEPS == 2) {
                read_frags(t, 1, fa1, fb1);
                mfma_all(fa0, fb0);
                if (XPREFETCH && t + 1 < nt) read_frags(t + 1, 0, fa0, fb0);
                mfma_all(fa1, fb1);
            } else {
                mfma_all(fa0, fb0);
            }
        }
        if (trc && tid == 0 && t < 24) trc[128 + t * 4 + 1] = clock64();
        if (!(abl & 16)) __syncthreads();
        if ((!XPREFETCH || KSTEPS == 1) && t + 1 < nt && !(abl & 4)) read_frags(t + 1, 0, fa0, fb0);
    }
    if (trc && tid == 0) trc[4] = clock64();
    // the staging LDS is dead (every fragment read retired at the last barrier): the consumers lay their accumulators down as a
    // row-major image of the C tile
    float* ctile = reinterpret_cast<float*>(smem_raw);
    {
        // the product was formed transposed (W fragment = the MFMA's first operand): a lane holds ONE row (lane & 31) of a 32 x 32 block and
        // columns 8g + 4h + {0..3} in registers 4g .. 4g+3 (h = lane >> 5) — 16 ds_write_b128 per lane instead of 64 ds_write_b32
        // (rows are 528 bytes apart: 8 consecutive lanes hit 8 x 4 different banks)
        const int row = lane & 31, h4 = (lane >> 5) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(ctile + (wm * WTM + i * 32 + row) * EP_LD + wn * WTN + j * 32 + 8 * g + h4) = v;
                }
    }
    }   // consumers
    // ---- epilogue, all eight waves: the producers have nothing left to do, and a 4-wave epilogue was 5-11 us of store-issue
    // latency per tile.  Each pass moves 16 rows x 512 B: one wave = two full rows, float4 per lane ----
    __syncthreads();
    gemm_epilogue_tile<BM, BN, EP_LD>(p, reinterpret_cast<const float*>(smem_raw), m0, n0, tid, slice);
    if (trc && tid == 0) { trc[5] = clock64(); trc[6] = wall_clock64(); }
}

template <int NP, int KS>
static hipError_t launch_split_ws_ks(const GemmParams& p, dim3 grid, hipStream_t s) {
    const dim3 block(512);
    if (p.h2) {                           // mode f32_half2: two fp16 planes, 32-deep stages, plain or P-layout A
        if constexpr (NP == 2 && KS == 32) {
            if (p.a_planes) {
                if (p.mode != GEMM_DENSE || p.a_planes != 2) return hipErrorInvalidValue;
                hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, 2, 32, false, true, true>), grid, block, 0, s, p);
            } else if (p.mode == GEMM_DENSE) hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, 2, 32, false, false, true>), grid, block, 0, s, p);
            else if (p.mode == GEMM_CONV_NHWC) hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_CONV_NHWC, 2, 32, false, false, true>), grid, block, 0, s, p);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    if (p.a_planes) {
        if constexpr (NP == 2 && KS == 32) {   // (the 3-plane form works too, but was 2 % slower per forward: rows 1.5x as long)
            if (p.mode != GEMM_DENSE || p.a_planes != NP) return hipErrorInvalidValue;
#ifdef BRN_DIAG_BUILD
            if (p.abl || p.trace) { hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, 2, 32, true, true>), grid, block, 0, s, p); return hipGetLastError(); }
#endif
            hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, 2, 32, false, true>), grid, block, 0, s, p);
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
#ifdef BRN_DIAG_BUILD
    if (p.mode == GEMM_DENSE && KS == 32 && (p.abl || p.trace)) { hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, NP, 32, true, false>), grid, block, 0, s, p); return hipGetLastError(); }
#endif
    if (p.mode == GEMM_DENSE) hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_DENSE, NP, KS, false, false>), grid, block, 0, s, p);
    else if (p.mode == GEMM_CONV_NHWC) hipLaunchKernelGGL((gemm_split_ws_kernel<GEMM_CONV_NHWC, NP, KS, false, false>), grid, block, 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
template <int NP>
static hipError_t launch_split_ws(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + 127) / 128) * ((p.N + 127) / 128) * p.splitk;
    const dim3 grid(tiles);
    // 3 planes: 32-deep stages need 120 KB of LDS (one workgroup per CU); 16-deep stages (2 x 36 KB) let two share a CU like the
    // 2-plane kernel's do, at twice the barriers per K: worth it as soon as there is more than one workgroup per CU to place
    if (NP == 3 && tiles > 256 && !(p.abl || p.trace)) return launch_split_ws_ks<NP, 16>(p, grid, s);
    return launch_split_ws_ks<NP, 32>(p, grid, s);
}

template <int BM, int BN, int WM, int WN, int NP>
static hipError_t launch_split_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
    dim3 grid(tiles), block(WM * WN * 64);
    if (p.h2) {
        if constexpr (NP == 2) {
            if (p.mode == GEMM_DENSE) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, GEMM_DENSE, 2, true>), grid, block, 0, s, p);
            else if (p.mode == GEMM_CONV_NHWC) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, GEMM_CONV_NHWC, 2, true>), grid, block, 0, s, p);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    if (p.mode == GEMM_DENSE) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, GEMM_DENSE, NP>), grid, block, 0, s, p);
    else if (p.mode == GEMM_CONV_NHWC) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, GEMM_CONV_NHWC, NP>), grid, block, 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

#ifdef BRN_DIAG_BUILD   // probe kernels: only in libbirefnet_hip_diag.so (include/birefnet_hip_diag.h)
// diagnostic: the consumer inner loop of the split kernels in isolation — fragments from LDS (ds_read_b128, conflict-free
// layout of the real kernel), NPAIR MFMAs per (i,j) sub-tile, no global memory, no barriers.  variant 0: reads of a k-step
// issued right before its MFMAs; variant 1: next k-step's fragments prefetched into a second register set.
template <int NP, int VARIANT>
__global__ void __launch_bounds__(256) lds_mfma_probe_kernel(int iters, float* sink) {
    constexpr int TM = 2, TN = 2, LD = 40;
    __shared__ __attribute__((aligned(16))) __bf16 smem[NP * 256 * LD];
    for (int i = threadIdx.x; i < NP * 256 * LD; i += 256) smem[i] = (__bf16)((float)((i * 7) & 15) * 0.0625f - 0.4f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const __bf16* a_frag = smem + (wm * 64 + (lane & 31)) * LD + (lane >> 5) * 8;
    const __bf16* b_frag = smem + NP * 128 * LD + (wn * 64 + (lane & 31)) * LD + (lane >> 5) * 8;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto rd = [&](int ks, bf16x8 (&af)[NP][TM], bf16x8 (&bf)[NP][TN]) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[pl][i] = *reinterpret_cast<const bf16x8*>(a_frag + (pl * 128 + i * 32) * LD + ks * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[pl][j] = *reinterpret_cast<const bf16x8*>(b_frag + (pl * 128 + j * 32) * LD + ks * 16);
        }
    };
    auto mm = [&](const bf16x8 (&af)[NP][TM], const bf16x8 (&bf)[NP][TN]) {
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], bf[pb][j], acc[i][j], 0, 0, 0);
            }
    };
    bf16x8 a0[NP][TM], b0[NP][TN], a1[NP][TM], b1[NP][TN];
    if (VARIANT == 0) {
        for (int it = 0; it < iters; ++it) {
            rd(0, a0, b0); mm(a0, b0);
            rd(1, a1, b1); mm(a1, b1);
            asm volatile("" ::: "memory");
        }
    } else {
        rd(0, a0, b0);
        for (int it = 0; it < iters; ++it) {
            rd(1, a1, b1); mm(a0, b0);
            rd(0, a0, b0); mm(a1, b1);
            asm volatile("" ::: "memory");
        }
    }
    float t = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) sink[0] = t;
}
hipError_t launch_lds_mfma_probe(int blocks, int iters, int np, int variant, float* sink, hipStream_t s) {
#define BRN_P(NP_, V_) hipLaunchKernelGGL((lds_mfma_probe_kernel<NP_, V_>), dim3(blocks), dim3(256), 0, s, iters, sink)
    if (np == 1) { if (variant) BRN_P(1, 1); else BRN_P(1, 0); }
    else if (np == 2) { if (variant) BRN_P(2, 1); else BRN_P(2, 0); }
    else { if (variant) BRN_P(3, 1); else BRN_P(3, 0); }
#undef BRN_P
    return hipGetLastError();
}

// diagnostic: how MFMA and plain VALU work share a SIMD.  8 waves (two per SIMD) per workgroup.
//   mode 0: waves 0-3 MFMA only, waves 4-7 exit        mode 1: waves 4-7 VALU only, waves 0-3 exit
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (specialised) mode 3: every wave 1/2 of both, VALU interleaved between its MFMAs
//   mode 4: every wave 1/2 of both, VALU in one block after the MFMAs
template <int MODE>
__global__ void __launch_bounds__(512) mfma_valu_probe_kernel(int iters, float* sink) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (threadIdx.x & 31) + e); b[e] = (__bf16)(0.5f - 0.03f * e); }
    float v[6];
    for (int e = 0; e < 6; ++e) v[e] = 1.0f + 0.001f * threadIdx.x + e;
    unsigned u[3] = {0, 0, 0};
#define BRN_MFMA(I) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[I]) : "v"(a), "v"(b))
    // the split's op mix per pair of elements: cvt_pk, shift, and, 2 sub, cvt_pk  (6 plain VALU)
#define BRN_VALU6(X, Y, U)                                                             \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(U) : "v"(X), "v"(Y));          \
    asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(X) : "v"(U), "v"(X)); \
    asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %0, %2, %0" : "=&v"(Y) : "v"(U), "v"(Y)); \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(U) : "v"(X), "v"(Y));
    if (MODE == 0 || MODE == 2) {
        if (wave < 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 24; ++m) BRN_MFMA(m & 3);
            }
        } else if (MODE == 2) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 12; ++g) { BRN_VALU6(v[(g % 3) * 2], v[(g % 3) * 2 + 1], u[g % 3]) }
            }
        }
    } else if (MODE == 1) {
        if (wave >= 4) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 12; ++g) { BRN_VALU6(v[(g % 3) * 2], v[(g % 3) * 2 + 1], u[g % 3]) }
            }
        }
    } else if (MODE >= 5) {
        // 5: waves 0-3 MFMA + fragment reads   6: waves 4-7 VALU + LDS stores   7: both   (the warp-specialised GEMM's K-tile shape)
        __shared__ __attribute__((aligned(16))) char lds[40960];
        const int lane = threadIdx.x & 63;
        if (wave < 4 && MODE != 6) {
            const unsigned ra = (unsigned)(size_t)lds + ((wave >> 1) * 64 + (lane & 31)) * 80 + (lane >> 5) * 16;
            f32x4 f[8];
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[q]) : "v"(ra), "n"(0) );
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int m = 0; m < 12; ++m)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(f[m & 7]), "v"(f[(m + 3) & 7]));
                }
            }
            v[0] += f[0][0];
        } else if (wave >= 4 && MODE != 5) {
            const int pt = threadIdx.x - 256;
            const unsigned wa = (unsigned)(size_t)lds + (pt >> 3) * 80 + (pt & 7) * 8;
            const unsigned wb = (unsigned)(size_t)lds + 20480 + (pt >> 2) * 80 + (pt & 3) * 16;
            f32x4 w4 = {v[0], v[1], v[2], v[3]};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g = 0; g < 12; ++g) { BRN_VALU6(v[(g % 3) * 2], v[(g % 3) * 2 + 1], u[g % 3]) }
                unsigned long long d0 = ((unsigned long long)u[0] << 32) | u[1], d1 = ((unsigned long long)u[2] << 32) | u[0];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(wa), "v"(d0), "v"(d1), "n"(0), "n"(20) : "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wb), "v"(w4), "n"(0) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    } else if (MODE == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                BRN_MFMA(m & 3);
                if (m & 1) { BRN_VALU6(v[(m % 3) * 2], v[(m % 3) * 2 + 1], u[m % 3]) }
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 12; ++m) BRN_MFMA(m & 3);
#pragma unroll
            for (int g = 0; g < 6; ++g) { BRN_VALU6(v[(g % 3) * 2], v[(g % 3) * 2 + 1], u[g % 3]) }
        }
    }
#undef BRN_MFMA
#undef BRN_VALU6
    float t = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + (float)(u[0] ^ u[1] ^ u[2]);
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 123.456f) sink[0] = t;
}
hipError_t launch_mfma_valu_probe(int blocks, int iters, int mode, float* sink, hipStream_t s) {
#define BRN_P(M_) hipLaunchKernelGGL((mfma_valu_probe_kernel<M_>), dim3(blocks), dim3(512), 0, s, iters, sink)
    switch (mode) { case 0: BRN_P(0); break; case 1: BRN_P(1); break; case 2: BRN_P(2); break; case 3: BRN_P(3); break; case 5: BRN_P(5); break; case 6: BRN_P(6); break; case 7: BRN_P(7); break; default: BRN_P(4); break; }
#undef BRN_P
    return hipGetLastError();
}

// diagnostic: back-to-back v_mfma_f32_32x32x2_f32 on register operands (4 independent accumulators per wave); lane 0 of
// each wave reports shader-clock / 100 MHz-realtime-clock ticks so the host can derive the sustained clock
__global__ void mfma_peak_kernel(int iters, float* sink, unsigned long long* clk) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    float x = (float)(threadIdx.x & 7) * 0.125f - 0.4f, y = (float)(threadIdx.x & 3) * 0.25f - 0.3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r];
    if (acc == 123.456f) sink[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ void mfma_peak_bf16_kernel(int iters, float* sink, unsigned long long* clk, int nacc) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)((float)((threadIdx.x + j) & 7) * 0.125f - 0.4f); y[j] = (__bf16)((float)((threadIdx.x * 3 + j) & 3) * 0.25f - 0.3f); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (nacc == 4) {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) acc += a0[r] + a1[r] + a2[r] + a3[r];
    if (acc == 123.456f) sink[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
hipError_t launch_mfma_peak_bf16(int blocks, int iters, float* sink, unsigned long long* clk, int nacc, hipStream_t s) {
    hipLaunchKernelGGL(mfma_peak_bf16_kernel, dim3(blocks), dim3(256), 0, s, iters, sink, clk, nacc);
    return hipGetLastError();
}
hipError_t launch_mfma_peak(int blocks, int iters, float* sink, unsigned long long* clk, hipStream_t s) {
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, iters, sink, clk);
    return hipGetLastError();
}

#endif  // BRN_DIAG_BUILD

// split-K second pass: fixed-order sum of the slices (deterministic) + the epilogue of gemm_f32_kernel
__global__ void splitk_reduce_kernel(const GemmParams p) {
    const long total = (long)p.M * p.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += p.part[(long)s * total + idx];
        if (p.bias) v += p.bias[n];
        if (p.bbias) v += p.bbias[(long)(m / p.bbias_rows) * p.N + n];
        if (p.scale) v = v * p.scale[n] + p.shift[n];
        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == ACT_GELU_ERF) v = gelu_erf(v);
        if (p.R) v += p.R[(long)m * p.ldr + p.r_coff + n];
        p.C[(long)m * p.ldc + p.c_coff + n] = v;
    }
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.splitk;
    dim3 grid(tiles), block(WM * WN * 64);
    switch (p.mode) {
        case GEMM_DENSE: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, GEMM_DENSE>), grid, block, 0, s, p); break;
        case GEMM_CONV_NHWC: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, GEMM_CONV_NHWC>), grid, block, 0, s, p); break;
        case GEMM_GATHER_NCHW: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, GEMM_GATHER_NCHW>), grid, block, 0, s, p); break;
        case GEMM_DEFORM_NHWC: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, GEMM_DEFORM_NHWC>), grid, block, 0, s, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Tile + split-K choice from a small cost model calibrated on MI355X sweeps (tools/gemm_bench.py): a CU's matrix pipe is
// shared by its resident workgroups, so a launch lasts ~ ceil(tiles / 256 CUs) x (tile area / efficiency of the config).
// When even the 64x64 tiling leaves most CUs idle (tall-K convs on small maps, half-scale Swin GEMMs at batch 1) the K
// loop is split so that ~512 workgroups exist.  W is padded to 128 rows so every config may over-read it.
GemmPlan plan_gemm(int M, int N, int K, int planes) {
    struct Cand { int cfg, bm, bn; double eff; };
    static const Cand cands[] = {{0, 128, 128, 1.00}, {1, 128, 64, 0.98}, {2, 64, 64, 0.93}, {5, 128, 128, 1.04}, {4, 256, 128, 1.06}};
    GemmPlan pl{2, 1, 0};
    double best = 1e300;
    for (const Cand& c : cands) {
        const long tiles = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        if ((c.cfg == 5 || c.cfg == 4) && tiles < 1024) continue;      // the 8-wave tiles only pay on large grids
        const double cost = (double)((tiles + 255) / 256) * c.bm * c.bn / c.eff;
        if (cost < best) { best = cost; pl.cfg = c.cfg; }
    }
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int nk = K / BK;
    if (planes > 0 && N >= 128) {
        // split-bf16 modes: the warp-specialised 128x128 kernel beats the 4-wave tiles by 1.3-1.5x whenever its tiles are
        // mostly full (measured, tools/gemm_sk_sweep.py), two workgroups per CU = 512 slots; under ~200 tiles the K loop is
        // cut so that ~480 workgroups exist, but never below 24 K tiles per slice (the reduce pass costs more than it buys)
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        const double waste = (double)t128 * 128.0 * 128.0 / ((double)M * N);
        if (waste <= 1.35) {          // N = 192 (1.33) still wins on the warp-specialised kernel: 214 vs 183 TF/s-eq at 81920 x 192 x 768
            pl.cfg = 0; pl.splitk = 1; pl.ws_floats = 0;
            static const int sk_t128 = getenv("BRN_SK_T128") ? atoi(getenv("BRN_SK_T128")) : 200;
            if (t128 < sk_t128) {
                int s = (int)(480 / t128);
                if (s > nk / 24) s = nk / 24;
                if (s > 8) s = 8;
                if (s > 1) { pl.splitk = s; pl.ws_floats = (size_t)s * M * N; }
            }
            return pl;
        }
    }
    if (t64 < 384 && nk >= 8) {
        pl.cfg = 2;
        int s = (int)((768 + t64 - 1) / t64);
        if (s > nk / 4) s = nk / 4;
        if (s > 64) s = 64;
        if (s > 1) { pl.splitk = s; pl.ws_floats = (size_t)s * M * N; }
    }
    return pl;
}

hipError_t launch_gemm(const GemmParams& p_in, const GemmPlan& pl, float* ws, hipStream_t s) {
    if (p_in.M <= 0 || p_in.N <= 0 || p_in.K <= 0 || (p_in.K % BK) != 0) return hipErrorInvalidValue;
    if ((p_in.mode == GEMM_CONV_NHWC || p_in.mode == GEMM_DEFORM_NHWC) && (p_in.Cin % BK) != 0) return hipErrorInvalidValue;
    GemmParams p = p_in;
    p.splitk = pl.splitk < 1 ? 1 : pl.splitk;
    p.part = ws;
    if (p.splitk > 1 && !ws) return hipErrorInvalidValue;
    if (p.a_bf16 && p.mode != GEMM_DEFORM_NHWC) return hipErrorInvalidValue;       // only the deformable loader reads bf16 maps here
    if (p.c_bf16 && (p.splitk > 1 || p.R || (p.planes > 0 && p.Wp && p.mode != GEMM_DEFORM_NHWC && p.mode != GEMM_GATHER_NCHW))) return hipErrorInvalidValue;
    if (p.h2 && !(p.planes == 2 && p.Wp && (p.mode == GEMM_DENSE || p.mode == GEMM_CONV_NHWC))) p.h2 = 0;     // (the fp32-MFMA kernel reads W itself: nothing is scaled)
    if (p.h2 && !(p.a_scale > 0.f && p.out_scale > 0.f)) return hipErrorInvalidValue;
    if (p.a_planes || p.c_planes) {         // P2 layouts: 2-plane split mode, dense, on the warp-specialised kernel only
        const bool ws_cfg = pl.cfg == 6 || pl.cfg == 0 || pl.cfg == 3 || pl.cfg == 4 || pl.cfg == 5;
        if (!(p.planes == 2 || p.planes == 3) || !p.Wp || p.mode != GEMM_DENSE || !ws_cfg) return hipErrorInvalidValue;
        if (p.a_planes && (p.a_planes != p.planes || p.lda % 16 || p.a_coff)) return hipErrorInvalidValue;
        if (p.c_planes && (p.c_planes != p.planes || p.splitk > 1 || p.R || p.N % 32 || p.ldc % 16 || p.c_coff % 32)) return hipErrorInvalidValue;
    }
    hipError_t e;
    if (p.planes > 0 && p.Wp && (p.mode == GEMM_DENSE || p.mode == GEMM_CONV_NHWC)) {
        // split-bf16 path: tile choice by the same plan (64x64 / 128x64 / 128x128 families); the 128x128 tile runs
        // warp-specialised
        if (pl.cfg == 6 || pl.cfg == 0 || pl.cfg == 3 || pl.cfg == 4 || pl.cfg == 5) {
            if (p.planes == 3) e = launch_split_ws<3>(p, s);
            else if (p.planes == 2) e = launch_split_ws<2>(p, s);
#ifdef BRN_DIAG_BUILD               // one bf16 plane (mode bf16_operands, superseded by the bf16-storage mode): diag build only
            else e = launch_split_ws<1>(p, s);
#else
            else e = hipErrorInvalidValue;
#endif
        }
        else if (p.planes == 3) {
            if (pl.cfg == 2) e = launch_split_cfg<64, 64, 2, 2, 3>(p, s);
            else if (pl.cfg == 1) e = launch_split_cfg<128, 64, 2, 2, 3>(p, s);
            else e = launch_split_cfg<128, 128, 2, 2, 3>(p, s);
        } else if (p.planes == 2) {
            if (pl.cfg == 2) e = launch_split_cfg<64, 64, 2, 2, 2>(p, s);
            else if (pl.cfg == 1) e = launch_split_cfg<128, 64, 2, 2, 2>(p, s);
            else e = launch_split_cfg<128, 128, 2, 2, 2>(p, s);
        } else {
#ifdef BRN_DIAG_BUILD
            if (pl.cfg == 2) e = launch_split_cfg<64, 64, 2, 2, 1>(p, s);
            else if (pl.cfg == 1) e = launch_split_cfg<128, 64, 2, 2, 1>(p, s);
            else e = launch_split_cfg<128, 128, 2, 2, 1>(p, s);
#else
            e = hipErrorInvalidValue;
#endif
        }
    } else
    if (pl.cfg == 0) e = launch_cfg<128, 128, 2, 2>(p, s);
    else if (pl.cfg == 1) e = launch_cfg<128, 64, 2, 2>(p, s);
    else if (pl.cfg == 3) e = launch_cfg<128, 128, 2, 4>(p, s);
    else if (pl.cfg == 4) e = launch_cfg<256, 128, 4, 2>(p, s);
    else if (pl.cfg == 5) e = launch_cfg<128, 128, 4, 2>(p, s);
    else e = launch_cfg<64, 64, 2, 2>(p, s);
    if (e != hipSuccess || p.splitk == 1) return e;
    long total = (long)p.M * p.N;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

}  // namespace brn
