// deform_bf16.hip — the modulated deformable convolution of compute mode BRN_BF16 on the bf16 matrix cores.
// Replaces DeformConvASPP::forward_metal / DeformableConv2d::forward_metal (aspp.rs:58-165, deform_conv.rs:101-215: deformable
// im2col -> [Cin k^2, B Ho Wo] column matrix -> matmul) for bf16 channels-last maps: the bilinear gather x modulator is the A-tile
// loader of an MFMA GEMM, the column matrix is never formed.  Sampling semantics (torchvision deform_conv2d): offset
// channel 2t = dy, 2t + 1 = dx of tap t = ky kw + kx; sample at (oy stride - pad + ky + dy, ox stride - pad + kx + dx), bilinear,
// zero outside (-1, H) x (-1, W), corners outside the image contribute zero; x modulator (2 sigmoid, aspp.rs:173-174); one offset group.
//
// Structure.  One workgroup (4 waves) owns 64 output pixels x 256 output channels; a K step is one tap x 64 input channels.
//   * A tile (64 pixels x 64 channels, bf16, 8 KB, two buffers): thread t gathers chunk c = t & 7 (8 channels = 16 bytes) of rows
//     t >> 3 and (t >> 3) + 32: four 16-byte corner loads each (8 lanes = one 128-byte pixel row: coalesced), fp32 bilinear
//     combination x modulator, one v_cvt_pk per pair, ds_write_b128 into the XOR-swizzled image of gemm_bf16.hip (rows 2p, 2p+1
//     share a 256-byte bank row whose sixteen 16-byte slots are permuted by p & 15: conflict-free fragment reads).  Corner addresses
//     are clamped into the image and the corner's weight is zeroed instead of predicating the load (no exec branches around loads:
//     the loads of K step t+1 stay in flight under the MFMAs of step t).
//   * W never touches LDS: the four waves own disjoint 64-column slices (no reuse inside the workgroup), and the weights are
//     stored at load time in MFMA fragment order — [n / 16][K step][k32 half][lane][8 bf16] — so a wave's fragment load is one
//     contiguous 1-KiB read (brn_weights.cpp, attach_deform_frags).
//   * v_mfma_f32_16x16x32_bf16, transposed product (W fragment first): a lane holds 4 consecutive output channels of one pixel.
//   * epilogue: bias / folded BN / ReLU on the fp32 accumulators, bf16 rows through LDS (XOR-ed 16-byte chunks), stored as whole
//     512-byte pixel rows (16 bytes per lane) into the consumer's column window.
// Bound: the gather (4 x 128 bytes per pixel and K step from L1 / L2) and its VALU work, not the matrix pipe: per K step a wave
// issues 32 MFMAs (512 matrix-pipe cycles) beside ~230 vector instructions.
#include "../brn_kernels.h"
#include "split_planes.h"
#include <cstdlib>
#include <type_traits>

// Compiled twice like gemm_bf16.hip: -DBRN_S16_F16=1 builds the fp16-storage flavour (compute mode BRN_F16) in namespace brn::hf.
#ifndef BRN_S16_F16
#define BRN_S16_F16 0
#endif
namespace brn {
#if BRN_S16_F16
namespace hf {
typedef _Float16 s16_t;
#define BRN_MFMA_16X16X32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)
#else
typedef __bf16 s16_t;
#define BRN_MFMA_16X16X32(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
#endif
typedef s16_t s16x8 __attribute__((ext_vector_type(8)));
#if BRN_S16_F16      // (an unqualified call would also find the brn:: function of the same name through its brn::GemmParams argument)
#define BRN_S16_SELF(FN) hf::FN
#else
#define BRN_S16_SELF(FN) FN
#endif
__device__ __forceinline__ float d16_lo_f32(unsigned r) {
#if BRN_S16_F16
    typedef _Float16 h2_d __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2_d, r)[0];
#else
    return __builtin_bit_cast(float, r << 16);
#endif
}
__device__ __forceinline__ float d16_hi_f32(unsigned r) {
#if BRN_S16_F16
    typedef _Float16 h2_d __attribute__((ext_vector_type(2)));
    return (float)__builtin_bit_cast(h2_d, r)[1];
#else
    return __builtin_bit_cast(float, r & 0xffff0000u);
#endif
}

typedef float f32x4_d __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_d __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_d __attribute__((ext_vector_type(2)));

constexpr int DBM = 64, DBN = 256, DBK = 64;

__device__ __forceinline__ unsigned dpack2(float lo, float hi) {
    typedef s16_t s16x2_d __attribute__((ext_vector_type(2)));
    const s16x2_d t = {(s16_t)lo, (s16_t)hi};
    return __builtin_bit_cast(unsigned, t);
}
// byte offset of 16-byte chunk c (0..7) of tile row r (128-byte rows) in the swizzled A image
__device__ __forceinline__ int a_slot(int r, int c) { return (r >> 1) * 256 + (((((r & 1) << 3) | c) ^ ((r >> 1) & 15)) << 4); }

__global__ void __launch_bounds__(256, 2) gemm_deform_bf16_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[DBM * DBN * 2];   // K loop: two 8-KB A tiles; epilogue: [64][256] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order (bijective): the workgroups that share an L2 walk a contiguous run of pixel tiles (neighbouring image rows)
    const int tilesM = (p.M + DBM - 1) / DBM, tilesN = (p.N + DBN - 1) / DBN;
    int swz;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_n = swz / tilesM, tile_m = swz - tile_n * tilesM;     // (tilesN is 1 for the ASPP modules)
    const int m0 = tile_m * DBM, n0 = tile_n * DBN;
    (void)tilesN;

    // ---- gather state of this thread's two rows ----
    const int gc = tid & 7, gr = tid >> 3;
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    const char* a_img[2];        // first byte of (image b, channel a_coff + 8 gc)
    const float* om_row[2];
    int iy0[2], ix0[2];
    bool rok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + gr + 32 * i;
        rok[i] = m < p.M;
        const int mm = rok[i] ? m : p.M - 1;
        const int hw = p.Hout * p.Wout;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        iy0[i] = oy * p.stride - p.pad;
        ix0[i] = ox * p.stride - p.pad;
        a_img[i] = reinterpret_cast<const char*>(Ab + (long)b * p.Hin * p.Win * p.lda + p.a_coff + gc * 8);
        om_row[i] = p.om + (long)mm * p.om_ld;
    }
    const int cpt = p.Cin / DBK;                     // K steps per tap
    const int nk = p.K / DBK;
    const int pix_bytes = p.lda * 2;

    u32x4_d gv[2][4];
    float gw[2][4];
    auto gather_issue = [&](int kt) {
        const int tap = kt / cpt, ci0 = (kt - tap * cpt) * DBK;
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float offy = om_row[i][2 * tap], offx = om_row[i][2 * tap + 1];
            float mk = om_row[i][p.om_mask_off + tap];
            if (p.om_sigmoid) mk = (1.0f / (1.0f + __expf(-mk))) * 2.0f;          // aspp.rs:173-174
            const float y = (float)(iy0[i] + ky * p.dil) + offy;
            const float x = (float)(ix0[i] + kx * p.dil) + offx;
            const bool inside = rok[i] && y > -1.f && y < (float)p.Hin && x > -1.f && x < (float)p.Win;
            const float yf = floorf(y), xf = floorf(x);
            const int yl = (int)yf, xl = (int)xf, yh = yl + 1, xh = xl + 1;
            const float ly = y - yf, lx = x - xf, hy = 1.f - ly, hx = 1.f - lx;
            const float s = inside ? mk : 0.f;
            const bool yl_ok = yl >= 0, yh_ok = yh <= p.Hin - 1, xl_ok = xl >= 0, xh_ok = xh <= p.Win - 1;
            gw[i][0] = (yl_ok && xl_ok) ? s * (hy * hx) : 0.f;
            gw[i][1] = (yl_ok && xh_ok) ? s * (hy * lx) : 0.f;
            gw[i][2] = (yh_ok && xl_ok) ? s * (ly * hx) : 0.f;
            gw[i][3] = (yh_ok && xh_ok) ? s * (ly * lx) : 0.f;
            // clamped corners: always a valid address (the value is multiplied by a zero weight when it is not a real corner);
            // integer clamps on the float->int results also tame NaN / huge offsets
            const int ylc = min(max(yl, 0), p.Hin - 1), yhc = min(max(yh, 0), p.Hin - 1);
            const int xlc = min(max(xl, 0), p.Win - 1), xhc = min(max(xh, 0), p.Win - 1);
            const char* base = a_img[i] + ci0 * 2;
            gv[i][0] = *reinterpret_cast<const u32x4_d*>(base + (long)(ylc * p.Win + xlc) * pix_bytes);
            gv[i][1] = *reinterpret_cast<const u32x4_d*>(base + (long)(ylc * p.Win + xhc) * pix_bytes);
            gv[i][2] = *reinterpret_cast<const u32x4_d*>(base + (long)(yhc * p.Win + xlc) * pix_bytes);
            gv[i][3] = *reinterpret_cast<const u32x4_d*>(base + (long)(yhc * p.Win + xhc) * pix_bytes);
        }
    };
    auto gather_finish = [&](char* abuf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4_d o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = 0.f, hi = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    lo = fmaf(gw[i][c], d16_lo_f32(gv[i][c][e]), lo);
                    hi = fmaf(gw[i][c], d16_hi_f32(gv[i][c][e]), hi);
                }
                o[e] = dpack2(lo, hi);
            }
            *reinterpret_cast<u32x4_d*>(abuf + a_slot(gr + 32 * i, gc)) = o;
        }
    };

    // ---- fragments ----
    // W: fragment (nb, kt, s) = 1 KiB at wf + (((nb nk + kt) 2 + s) 64 + lane) 16 bytes; this wave owns n16 blocks n0/16 + 4 wave + j
    const char* wf = reinterpret_cast<const char*>(p.Wp) + ((long)((n0 >> 4) + 4 * wave) * nk * 2 * 64 + lane) * 16;
    const long wf_nb = (long)nk * 2 * 1024;          // bytes between consecutive n16 blocks
    // A: lane reads tile row 16 i + (lane & 15), chunk 4 s + (lane >> 4)
    int a_foff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s) a_foff[i][s] = a_slot(16 * i + (lane & 15), 4 * s + (lane >> 4));

    f32x4_d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_d{0.f, 0.f, 0.f, 0.f};

    gather_issue(0);
    gather_finish(smem);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const char* abuf = smem + (kt & 1) * (DBM * DBK * 2);
        s16x8 wfr[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) wfr[j][s] = *reinterpret_cast<const s16x8*>(wf + j * wf_nb + (long)(kt * 2 + s) * 1024);
        if (kt + 1 < nk) gather_issue(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const s16x8*>(abuf + a_foff[i][s]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = BRN_MFMA_16X16X32(wfr[j][s], af[i], acc[i][j]);
        }
        if (kt + 1 < nk) gather_finish(smem + ((kt + 1) & 1) * (DBM * DBK * 2));
        __syncthreads();
    }

    // ---- epilogue: lane holds pixel 16 i + (lane & 15), channels 64 wave + 16 j + 4 (lane >> 4) + {0..3} ----
    {
        const int q4 = lane >> 4, pr = lane & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = 64 * wave + 16 * j + 4 * q4;          // channel within the tile
            const int n = n0 + nl;
            f32x4_d bias = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {                                         // N % 4 == 0 (launcher)
                if (p.bias) bias = *reinterpret_cast<const f32x4_d*>(p.bias + n);
                if (p.scale) { sc = *reinterpret_cast<const f32x4_d*>(p.scale + n); sh = *reinterpret_cast<const f32x4_d*>(p.shift + n); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4_d v = (acc[i][j] + bias) * sc + sh;
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const int row = 16 * i + pr;
                const int chunk = (nl >> 3) ^ (row & 31);          // 32 chunks of 16 bytes per 512-byte row
                const u32x2_d o = {dpack2(v[0], v[1]), dpack2(v[2], v[3])};
                *reinterpret_cast<u32x2_d*>(smem + row * (DBN * 2) + chunk * 16 + (nl & 4) * 2) = o;
            }
        }
    }
    __syncthreads();
    {
        s16_t* Cb = reinterpret_cast<s16_t*>(p.C);
#pragma unroll
        for (int ps = 0; ps < DBM / 8; ++ps) {
            const int row = ps * 8 + (tid >> 5), c = tid & 31;
            const int m = m0 + row, n = n0 + c * 8;
            const u32x4_d v = *reinterpret_cast<const u32x4_d*>(smem + row * (DBN * 2) + ((c ^ (row & 31)) << 4));
            if (m < p.M && n < p.N) *reinterpret_cast<u32x4_d*>(Cb + (long)m * p.ldc + p.c_coff + n) = v;   // N % 8 == 0 (launcher)
        }
    }
}

// =====================================================================================================================
// gemm_deform_bf16_v2_kernel (round 4) — the same tile and MFMA layout; what changed is WHO computes the sampling parameters and HOW FAR
// AHEAD the gather runs.
//   * v1: every thread loaded its rows' offsets / modulator, did floor / clamp / bilinear weights / 2 sigmoid itself — the 8 lanes that
//     share a pixel all the same work — and only then could issue its 8 corner loads: two dependent memory round trips (offsets, then
//     corners) per K step with one K step of cover (the PMC view: instructions issuing on 0.37 of wave cycles, MFMA busy 0.15).
//   * v2: a wave owns 16 of the tile's 64 pixels; every 4 taps ONE lane per (pixel, tap) — 16 pixels x 4 taps = 64 lanes — computes
//     the four clamped corner offsets (bytes from the tile's first image: buffer-load offsets) and the four weights x modulator and parks
//     them in a wave-private LDS table (2 KB, no workgroup barrier involved).  Its offset / modulator loads go out one 4-tap block
//     ahead.  A thread's gather for a K step is then 2 x (2 ds_read_b128 + 4 buffer_load_dwordx4): no address arithmetic beyond one add
//     per corner, no transcendental, and the corner loads of K step t + 2 are issued during step t (two register sets: a whole K step
//     more latency cover); the W fragments of the next half step load under the MFMAs of the current one.
// =====================================================================================================================
template <int UNUSED = 0>
__global__ void __launch_bounds__(256, 2) gemm_deform_bf16_v2_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) char smem[DBM * DBN * 2];   // K loop: two 8-KB A tiles + 4 x 2-KB parameter tables; epilogue: [64][256] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tilesM = (p.M + DBM - 1) / DBM;
    int swz;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile_n = swz / tilesM, tile_m = swz - tile_n * tilesM;
    const int m0 = tile_m * DBM, n0 = tile_n * DBN;
    const int hw = p.Hout * p.Wout;
    const int ntaps = p.kh * p.kw;
    const int cpt = p.Cin / DBK;                     // K steps per tap
    const int nk = p.K / DBK;
    const int pix_bytes = p.lda * 2;
    const int b0 = min(m0, p.M - 1) / hw;            // first image of the tile: the base of the buffer resource
    const s16_t* Ab = reinterpret_cast<const s16_t*>(p.A);
    const __amdgpu_buffer_rsrc_t a_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<s16_t*>(Ab + (long)b0 * p.Hin * p.Win * p.lda + p.a_coff), 0, (int)0xffffffffu, 0x00020000);

    // ---- parameter role: lane = (pixel pi of this wave's 16, tap tq of the current block of 4) ----
    const int pi = lane & 15, tq = lane >> 4;
    const int prow = 8 * wave + (pi & 7) + 32 * (pi >> 3);               // tile row of that pixel (the rows this wave gathers)
    int p_iy0, p_ix0; unsigned p_img; bool p_ok; const float* p_om;
    {
        const int m = m0 + prow;
        p_ok = m < p.M;
        const int mm = p_ok ? m : p.M - 1;
        const int b = mm / hw, rem = mm - b * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        p_iy0 = oy * p.stride - p.pad;
        p_ix0 = ox * p.stride - p.pad;
        p_img = (unsigned)(b - b0) * (unsigned)(p.Hin * p.Win) * (unsigned)pix_bytes;
        p_om = p.om + (long)mm * p.om_ld;
    }
    // wave-private tables of 64 entries (pixel pi + 16 tq) x 16 bytes: corner offsets [0, 1024) and, double-buffered by block parity,
    // weights [1024, 3072).  Offsets of a K step are read two steps ahead (gather_issue), its weights one step ahead (gather_finish): the
    // two tables of a block are last read in different iterations, hence the second weight buffer (see the schedule in kstep).
    char* ptab = smem + 2 * (DBM * DBK * 2) + wave * 3072;
    float om_y = 0.f, om_x = 0.f, om_m = 0.f;
    auto param_load = [&](int blk) {                                     // the offsets / modulator logit of (pixel, tap 4 blk + tq)
        const int tap = min(4 * blk + tq, ntaps - 1);
        om_y = p_om[2 * tap]; om_x = p_om[2 * tap + 1]; om_m = p_om[p.om_mask_off + tap];
    };
    auto param_store = [&](int blk) {
        const int tap = min(4 * blk + tq, ntaps - 1);
        const int ky = tap / p.kw, kx = tap - ky * p.kw;
        float mk = om_m;
        if (p.om_sigmoid) mk = (1.0f / (1.0f + __expf(-mk))) * 2.0f;      // aspp.rs:173-174
        const float y = (float)(p_iy0 + ky * p.dil) + om_y;
        const float x = (float)(p_ix0 + kx * p.dil) + om_x;
        const bool inside = p_ok && y > -1.f && y < (float)p.Hin && x > -1.f && x < (float)p.Win;
        const float yf = floorf(y), xf = floorf(x);
        const int yl = (int)yf, xl = (int)xf, yh = yl + 1, xh = xl + 1;
        const float ly = y - yf, lx = x - xf, hy = 1.f - ly, hx = 1.f - lx;
        const float s = inside ? mk : 0.f;
        const bool yl_ok = yl >= 0, yh_ok = yh <= p.Hin - 1, xl_ok = xl >= 0, xh_ok = xh <= p.Win - 1;
        f32x4_d w4;
        w4[0] = (yl_ok && xl_ok) ? s * (hy * hx) : 0.f;
        w4[1] = (yl_ok && xh_ok) ? s * (hy * lx) : 0.f;
        w4[2] = (yh_ok && xl_ok) ? s * (ly * hx) : 0.f;
        w4[3] = (yh_ok && xh_ok) ? s * (ly * lx) : 0.f;
        // clamped corners: always a valid address (a corner that is not real carries a zero weight); the integer clamps also tame NaN / huge offsets
        const int ylc = min(max(yl, 0), p.Hin - 1), yhc = min(max(yh, 0), p.Hin - 1);
        const int xlc = min(max(xl, 0), p.Win - 1), xhc = min(max(xh, 0), p.Win - 1);
        u32x4_d o4;
        o4[0] = p_img + (unsigned)(ylc * p.Win + xlc) * (unsigned)pix_bytes;
        o4[1] = p_img + (unsigned)(ylc * p.Win + xhc) * (unsigned)pix_bytes;
        o4[2] = p_img + (unsigned)(yhc * p.Win + xlc) * (unsigned)pix_bytes;
        o4[3] = p_img + (unsigned)(yhc * p.Win + xhc) * (unsigned)pix_bytes;
        *reinterpret_cast<u32x4_d*>(ptab + lane * 16) = o4;
        *reinterpret_cast<f32x4_d*>(ptab + 1024 + (blk & 1) * 1024 + lane * 16) = w4;
    };

    // ---- gather role: thread = (row gr and gr + 32, 16-byte channel chunk gc) ----
    const int gc = tid & 7, gr = tid >> 3;
    const int e0 = (gr & 7) * 16;                                        // byte offset of this thread's first row in a table (entry pi = gr & 7; second row: pi + 8)
    u32x4_d gv[2][2][4];
    auto gather_issue = [&](auto set_c, int kt) {
        constexpr int S = decltype(set_c)::value;
        const int tap = kt / cpt;
        const unsigned ch = (unsigned)((kt - tap * cpt) * DBK * 2 + gc * 16);
        const char* te = ptab + (tap & 3) * 256 + e0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4_d o = *reinterpret_cast<const u32x4_d*>(te + i * 128);
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[S][i][c] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, o[c] + ch, 0, 0);
        }
    };
    // (the weights of K step kt are read from the table when the step's corners are combined — its block is still the one in the table,
    // see the parameter-block schedule in kstep — so no weight register lives across a K step)
    auto gather_finish = [&](auto set_c, int kt, char* abuf) {
        constexpr int S = decltype(set_c)::value;
        const int tap = kt / cpt;
        const char* te = ptab + 1024 + ((tap >> 2) & 1) * 1024 + (tap & 3) * 256 + e0;
        f32x4_d gw[1][2];
        gw[0][0] = *reinterpret_cast<const f32x4_d*>(te);
        gw[0][1] = *reinterpret_cast<const f32x4_d*>(te + 128);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4_d o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float lo = 0.f, hi = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    lo = fmaf(gw[0][i][c], d16_lo_f32(gv[S][i][c][e]), lo);
                    hi = fmaf(gw[0][i][c], d16_hi_f32(gv[S][i][c][e]), hi);
                }
                o[e] = dpack2(lo, hi);
            }
            *reinterpret_cast<u32x4_d*>(abuf + a_slot(gr + 32 * i, gc)) = o;
        }
    };

    // ---- fragments (as v1) ----
    const char* wf = reinterpret_cast<const char*>(p.Wp) + ((long)((n0 >> 4) + 4 * wave) * nk * 2 * 64 + lane) * 16;
    const long wf_nb = (long)nk * 2 * 1024;
    // A fragment (row block i, half step s): tile row 16 i + (lane & 15), chunk 4 s + (lane >> 4).  Against (i, s) = (0, 0) the swizzled
    // offset only has bit 6 flipped by s and bit 7 by an odd i (16 rows = 8 bank rows: the XOR key's bit 3), + 2 KB per i: one register
    const int a_f0 = a_slot(lane & 15, lane >> 4);
    auto a_foff = [&](int i, int s) { return i * 2048 + (a_f0 ^ (s * 64) ^ ((i & 1) * 128)); };
    f32x4_d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_d{0.f, 0.f, 0.f, 0.f};
    s16x8 wfr[2][4];                                                    // [half step s][n16 block j]
    auto load_w = [&](auto s_c, int kt) {
        constexpr int S = decltype(s_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) wfr[S][j] = *reinterpret_cast<const s16x8*>(wf + j * wf_nb + (long)(kt * 2 + S) * 1024);
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;

    // ---- prologue: parameter block 0, the gathers of K steps 0 and 1, the first W fragments ----
    const int nblk = (ntaps + 3) / 4;
    const int blk_steps = 4 * cpt;                                       // K steps per parameter block
    param_load(0);
    param_store(0);
    if (nblk > 1) param_load(1);
    gather_issue(C0{}, 0);
    if (nk > 1) gather_issue(C1{}, 1);                                   // (block 0 spans blk_steps >= 4 K steps)
    load_w(C0{}, 0);
    gather_finish(C0{}, 0, smem);
    __syncthreads();
    // one K step; PAR = kt & 1 (register set of this step's successor data: static indices after unrolling by two)
    auto kstep = [&](auto par_c, int kt) {
        constexpr int PAR = decltype(par_c)::value;
        using CP = std::integral_constant<int, PAR>; using CN = std::integral_constant<int, 1 - PAR>;
        const char* abuf = smem + PAR * (DBM * DBK * 2);
        load_w(C1{}, kt);
        if (kt + 2 < nk) gather_issue(CP{}, kt + 2);                      // set PAR held step kt: consumed by gather_finish in the previous iteration
        {
            s16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const s16x8*>(abuf + a_foff(i, 0));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = BRN_MFMA_16X16X32(wfr[0][j], af[i], acc[i][j]);
        }
        if (kt + 1 < nk) load_w(C0{}, kt + 1);
        {
            s16x8 af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const s16x8*>(abuf + a_foff(i, 1));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i][j] = BRN_MFMA_16X16X32(wfr[1][j], af[i], acc[i][j]);
        }
        if (kt + 1 < nk) gather_finish(CN{}, kt + 1, smem + (1 - PAR) * (DBM * DBK * 2));
        // offsets of block g are first read by gather_issue(kt' + 2) with kt' + 2 = g blk_steps; those of block g - 1 were last read by
        // gather_issue(kt + 2) above when kt + 3 = g blk_steps: here, and only here, the offset table can be rewritten.  The weights of
        // block g - 1 are still read by gather_finish in the NEXT iteration (K step g blk_steps - 1): block g's go to the other buffer,
        // whose last reader (block g - 2) is more than a block behind.
        if ((kt + 3) % blk_steps == 0) {
            const int g = (kt + 3) / blk_steps;
            if (g < nblk) {
                param_store(g);
                if (g + 1 < nblk) param_load(g + 1);
            }
        }
        __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { kstep(C0{}, kt); kstep(C1{}, kt + 1); }
    if (kt < nk) kstep(C0{}, kt);

    // ---- epilogue (as v1): lane holds pixel 16 i + (lane & 15), channels 64 wave + 16 j + 4 (lane >> 4) + {0..3} ----
    {
        const int q4 = lane >> 4, pr = lane & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = 64 * wave + 16 * j + 4 * q4;
            const int n = n0 + nl;
            f32x4_d bias = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N) {
                if (p.bias) bias = *reinterpret_cast<const f32x4_d*>(p.bias + n);
                if (p.scale) { sc = *reinterpret_cast<const f32x4_d*>(p.scale + n); sh = *reinterpret_cast<const f32x4_d*>(p.shift + n); }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4_d v = (acc[i][j] + bias) * sc + sh;
                if (p.act == ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const int row = 16 * i + pr;
                const int chunk = (nl >> 3) ^ (row & 31);
                const u32x2_d o = {dpack2(v[0], v[1]), dpack2(v[2], v[3])};
                *reinterpret_cast<u32x2_d*>(smem + row * (DBN * 2) + chunk * 16 + (nl & 4) * 2) = o;
            }
        }
    }
    __syncthreads();
    {
        s16_t* Cb = reinterpret_cast<s16_t*>(p.C);
#pragma unroll
        for (int ps = 0; ps < DBM / 8; ++ps) {
            const int row = ps * 8 + (tid >> 5), c = tid & 31;
            const int m = m0 + row, n = n0 + c * 8;
            const u32x4_d v = *reinterpret_cast<const u32x4_d*>(smem + row * (DBN * 2) + ((c ^ (row & 31)) << 4));
            if (m < p.M && n < p.N) *reinterpret_cast<u32x4_d*>(Cb + (long)m * p.ldc + p.c_coff + n) = v;
        }
    }
}

bool deform_bf16_eligible(const GemmParams& p) {
    return p.mode == GEMM_DEFORM_NHWC && p.Wp && p.om && p.Cin >= DBK && (p.Cin % DBK) == 0 && p.K == p.kh * p.kw * p.Cin && (p.N & 7) == 0 &&
           ((p.lda | p.a_coff | p.ldc | p.c_coff) & 7) == 0 && !p.R && !p.bbias && !p.c_f32 && p.act != ACT_GELU_ERF &&
           (double)p.Hin * p.Win * p.lda * 2.0 < 2147483648.0;
}

hipError_t launch_deform_bf16(const GemmParams& p, hipStream_t s) {
    if (!BRN_S16_SELF(deform_bf16_eligible)(p) || p.M <= 0) return hipErrorInvalidValue;
    const int tiles = ((p.M + DBM - 1) / DBM) * ((p.N + DBN - 1) / DBN);
    // BRN_DEFORM_V=1: the round-3 kernel (every thread computes its own sampling parameters), for same-box A/B runs and the bit-equality test
    static const int ver = getenv("BRN_DEFORM_V") ? atoi(getenv("BRN_DEFORM_V")) : 2;
    // v2 addresses the corners as 32-bit byte offsets from the tile's first image: the whole batch of maps must span < 4 GiB
    const double map_bytes = (double)p.M / ((double)p.Hout * p.Wout) * p.Hin * p.Win * p.lda * 2.0;
    if (ver != 1 && map_bytes < 4294967296.0) hipLaunchKernelGGL(gemm_deform_bf16_v2_kernel<0>, dim3(tiles), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_deform_bf16_kernel, dim3(tiles), dim3(256), 0, s, p);
    return hipGetLastError();
}

#if BRN_S16_F16
}  // namespace hf
#endif
}  // namespace brn
