// brn_graph.cpp — the forward graph: BiRefNet::forward_logits (birefnet.rs:412-461), SwinTransformer::forward
// (swin.rs:768-797), BiRefNetDecoder::forward (birefnet.rs:278-376), BasicDecBlk/ASPPDeformable (decoder.rs:126-141,
// aspp.rs:303-333) expressed as launches of the gfx950 kernels on one HIP stream, activations channels-last in an HBM
// arena, every concatenation written in place into column windows of its consumer's input map.
#include "brn_host.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>

namespace brn {

static inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

// ---- arena ----------------------------------------------------------------------------------------------------------
float* Arena::alloc(size_t nfloats) {
    const size_t bytes = (nfloats * sizeof(float) + 255) & ~(size_t)255;
    const size_t off = top;
    top += bytes;
    if (top > peak) peak = top;
    if (dry) return reinterpret_cast<float*>(off);   // never dereferenced in a dry run
    if (top > cap) fail(BRN_ERR_OOM, "workspace arena overflow: need %zu bytes, have %zu", top, cap);
    return reinterpret_cast<float*>(base + off);
}

Map new_map(Ctx& c, int B, int H, int W, int C) {
    Map m;
    m.B = B; m.H = H; m.W = W; m.C = C; m.ld = C; m.coff = 0;
    m.p = c.act_alloc((size_t)B * H * W * C);      // fp32 map, or bf16 in compute mode BRN_BF16
    return m;
}

// ---- launch bracket -----------------------------------------------------------------------------------------------
struct Bracket {
    Ctx& c; bool on;
    Bracket(Ctx& c_, int fam, double flop, double bytes, int M = 0, int N = 0, int K = 0) : c(c_), on(c_.profile && !c_.dry) {
        if (!on) return;
        auto next = [&]() -> hipEvent_t {
            if (*c.event_next >= c.event_pool->size()) {
                hipEvent_t e; BRN_HIP(hipEventCreate(&e)); c.event_pool->push_back(e);
            }
            return (*c.event_pool)[(*c.event_next)++];
        };
        LaunchRecord r; r.fam = fam; r.flop = flop; r.bytes = bytes; r.e0 = next(); r.e1 = next(); r.M = M; r.N = N; r.K = K; r.region = c.region;
        BRN_HIP(hipEventRecord(r.e0, c.stream));
        c.records->push_back(r);
    }
    ~Bracket() { if (on) (void)hipEventRecord(c.records->back().e1, c.stream); }
};
#define BRN_LAUNCH(expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) fail(BRN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
    } while (0)

// ---- graph branches on auxiliary streams --------------------------------------------------------------------------------
// While a Branch is alive, launches go to aux stream k, ordered after everything enqueued on the main stream so far; join_branches
// makes the main stream wait for every branch enqueued since the last join.  Same kernels, same arguments, same results: only
// the order in which independent launches may run changes.  Buffers a branch writes must stay allocated until the join (Arena::hold).
struct Branch {
    Ctx& c; hipStream_t main; int k; bool on;
    Branch(Ctx& c_, int k_) : c(c_), main(c_.stream), k(k_), on(c_.br && !c_.dry && !c_.profile && k_ >= 0 && k_ < BRN_AUX_STREAMS && ((c_.br_mask >> k_) & 1u)) {
        if (!on) return;
        BRN_HIP(hipEventRecord(c.br->fork_ev[k], main));
        BRN_HIP(hipStreamWaitEvent(c.br->stream[k], c.br->fork_ev[k], 0));
        c.stream = c.br->stream[k];
    }
    ~Branch() {
        if (!on) return;
        (void)hipEventRecord(c.br->join_ev[k], c.stream);
        c.stream = main;
        c.pending |= 1u << k;
    }
};
static void join_branches(Ctx& c, unsigned mask) {
    for (int k = 0; k < BRN_AUX_STREAMS; ++k)
        if (c.pending & mask & (1u << k)) BRN_HIP(hipStreamWaitEvent(c.stream, c.br->join_ev[k], 0));
    c.pending &= ~mask;
}
constexpr int AUX_IPT = 3, AUX_LAT = 4;
constexpr unsigned AUX_ASPP_MASK = 7u;
struct ArenaHold {
    Arena& a;
    explicit ArenaHold(Arena& a_) : a(a_) { ++a.hold; }
    ~ArenaHold() { --a.hold; }
};

static bool att_h2_on() { static const bool on = !(getenv("BRN_H2_ATT") && atoi(getenv("BRN_H2_ATT")) == 0); return on; }
// compute mode BRN_F16 (c.bf16 == 2): the 16-bit kernels of namespace brn::hf (fp16 storage / MFMA operands)
#define S16F(C_, FN_) ((C_).bf16 == 2 ? hf::FN_ : FN_)
// ---- GEMM-shaped pieces -------------------------------------------------------------------------------------------------
static void fill_epilogue(GemmParams& p, const GemmW& w) {
    p.bias = w.bias; p.scale = w.scale; p.shift = w.shift; p.act = w.act;
    p.Wp = w.wp; p.planes = w.planes; p.wp_rows = w.wp_rows;
    p.h2 = w.half;
    if (w.half) { p.a_scale = half2_act_scale(); p.out_scale = 1.f / (p.a_scale * w.w_scale); }
}

// bf16-storage mode: the same GEMM on kernels/gemm_bf16.hip (A bf16; C / R bf16 unless flagged fp32)
static void run_gemm_bf16(Ctx& c, const GemmW& w, GemmParams& p, int fam) {
    if (!w.wb) fail(BRN_ERR_INVALID_ARG, "bf16 mode: weight without a bf16 copy");
    const GemmPlan pl = S16F(c, plan_gemm_bf16)(p.M, p.N, p.K, p.c_f32 && p.R && p.r_f32, p.act == ACT_GELU_ERF);
    const size_t mk = c.arena->mark();
    float* ws = pl.ws_floats ? c.arena->alloc(pl.ws_floats) : nullptr;
    c.arena->release(mk);
    if (c.dry) return;
    p.Wp = w.wb; p.wp_rows = w.wb_rows; p.wp_ld = w.wb_ld; p.planes = 1;
    static const int wstat_mask = getenv("BRN_WSTAT") ? atoi(getenv("BRN_WSTAT")) : 3;       // bit 0: K = 192, bit 1: K = 384
    if (w.wf && w.mode == GEMM_DENSE && (wstat_mask & (p.K == 384 ? 2 : 1))) {   // short K, wide A: the weights stay in registers (gemm_wstat_bf16_kernel)
        GemmParams q = p;
        q.Wp = w.wf;
        if (S16F(c, gemm_wstat_eligible)(q)) {
            const double flop_ = 2.0 * q.M * (double)q.N * q.K;
            const double bytes_ = 2.0 * ((double)q.M * q.K + (double)q.N * q.K + (double)q.M * q.N);
            Bracket b(c, fam, flop_, bytes_, q.M, q.N, q.K);
            BRN_LAUNCH(S16F(c, launch_gemm_wstat)(q, c.stream));
            return;
        }
    }
    const double flop = 2.0 * p.M * (double)p.N * p.K;
    const double a_elems = p.mode == GEMM_DENSE ? (double)p.M * p.K : (double)p.M / ((double)p.Hout * p.Wout) * p.Hin * p.Win * p.Cin;
    const double bytes = 2.0 * (a_elems + (double)p.N * p.K) + (p.c_f32 ? 4.0 : 2.0) * (double)p.M * p.N + (p.R ? (p.r_f32 ? 4.0 : 2.0) * (double)p.M * p.N : 0.0);
    Bracket b(c, fam, flop, bytes, p.M, p.N, p.K);
    const hipError_t e = S16F(c, launch_gemm_bf16)(p, pl, ws, c.stream);
    if (e != hipSuccess)
        fail(BRN_ERR_HIP, "launch_gemm_bf16 (M %d, N %d, K %d, mode %d, Cin %d, %d x %d -> %d x %d, lda %d + %d, ldc %d + %d, tile cfg %d, split-K %d, chunk-major %d): %s",
             p.M, p.N, p.K, p.mode, p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.lda, p.a_coff, p.ldc, p.c_coff, pl.cfg, pl.splitk, p.k_chunk_major, hipGetErrorString(e));
}

void run_gemm(Ctx& c, const GemmW& w, const float* A, int M, int lda, float* C, int ldc, int c_coff, const float* R, int ldr,
              int r_coff, const float* bbias, int bbias_rows, int a_planes, int c_planes, int c_f32, int r_f32) {
    if (c.bf16) {
        GemmParams p{};
        p.A = A; p.C = C; p.M = M; p.N = w.N; p.K = w.K; p.mode = GEMM_DENSE; p.lda = lda;
        p.bias = w.bias; p.scale = w.scale; p.shift = w.shift; p.act = w.act;
        p.bbias = bbias; p.bbias_rows = bbias_rows > 0 ? bbias_rows : 1;
        p.R = R; p.ldr = ldr; p.r_coff = r_coff; p.ldc = ldc; p.c_coff = c_coff;
        p.c_f32 = c_f32; p.r_f32 = r_f32;
        run_gemm_bf16(c, w, p, FAM_GEMM_DENSE);
        return;
    }
    GemmParams p{};
    p.A = A; p.W = w.w; p.C = C; p.M = M; p.N = w.N; p.K = w.K; p.mode = GEMM_DENSE; p.lda = lda;
    fill_epilogue(p, w);
    p.bbias = bbias; p.bbias_rows = bbias_rows > 0 ? bbias_rows : 1;
    p.R = R; p.ldr = ldr; p.r_coff = r_coff; p.ldc = ldc; p.c_coff = c_coff;
    p.a_planes = a_planes; p.c_planes = c_planes;
    // A already split by its producer (P layout): the LDS-DMA plane kernel (kernels/gemm_planes.hip), when the shape allows
    // (opt-in, BRN_PLANES_KERNEL=1: at batch 1 it measured 10-12 % SLOWER per forward than the warp-specialised kernel in both
    // split modes — MFMA utilisation 0.40 vs 0.44, profiles/r02_pmc_sq_c2_planes.csv — see DESIGN.md 3.1c)
    // — the kernel lives in the diag build only (make diag), the product library does not carry it)
#ifdef BRN_DIAG_BUILD
    static const bool planes_on = getenv("BRN_PLANES_KERNEL") && atoi(getenv("BRN_PLANES_KERNEL")) != 0;
    const bool planes_kernel = planes_on && a_planes && gemm_planes_eligible(p);
#else
    constexpr bool planes_kernel = false;
#endif
#ifdef BRN_DIAG_BUILD
    GemmPlan pl = planes_kernel ? plan_gemm_planes(M, w.N, w.K, w.planes, c_planes != 0) : plan_gemm(M, w.N, w.K, w.wp ? w.planes : 0);
#else
    GemmPlan pl = plan_gemm(M, w.N, w.K, w.wp ? w.planes : 0);
#endif
    if (!planes_kernel && (a_planes || c_planes)) {
        // otherwise P operands exist only on the warp-specialised kernel; a P output cannot go through the split-K reduce pass
        if (!(w.wp && (w.planes == 2 || w.planes == 3))) fail(BRN_ERR_INVALID_ARG, "P activation layout outside the split modes");
        pl.cfg = 0;
        if (c_planes) { pl.splitk = 1; pl.ws_floats = 0; }
    }
    const size_t mk = c.arena->mark();
    float* ws = pl.ws_floats ? c.arena->alloc(pl.ws_floats) : nullptr;
    c.arena->release(mk);          // scratch is dead as soon as the reduce pass has been enqueued (in-order stream)
    if (c.dry) return;
    const double flop = 2.0 * M * (double)w.N * w.K;
    const double bytes = 4.0 * ((double)M * w.K + (double)w.N * w.K + (double)M * w.N * (R ? 2 : 1));
    Bracket b(c, FAM_GEMM_DENSE, flop, bytes, M, w.N, w.K);
#ifdef BRN_DIAG_BUILD
    if (planes_kernel) { BRN_LAUNCH(launch_gemm_planes(p, pl, ws, c.stream)); return; }
#endif
    BRN_LAUNCH(launch_gemm(p, pl, ws, c.stream));
}

// geometry + operands of a deformable conv on kernels/deform_bf16.hip (compute mode BRN_BF16)
static GemmParams deform_bf16_params(const GemmW& w, const Map& in, const Map& out, int Hout, int Wout, const float* om, int om_ld, int om_mask_off,
                                     int om_sigmoid) {
    GemmParams p{};
    p.A = in.p; p.C = out.p; p.M = out.B * Hout * Wout; p.N = w.N; p.K = w.K; p.mode = GEMM_DEFORM_NHWC;
    p.lda = in.ld; p.a_coff = in.coff;
    p.Hin = in.H; p.Win = in.W; p.Cin = w.Cinp; p.kh = w.kh; p.kw = w.kw; p.stride = w.stride; p.pad = w.pad; p.dil = w.dil;
    p.Hout = Hout; p.Wout = Wout; p.Kreal = w.Kreal;
    p.om = om; p.om_ld = om_ld; p.om_mask_off = om_mask_off; p.om_sigmoid = om_sigmoid;
    p.bias = w.bias; p.scale = w.scale; p.shift = w.shift; p.act = w.act;
    p.bbias_rows = 1; p.ldc = out.ld; p.c_coff = out.coff;
    p.Wp = w.wf; p.planes = 1;
    return p;
}
bool deform_fused_sigmoid(const Ctx& c, const GemmW& w) {
    static const bool off = getenv("BRN_DEFORM_F32_KERNEL") && atoi(getenv("BRN_DEFORM_F32_KERNEL")) != 0;   // A/B: the fp32-MFMA gather kernel
    return c.bf16 && w.mode == GEMM_DEFORM_NHWC && w.wf && !off && w.Cinp % 64 == 0 && (w.N & 7) == 0 && w.act != ACT_GELU_ERF;
}

void run_conv(Ctx& c, const GemmW& w, const Map& in, const Map& out, const float* om, int om_ld, int om_mask_off, int c_f32, int om_sigmoid) {
    const int Hout = (in.H + 2 * w.pad - w.dil * (w.kh - 1) - 1) / w.stride + 1;
    const int Wout = (in.W + 2 * w.pad - w.dil * (w.kw - 1) - 1) / w.stride + 1;
    if (out.H != Hout || out.W != Wout || out.B != in.B || out.C != w.N)
        fail(BRN_ERR_INVALID_ARG, "conv output map [%d,%d,%d,%d] does not match expected [%d,%d,%d,%d]", out.B, out.H, out.W,
             out.C, in.B, Hout, Wout, w.N);
    if (in.C > w.Cinp || in.coff + w.Cinp > in.ld)
        fail(BRN_ERR_INVALID_ARG, "conv input window (C=%d coff=%d ld=%d) cannot supply %d channels", in.C, in.coff, in.ld, w.Cinp);
    const int M = out.B * Hout * Wout;
    if (w.mode == GEMM_DENSE) {
        run_gemm(c, w, c.at(in.p, in.coff), M, in.ld, out.p, out.ld, out.coff, nullptr, 0, 0, nullptr, 0, 0, 0, c_f32);
        return;
    }
    if (c.bf16 && w.mode == GEMM_CONV_NHWC) {
        GemmParams p{};
        p.A = in.p; p.C = out.p; p.M = M; p.N = w.N; p.K = w.K; p.mode = GEMM_CONV_NHWC;
        p.lda = in.ld; p.a_coff = in.coff;
        p.Hin = in.H; p.Win = in.W; p.Cin = w.Cinp; p.kh = w.kh; p.kw = w.kw; p.stride = w.stride; p.pad = w.pad; p.dil = w.dil;
        p.Hout = Hout; p.Wout = Wout; p.Kreal = w.Kreal;
        p.bias = w.bias; p.scale = w.scale; p.shift = w.shift; p.act = w.act;
        p.bbias_rows = 1; p.ldc = out.ld; p.c_coff = out.coff; p.c_f32 = c_f32;
        p.k_chunk_major = w.wb_chunk_major;
        run_gemm_bf16(c, w, p, FAM_GEMM_CONV);
        return;
    }
    if (deform_fused_sigmoid(c, w) && !c_f32) {
        if (!om) fail(BRN_ERR_INVALID_ARG, "deformable conv without an offset/modulator map");
        GemmParams p = deform_bf16_params(w, in, out, Hout, Wout, om, om_ld, om_mask_off, om_sigmoid);
        if (S16F(c, deform_bf16_eligible)(p)) {
            if (c.dry) return;
            const double flop = 2.0 * M * (double)w.N * w.K;
            // algorithmic bytes: the sampled map once, the offset / modulator map, the weights, the result
            const double bytes = 2.0 * ((double)in.pixels() * w.Cinp + (double)w.N * w.K + (double)M * w.N) + 4.0 * M * 3.0 * w.kh * w.kw;
            Bracket b(c, FAM_GEMM_DEFORM, flop, bytes, M, w.N, w.K);
            BRN_LAUNCH(S16F(c, launch_deform_bf16)(p, c.stream));
            return;
        }
        if (om_sigmoid) fail(BRN_ERR_INVALID_ARG, "deformable conv: raw modulator logits passed to a shape the bf16 gather kernel does not cover");
    } else if (om_sigmoid) fail(BRN_ERR_INVALID_ARG, "deformable conv: raw modulator logits outside the bf16 gather kernel");
    GemmPlan pl = plan_gemm(M, w.N, w.K, (w.wp && w.mode == GEMM_CONV_NHWC) ? w.planes : 0);
    if (c.bf16) { pl.splitk = 1; pl.ws_floats = 0; }     // (the fp32 split-K reduce pass has no bf16 output; deformable convs only)
    const size_t mk = c.arena->mark();
    float* ws = pl.ws_floats ? c.arena->alloc(pl.ws_floats) : nullptr;
    c.arena->release(mk);
    if (c.dry) return;
    GemmParams p{};
    p.A = in.p; p.W = w.w; p.C = out.p; p.M = M; p.N = w.N; p.K = w.K; p.mode = w.mode;
    p.lda = in.ld; p.a_coff = in.coff;
    p.Hin = in.H; p.Win = in.W; p.Cin = w.Cinp; p.kh = w.kh; p.kw = w.kw; p.stride = w.stride; p.pad = w.pad; p.dil = w.dil;
    p.Hout = Hout; p.Wout = Wout; p.Kreal = w.Kreal;
    p.om = om; p.om_ld = om_ld; p.om_mask_off = om_mask_off;
    fill_epilogue(p, w);
    p.bbias_rows = 1; p.ldc = out.ld; p.c_coff = out.coff;
    if (c.bf16) { p.a_bf16 = c.bf16; p.c_bf16 = c_f32 ? 0 : c.bf16; }   // deformable gather in bf16 mode: bf16 map in / out on the fp32-MFMA kernel
    if (w.mode == GEMM_DEFORM_NHWC && !om) fail(BRN_ERR_INVALID_ARG, "deformable conv without an offset/modulator map");
    const double flop = 2.0 * M * (double)w.N * w.K;
    const double bytes = 4.0 * ((double)in.pixels() * w.Cinp + (double)w.N * w.K + (double)M * w.N);
    Bracket b(c, w.mode == GEMM_DEFORM_NHWC ? FAM_GEMM_DEFORM : FAM_GEMM_CONV, flop, bytes, M, w.N, w.K);
    BRN_LAUNCH(launch_gemm(p, pl, ws, c.stream));
}

void run_conv_nchw(Ctx& c, const GemmW& w, const float* x, int B, int Hin, int Win, const Map& out, bool pad_to_stride) {
    int Hout = (Hin + 2 * w.pad - w.dil * (w.kh - 1) - 1) / w.stride + 1;
    int Wout = (Win + 2 * w.pad - w.dil * (w.kw - 1) - 1) / w.stride + 1;
    // PatchEmbed (swin.rs:696-702) first pads the image with zeros on the right / bottom to a multiple of the patch: for a
    // k == stride, pad 0 conv that is the ceil-mode output size, and the gather loader already returns 0 beyond the border
    if (pad_to_stride && w.kh == w.stride && w.kw == w.stride && w.pad == 0 && w.dil == 1) { Hout = (Hin + w.stride - 1) / w.stride; Wout = (Win + w.stride - 1) / w.stride; }
    if (out.H != Hout || out.W != Wout || out.B != B || out.C != w.N)
        fail(BRN_ERR_INVALID_ARG, "conv(nchw) output map mismatch");
    const int M = B * Hout * Wout;
    const GemmPlan pl = plan_gemm(M, w.N, w.K);
    const size_t mk = c.arena->mark();
    float* ws = pl.ws_floats ? c.arena->alloc(pl.ws_floats) : nullptr;
    c.arena->release(mk);
    if (c.dry) return;
    GemmParams p{};
    p.A = x; p.W = w.w; p.C = out.p; p.M = M; p.N = w.N; p.K = w.K; p.mode = GEMM_GATHER_NCHW;
    p.Hin = Hin; p.Win = Win; p.Cin = w.Cin; p.kh = w.kh; p.kw = w.kw; p.stride = w.stride; p.pad = w.pad; p.dil = w.dil;
    p.Hout = Hout; p.Wout = Wout; p.Kreal = w.Kreal;
    fill_epilogue(p, w);
    p.bbias_rows = 1; p.ldc = out.ld; p.c_coff = out.coff;
    const double flop = 2.0 * M * (double)w.N * w.Kreal;
    const double bytes = 4.0 * ((double)B * w.Cin * Hin * Win + (double)w.N * w.K + (double)M * w.N);
    Bracket b(c, FAM_GEMM_GATHER, flop, bytes, M, w.N, w.K);
    BRN_LAUNCH(launch_gemm(p, pl, ws, c.stream));
}

void run_layernorm(Ctx& c, const LNW& ln, const float* x, int rows, int ldx, float* y, int ldy, int y_coff, int y_planes, int y_bf16) {
    if (c.dry) return;
    LayerNormParams p{};
    p.x = x; p.y = y; p.rows = rows; p.C = ln.C; p.gamma = ln.g; p.beta = ln.b; p.eps = 1e-5f;
    p.ldx = ldx; p.ldy = ldy; p.y_coff = y_coff; p.mode = 0; p.y_planes = y_planes; p.y_bf16 = y_bf16;
    p.y_h2 = y_planes == 2 ? c.h2_scale : 0.f;
    Bracket b(c, FAM_LAYERNORM, 0.0, (y_bf16 ? 6.0 : 8.0) * rows * (double)ln.C, rows, ln.C, 0);
    BRN_LAUNCH(launch_layernorm(p, c.stream));
}

void run_resize(Ctx& c, const Map& in, const Map& out, bool accumulate) {
    if (in.C != out.C || in.B != out.B) fail(BRN_ERR_INVALID_ARG, "resize: channel/batch mismatch");
    if (c.dry) return;
    Bracket b(c, FAM_RESIZE, 0.0, (double)c.esz() * ((double)in.pixels() + (double)out.pixels() * (accumulate ? 2 : 1)) * in.C);
    BRN_LAUNCH(launch_resize_nhwc(in.p, in.B, in.H, in.W, in.C, in.ld, in.coff, out.p, out.H, out.W, out.ld, out.coff, c.stream, c.bf16, accumulate ? 1 : 0));
}

// ---- Swin --------------------------------------------------------------------------------------------------------------
void swin_stage_dims(int H, int W, int patch, int hs[4], int ws[4]) {
    int h = (H + patch - 1) / patch, w = (W + patch - 1) / patch;   // PatchEmbed pads to a multiple (swin.rs:696-702)
    for (int i = 0; i < 4; ++i) {
        hs[i] = h; ws[i] = w;
        h = (h + 1) / 2; w = (w + 1) / 2;                            // swin.rs:595
    }
}

// proj + residual + the block's second LayerNorm in one launch (compute mode BRN_BF16: gemm_wstat_ln_bf16_kernel for C = 192, M >= 32768;
// gemm_rowln_bf16_kernel for C = 768 / 384, M >= 4096);
// false = not applicable, nothing enqueued (no workspace is involved either way, so a dry run and a real run agree trivially)
static bool run_gemm_ln(Ctx& c, const GemmW& w, const float* A, int M, int lda, float* x, const LNW& ln, float* y, int ldy) {
    return linear_residual_ln(c, w, A, M, lda, x, ln, y, ldy, false);
}
bool linear_residual_ln(Ctx& c, const GemmW& w, const float* A, int M, int lda, float* x, const LNW& ln, float* y, int ldy, bool every_fused_kernel) {
    static const bool off = getenv("BRN_WSTAT_LN") && atoi(getenv("BRN_WSTAT_LN")) == 0;
    // gemm_rowln_bf16_kernel (N = 768 / 384) is built and tested but NOT used by the model by default: it streams the whole W through LDS per
    // 64 rows, and that L2 -> LDS intake costs what the saved fp32 re-read of x is worth (measured at c3: 159 us per stage-2 launch against
    // 96 + 36 us for projection + LayerNorm; -0.6 % end to end).  BRN_ROWLN: bit 0 = N 768, bit 1 = N 384.
    static const int rowln_env = getenv("BRN_ROWLN") ? atoi(getenv("BRN_ROWLN")) : 0;
    const int rowln_mask = every_fused_kernel ? 3 : rowln_env;
    if (!c.bf16 || w.mode != GEMM_DENSE || ln.C != w.N || !ln.g || !ln.b) return false;
    GemmParams p{};
    p.A = A; p.C = x; p.M = M; p.N = w.N; p.K = w.K; p.mode = GEMM_DENSE; p.lda = lda;
    p.bias = w.bias; p.scale = w.scale; p.shift = w.shift; p.act = w.act;
    p.R = x; p.ldr = w.N; p.ldc = w.N; p.c_f32 = 1; p.r_f32 = 1;
    p.planes = 1;
    // wide stages: the workgroup owns 64 whole rows and W streams through LDS (gemm_rowln_bf16_kernel)
    if (w.wb && (rowln_mask & (w.N == 768 ? 1 : w.N == 384 ? 2 : 0))) {
        p.Wp = w.wb; p.wp_rows = w.wb_rows; p.wp_ld = w.wb_ld;
        if (S16F(c, gemm_rowln_eligible)(p)) {
            if (c.dry) return true;
            const double flop = 2.0 * M * (double)w.N * w.K;
            const double bytes = 2.0 * ((double)M * w.K + (double)w.N * w.K) + (4.0 + 4.0 + 2.0) * (double)M * w.N;
            Bracket b(c, FAM_GEMM_DENSE, flop, bytes, M, w.N, w.K);
            BRN_LAUNCH(S16F(c, launch_gemm_rowln)(p, ln.g, ln.b, 1e-5f, y, ldy, c.stream));
            return true;
        }
    }
    if (off || !w.wf) return false;
    p.Wp = w.wf; p.wp_rows = 0; p.wp_ld = 0;
    if (!S16F(c, gemm_wstat_ln_eligible)(p)) return false;
    if (c.dry) return true;
    const double flop = 2.0 * M * (double)w.N * w.K;
    const double bytes = 2.0 * ((double)M * w.K + (double)w.N * w.K) + (4.0 + 4.0 + 2.0) * (double)M * w.N;
    Bracket b(c, FAM_GEMM_DENSE, flop, bytes, M, w.N, w.K);
    BRN_LAUNCH(S16F(c, launch_gemm_wstat_ln)(p, ln.g, ln.b, 1e-5f, y, ldy, c.stream));
    return true;
}

// The attention half of a block for `nin` token sets that share the weights (the full- and half-scale backbone passes
// of birefnet.rs:416,426 are run as ONE pass over concatenated token rows: every per-token op sees M = M_full + M_half).
// ln2 / xn2: the block's norm2 and its output matrix — when the projection can take the LayerNorm into its epilogue (run_gemm_ln)
// it is done here and the function returns true.
static bool swin_attention_multi(Ctx& c, const SwinBlockW& blk, const float* xn, int B, int nin, const int* hs, const int* wsz, int C,
                                 int shift, float* y, const float* residual, int p2 = 0, int window = 12, const LNW* ln2 = nullptr,
                                 float* xn2 = nullptr, int ld_xn2 = 0) {
    const size_t mk = c.arena->mark();
    int M = 0;
    for (int k = 0; k < nin; ++k) M += B * hs[k] * wsz[k];
    float* qkv = c.act_alloc((size_t)M * 3 * C);                        // (bf16 in compute mode BRN_BF16, like att and xn)
    const int ldp = p2 ? C * p2 / 2 : C;                                // row stride (floats) of a P-layout [M][C] buffer
    float* att = c.act_alloc((size_t)M * ldp);
    run_gemm(c, blk.qkv, xn, M, ldp, qkv, 3 * C, 0, nullptr, 0, 0, nullptr, 0, p2, 0);   // swin.rs:217 (pad rows are synthesised by the kernel)
    if (!c.dry) {
        // one launch for all maps of the pass (full + half scale): fewer ramps and tails than one launch per geometry
        WindowAttnParams ps[2]{};
        size_t off = 0;
        double nwin = 0.0;
        for (int k = 0; k < nin; ++k) {
            WindowAttnParams& p = ps[k];
            p.qkv = c.at(qkv, off * 3 * C); p.qkv_bias = blk.qkv.bias; p.rel_table = blk.rel_table; p.out = c.at(att, off * ldp);
            p.io_bf16 = c.bf16;
            p.B = B; p.H = hs[k]; p.W = wsz[k]; p.C = C; p.heads = blk.heads;
            p.Hp = roundup(hs[k], window); p.Wp = roundup(wsz[k], window);   // swin.rs:359-360
            p.ws = window;
            p.shift = shift; p.scale = 1.0f / sqrtf(32.0f);          // head_dim^-0.5 (swin.rs:134)
            p.planes = (!c.bf16 && window == 12 && (blk.qkv.planes == 2 || blk.qkv.planes == 1)) ? blk.qkv.planes : 0;
            p.h2 = (p.planes == 2 && blk.qkv.half && att_h2_on()) ? 1 : 0;
            if (blk.qkv.half && !p.h2) p.planes = 0;        // BRN_H2_ATT=0: the fp32-MFMA kernel, like f32_split3
            p.out_planes = p2;
            p.out_h2 = (p2 == 2 && blk.qkv.half) ? c.h2_scale : 0.f;
            nwin += (double)B * (p.Hp / window) * (p.Wp / window) * blk.heads;
            off += (size_t)B * hs[k] * wsz[k];
        }
        const double ntok = (double)window * window;
        Bracket b(c, FAM_ATTENTION, nwin * 2.0 * 2.0 * ntok * ntok * 32, (double)c.esz() * ((double)M * 4 * C), M, C, shift);
        BRN_LAUNCH(launch_window_attention2(ps[0], nin > 1 ? &ps[1] : nullptr, c.stream));
    }
    // swin.rs:310 (+ shortcut, swin.rs:406); the residual stream y / residual stays fp32 in every mode
    bool ln_done = false;
    if (ln2 && xn2 && !p2 && y == residual) ln_done = run_gemm_ln(c, blk.proj, att, M, ldp, y, *ln2, xn2, ld_xn2);
    if (!ln_done) run_gemm(c, blk.proj, att, M, ldp, y, C, 0, residual, C, 0, nullptr, 0, p2, 0, c.bf16, c.bf16);
    c.arena->release(mk);
    return ln_done;
}

void swin_attention(Ctx& c, const SwinBlockW& blk, const float* xn, int B, int H, int W, int C, int shift, float* y,
                    const float* residual, int window) {
    swin_attention_multi(c, blk, xn, B, 1, &H, &W, C, shift, y, residual, 0, window);
}

void swin_forward_multi(Ctx& c, const SwinW& w, const SwinIn* ins, int nin, int B, bool outs_f32) {
    if (nin < 1 || nin > 2) fail(BRN_ERR_INVALID_ARG, "swin_forward_multi: 1 or 2 inputs");
    int hs[2][4], wsz[2][4];
    for (int k = 0; k < nin; ++k) swin_stage_dims(ins[k].H, ins[k].W, w.patch, hs[k], wsz[k]);
    auto rows = [&](int k, int i) { return B * hs[k][i] * wsz[k][i]; };
    auto total = [&](int i) { int m = 0; for (int k = 0; k < nin; ++k) m += rows(k, i); return m; };
    const size_t mk0 = c.arena->mark();
    const int E = w.embed_dim;
    // PatchEmbed (swin.rs:692-714): conv k4 s4 straight from the NCHW image (zero beyond the border = pad_with_zeros), LN
    float* x = c.arena->alloc((size_t)total(0) * E);
    // compute mode BRN_BF16, Swin-L geometry, image sides multiples of 4: conv + bias + LayerNorm in one kernel per image scale
    // (kernels/patch_embed.hip) — neither the conv output nor a second pass over it touches HBM
    static const int pe_env = getenv("BRN_PATCH_LN") ? atoi(getenv("BRN_PATCH_LN")) : 1;     // 0: two kernels; 2: fused without the first block's norm1
    const bool pe_off = pe_env == 0;
    bool pe_fused = c.bf16 && !pe_off && w.patch_proj.w && w.patch_proj.mode == GEMM_GATHER_NCHW && w.patch_proj.pad == 0 && w.patch_proj.dil == 1 &&
                    w.patch_proj.kh == w.patch_proj.kw && w.patch_norm.C == E && w.patch_norm.g && w.patch_norm.b;
    for (int k = 0; k < nin && pe_fused; ++k)
        pe_fused = patch_embed_ln_eligible(w.patch_proj.Cin, w.patch_proj.N, w.patch_proj.kh, w.patch_proj.stride, ins[k].H, ins[k].W, w.patch_proj.K, E);
    // ... and, while the row is in registers, the first block's norm1 of it (the bf16 operand of that block's qkv GEMM)
    float* xn0 = nullptr;
    if (pe_fused && pe_env != 2 && !w.stages[0].blocks.empty() && w.stages[0].C == E && w.stages[0].blocks[0].norm1.C == E && w.stages[0].blocks[0].norm1.g &&
        w.stages[0].blocks[0].norm1.b)
        xn0 = c.act_alloc((size_t)total(0) * E);
    if (pe_fused) {
        size_t off = 0;
        for (int k = 0; k < nin; ++k) {
            if (!c.dry) {
                const double M = (double)rows(k, 0);
                Bracket b(c, FAM_GEMM_GATHER, 2.0 * M * E * w.patch_proj.Kreal, 4.0 * ((double)B * 3 * ins[k].H * ins[k].W + M * E) + (xn0 ? 2.0 * M * E : 0.0), (int)M, E, w.patch_proj.K);
                const LNW* n1 = xn0 ? &w.stages[0].blocks[0].norm1 : nullptr;
                BRN_LAUNCH(launch_patch_embed_ln(ins[k].img, B, ins[k].H, ins[k].W, w.patch_proj.w, w.patch_proj.K, w.patch_proj.bias, w.patch_norm.g,
                                                 w.patch_norm.b, 1e-5f, x + off * E, E, c.stream, n1 ? n1->g : nullptr, n1 ? n1->b : nullptr,
                                                 xn0 ? c.at(xn0, off * E) : nullptr, E, c.bf16 == 2));
            }
            off += rows(k, 0);
        }
    } else {
        const size_t mk = c.arena->mark();
        float* t = c.arena->alloc((size_t)total(0) * E);
        size_t off = 0;
        for (int k = 0; k < nin; ++k) {
            Map tm; tm.p = t + off * E; tm.B = B; tm.H = hs[k][0]; tm.W = wsz[k][0]; tm.C = E; tm.ld = E; tm.coff = 0;
            run_conv_nchw(c, w.patch_proj, ins[k].img, B, ins[k].H, ins[k].W, tm, true);
            off += rows(k, 0);
        }
        run_layernorm(c, w.patch_norm, t, total(0), E, x, E, 0);
        c.arena->release(mk);
    }
    for (int i = 0; i < 4; ++i) {
        const SwinStageW& st = w.stages[i];
        const int C = st.C, M = total(i);
        int hh[2], ww[2];
        for (int k = 0; k < nin; ++k) { hh[k] = hs[k][i]; ww[k] = wsz[k][i]; }
        float* xnext = nullptr;
        if (st.has_down) xnext = c.arena->alloc((size_t)total(i + 1) * 2 * C);
        const size_t mk = c.arena->mark();
        // P layout (kernels/split_planes.h) of the blocks' GEMM inputs in the split modes: 2 planes = the fp32 row size, 3 planes = 1.5x
        const int hidden = st.blocks.empty() ? 4 * C : st.blocks[0].fc1.N;
        int stage_pl = 0;
        if (!st.blocks.empty()) {
            const SwinBlockW& b0 = st.blocks[0];
            const int np = b0.qkv.planes;
            // (3 planes = rows 1.5x as long: measured 2 % SLOWER per forward in f32_split3 with the warp-specialised kernel, and the
            // LDS-DMA plane kernel, kernels/gemm_planes.hip, which needs P3 input, did not beat it at batch 1: 2 planes only by default)
#ifdef BRN_DIAG_BUILD
            static const bool planes_on = getenv("BRN_PLANES_KERNEL") && atoi(getenv("BRN_PLANES_KERNEL")) != 0;
#else
            constexpr bool planes_on = false;
#endif
            if (w.window == 12 && (np == 2 || (np == 3 && planes_on)) && b0.qkv.wp && b0.proj.wp && b0.fc1.wp && b0.fc2.wp && C % 32 == 0 && hidden % 32 == 0) stage_pl = np;
        }
        const int ldx = stage_pl ? C * stage_pl / 2 : C, ldh = stage_pl ? hidden * stage_pl / 2 : hidden;
        c.h2_scale = (stage_pl == 2 && st.blocks[0].qkv.half) ? half2_act_scale() : 0.f;    // mode f32_half2: the P2 planes are fp16 planes of the scaled activations
        // compute mode BRN_BF16: x (the residual stream) stays fp32; every GEMM operand (xn, qkv, att, hid, pm) is bf16
        const int yb = c.bf16;
        const bool xn_ready = i == 0 && xn0 && !stage_pl && ldx == C;     // block 0's norm1 came out of the PatchEmbed kernel
        float* xn = xn_ready ? xn0 : c.act_alloc((size_t)M * ldx);
        float* hid = c.act_alloc((size_t)M * ldh);
        for (size_t j = 0; j < st.blocks.size(); ++j) {
            const SwinBlockW& bk = st.blocks[j];
            const int shift = (j % 2 == 0) ? 0 : w.window / 2;                       // swin.rs:552
            // split modes: every GEMM input of the block is written by its producer in the P layout (the bf16 planes the GEMM
            // would split out while staging), so the GEMMs' staging waves only copy
            const int p2 = stage_pl;
            if (!(xn_ready && j == 0)) run_layernorm(c, bk.norm1, x, M, C, xn, ldx, 0, p2, yb);   // swin.rs:355
            const bool ln2_done = swin_attention_multi(c, bk, xn, B, nin, hh, ww, C, shift, x, x, p2, w.window, &bk.norm2, xn, ldx);   // x = shortcut + attn (swin.rs:406)
            if (!ln2_done) run_layernorm(c, bk.norm2, x, M, C, xn, ldx, 0, p2, yb);   // swin.rs:407
            run_gemm(c, bk.fc1, xn, M, ldx, hid, ldh, 0, nullptr, 0, 0, nullptr, 0, p2, p2);   // fc1 + gelu_erf (swin.rs:104-105)
            run_gemm(c, bk.fc2, hid, M, ldh, x, C, 0, x, C, 0, nullptr, 0, p2, 0, yb, yb);     // x + fc2(...) (swin.rs:106,407)
        }
        // stage output = norm_i(x_out), pre-downsample (swin.rs:591,784-789); written into its consumer's window
        size_t off = 0, off2 = 0;
        float* pm = nullptr;
        const int pm_pl = (st.has_down && st.reduction.wp && (st.reduction.planes == 2 || (st.reduction.planes == 3 && stage_pl == 3)) && (4 * C) % 32 == 0) ? st.reduction.planes : 0;
        const int ldpm = pm_pl ? 4 * C * pm_pl / 2 : 4 * C;
        if (st.has_down) pm = c.act_alloc((size_t)total(i + 1) * ldpm);
        for (int k = 0; k < nin; ++k) {
            const Map& o = ins[k].outs[i];
            if (o.B != B || o.H != hh[k] || o.W != ww[k] || o.C != C) fail(BRN_ERR_INVALID_ARG, "swin output window %d has the wrong shape", i);
            run_layernorm(c, st.out_norm, x + off * C, rows(k, i), C, o.p, o.ld, o.coff, 0, outs_f32 ? 0 : yb);
            if (st.has_down) {
                // PatchMerging (swin.rs:491-527): gather 2x2 + LN(4C) fused, then the bias-free reduction (below, once)
                const int M2 = rows(k, i + 1);
                if (!c.dry) {
                    LayerNormParams p{};
                    p.x = x + off * C; p.y = c.at(pm, off2 * ldpm); p.rows = M2; p.y_bf16 = yb; p.C = 4 * C; p.gamma = st.down_norm.g; p.beta = st.down_norm.b;
                    p.eps = 1e-5f; p.ldy = ldpm; p.y_coff = 0; p.mode = 1; p.H = hh[k]; p.W = ww[k]; p.Cin = C;
                    p.y_planes = pm_pl;                                            // P layout for the reduction GEMM
                    p.y_h2 = (pm_pl == 2 && st.reduction.half) ? half2_act_scale() : 0.f;
                    Bracket b(c, FAM_LAYERNORM, 0.0, 8.0 * M2 * 4.0 * C, M2, 4 * C, 1);
                    BRN_LAUNCH(launch_layernorm(p, c.stream));
                }
                off2 += M2;
            }
            off += rows(k, i);
        }
        if (st.has_down) run_gemm(c, st.reduction, pm, total(i + 1), ldpm, xnext, 2 * C, 0, nullptr, 0, 0, nullptr, 0, pm_pl, 0, yb, 0);
        c.h2_scale = 0.f;
        c.arena->release(mk);
        x = xnext;
    }
    c.arena->release(mk0);
}

void swin_forward(Ctx& c, const SwinW& w, const float* img, int B, int H, int W, const Map outs[4]) {
    SwinIn in{img, H, W, outs};
    swin_forward_multi(c, w, &in, 1, B);
}

// ---- BasicDecBlk (decoder.rs:126-141) with ASPPDeformable (aspp.rs:303-333) ------------------------------------------------
void decblk_forward(Ctx& c, const DecBlkW& w, const Map& in, const Map& out, int deform_mode, int out_f32) {
    const size_t mk = c.arena->mark();
    // the maps between the convs carry w.icp channels: inter_channels rounded up to the channel granule (64 -> 64 in the model); a conv
    // writes its real output channels, so the pad channels of a fresh map are zeroed once
    auto inter_map = [&](int chans, int padded) {
        Map m_ = new_map(c, in.B, in.H, in.W, padded);
        if (padded != chans && !c.dry) BRN_HIP(hipMemsetAsync(m_.p, 0, m_.pixels() * (size_t)padded * c.esz(), c.stream));
        Map v = m_.window(0, chans);
        return std::make_pair(m_, v);
    };
    auto [t, t_out] = inter_map(w.ic, w.icp);
    run_conv(c, w.conv_in, in, t_out);                               // conv_in + bn_in + relu
    if (!w.has_aspp) {                                               // dec_att is None (decoder.rs:131-135)
        run_conv(c, w.conv_out, t, out, nullptr, 0, 0, out_f32);
        c.arena->release(mk);
        return;
    }
    auto [u, u_out] = inter_map(w.aspp.oc, w.icp);                   // (ASPPDeformable(inter, None): out_channels = inter_channels)
    aspp_forward(c, w.aspp, t, u_out, deform_mode);
    run_conv(c, w.conv_out, u, out, nullptr, 0, 0, out_f32);         // conv_out + bn_out (no ReLU)
    c.arena->release(mk);
}

void aspp_forward(Ctx& c, const ASPPW& a, const Map& t, const Map& u, int deform_mode) {
    if (t.C != a.icp || t.ld != a.icp || t.coff || u.C != a.oc || u.B != t.B || u.H != t.H || u.W != t.W)
        fail(BRN_ERR_INVALID_ARG, "ASPPDeformable(%d -> %d): input map [C %d, ld %d, coff %d] must be a whole map of %d channels, output map C %d", a.ic, a.oc, t.C,
             t.ld, t.coff, a.icp, u.C);
    const size_t mk = c.arena->mark();
    const int B = t.B, H = t.H, W = t.W, M = B * H * W, IC = a.icp, OC = a.oc;
    const int region0 = c.region;
    c.region = REGION_ASPP;
    Map cat = new_map(c, B, H, W, 1024);                             // [aspp1 | deform k1 | k3 | k7]; pooled branch -> bias
    float* g0 = c.arena->alloc((size_t)B * IC);
    float* g1 = c.arena->alloc((size_t)B * 256);
    float* gb = c.arena->alloc((size_t)B * OC);
    float* gscr = c.arena->alloc(gap_scratch_floats(B, H * W, IC));
    {
        // the branches only share their input t: each runs on its own stream (the 7 x 7 branch, the longest, stays on the main one)
        ArenaHold hold(*c.arena);
        {
            // pooled branch: mean over H then W (aspp.rs:314), 1x1 conv (no bias) + BN + ReLU, nearest-broadcast (aspp.rs:315-318)
            Branch br(c, 2);
            if (!c.dry) {
                Bracket b(c, FAM_ELEMENTWISE, 0.0, 4.0 * M * IC);
                BRN_LAUNCH(launch_gap_nhwc(t.p, B, H * W, IC, IC, 0, gscr, g0, c.stream, c.bf16));
                BRN_LAUNCH(launch_small_fc(g0, B, IC, a.gap_w, IC, 0, 256, a.gap_scale, a.gap_shift, ACT_RELU, g1, c.stream));
                BRN_LAUNCH(launch_small_fc(g1, B, 256, a.conv1_full, 1280, 1024, OC, nullptr, nullptr, ACT_NONE, gb, c.stream));
            }
        }
        if (deform_mode == BRN_DEFORM_REFERENCE_CPU) {
            { Branch br(c, 0); run_gemm(c, a.k1pair, t.p, M, IC, cat.p, 1024, 0); }          // aspp1 + aspp_deforms.0 (regular 1x1, BN, ReLU)
            { Branch br(c, 1); run_conv(c, a.d[2].regular, t, cat.window(512, 256)); }       // k3
            run_conv(c, a.d[3].regular, t, cat.window(768, 256));                             // k7
        } else {
            for (int i = 0; i < 4; ++i) {
                Branch br(c, i < 3 ? i : -1);
                const DeformW& d = a.d[i];
                const int kk = d.k * d.k, ldom = d.offmod.N;         // 3 k^2 rounded up to 8 (zero filters: build_aspp_weights)
                Map om; om.B = B; om.H = H; om.W = W; om.C = ldom; om.ld = ldom; om.coff = 0;
                om.p = c.arena->alloc((size_t)M * ldom);             // offsets / modulator stay fp32 in every mode
                run_conv(c, d.offmod, t, om, nullptr, 0, 0, 1);      // offset_conv | modulator_conv (aspp.rs:171,173)
                const bool fused_sig = deform_fused_sigmoid(c, d.regular);   // bf16 gather kernel: 2*sigmoid applied where the modulator is read
                if (!c.dry && !fused_sig) {
                    Bracket b(c, FAM_ELEMENTWISE, 0.0, 8.0 * M * kk);
                    BRN_LAUNCH(launch_mod_sigmoid2(om.p, (size_t)M, ldom, 2 * kk, 3 * kk, c.stream));   // 2*sigmoid (aspp.rs:174)
                }
                run_conv(c, d.regular, t, cat.window(256 * i, 256), om.p, ldom, 2 * kk, 0, fused_sig ? 1 : 0);
            }
        }
        join_branches(c, AUX_ASPP_MASK);
    }
    run_gemm(c, a.conv1_main, cat.p, M, 1024, u.p, u.ld, u.coff, nullptr, 0, 0, gb, H * W);   // conv1 + bn1 + relu (aspp.rs:329-331)
    c.region = region0;
    c.arena->release(mk);
}

// ---- decoder (birefnet.rs:278-376) ---------------------------------------------------------------------------------------
static void ipt_block(Ctx& c, const SimpleConvsW& w, const float* img, int B, int H, int W, int th, int tw, int cin,
                      const Map& out) {
    const size_t mk = c.arena->mark();
    const int cinp = roundup(cin, 32);
    Map pt = new_map(c, B, th, tw, cinp);
    if (!c.dry) {
        Bracket b(c, FAM_ELEMENTWISE, 0.0, 8.0 * B * 3.0 * H * W);
        BRN_LAUNCH(launch_image2patches(img, B, 3, H, W, th, tw, pt.p, cinp, cinp, c.stream, c.bf16));   // birefnet.rs:288-300
    }
    Map mid = new_map(c, B, th, tw, 64);
    run_conv(c, w.conv1, pt, mid);        // no activation between the two convs (decoder.rs:52)
    run_conv(c, w.conv_out, mid, out);
    c.arena->release(mk);
}

static void gdt_gate(Ctx& c, const DecoderW& d, int i, const Map& p) {
    const size_t mk = c.arena->mark();
    Map g = new_map(c, p.B, p.H, p.W, 16);
    run_conv(c, d.gdt[i], p, g);                                      // conv3x3 -> 16, BN, ReLU (birefnet.rs:111-117)
    if (!c.dry) {
        Bracket b(c, FAM_ELEMENTWISE, 0.0, 8.0 * p.pixels() * p.C);
        BRN_LAUNCH(launch_gdt_gate(p.p, (int)p.pixels(), p.C, p.ld, p.coff, g.p, 16, d.gdt_attn_w[i], d.gdt_attn_b[i], c.stream, c.bf16));
    }
    c.arena->release(mk);
}

static DecMaps alloc_dec_maps(Ctx& c, const Model& m, int B, int H, int W) {
    const DecoderW& d = m.dec;
    DecMaps dm;
    dm.d3 = new_map(c, B, H / 16, W / 16, 1920);
    dm.d2 = new_map(c, B, H / 8, W / 8, 960);
    // (bf16-storage mode: 512 channels, the last 32 zeros written by ipt_blk2's padded conv_out: decoder_block1.conv_in runs chunk-major)
    const int d1pad = d.dec[3].conv_in.Cinp > 480 ? d.dec[3].conv_in.Cinp - 480 : 0;
    if (d.ipt[1].conv_out.N != 96 + d1pad) fail(BRN_ERR_INVALID_ARG, "ipt_blk2 / decoder_block1 channel padding mismatch");
    dm.d1 = new_map(c, B, H / 4, W / 4, 480 + d1pad);
    dm.d1.C = 480;
    return dm;
}
// ipt_blk5 .. ipt_blk2 (birefnet.rs:304-305,335-337,350-352,365-366): they read only the image and write the last channels of
// the concat maps, so they can run any time before the decoder block that reads the map
static void ipt_blocks(Ctx& c, const Model& m, const float* img, int B, int H, int W, const Map& d4, const DecMaps& dm) {
    const DecoderW& d = m.dec;
    ipt_block(c, d.ipt[4], img, B, H, W, H / 32, W / 32, 3072, d4.window(3072, 384));
    ipt_block(c, d.ipt[3], img, B, H, W, H / 16, W / 16, 768, dm.d3.window(1536, 384));   // ipt4_up is a same-size resize = identity
    ipt_block(c, d.ipt[2], img, B, H, W, H / 8, W / 8, 192, dm.d2.window(768, 192));
    ipt_block(c, d.ipt[1], img, B, H, W, H / 4, W / 4, 48, dm.d1.window(384, dm.d1.ld - 384));
}

// lateral_block4 / 3 / 2 (1x1 convs of the backbone maps, birefnet.rs:333,348,363) written into [0:C) of the concat maps BEFORE the
// up-sampled decoder map is added there (run_resize accumulates): they depend on the backbone only, so they can overlap the
// squeeze module and decoder_block4, whose launches fill a fraction of the chip.  fp32 maps only: (conv + bias) + resized and
// resized + (conv + bias) are the same fp32 sum, while on bf16 maps the stored conv result would be rounded once more.
static void lateral_blocks(Ctx& c, const Model& m, int B, int H, int W, const Map& x1, const Map& x2, const Map& x3, const DecMaps& dm) {
    const DecoderW& d = m.dec;
    run_gemm(c, d.lat[0], c.at(x3.p, x3.coff), B * (H / 16) * (W / 16), x3.ld, dm.d3.p, dm.d3.ld, 0);
    run_gemm(c, d.lat[1], c.at(x2.p, x2.coff), B * (H / 8) * (W / 8), x2.ld, dm.d2.p, dm.d2.ld, 0);
    run_gemm(c, d.lat[2], c.at(x1.p, x1.coff), B * (H / 4) * (W / 4), x1.ld, dm.d1.p, dm.d1.ld, 0);
}

void decoder_forward(Ctx& c, const Model& m, const float* img, int B, int H, int W, const Map& x1, const Map& x2, const Map& x3,
                     const Map& d4, float* out, int apply_sigmoid, const DecMaps* pre) {
    const DecoderW& d = m.dec;
    const int dm = m.cfg.deform_mode;
    const int h4 = H / 32, w4 = W / 32, h3 = H / 16, w3 = W / 16, h2 = H / 8, w2 = W / 8, h1 = H / 4, w1 = W / 4;
    const size_t mk = c.arena->mark();
    const DecMaps maps = pre ? *pre : alloc_dec_maps(c, m, B, H, W);
    if (!pre) ipt_blocks(c, m, img, B, H, W, d4, maps);
    else join_branches(c, 1u << AUX_IPT);
    const bool lat_done = pre && pre->lat_done;
    const Map &d3 = maps.d3, &d2 = maps.d2, &d1 = maps.d1;
    // stage 4: cat(x4, ipt5) -> decoder_block4 -> gate (birefnet.rs:304-305, 323-329)
    Map p4 = new_map(c, B, h4, w4, 1536);
    decblk_forward(c, d.dec[0], d4, p4, dm);
    gdt_gate(c, d, 0, p4);
    // stage 3 (birefnet.rs:332-344)
    if (lat_done) join_branches(c, 1u << AUX_LAT);
    run_resize(c, p4, d3.window(0, 1536), lat_done);
    if (!lat_done) run_gemm(c, d.lat[0], c.at(x3.p, x3.coff), B * h3 * w3, x3.ld, d3.p, d3.ld, 0, d3.p, d3.ld, 0);   // + lateral_block4(x3)
    Map p3 = new_map(c, B, h3, w3, 768);
    decblk_forward(c, d.dec[1], d3, p3, dm);
    gdt_gate(c, d, 1, p3);
    // stage 2 (birefnet.rs:347-359)
    run_resize(c, p3, d2.window(0, 768), lat_done);
    if (!lat_done) run_gemm(c, d.lat[1], c.at(x2.p, x2.coff), B * h2 * w2, x2.ld, d2.p, d2.ld, 0, d2.p, d2.ld, 0);
    Map p2 = new_map(c, B, h2, w2, 384);
    decblk_forward(c, d.dec[2], d2, p2, dm);
    gdt_gate(c, d, 2, p2);
    // stage 1 (birefnet.rs:362-369)
    run_resize(c, p2, d1.window(0, 384), lat_done);
    if (!lat_done) run_gemm(c, d.lat[2], c.at(x1.p, x1.coff), B * h1 * w1, x1.ld, d1.p, d1.ld, 0, d1.p, d1.ld, 0);
    // p1 is the last map of the chain and feeds a 192-term dot product per pixel (the head): in compute mode BRN_BF16 it is kept fp32
    // (BRN_P1_F32=0: bf16 like every other map) — its rounding is the one error of the decoder that nothing downstream averages
    static const bool p1_f32_env = !(getenv("BRN_P1_F32") && atoi(getenv("BRN_P1_F32")) == 0);
    const bool p1_f32 = c.bf16 && p1_f32_env;
    Map p1;
    if (p1_f32) { p1.B = B; p1.H = h1; p1.W = w1; p1.C = 192; p1.ld = 192; p1.coff = 0; p1.p = c.arena->alloc((size_t)B * h1 * w1 * 192); }
    else p1 = new_map(c, B, h1, w1, 192);
    decblk_forward(c, d.dec[3], d1, p1, dm, p1_f32 ? 1 : 0);
    // head (birefnet.rs:372-375): q = <p1, w[0:192]> at 1/4 res; t = the whole ipt_blk1 branch (conv1 -> conv_out -> its
    // slice of conv_out1) as one composed 5x5 stencil on the image (brn_weights.cpp): no 64-channel 1024^2 map exists
    float* q = c.arena->alloc((size_t)B * h1 * w1);
    float* tl = c.arena->alloc((size_t)B * H * W);
    if (!c.dry) {
        Bracket b(c, FAM_ELEMENTWISE, 2.0 * B * H * (double)W * 75, 4.0 * B * H * (double)W * 5);
        BRN_LAUNCH(launch_pixel_dot(p1.p, B * h1 * w1, 192, p1.ld, p1.coff, d.out_w, 0.f, q, c.stream, p1_f32 ? 0 : c.bf16));
        BRN_LAUNCH(launch_head_stencil5x5(img, B, H, W, d.head_k, d.head_b, tl, c.stream));
        BRN_LAUNCH(launch_final_head(q, B, h1, w1, tl, d.out_b, H, W, apply_sigmoid, out, c.stream));
    }
    c.arena->release(mk);
}

// ---- BiRefNet::forward_logits (birefnet.rs:412-461) ----------------------------------------------------------------------------
void model_forward(Model& m, Ctx& c, const float* img, int B, int H, int W, float* out, int apply_sigmoid) {
    if (H % 32 || W % 32 || H < 32 || W < 32)
        fail(BRN_ERR_INVALID_ARG, "input %dx%d: H and W must be positive multiples of 32 (image2patches, birefnet.rs:288-300)", H, W);
    const bool prof = c.profile && !c.dry && m.stage_ev_ok;
    auto stamp = [&](int i) { if (prof) BRN_HIP(hipEventRecord(m.stage_ev[i], c.stream)); };
    // Ctx::bf16 says what the maps being allocated / the kernels being launched hold: the backbone's setting (m.bf16) inside
    // swin_forward_multi, the decoder side's (m.dec_bf16) everywhere else — the two differ only in the mixed mode BRN_BF16_DEC_SPLIT2
    struct Bf16Scope { Ctx& c; int old; Bf16Scope(Ctx& c_, int v) : c(c_), old(c_.bf16) { c.bf16 = v; } ~Bf16Scope() { c.bf16 = old; } };
    Bf16Scope dec_scope(c, m.dec_bf16);
    const size_t mk = c.arena->mark();
    const int h1 = H / 4, w1 = W / 4, h2 = H / 8, w2 = W / 8, h3 = H / 16, w3 = W / 16, h4 = H / 32, w4 = W / 32;
    // multi-scale concat targets (birefnet.rs:440-443) and the context concat (birefnet.rs:453): [x1|x2|x3|x4] at 1/32
    Map X1 = new_map(c, B, h1, w1, 384), X2 = new_map(c, B, h2, w2, 768), X3 = new_map(c, B, h3, w3, 1536);
    Map X4 = new_map(c, B, h4, w4, 5760);
    Map D4 = new_map(c, B, h4, w4, 3456);
    const DecMaps dmaps = alloc_dec_maps(c, m, B, H, W);
    stamp(0);
    {
        // the decoder's image-patch convolutions depend on nothing but the image: enqueued first, on an auxiliary stream, they fill
        // the CUs the batch-1 backbone leaves idle (their temporaries stay allocated: the branch is joined in decoder_forward)
        ArenaHold hold(*c.arena);
        Branch br(c, AUX_IPT);
        ipt_blocks(c, m, img, B, H, W, D4, dmaps);
    }
    {
        // both backbone passes (birefnet.rs:416 and :426) as one pass over concatenated token rows
        const size_t mk2 = c.arena->mark();
        const int Hh = H / 2, Wh = W / 2;
        float* half = c.arena->alloc((size_t)B * 3 * Hh * Wh);
        if (!c.dry) {
            Bracket b(c, FAM_RESIZE, 0.0, 4.0 * B * 3.0 * (H * (double)W + Hh * (double)Wh));
            BRN_LAUNCH(launch_resize_nchw(img, B * 3, H, W, half, Hh, Wh, c.stream));  // birefnet.rs:425
        }
        int hs[4], ws[4];
        swin_stage_dims(Hh, Wh, m.swin.patch, hs, ws);
        Map hm[4];
        for (int i = 0; i < 4; ++i) hm[i] = new_map(c, B, hs[i], ws[i], 192 << i);
        Map outs[4] = {X1.window(0, 192), X2.window(0, 384), X3.window(0, 768), X4.window(2688, 1536)};
        SwinIn ins[2] = {{img, H, W, outs}, {half, Hh, Wh, hm}};
        {
            Bf16Scope bb_scope(c, m.bf16);
            swin_forward_multi(c, m.swin, ins, 2, B, m.bf16 && !m.dec_bf16);
        }
        stamp(1);
        run_resize(c, hm[0], X1.window(192, 192));                                    // birefnet.rs:435-443
        run_resize(c, hm[1], X2.window(384, 384));
        run_resize(c, hm[2], X3.window(768, 768));
        run_resize(c, hm[3], X4.window(4224, 1536));
        c.arena->release(mk2);
        // context: x1, x2, x3 bilinearly DOWN-sampled to 1/32 (no antialias), birefnet.rs:450-453
        run_resize(c, X1, X4.window(0, 384));
        run_resize(c, X2, X4.window(384, 768));
        run_resize(c, X3, X4.window(1152, 1536));
    }
    DecMaps dmaps2 = dmaps;
    if (!c.bf16) {
        ArenaHold hold(*c.arena);
        Branch br(c, AUX_LAT);
        lateral_blocks(c, m, B, H, W, X1, X2, X3, dmaps2);
        dmaps2.lat_done = true;
    }
    stamp(2);
    decblk_forward(c, m.squeeze, X4, D4.window(0, 3072), m.cfg.deform_mode);          // birefnet.rs:457
    stamp(3);
    decoder_forward(c, m, img, B, H, W, X1, X2, X3, D4, out, apply_sigmoid, &dmaps2); // birefnet.rs:460
    stamp(4);
    join_branches(c, ~0u);                 // (every branch is joined where its result is read; nothing may outlive the forward)
    c.arena->release(mk);
}

Model::~Model() {
    if (arena.base) (void)hipFree(arena.base);
    if (io.base) (void)hipFree(io.base);
    for (Side& sd : sides) {
        if (sd.arena.base) (void)hipFree(sd.arena.base);
        if (sd.stream) (void)hipStreamDestroy(sd.stream);
        if (sd.join_ev) (void)hipEventDestroy(sd.join_ev);
    }
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    for (int k = 0; k < 2; ++k) {
        if (cu_stream[k]) (void)hipStreamDestroy(cu_stream[k]);
        if (cu_join_ev[k]) (void)hipEventDestroy(cu_join_ev[k]);
    }
    for (BranchSet& bs : branch_sets)
        for (int i = 0; i < BRN_AUX_STREAMS; ++i) {
            if (bs.stream[i]) (void)hipStreamDestroy(bs.stream[i]);
            if (bs.fork_ev[i]) (void)hipEventDestroy(bs.fork_ev[i]);
            if (bs.join_ev[i]) (void)hipEventDestroy(bs.join_ev[i]);
        }
    for (hipEvent_t e : event_pool) (void)hipEventDestroy(e);
    if (stage_ev_ok) for (int i = 0; i < 6; ++i) (void)hipEventDestroy(stage_ev[i]);
    if (done_ev) (void)hipEventDestroy(done_ev);
}

}  // namespace brn
