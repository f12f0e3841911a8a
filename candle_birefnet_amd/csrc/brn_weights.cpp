// brn_weights.cpp — BiRefNet::new (birefnet.rs:389-409) on the device side: look the tensors up by their
// VarBuilder names, repack them for gemm_f32 (K-contiguous, channels-last tap order, rows padded
// to the tile), fold eval-mode BatchNorm into per-channel scale/shift, precompute the transposed relative-position
// bias per block (the analogue of WindowAttention::cached_bias, swin.rs:147-152), upload once.
#include "brn_host.h"
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <cstdio>

namespace brn {

static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
const char* last_error_cstr() { return g_last_error.c_str(); }

void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void ensure_device(int ordinal) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) fail(BRN_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (ordinal < 0 || ordinal >= n) fail(BRN_ERR_NO_DEVICE, "device ordinal %d out of range (have %d)", ordinal, n);
    BRN_HIP(hipSetDevice(ordinal));
}

// ---- device ownership -----------------------------------------------------------------------------------------
float* DeviceOwner::upload(const float* host, size_t n) {
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(float) + 16);
    if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", n * sizeof(float), hipGetErrorString(e));
    ptrs.push_back(d);
    BRN_HIP(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    return (float*)d;
}
DeviceOwner::~DeviceOwner() {
    for (void* p : ptrs) (void)hipFree(p);
}

// ---- weight table -----------------------------------------------------------------------------------------------
WeightTable::WeightTable(const brn_named_tensor* w, size_t n) {
    for (size_t i = 0; i < n; ++i)
        if (w[i].name) map[w[i].name] = &w[i];
}
const brn_named_tensor* WeightTable::get(const std::string& name, std::initializer_list<int64_t> shape) const {
    auto it = map.find(name);
    if (it == map.end()) fail(BRN_ERR_MISSING_TENSOR, "cannot find tensor %s", name.c_str());
    const brn_named_tensor* t = it->second;
    bool ok = t->ndim == (int)shape.size() && t->data != nullptr;
    if (ok) {
        int i = 0;
        for (int64_t d : shape) ok = ok && t->shape[i++] == d;
    }
    if (!ok) {
        std::string want, got;
        for (int64_t d : shape) want += (want.empty() ? "" : ", ") + std::to_string(d);
        for (int i = 0; i < t->ndim; ++i) got += (got.empty() ? "" : ", ") + std::to_string(t->shape[i]);
        fail(BRN_ERR_SHAPE, "shape mismatch for %s: expected [%s] got [%s]", name.c_str(), want.c_str(), got.c_str());
    }
    return t;
}

static inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

static thread_local int g_build_planes = 0;
static thread_local bool g_build_f16 = false;
void set_build_f16(bool on) { g_build_f16 = on; }
void set_build_planes(int planes) { g_build_planes = planes; }
int build_planes() { return g_build_planes; }

static inline uint16_t bf16_rne(float x) {            // round-to-nearest-even fp32 -> bf16 (finite inputs)
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f16_rne(float x) {             // round-to-nearest-even fp32 -> fp16 (finite inputs; |x| >= 65520 -> Inf)
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x47800000u) return sign | 0x7c00u;
    if (u < 0x38800000u) {                             // below 2^-14: a multiple of 2^-24 — the ulp of fp32 numbers in [0.5, 1)
        float f;
        memcpy(&f, &u, 4);
        f += 0.5f;
        uint32_t v;
        memcpy(&v, &f, 4);
        return sign | (uint16_t)(v - 0x3f000000u);
    }
    uint32_t v = u - 0x38000000u;
    v += 0xfffu + ((v >> 13) & 1u);
    return sign | (uint16_t)(v >> 13);
}
static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    float f;
    if (e == 0) { f = (float)m * 5.9604644775390625e-08f; uint32_t u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
    const uint32_t u = sign | ((e == 31 ? 255u : e + 112u) << 23) | (m << 13);
    memcpy(&f, &u, 4);
    return f;
}
float half2_act_scale() {
    static const float s = [] {
        const char* e = getenv("BRN_H2_ASCALE");
        const int k = e ? atoi(e) : 3;                  // log2 of the scale
        return ldexpf(1.f, k < 0 ? 0 : (k > 8 ? 8 : k));
    }();
    return s;
}
static inline uint16_t s16_rne(float x) { return g_build_f16 ? f16_rne(x) : bf16_rne(x); }   // the 16-bit storage type of the build (BRN_BF16 / BRN_F16)
// error-free split of the packed fp32 matrix [rows][K] into bf16 planes: plane p = RN_bf16(x - sum of the previous planes),
// stored interleaved per 32-deep K tile: [row][K/32][plane][32] (a (row, K tile) is NP x 64 contiguous bytes)
static void attach_planes(DeviceOwner& own, GemmW& g, const std::vector<float>& pk, int rows) {
    const int np = g_build_planes;
    if (np <= 0) return;
    if (np == BUILD_BF16) {      // bf16-storage mode: W = RNE bf16 of the packed matrix, rows padded to 256, K to 64 (gemm_bf16.hip)
        const size_t K = (size_t)g.K, ld = (K + 63) / 64 * 64;
        const size_t nrows = pk.size() / K, prow = (nrows + 255) / 256 * 256;
        std::vector<uint16_t> wb(prow * ld, 0);
        // channels-last convs over more than one 64-channel chunk: K order (chunk, tap, channel in chunk) instead of pk's (tap, channel), so
        // that the kh x kw taps of a chunk — which re-read the same input pixels — are consecutive K steps: the re-reads then hit L2 (tap-major
        // order puts a whole Cin sweep of the tile's neighbourhood, > 4 MB for 32 concurrent tiles of the decoder's conv_in, between them)
        static const bool cm_off = getenv("BRN_CONV_CHUNK_MAJOR") && atoi(getenv("BRN_CONV_CHUNK_MAJOR")) == 0;
        const bool cm = !cm_off && g.mode == GEMM_CONV_NHWC && g.Cinp % 64 == 0 && g.Cinp > 64 && (size_t)g.kh * g.kw * g.Cinp == K;
        const size_t kk = (size_t)g.kh * g.kw, cinp = (size_t)g.Cinp;
        for (size_t r = 0; r < nrows; ++r)
            for (size_t k = 0; k < K; ++k) {
                size_t kd = k;
                if (cm) { const size_t t = k / cinp, ci = k - t * cinp; kd = ((ci >> 6) * kk + t) * 64 + (ci & 63); }
                wb[r * ld + kd] = s16_rne(pk[r * K + k]);
            }
        g.wb_chunk_major = cm ? 1 : 0;
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, wb.size() * 2 + 16);
        if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", wb.size() * 2, hipGetErrorString(e));
        own.ptrs.push_back(d);
        BRN_HIP(hipMemcpy(d, wb.data(), wb.size() * 2, hipMemcpyHostToDevice));
        g.wb = d; g.wb_rows = (int)prow; g.wb_ld = (int)ld;
        (void)rows;
        return;
    }
    const size_t n = pk.size();
    const size_t K = (size_t)g.K;
    if (np == BUILD_HALF2) {
        // mode f32_half2: hi = RN_f16(s w), lo = RN_f16(s w - hi), s = 2^k with max |w| s in (2^13, 2^14] — both planes of every weight that
        // matters are normal fp16 numbers (hi + lo = s w up to 2^-22), and what falls below 2^-14 is 2^-38 of the largest weight
        float mx = 0.f;
        for (size_t i = 0; i < n; ++i) mx = std::max(mx, fabsf(pk[i]));
        int k = 0;
        if (mx > 0.f && std::isfinite(mx)) { int ex; frexpf(mx, &ex); k = 14 - ex; }     // mx = f 2^ex, f in [0.5, 1): mx 2^k in [2^13, 2^14)
        k = std::max(-100, std::min(100, k));
        const float sc = ldexpf(1.f, k);
        std::vector<uint16_t> planes(n * 2);
        for (size_t i = 0; i < n; ++i) {
            const size_t row = i / K, kk = i - row * K;
            const float x = pk[i] * sc;
            const uint16_t h = f16_rne(x), l = f16_rne(x - f16_to_f32(h));
            const size_t o = (row * (K / 32) + kk / 32) * (size_t)64 + (kk & 31);
            planes[o] = h;
            planes[o + 32] = l;
        }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, planes.size() * 2 + 16);
        if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", planes.size() * 2, hipGetErrorString(e));
        own.ptrs.push_back(d);
        BRN_HIP(hipMemcpy(d, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
        g.wp = d; g.planes = 2; g.wp_rows = rows; g.half = 1; g.w_scale = sc;
        return;
    }
    std::vector<uint16_t> planes(n * np);
    for (size_t i = 0; i < n; ++i) {
        const size_t row = i / K, k = i - row * K;
        float r = pk[i];
        for (int p = 0; p < np; ++p) {
            const uint16_t h = bf16_rne(r);
            planes[(row * (K / 32) + k / 32) * (size_t)np * 32 + (size_t)p * 32 + (k & 31)] = h;
            r -= bf16_to_f32(h);
        }
    }
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, planes.size() * 2 + 16);
    if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", planes.size() * 2, hipGetErrorString(e));
    own.ptrs.push_back(d);
    BRN_HIP(hipMemcpy(d, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
    g.wp = d; g.planes = np; g.wp_rows = rows;
}

// ---- repack helpers ------------------------------------------------------------------------------------------------
GemmW make_linear(DeviceOwner& own, const float* w, const float* bias, int N, int K) {
    if (K % 32) fail(BRN_ERR_INVALID_ARG, "linear in_features %d must be a multiple of 32", K);
    GemmW g;
    g.N = N; g.K = K; g.Kreal = K; g.Cin = K; g.Cinp = K; g.mode = GEMM_DENSE;
    std::vector<float> pk((size_t)roundup(N, 128) * K, 0.f);
    memcpy(pk.data(), w, (size_t)N * K * sizeof(float));
    g.w = own.upload(pk);
    attach_planes(own, g, pk, roundup(N, 128));
    attach_dense_frags(own, g, w);
    if (bias) g.bias = own.upload(bias, N);
    return g;
}

GemmW make_conv_nhwc(DeviceOwner& own, const float* w, const float* bias, int O, int Cin, int cinp, int kh, int kw,
                     int stride, int pad, int dil) {
    if (cinp % 32 || cinp < Cin) fail(BRN_ERR_INVALID_ARG, "padded channel count %d invalid for Cin %d", cinp, Cin);
    GemmW g;
    g.N = O; g.K = kh * kw * cinp; g.Kreal = kh * kw * Cin; g.Cin = Cin; g.Cinp = cinp;
    g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil = dil;
    g.mode = (kh == 1 && kw == 1 && stride == 1 && pad == 0) ? GEMM_DENSE : GEMM_CONV_NHWC;
    std::vector<float> pk((size_t)roundup(O, 128) * g.K, 0.f);
    for (int o = 0; o < O; ++o)
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx)
                    pk[(size_t)o * g.K + (size_t)(ky * kw + kx) * cinp + ci] = w[(((size_t)o * Cin + ci) * kh + ky) * kw + kx];
    g.w = own.upload(pk);
    attach_planes(own, g, pk, roundup(O, 128));
    if (g.mode == GEMM_DENSE && cinp == Cin) attach_dense_frags(own, g, w);      // 1x1 conv = Linear: [O][Cin] as stored
    if (bias) g.bias = own.upload(bias, O);
    return g;
}

// bf16-storage mode, deformable convs: W[n][k] (k = (ky, kx, ci), ci padded to Cinp) as bf16 in the order the MFMA consumes it —
// fragment (n16 block nb, K step kt of 64, k32 half s) is 1 KiB: lane l = 16 (k / 8 % 4) + n % 16 holds 8 consecutive k — so a
// wave's fragment load in kernels/deform_bf16.hip is one contiguous read.  Rows padded to 256 with zeros.
void attach_deform_frags(DeviceOwner& own, GemmW& g, const float* w) {
    if (g_build_planes != BUILD_BF16 || g.Cinp % 64 || g.K != g.kh * g.kw * g.Cinp) return;
    const int nk = g.K / 64, nb_n = roundup(g.N, 256) / 16, kk = g.kh * g.kw;
    std::vector<uint16_t> wf((size_t)nb_n * nk * 2 * 64 * 8, 0);
    for (int n = 0; n < g.N; ++n)
        for (int t = 0; t < kk; ++t)
            for (int ci = 0; ci < g.Cin; ++ci) {
                const int k = t * g.Cinp + ci;
                const int kt = k >> 6, s = (k >> 5) & 1, lane = ((k >> 3) & 3) * 16 + (n & 15), e = k & 7;
                wf[((((size_t)(n >> 4) * nk + kt) * 2 + s) * 64 + lane) * 8 + e] = s16_rne(w[((size_t)n * g.Cin + ci) * kk + t]);
            }
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, wf.size() * 2 + 16);
    if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", wf.size() * 2, hipGetErrorString(e));
    own.ptrs.push_back(d);
    BRN_HIP(hipMemcpy(d, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
    g.wf = d;
}

void attach_dense_frags(DeviceOwner& own, GemmW& g, const float* w) {
    if (g_build_planes != BUILD_BF16 || g.mode != GEMM_DENSE || (g.K != 192 && g.K != 384) || g.N < 192 || g.N % 192) return;
    const int k32 = g.K / 32, nb_n = g.N / 16;
    std::vector<uint16_t> wf((size_t)nb_n * k32 * 64 * 8, 0);
    for (int n = 0; n < g.N; ++n)
        for (int k = 0; k < g.K; ++k) {
            const int lane = ((k >> 3) & 3) * 16 + (n & 15);
            wf[(((size_t)(n >> 4) * k32 + (k >> 5)) * 64 + lane) * 8 + (k & 7)] = s16_rne(w[(size_t)n * g.K + k]);
        }
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, wf.size() * 2 + 16);
    if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", wf.size() * 2, hipGetErrorString(e));
    own.ptrs.push_back(d);
    BRN_HIP(hipMemcpy(d, wf.data(), wf.size() * 2, hipMemcpyHostToDevice));
    g.wf = d;
}

GemmW make_conv_gather(DeviceOwner& own, const float* w, const float* bias, int O, int Cin, int kh, int kw, int stride,
                       int pad, int dil) {
    GemmW g;
    g.N = O; g.Kreal = Cin * kh * kw; g.K = roundup(g.Kreal, 32); g.Cin = Cin; g.Cinp = Cin;
    g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil = dil; g.mode = GEMM_GATHER_NCHW;
    std::vector<float> pk((size_t)roundup(O, 128) * g.K, 0.f);
    for (int o = 0; o < O; ++o) memcpy(&pk[(size_t)o * g.K], &w[(size_t)o * g.Kreal], (size_t)g.Kreal * sizeof(float));
    g.w = own.upload(pk);
    if (bias) g.bias = own.upload(bias, O);
    return g;
}

// eval-mode batch_norm after a conv (decoder.rs:128-129): y = ((acc + b) - mean) / sqrt(var + eps) * gamma + beta
//                                                        = acc * scale + shift
void fold_bn(DeviceOwner& own, GemmW& g, const float* conv_bias, const float* gamma, const float* beta,
             const float* mean, const float* var, float eps) {
    std::vector<float> sc(g.N), sh(g.N);
    for (int n = 0; n < g.N; ++n) {
        const double s = (double)gamma[n] / std::sqrt((double)var[n] + (double)eps);
        const double b = conv_bias ? (double)conv_bias[n] : 0.0;
        sc[n] = (float)s;
        sh[n] = (float)((b - (double)mean[n]) * s + (double)beta[n]);
    }
    g.scale = own.upload(sc);
    g.shift = own.upload(sh);
    g.bias = nullptr;
}

struct BNHost { const float *g, *b, *m, *v; };
static BNHost get_bn(const WeightTable& wt, const std::string& p, int C) {
    BNHost h;
    h.g = wt.get(p + ".weight", {C})->data;
    h.b = wt.get(p + ".bias", {C})->data;
    h.m = wt.get(p + ".running_mean", {C})->data;
    h.v = wt.get(p + ".running_var", {C})->data;
    return h;
}
static LNW get_ln(const WeightTable& wt, const std::string& p, int C, DeviceOwner& own) {
    LNW l;
    l.C = C;
    l.g = own.upload(wt.get(p + ".weight", {C})->data, C);
    l.b = own.upload(wt.get(p + ".bias", {C})->data, C);
    return l;
}
static GemmW get_linear(const WeightTable& wt, const std::string& p, int N, int K, bool bias, DeviceOwner& own) {
    const float* w = wt.get(p + ".weight", {N, K})->data;
    const float* b = bias ? wt.get(p + ".bias", {N})->data : nullptr;
    return make_linear(own, w, b, N, K);
}

// ---- Swin (swin.rs:725-764) -----------------------------------------------------------------------------------------
void build_swin_weights(const WeightTable& wt, const std::string& pre, const brn_config& cfg, DeviceOwner& own, SwinW& out) {
    if (cfg.window_size != 12 && cfg.window_size != 7)
        fail(BRN_ERR_INVALID_ARG, "window_size %d unsupported: the attention kernels are built for 12 (Swin-B / L, swin.rs:60,74) and 7 (Swin-T / S, swin.rs:32,46)", cfg.window_size);
    if (cfg.patch_size < 1 || cfg.in_channels < 1) fail(BRN_ERR_INVALID_ARG, "bad patch_size/in_channels");
    out.embed_dim = cfg.embed_dim; out.window = cfg.window_size; out.patch = cfg.patch_size; out.in_ch = cfg.in_channels;
    const int E = cfg.embed_dim, P = cfg.patch_size, IC = cfg.in_channels;
    if (E % 32) fail(BRN_ERR_INVALID_ARG, "embed_dim %d must be a multiple of 32", E);
    {
        const float* w = wt.get(pre + "patch_embed.proj.weight", {E, IC, P, P})->data;
        const float* b = wt.get(pre + "patch_embed.proj.bias", {E})->data;
        out.patch_proj = make_conv_gather(own, w, b, E, IC, P, P, P, 0, 1);
        out.patch_norm = get_ln(wt, pre + "patch_embed.norm", E, own);
    }
    const int ws = cfg.window_size, T = (2 * ws - 1) * (2 * ws - 1);
    for (int i = 0; i < 4; ++i) {
        SwinStageW& st = out.stages[i];
        const int C = E << i, heads = cfg.num_heads[i];
        st.C = C; st.heads = heads;
        if (heads <= 0 || C != heads * 32)
            fail(BRN_ERR_INVALID_ARG, "stage %d: dim %d / heads %d: head_dim must be 32 (Swin-L geometry)", i, C, heads);
        const int hidden = (int)((double)C * (double)cfg.mlp_ratio);   // swin.rs:337
        const std::string lp = pre + "layers." + std::to_string(i) + ".";
        st.blocks.resize(cfg.depths[i]);
        for (int j = 0; j < cfg.depths[i]; ++j) {
            SwinBlockW& bk = st.blocks[j];
            const std::string bp = lp + "blocks." + std::to_string(j) + ".";
            bk.heads = heads;
            bk.norm1 = get_ln(wt, bp + "norm1", C, own);
            bk.norm2 = get_ln(wt, bp + "norm2", C, own);
            bk.qkv = get_linear(wt, bp + "attn.qkv", 3 * C, C, true, own);
            bk.proj = get_linear(wt, bp + "attn.proj", C, C, true, own);
            bk.fc1 = get_linear(wt, bp + "mlp.fc1", hidden, C, true, own);
            bk.fc1.act = ACT_GELU_ERF;                                  // swin.rs:105
            bk.fc2 = get_linear(wt, bp + "mlp.fc2", C, hidden, true, own);
            const float* table = wt.get(bp + "attn.relative_position_bias_table", {T, heads})->data;
            {   // [T][heads] -> [heads][T] so a workgroup reads its head's column contiguously; indexed in-kernel by
                // (qi-ki+ws-1)*(2ws-1) + (qj-kj+ws-1) (swin.rs:182-184)
                std::vector<float> tt((size_t)T * heads);
                for (int t = 0; t < T; ++t) for (int h = 0; h < heads; ++h) tt[(size_t)h * T + t] = table[(size_t)t * heads + h];
                bk.rel_table = own.upload(tt);
            }
        }
        st.has_down = i < 3;
        if (st.has_down) {
            st.down_norm = get_ln(wt, lp + "downsample.norm", 4 * C, own);
            st.reduction = get_linear(wt, lp + "downsample.reduction", 2 * C, 4 * C, false, own);
        }
        st.out_norm = get_ln(wt, pre + "norm" + std::to_string(i), C, own);
    }
}

// ---- BasicDecBlk + ASPPDeformable (decoder.rs:86-124, aspp.rs:236-300) ---------------------------------------------------
static GemmW conv_bn(const WeightTable& wt, const std::string& conv, bool has_bias, const std::string& bn, int O, int Cin,
                     int cinp, int k, int pad, int act, DeviceOwner& own) {
    const float* w = wt.get(conv + ".weight", {O, Cin, k, k})->data;
    const float* b = has_bias ? wt.get(conv + ".bias", {O})->data : nullptr;
    GemmW g = make_conv_nhwc(own, w, nullptr, O, Cin, cinp, k, k, 1, pad, 1);
    if (!bn.empty()) {
        BNHost h = get_bn(wt, bn, O);
        fold_bn(own, g, b, h.g, h.b, h.m, h.v, 1e-5f);
    } else if (b) {
        g.bias = own.upload(b, O);
    }
    g.act = act;
    return g;
}

void build_decblk_weights(const WeightTable& wt, const std::string& p, int cin, int cout, int deform_mode, DeviceOwner& own,
                          DecBlkW& out, bool use_aspp, int inter_channels) {
    const int IC = inter_channels;  // decoder.rs:94-98
    if (IC < 1) fail(BRN_ERR_INVALID_ARG, "decoder block inter_channels %d", IC);
    const int gran = g_build_planes == BUILD_BF16 ? 64 : 32;
    out.cin = cin; out.cout = cout; out.has_aspp = use_aspp; out.ic = IC; out.icp = roundup(IC, gran);
    // The channels-last convs read whole 32-channel granules (64 in the bf16-storage mode, where e.g. decoder_block1's 480 is padded so
    // that the conv runs chunk-major): a channel count off the granule is padded with zero weights to the next multiple; the map then
    // carries that many channels, the pad ones zero (conv_in's pad OUTPUT channels have zero weights, scale and shift).  Every block of
    // the model has in_channels % 32 == 0 and inter_channels = 64.
    const int cin_pad = roundup(cin, gran);
    out.conv_in = conv_bn(wt, p + "conv_in", true, p + "bn_in", IC, cin, cin_pad, 3, 1, ACT_RELU, own);
    out.conv_out = conv_bn(wt, p + "conv_out", true, p + "bn_out", cout, IC, out.icp, 3, 1, ACT_NONE, own);   // no ReLU (decoder.rs:138-139)
    if (use_aspp) build_aspp_weights(wt, p + "dec_att.", deform_mode, own, out.aspp, IC, 0);                    // decoder.rs:107-111
}

// ASPPDeformable::new (aspp.rs:236-300) under prefix `ap`: in_channels IC (64 inside the model's BasicDecBlk), out_channels OC
void build_aspp_weights(const WeightTable& wt, const std::string& ap, int deform_mode, DeviceOwner& own, ASPPW& a, int in_channels, int out_channels) {
    const int IC = in_channels, OC = out_channels > 0 ? out_channels : in_channels, PL = 256;   // aspp.rs:242-243
    if (IC < 1) fail(BRN_ERR_INVALID_ARG, "ASPPDeformable in_channels %d", IC);
    const int ICP = roundup(IC, g_build_planes == BUILD_BF16 ? 64 : 32);
    a.ic = IC; a.icp = ICP; a.oc = OC;
    const int ks[4] = {1, 1, 3, 7};
    const std::string mods[4] = {ap + "aspp1.", ap + "aspp_deforms.0.", ap + "aspp_deforms.1.", ap + "aspp_deforms.2."};
    for (int i = 0; i < 4; ++i) {
        const int k = ks[i], kk = k * k;
        DeformW& d = a.d[i];
        d.k = k;
        const std::string cp = mods[i] + "atrous_conv.";
        // all three convs must exist, as DeformConvASPP::new loads them (aspp.rs:39-45)
        const float* ow = wt.get(cp + "offset_conv.weight", {2 * kk, IC, k, k})->data;
        const float* ob = wt.get(cp + "offset_conv.bias", {2 * kk})->data;
        const float* mw = wt.get(cp + "modulator_conv.weight", {kk, IC, k, k})->data;
        const float* mb = wt.get(cp + "modulator_conv.bias", {kk})->data;
        d.regular = conv_bn(wt, cp + "regular_conv", false, mods[i] + "bn", PL, IC, ICP, k, k / 2, ACT_RELU, own);
        if (deform_mode == BRN_DEFORM_DEFORMABLE) {
            d.regular.mode = GEMM_DEFORM_NHWC;
            attach_deform_frags(own, d.regular, wt.get(cp + "regular_conv.weight", {PL, IC, k, k})->data);
            // offset_conv and modulator_conv stacked on N = 3 k^2, padded with zero filters to a multiple of 8 (3 / 27 / 147 -> 8 / 32 / 152): the
            // fp32 offset map then leaves the bf16 GEMM through its vector-store epilogue instead of the per-element one
            static const bool pad8 = !(getenv("BRN_OFFMOD_PAD8") && atoi(getenv("BRN_OFFMOD_PAD8")) == 0);   // (A/B switch: 0 = pad to 4 as before)
            const int n3p = roundup(3 * kk, pad8 ? 8 : 4);
            std::vector<float> w3((size_t)n3p * IC * kk, 0.f), b3((size_t)n3p, 0.f);
            memcpy(w3.data(), ow, (size_t)2 * kk * IC * kk * sizeof(float));
            memcpy(w3.data() + (size_t)2 * kk * IC * kk, mw, (size_t)kk * IC * kk * sizeof(float));
            memcpy(b3.data(), ob, (size_t)2 * kk * sizeof(float));
            memcpy(b3.data() + 2 * kk, mb, (size_t)kk * sizeof(float));
            d.offmod = make_conv_nhwc(own, w3.data(), b3.data(), n3p, IC, ICP, k, k, 1, k / 2, 1);
        }
    }
    if (deform_mode == BRN_DEFORM_REFERENCE_CPU) {
        // aspp1 and aspp_deforms.0 are both plain 1x1 IC->256 convs of the same input on the CPU path
        // (aspp.rs:183-185): one GEMM with N = 512 writes both concat slices (K = ICP: zero columns for the pad channels).
        const float* w0 = wt.get(mods[0] + "atrous_conv.regular_conv.weight", {PL, IC, 1, 1})->data;
        const float* w1 = wt.get(mods[1] + "atrous_conv.regular_conv.weight", {PL, IC, 1, 1})->data;
        std::vector<float> w2((size_t)2 * PL * ICP, 0.f);
        for (int n = 0; n < PL; ++n) {
            memcpy(&w2[(size_t)n * ICP], &w0[(size_t)n * IC], (size_t)IC * sizeof(float));
            memcpy(&w2[(size_t)(PL + n) * ICP], &w1[(size_t)n * IC], (size_t)IC * sizeof(float));
        }
        a.k1pair = make_linear(own, w2.data(), nullptr, 2 * PL, ICP);
        BNHost h0 = get_bn(wt, mods[0] + "bn", PL), h1 = get_bn(wt, mods[1] + "bn", PL);
        std::vector<float> g(2 * PL), b(2 * PL), m(2 * PL), v(2 * PL);
        for (int n = 0; n < PL; ++n) {
            g[n] = h0.g[n]; b[n] = h0.b[n]; m[n] = h0.m[n]; v[n] = h0.v[n];
            g[PL + n] = h1.g[n]; b[PL + n] = h1.b[n]; m[PL + n] = h1.m[n]; v[PL + n] = h1.v[n];
        }
        fold_bn(own, a.k1pair, nullptr, g.data(), b.data(), m.data(), v.data(), 1e-5f);
        a.k1pair.act = ACT_RELU;
    }
    // global_avg_pool.1 (conv, no bias) + .2 (BN) (aspp.rs:271-278); rows padded to ICP like the pooled vector
    {
        const float* gw = wt.get(ap + "global_avg_pool.1.weight", {PL, IC, 1, 1})->data;
        std::vector<float> gp((size_t)PL * ICP, 0.f);
        for (int n = 0; n < PL; ++n) memcpy(&gp[(size_t)n * ICP], &gw[(size_t)n * IC], (size_t)IC * sizeof(float));
        a.gap_w = own.upload(gp);
        BNHost h = get_bn(wt, ap + "global_avg_pool.2", PL);
        std::vector<float> sc(PL), sh(PL);
        for (int n = 0; n < PL; ++n) {
            const double s = (double)h.g[n] / std::sqrt((double)h.v[n] + 1e-5);
            sc[n] = (float)s; sh[n] = (float)((0.0 - (double)h.m[n]) * s + (double)h.b[n]);
        }
        a.gap_scale = own.upload(sc); a.gap_shift = own.upload(sh);
    }
    // conv1 1x1 1280->OC no bias + bn1 + ReLU (aspp.rs:282-290, 329-331).  The first 1024 input channels are the four
    // spatial branches (a GEMM); the last 256 are the pooled branch, constant over the map -> a per-image bias.
    {
        const float* cw = wt.get(ap + "conv1.weight", {OC, 5 * PL, 1, 1})->data;
        a.conv1_full = own.upload(cw, (size_t)OC * 5 * PL);
        std::vector<float> mainw((size_t)OC * 4 * PL);
        for (int o = 0; o < OC; ++o) memcpy(&mainw[(size_t)o * 4 * PL], &cw[(size_t)o * 5 * PL], (size_t)4 * PL * sizeof(float));
        a.conv1_main = make_linear(own, mainw.data(), nullptr, OC, 4 * PL);
        BNHost h = get_bn(wt, ap + "bn1", OC);
        fold_bn(own, a.conv1_main, nullptr, h.g, h.b, h.m, h.v, 1e-5f);
        a.conv1_main.act = ACT_RELU;
    }
}

// ---- BiRefNetDecoder::new (birefnet.rs:170-273) -------------------------------------------------------------------------
void build_decoder_weights(const WeightTable& wt, const std::string& p, const brn_config& cfg, DeviceOwner& own, DecoderW& out) {
    int lat[4];
    brn_config_lateral_channels(&cfg, lat);                      // [384,768,1536,3072]
    const int ipt_out[5] = {48, 96, 192, 384, 384};              // birefnet.rs:180
    const int ipt_in[5] = {3, ipt_out[0], lat[0] / 2, lat[2] / 2, lat[3]};   // birefnet.rs:189-193
    // what image2patches really delivers at each scale (birefnet.rs:304-316): 3*g*g channels
    const int patch_ch[5] = {3, 48, 192, 768, 3072};
    for (int i = 0; i < 5; ++i)
        if (ipt_in[i] != patch_ch[i])
            fail(BRN_ERR_INVALID_ARG, "ipt_blk%d expects %d channels but image2patches yields %d: only the Swin-L channel plan "
                 "[192,384,768,1536] with mul_scl_ipt is self-consistent in the reference (birefnet.rs:189-193,304-316)",
                 i + 1, ipt_in[i], patch_ch[i]);
    // ipt_blk1 (conv1 3->64, conv_out 64->48, no activation) is composed with conv_out1's ipt slice into one stencil below
    for (int i = 1; i < 5; ++i) {
        const std::string ip = p + "ipt_blk" + std::to_string(i + 1) + ".";
        const int cin = ipt_in[i], cinp = roundup(cin, 32);
        const float* w = wt.get(ip + "conv1.weight", {64, cin, 3, 3})->data;
        const float* b = wt.get(ip + "conv1.bias", {64})->data;
        out.ipt[i].conv1 = make_conv_nhwc(own, w, b, 64, cin, cinp, 3, 3, 1, 1, 1);
        const float* w2 = wt.get(ip + "conv_out.weight", {ipt_out[i], 64, 3, 3})->data;
        const float* b2 = wt.get(ip + "conv_out.bias", {ipt_out[i]})->data;
        // bf16-storage mode: the consumer of ipt_blk2 (decoder_block1.conv_in, 384 + 96 = 480 input channels) reads a map padded to 512
        // channels so that its K loop can run chunk-major (attach_planes); the 32 pad channels must hold zeros, and the cheapest writer
        // is this conv with 32 extra all-zero output channels (weights and bias zero: exact zeros, no activation follows)
        const int opad = (g_build_planes == BUILD_BF16 && ipt_out[i] == 96) ? 128 : ipt_out[i];
        if (opad != ipt_out[i]) {
            std::vector<float> wz((size_t)opad * 64 * 9, 0.f), bz(opad, 0.f);
            memcpy(wz.data(), w2, (size_t)ipt_out[i] * 64 * 9 * sizeof(float));
            memcpy(bz.data(), b2, (size_t)ipt_out[i] * sizeof(float));
            out.ipt[i].conv_out = make_conv_nhwc(own, wz.data(), bz.data(), opad, 64, 64, 3, 3, 1, 1, 1);
        } else
        out.ipt[i].conv_out = make_conv_nhwc(own, w2, b2, ipt_out[i], 64, 64, 3, 3, 1, 1, 1);
    }
    const int dec_out[4] = {lat[2], lat[1], lat[0], lat[0] / 2};          // [1536,768,384,192] birefnet.rs:202
    const int dec_in[4] = {lat[3] + ipt_out[4], dec_out[0] + ipt_out[3], dec_out[1] + ipt_out[2], dec_out[2] + ipt_out[1]};
    const char* dnames[4] = {"decoder_block4.", "decoder_block3.", "decoder_block2.", "decoder_block1."};
    for (int i = 0; i < 4; ++i)
        build_decblk_weights(wt, p + dnames[i], dec_in[i], dec_out[i], cfg.deform_mode, own, out.dec[i]);
    const char* lnames[3] = {"lateral_block4.conv", "lateral_block3.conv", "lateral_block2.conv"};
    const int lch[3] = {lat[2], lat[1], lat[0]};
    for (int i = 0; i < 3; ++i) {
        const float* w = wt.get(p + lnames[i] + ".weight", {lch[i], lch[i], 1, 1})->data;
        const float* b = wt.get(p + lnames[i] + ".bias", {lch[i]})->data;
        out.lat[i] = make_linear(own, w, b, lch[i], lch[i]);
    }
    const char* sfx[3] = {"4", "3", "2"};
    for (int i = 0; i < 3; ++i) {
        const std::string gp = p + "gdt_convs_" + sfx[i];
        out.gdt[i] = conv_bn(wt, gp + ".0", true, gp + ".1", 16, dec_out[i], dec_out[i], 3, 1, ACT_RELU, own);
        const std::string apn = p + "gdt_convs_attn_" + sfx[i] + ".0";
        out.gdt_attn_w[i] = own.upload(wt.get(apn + ".weight", {1, 16, 1, 1})->data, 16);
        out.gdt_attn_b[i] = wt.get(apn + ".bias", {1})->data[0];
        // loaded-but-unused heads must exist (birefnet.rs:230-232, 241-243)
        const std::string ppn = p + "gdt_convs_pred_" + sfx[i] + ".0";
        (void)wt.get(ppn + ".weight", {1, 16, 1, 1}); (void)wt.get(ppn + ".bias", {1});
        const std::string msn = p + "conv_ms_spvn_" + sfx[i];
        (void)wt.get(msn + ".weight", {1, dec_out[i], 1, 1}); (void)wt.get(msn + ".bias", {1});
    }
    // conv_out1.0: 1x1, 240 -> 1 (birefnet.rs:237-238).  w[0:192] acts on p1 (at 1/4 resolution, see final_head_kernel);
    // w[192:240] acts on ipt1 = ipt_blk1.conv_out(u) + b: composed here into one 3x3 64->1 stencil (fp64 accumulate).
    {
        const int fin = dec_out[3] + ipt_out[0];
        const float* ow = wt.get(p + "conv_out1.0.weight", {1, fin, 1, 1})->data;
        out.out_b = wt.get(p + "conv_out1.0.bias", {1})->data[0];
        out.out_w = own.upload(ow, fin);
        const float* w2 = wt.get(p + "ipt_blk1.conv_out.weight", {ipt_out[0], 64, 3, 3})->data;
        const float* b2 = wt.get(p + "ipt_blk1.conv_out.bias", {ipt_out[0]})->data;
        std::vector<float> tw(9 * 64);
        for (int t = 0; t < 9; ++t)
            for (int ci = 0; ci < 64; ++ci) {
                double s = 0.0;
                for (int o = 0; o < ipt_out[0]; ++o) s += (double)ow[dec_out[3] + o] * (double)w2[((size_t)o * 64 + ci) * 9 + t];
                tw[t * 64 + ci] = (float)s;
            }
        double tb = 0.0;
        for (int o = 0; o < ipt_out[0]; ++o) tb += (double)ow[dec_out[3] + o] * (double)b2[o];
        // One level further: ipt_blk1.conv1 (3x3, 3 -> 64, pad 1, bias b1) feeds that stencil with nothing in between
        // (SimpleConvs has no activation, decoder.rs:52), so t = stencil3x3(conv1(x)) is ONE 5x5, 3 -> 1 stencil on the image:
        //   t(p) = tb + sum_{d in D(p)} sum_ci tw[d][ci] * (b1[ci] + sum_e sum_c W1[ci][c][e] * x[c][p+d+e])
        // where D(p) = the taps d of the outer 3x3 whose centre p+d lies inside the image (the outer conv zero-pads conv1's
        // OUTPUT, it does not extend it): 3 x 3 border cases (first / inner / last row, same for columns), each with its own
        // composed kernel K[f = d+e][c] and bias.  x itself is zero-padded, as conv1 does.  fp64 accumulation on the host.
        const float* w1 = wt.get(p + "ipt_blk1.conv1.weight", {64, 3, 3, 3})->data;
        const float* b1 = wt.get(p + "ipt_blk1.conv1.bias", {64})->data;
        std::vector<float> hk(9 * 75), hb(9);
        for (int cy = 0; cy < 3; ++cy)
            for (int cx = 0; cx < 3; ++cx) {
                std::vector<double> K(75, 0.0);
                double bias = tb;
                for (int dy = -1; dy <= 1; ++dy) {
                    if ((cy == 0 && dy < 0) || (cy == 2 && dy > 0)) continue;
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((cx == 0 && dx < 0) || (cx == 2 && dx > 0)) continue;
                        const int t = (dy + 1) * 3 + (dx + 1);
                        for (int ci = 0; ci < 64; ++ci) {
                            const double a = tw[t * 64 + ci];
                            bias += a * (double)b1[ci];
                            for (int c = 0; c < 3; ++c)
                                for (int ey = -1; ey <= 1; ++ey)
                                    for (int ex = -1; ex <= 1; ++ex)
                                        K[((dy + ey + 2) * 5 + (dx + ex + 2)) * 3 + c] +=
                                            a * (double)w1[(((size_t)ci * 3 + c) * 3 + (ey + 1)) * 3 + (ex + 1)];
                        }
                    }
                }
                for (int i = 0; i < 75; ++i) hk[(cy * 3 + cx) * 75 + i] = (float)K[i];
                hb[cy * 3 + cx] = (float)bias;
            }
        out.head_k = own.upload(hk);
        out.head_b = own.upload(hb);
    }
}

}  // namespace brn
