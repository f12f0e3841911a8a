// brn_oracle.cpp — CPU ORACLE for the BiRefNet forward_logits hot path.  TEST INFRASTRUCTURE ONLY.
//
//   * This is a plain C++ restatement of the reference's algorithm (imperatormk/candle-birefnet, /root/reference/src),
//     op by op and UNFUSED, in the reference's own NCHW / [B,L,C] layouts: every pad / roll / partition / cat / permute
//     the Rust code performs is materialised here too.  Each function cites the reference file:line it follows.
//   * PARITY UNPINNED: the reference ships no golden vectors, no known-answer tests and cannot be built in this
//     environment (Rust, un-vendored candle fork).  The arithmetic of candle-core/candle-nn 0.9.2 @
//     imperatormk/candle 674fa161 (LayerNorm, BatchNorm, conv2d, softmax_last_dim, gelu_erf, upsample_bilinear2d) is
//     restated from its published semantics = PyTorch's (biased variance, eps inside the sqrt; cross-correlation with
//     zero padding; max-subtracted softmax; exact-erf GELU; align_corners bilinear src = dst*(in-1)/(out-1)).
//     What pins this file instead: an independent torch restatement (tests/torch_ref.py) and the fixtures under
//     tests/golden/ generated from it in fp64 (tests/golden/make_golden.py).
//   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
//     (candle_birefnet_amd/, libbirefnet_hip.so) never links, imports or calls it.
//
// fp32 storage and fp32 accumulation (as candle's `gemm` crate); OpenMP over all host cores (as candle's rayon pool).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <unordered_map>
#include <stdexcept>
#include <algorithm>
#include <immintrin.h>
#include <omp.h>
#include "../include/birefnet_hip.h"   // brn_config / brn_named_tensor struct definitions only

namespace orc {

// ------------------------------------------------------------------------------------------------------------------
// tensor
// ------------------------------------------------------------------------------------------------------------------
struct Tensor {
    std::vector<int64_t> s;
    std::vector<float> d;
    Tensor() {}
    explicit Tensor(std::vector<int64_t> shape) : s(std::move(shape)) { d.assign((size_t)numel(), 0.f); }
    int64_t numel() const { int64_t n = 1; for (auto v : s) n *= v; return n; }
    int64_t dim(int i) const { return s[i < 0 ? (int)s.size() + i : i]; }
    float* p() { return d.data(); }
    const float* p() const { return d.data(); }
    Tensor reshaped(std::vector<int64_t> ns) const {
        Tensor t; t.s = std::move(ns); t.d = d;
        if (t.numel() != numel()) throw std::runtime_error("reshape: element count mismatch");
        return t;
    }
};

[[noreturn]] static void die(const std::string& m) { throw std::runtime_error(m); }

struct Weights {
    std::unordered_map<std::string, const brn_named_tensor*> m;
    Weights(const brn_named_tensor* w, size_t n) { for (size_t i = 0; i < n; ++i) m[w[i].name] = &w[i]; }
    const float* get(const std::string& name, std::initializer_list<int64_t> shape) const {   // VarBuilder::get
        auto it = m.find(name);
        if (it == m.end()) die("cannot find tensor " + name);
        const brn_named_tensor* t = it->second;
        bool ok = t->ndim == (int)shape.size();
        int i = 0;
        if (ok) for (auto v : shape) ok = ok && t->shape[i++] == v;
        if (!ok) die("shape mismatch for " + name);
        return t->data;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// SGEMM  C[M,N] = A[M,K] * B[N,K]^T   (fp32 accumulate; AVX2+FMA 6x16 outer-product micro-kernel, B packed per K block)
// ------------------------------------------------------------------------------------------------------------------
static void sgemm_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                     int64_t ldc) {
    const int64_t KC = 384, NR = 16, MR = 6;
    const int64_t npan = (N + NR - 1) / NR;
    std::vector<float> Bp((size_t)npan * NR * KC);
    for (int64_t k0 = 0; k0 < K; k0 += KC) {
        const int64_t kc = std::min(KC, K - k0);
#pragma omp parallel for schedule(static)
        for (int64_t pn = 0; pn < npan; ++pn) {
            float* dst = &Bp[(size_t)pn * NR * KC];
            for (int64_t k = 0; k < kc; ++k)
                for (int64_t j = 0; j < NR; ++j) {
                    const int64_t n = pn * NR + j;
                    dst[k * NR + j] = n < N ? B[n * ldb + k0 + k] : 0.f;
                }
        }
        const int64_t mblocks = (M + MR - 1) / MR;
#pragma omp parallel for schedule(dynamic, 8)
        for (int64_t mb = 0; mb < mblocks; ++mb) {
            const int64_t m0 = mb * MR, mr = std::min(MR, M - m0);
            for (int64_t pn = 0; pn < npan; ++pn) {
                const float* bp = &Bp[(size_t)pn * NR * KC];
                __m256 acc[MR][2];
                for (int i = 0; i < MR; ++i) { acc[i][0] = _mm256_setzero_ps(); acc[i][1] = _mm256_setzero_ps(); }
                const float* a0 = A + m0 * lda + k0;
                if (mr == MR) {
                    for (int64_t k = 0; k < kc; ++k) {
                        const __m256 b0 = _mm256_loadu_ps(bp + k * NR), b1 = _mm256_loadu_ps(bp + k * NR + 8);
                        for (int i = 0; i < MR; ++i) {
                            const __m256 a = _mm256_broadcast_ss(a0 + i * lda + k);
                            acc[i][0] = _mm256_fmadd_ps(a, b0, acc[i][0]);
                            acc[i][1] = _mm256_fmadd_ps(a, b1, acc[i][1]);
                        }
                    }
                } else {
                    for (int64_t k = 0; k < kc; ++k) {
                        const __m256 b0 = _mm256_loadu_ps(bp + k * NR), b1 = _mm256_loadu_ps(bp + k * NR + 8);
                        for (int i = 0; i < mr; ++i) {
                            const __m256 a = _mm256_broadcast_ss(a0 + i * lda + k);
                            acc[i][0] = _mm256_fmadd_ps(a, b0, acc[i][0]);
                            acc[i][1] = _mm256_fmadd_ps(a, b1, acc[i][1]);
                        }
                    }
                }
                const int64_t n0 = pn * NR, nr = std::min(NR, N - n0);
                for (int i = 0; i < mr; ++i) {
                    float tmp[16];
                    _mm256_storeu_ps(tmp, acc[i][0]);
                    _mm256_storeu_ps(tmp + 8, acc[i][1]);
                    float* c = C + (m0 + i) * ldc + n0;
                    if (k0 == 0) for (int64_t j = 0; j < nr; ++j) c[j] = tmp[j];
                    else for (int64_t j = 0; j < nr; ++j) c[j] += tmp[j];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// candle ops (semantics: see header)
// ------------------------------------------------------------------------------------------------------------------
// candle_nn::Linear::forward: x[..., K] @ W[N,K]^T + b
static Tensor linear(const Tensor& x, const float* W, const float* b, int64_t N) {
    const int64_t K = x.dim(-1), M = x.numel() / K;
    std::vector<int64_t> os = x.s; os.back() = N;
    Tensor y(os);
    sgemm_nt(M, N, K, x.p(), K, W, K, y.p(), N);
    if (b) {
#pragma omp parallel for
        for (int64_t m = 0; m < M; ++m) { float* r = y.p() + m * N; for (int64_t n = 0; n < N; ++n) r[n] += b[n]; }
    }
    return y;
}

// candle_nn::LayerNorm::forward over the last dim: mean, biased variance, (x-mean)/sqrt(var+eps)*g+b
static Tensor layer_norm(const Tensor& x, const float* g, const float* b, float eps) {
    const int64_t C = x.dim(-1), R = x.numel() / C;
    Tensor y(x.s);
#pragma omp parallel for
    for (int64_t r = 0; r < R; ++r) {
        const float* xr = x.p() + r * C; float* yr = y.p() + r * C;
        float s = 0.f; for (int64_t c = 0; c < C; ++c) s += xr[c];
        const float mean = s / (float)C;
        float q = 0.f; for (int64_t c = 0; c < C; ++c) { const float dd = xr[c] - mean; q += dd * dd; }
        const float rstd = 1.0f / std::sqrt(q / (float)C + eps);
        for (int64_t c = 0; c < C; ++c) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
    }
    return y;
}

// candle_nn::Conv2d::forward, NCHW, zero padding, cross-correlation; im2col + GEMM like candle's CPU backend
static Tensor conv2d(const Tensor& x, const float* W, const float* bias, int64_t O, int64_t kh, int64_t kw, int64_t stride,
                     int64_t pad, int64_t dil) {
    const int64_t B = x.dim(0), C = x.dim(1), H = x.dim(2), Wd = x.dim(3);
    const int64_t Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (Wd + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
    const int64_t CK = C * kh * kw, P = Ho * Wo;
    Tensor y({B, O, Ho, Wo});
    const int64_t chunk = std::max<int64_t>(256, std::min<int64_t>(P, (int64_t)(64 << 20) / std::max<int64_t>(1, CK)));
    std::vector<float> col((size_t)chunk * CK), out((size_t)chunk * O);
    for (int64_t b = 0; b < B; ++b) {
        for (int64_t p0 = 0; p0 < P; p0 += chunk) {
            const int64_t pc = std::min(chunk, P - p0);
#pragma omp parallel for
            for (int64_t pp = 0; pp < pc; ++pp) {
                const int64_t oy = (p0 + pp) / Wo, ox = (p0 + pp) % Wo;
                float* cr = &col[(size_t)pp * CK];
                for (int64_t c = 0; c < C; ++c)
                    for (int64_t ky = 0; ky < kh; ++ky)
                        for (int64_t kx = 0; kx < kw; ++kx) {
                            const int64_t iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
                            cr[(c * kh + ky) * kw + kx] =
                                (iy >= 0 && iy < H && ix >= 0 && ix < Wd) ? x.d[((b * C + c) * H + iy) * Wd + ix] : 0.f;
                        }
            }
            sgemm_nt(pc, O, CK, col.data(), CK, W, CK, out.data(), O);
#pragma omp parallel for
            for (int64_t o = 0; o < O; ++o) {
                float* yr = y.p() + ((b * O + o) * P) + p0;
                const float bb = bias ? bias[o] : 0.f;
                for (int64_t pp = 0; pp < pc; ++pp) yr[pp] = out[(size_t)pp * O + o] + bb;
            }
        }
    }
    return y;
}

// candle_nn::BatchNorm::forward_t(x, train=false): (x - running_mean) / sqrt(running_var + eps) * weight + bias
static void batch_norm_(Tensor& x, const float* g, const float* b, const float* mean, const float* var, float eps) {
    const int64_t B = x.dim(0), C = x.dim(1), P = x.numel() / (B * C);
#pragma omp parallel for
    for (int64_t bc = 0; bc < B * C; ++bc) {
        const int64_t c = bc % C;
        const float inv = 1.0f / std::sqrt(var[c] + eps);
        float* r = x.p() + bc * P;
        for (int64_t i = 0; i < P; ++i) r[i] = (r[i] - mean[c]) * inv * g[c] + b[c];
    }
}
static void relu_(Tensor& x) {
#pragma omp parallel for
    for (int64_t i = 0; i < x.numel(); ++i) x.d[i] = x.d[i] > 0.f ? x.d[i] : 0.f;
}
static void gelu_erf_(Tensor& x) {   // Tensor::gelu_erf (swin.rs:105)
#pragma omp parallel for
    for (int64_t i = 0; i < x.numel(); ++i) { const float v = x.d[i]; x.d[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
}
static void sigmoid_(Tensor& x) {    // candle_nn::ops::sigmoid
#pragma omp parallel for
    for (int64_t i = 0; i < x.numel(); ++i) x.d[i] = 1.0f / (1.0f + std::exp(-x.d[i]));
}
static void add_(Tensor& a, const Tensor& b) {
    if (a.numel() != b.numel()) die("add: size mismatch");
#pragma omp parallel for
    for (int64_t i = 0; i < a.numel(); ++i) a.d[i] += b.d[i];
}

// Tensor::upsample_bilinear2d(h, w, align_corners = true), NCHW
static Tensor upsample_bilinear2d(const Tensor& x, int64_t oh, int64_t ow) {
    const int64_t B = x.dim(0), C = x.dim(1), H = x.dim(2), W = x.dim(3);
    Tensor y({B, C, oh, ow});
    const float sy = oh > 1 ? (float)(H - 1) / (float)(oh - 1) : 0.f, sx = ow > 1 ? (float)(W - 1) / (float)(ow - 1) : 0.f;
#pragma omp parallel for
    for (int64_t bc = 0; bc < B * C; ++bc) {
        const float* src = x.p() + bc * H * W; float* dst = y.p() + bc * oh * ow;
        for (int64_t oy = 0; oy < oh; ++oy) {
            const float fy = sy * (float)oy; int64_t y0 = (int64_t)fy; if (y0 > H - 1) y0 = H - 1;
            const int64_t y1 = y0 + (y0 < H - 1 ? 1 : 0); const float ly = fy - (float)y0;
            for (int64_t ox = 0; ox < ow; ++ox) {
                const float fx = sx * (float)ox; int64_t x0 = (int64_t)fx; if (x0 > W - 1) x0 = W - 1;
                const int64_t x1 = x0 + (x0 < W - 1 ? 1 : 0); const float lx = fx - (float)x0;
                const float v00 = src[y0 * W + x0], v01 = src[y0 * W + x1], v10 = src[y1 * W + x0], v11 = src[y1 * W + x1];
                const float top = v00 + (v01 - v00) * lx, bot = v10 + (v11 - v10) * lx;
                dst[oy * ow + ox] = top + (bot - top) * ly;
            }
        }
    }
    return y;
}

// Tensor::cat(&[...], 1) for NCHW
static Tensor cat_channels(const std::vector<const Tensor*>& ts) {
    const int64_t B = ts[0]->dim(0), H = ts[0]->dim(2), W = ts[0]->dim(3);
    int64_t Ct = 0; for (auto t : ts) Ct += t->dim(1);
    Tensor y({B, Ct, H, W});
    for (auto t : ts)
        if (t->dim(0) != B || t->dim(2) != H || t->dim(3) != W) die("cat: shape mismatch");
    for (int64_t b = 0; b < B; ++b) {
        int64_t c0 = 0;
        for (auto t : ts) {
            const int64_t C = t->dim(1);
#pragma omp parallel for
            for (int64_t c = 0; c < C; ++c)
                memcpy(y.p() + (b * Ct + c0 + c) * H * W, t->p() + (b * C + c) * H * W, (size_t)H * W * sizeof(float));
            c0 += C;
        }
    }
    return y;
}

// ------------------------------------------------------------------------------------------------------------------
// Swin (swin.rs)
// ------------------------------------------------------------------------------------------------------------------
struct SwinCfg { int embed, depths[4], heads[4], ws, patch, in_ch; float mlp_ratio; };

// WindowAttention::build_relative_position_index (swin.rs:166-210)
static std::vector<int64_t> build_relative_position_index(int ws) {
    const int n = ws * ws;
    std::vector<int64_t> rc((size_t)n * n), idx((size_t)n * n);
    for (int i = 0; i < ws; ++i) for (int j = 0; j < ws; ++j) for (int k = 0; k < ws; ++k) for (int l = 0; l < ws; ++l) {
        const int64_t rel_h = i - k + (ws - 1), rel_w = j - l + (ws - 1);
        rc[((size_t)(i * ws + j) * ws + k) * ws + l] = rel_h * (2 * ws - 1) + rel_w;
    }
    for (int i = 0; i < ws; ++i) for (int j = 0; j < ws; ++j) for (int k = 0; k < ws; ++k) for (int l = 0; l < ws; ++l)
        idx[(size_t)(i * ws + j) * n + (k * ws + l)] = rc[((size_t)(i * ws + j) * ws + k) * ws + l];
    return idx;
}

// BasicLayer::create_attention_mask (swin.rs:603-655): [nW, N, N], -100 where region ids differ
static Tensor create_attention_mask(int hp, int wp, int ws, int shift) {
    std::vector<float> img((size_t)hp * wp, 0.f);
    const int hsl[3][2] = {{0, hp - ws}, {hp - ws, hp - shift}, {hp - shift, hp}};
    const int wsl[3][2] = {{0, wp - ws}, {wp - ws, wp - shift}, {wp - shift, wp}};
    int cnt = 0;
    for (auto& hs : hsl) for (auto& wss : wsl) {
        for (int i = hs[0]; i < hs[1]; ++i) for (int j = wss[0]; j < wss[1]; ++j) img[(size_t)i * wp + j] = (float)cnt;
        ++cnt;
    }
    const int nwh = hp / ws, nww = wp / ws, nW = nwh * nww, N = ws * ws;
    std::vector<float> m((size_t)nW * N);
    for (int a = 0; a < nwh; ++a) for (int b = 0; b < nww; ++b) for (int i = 0; i < ws; ++i) for (int j = 0; j < ws; ++j)
        m[(size_t)(a * nww + b) * N + i * ws + j] = img[(size_t)(a * ws + i) * wp + b * ws + j];
    Tensor out({nW, N, N});
    for (int w = 0; w < nW; ++w) for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
        // mask_1 = mask.unsqueeze(1) [nW,1,N], mask_2 = mask.unsqueeze(2) [nW,N,1]; attn_mask = mask_1 - mask_2
        const float dlt = m[(size_t)w * N + j] - m[(size_t)w * N + i];
        out.d[((size_t)w * N + i) * N + j] = dlt != 0.f ? -100.0f : 0.0f;
    }
    return out;
}

// SwinTransformerBlock::roll_2d (swin.rs:412-444) on [B,H,W,C]
static Tensor roll_2d(const Tensor& x, int64_t sh, int64_t sw) {
    const int64_t B = x.dim(0), H = x.dim(1), W = x.dim(2), C = x.dim(3);
    sh = ((sh % H) + H) % H; sw = ((sw % W) + W) % W;
    Tensor y(x.s);
#pragma omp parallel for
    for (int64_t bh = 0; bh < B * H; ++bh) {
        const int64_t b = bh / H, i = bh % H;
        // cat(part1 = x[H-sh:], part2 = x[:H-sh]): out[i] = x[(i - sh) mod H]
        const int64_t si = ((i - sh) % H + H) % H;
        for (int64_t j = 0; j < W; ++j) {
            const int64_t sj = ((j - sw) % W + W) % W;
            memcpy(y.p() + ((b * H + i) * W + j) * C, x.p() + ((b * H + si) * W + sj) * C, (size_t)C * sizeof(float));
        }
    }
    return y;
}

// window_partition (swin.rs:446-459): [B,H,W,C] -> [B*nW, ws*ws, C]
static Tensor window_partition(const Tensor& x, int ws) {
    const int64_t B = x.dim(0), H = x.dim(1), W = x.dim(2), C = x.dim(3), nh = H / ws, nw = W / ws;
    Tensor y({B * nh * nw, (int64_t)ws * ws, C});
#pragma omp parallel for
    for (int64_t t = 0; t < B * nh * nw; ++t) {
        const int64_t b = t / (nh * nw), a = (t / nw) % nh, c = t % nw;
        for (int i = 0; i < ws; ++i) for (int j = 0; j < ws; ++j)
            memcpy(y.p() + (t * ws * ws + i * ws + j) * C, x.p() + ((b * H + a * ws + i) * W + c * ws + j) * C, (size_t)C * sizeof(float));
    }
    return y;
}
// window_reverse (swin.rs:461-475)
static Tensor window_reverse(const Tensor& wnd, int ws, int64_t H, int64_t W) {
    const int64_t C = wnd.dim(2), nh = H / ws, nw = W / ws, B = wnd.dim(0) / (nh * nw);
    Tensor y({B, H, W, C});
#pragma omp parallel for
    for (int64_t t = 0; t < B * nh * nw; ++t) {
        const int64_t b = t / (nh * nw), a = (t / nw) % nh, c = t % nw;
        for (int i = 0; i < ws; ++i) for (int j = 0; j < ws; ++j)
            memcpy(y.p() + ((b * H + a * ws + i) * W + c * ws + j) * C, wnd.p() + (t * ws * ws + i * ws + j) * C, (size_t)C * sizeof(float));
    }
    return y;
}

// WindowAttention::forward + forward_standard (swin.rs:212-312).  x [B_, N, C]; mask [nW,N,N] or null
static Tensor window_attention(const Tensor& x, const Weights& w, const std::string& p, int heads, int ws, const Tensor* mask) {
    const int64_t B_ = x.dim(0), N = x.dim(1), C = x.dim(2), hd = C / heads;
    const float* qw = w.get(p + "qkv.weight", {3 * C, C}); const float* qb = w.get(p + "qkv.bias", {3 * C});
    const float* pw = w.get(p + "proj.weight", {C, C}); const float* pb = w.get(p + "proj.bias", {C});
    const int64_t T = (2 * ws - 1) * (2 * ws - 1);
    const float* table = w.get(p + "relative_position_bias_table", {T, (int64_t)heads});
    const std::vector<int64_t> index = build_relative_position_index(ws);
    // cached_bias [heads, N, N] = table[index].reshape(N,N,h).permute(2,0,1)  (swin.rs:147-152)
    std::vector<float> bias((size_t)heads * N * N);
    for (int64_t i = 0; i < N * N; ++i) for (int h = 0; h < heads; ++h) bias[(size_t)h * N * N + i] = table[index[i] * heads + h];
    Tensor qkv = linear(x, qw, qb, 3 * C);                         // [B_, N, 3C] == [B_, N, 3, heads, hd] (swin.rs:217-218)
    const float scale = (float)std::pow((double)hd, -0.5);        // swin.rs:134 (f64 powf, used as f64 * f32 tensor)
    Tensor o({B_, N, C});
    const int64_t nW = mask ? mask->dim(0) : 1;
#pragma omp parallel
    {
        std::vector<float> q((size_t)N * hd), attn((size_t)N * N);
#pragma omp for schedule(dynamic, 4)
        for (int64_t bh = 0; bh < B_ * heads; ++bh) {
            const int64_t b = bh / heads, h = bh % heads;
            const float* base = qkv.p() + b * N * 3 * C;
            for (int64_t i = 0; i < N; ++i) for (int64_t d = 0; d < hd; ++d) q[i * hd + d] = base[i * 3 * C + h * hd + d] * scale;   // swin.rs:278
            for (int64_t i = 0; i < N; ++i) {
                float* ar = &attn[i * N];
                for (int64_t j = 0; j < N; ++j) {                  // q @ k^T (swin.rs:281)
                    const float* kr = base + j * 3 * C + C + h * hd;
                    float s = 0.f; for (int64_t d = 0; d < hd; ++d) s += q[i * hd + d] * kr[d];
                    ar[j] = s;
                }
                const float* br = &bias[((size_t)h * N + i) * N];
                for (int64_t j = 0; j < N; ++j) ar[j] += br[j];   // broadcast_add(bias) (swin.rs:285)
                if (mask) {                                        // window index = b % nW (swin.rs:291-294)
                    const float* mr = mask->p() + ((b % nW) * N + i) * N;
                    for (int64_t j = 0; j < N; ++j) ar[j] += mr[j];
                }
                float mx = ar[0]; for (int64_t j = 1; j < N; ++j) mx = std::max(mx, ar[j]);   // softmax_last_dim (swin.rs:300)
                float sum = 0.f; for (int64_t j = 0; j < N; ++j) { ar[j] = std::exp(ar[j] - mx); sum += ar[j]; }
                for (int64_t j = 0; j < N; ++j) ar[j] /= sum;
                float* orow = o.p() + (b * N + i) * C + h * hd;   // attn @ v, transpose(1,2).reshape (swin.rs:303-307)
                for (int64_t d = 0; d < hd; ++d) orow[d] = 0.f;
                for (int64_t j = 0; j < N; ++j) {
                    const float a = ar[j]; const float* vr = base + j * 3 * C + 2 * C + h * hd;
                    for (int64_t d = 0; d < hd; ++d) orow[d] += a * vr[d];
                }
            }
        }
    }
    return linear(o, pw, pb, C);                                   // proj (swin.rs:310)
}

// the part of SwinTransformerBlock::forward between norm1 and the residual (swin.rs:356-403): xn [B,H,W,C] -> [B,H,W,C]
static Tensor attention_half(const Tensor& xn, const Weights& w, const std::string& p, int heads, int ws, int shift, const Tensor* mask) {
    const int64_t B = xn.dim(0), H = xn.dim(1), W = xn.dim(2), C = xn.dim(3);
    const int64_t pad_r = (ws - W % ws) % ws, pad_b = (ws - H % ws) % ws, hp = H + pad_b, wp = W + pad_r;
    Tensor x({B, hp, wp, C});                                       // pad_with_zeros (swin.rs:359-366)
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t i = 0; i < H; ++i)
        memcpy(x.p() + ((b * hp + i) * wp) * C, xn.p() + ((b * H + i) * W) * C, (size_t)W * C * sizeof(float));
    if (shift > 0) x = roll_2d(x, -shift, -shift);                  // swin.rs:371-377
    Tensor xw = window_partition(x, ws);                            // swin.rs:380
    Tensor local_mask;
    if (shift > 0 && !mask) { local_mask = create_attention_mask((int)hp, (int)wp, ws, shift); mask = &local_mask; }
    Tensor aw = window_attention(xw, w, p + "attn.", heads, ws, shift > 0 ? mask : nullptr);   // swin.rs:383-384
    Tensor sx = window_reverse(aw, ws, hp, wp);                     // swin.rs:387
    if (shift > 0) sx = roll_2d(sx, shift, shift);                  // swin.rs:390-394
    Tensor y({B, H, W, C});                                         // narrow (swin.rs:397-401)
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t i = 0; i < H; ++i)
        memcpy(y.p() + ((b * H + i) * W) * C, sx.p() + ((b * hp + i) * wp) * C, (size_t)W * C * sizeof(float));
    return y;
}

// SwinTransformerBlock::forward (swin.rs:350-410)
static Tensor swin_block(const Tensor& x, int64_t H, int64_t W, const Weights& w, const std::string& p, int heads, int ws, int shift,
                         const Tensor* mask) {
    const int64_t B = x.dim(0), L = x.dim(1), C = x.dim(2);
    if (L != H * W) die("Input feature has wrong size");           // assert_eq! swin.rs:352
    Tensor xn = layer_norm(x, w.get(p + "norm1.weight", {C}), w.get(p + "norm1.bias", {C}), 1e-5f).reshaped({B, H, W, C});
    Tensor a = attention_half(xn, w, p, heads, ws, shift, mask);
    Tensor x1 = x; add_(x1, a);                                     // shortcut + x (swin.rs:406)
    Tensor h = layer_norm(x1, w.get(p + "norm2.weight", {C}), w.get(p + "norm2.bias", {C}), 1e-5f);
    const float* w1 = nullptr; int64_t hidden = 0;
    {
        auto it = w.m.find(p + "mlp.fc1.weight");
        if (it == w.m.end()) die("cannot find tensor " + p + "mlp.fc1.weight");
        hidden = it->second->shape[0]; w1 = it->second->data;
    }
    h = linear(h, w1, w.get(p + "mlp.fc1.bias", {hidden}), hidden);
    gelu_erf_(h);
    h = linear(h, w.get(p + "mlp.fc2.weight", {C, hidden}), w.get(p + "mlp.fc2.bias", {C}), C);
    add_(x1, h);                                                    // swin.rs:407
    return x1;
}

// PatchMerging::forward (swin.rs:491-527)
static Tensor patch_merging(const Tensor& x, int64_t H, int64_t W, const Weights& w, const std::string& p) {
    const int64_t B = x.dim(0), C = x.dim(2);
    const int64_t H2 = (H + 1) / 2, W2 = (W + 1) / 2;
    Tensor c4({B, H2 * W2, 4 * C});
    auto at = [&](int64_t b, int64_t i, int64_t j) -> const float* { return (i < H && j < W) ? x.p() + ((b * H + i) * W + j) * C : nullptr; };
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t i = 0; i < H2; ++i) for (int64_t j = 0; j < W2; ++j) {
        float* dst = c4.p() + ((b * H2 + i) * W2 + j) * 4 * C;
        const float* src[4] = {at(b, 2 * i, 2 * j), at(b, 2 * i + 1, 2 * j), at(b, 2 * i, 2 * j + 1), at(b, 2 * i + 1, 2 * j + 1)};   // x0,x1,x2,x3 swin.rs:509-516
        for (int k = 0; k < 4; ++k) {
            if (src[k]) memcpy(dst + k * C, src[k], (size_t)C * sizeof(float)); else memset(dst + k * C, 0, (size_t)C * sizeof(float));
        }
    }
    Tensor n = layer_norm(c4, w.get(p + "norm.weight", {4 * C}), w.get(p + "norm.bias", {4 * C}), 1e-5f);
    return linear(n, w.get(p + "reduction.weight", {2 * C, 4 * C}), nullptr, 2 * C);
}

// SwinTransformer::forward (swin.rs:768-797) -> 4 NCHW feature maps
static std::vector<Tensor> swin_forward(const Tensor& x_in, const Weights& w, const std::string& p, const SwinCfg& cfg) {
    const int64_t B = x_in.dim(0), P = cfg.patch, E = cfg.embed;
    Tensor x = x_in;
    {   // PatchEmbed::forward (swin.rs:692-714): pad right/bottom to a multiple of patch, conv, LN over channels
        const int64_t H = x.dim(2), W = x.dim(3);
        if (H % P || W % P) {
            const int64_t ph = (P - H % P) % P, pw = (P - W % P) % P;
            Tensor xp({B, x.dim(1), H + ph, W + pw});
            for (int64_t bc = 0; bc < B * x.dim(1); ++bc) for (int64_t i = 0; i < H; ++i)
                memcpy(xp.p() + (bc * (H + ph) + i) * (W + pw), x.p() + (bc * H + i) * W, (size_t)W * sizeof(float));
            x = xp;
        }
    }
    Tensor pe = conv2d(x, w.get(p + "patch_embed.proj.weight", {E, (int64_t)cfg.in_ch, P, P}), w.get(p + "patch_embed.proj.bias", {E}), E, P, P, P, 0, 1);
    int64_t h = pe.dim(2), wd = pe.dim(3);
    Tensor tok({B, h * wd, E});                                     // flatten(2).transpose(1,2) (swin.rs:708, 774)
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t i = 0; i < h * wd; ++i) for (int64_t c = 0; c < E; ++c)
        tok.d[(b * h * wd + i) * E + c] = pe.d[(b * E + c) * h * wd + i];
    tok = layer_norm(tok, w.get(p + "patch_embed.norm.weight", {E}), w.get(p + "patch_embed.norm.bias", {E}), 1e-5f);
    std::vector<Tensor> outs;
    for (int i = 0; i < 4; ++i) {
        const int64_t C = E << i;
        const std::string lp = p + "layers." + std::to_string(i) + ".";
        const int ws = cfg.ws, shift = ws / 2;
        const int64_t hp = (h + ws - 1) / ws * ws, wp = (wd + ws - 1) / ws * ws;   // swin.rs:580-581
        Tensor mask = create_attention_mask((int)hp, (int)wp, ws, shift);          // swin.rs:584
        for (int j = 0; j < cfg.depths[i]; ++j)
            tok = swin_block(tok, h, wd, w, lp + "blocks." + std::to_string(j) + ".", cfg.heads[i], ws, (j % 2 == 0) ? 0 : shift, &mask);
        Tensor n = layer_norm(tok, w.get(p + "norm" + std::to_string(i) + ".weight", {C}), w.get(p + "norm" + std::to_string(i) + ".bias", {C}), 1e-5f);
        Tensor o({B, C, h, wd});                                    // reshape(B,h,w,C).permute(0,3,1,2) (swin.rs:786-788)
#pragma omp parallel for collapse(2)
        for (int64_t b = 0; b < B; ++b) for (int64_t c = 0; c < C; ++c) for (int64_t t = 0; t < h * wd; ++t)
            o.d[(b * C + c) * h * wd + t] = n.d[(b * h * wd + t) * C + c];
        outs.push_back(std::move(o));
        if (i < 3) {
            tok = patch_merging(tok, h, wd, w, lp + "downsample.");
            h = (h + 1) / 2; wd = (wd + 1) / 2;
        }
    }
    return outs;
}

// ------------------------------------------------------------------------------------------------------------------
// decoder side (aspp.rs, decoder.rs, birefnet.rs)
// ------------------------------------------------------------------------------------------------------------------
static Tensor conv_named(const Tensor& x, const Weights& w, const std::string& p, int64_t O, int64_t k, int64_t pad, bool bias, int64_t stride = 1) {
    const int64_t C = x.dim(1);
    return conv2d(x, w.get(p + ".weight", {O, C, k, k}), bias ? w.get(p + ".bias", {O}) : nullptr, O, k, k, stride, pad, 1);
}
static void bn_named_(Tensor& x, const Weights& w, const std::string& p) {
    const int64_t C = x.dim(1);
    batch_norm_(x, w.get(p + ".weight", {C}), w.get(p + ".bias", {C}), w.get(p + ".running_mean", {C}), w.get(p + ".running_var", {C}), 1e-5f);
}

// torchvision.ops.deform_conv2d semantics (the Metal path's deformable_im2col + matmul, aspp.rs:58-165; 1 offset group,
// modulated): offset channel 2*(i*kw+j) = dy, +1 = dx; bilinear with zero outside (-1, H) x (-1, W)
static Tensor deform_conv2d(const Tensor& x, const Tensor& offset, const Tensor& mask, const float* W, const float* bias, int64_t O,
                            int64_t k, int64_t stride, int64_t pad) {
    const int64_t B = x.dim(0), C = x.dim(1), H = x.dim(2), Wd = x.dim(3), Ho = offset.dim(2), Wo = offset.dim(3);
    const int64_t CK = C * k * k, P = Ho * Wo;
    Tensor y({B, O, Ho, Wo});
    std::vector<float> col((size_t)P * CK), out((size_t)P * O);
    for (int64_t b = 0; b < B; ++b) {
#pragma omp parallel for
        for (int64_t pp = 0; pp < P; ++pp) {
            const int64_t oy = pp / Wo, ox = pp % Wo;
            for (int64_t i = 0; i < k; ++i) for (int64_t j = 0; j < k; ++j) {
                const int64_t t = i * k + j;
                const float dy = offset.d[((b * 2 * k * k + 2 * t) * Ho + oy) * Wo + ox], dx = offset.d[((b * 2 * k * k + 2 * t + 1) * Ho + oy) * Wo + ox];
                const float mk = mask.d[((b * k * k + t) * Ho + oy) * Wo + ox];
                const float py = (float)(oy * stride - pad + i) + dy, px = (float)(ox * stride - pad + j) + dx;
                const bool inside = py > -1.f && py < (float)H && px > -1.f && px < (float)Wd;
                const int64_t yl = (int64_t)std::floor(py), xl = (int64_t)std::floor(px), yh = yl + 1, xh = xl + 1;
                const float ly = py - (float)yl, lx = px - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
                for (int64_t c = 0; c < C; ++c) {
                    float v = 0.f;
                    if (inside) {
                        const float* im = x.p() + (b * C + c) * H * Wd;
                        const float v1 = (yl >= 0 && xl >= 0) ? im[yl * Wd + xl] : 0.f;
                        const float v2 = (yl >= 0 && xh <= Wd - 1) ? im[yl * Wd + xh] : 0.f;
                        const float v3 = (yh <= H - 1 && xl >= 0) ? im[yh * Wd + xl] : 0.f;
                        const float v4 = (yh <= H - 1 && xh <= Wd - 1) ? im[yh * Wd + xh] : 0.f;
                        v = hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
                    }
                    col[(size_t)pp * CK + (c * k + i) * k + j] = v * mk;
                }
            }
        }
        sgemm_nt(P, O, CK, col.data(), CK, W, CK, out.data(), O);
#pragma omp parallel for
        for (int64_t o = 0; o < O; ++o) for (int64_t pp = 0; pp < P; ++pp) y.d[(b * O + o) * P + pp] = out[(size_t)pp * O + o] + (bias ? bias[o] : 0.f);
    }
    return y;
}

// DeformConvASPP::forward (aspp.rs:168-187).  mode 0: CPU path (offset & modulator computed, discarded, regular_conv);
// mode 1: Metal path semantics (forward_metal, aspp.rs:58-165)
static Tensor deform_conv_aspp(const Tensor& x, const Weights& w, const std::string& p, int64_t k, int mode) {
    const int64_t pad = k / 2, C = x.dim(1);
    if (mode == 0) {
        (void)w.get(p + "offset_conv.weight", {2 * k * k, C, k, k}); (void)w.get(p + "modulator_conv.weight", {k * k, C, k, k});
        return conv_named(x, w, p + "regular_conv", 256, k, pad, false);   // aspp.rs:183-185 (their values never reach the output)
    }
    Tensor offset = conv_named(x, w, p + "offset_conv", 2 * k * k, k, pad, true);   // aspp.rs:171
    Tensor mask = conv_named(x, w, p + "modulator_conv", k * k, k, pad, true);     // aspp.rs:173
    for (auto& v : mask.d) v = (1.0f / (std::exp(-v) + 1.0f)) * 2.0f;              // aspp.rs:174
    return deform_conv2d(x, offset, mask, w.get(p + "regular_conv.weight", {256, C, k, k}), nullptr, 256, k, 1, pad);
}

// ASPPDeformable::forward (aspp.rs:303-333)
static Tensor aspp_deformable(const Tensor& x, const Weights& w, const std::string& p, int mode, int64_t out_channels = 0) {
    const int64_t B = x.dim(0), C = x.dim(1), H = x.dim(2), W = x.dim(3);
    auto module = [&](const std::string& q, int64_t k) {              // ASPPModuleDeformable::forward (aspp.rs:217-223)
        Tensor t = deform_conv_aspp(x, w, q + "atrous_conv.", k, mode);
        bn_named_(t, w, q + "bn"); relu_(t); return t;
    };
    Tensor x1 = module(p + "aspp1.", 1);
    Tensor d0 = module(p + "aspp_deforms.0.", 1), d1 = module(p + "aspp_deforms.1.", 3), d2 = module(p + "aspp_deforms.2.", 7);
    Tensor g({B, C, 1, 1});                                           // mean_keepdim(H) then mean_keepdim(W) (aspp.rs:314)
#pragma omp parallel for
    for (int64_t bc = 0; bc < B * C; ++bc) {
        std::vector<float> colmean((size_t)W, 0.f);
        for (int64_t j = 0; j < W; ++j) { float s = 0.f; for (int64_t i = 0; i < H; ++i) s += x.d[(bc * H + i) * W + j]; colmean[j] = s / (float)H; }
        float s = 0.f; for (int64_t j = 0; j < W; ++j) s += colmean[j];
        g.d[bc] = s / (float)W;
    }
    Tensor x5 = conv_named(g, w, p + "global_avg_pool.1", 256, 1, 0, false);
    bn_named_(x5, w, p + "global_avg_pool.2"); relu_(x5);
    Tensor x5u({B, 256, H, W});                                       // upsample_nearest2d (aspp.rs:318)
#pragma omp parallel for
    for (int64_t bc = 0; bc < B * 256; ++bc) std::fill(x5u.p() + bc * H * W, x5u.p() + (bc + 1) * H * W, x5.d[bc]);
    Tensor cat = cat_channels({&x1, &d0, &d1, &d2, &x5u});            // aspp.rs:321-327
    const int64_t OC = out_channels > 0 ? out_channels : C;           // out_channels.unwrap_or(in_channels) (aspp.rs:242)
    Tensor out = conv_named(cat, w, p + "conv1", OC, 1, 0, false);
    bn_named_(out, w, p + "bn1"); relu_(out);
    return out;
}

// BasicDecBlk::forward (decoder.rs:126-141)
static Tensor dec_blk(const Tensor& x, const Weights& w, const std::string& p, int64_t cout, int mode, bool use_aspp = true, int64_t inter = 64) {
    Tensor t = conv_named(x, w, p + "conv_in", inter, 3, 1, true);    // inter_channels: 64, or in_channels / 4 (decoder.rs:94-98)
    bn_named_(t, w, p + "bn_in"); relu_(t);
    if (use_aspp) t = aspp_deformable(t, w, p + "dec_att.", mode);    // dec_att is None when DecoderConfig::use_aspp_deformable is false (decoder.rs:107-111,131-135)
    t = conv_named(t, w, p + "conv_out", cout, 3, 1, true);
    bn_named_(t, w, p + "bn_out");
    return t;
}
// SimpleConvs::forward (decoder.rs:50-56): no activation in between
static Tensor simple_convs(const Tensor& x, const Weights& w, const std::string& p, int64_t cout) {
    Tensor t = conv_named(x, w, p + "conv1", 64, 3, 1, true);
    return conv_named(t, w, p + "conv_out", cout, 3, 1, true);
}
// image2patches (birefnet.rs:288-300)
static Tensor image2patches(const Tensor& x, int64_t th, int64_t tw) {
    const int64_t B = x.dim(0), C = x.dim(1), H = x.dim(2), W = x.dim(3), gh = H / th, gw = W / tw;
    Tensor y({B, C * gh * gw, th, tw});
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t c = 0; c < C; ++c) for (int64_t a = 0; a < gh; ++a) for (int64_t e = 0; e < gw; ++e)
        for (int64_t i = 0; i < th; ++i)
            memcpy(y.p() + (((b * C * gh * gw) + (c * gh + a) * gw + e) * th + i) * tw, x.p() + ((b * C + c) * H + a * th + i) * W + e * tw, (size_t)tw * sizeof(float));
    return y;
}
static void broadcast_mul_(Tensor& p, const Tensor& a) {   // p [B,C,H,W] * a [B,1,H,W] (birefnet.rs:329)
    const int64_t B = p.dim(0), C = p.dim(1), P = p.dim(2) * p.dim(3);
#pragma omp parallel for collapse(2)
    for (int64_t b = 0; b < B; ++b) for (int64_t c = 0; c < C; ++c) for (int64_t i = 0; i < P; ++i) p.d[(b * C + c) * P + i] *= a.d[b * P + i];
}

// BiRefNetDecoder::forward (birefnet.rs:278-376)
static Tensor decoder_forward(const Tensor& x, const Tensor& x1, const Tensor& x2, const Tensor& x3, const Tensor& x4, const Weights& w,
                              const std::string& p, int mode) {
    const int64_t H = x.dim(2), W = x.dim(3);
    const int64_t h3 = x3.dim(2), w3 = x3.dim(3), h2 = x2.dim(2), w2 = x2.dim(3), h1 = x1.dim(2), w1 = x1.dim(3);
    Tensor ipt5 = simple_convs(image2patches(x, H / 32, W / 32), w, p + "ipt_blk5.", 384);
    Tensor ipt4 = simple_convs(image2patches(x, H / 16, W / 16), w, p + "ipt_blk4.", 384);
    Tensor ipt3 = simple_convs(image2patches(x, H / 8, W / 8), w, p + "ipt_blk3.", 192);
    Tensor ipt2 = simple_convs(image2patches(x, H / 4, W / 4), w, p + "ipt_blk2.", 96);
    Tensor ipt1 = simple_convs(x, w, p + "ipt_blk1.", 48);
    auto gate = [&](Tensor& pp, const char* n) {                      // birefnet.rs:327-329
        Tensor g = conv_named(pp, w, p + "gdt_convs_" + n + ".0", 16, 3, 1, true);
        bn_named_(g, w, p + "gdt_convs_" + n + ".1"); relu_(g);
        Tensor a = conv_named(g, w, p + "gdt_convs_attn_" + n + ".0", 1, 1, 0, true);
        sigmoid_(a);
        broadcast_mul_(pp, a);
    };
    Tensor p4 = dec_blk(cat_channels({&x4, &ipt5}), w, p + "decoder_block4.", 1536, mode);
    gate(p4, "4");
    Tensor p3_in = upsample_bilinear2d(p4, h3, w3);
    add_(p3_in, conv_named(x3, w, p + "lateral_block4.conv", 1536, 1, 0, true));
    Tensor ipt4u = upsample_bilinear2d(ipt4, h3, w3);
    Tensor p3 = dec_blk(cat_channels({&p3_in, &ipt4u}), w, p + "decoder_block3.", 768, mode);
    gate(p3, "3");
    Tensor p2_in = upsample_bilinear2d(p3, h2, w2);
    add_(p2_in, conv_named(x2, w, p + "lateral_block3.conv", 768, 1, 0, true));
    Tensor ipt3u = upsample_bilinear2d(ipt3, h2, w2);
    Tensor p2 = dec_blk(cat_channels({&p2_in, &ipt3u}), w, p + "decoder_block2.", 384, mode);
    gate(p2, "2");
    Tensor p1_in = upsample_bilinear2d(p2, h1, w1);
    add_(p1_in, conv_named(x1, w, p + "lateral_block2.conv", 384, 1, 0, true));
    Tensor ipt2u = upsample_bilinear2d(ipt2, h1, w1);
    Tensor p1 = dec_blk(cat_channels({&p1_in, &ipt2u}), w, p + "decoder_block1.", 192, mode);
    Tensor p1u = upsample_bilinear2d(p1, H, W), ipt1u = upsample_bilinear2d(ipt1, H, W);
    return conv_named(cat_channels({&p1u, &ipt1u}), w, p + "conv_out1.0", 1, 1, 0, true);
}

struct Parts { std::vector<Tensor> f, fh; Tensor x1, x2, x3, x4, x4s; };

// BiRefNet::forward_logits (birefnet.rs:412-461)
static Tensor forward_logits(const Tensor& x, const Weights& w, const SwinCfg& sc, int mode, Parts* parts) {
    const int64_t H = x.dim(2), W = x.dim(3);
    std::vector<Tensor> f = swin_forward(x, w, "bb.", sc);
    Tensor xh = upsample_bilinear2d(x, H / 2, W / 2);                 // birefnet.rs:425
    std::vector<Tensor> fh = swin_forward(xh, w, "bb.", sc);
    Tensor xs[4];
    for (int i = 0; i < 4; ++i) {
        Tensor u = upsample_bilinear2d(fh[i], f[i].dim(2), f[i].dim(3));   // birefnet.rs:435-438
        xs[i] = cat_channels({&f[i], &u});                            // birefnet.rs:440-443
    }
    const int64_t h4 = xs[3].dim(2), w4 = xs[3].dim(3);
    Tensor a = upsample_bilinear2d(xs[0], h4, w4), b = upsample_bilinear2d(xs[1], h4, w4), c = upsample_bilinear2d(xs[2], h4, w4);
    Tensor x4 = cat_channels({&a, &b, &c, &xs[3]});                   // birefnet.rs:450-453
    Tensor x4s = dec_blk(x4, w, "squeeze_module.0.", 3072, mode);     // birefnet.rs:457
    Tensor out = decoder_forward(x, xs[0], xs[1], xs[2], x4s, w, "decoder.", mode);
    if (parts) { parts->f = f; parts->fh = fh; parts->x1 = xs[0]; parts->x2 = xs[1]; parts->x3 = xs[2]; parts->x4 = x4; parts->x4s = x4s; }
    return out;
}

static SwinCfg swin_cfg(const brn_config* c) {
    SwinCfg s;
    s.embed = c->embed_dim; s.ws = c->window_size; s.patch = c->patch_size; s.in_ch = c->in_channels; s.mlp_ratio = c->mlp_ratio;
    for (int i = 0; i < 4; ++i) { s.depths[i] = c->depths[i]; s.heads[i] = c->num_heads[i]; }
    return s;
}

static thread_local std::string g_err;
template <class F> static int guarded(F&& f) {
    try { f(); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
static Tensor from_ptr(const float* p, std::vector<int64_t> s) { Tensor t(std::move(s)); memcpy(t.p(), p, (size_t)t.numel() * sizeof(float)); return t; }

}  // namespace orc

using namespace orc;

extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }
int orc_num_threads(void) { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

int orc_forward_logits(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const float* x, int B, int H, int W, float* out) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor y = forward_logits(from_ptr(x, {B, 3, H, W}), w, swin_cfg(cfg), cfg->deform_mode, nullptr);
        memcpy(out, y.p(), (size_t)y.numel() * sizeof(float));
    });
}

// forward_logits plus the intermediate tensors bench_inference.rs times one by one; any out pointer may be NULL
int orc_forward_parts(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const float* x, int B, int H, int W, float* out,
                      float* const f[4], float* x1, float* x2, float* x3, float* x4, float* x4s) {
    return guarded([&] {
        Weights w(weights, n);
        Parts p;
        Tensor y = forward_logits(from_ptr(x, {B, 3, H, W}), w, swin_cfg(cfg), cfg->deform_mode, &p);
        auto cp = [](float* d, const Tensor& t) { if (d) memcpy(d, t.p(), (size_t)t.numel() * sizeof(float)); };
        cp(out, y);
        if (f) for (int i = 0; i < 4; ++i) cp(f[i], p.f[i]);
        cp(x1, p.x1); cp(x2, p.x2); cp(x3, p.x3); cp(x4, p.x4); cp(x4s, p.x4s);
    });
}

int orc_swin_forward(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const char* prefix, const float* x, int B, int H,
                     int W, float* const outs[4]) {
    return guarded([&] {
        Weights w(weights, n);
        std::vector<Tensor> f = swin_forward(from_ptr(x, {B, (int64_t)cfg->in_channels, H, W}), w, prefix ? prefix : "", swin_cfg(cfg));
        for (int i = 0; i < 4; ++i) memcpy(outs[i], f[i].p(), (size_t)f[i].numel() * sizeof(float));
    });
}

int orc_linear(const float* x, int M, int K, const float* w, const float* bias, int N, int act, const float* residual, float* y) {
    return guarded([&] {
        Tensor t = linear(from_ptr(x, {M, K}), w, bias, N);
        if (act == 1) relu_(t); else if (act == 2) gelu_erf_(t);
        if (residual) for (int64_t i = 0; i < t.numel(); ++i) t.d[i] += residual[i];
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
int orc_layer_norm(const float* x, int rows, int C, const float* g, const float* b, float eps, float* y) {
    return guarded([&] { Tensor t = layer_norm(from_ptr(x, {rows, C}), g, b, eps); memcpy(y, t.p(), (size_t)t.numel() * sizeof(float)); });
}
int orc_conv2d(const float* x, int B, int C, int H, int W, const float* w, const float* bias, int O, int kh, int kw, int stride, int pad,
               int dil, const float* bn_g, const float* bn_b, const float* bn_m, const float* bn_v, float eps, int act, float* y) {
    return guarded([&] {
        Tensor t = conv2d(from_ptr(x, {B, C, H, W}), w, bias, O, kh, kw, stride, pad, dil);
        if (bn_g) batch_norm_(t, bn_g, bn_b, bn_m, bn_v, eps);
        if (act == 1) relu_(t); else if (act == 2) gelu_erf_(t);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
int orc_upsample_bilinear2d(const float* x, int B, int C, int H, int W, int oh, int ow, float* y) {
    return guarded([&] { Tensor t = upsample_bilinear2d(from_ptr(x, {B, C, H, W}), oh, ow); memcpy(y, t.p(), (size_t)t.numel() * sizeof(float)); });
}
// weights: names "attn.qkv.weight" ... under `prefix`
int orc_window_attention(const float* x, int B, int H, int W, int C, int heads, int ws, int shift, const brn_named_tensor* weights, size_t n,
                         const char* prefix, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = attention_half(from_ptr(x, {B, H, W, C}), w, prefix ? prefix : "", heads, ws, shift, nullptr);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
int orc_patch_merging(const float* x, int B, int H, int W, int C, const brn_named_tensor* weights, size_t n, const char* prefix, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = patch_merging(from_ptr(x, {B, (int64_t)H * W, C}), H, W, w, prefix ? prefix : "");
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
// DeformableConv2d::forward (deform_conv.rs:82-99 / :101-215)
int orc_deform_conv2d(const float* x, int B, int C, int H, int W, const float* ow, const float* ob, const float* mw, const float* mb,
                      const float* w, const float* bias, int O, int k, int stride, int pad, int mode, float* y) {
    return guarded([&] {
        Tensor xt = from_ptr(x, {B, C, H, W});
        Tensor offset = conv2d(xt, ow, ob, 2 * k * k, k, k, stride, pad, 1);   // deform_conv.rs:83
        Tensor mask = conv2d(xt, mw, mb, k * k, k, k, stride, pad, 1);         // deform_conv.rs:85
        for (auto& v : mask.d) v = (1.0f / (std::exp(-v) + 1.0f)) * 2.0f;      // deform_conv.rs:86
        Tensor t = mode == 0 ? conv2d(xt, w, bias, O, k, k, stride, pad, 1)    // deform_conv.rs:95-98
                             : deform_conv2d(xt, offset, mask, w, bias, O, k, stride, pad);   // deform_conv.rs:101-215
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
// ASPPDeformable::forward (aspp.rs:303-333) on a 64-channel map; weights under `prefix`
int orc_aspp(const brn_named_tensor* weights, size_t n, const char* prefix, int in_channels, int out_channels, int mode, const float* x, int B, int H,
             int W_, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = aspp_deformable(from_ptr(x, {B, in_channels, H, W_}), w, prefix ? prefix : "", mode, out_channels);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
// BasicDecBlk::new(in_channels, out_channels, &DecoderConfig{use_aspp_deformable, inter_channels_adaptive: false}, vb.pp(prefix)) + forward
// (decoder.rs:87-141)
int orc_decblk(const brn_named_tensor* weights, size_t n, const char* prefix, int cin, int cout, int inter, int use_aspp, int mode, const float* x, int B,
               int H, int W_, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = dec_blk(from_ptr(x, {B, cin, H, W_}), w, prefix ? prefix : "", cout, mode, use_aspp != 0, inter > 0 ? inter : 64);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
int orc_squeeze(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const float* x4, int B, int h, int w_, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = dec_blk(from_ptr(x4, {B, 5760, h, w_}), w, "squeeze_module.0.", 3072, cfg->deform_mode);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}
int orc_decoder(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const float* x, const float* x1, const float* x2,
                const float* x3, const float* x4, int B, int H, int W, float* y) {
    return guarded([&] {
        Weights w(weights, n);
        Tensor t = decoder_forward(from_ptr(x, {B, 3, H, W}), from_ptr(x1, {B, 384, H / 4, W / 4}), from_ptr(x2, {B, 768, H / 8, W / 8}),
                                   from_ptr(x3, {B, 1536, H / 16, W / 16}), from_ptr(x4, {B, 3072, H / 32, W / 32}), w, "decoder.", cfg->deform_mode);
        memcpy(y, t.p(), (size_t)t.numel() * sizeof(float));
    });
}

}  // extern "C"
