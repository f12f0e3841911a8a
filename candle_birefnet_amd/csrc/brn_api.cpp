// brn_api.cpp — the extern "C" boundary (include/birefnet_hip.h).  Exceptions stop here; every entry returns a status
// and leaves a thread-local message for brn_last_error().
#include "brn_host.h"
#include <cstring>
#include <cstdio>
#include <cmath>
#include <functional>
#include <memory>
#include <cstdlib>

namespace brn {
const char* last_error_cstr();

template <class F>
static brn_status guarded(F&& f) {
    try {
        (void)hipGetLastError();              // an error another library (or an earlier failed call) left in this thread is not ours to report
        f();
        return BRN_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return BRN_ERR_OOM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return BRN_ERR_INVALID_ARG;
    }
}

// per-call staging of caller buffers: BRN_MEM_HOST buffers travel through temporary HBM allocations
struct Staging {
    hipStream_t s; brn_mem loc;
    std::vector<void*> tmp;
    struct Out { float* host; float* dev; size_t n; };
    std::vector<Out> outs;
    Staging(void* stream, brn_mem l) : s((hipStream_t)stream), loc(l) {}
    float* dalloc(size_t n) {
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, n * sizeof(float) + 16);
        if (e != hipSuccess) fail(BRN_ERR_OOM, "hipMalloc of %zu bytes failed: %s", n * sizeof(float), hipGetErrorString(e));
        tmp.push_back(d);
        return (float*)d;
    }
    const float* in(const float* p, size_t n) {
        if (!p) fail(BRN_ERR_INVALID_ARG, "null input pointer");
        if (loc == BRN_MEM_DEVICE) return p;
        float* d = dalloc(n);
        BRN_HIP(hipMemcpyAsync(d, p, n * sizeof(float), hipMemcpyHostToDevice, s));
        return d;
    }
    float* out(float* p, size_t n) {
        if (!p) fail(BRN_ERR_INVALID_ARG, "null output pointer");
        if (loc == BRN_MEM_DEVICE) return p;
        float* d = dalloc(n);
        outs.push_back({p, d, n});
        return d;
    }
    void finish() {
        for (auto& o : outs) BRN_HIP(hipMemcpyAsync(o.host, o.dev, o.n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (loc == BRN_MEM_HOST) BRN_HIP(hipStreamSynchronize(s));
    }
    ~Staging() {
        if (!tmp.empty()) (void)hipStreamSynchronize(s);
        for (void* p : tmp) (void)hipFree(p);
    }
};

// run a graph fragment with a private arena: plan (dry), allocate, run
static void with_arena(hipStream_t s, const std::function<void(Ctx&)>& fn, int bf16 = 0) {   // bf16: 0 fp32 maps, 1 bf16, 2 fp16 (Ctx::bf16)
    Arena a;
    a.dry = true;
    Ctx c{&a, s, true, false, nullptr, nullptr, nullptr};
    c.bf16 = bf16;
    fn(c);
    Arena real;
    real.cap = a.peak + 256;
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, real.cap);
    if (e != hipSuccess) fail(BRN_ERR_OOM, "workspace hipMalloc of %zu bytes failed: %s", real.cap, hipGetErrorString(e));
    real.base = (char*)d;
    Ctx c2{&real, s, false, false, nullptr, nullptr, nullptr};
    c2.bf16 = bf16;
    try {
        fn(c2);
        BRN_HIP(hipStreamSynchronize(s));
    } catch (...) {
        (void)hipStreamSynchronize(s);
        (void)hipFree(d);
        throw;
    }
    (void)hipFree(d);
}

// Size the workspace for a (B, H, W) request: a dry run of the forward with a counting arena.  Every request shape is planned
// on its own and the workspace only grows to the largest need seen (batch 16 at 512^2 followed by batch 1 at 2080^2 must not
// reserve batch 16 at 2080^2); a request that is <= a planned shape in every dimension fits without a new dry run.
static void plan_model(Model& m, int B, int H, int W) {
    if (m.arena.base)
        for (const Model::Planned& q : m.planned)
            if (B <= q.B && H <= q.H && W <= q.W) return;
    Arena dry;
    dry.dry = true;
    Ctx c{&dry, nullptr, true, false, nullptr, nullptr, nullptr};
    c.bf16 = m.bf16;
    model_forward(m, c, nullptr, B, H, W, nullptr, 0);
    size_t need = dry.peak + 4096;
    // staging for host-resident input / output of the full model
    need += ((size_t)B * 3 * H * W + (size_t)B * H * W) * sizeof(float) + 1024;
    if (!m.arena.base || need > m.arena.cap) {
        if (m.arena.base) { BRN_HIP(hipDeviceSynchronize()); (void)hipFree(m.arena.base); m.arena.base = nullptr; m.arena.cap = 0; m.planned.clear(); }
        void* d = nullptr;
        hipError_t e = hipMalloc(&d, need);
        if (e != hipSuccess) {
            (void)hipGetLastError();          // (the failed allocation must not be reported again by the next launch check)
            fail(BRN_ERR_OOM, "workspace hipMalloc of %zu bytes (B=%d, %dx%d) failed: %s", need, B, H, W, hipGetErrorString(e));
        }
        m.arena.base = (char*)d; m.arena.cap = need; m.arena.top = 0; m.arena.peak = 0; m.arena.dry = false;
    }
    if (m.planned.size() >= 16) m.planned.erase(m.planned.begin());
    m.planned.push_back({B, H, W});
    if (B > m.plan_B) m.plan_B = B;
    if (H > m.plan_H) m.plan_H = H;
    if (W > m.plan_W) m.plan_W = W;
}

static void collect_profile(Model& m, hipStream_t s) {
    BRN_HIP(hipStreamSynchronize(s));
    for (int f = 0; f < FAM_COUNT + REGION_COUNT; ++f) { m.fam_launches[f] = 0; m.fam_ms[f] = 0.f; m.fam_flop[f] = 0.0; m.fam_bytes[f] = 0.0; }
    // BRN_DUMP_LAUNCHES=<path>: one CSV row per launch of the last profiled forward (tuning aid)
    const char* dump = getenv("BRN_DUMP_LAUNCHES");
    FILE* df = dump ? fopen(dump, "w") : nullptr;
    if (df) fprintf(df, "family,M,N,K,ms,gflop,tflops,gbytes,region\n");
    for (auto& r : m.records) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0.f;
        if (df) fprintf(df, "%s,%d,%d,%d,%.4f,%.3f,%.2f,%.4f,%d\n", brn_kernel_family_name(r.fam), r.M, r.N, r.K, ms, r.flop / 1e9, ms > 0 ? r.flop / ms / 1e9 : 0.0, r.bytes / 1e9, r.region);
        m.fam_launches[r.fam]++; m.fam_ms[r.fam] += ms; m.fam_flop[r.fam] += r.flop; m.fam_bytes[r.fam] += r.bytes;
        if (r.region > REGION_NONE && r.region < REGION_COUNT) {
            const int g = FAM_COUNT + r.region;
            m.fam_launches[g]++; m.fam_ms[g] += ms; m.fam_flop[g] += r.flop; m.fam_bytes[g] += r.bytes;
        }
    }
    if (df) fclose(df);
    if (m.stage_ev_ok) {
        for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&m.last_ms[i], m.stage_ev[i], m.stage_ev[i + 1]);
        (void)hipEventElapsedTime(&m.last_ms[4], m.stage_ev[0], m.stage_ev[4]);
    }
}

static void validate_config(const brn_config& c) {
    const int bc[4] = {192, 384, 768, 1536};
    for (int i = 0; i < 4; ++i)
        if (c.backbone_channels[i] != bc[i] || c.backbone_channels[i] != (c.embed_dim << i))
            fail(BRN_ERR_INVALID_ARG, "backbone_channels/embed_dim must be the Swin-L plan [192,384,768,1536]: the decoder's ipt blocks "
                 "hard-wire it (birefnet.rs:189-193)");
    if (!c.mul_scl_ipt || c.n_cxt != 3 || c.cxt[0] != 192 || c.cxt[1] != 384 || c.cxt[2] != 768)
        fail(BRN_ERR_INVALID_ARG, "only mul_scl_ipt=true with cxt=[192,384,768] is supported (the reference decoder's channel plan "
             "is inconsistent otherwise, birefnet.rs:176-207)");
    if (c.deform_mode != BRN_DEFORM_REFERENCE_CPU && c.deform_mode != BRN_DEFORM_DEFORMABLE)
        fail(BRN_ERR_INVALID_ARG, "unknown deform_mode %d", c.deform_mode);
    for (int i = 0; i < 4; ++i)
        if (c.depths[i] < 1 || c.depths[i] > 64) fail(BRN_ERR_INVALID_ARG, "depths[%d]=%d out of range", i, c.depths[i]);
}

static void run_model_locked(Model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc, void* stream, int apply_sigmoid);

// sub-batches a device-resident batch of B images runs as (BRN_SPLIT_STREAMS, default 2; at least two images per part)
static int sub_batch_parts(int B, int override_parts = 0) {
    static const int parts_env = getenv("BRN_SPLIT_STREAMS") ? atoi(getenv("BRN_SPLIT_STREAMS")) : 2;
    const int want = override_parts > 0 ? override_parts : parts_env;     // brn_model_set_streams beats the environment
    int parts = want < 1 ? 1 : (want > 8 ? 8 : want);
    if (parts > B / 2) parts = B / 2;
    return parts < 1 ? 1 : parts;
}
// streams, events and workspaces of parts 1 .. parts-1, each workspace as large as the main one (which plan_model sized for the
// largest part).  false = a workspace could not be allocated (the caller then runs the batch as one part); nothing is left half-made
static bool ensure_side_arenas(Model& m, int parts) {
    if ((int)m.sides.size() < parts - 1) m.sides.resize(parts - 1);
    for (int k = 0; k < parts - 1; ++k) {
        Model::Side& sd = m.sides[k];
        if (!sd.stream) {
            BRN_HIP(hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking));
            BRN_HIP(hipEventCreateWithFlags(&sd.join_ev, hipEventDisableTiming));
        }
        if (sd.arena.base && sd.arena.cap >= m.arena.cap) continue;
        if (sd.arena.base) { BRN_HIP(hipDeviceSynchronize()); (void)hipFree(sd.arena.base); sd.arena.base = nullptr; sd.arena.cap = 0; }
        void* d = nullptr;
        const char* fault = getenv("BRN_FAULT_SIDE_ARENA");       // test hook: behave as if this allocation had failed
        hipError_t e = (fault && atoi(fault) != 0) ? hipErrorOutOfMemory : hipMalloc(&d, m.arena.cap);
        if (e != hipSuccess) {
            (void)hipGetLastError();                               // (must not be reported by the next launch check)
            return false;
        }
        sd.arena.base = (char*)d; sd.arena.cap = m.arena.cap;
    }
    return true;
}

static void run_model(Model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc, void* stream,
                      int apply_sigmoid) {
    if (!m || !x || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
    if (B < 1) fail(BRN_ERR_INVALID_ARG, "batch must be >= 1");
    if (m->decoder_only) fail(BRN_ERR_INVALID_ARG, "this handle holds only the decoder (brn_decoder_create): forward_logits needs a whole model");
    std::lock_guard<std::mutex> lk(m->mu);
    try {
        run_model_locked(m, x, B, H, W, in_loc, out, out_loc, stream, apply_sigmoid);
    } catch (...) {
        // a forward that failed half-way may have work in flight on its sub-batch / branch streams that no event of the next call
        // orders against: drain the device before the workspace can be handed out again
        (void)hipDeviceSynchronize();
        throw;
    }
}

static void run_model_locked(Model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc, void* stream,
                             int apply_sigmoid) {
    BRN_HIP(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    // a device-resident batch runs as `parts` sub-batches, each on its own stream with its own workspace (below): what has to fit a
    // workspace is then the LARGEST PART, and that is the shape the dry run plans (its GEMM plans — tiles, split-K scratch — are those of
    // the part, not of the whole batch); every workspace, the main one included, is sized to that peak
    int parts = sub_batch_parts(B, m->opt_parts);
    bool split = parts > 1 && !m->profiling && in_loc == BRN_MEM_DEVICE && out_loc == BRN_MEM_DEVICE;
    plan_model(*m, split ? (B + parts - 1) / parts : B, H, W);
    if (split && !ensure_side_arenas(*m, parts)) {
        // no memory for a second workspace: the batch runs as one part on one stream (planned as such) instead of failing
        split = false; parts = 1;
        plan_model(*m, B, H, W);
    }
    if (!m->done_ev) BRN_HIP(hipEventCreateWithFlags(&m->done_ev, hipEventDisableTiming));
    if (m->has_last && m->last_stream != s) BRN_HIP(hipStreamWaitEvent(s, m->done_ev, 0));   // previous forward still owns the arena
    m->arena.top = 0;
    const size_t n_in = (size_t)B * 3 * H * W, n_out = (size_t)B * H * W;
    const float* dx = x;
    float* dout = out;
    if (in_loc == BRN_MEM_HOST) {
        float* t = m->arena.alloc(n_in);
        BRN_HIP(hipMemcpyAsync(t, x, n_in * sizeof(float), hipMemcpyHostToDevice, s));
        dx = t;
    }
    if (out_loc == BRN_MEM_HOST) dout = m->arena.alloc(n_out);
    m->records.clear(); m->event_next = 0;
    if (m->profiling && !m->stage_ev_ok) {
        for (int i = 0; i < 6; ++i) BRN_HIP(hipEventCreate(&m->stage_ev[i]));
        m->stage_ev_ok = true;
    }
    // A batch as `parts` sub-batches on `parts` streams (images are independent units): the kernels of one part fill the CUs another
    // part's launch leaves idle in its last, partial round of tiles, and the write bursts of one part's epilogues fall into the K
    // loops of the others (measured at batch 8, 1024^2, bf16: +5.6 % with 2 parts).  BRN_SPLIT_STREAMS = number of
    // parts (default 2; 1 = one stream).  Each part has its own workspace; the results do not depend on how the host interleaves the
    // enqueues (same kernels, same plans per part, no atomics).  Profiled forwards run on one stream (per-launch events).
    // Independent branches of one forward (ASPP branches, the image-patch convolutions) go to auxiliary streams (brn_graph.cpp: Branch);
    // BRN_BRANCH_STREAMS=0 keeps everything on the forward's own stream.
    // Default: on when the batch runs as ONE part (measured: +1.4 % at batch 1, 1024^2; with two sub-batch streams the extra
    // concurrency costs 2.5 % at batch 8); a positive value is the mask of auxiliary streams to use, for every batch
    // (31 = all: 7 the ASPP branches, 8 the image-patch convolutions, 16 the lateral convolutions).
    static const int branches_env0 = getenv("BRN_BRANCH_STREAMS") ? atoi(getenv("BRN_BRANCH_STREAMS")) : -1;
    const int branches_env = m->opt_branches > -2 ? m->opt_branches : branches_env0;    // brn_model_set_streams beats the environment
    bool branches_on = branches_env != 0;
    auto branch_set = [&](int k) -> BranchSet* {
        if (!branches_on || m->profiling) return nullptr;
        if ((int)m->branch_sets.size() <= k) m->branch_sets.resize(k + 1);
        BranchSet& bs = m->branch_sets[k];
        for (int i = 0; i < BRN_AUX_STREAMS; ++i) {
            if (bs.stream[i]) continue;
            BRN_HIP(hipStreamCreateWithFlags(&bs.stream[i], hipStreamNonBlocking));
            BRN_HIP(hipEventCreateWithFlags(&bs.fork_ev[i], hipEventDisableTiming));
            BRN_HIP(hipEventCreateWithFlags(&bs.join_ev[i], hipEventDisableTiming));
        }
        return &bs;
    };
    if (split) {
        if (branches_env < 0) branches_on = false;
        if (!m->fork_ev) BRN_HIP(hipEventCreateWithFlags(&m->fork_ev, hipEventDisableTiming));
        for (int k = 0; k < parts - 1; ++k) {
            Model::Side& sd = m->sides[k];
            sd.arena.top = 0; sd.arena.peak = 0; sd.arena.dry = false;
        }
        // BRN_CU_PARTITION (A/B switch, two parts only): each part's stream is confined to its own half of the chip by a CU mask
        // (hipExtStreamCreateWithCUMask; mask bit i = CU i / 8 of XCD i % 8): 1 = half the CUs of every XCD (both parts share every L2),
        // 2 = four whole XCDs each.  The persistent GEMM grids are sized for the CUs of the mask (set_launch_cus).
        static const int cu_part = getenv("BRN_CU_PARTITION") ? atoi(getenv("BRN_CU_PARTITION")) : 0;
        const bool masked = cu_part > 0 && parts == 2;
        if (masked && !m->cu_stream[0]) {
            for (int k = 0; k < 2; ++k) {
                uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < 256; ++i) {
                    const bool mine = cu_part == 2 ? ((i % 8 < 4) == (k == 0)) : ((i < 128) == (k == 0));
                    if (mine) mask[i >> 5] |= 1u << (i & 31);
                }
                BRN_HIP(hipExtStreamCreateWithCUMask(&m->cu_stream[k], 8, mask));
                BRN_HIP(hipEventCreateWithFlags(&m->cu_join_ev[k], hipEventDisableTiming));
            }
        }
        struct CusGuard { bool on; explicit CusGuard(bool o) : on(o) { if (on) set_launch_cus(128); } ~CusGuard() { if (on) set_launch_cus(256); } } cus_guard(masked);
        BRN_HIP(hipEventRecord(m->fork_ev, s));
        int b0 = 0;
        for (int k = 0; k < parts; ++k) {
            const int bk = B / parts + (k < B % parts ? 1 : 0);
            hipStream_t sk = masked ? m->cu_stream[k] : (k == 0 ? s : m->sides[k - 1].stream);
            if (k > 0 || masked) BRN_HIP(hipStreamWaitEvent(sk, m->fork_ev, 0));
            Ctx ck{k == 0 ? &m->arena : &m->sides[k - 1].arena, sk, false, false, nullptr, nullptr, nullptr};
            ck.bf16 = m->bf16;
            ck.br = branch_set(k);
            if (branches_env > 0) ck.br_mask = (unsigned)branches_env;
            model_forward(*m, ck, dx + (size_t)b0 * 3 * H * W, bk, H, W, dout + (size_t)b0 * H * W, apply_sigmoid);
            if (masked) BRN_HIP(hipEventRecord(m->cu_join_ev[k], sk));
            else if (k > 0) BRN_HIP(hipEventRecord(m->sides[k - 1].join_ev, sk));
            b0 += bk;
        }
        if (masked) { for (int k = 0; k < 2; ++k) BRN_HIP(hipStreamWaitEvent(s, m->cu_join_ev[k], 0)); }
        else for (int k = 1; k < parts; ++k) BRN_HIP(hipStreamWaitEvent(s, m->sides[k - 1].join_ev, 0));
        BRN_HIP(hipEventRecord(m->done_ev, s));
        m->last_stream = s; m->has_last = true;
        return;
    }
    Ctx c{&m->arena, s, false, m->profiling, &m->records, &m->event_pool, &m->event_next};
    c.bf16 = m->bf16;
    c.br = branch_set(0);
    if (branches_env > 0) c.br_mask = (unsigned)branches_env;
    model_forward(*m, c, dx, B, H, W, dout, apply_sigmoid);
    if (out_loc == BRN_MEM_HOST) {
        BRN_HIP(hipMemcpyAsync(out, dout, n_out * sizeof(float), hipMemcpyDeviceToHost, s));
        BRN_HIP(hipStreamSynchronize(s));
    }
    BRN_HIP(hipEventRecord(m->done_ev, s));
    m->last_stream = s; m->has_last = true;
    if (m->profiling) collect_profile(*m, s);
}

}  // namespace brn

using namespace brn;

struct brn_model { Model m; };
struct brn_swin { brn_config cfg; int device; DeviceOwner own; SwinW w; std::mutex mu; int bf16 = 0; };

extern "C" {

int brn_abi_version(void) { return BRN_ABI_VERSION; }
const char* brn_last_error(void) { return last_error_cstr(); }
const char* brn_build_info(void) {
#define BRN_STR2(x) #x
#define BRN_STR(x) BRN_STR2(x)
    return "libbirefnet_hip gfx950 (CDNA4): compute modes f32 (fp32 MFMA), f32_split3 / f32_split2 (split-bf16 MFMA, fp32 storage), f32_half2 (fp16-pair MFMA, fp32 storage), bf16 / f16 (bf16 / fp16 storage + MFMA); HIP " BRN_STR(HIP_VERSION_MAJOR) "." BRN_STR(HIP_VERSION_MINOR) "." BRN_STR(HIP_VERSION_PATCH);
}
brn_status brn_device_count(int* n) {
    return guarded([&] {
        if (!n) fail(BRN_ERR_INVALID_ARG, "null argument");
        int k = 0;
        hipError_t e = hipGetDeviceCount(&k);
        *n = (e == hipSuccess) ? k : 0;
    });
}

void brn_config_default_swin_l(brn_config* c) {
    if (!c) return;
    memset(c, 0, sizeof *c);
    c->size_w = 1024; c->size_h = 1024;                       // birefnet.rs:35
    strncpy(c->backbone, "swin_v1_l", sizeof c->backbone - 1);  // birefnet.rs:36
    const int bc[4] = {192, 384, 768, 1536};                   // birefnet.rs:38
    const int dp[4] = {2, 2, 18, 2}, nh[4] = {6, 12, 24, 48};  // swin.rs:72-73
    for (int i = 0; i < 4; ++i) { c->backbone_channels[i] = bc[i]; c->depths[i] = dp[i]; c->num_heads[i] = nh[i]; }
    c->mul_scl_ipt = 1; c->ms_supervision = 1; c->dec_ipt = 1; c->use_aspp_deformable = 1;   // birefnet.rs:39-42
    c->cxt[0] = 192; c->cxt[1] = 384; c->cxt[2] = 768; c->n_cxt = 3;                          // birefnet.rs:43
    c->embed_dim = 192; c->window_size = 12; c->mlp_ratio = 4.0f; c->patch_size = 4; c->in_channels = 3;
    c->drop_path_rate = 0.2f;                                  // swin.rs:71-78
    c->deform_mode = BRN_DEFORM_REFERENCE_CPU;
}
void brn_config_lateral_channels(const brn_config* c, int out[4]) {
    const int mult = c->mul_scl_ipt ? 2 : 1;                   // birefnet.rs:51
    for (int i = 0; i < 4; ++i) out[i] = c->backbone_channels[i] * mult;
}
int brn_config_x4_channels(const brn_config* c) {
    const int mult = c->mul_scl_ipt ? 2 : 1;                   // birefnet.rs:57-60
    int s = c->backbone_channels[3] * mult;
    for (int i = 0; i < c->n_cxt; ++i) s += c->cxt[i] * mult;
    return s;
}

brn_status brn_model_create(const brn_config* cfg, const brn_named_tensor* weights, size_t n, int device, brn_dtype dt,
                            int max_batch, int max_h, int max_w, brn_model** out) {
    return guarded([&] {
        if (!cfg || !weights || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        int planes = 0;
        if (dt == BRN_F32) planes = 0;
        else if (dt == BRN_F32_SPLIT3) planes = 3;
        else if (dt == BRN_F32_SPLIT2) planes = 2;
        else if (dt == BRN_F32_HALF2) planes = BUILD_HALF2;
#ifdef BRN_DIAG_BUILD
        else if (dt == BRN_BF16_OPERANDS) planes = 1;
#else
        else if (dt == BRN_BF16_OPERANDS) fail(BRN_ERR_INVALID_ARG, "compute dtype BRN_BF16_OPERANDS is superseded by BRN_BF16 and only built into libbirefnet_hip_diag.so");
#endif
        else if (dt == BRN_BF16 || dt == BRN_BF16_DEC_SPLIT2 || dt == BRN_F16) planes = BUILD_BF16;
        else fail(BRN_ERR_INVALID_ARG, "unsupported compute dtype %d", (int)dt);
        struct PlanesGuard { PlanesGuard(int p, bool h) { set_build_planes(p); set_build_f16(h); } ~PlanesGuard() { set_build_planes(0); set_build_f16(false); } } guard(planes, dt == BRN_F16);
        *out = nullptr;
        ensure_device(device);
        validate_config(*cfg);
        std::unique_ptr<brn_model> h(new brn_model());
        Model& m = h->m;
        m.cfg = *cfg; m.device = device; m.bf16 = planes == BUILD_BF16 ? (dt == BRN_F16 ? 2 : 1) : 0; m.dec_bf16 = dt == BRN_BF16 ? 1 : (dt == BRN_F16 ? 2 : 0);
        WeightTable wt(weights, n);
        build_swin_weights(wt, "bb.", *cfg, m.own, m.swin);                               // birefnet.rs:393
        if (dt == BRN_BF16_DEC_SPLIT2) set_build_planes(2);                               // squeeze + decoder weights as two bf16 planes (mode f32_split2)
        int lat[4];
        brn_config_lateral_channels(cfg, lat);
        build_decblk_weights(wt, "squeeze_module.0.", brn_config_x4_channels(cfg), lat[3], cfg->deform_mode, m.own, m.squeeze);   // birefnet.rs:397-399
        build_decoder_weights(wt, "decoder.", *cfg, m.own, m.dec);                        // birefnet.rs:401
        m.has_decoder = true;
        // (the batch the caller announces will run as sub_batch_parts(max_batch) parts when it is device-resident: plan the part;
        // a host-resident or profiled call of that batch re-plans for the whole batch when it comes)
        if (max_batch > 0 && max_h > 0 && max_w > 0) { const int pp = sub_batch_parts(max_batch, m.opt_parts); plan_model(m, (max_batch + pp - 1) / pp, max_h, max_w); }
        *out = h.release();
    });
}
brn_status brn_decoder_create(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const char* prefix, int device, brn_dtype dt,
                              brn_model** out) {
    return guarded([&] {
        if (!cfg || !weights || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        int planes = 0;
        if (dt == BRN_F32) planes = 0;
        else if (dt == BRN_F32_SPLIT3) planes = 3;
        else if (dt == BRN_F32_SPLIT2) planes = 2;
        else if (dt == BRN_F32_HALF2) planes = BUILD_HALF2;
        else if (dt == BRN_BF16 || dt == BRN_F16) planes = BUILD_BF16;
        else if (dt == BRN_BF16_DEC_SPLIT2) planes = 2;                                    // (a decoder on its own in the mixed mode = mode f32_split2)
        else fail(BRN_ERR_INVALID_ARG, "unsupported compute dtype %d", (int)dt);
        struct PlanesGuard { PlanesGuard(int p, bool h) { set_build_planes(p); set_build_f16(h); } ~PlanesGuard() { set_build_planes(0); set_build_f16(false); } } guard(planes, dt == BRN_F16);
        *out = nullptr;
        ensure_device(device);
        validate_config(*cfg);
        std::unique_ptr<brn_model> h(new brn_model());
        Model& m = h->m;
        m.cfg = *cfg; m.device = device; m.bf16 = dt == BRN_BF16 ? 1 : (dt == BRN_F16 ? 2 : 0); m.dec_bf16 = m.bf16;
        WeightTable wt(weights, n);
        build_decoder_weights(wt, prefix ? prefix : "", *cfg, m.own, m.dec);               // birefnet.rs:170-273
        m.has_decoder = true; m.decoder_only = true;
        *out = h.release();
    });
}
brn_status brn_model_create_from_safetensors(const brn_config* cfg, const char* path, const char* prefix, int device, brn_dtype dt,
                                            int max_batch, int max_h, int max_w, brn_model** out) {
    brn_status st = guarded([&] {
        if (!cfg || !path || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        *out = nullptr;
    });
    if (st != BRN_OK) return st;
    SafetensorsFile f;
    std::vector<brn_named_tensor> named;
    st = guarded([&] { f.open(path); named = f.named(prefix); if (named.empty()) fail(BRN_ERR_MISSING_TENSOR, "no tensor under prefix '%s' in '%s'", prefix ? prefix : "", path); });
    if (st != BRN_OK) return st;
    return brn_model_create(cfg, named.data(), named.size(), device, dt, max_batch, max_h, max_w, out);   // copies; the mapping goes with f
}
void brn_model_destroy(brn_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->m.device);
    (void)hipDeviceSynchronize();
    delete m;
}

brn_status brn_forward_logits(brn_model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc,
                              void* stream) {
    return guarded([&] { run_model(m ? &m->m : nullptr, x, B, H, W, in_loc, out, out_loc, stream, 0); });
}
brn_status brn_forward(brn_model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc,
                       void* stream) {
    return guarded([&] { run_model(m ? &m->m : nullptr, x, B, H, W, in_loc, out, out_loc, stream, 1); });
}

brn_status brn_model_set_streams(brn_model* m, int sub_batch_streams, int branch_stream_mask) {
    return guarded([&] {
        if (!m) fail(BRN_ERR_INVALID_ARG, "null model");
        if (sub_batch_streams < 0 || sub_batch_streams > 8 || branch_stream_mask < -1 || branch_stream_mask > 31)
            fail(BRN_ERR_INVALID_ARG, "sub_batch_streams in 0 .. 8 (0 = default), branch_stream_mask in -1 .. 31 (-1 = automatic)");
        std::lock_guard<std::mutex> lk(m->m.mu);
        m->m.opt_parts = sub_batch_streams;
        m->m.opt_branches = branch_stream_mask;
    });
}
brn_status brn_model_set_profiling(brn_model* m, int enable) {
    return guarded([&] {
        if (!m) fail(BRN_ERR_INVALID_ARG, "null model");
        std::lock_guard<std::mutex> lk(m->m.mu);
        m->m.profiling = enable != 0;
    });
}
brn_status brn_model_last_timings(brn_model* m, float ms[5]) {
    return guarded([&] {
        if (!m || !ms) fail(BRN_ERR_INVALID_ARG, "null argument");
        for (int i = 0; i < 5; ++i) ms[i] = m->m.last_ms[i];
    });
}
brn_status brn_model_last_kernel_stats(brn_model* m, int n, int* launches, float* ms, double* flop, double* bytes, int* n_out) {
    return guarded([&] {
        if (!m || !launches || !ms || !flop || !bytes) fail(BRN_ERR_INVALID_ARG, "null argument");
        const int rows = FAM_COUNT + REGION_COUNT;      // families, then regions (row FAM_COUNT + REGION_NONE stays zero)
        const int k = n < rows ? n : rows;
        for (int f = 0; f < k; ++f) {
            launches[f] = m->m.fam_launches[f]; ms[f] = m->m.fam_ms[f]; flop[f] = m->m.fam_flop[f]; bytes[f] = m->m.fam_bytes[f];
        }
        if (n_out) *n_out = rows;
    });
}
const char* brn_kernel_family_name(int f) {
    static const char* names[FAM_COUNT] = {"gemm_dense", "gemm_conv_nhwc", "gemm_gather_nchw", "gemm_deform_nhwc",
                                           "window_attention", "layernorm", "resize", "elementwise"};
    static const char* regions[REGION_COUNT] = {"region_none", "region_aspp"};
    if (f >= FAM_COUNT && f < FAM_COUNT + REGION_COUNT) return regions[f - FAM_COUNT];
    return (f >= 0 && f < FAM_COUNT) ? names[f] : "?";
}

// ---- pieces of the model used individually by bench_inference.rs -------------------------------------------------------
static void swin_outputs_nchw(Ctx& c, const SwinW& w, const float* dx, int B, int H, int W, float* const douts[4]) {
    int hs[4], ws[4];
    swin_stage_dims(H, W, w.patch, hs, ws);
    Map hm[4];
    for (int i = 0; i < 4; ++i) hm[i] = new_map(c, B, hs[i], ws[i], w.embed_dim << i);
    swin_forward(c, w, dx, B, H, W, hm);
    if (!c.dry)
        for (int i = 0; i < 4; ++i)     // NHWC -> NCHW: the permute(0,3,1,2) of swin.rs:786-788
            BRN_HIP(launch_nhwc_to_nchw(hm[i].p, B, hm[i].C, hs[i], ws[i], hm[i].ld, 0, douts[i], c.stream, c.bf16));
}

static void swin_entry(const SwinW& w, int device, const float* x, int B, int H, int W, brn_mem in_loc, float* const outs[4],
                       brn_mem out_loc, void* stream, int bf16 = 0) {
    if (!x || !outs) fail(BRN_ERR_INVALID_ARG, "null argument");
    if (B < 1 || H < 1 || W < 1) fail(BRN_ERR_INVALID_ARG, "bad input shape");
    BRN_HIP(hipSetDevice(device));
    Staging si(stream, in_loc), so(stream, out_loc);
    const float* dx = si.in(x, (size_t)B * 3 * H * W);
    int hs[4], ws[4];
    swin_stage_dims(H, W, w.patch, hs, ws);
    float* douts[4];
    for (int i = 0; i < 4; ++i) douts[i] = so.out(outs[i], (size_t)B * (w.embed_dim << i) * hs[i] * ws[i]);
    with_arena((hipStream_t)stream, [&](Ctx& c) { swin_outputs_nchw(c, w, dx, B, H, W, douts); }, bf16);
    so.finish();
}

brn_status brn_model_backbone_forward(brn_model* m, const float* x, int B, int H, int W, brn_mem in_loc, float* const outs[4],
                                      brn_mem out_loc, void* stream) {
    return guarded([&] {
        if (!m) fail(BRN_ERR_INVALID_ARG, "null model");
        if (m->m.decoder_only) fail(BRN_ERR_INVALID_ARG, "this handle holds only the decoder (brn_decoder_create)");
        std::lock_guard<std::mutex> lk(m->m.mu);
        swin_entry(m->m.swin, m->m.device, x, B, H, W, in_loc, outs, out_loc, stream, m->m.bf16);
    });
}

brn_status brn_model_squeeze_forward(brn_model* m, const float* x4, int B, int h, int w, brn_mem in_loc, float* out,
                                     brn_mem out_loc, void* stream) {
    return guarded([&] {
        if (!m || !x4 || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        if (m->m.decoder_only) fail(BRN_ERR_INVALID_ARG, "this handle holds only the decoder (brn_decoder_create)");
        std::lock_guard<std::mutex> lk(m->m.mu);
        BRN_HIP(hipSetDevice(m->m.device));
        const int cin = m->m.squeeze.cin, cout = m->m.squeeze.cout;
        Staging si(stream, in_loc), so(stream, out_loc);
        const float* dx = si.in(x4, (size_t)B * cin * h * w);
        float* dy = so.out(out, (size_t)B * cout * h * w);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map X = new_map(c, B, h, w, cin), Y = new_map(c, B, h, w, cout);
            if (!c.dry) BRN_HIP(launch_nchw_to_nhwc(dx, B, cin, h, w, X.p, X.ld, 0, c.stream, c.bf16));
            decblk_forward(c, m->m.squeeze, X, Y, m->m.cfg.deform_mode);
            if (!c.dry) BRN_HIP(launch_nhwc_to_nchw(Y.p, B, cout, h, w, Y.ld, 0, dy, c.stream, c.bf16));
        }, m->m.dec_bf16);
        so.finish();
    });
}

brn_status brn_model_decoder_forward(brn_model* m, const float* x, const float* x1, const float* x2, const float* x3,
                                     const float* x4, int B, int H, int W, brn_mem in_loc, float* out, brn_mem out_loc,
                                     void* stream) {
    return guarded([&] {
        if (!m || !x || !x1 || !x2 || !x3 || !x4 || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        if (H % 32 || W % 32 || H < 32 || W < 32) fail(BRN_ERR_INVALID_ARG, "H and W must be positive multiples of 32");
        std::lock_guard<std::mutex> lk(m->m.mu);
        BRN_HIP(hipSetDevice(m->m.device));
        Staging si(stream, in_loc), so(stream, out_loc);
        const int hh[4] = {H / 4, H / 8, H / 16, H / 32}, ww[4] = {W / 4, W / 8, W / 16, W / 32};
        const int ch[4] = {384, 768, 1536, 3072};
        const float* src[4] = {x1, x2, x3, x4};
        const float* dsrc[4];
        for (int i = 0; i < 4; ++i) dsrc[i] = si.in(src[i], (size_t)B * ch[i] * hh[i] * ww[i]);
        const float* dx = si.in(x, (size_t)B * 3 * H * W);
        float* dy = so.out(out, (size_t)B * H * W);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map X1 = new_map(c, B, hh[0], ww[0], 384), X2 = new_map(c, B, hh[1], ww[1], 768), X3 = new_map(c, B, hh[2], ww[2], 1536);
            Map D4 = new_map(c, B, hh[3], ww[3], 3456);
            if (!c.dry) {
                BRN_HIP(launch_nchw_to_nhwc(dsrc[0], B, 384, hh[0], ww[0], X1.p, X1.ld, 0, c.stream, c.bf16));
                BRN_HIP(launch_nchw_to_nhwc(dsrc[1], B, 768, hh[1], ww[1], X2.p, X2.ld, 0, c.stream, c.bf16));
                BRN_HIP(launch_nchw_to_nhwc(dsrc[2], B, 1536, hh[2], ww[2], X3.p, X3.ld, 0, c.stream, c.bf16));
                BRN_HIP(launch_nchw_to_nhwc(dsrc[3], B, 3072, hh[3], ww[3], D4.p, D4.ld, 0, c.stream, c.bf16));
            }
            decoder_forward(c, m->m, dx, B, H, W, X1, X2, X3, D4, dy, 0);
        }, m->m.dec_bf16);
        so.finish();
    });
}

// ---- image pre/post-processing (infer_image.rs:44-67, 84-110) -----------------------------------------------------------------
extern "C++" {
namespace {

// One axis of image 0.25.9's resampler (imageops/sample.rs, horizontal_sample / vertical_sample): for every output index the
// first input index, the tap count and the normalised weights, computed in f32 in the crate's order of operations.
struct ResampleAxis {
    int in_n = 0, out_n = 0, max_taps = 0;
    std::vector<int> left, count;
    std::vector<float> w;      // [out_n][max_taps]
};
enum { FILTER_TRIANGLE = 0, FILTER_LANCZOS3 = 1 };

float sincf_image(float t) {
    const float a = t * 3.14159265358979323846f;     // f32::consts::PI
    return t == 0.0f ? 1.0f : sinf(a) / a;
}
float filter_kernel(int filter, float x) {
    if (filter == FILTER_TRIANGLE) return fabsf(x) < 1.0f ? 1.0f - fabsf(x) : 0.0f;
    return fabsf(x) < 3.0f ? sincf_image(x) * sincf_image(x / 3.0f) : 0.0f;
}
ResampleAxis make_axis(int in_n, int out_n, int filter) {
    ResampleAxis ax;
    ax.in_n = in_n; ax.out_n = out_n;
    const float support = filter == FILTER_TRIANGLE ? 1.0f : 3.0f;
    const float ratio = (float)in_n / (float)out_n;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = support * sratio;
    ax.left.resize(out_n); ax.count.resize(out_n);
    std::vector<std::vector<float>> ws(out_n);
    for (int o = 0; o < out_n; ++o) {
        float inputx = ((float)o + 0.5f) * ratio;
        long l = (long)floorf(inputx - src_support);
        l = std::min<long>(std::max<long>(l, 0), (long)in_n - 1);
        long r = (long)ceilf(inputx + src_support);
        r = std::min<long>(std::max<long>(r, l + 1), (long)in_n);
        inputx = inputx - 0.5f;
        float sum = 0.0f;
        for (long i = l; i < r; ++i) {
            const float wv = filter_kernel(filter, ((float)i - inputx) / sratio);
            ws[o].push_back(wv);
            sum += wv;
        }
        for (float& v : ws[o]) v /= sum;
        ax.left[o] = (int)l; ax.count[o] = (int)(r - l);
        ax.max_taps = std::max(ax.max_taps, (int)(r - l));
    }
    ax.w.assign((size_t)out_n * ax.max_taps, 0.0f);
    for (int o = 0; o < out_n; ++o) std::copy(ws[o].begin(), ws[o].end(), ax.w.begin() + (size_t)o * ax.max_taps);
    return ax;
}
struct DevAxis { int* left; int* count; float* w; int max_taps; };
DevAxis upload_axis(DeviceOwner& own, const ResampleAxis& ax) {
    DevAxis d;
    d.left = reinterpret_cast<int*>(own.upload(reinterpret_cast<const float*>(ax.left.data()), ax.left.size()));
    d.count = reinterpret_cast<int*>(own.upload(reinterpret_cast<const float*>(ax.count.data()), ax.count.size()));
    d.w = own.upload(ax.w);
    d.max_taps = ax.max_taps;
    return d;
}
unsigned char* dev_bytes(DeviceOwner& own, size_t n) {
    std::vector<float> z((n + 3) / 4 + 4, 0.f);
    return reinterpret_cast<unsigned char*>(own.upload(z));
}

}  // namespace
}  // extern "C++"

brn_status brn_preprocess_image(const unsigned char* pixels, int h, int w, int channels, int S, float* x_nchw, brn_mem out_loc,
                                int device, void* stream) {
    return guarded([&] {
        if (!pixels || !x_nchw) fail(BRN_ERR_INVALID_ARG, "null argument");
        if (h < 1 || w < 1 || S < 1 || !(channels == 3 || channels == 4))
            fail(BRN_ERR_INVALID_ARG, "preprocess: %dx%d image with %d channels to %d: need RGB8 or RGBA8 and positive sizes", h, w, channels, S);
        ensure_device(device);
        hipStream_t s = (hipStream_t)stream;
        DeviceOwner own;
        Staging so(stream, out_loc);
        float* dout = so.out(x_nchw, (size_t)3 * S * S);
        const ResampleAxis ay = make_axis(h, S, FILTER_TRIANGLE), ax = make_axis(w, S, FILTER_TRIANGLE);   // resize_exact(S, S, Triangle)
        const DevAxis dy = upload_axis(own, ay), dx = upload_axis(own, ax);
        unsigned char* din = dev_bytes(own, (size_t)h * w * channels);
        BRN_HIP(hipMemcpyAsync(din, pixels, (size_t)h * w * channels, hipMemcpyHostToDevice, s));
        std::vector<float> z((size_t)S * w * channels, 0.f);
        float* tmp = own.upload(z);                                    // the crate's intermediate Rgba32FImage (vertical pass first)
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};    // infer_image.rs:53-54
        BRN_HIP(launch_resample_v_u8(din, h, w, channels, S, dy.left, dy.count, dy.w, dy.max_taps, tmp, s));
        BRN_HIP(launch_resample_h(tmp, S, w, channels, S, dx.left, dx.count, dx.w, dx.max_taps, nullptr, dout, mean, stdv, s));
        so.finish();
        BRN_HIP(hipStreamSynchronize(s));                               // `own` frees the temporaries when this scope ends
    });
}

brn_status brn_postprocess_mask(const float* logits, int S, brn_mem in_loc, int apply_sigmoid, int out_h, int out_w,
                                unsigned char* mask, int device, void* stream) {
    return guarded([&] {
        if (!logits || !mask) fail(BRN_ERR_INVALID_ARG, "null argument");
        if (S < 1 || out_h < 1 || out_w < 1) fail(BRN_ERR_INVALID_ARG, "postprocess: sizes must be positive");
        ensure_device(device);
        hipStream_t s = (hipStream_t)stream;
        DeviceOwner own;
        Staging si(stream, in_loc);
        const float* dl = si.in(logits, (size_t)S * S);
        unsigned char* m8 = dev_bytes(own, (size_t)S * S);
        BRN_HIP(launch_mask_u8(dl, (long)S * S, apply_sigmoid, m8, s));                       // infer_image.rs:84-99
        const ResampleAxis ay = make_axis(S, out_h, FILTER_LANCZOS3), ax = make_axis(S, out_w, FILTER_LANCZOS3);   // :103-108
        const DevAxis dy = upload_axis(own, ay), dx = upload_axis(own, ax);
        std::vector<float> z((size_t)out_h * S, 0.f);
        float* tmp = own.upload(z);
        unsigned char* dout = dev_bytes(own, (size_t)out_h * out_w);
        BRN_HIP(launch_resample_v_u8(m8, S, S, 1, out_h, dy.left, dy.count, dy.w, dy.max_taps, tmp, s));
        BRN_HIP(launch_resample_h(tmp, out_h, S, 1, out_w, dx.left, dx.count, dx.w, dx.max_taps, dout, nullptr, nullptr, nullptr, s));
        BRN_HIP(hipMemcpyAsync(mask, dout, (size_t)out_h * out_w, hipMemcpyDeviceToHost, s));
        BRN_HIP(hipStreamSynchronize(s));
    });
}

// examples/infer_image.rs:44-110 for a batch (see the header).  The staging pool: [raw images | vertical-pass temporaries (pre) |
// x batch | mask probabilities | u8 masks at S | vertical-pass temporaries (post) | u8 masks at the images' sizes]
brn_status brn_infer_images_u8(brn_model* mh, int n, const unsigned char* const* pixels, const int* heights, const int* widths, int channels, int S,
                               unsigned char* const* masks, void* stream) {
    return guarded([&] {
        if (!mh || !pixels || !heights || !widths || !masks || n < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (!(channels == 3 || channels == 4) || S < 32 || S % 32) fail(BRN_ERR_INVALID_ARG, "infer_images: RGB8 / RGBA8 input and a model size that is a positive multiple of 32 (got %d channels, S = %d)", channels, S);
        Model& m = mh->m;
        if (m.decoder_only) fail(BRN_ERR_INVALID_ARG, "this handle holds only the decoder (brn_decoder_create)");
        for (int i = 0; i < n; ++i)
            if (!pixels[i] || !masks[i] || heights[i] < 1 || widths[i] < 1) fail(BRN_ERR_INVALID_ARG, "infer_images: image %d is null or empty", i);
        hipStream_t s = (hipStream_t)stream;
        std::lock_guard<std::mutex> io_lock(m.io_mu);      // the staging pool and the table cache belong to one call at a time
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        std::vector<size_t> o_raw(n), o_tv(n), o_pv(n), o_out(n);
        size_t total = 0;
        for (int i = 0; i < n; ++i) { o_raw[i] = total; total += al((size_t)heights[i] * widths[i] * channels); }
        for (int i = 0; i < n; ++i) { o_tv[i] = total; total += al((size_t)S * widths[i] * channels * sizeof(float)); }
        const size_t o_x = total; total += al((size_t)n * 3 * S * S * sizeof(float));
        const size_t o_p = total; total += al((size_t)n * S * S * sizeof(float));
        const size_t o_m8 = total; total += al((size_t)n * S * S);
        for (int i = 0; i < n; ++i) { o_pv[i] = total; total += al((size_t)heights[i] * S * sizeof(float)); }
        for (int i = 0; i < n; ++i) { o_out[i] = total; total += al((size_t)heights[i] * widths[i]); }
        {
            std::lock_guard<std::mutex> lk(m.mu);
            BRN_HIP(hipSetDevice(m.device));
            if (total > m.io.cap) {
                if (m.io.base) { BRN_HIP(hipDeviceSynchronize()); (void)hipFree(m.io.base); m.io.base = nullptr; m.io.cap = 0; }
                void* d = nullptr;
                hipError_t e = hipMalloc(&d, total);
                if (e != hipSuccess) { (void)hipGetLastError(); fail(BRN_ERR_OOM, "hipMalloc of %zu bytes for the image staging failed: %s", total, hipGetErrorString(e)); }
                m.io.base = (char*)d; m.io.cap = total;
            }
        }
        // resampling tables, cached with the handle by (input size, output size, filter); uploaded once
        auto axis = [&](int in_n, int out_n, int filter) -> const Model::AxisDev& {
            std::lock_guard<std::mutex> lk(m.mu);
            for (const Model::AxisDev& a : m.axes) if (a.in_n == in_n && a.out_n == out_n && a.filter == filter) return a;
            const ResampleAxis ax = make_axis(in_n, out_n, filter);
            const DevAxis d = upload_axis(m.own, ax);
            m.axes.push_back({in_n, out_n, filter, d.max_taps, d.left, d.count, d.w});
            return m.axes.back();
        };
        char* base = m.io.base;
        float* x = reinterpret_cast<float*>(base + o_x);
        float* prob = reinterpret_cast<float*>(base + o_p);
        const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};    // infer_image.rs:53-54
        for (int i = 0; i < n; ++i) {
            const int h = heights[i], w = widths[i];
            unsigned char* raw = reinterpret_cast<unsigned char*>(base + o_raw[i]);
            BRN_HIP(hipMemcpyAsync(raw, pixels[i], (size_t)h * w * channels, hipMemcpyHostToDevice, s));
            const Model::AxisDev ay = axis(h, S, FILTER_TRIANGLE), ax = axis(w, S, FILTER_TRIANGLE);          // resize_exact(S, S, Triangle)
            float* tv = reinterpret_cast<float*>(base + o_tv[i]);
            BRN_HIP(launch_resample_v_u8(raw, h, w, channels, S, ay.left, ay.count, ay.w, ay.max_taps, tv, s));
            BRN_HIP(launch_resample_h(tv, S, w, channels, S, ax.left, ax.count, ax.w, ax.max_taps, nullptr, x + (size_t)i * 3 * S * S, mean, stdv, s));
        }
        run_model(&m, x, n, S, S, BRN_MEM_DEVICE, prob, BRN_MEM_DEVICE, stream, 1);                           // forward(): sigmoid fused (birefnet.rs:466-469)
        unsigned char* m8 = reinterpret_cast<unsigned char*>(base + o_m8);
        BRN_HIP(launch_mask_u8(prob, (long)n * S * S, 0, m8, s));                                              // infer_image.rs:84-99
        for (int i = 0; i < n; ++i) {
            const int h = heights[i], w = widths[i];
            const Model::AxisDev ay = axis(S, h, FILTER_LANCZOS3), ax = axis(S, w, FILTER_LANCZOS3);          // :103-108
            float* pv = reinterpret_cast<float*>(base + o_pv[i]);
            unsigned char* dout = reinterpret_cast<unsigned char*>(base + o_out[i]);
            BRN_HIP(launch_resample_v_u8(m8 + (size_t)i * S * S, S, S, 1, h, ay.left, ay.count, ay.w, ay.max_taps, pv, s));
            BRN_HIP(launch_resample_h(pv, h, S, 1, w, ax.left, ax.count, ax.w, ax.max_taps, dout, nullptr, nullptr, nullptr, s));
            BRN_HIP(hipMemcpyAsync(masks[i], dout, (size_t)h * w, hipMemcpyDeviceToHost, s));
        }
        BRN_HIP(hipStreamSynchronize(s));
    });
}

// ---- op-level entry points (weights are always host pointers; x / y / residual follow `loc`) -------------------------------------
static thread_local int g_op_planes = 0;
static thread_local bool g_op_f16 = false;      // with BUILD_BF16: fp16 storage (BRN_F16)
static inline int op_s16(bool bf) { return bf ? (g_op_f16 ? 2 : 1) : 0; }
brn_status brn_set_op_compute(int dtype) {
    return guarded([&] {
        if (dtype == BRN_F32) g_op_planes = 0;
        else if (dtype == BRN_F32_SPLIT3) g_op_planes = 3;
        else if (dtype == BRN_F32_SPLIT2) g_op_planes = 2;
        else if (dtype == BRN_F32_HALF2) g_op_planes = BUILD_HALF2;
#ifdef BRN_DIAG_BUILD
        else if (dtype == BRN_BF16_OPERANDS) g_op_planes = 1;
#else
        else if (dtype == BRN_BF16_OPERANDS) fail(BRN_ERR_INVALID_ARG, "compute dtype BRN_BF16_OPERANDS is superseded by BRN_BF16 and only built into libbirefnet_hip_diag.so");
#endif
        else if (dtype == BRN_BF16 || dtype == BRN_F16) g_op_planes = BUILD_BF16;
        else fail(BRN_ERR_INVALID_ARG, "unsupported compute dtype %d", dtype);
        g_op_f16 = dtype == BRN_F16;
    });
}
struct OpPlanes { OpPlanes() { set_build_planes(g_op_planes); set_build_f16(g_op_f16); } ~OpPlanes() { set_build_planes(0); set_build_f16(false); } };

// ---- stand-alone SwinTransformer -----------------------------------------------------------------------------------------
brn_status brn_swin_create(const brn_config* cfg, const brn_named_tensor* weights, size_t n, const char* prefix, int device,
                           brn_swin** out) {
    return guarded([&] {
        if (!cfg || !weights || !out) fail(BRN_ERR_INVALID_ARG, "null argument");
        *out = nullptr;
        ensure_device(device);
        std::unique_ptr<brn_swin> h(new brn_swin());
        h->cfg = *cfg; h->device = device;
        WeightTable wt(weights, n);
        OpPlanes op_planes;                    // the arithmetic brn_set_op_compute selected on this thread (default BRN_F32)
        h->bf16 = op_s16(g_op_planes == BUILD_BF16);
        build_swin_weights(wt, prefix ? prefix : "", *cfg, h->own, h->w);
        *out = h.release();
    });
}
void brn_swin_destroy(brn_swin* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    delete s;
}
brn_status brn_swin_forward(brn_swin* s, const float* x, int B, int H, int W, brn_mem in_loc, float* const outs[4],
                            brn_mem out_loc, void* stream) {
    return guarded([&] {
        if (!s) fail(BRN_ERR_INVALID_ARG, "null handle");
        std::lock_guard<std::mutex> lk(s->mu);
        swin_entry(s->w, s->device, x, B, H, W, in_loc, outs, out_loc, stream, s->bf16);
    });
}


brn_status brn_linear_forward(const float* x, int M, int K, const float* w, const float* bias, int N, int act,
                              const float* residual, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !w || !y || M < 1 || N < 1 || K < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        ensure_device(device);
        DeviceOwner own;
        OpPlanes op_planes;
        GemmW g = make_linear(own, w, bias, N, K);
        g.act = act;
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)M * K);
        const float* dr = residual ? st.in(residual, (size_t)M * N) : nullptr;
        float* dy = st.out(y, (size_t)M * N);
        if (g_op_planes == BUILD_BF16) {
            // compute mode BRN_BF16 at op level: x is rounded to bf16 at the edge (as the producing kernel of the model would have
            // written it), the product runs on kernels/gemm_bf16.hip, y (and the residual) stay fp32 at this boundary
            with_arena((hipStream_t)stream, [&](Ctx& c) {
                float* xb = c.arena->alloc_bytes((size_t)M * K * 2);
                if (!c.dry) BRN_HIP(launch_f32_to_bf16(dx, (size_t)M * K, xb, c.stream, g_op_f16));
                run_gemm(c, g, xb, M, K, dy, N, 0, dr, N, 0, nullptr, 0, 0, 0, 1, 1);
            }, op_s16(true));
        } else
        with_arena((hipStream_t)stream, [&](Ctx& c) { run_gemm(c, g, dx, M, K, dy, N, 0, dr, N, 0); });
        st.finish();
    });
}

brn_status brn_linear_residual_layer_norm_forward(const float* x, int M, int K, const float* w, const float* bias, int N, const float* residual,
                                                  const float* gamma, const float* beta, float eps, float* x_out, float* y_out, brn_mem loc,
                                                  int device, void* stream) {
    return guarded([&] {
        if (!x || !w || !residual || !gamma || !beta || !x_out || !y_out || M < 1 || N < 1 || K < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (N % 4 || N > 3072) fail(BRN_ERR_INVALID_ARG, "layer_norm width %d unsupported (multiple of 4, <= 3072)", N);
        if (eps != 1e-5f) fail(BRN_ERR_INVALID_ARG, "the fused projection + LayerNorm kernels are built for eps = 1e-5 (swin.rs:333-335)");
        ensure_device(device);
        DeviceOwner own;
        OpPlanes op_planes;
        GemmW g = make_linear(own, w, bias, N, K);
        if (g_op_planes == BUILD_BF16 && K == 192 && N == 192) attach_dense_frags(own, g, w);
        LNW ln; ln.C = N; ln.g = own.upload(gamma, N); ln.b = own.upload(beta, N);
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)M * K);
        const float* dr = st.in(residual, (size_t)M * N);
        float* dxo = st.out(x_out, (size_t)M * N);
        float* dyo = st.out(y_out, (size_t)M * N);
        const bool bf = g_op_planes == BUILD_BF16;
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            const float* a = dx;
            float* yb = dyo;
            if (bf) {                                                   // x rounded to bf16 at the edge, y produced as a bf16 matrix and widened
                float* xb = c.arena->alloc_bytes((size_t)M * K * 2);
                yb = c.arena->alloc_bytes((size_t)M * N * 2);
                if (!c.dry) BRN_HIP(launch_f32_to_bf16(dx, (size_t)M * K, xb, c.stream, g_op_f16));
                a = xb;
            }
            // the residual stream is updated in place inside the model: here x_out starts as a copy of the residual
            if (!c.dry) BRN_HIP(hipMemcpyAsync(dxo, dr, (size_t)M * N * sizeof(float), hipMemcpyDeviceToDevice, c.stream));
            if (!linear_residual_ln(c, g, a, M, K, dxo, ln, yb, N, true)) {
                run_gemm(c, g, a, M, K, dxo, N, 0, dxo, N, 0, nullptr, 0, 0, 0, bf ? 1 : 0, bf ? 1 : 0);
                run_layernorm(c, ln, dxo, M, N, yb, N, 0, 0, bf ? 1 : 0);
            }
            if (bf && !c.dry) BRN_HIP(launch_bf16_to_f32(yb, (size_t)M * N, dyo, c.stream, g_op_f16));
        }, op_s16(bf));
        st.finish();
    });
}

brn_status brn_layer_norm_forward(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, float* y,
                                  brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !gamma || !beta || !y || rows < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (C % 4 || C > 3072) fail(BRN_ERR_INVALID_ARG, "layer_norm width %d unsupported (multiple of 4, <= 3072)", C);
        ensure_device(device);
        DeviceOwner own;
        LNW ln; ln.C = C; ln.g = own.upload(gamma, C); ln.b = own.upload(beta, C);
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)rows * C);
        float* dy = st.out(y, (size_t)rows * C);
        LayerNormParams p{};
        p.x = dx; p.y = dy; p.rows = rows; p.C = C; p.gamma = ln.g; p.beta = ln.b; p.eps = eps; p.ldx = C; p.ldy = C;
        BRN_HIP(launch_layernorm(p, (hipStream_t)stream));
        BRN_HIP(hipStreamSynchronize((hipStream_t)stream));
        st.finish();
    });
}

brn_status brn_conv2d_forward(const float* x, int B, int C, int H, int W, const float* w, const float* bias, int O, int kh,
                              int kw, int stride, int pad, int dil, const float* bn_g, const float* bn_b, const float* bn_m,
                              const float* bn_v, float bn_eps, int act, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !w || !y || B < 1 || C < 1 || O < 1 || kh < 1 || kw < 1 || stride < 1 || dil < 1 || pad < 0)
            fail(BRN_ERR_INVALID_ARG, "bad argument");
        const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
        if (Ho < 1 || Wo < 1) fail(BRN_ERR_INVALID_ARG, "empty conv output");
        ensure_device(device);
        DeviceOwner own;
        OpPlanes op_planes;
        const bool nhwc = (C % 32) == 0;
        GemmW g = nhwc ? make_conv_nhwc(own, w, nullptr, O, C, C, kh, kw, stride, pad, dil)
                       : make_conv_gather(own, w, nullptr, O, C, kh, kw, stride, pad, dil);
        if (bn_g) fold_bn(own, g, bias, bn_g, bn_b, bn_m, bn_v, bn_eps);
        else if (bias) g.bias = own.upload(bias, O);
        g.act = act;
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * C * H * W);
        float* dy = st.out(y, (size_t)B * O * Ho * Wo);
        const bool bf = g_op_planes == BUILD_BF16 && nhwc && C >= 64;   // bf16 mode: bf16 map in, bf16 map out, like inside the model
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map Y = new_map(c, B, Ho, Wo, O);
            if (nhwc) {
                Map X = new_map(c, B, H, W, C);
                if (!c.dry) BRN_HIP(launch_nchw_to_nhwc(dx, B, C, H, W, X.p, X.ld, 0, c.stream, c.bf16));
                run_conv(c, g, X, Y);
            } else {
                run_conv_nchw(c, g, dx, B, H, W, Y);
            }
            if (!c.dry) BRN_HIP(launch_nhwc_to_nchw(Y.p, B, O, Ho, Wo, Y.ld, 0, dy, c.stream, c.bf16));
        }, op_s16(bf));
        st.finish();
    });
}

brn_status brn_upsample_bilinear2d(const float* x, int B, int C, int H, int W, int oh, int ow, float* y, brn_mem loc,
                                   int device, void* stream) {
    return guarded([&] {
        if (!x || !y || B < 1 || C < 1 || H < 1 || W < 1 || oh < 1 || ow < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        ensure_device(device);
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * C * H * W);
        float* dy = st.out(y, (size_t)B * C * oh * ow);
        BRN_HIP(launch_resize_nchw(dx, B * C, H, W, dy, oh, ow, (hipStream_t)stream));
        BRN_HIP(hipStreamSynchronize((hipStream_t)stream));
        st.finish();
    });
}

brn_status brn_window_attention_forward(const float* x, int B, int H, int W, int C, int heads, int window_size, int shift,
                                        const float* qkv_w, const float* qkv_b, const float* proj_w, const float* proj_b,
                                        const float* rel_table, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !qkv_w || !qkv_b || !proj_w || !proj_b || !rel_table || !y) fail(BRN_ERR_INVALID_ARG, "null argument");
        if (!(window_size == 12 || window_size == 7) || heads < 1 || C != heads * 32 || !(shift == 0 || shift == window_size / 2))
            fail(BRN_ERR_INVALID_ARG, "window attention needs window_size 12 or 7, head_dim 32, shift 0 or window_size / 2");
        ensure_device(device);
        DeviceOwner own;
        // reuse the model's weight builder through a one-block table
        const int T = (2 * window_size - 1) * (2 * window_size - 1);
        int64_t s_qw[2] = {3 * C, C}, s_qb[1] = {3 * C}, s_pw[2] = {C, C}, s_pb[1] = {C}, s_t[2] = {T, heads};
        SwinBlockW bk;
        bk.heads = heads;
        OpPlanes op_planes;
        bk.qkv = make_linear(own, qkv_w, qkv_b, 3 * C, C);
        bk.proj = make_linear(own, proj_w, proj_b, C, C);
        (void)s_qw; (void)s_qb; (void)s_pw; (void)s_pb; (void)s_t;
        {
            std::vector<float> tt((size_t)T * heads);
            for (int t = 0; t < T; ++t) for (int h = 0; h < heads; ++h) tt[(size_t)h * T + t] = rel_table[(size_t)t * heads + h];
            bk.rel_table = own.upload(tt);
        }
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * H * W * C);
        float* dy = st.out(y, (size_t)B * H * W * C);
        if (g_op_planes == BUILD_BF16) {
            // compute mode BRN_BF16 at op level: x is rounded to bf16 at the edge (inside the model LayerNorm writes it as bf16), qkv and
            // the attention output are bf16 matrices (window_attention_bf16_kernel), y = proj(...) stays fp32 like the residual stream
            with_arena((hipStream_t)stream, [&](Ctx& c) {
                float* xb = c.arena->alloc_bytes((size_t)B * H * W * C * 2);
                if (!c.dry) BRN_HIP(launch_f32_to_bf16(dx, (size_t)B * H * W * C, xb, c.stream, g_op_f16));
                swin_attention(c, bk, xb, B, H, W, C, shift, dy, nullptr, window_size);
            }, op_s16(true));
        } else
        with_arena((hipStream_t)stream, [&](Ctx& c) { swin_attention(c, bk, dx, B, H, W, C, shift, dy, nullptr, window_size); });
        st.finish();
    });
}

brn_status brn_patch_merging_forward(const float* x, int B, int H, int W, int C, const float* ng, const float* nb,
                                     const float* rw, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !ng || !nb || !rw || !y || B < 1 || H < 1 || W < 1) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (C % 32 || 4 * C > 3072) fail(BRN_ERR_INVALID_ARG, "patch merging width %d unsupported", C);
        ensure_device(device);
        DeviceOwner own;
        LNW ln; ln.C = 4 * C; ln.g = own.upload(ng, 4 * C); ln.b = own.upload(nb, 4 * C);
        GemmW red = make_linear(own, rw, nullptr, 2 * C, 4 * C);
        const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, M2 = B * Ho * Wo;
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * H * W * C);
        float* dy = st.out(y, (size_t)M2 * 2 * C);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            float* pm = c.arena->alloc((size_t)M2 * 4 * C);
            if (!c.dry) {
                LayerNormParams p{};
                p.x = dx; p.y = pm; p.rows = M2; p.C = 4 * C; p.gamma = ln.g; p.beta = ln.b; p.eps = 1e-5f;
                p.ldy = 4 * C; p.mode = 1; p.H = H; p.W = W; p.Cin = C;
                BRN_HIP(launch_layernorm(p, c.stream));
            }
            run_gemm(c, red, pm, M2, 4 * C, dy, 2 * C, 0);
        });
        st.finish();
    });
}

brn_status brn_aspp_deformable_forward(const brn_named_tensor* weights, size_t n, const char* prefix, int in_channels, int out_channels, int mode,
                                       const float* x, int B, int H, int W, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!weights || !x || !y || B < 1 || H < 1 || W < 1 || in_channels < 1 || out_channels < 0) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (mode != BRN_DEFORM_REFERENCE_CPU && mode != BRN_DEFORM_DEFORMABLE) fail(BRN_ERR_INVALID_ARG, "unknown deform mode %d", mode);
        ensure_device(device);
        DeviceOwner own;
        OpPlanes op_planes;
        WeightTable wt(weights, n);
        ASPPW a;
        build_aspp_weights(wt, prefix ? prefix : "", mode, own, a, in_channels, out_channels);
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * a.ic * H * W);
        float* dy = st.out(y, (size_t)B * a.oc * H * W);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map T = new_map(c, B, H, W, a.icp), U = new_map(c, B, H, W, a.oc);
            if (!c.dry) {
                if (a.icp != a.ic) BRN_HIP(hipMemsetAsync(T.p, 0, (size_t)B * H * W * a.icp * c.esz(), c.stream));   // the pad channels
                BRN_HIP(launch_nchw_to_nhwc(dx, B, a.ic, H, W, T.p, T.ld, 0, c.stream, c.bf16));
            }
            aspp_forward(c, a, T, U, mode);
            if (!c.dry) BRN_HIP(launch_nhwc_to_nchw(U.p, B, a.oc, H, W, U.ld, 0, dy, c.stream, c.bf16));
        }, op_s16(g_op_planes == BUILD_BF16));
        st.finish();
    });
}

brn_status brn_decblk_forward(const brn_named_tensor* weights, size_t n, const char* prefix, int cin, int cout, int inter, int use_aspp, int mode,
                              const float* x, int B, int H, int W, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!weights || !x || !y || B < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || inter < 0) fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (mode != BRN_DEFORM_REFERENCE_CPU && mode != BRN_DEFORM_DEFORMABLE) fail(BRN_ERR_INVALID_ARG, "unknown deform mode %d", mode);
        ensure_device(device);
        DeviceOwner own;
        OpPlanes op_planes;
        WeightTable wt(weights, n);
        DecBlkW blk;
        build_decblk_weights(wt, prefix ? prefix : "", cin, cout, mode, own, blk, use_aspp != 0, inter > 0 ? inter : 64);
        const int cinp = blk.conv_in.Cinp;                   // in_channels rounded up to the kernels' channel granule (zero weights there)
        const bool bf = g_op_planes == BUILD_BF16;
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * cin * H * W);
        float* dy = st.out(y, (size_t)B * cout * H * W);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map X = new_map(c, B, H, W, cinp), Y = new_map(c, B, H, W, cout);
            if (!c.dry) {
                if (cinp != cin) BRN_HIP(hipMemsetAsync(X.p, 0, (size_t)B * H * W * cinp * c.esz(), c.stream));   // the pad channels
                BRN_HIP(launch_nchw_to_nhwc(dx, B, cin, H, W, X.p, X.ld, 0, c.stream, c.bf16));
            }
            decblk_forward(c, blk, X, Y, mode);
            if (!c.dry) BRN_HIP(launch_nhwc_to_nchw(Y.p, B, cout, H, W, Y.ld, 0, dy, c.stream, c.bf16));
        }, op_s16(bf));
        st.finish();
    });
}

brn_status brn_deform_conv2d_forward(const float* x, int B, int C, int H, int W, const float* offset_w, const float* offset_b,
                                     const float* mod_w, const float* mod_b, const float* w, const float* bias, int O, int k,
                                     int stride, int pad, int mode, float* y, brn_mem loc, int device, void* stream) {
    return guarded([&] {
        if (!x || !offset_w || !offset_b || !mod_w || !mod_b || !w || !y || k < 1 || stride < 1 || pad < 0)
            fail(BRN_ERR_INVALID_ARG, "bad argument");
        if (mode == BRN_DEFORM_REFERENCE_CPU) {
            // deform_conv.rs:95-98: offsets and modulator are computed and discarded; the result is regular_conv(x)
            brn_status s = brn_conv2d_forward(x, B, C, H, W, w, bias, O, k, k, stride, pad, 1, nullptr, nullptr, nullptr,
                                              nullptr, 0.f, BRN_ACT_NONE, y, loc, device, stream);
            if (s != BRN_OK) fail(s, "%s", brn_last_error());
            return;
        }
        if (mode != BRN_DEFORM_DEFORMABLE) fail(BRN_ERR_INVALID_ARG, "unknown deform mode %d", mode);
        const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1, kk = k * k;
        ensure_device(device);
        DeviceOwner own;
        // compute mode BRN_BF16 at op level (brn_set_op_compute): bf16 map in / out as inside the model, offsets / modulator fp32, the
        // gather on kernels/deform_bf16.hip where the shape allows; every other mode runs the fp32-MFMA gather kernel.
        // Any in_channels (deform_conv.rs:29-36): the channels-last map is padded with zero channels to the kernels' granule (32; 64 for
        // the bf16 gather kernel), the weights with zero columns
        const bool bf = g_op_planes == BUILD_BF16 && (O % 8) == 0;
        const int Cp = (C + (bf ? 63 : 31)) / (bf ? 64 : 32) * (bf ? 64 : 32);
        struct Planes { Planes(int p) { set_build_planes(p); set_build_f16(p == BUILD_BF16 && g_op_f16); } ~Planes() { set_build_planes(0); set_build_f16(false); } } planes_guard(bf ? BUILD_BF16 : 0);
        std::vector<float> w3((size_t)3 * kk * C * kk), b3((size_t)3 * kk);
        memcpy(w3.data(), offset_w, (size_t)2 * kk * C * kk * sizeof(float));
        memcpy(w3.data() + (size_t)2 * kk * C * kk, mod_w, (size_t)kk * C * kk * sizeof(float));
        memcpy(b3.data(), offset_b, (size_t)2 * kk * sizeof(float));
        memcpy(b3.data() + 2 * kk, mod_b, (size_t)kk * sizeof(float));
        GemmW om = make_conv_nhwc(own, w3.data(), b3.data(), 3 * kk, C, Cp, k, k, stride, pad, 1);
        om.mode = GEMM_CONV_NHWC;
        GemmW reg = make_conv_nhwc(own, w, bias, O, C, Cp, k, k, stride, pad, 1);
        reg.mode = GEMM_DEFORM_NHWC;
        if (bf) attach_deform_frags(own, reg, w);
        Staging st(stream, loc);
        const float* dx = st.in(x, (size_t)B * C * H * W);
        float* dy = st.out(y, (size_t)B * O * Ho * Wo);
        with_arena((hipStream_t)stream, [&](Ctx& c) {
            Map X = new_map(c, B, H, W, Cp), Y = new_map(c, B, Ho, Wo, O);
            if (Cp != C && !c.dry) BRN_HIP(hipMemsetAsync(X.p, 0, (size_t)B * H * W * Cp * c.esz(), c.stream));
            const int ldom = (3 * kk + 3) / 4 * 4;
            Map OM; OM.B = B; OM.H = Ho; OM.W = Wo; OM.C = 3 * kk; OM.ld = ldom; OM.coff = 0;
            OM.p = c.arena->alloc((size_t)B * Ho * Wo * ldom);
            if (!c.dry) BRN_HIP(launch_nchw_to_nhwc(dx, B, C, H, W, X.p, X.ld, 0, c.stream, c.bf16));
            run_conv(c, om, X, OM, nullptr, 0, 0, 1);
            const bool fused_sig = deform_fused_sigmoid(c, reg);
            if (!c.dry && !fused_sig) BRN_HIP(launch_mod_sigmoid2(OM.p, (size_t)B * Ho * Wo, ldom, 2 * kk, 3 * kk, c.stream));
            run_conv(c, reg, X, Y, OM.p, ldom, 2 * kk, 0, fused_sig ? 1 : 0);
            if (!c.dry) BRN_HIP(launch_nhwc_to_nchw(Y.p, B, O, Ho, Wo, Y.ld, 0, dy, c.stream, c.bf16));
        }, op_s16(bf));
        st.finish();
    });
}

}  // extern "C"
