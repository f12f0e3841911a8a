// brn_host.h — host-side runtime of libbirefnet_hip.so: error plumbing, HBM arena, prepared weights, the model graph.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <string>
#include <deque>
#include <vector>
#include <unordered_map>
#include <stdexcept>
#include <mutex>
#include <type_traits>
#include "../../include/birefnet_hip.h"
#include "brn_kernels.h"

namespace brn {

// ---- errors: C++ exceptions inside, status codes at the ABI ------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
[[noreturn]] void fail(int code, const char* fmt, ...);
void set_last_error(const std::string& s);
#define BRN_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) ::brn::fail(BRN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- kernel families for the per-launch accounting (brn_model_last_kernel_stats) --------------------------
enum Family {
    FAM_GEMM_DENSE = 0, FAM_GEMM_CONV, FAM_GEMM_GATHER, FAM_GEMM_DEFORM, FAM_ATTENTION, FAM_LAYERNORM,
    FAM_RESIZE, FAM_ELEMENTWISE, FAM_COUNT
};
// coarse graph regions, accounted beside the families (rows FAM_COUNT.. of brn_model_last_kernel_stats): the ASPPDeformable modules
// (aspp.rs:303-333: branch convs incl. offset / modulator convs and the deformable gathers, pooled branch, conv1) — BASELINE.md C5 asks
// for their HBM rate
enum Region { REGION_NONE = 0, REGION_ASPP = 1, REGION_COUNT };

struct LaunchRecord { int fam; double flop; double bytes; hipEvent_t e0, e1; int M, N, K; int region; };

// ---- HBM arena: one allocation, stack discipline (mark / release) ------------------------------------------
// In dry-run mode nothing is allocated or launched: the same graph code walks the plan and records the peak.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, top = 0, peak = 0;
    bool dry = false;
    float* alloc(size_t nfloats);
    float* alloc_bytes(size_t bytes) { return alloc((bytes + 3) / 4); }
    size_t mark() const { return top; }
    // hold > 0: nothing is handed out twice — the launches being enqueued run on several streams (graph branches), so "dead as soon
    // as the next launch is enqueued" does not hold for scratch; the owner of the hold releases to its own mark afterwards
    int hold = 0;
    void release(size_t m) { if (!hold) top = m; }
};

// auxiliary streams of one forward: branches of the graph that do not depend on each other (the four ASPP branches of a decoder
// block, the image-patch convolutions beside the backbone) are enqueued on them between a fork and a join event (brn_graph.cpp)
constexpr int BRN_AUX_STREAMS = 5;   // 0-2: ASPP branches, 3: image-patch convolutions, 4: lateral convolutions
struct BranchSet {
    hipStream_t stream[BRN_AUX_STREAMS] = {};
    hipEvent_t fork_ev[BRN_AUX_STREAMS] = {}, join_ev[BRN_AUX_STREAMS] = {};
};

struct Ctx {
    Arena* arena;
    hipStream_t stream;
    bool dry;                       // plan only
    bool profile;                   // bracket every launch with events
    std::vector<LaunchRecord>* records;
    std::vector<hipEvent_t>* event_pool; size_t* event_next;
    // compute mode BRN_BF16: activation maps are bf16 in HBM (esz = 2); pointers stay typed float* and are opaque to the host
    int bf16 = 0;                   // 0: fp32 maps; 1: bf16 (BRN_BF16); 2: fp16 (BRN_F16: the same graph, kernels of namespace brn::hf / the fp16 flavours)
    float h2_scale = 0.f;   // > 0 inside a Swin stage of mode f32_half2: P2-layout producers write fp16 planes of h2_scale * x (kernels/split_planes.h)
    int region = REGION_NONE;       // tag of the launches being recorded (profiling only)
    BranchSet* br = nullptr;        // null: every branch stays on `stream` (profiled forwards, the op-level entry points)
    unsigned pending = 0;           // branches enqueued on aux streams and not yet joined (bit k = aux stream k)
    unsigned br_mask = ~0u;         // aux streams in use (bit k); BRN_BRANCH_STREAMS: 7 = ASPP branches, 8 = image-patch convs, 16 = laterals
    int esz() const { return bf16 ? 2 : 4; }
    float* act_alloc(size_t elems) { return arena->alloc_bytes(elems * (size_t)esz()); }
    template <class T> T* at(T* p, size_t elems) const { return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + elems * (size_t)esz()); }
};

// channels-last window view: logical [B,H,W,C] living in columns [coff, coff+C) of rows that are ld floats wide
struct Map {
    float* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0, ld = 0, coff = 0;
    size_t pixels() const { return (size_t)B * H * W; }
    Map window(int c0, int c) const { Map m = *this; m.coff = coff + c0; m.C = c; return m; }
};
Map new_map(Ctx& c, int B, int H, int W, int C);

// ---- prepared weights ---------------------------------------------------------------------------------------
struct DevVec { float* p = nullptr; size_t n = 0; };

struct GemmW {          // one Linear / Conv2d, repacked for gemm_f32
    float* w = nullptr; // [roundup(N,128)][K]
    void* wp = nullptr; // split-bf16 path: [planes][roundup(N,128)][K] bf16 planes of the same matrix (or null)
    int planes = 0, wp_rows = 0;
    int half = 0;       // mode f32_half2: wp = two fp16 planes of w_scale * W (w_scale a power of two chosen per tensor: max |w| w_scale in (2^13, 2^14])
    float w_scale = 1.f;
    void* wb = nullptr; // bf16-storage mode (BRN_BF16): plain [wb_rows][wb_ld] bf16, rows padded to 256, K padded to 64 (zeros)
    int wb_rows = 0, wb_ld = 0;
    int wb_chunk_major = 0;   // wb's K order is (64-channel chunk, tap, channel in chunk) (channels-last convs with Cinp % 64 == 0, Cinp > 64: gemm_bf16.hip)
    void* wf = nullptr; // bf16-storage mode, deformable convs: the same matrix in MFMA fragment order [n/16][K/64][2][64 lanes][8] (kernels/deform_bf16.hip)
    int N = 0, K = 0, Kreal = 0;
    int Cin = 0, Cinp = 0, kh = 1, kw = 1, stride = 1, pad = 0, dil = 1;
    int mode = GEMM_DENSE;
    float* bias = nullptr;   // [N] or null
    float* scale = nullptr;  // folded eval-BN (with the conv bias folded into shift) or null
    float* shift = nullptr;
    int act = ACT_NONE;
};
struct LNW { float* g = nullptr; float* b = nullptr; int C = 0; };

struct SwinBlockW {
    LNW norm1, norm2;
    GemmW qkv, proj, fc1, fc2;
    float* rel_table = nullptr;   // [529][heads], as stored
    int heads = 0;
};
struct SwinStageW {
    std::vector<SwinBlockW> blocks;
    bool has_down = false;
    LNW down_norm; GemmW reduction;
    LNW out_norm;
    int C = 0, heads = 0;
};
struct SwinW {
    GemmW patch_proj; LNW patch_norm;
    SwinStageW stages[4];
    int embed_dim = 0, window = 12, patch = 4, in_ch = 3;
};
struct DeformW {           // DeformConvASPP (aspp.rs:13-56)
    int k = 1;
    GemmW offmod;          // offset_conv and modulator_conv stacked on N: [2k^2 | k^2] (deformable mode only)
    GemmW regular;         // regular_conv + the ASPP module's BN + ReLU folded (aspp.rs:217-222)
};
struct ASPPW {             // ASPPDeformable (aspp.rs:227-333)
    GemmW k1pair;          // reference_cpu mode: aspp1 and aspp_deforms.0 (both 1x1) stacked on N = 512
    DeformW d[4];          // aspp1, deform k1, k3, k7
    float* gap_w = nullptr; float* gap_scale = nullptr; float* gap_shift = nullptr;   // global_avg_pool.1/.2
    float* conv1_full = nullptr;   // [64][1280] (for the pooled branch's contribution)
    GemmW conv1_main;      // [oc][1024] + bn1 + relu
    int ic = 64, icp = 64, oc = 64;   // in_channels, the same rounded up to the kernels' channel granule (the input map carries icp channels), out_channels
};
struct DecBlkW { GemmW conv_in; ASPPW aspp; GemmW conv_out; int cin = 0, cout = 0; bool has_aspp = true; /* dec_att is Some (decoder.rs:107-111) */
                 int ic = 64, icp = 64; /* inter_channels and the channel count of the maps between the convs (granule-padded) */ };
struct SimpleConvsW { GemmW conv1, conv_out; };
struct DecoderW {
    SimpleConvsW ipt[5];   // ipt_blk2..5 at [1..4]; ipt_blk1 ([0]) is composed into head_k / head_b
    float* head_k = nullptr; float* head_b = nullptr;   // [3x3 border cases][5][5][3] + [9]: that stencil o ipt_blk1.conv1 (see brn_weights.cpp)
    DecBlkW dec[4];        // decoder_block4,3,2,1
    GemmW lat[3];          // lateral_block4,3,2
    GemmW gdt[3];          // gdt_convs_4,3,2 (conv + BN + ReLU)
    float* gdt_attn_w[3] = {nullptr, nullptr, nullptr}; float gdt_attn_b[3] = {0, 0, 0};
    float* out_w = nullptr; float out_b = 0.f;     // conv_out1.0 weight [240] (first 192 used by pixel_dot)
};

struct WeightTable {
    std::unordered_map<std::string, const brn_named_tensor*> map;
    WeightTable(const brn_named_tensor* w, size_t n);
    const brn_named_tensor* get(const std::string& name, std::initializer_list<int64_t> shape) const;
};

// a memory-mapped .safetensors file as brn_named_tensor views (brn_safetensors.cpp)
struct SafetensorsFile {
    struct Entry { std::string name; std::vector<int64_t> shape; const float* data = nullptr; };
    void* map = nullptr; size_t map_len = 0;
    std::vector<Entry> entries;
    std::deque<std::vector<float>> converted;       // F16 / BF16 (or misaligned F32) tensors widened to fp32
    SafetensorsFile() = default;
    SafetensorsFile(const SafetensorsFile&) = delete;
    SafetensorsFile& operator=(const SafetensorsFile&) = delete;
    ~SafetensorsFile();
    void open(const char* path);
    std::vector<brn_named_tensor> named(const char* prefix) const;   // views valid while *this lives; names have prefix stripped
};

struct DeviceOwner {     // every hipMalloc of a model, freed together
    std::vector<void*> ptrs;
    float* upload(const float* host, size_t n);
    float* upload(const std::vector<float>& v) { return upload(v.data(), v.size()); }
    ~DeviceOwner();
};

// ---- model ---------------------------------------------------------------------------------------------------
struct Model {
    brn_config cfg;
    int device = 0;
    DeviceOwner own;
    SwinW swin;
    DecBlkW squeeze;
    DecoderW dec;
    bool has_decoder = false;
    bool decoder_only = false;                // brn_decoder_create: no backbone / squeeze weights behind this handle
    Arena arena;
    int plan_B = 0, plan_H = 0, plan_W = 0;   // the largest request planned last (a request <= it in every dimension fits)
    struct Planned { int B, H, W; };
    std::vector<Planned> planned;             // shapes known to fit the workspace (requests are planned one by one, not as the product of the maxima)
    std::mutex mu;            // forward calls on one handle are serialised (one workspace)
    // the workspace is reused by every forward: a call on a different stream than the previous one first waits (on the GPU)
    // for the previous forward's last kernel, so two streams never overlap inside the arena
    hipEvent_t done_ev = nullptr; hipStream_t last_stream = nullptr; bool has_last = false;
    // two half batches on two streams (run_model): second workspace, side stream, fork / join events
    struct Side { Arena arena; hipStream_t stream = nullptr; hipEvent_t join_ev = nullptr; };
    std::vector<Side> sides; hipEvent_t fork_ev = nullptr;
    hipStream_t cu_stream[2] = {nullptr, nullptr}; hipEvent_t cu_join_ev[2] = {nullptr, nullptr};   // BRN_CU_PARTITION: CU-masked streams of the two parts
    std::vector<BranchSet> branch_sets;   // one per sub-batch stream (run_model)
    // brn_infer_images_u8: device staging of a batch of u8 images + their masks, grown on demand, and the resampling tables by (in, out, filter)
    struct IoPool { char* base = nullptr; size_t cap = 0; };
    IoPool io;
    std::mutex io_mu;         // brn_infer_images_u8 calls on one handle are serialised end to end (they share the staging pool and the tables)
    struct AxisDev { int in_n, out_n, filter, max_taps; int* left; int* count; float* w; };
    std::vector<AxisDev> axes;
    bool profiling = false;
    int opt_parts = 0;        // brn_model_set_streams: sub-batch streams of a device-resident batch (0 = BRN_SPLIT_STREAMS / default 2)
    int opt_branches = -2;    // ... and the mask of auxiliary branch streams (-2 = BRN_BRANCH_STREAMS / default; -1 = automatic; 0 = none)
    int bf16 = 0;             // BRN_BF16 / BRN_BF16_DEC_SPLIT2: the backbone's activations / weights are bf16 in HBM (1); BRN_F16: fp16 (2)
    int dec_bf16 = 0;        // the fusion / squeeze / decoder part too (BRN_BF16); false in BRN_BF16_DEC_SPLIT2: fp32 maps, split-bf16 GEMMs
    std::vector<LaunchRecord> records;
    std::vector<hipEvent_t> event_pool; size_t event_next = 0;
    hipEvent_t stage_ev[6]; bool stage_ev_ok = false;
    float last_ms[5] = {0, 0, 0, 0, 0};
    int fam_launches[FAM_COUNT + REGION_COUNT]; float fam_ms[FAM_COUNT + REGION_COUNT]; double fam_flop[FAM_COUNT + REGION_COUNT]; double fam_bytes[FAM_COUNT + REGION_COUNT];
    ~Model();
};

void build_swin_weights(const WeightTable& wt, const std::string& prefix, const brn_config& cfg, DeviceOwner& own, SwinW& out);
// ASPPDeformable::new(in_channels, out_channels (0 = in_channels), vb.pp(prefix)) (aspp.rs:236-300)
void build_aspp_weights(const WeightTable& wt, const std::string& prefix, int deform_mode, DeviceOwner& own, ASPPW& out, int in_channels = 64,
                        int out_channels = 0);
void build_decblk_weights(const WeightTable& wt, const std::string& prefix, int cin, int cout, int deform_mode, DeviceOwner& own, DecBlkW& out,
                          bool use_aspp = true, int inter_channels = 64 /* decoder.rs:94-98: 64, or in_channels / 4 when inter_channels_adaptive */);
void build_decoder_weights(const WeightTable& wt, const std::string& prefix, const brn_config& cfg, DeviceOwner& own, DecoderW& out);

// generic weight repack helpers (also used by the op-level entry points)
GemmW make_linear(DeviceOwner& own, const float* w, const float* bias, int N, int K);
// conv for a channels-last input whose channel count is cin_padded (>= Cin, % 32 == 0); NHWC K order
GemmW make_conv_nhwc(DeviceOwner& own, const float* w, const float* bias, int O, int Cin, int cin_padded, int kh, int kw,
                     int stride, int pad, int dil);
// conv that gathers straight from an NCHW tensor; candle K order
GemmW make_conv_gather(DeviceOwner& own, const float* w, const float* bias, int O, int Cin, int kh, int kw, int stride, int pad, int dil);
void fold_bn(DeviceOwner& own, GemmW& g, const float* conv_bias_host, const float* gamma, const float* beta,
             const float* mean, const float* var, float eps);

// ---- graph pieces ------------------------------------------------------------------------------------------------
void run_gemm(Ctx& c, const GemmW& w, const float* A, int M, int lda, float* C, int ldc, int c_coff,
              const float* R = nullptr, int ldr = 0, int r_coff = 0, const float* bbias = nullptr, int bbias_rows = 1,
              int a_planes = 0, int c_planes = 0 /* 2: operand in the P2 layout (kernels/split_planes.h) */,
              int c_f32 = 0, int r_f32 = 0 /* compute mode BRN_BF16 only: C written / R read as fp32 (the residual stream) */);
void run_conv(Ctx& c, const GemmW& w, const Map& in, const Map& out, const float* om = nullptr, int om_ld = 0, int om_mask_off = 0,
              int c_f32 = 0, int om_sigmoid = 0 /* the modulator columns are raw logits (only with deform_fused_sigmoid(c, w)) */);
// true when the deformable conv `w` runs on kernels/deform_bf16.hip, which applies 2 * sigmoid to the modulator itself
bool deform_fused_sigmoid(const Ctx& c, const GemmW& w);
// bf16-storage mode: attach the fragment-ordered copy of a channels-last conv weight (w: candle [O][Cin][kh][kw]) for deform_bf16
void attach_deform_frags(DeviceOwner& own, GemmW& g, const float* w_oihw);
// bf16-storage mode: the fragment-ordered copy of a Linear's [N][K] weight for gemm_wstat_bf16_kernel (K = 192, N % 192 == 0 only)
void attach_dense_frags(DeviceOwner& own, GemmW& g, const float* w);
void run_conv_nchw(Ctx& c, const GemmW& w, const float* x_nchw, int B, int Hin, int Win, const Map& out, bool pad_to_stride = false);
void run_layernorm(Ctx& c, const LNW& ln, const float* x, int rows, int ldx, float* y, int ldy, int y_coff, int y_planes = 0, int y_bf16 = 0);
void run_resize(Ctx& c, const Map& in, const Map& out, bool accumulate = false);
// compute mode BRN_BF16: x = A W^T + bias + x (fp32, in place) and y = LayerNorm(x) as a bf16 matrix in ONE launch where a row-owning
// kernel covers the shape (N = 192: gemm_wstat_ln_bf16_kernel; N = 768 / 384: gemm_rowln_bf16_kernel); false = nothing enqueued
// every_fused_kernel: also the kernels the model does not use by default (the op-level entry point exercises them all)
bool linear_residual_ln(Ctx& c, const GemmW& w, const float* A, int M, int lda, float* x, const LNW& ln, float* y, int ldy, bool every_fused_kernel = false);

// SwinTransformer::forward (swin.rs:768-797): outs[i] are destination windows (stage outputs after norm_i)
void swin_forward(Ctx& c, const SwinW& w, const float* img_nchw, int B, int H, int W, const Map outs[4]);
void swin_stage_dims(int H, int W, int patch, int hs[4], int ws[4]);
struct SwinIn { const float* img; int H, W; const Map* outs; };   // one backbone input and its 4 destination windows
// 1 or 2 inputs through the same weights in one pass over concatenated token rows; outs_f32: in compute mode BRN_BF16 the stage outputs
// (norm_i of the fp32 residual stream) are written as fp32 maps (the mixed mode, whose decoder runs on fp32 maps)
void swin_forward_multi(Ctx& c, const SwinW& w, const SwinIn* ins, int nin, int B, bool outs_f32 = false);
// the attention half of one block (swin.rs:356-403), x is the norm1 output, y = proj(attn) (no residual) or += residual
void swin_attention(Ctx& c, const SwinBlockW& blk, const float* xn, int B, int H, int W, int C, int shift,
                    float* y, const float* residual, int window = 12);
void decblk_forward(Ctx& c, const DecBlkW& w, const Map& in, const Map& out, int deform_mode, int out_f32 = 0 /* BRN_BF16: `out` is an fp32 map */);
// ASPPDeformable::forward (aspp.rs:303-333) on a whole map t of a.icp channels (the first a.ic real, the rest zero) -> map u of a.oc channels
void aspp_forward(Ctx& c, const ASPPW& a, const Map& t, const Map& u, int deform_mode);
// the decoder's concat maps (birefnet.rs:332,347,362) — allocated by the caller when the image-patch convolutions that fill their
// last channels are enqueued early, beside the backbone (model_forward)
struct DecMaps { Map d3, d2, d1; bool lat_done = false; };   // lat_done: lateral_block4/3/2 already wrote [0:C) of d3 / d2 / d1
void decoder_forward(Ctx& c, const Model& m, const float* img_nchw, int B, int H, int W, const Map& x1, const Map& x2,
                     const Map& x3, const Map& d4 /* [.., 3456] with [0:3072) = squeezed x4 */, float* out, int apply_sigmoid, const DecMaps* pre = nullptr);
void model_forward(Model& m, Ctx& c, const float* img_nchw, int B, int H, int W, float* out, int apply_sigmoid);

void ensure_device(int ordinal);
// number of bf16 planes the weight builders attach to every dense / channels-last conv GemmW (0 = fp32 MFMA path only);
// BUILD_BF16: attach the plain bf16 matrix of the bf16-storage mode instead
constexpr int BUILD_BF16 = 16;
void set_build_f16(bool on);      // with BUILD_BF16: the 16-bit copies are fp16 (compute mode BRN_F16)
constexpr int BUILD_HALF2 = 18;   // two fp16 planes of the scaled matrix (mode f32_half2)
float half2_act_scale();          // the power of two GEMM activations are scaled by before the fp16 split (BRN_H2_ASCALE, default 8)
void set_build_planes(int planes);
int build_planes();

}  // namespace brn
