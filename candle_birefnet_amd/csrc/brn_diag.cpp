// brn_diag.cpp — brn_gemm_microbench (include/birefnet_hip_diag.h): compiled only into libbirefnet_hip_diag.so
// (make diag, -DBRN_DIAG_BUILD); the product library carries neither this entry nor the probe kernels it drives.
#include "brn_host.h"
#include "../../include/birefnet_hip_diag.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace brn {
template <class F>
static brn_status guarded_diag(F&& f) {
    try { f(); return BRN_OK; }
    catch (const Error& e) { set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { set_last_error(e.what()); return BRN_ERR_INVALID_ARG; }
}
}  // namespace brn
using namespace brn;
#define guarded guarded_diag

extern "C" {

// ---- diagnostics -----------------------------------------------------------------------------------------------------
brn_status brn_gemm_microbench(int M, int N, int K, int tile_cfg, int splitk, int iters, int device, float* ms_per_launch) {
    return guarded([&] {
        if (tile_cfg >= 130 && tile_cfg < 140) {   // MFMA / VALU SIMD-sharing probe, mode = tile_cfg - 130; returns us per launch
            ensure_device(device);
            DeviceOwner own;
            std::vector<float> z(16, 0.f);
            float* sink = own.upload(z);
            hipEvent_t e0, e1;
            BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
            BRN_HIP(launch_mfma_valu_probe(M, N, tile_cfg - 130, sink, nullptr));
            BRN_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) BRN_HIP(launch_mfma_valu_probe(M, N, tile_cfg - 130, sink, nullptr));
            BRN_HIP(hipEventRecord(e1, nullptr));
            BRN_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
            ms_per_launch[0] = ms / iters * 1e3f;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return;
        }
        if (tile_cfg >= 110 && tile_cfg < 130) {   // LDS-fragment + MFMA consumer-loop probe: 11x = variant 0, 12x = variant 1, x = planes
            ensure_device(device);
            DeviceOwner own;
            std::vector<float> z(16, 0.f);
            float* sink = own.upload(z);
            const int np = tile_cfg % 10, variant = tile_cfg >= 120;
            hipEvent_t e0, e1;
            BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
            BRN_HIP(launch_lds_mfma_probe(M, N, np, variant, sink, nullptr));
            BRN_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) BRN_HIP(launch_lds_mfma_probe(M, N, np, variant, sink, nullptr));
            BRN_HIP(hipEventRecord(e1, nullptr));
            BRN_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
            const int npair = np * (np + 1) / 2;
            const double mfmas = (double)M * 4 * (double)N * 2 * 4 * npair;      // wgs * waves * iters * ksteps * tiles * pairs
            ms_per_launch[0] = (float)(mfmas * 32768.0 / (ms / iters * 1e-3) / 1e12);   // bf16 TF/s
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return;
        }
        if (tile_cfg == 101 || tile_cfg == 102) {   // bf16 32x32x16 MFMA peak probe (101: 4 accumulators, 102: one dependent chain)
            ensure_device(device);
            DeviceOwner own;
            std::vector<float> z(16, 0.f);
            float* sink = own.upload(z);
            unsigned long long* clk = (unsigned long long*)own.upload(z);
            hipEvent_t e0, e1;
            BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
            const int nacc = tile_cfg == 101 ? 4 : 1;
            BRN_HIP(launch_mfma_peak_bf16(M, N, sink, clk, nacc, nullptr));
            BRN_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) BRN_HIP(launch_mfma_peak_bf16(M, N, sink, clk, nacc, nullptr));
            BRN_HIP(hipEventRecord(e1, nullptr));
            BRN_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[2];
            BRN_HIP(hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost));
            const double flop = (double)M * 4 * (double)N * 4 * 32768.0;
            ms_per_launch[0] = (float)(flop / (ms / iters * 1e-3) / 1e12);
            if (K > 1) ms_per_launch[1] = hc[1] ? (float)((double)hc[0] / (double)hc[1] * 100.0) : 0.f;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return;
        }
        if (tile_cfg == 100) {   // MFMA peak probe: M = workgroups, N = MFMA iterations (x4) per wave; returns TF/s, clock via splitk ptr hack-free
            ensure_device(device);
            DeviceOwner own;
            std::vector<float> z(16, 0.f);
            float* sink = own.upload(z);
            unsigned long long* clk = (unsigned long long*)own.upload(z);
            hipEvent_t e0, e1;
            BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
            BRN_HIP(launch_mfma_peak(M, N, sink, clk, nullptr));
            BRN_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) BRN_HIP(launch_mfma_peak(M, N, sink, clk, nullptr));
            BRN_HIP(hipEventRecord(e1, nullptr));
            BRN_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[2];
            BRN_HIP(hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost));
            const double flop = (double)M * 4 /*waves*/ * (double)N * 4 * 4096.0;
            ms_per_launch[0] = (float)(flop / (ms / iters * 1e-3) / 1e12);                    // TF/s
            if (K > 1) ms_per_launch[1] = hc[1] ? (float)((double)hc[0] / (double)hc[1] * 100.0) : 0.f;   // MHz (K>1: 2 floats out)
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            return;
        }
        if (M < 1 || N < 1 || K < 32 || K % 32 || iters < 1 || !ms_per_launch) fail(BRN_ERR_INVALID_ARG, "bad argument");
        ensure_device(device);
        DeviceOwner own;
        std::vector<float> ha((size_t)M * K), hw((size_t)((N + 127) / 128 * 128) * K, 0.f);
        uint32_t s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : ha) v = rnd();
        for (size_t i = 0; i < (size_t)N * K; ++i) hw[i] = rnd();
        int planes = 0;
        if (tile_cfg >= 1000) { planes = tile_cfg / 1000; tile_cfg %= 1000; if (tile_cfg == 999) tile_cfg = -1; }
        if (planes == 4) {        // 4000 + cfg (4999 = the library's plan): the bf16-storage kernel (kernels/gemm_bf16.hip), bf16 A and C
            auto bf = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(r >> 16); };
            std::vector<uint16_t> hab((size_t)M * K + 64);
            for (size_t i = 0; i < (size_t)M * K; ++i) hab[i] = bf(ha[i]);
            void* dAb = nullptr;
            BRN_HIP(hipMalloc(&dAb, hab.size() * 2));
            own.ptrs.push_back(dAb);
            BRN_HIP(hipMemcpy(dAb, hab.data(), hab.size() * 2, hipMemcpyHostToDevice));
            set_build_planes(BUILD_BF16);
            GemmW gw = make_linear(own, hw.data(), nullptr, N, K);
            set_build_planes(0);
            const bool epi1 = getenv("BRN_GEMM_EPI1") != nullptr;      // fp32 C + fp32 residual in place (proj / fc2) instead of bf16 C
            std::vector<float> hc(epi1 ? (size_t)M * N : (size_t)M * N / 2 + 64, 0.f);
            float* dC = own.upload(hc);
            GemmPlan pl = plan_gemm_bf16(M, N, K, epi1, getenv("BRN_GEMM_ACT") && atoi(getenv("BRN_GEMM_ACT")) == 2);
            if (tile_cfg >= 0) { pl.cfg = tile_cfg; pl.splitk = splitk > 1 ? splitk : 1; pl.ws_floats = pl.splitk > 1 ? (size_t)pl.splitk * M * N : 0; }
            float* ws = nullptr;
            if (pl.ws_floats) { std::vector<float> z(pl.ws_floats, 0.f); ws = own.upload(z); }
            GemmParams p{};
            p.A = (const float*)dAb; p.C = dC; p.M = M; p.N = N; p.K = K; p.mode = GEMM_DENSE; p.lda = K; p.ldc = N; p.bbias_rows = 1;
            p.Wp = gw.wb; p.wp_rows = gw.wb_rows; p.wp_ld = gw.wb_ld; p.planes = 1;
            if (epi1) { p.c_f32 = 1; p.R = dC; p.r_f32 = 1; p.ldr = N; }
            if (const char* ab = getenv("BRN_GEMM_ABLATE")) p.abl = atoi(ab);
            std::vector<float> hb((size_t)N, 0.1f);
            if (const char* ac = getenv("BRN_GEMM_ACT")) { p.act = atoi(ac); p.bias = own.upload(hb); }
            hipEvent_t e0, e1;
            BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) BRN_HIP(launch_gemm_bf16(p, pl, ws, nullptr));
            BRN_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < iters; ++i) BRN_HIP(launch_gemm_bf16(p, pl, ws, nullptr));
            BRN_HIP(hipEventRecord(e1, nullptr));
            BRN_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            *ms_per_launch = ms / iters;
            return;
        }
        float* dA = own.upload(ha);
        set_build_planes(planes);
        GemmW gw = make_linear(own, hw.data(), nullptr, N, K);
        set_build_planes(0);
        float* dW = gw.w;
        std::vector<float> hc((size_t)M * N, 0.f);
        float* dC = own.upload(hc);
        GemmPlan pl = plan_gemm(M, N, K, gw.wp ? gw.planes : 0);
        bool a_p2 = false;
        if (tile_cfg == 9) { tile_cfg = 0; a_p2 = true; }     // 2009: warp-specialised kernel fed an A that is already in the P2 layout
        if (tile_cfg >= 0) { pl.cfg = tile_cfg; pl.splitk = splitk > 1 ? splitk : 1; pl.ws_floats = pl.splitk > 1 ? (size_t)pl.splitk * M * N : 0; }
        if (a_p2) {
            if (planes != 2 || K % 32) fail(BRN_ERR_INVALID_ARG, "P2 input needs the 2-plane mode");
            auto bf = [](float x) { uint32_t u; std::memcpy(&u, &x, 4); const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(r >> 16); };
            auto fl = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
            std::vector<float> p2(ha.size());
            uint16_t* q = reinterpret_cast<uint16_t*>(p2.data());
            for (int m = 0; m < M; ++m)
                for (int k = 0; k < K; ++k) {
                    const float x = ha[(size_t)m * K + k];
                    const uint16_t h = bf(x), l = bf(x - fl(h));
                    uint16_t* row = q + (size_t)m * K * 2;
                    row[(k / 32) * 64 + (k % 32)] = h;
                    row[(k / 32) * 64 + 32 + (k % 32)] = l;
                }
            BRN_HIP(hipMemcpy(dA, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
        }
        float* ws = nullptr;
        if (pl.ws_floats) { std::vector<float> z(pl.ws_floats, 0.f); ws = own.upload(z); }
        GemmParams p{};
        p.A = dA; p.W = dW; p.C = dC; p.M = M; p.N = N; p.K = K; p.mode = GEMM_DENSE; p.lda = K; p.ldc = N; p.bbias_rows = 1;
        p.Wp = gw.wp; p.planes = gw.planes; p.wp_rows = gw.wp_rows;
        p.a_planes = a_p2 ? 2 : 0;
        if (const char* ab = getenv("BRN_GEMM_ABLATE")) p.abl = atoi(ab);
        std::vector<float> hb((size_t)N, 0.1f);
        if (const char* ac = getenv("BRN_GEMM_ACT")) { p.act = atoi(ac); p.bias = own.upload(hb); }          // epilogue cost probes
        if (getenv("BRN_GEMM_RES")) { p.R = dC; p.ldr = N; }
        hipEvent_t e0, e1;
        BRN_HIP(hipEventCreate(&e0)); BRN_HIP(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) BRN_HIP(launch_gemm(p, pl, ws, nullptr));
        BRN_HIP(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) BRN_HIP(launch_gemm(p, pl, ws, nullptr));
        BRN_HIP(hipEventRecord(e1, nullptr));
        BRN_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        BRN_HIP(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *ms_per_launch = ms / iters;
        if (const char* tp = getenv("BRN_GEMM_TRACE")) {   // one more launch with per-workgroup cycle stamps, dumped raw
            const size_t nwg = (size_t)((M + 127) / 128) * ((N + 127) / 128) * (size_t)std::max(1, pl.splitk);
            std::vector<unsigned long long> tr(nwg * 256, 0ull);
            unsigned long long* dtr = nullptr;
            BRN_HIP(hipMalloc(&dtr, tr.size() * 8));
            BRN_HIP(hipMemset(dtr, 0, tr.size() * 8));
            p.trace = dtr;
            BRN_HIP(launch_gemm(p, pl, ws, nullptr));
            BRN_HIP(hipDeviceSynchronize());
            BRN_HIP(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
            (void)hipFree(dtr);
            if (FILE* f = fopen(tp, "wb")) { fwrite(tr.data(), 8, tr.size(), f); fclose(f); }
        }
    });
}


}  // extern "C"
