/*
 * birefnet_hip_diag.h — tuning / diagnostic entry points.  NOT part of the product library: they exist only in
 * candle_birefnet_amd/libbirefnet_hip_diag.so (`make -C candle_birefnet_amd/csrc diag`), a build of the same sources with
 * -DBRN_DIAG_BUILD that additionally carries the probe kernels (MFMA peak, LDS-fragment + MFMA loop, MFMA / VALU SIMD sharing)
 * and the traced / ablated variants of the warp-specialised GEMM.  tools/ *.py load it through BRN_LIB_PATH.
 */
#ifndef BIREFNET_HIP_DIAG_H
#define BIREFNET_HIP_DIAG_H
#include "birefnet_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Times `iters` launches of the dense GEMM kernels on random device data (M x K times N x K^T).  tile_cfg: -1 = the
 * library's own plan, 0 = 128x128, 1 = 128x64, 2 = 64x64 block tile; splitk only with tile_cfg >= 0; add 1000*planes
 * (planes 1..3) for the split-bf16 kernels (1999/2999/3999 = library plan with 1/2/3 planes), 4000 + cfg for the bf16-storage
 * kernel; 100..139 select the probe kernels (see brn_diag.cpp). */
brn_status brn_gemm_microbench(int M, int N, int K, int tile_cfg, int splitk, int iters, int device_ordinal,
                               float* ms_per_launch);
#ifdef __cplusplus
}
#endif
#endif
