// Library identification + entry points that are declared in the ABI but not built yet in this round
// (they return VIPE_EUNSUPPORTED so the Python side raises NotImplementedError - never a silent fallback).
#include "common.cuh"

VIPE_EXPORT const char* vipe_amd_version(void) { return "vipe_amd 0.1 (gfx950, hipcc)"; }
VIPE_EXPORT int vipe_amd_abi_version(void) { return 1; }

VIPE_EXPORT int vipe_corr_pyramid_build(const void*, const void*, void* const*, int, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_altcorr_forward(const void*, const void*, const float*, void*, int, int, int, int, int, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_altcorr_backward(const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int64_t vipe_ba_workspace_bytes(int, int, int, int) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_ba(float*, float*, const float*, const float*, const float*, const float*, const float*, const int64_t*, const int64_t*, int, int, int, int, int, int, int, int, float, float, int, float*, float*, void*, int64_t, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_frame_distance(const float*, const float*, const float*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, const int64_t*, float*, int, int, int, float, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_depth_filter(const float*, const float*, const float*, const int64_t*, const float*, float*, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_projmap(const float*, const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_iproj(const float*, const float*, const float*, float*, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_scatter(const void*, const int64_t*, void*, int64_t*, int64_t, int64_t, int64_t, int64_t, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_scatter_mean_rows_f16(const void*, const int64_t*, void*, int, int, int64_t, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_corr_sampler_forward(const void*, const void*, void*, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_corr_sampler_backward(const void*, const void*, const void*, void*, void*, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, void*) { return VIPE_EUNSUPPORTED; }
