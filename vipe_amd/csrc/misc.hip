// Library identification + entry points that are declared in the ABI but not built yet in this round
// (they return VIPE_EUNSUPPORTED so the Python side raises NotImplementedError - never a silent fallback).
#include "common.cuh"

VIPE_EXPORT const char* vipe_amd_version(void) { return "vipe_amd 0.1 (gfx950, hipcc)"; }
VIPE_EXPORT int vipe_amd_abi_version(void) { return 1; }

VIPE_EXPORT int64_t vipe_ba_workspace_bytes(int, int, int, int) { return VIPE_EUNSUPPORTED; }
VIPE_EXPORT int vipe_ba(float*, float*, const float*, const float*, const float*, const float*, const float*, const int64_t*, const int64_t*, int, int, int, int, int, int, int, int, float, float, int, float*, float*, void*, int64_t, void*) { return VIPE_EUNSUPPORTED; }
