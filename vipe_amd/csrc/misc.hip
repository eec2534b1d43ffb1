// Library identification.
#include "common.cuh"

VIPE_EXPORT const char* vipe_amd_version(void) { return "vipe_amd 0.1 (gfx950, hipcc)"; }
VIPE_EXPORT int vipe_amd_abi_version(void) { return 4; }  // 4: n_prepared, slot-indexed operator state; 3: vipe_ba_params.solver_options (2: overlap_*, vipe_update_buffers.{side_stream,pzr,gate_state})

