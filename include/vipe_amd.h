/*
 * vipe_amd.h - C ABI of libvipe_amd.so, the MI355X (gfx950) backend for the ViPE dense-SLAM
 * update iteration.  One entry point per function the reference binds through pybind11 in
 * csrc/bind.cpp:28-49 (submodules droid_net_ext, slam_ext, lietorch_ext, scatter_ext, corr_ext),
 * plus the fused entry points the MI355X-first design adds (marked [fused]).
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (hipMalloc / torch CUDA tensor .data_ptr());
 *     h_* is host memory.  All tensors are contiguous row-major with the shapes given.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream; 0 = null stream).
 *     Nothing here synchronises the host, allocates device memory or copies D2H: callers pass
 *     workspaces (sizes from the *_workspace_bytes functions), so every call is graph-capturable.
 *   - return value: 0 on success, a negative VIPE_E* code on argument errors (no launch made),
 *     or a positive hipError_t if a launch failed.
 *   - dtype codes: VIPE_F16 / VIPE_F32 / VIPE_F64 (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF).
 */
#ifndef VIPE_AMD_H
#define VIPE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIPE_F16 0
#define VIPE_F32 1
#define VIPE_F64 2

#define VIPE_OK 0
#define VIPE_EINVAL (-1)   /* bad shape / null pointer / unsupported dtype */
#define VIPE_ENOSPACE (-2) /* workspace too small */
#define VIPE_EUNSUPPORTED (-3)

/* storage layout of a correlation pyramid (level i of an edge b, source pixel p1 = y1 * w + x1, target (y, x)):
 *   REFERENCE  level i = [B][h*w][h>>i][w>>i] for every level - CorrBlock.corr_pyramid, droid_net.py:56-69.
 *   BLOCKED    levels 0 and 1 regrouped into 64-byte tiles of 4 rows x 8 columns and 16 KiB / 8 KiB runs per 64
 *              source pixels, on the grid PADDED to G = ceil(h w / 64) groups of source pixels, S = ceil(w / 32) strips
 *              of 32 target columns and R = 2 ceil(h / 8) groups of 4 target rows (vipe_corr_blocked_dims):
 *                level0[b][p1/64][(x/32)*R     + y/4][p1%64][(x%32)/8][y%4][x%8]
 *                level1[b][p1/64][(x/16)*(R/2) + y/4][p1%64][(x%16)/8][y%4][x%8]     (x, y level-1 coordinates)
 *                level2[b][G*64][R][8 S]      level3[b][G*64][R/2][round_up(4 S, 8)]  (one slab per source pixel)
 *              For w % 64 == 0, h % 8 == 0 (the DROID maps at 1/8 of 512 x 384 and multiples) nothing is padded: the
 *              same bytes per level as REFERENCE, levels 2 / 3 ARE the reference's.  For every other grid (41 x 73 for
 *              16:9 video, vipe/slam/system.py:46-59) tiles wholly outside the h x w targets are never written or read,
 *              and entries outside the floored (h >> i) x (w >> i) level (droid_net.py:66-68) hold zero.  The internal
 *              layout of pooled pyramids: built by vipe_corr_pyramid_build_indexed / _prepared, read by
 *              vipe_corr_lookup_conv1x1. */
#define VIPE_PYRAMID_REFERENCE 0
#define VIPE_PYRAMID_BLOCKED 1

#define VIPE_CAM_PINHOLE 0 /* vipe/utils/cameras.py:123 */
#define VIPE_CAM_MEI 1     /* vipe/utils/cameras.py:220 */

/* library / build identification ("gfx950", ABI version) */
const char* vipe_amd_version(void);
int vipe_amd_abi_version(void); /* 4 since round 4 (vipe_corr_pyramid_build_prepared.n_prepared, slot-indexed operator state); 3: vipe_ba_params.solver_options; rounds 1-2: 1, 2 */

/* ---------------------------------------------------------------------------------------------
 * droid_net_ext  (csrc/droid_net_ext/droid.cpp:57-63)
 * ------------------------------------------------------------------------------------------- */

/* corr_index_forward: replaces corr_index_cuda_forward, correlation_kernels.cu:115-136.
 * volume [B,h1,w1,h2,w2] dtype; coords [B,2,h1,w1] f32; corr [B,2r+1,2r+1,h1,w1] dtype (fully written). */
int vipe_corr_index_forward(const void* d_volume, const float* d_coords, void* d_corr, int B, int h1, int w1,
                            int h2, int w2, int radius, int dtype, void* stream);

/* corr_index_backward: replaces corr_index_cuda_backward, correlation_kernels.cu:138-159.
 * corr_grad [B,2r+1,2r+1,h1,w1]; volume_grad [B,h1,w1,h2,w2] (fully written, zero outside the windows). */
int vipe_corr_index_backward(const float* d_coords, const void* d_corr_grad, void* d_volume_grad, int B, int h1,
                             int w1, int h2, int w2, int radius, int dtype, void* stream);

/* [fused] CorrBlock.__call__ (droid_net.py:71-82): all pyramid levels in one launch.
 * h_levels: host array of num_levels device pointers, level i is [B,h1,w1,h2>>i,w2>>i];
 * coords [B,h1,w1,2] f32 (level i uses coords / 2^i); out [B,num_levels*(2r+1)^2,h1,w1]. */
int vipe_corr_pyramid_lookup(const void* const* h_levels, const float* d_coords, void* d_out, int B, int h1, int w1,
                             int h2, int w2, int num_levels, int radius, int dtype, void* stream);

/* [fused] same lookup written channels-last for the flow-update operator: out [B,h1,w1,channel_stride],
 * channel = level*(2r+1)^2 + x_off*(2r+1) + y_off, channels beyond num_levels*(2r+1)^2 zero filled. */
int vipe_corr_pyramid_lookup_nhwc(const void* const* h_levels, const float* d_coords, void* d_out, int B, int h1,
                                  int w1, int h2, int w2, int num_levels, int radius, int dtype, int channel_stride,
                                  void* stream);

/* [fused] 4-level radius-3 lookup (droid_net.py:71-82) + the correlation encoder's first layer, Conv2d(196, Cout, 1)
 * + bias + activation (droid_net.py:436-437, 481): out[B,h1,w1,out_ctot] channels [out_coff, out_coff+Cout) fp16.
 * d_w_packed / d_bias as produced by vipe_conv_pack_weights for a [Cout, 200, 1, 1] weight (196 + 4 zero inputs).
 * fp16 volume levels, Cout == 128; other configurations return VIPE_EUNSUPPORTED (use lookup_nhwc + conv2d).
 * d_slots (optional, [B] int32): edge b reads slot d_slots[b] of the level buffers (a pool of pyramids with capacity
 * >= B whose edges come and go without compaction, factor_graph.py:147-152,194-196); null: slot b.
 * layout: VIPE_PYRAMID_REFERENCE (level widths multiples of 8 down to level 3) or VIPE_PYRAMID_BLOCKED (h1 == h2,
 * w1 == w2; any grid with h, w >= 8). */
int vipe_corr_lookup_conv1x1(const void* const* h_levels, const float* d_coords, const void* d_w_packed,
                             const float* d_bias, void* d_out, int out_ctot, int out_coff, int B, int h1, int w1,
                             int h2, int w2, int Cout, int act, const int* d_slots, int layout, void* stream);

/* [fused] CorrBlock.corr + pyramid (droid_net.py:56-69,94-102): volume = (f1/4)^T (f2/4) on MFMA (fp16 in,
 * fp32 accumulate, stored as dtype), then 2x2 average pooling of the target dims for levels 1..num_levels-1.
 * fmap1, fmap2 [B,C,h,w] f16; h_levels: host array of num_levels device pointers (outputs). C % 32 == 0. */
int vipe_corr_pyramid_build(const void* d_fmap1, const void* d_fmap2, void* const* h_levels, int B, int C, int h,
                            int w, int num_levels, void* stream);

/* [fused] the same for the edges of a factor graph, straight from the keyframe buffer into a pooled store
 * (factor_graph.py:147-152: `CorrBlock(fmaps[ii], fmaps[jj])` then `corr.cat`): edge b correlates frame d_idx1[b]
 * with frame d_idx2[b] of d_fmaps [n_frames,C,h,w] f16 - the gathered [B,C,h,w] copies never exist - and is written
 * to slot d_slots[b] of the level buffers (null: slot b) in `layout` (VIPE_PYRAMID_*). */
int vipe_corr_pyramid_build_indexed(const void* d_fmaps, const int64_t* d_idx1, const int64_t* d_idx2,
                                    const int* d_slots, void* const* h_levels, int B, int C, int h, int w,
                                    int num_levels, int layout, void* stream);

/* [fused] the same for ANY grid (C == 128): the keyframes' maps are first rewritten, once per frame, into the zero-padded
 * operand images the MFMA kernel loads with aligned 16-byte accesses (an h x w map with an odd pixel count has no
 * aligned rows):  vipe_corr_prep(d_fmaps [n,C,h,w] f16 -> d_prep [n][vipe_corr_prep_halves(C,h,w)] f16), then
 * vipe_corr_pyramid_build_prepared with frame indices d_idx1 / d_idx2 counted from `frame_base` (d_prep holds frames
 * frame_base .. frame_base + n_prepared - 1 of the keyframe buffer; an edge one of whose frames lies outside that range
 * is SKIPPED - its slot keeps its old contents - instead of reading outside d_prep).  Output: VIPE_PYRAMID_BLOCKED on the padded grid.
 * vipe_corr_blocked_dims: dims6 = {G, S, R, level-2 row pitch, level-3 row pitch, 1 if vipe_corr_pyramid_build_indexed
 * tiles this grid directly (no preparation needed) else 0}. */
int vipe_corr_blocked_dims(int h, int w, int* dims6);
int64_t vipe_corr_prep_halves(int C, int h, int w);
int vipe_corr_prep(const void* d_fmaps, void* d_prep, int n, int C, int h, int w, void* stream);
int vipe_corr_pyramid_build_prepared(const void* d_prep, int64_t frame_base, int64_t n_prepared, const int64_t* d_idx1, const int64_t* d_idx2,
                                     const int* d_slots, void* const* h_levels, int B, int C, int h, int w, int num_levels,
                                     void* stream);

/* CorrBlock.corr for any dtype / channel count (droid_net.py:94-102): volume[b][p1][p2] = sum_c (f1[b][c][p1] / 4)
 * (f2[b][c][p2] / 4); fmaps [B,C,P] dtype, volume [B,P,P] dtype (plain tiled kernel: what the MFMA kernel of
 * vipe_corr_pyramid_build does not take).  vipe_avg_pool2x2: F.avg_pool2d(x, 2, stride=2) over the last two dims of
 * [n,h,w] -> [n,h>>1,w>>1] (droid_net.py:66-68), at::native rounding. */
int vipe_corr_volume(const void* d_fmap1, const void* d_fmap2, void* d_volume, int B, int C, int P, int dtype, void* stream);
int vipe_avg_pool2x2(const void* d_x, void* d_y, int64_t n, int h, int w, int dtype, void* stream);

/* altcorr_forward: replaces altcorr_cuda_forward, altcorr_kernel.cu:266-290.
 * fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] dtype (f16/f32); coords [B,N,H1,W1,2] f32; corr [B,N,(2r+1)^2,H1,W1]. */
int vipe_altcorr_forward(const void* d_fmap1, const void* d_fmap2, const float* d_coords, void* d_corr, int B,
                         int H1, int W1, int H2, int W2, int N, int C, int radius, int dtype, void* stream);

/* altcorr_backward: replaces altcorr_cuda_backward, altcorr_kernel.cu:292-320 (float32 only).
 * fmap1_grad / fmap2_grad are accumulated into (caller zero-initialises), coords receive no gradient. */
int vipe_altcorr_backward(const float* d_fmap1, const float* d_fmap2, const float* d_coords,
                          const float* d_corr_grad, float* d_fmap1_grad, float* d_fmap2_grad, int B, int H1, int W1,
                          int H2, int W2, int N, int C, int radius, void* stream);

/* ---------------------------------------------------------------------------------------------
 * lietorch_ext  (csrc/lietorch_ext/lietorch.cpp:305-336).  group_id 1=SO3 2=RxSO3 3=SE3 4=Sim3
 * (dispatch.h:18-40); dtype VIPE_F32 / VIPE_F64.  `on_device` 0 runs the same closed forms on the
 * host (the reference's lietorch_cpu.cpp path), 1 launches on `stream`.
 * Rows: X [n,N] (N = 4,5,7,8), tangent a [n,K] (K = 3,4,6,7).
 * ------------------------------------------------------------------------------------------- */
int vipe_lie_expm(int group_id, const void* a, void* X, int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_logm(int group_id, const void* X, void* a, int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_inv(int group_id, const void* X, void* Y, int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_mul(int group_id, const void* X, const void* Y, void* Z, int64_t n, int dtype, int on_device,
                 void* stream);
int vipe_lie_adj(int group_id, const void* X, const void* a, void* b, int64_t n, int dtype, int on_device,
                 void* stream);
int vipe_lie_adjT(int group_id, const void* X, const void* a, void* b, int64_t n, int dtype, int on_device,
                  void* stream);
int vipe_lie_act(int group_id, const void* X, const void* p, void* q, int64_t n, int dtype, int on_device,
                 void* stream); /* p,q [n,3] */
int vipe_lie_act4(int group_id, const void* X, const void* p, void* q, int64_t n, int dtype, int on_device,
                  void* stream); /* p,q [n,4] */
int vipe_lie_as_matrix(int group_id, const void* X, void* T, int64_t n, int dtype, int on_device,
                       void* stream); /* T [n,4,4] */
int vipe_lie_projector(int group_id, const void* X, void* P, int64_t n, int dtype, int on_device,
                       void* stream); /* P [n,N,N] */
int vipe_lie_jinv(int group_id, const void* X, const void* a, void* b, int64_t n, int dtype, int on_device,
                  void* stream);
/* [fused] the keyframe frontend's preparation of the NEXT frame's slot (vipe/slam/components/frontend.py:70-76 `_init_pose`,
 * :118-122 / :147-151): poses[t1] = Exp(0.5 Log(G_{t1-1} G_{t1-2}^-1)) G_{t1-1} when init_pose, and disps[t1, v] =
 * mean(disps[t1 - n_mean .. t1 - 1, v]) for every view.  poses [>= t1+1, 7], disps [>= t1+1, n_views, P] f32. */
int vipe_frontend_next_frame(float* d_poses, float* d_disps, int t1, int n_views, int P, int n_mean, int init_pose,
                             void* stream);
/* [fused] broadcast forms: one group element per `rows_per_elem` consecutive rows of a/p (the Python
 * wrapper in the reference replicates X per row, broadcasting.py:32-35). */
int vipe_lie_adjT_bcast(int group_id, const void* X, const void* a, void* b, int64_t n_elem, int64_t rows_per_elem,
                        int dtype, void* stream);
int vipe_lie_act4_bcast(int group_id, const void* X, const void* p, void* q, int64_t n_elem, int64_t rows_per_elem,
                        int dtype, void* stream);
/* backward passes (lietorch.cpp:317-331): grad wrt the op's inputs */
int vipe_lie_expm_backward(int group_id, const void* grad, const void* a, void* da, int64_t n, int dtype,
                           int on_device, void* stream);
int vipe_lie_logm_backward(int group_id, const void* grad, const void* X, void* dX, int64_t n, int dtype,
                           int on_device, void* stream);
int vipe_lie_inv_backward(int group_id, const void* grad, const void* X, void* dX, int64_t n, int dtype,
                          int on_device, void* stream);
int vipe_lie_mul_backward(int group_id, const void* grad, const void* X, const void* Y, void* dX, void* dY,
                          int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_adj_backward(int group_id, const void* grad, const void* X, const void* a, void* dX, void* da,
                          int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_adjT_backward(int group_id, const void* grad, const void* X, const void* a, void* dX, void* da,
                           int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_act_backward(int group_id, const void* grad, const void* X, const void* p, void* dX, void* dp,
                          int64_t n, int dtype, int on_device, void* stream);
int vipe_lie_act4_backward(int group_id, const void* grad, const void* X, const void* p, void* dX, void* dp,
                           int64_t n, int dtype, int on_device, void* stream);

/* ---------------------------------------------------------------------------------------------
 * slam_ext  (csrc/slam_ext/slam.cpp:31-37) and the live dense BA it stands behind
 * ------------------------------------------------------------------------------------------- */

/* [fused] GraphBuffer.reproject_dense_disp (buffer.py:527-548 -> geom.py:187-263, jacobian=False).
 * poses [n_poses,7] world->cam; disps [n_poses*V,ht,wd]; intrinsics [V,4+D] FULL-RES (scaled by
 * 1/intr_factor inside, terms.py:182); rig [V,7]; pi,qi,pj,qj,di [M] int64 (expand_edge_multiview).
 * coords [M,ht,wd,2], valid [M,ht,wd] f32 (valid may be NULL). */
int vipe_reproject(const float* d_poses, const float* d_disps, const float* d_intrinsics, const float* d_rig,
                   const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj, const int64_t* d_qj,
                   const int64_t* d_di, float* d_coords, float* d_valid, int M, int ht, int wd, int n_views,
                   int camera, float intr_factor, void* stream);

/* [fused] FactorGraph.update motion features (factor_graph.py:253-261): coords1 = reproject(ii,jj) and
 * motn = clamp(cat[coords1 - grid, target - coords1], +-64) written as [M,4,ht,wd] in `motn_dtype`. */
int vipe_reproject_motion(const float* d_poses, const float* d_disps, const float* d_intrinsics, const float* d_rig,
                          const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj, const int64_t* d_qj,
                          const int64_t* d_di, const float* d_target, float* d_coords, void* d_motn, int M, int ht,
                          int wd, int n_views, int camera, float intr_factor, int motn_dtype, void* stream);
/* same with motn written channels-last [M,ht,wd,4] fp16 (the layout the MFMA convolutions read) */
int vipe_reproject_motion_nhwc(const float* d_poses, const float* d_disps, const float* d_intrinsics,
                               const float* d_rig, const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj,
                               const int64_t* d_qj, const int64_t* d_di, const float* d_target, float* d_coords,
                               void* d_motn, int M, int ht, int wd, int n_views, int camera, float intr_factor,
                               void* stream);

/* Dense bundle adjustment with the LIVE semantics of GraphBuffer.bundle_adjustment (buffer.py:373-525,
 * solver.py:117-197, terms.py:94-303): Gauss-Newton on SE3 (+) per-pixel inverse depth (+ optional
 * focal/distortion), Schur reduction over the depths, dense block Cholesky (fp64) of the reduced system,
 * back-substitution and retraction, n_iters times, entirely on `stream`.
 *   poses [n_poses,7] (updated in place), disps [n_poses*V,ht,wd] (in place; finally clamped >= 1e-3 over
 *   all n_disp_frames), disps_sens like disps, intrinsics [V,4+D] (in place when optimize_intrinsics),
 *   rig [V,7], target/weight [M,ht*wd,2], disp_damping [n_poses*V,ht,wd] (GraphAgg eta),
 *   pi,qi,pj,qj,di [M] int64.  Pose p is fixed iff it occurs in pi and (p < t0 or p >= t1)
 *   (buffer.py:462-465); t0 == t1 fixes every pose.
 *   d_workspace: vipe_dense_ba_workspace_bytes(...) bytes, contents need not be initialised.
 *   d_info (optional, 8 ints): [n_free_poses, n_free_disp_frames, cholesky_failures, n_regular_unknowns,
 *   band width in 6x6 blocks, 1 if the LDS band solver solved the last iteration, largest source-frame degree, 0]. */
/* Co-scheduling hook of vipe_dense_ba (optional).  Most of a Gauss-Newton iteration is the reduced-system solve: ONE
 * workgroup busy on a 256-CU part.  A caller with independent work (FactorGraph.update: the hidden-state part of the next
 * iteration's GRU gates, vipe_update_gate_state_piece) hands it over in `n_pieces = max(n_iters, 1)` pieces: piece k is
 * enqueued on `overlap_stream` behind an event recorded after iteration k's accumulate kernels and a few microseconds
 * of delay, i.e. it starts once the solve of iteration k owns its CU and fills the other 255.  overlap_fn is called on
 * the host, from inside vipe_dense_ba, exactly n_pieces times (also when the BA has nothing to do); it must only enqueue
 * work on the stream it is given and return VIPE_OK.  The caller orders overlap_stream after its producers before the
 * call and joins it afterwards. */
typedef int (*vipe_overlap_fn)(void* user, int piece, int n_pieces, void* stream);

typedef struct {
  int n_poses;       /* rows of poses; disps has n_poses*n_views frames */
  int n_views;
  int ht, wd;
  int M;             /* number of terms (edges * views) */
  int t0, t1;
  int n_iters;
  float pose_damping, pose_ep;
  int motion_only;
  int limited_disp;
  int optimize_intrinsics;
  int optimize_rig_rotation; /* rotation-only blocks for the views >= 1 of a rig (view 0 is always fixed, buffer.py:497-506,
                                retractor.py:32-37); no effect for n_views == 1.  Rigs of up to 8 views. */
  int camera;        /* VIPE_CAM_* */
  float alpha;       /* ba.dense_disp_alpha, configs/slam/default.yaml:48-49 */
  float weight_scale;/* 0.001, buffer.py:396 */
  float intr_factor; /* 8.0, buffer.py:415 */
  int reuse_plan;    /* 1: the workspace still holds the plan (term order, pose slots, band, flags) of the previous
                        call with IDENTICAL index arrays and parameters - skip rebuilding it.  The caller owns that
                        guarantee (same workspace, nothing else ran in it); everything data dependent (sensor-depth
                        frames, damping) is re-read every call. */
  int path_hint;     /* 0: unknown - every kernel of both accumulate / solve paths is launched and the inapplicable ones
                        exit at once (the choice depends on the plan, which lives on the device).  A caller that has read
                        d_info[4..7] of an earlier call with the SAME plan may pass what it learnt so that those launches are
                        not made: bit 0 source degree <= 6 (matrix-core accumulate), bit 1 degree > 6 (walk + Schur),
                        bit 2 an LDS solver takes the system (the global-memory Cholesky is not launched), bit 3 the LDS
                        band solver does not take it, bit 4 the LDS dense solver does not take it. */
  void* overlap_stream;        /* co-scheduling hook (see vipe_overlap_fn); NULL / NULL: none */
  vipe_overlap_fn overlap_fn;
  void* overlap_user;
  int solver_options;          /* 0 = the default kernel selection.  VIPE_BA_OPT_* bits pick the equivalent general forms, for
                                  validating the specialised kernels against them (tests/test_gpu_parity.py) */
  void* profile_ev0;           /* optional pair of hipEvent_t (created with timing enabled by the caller): recorded on the BA */
  void* profile_ev1;           /* stream right before / right after the accumulate kernels (ba_accum_mfma_kernel, or the walk +
                                  Schur pair) of Gauss-Newton iteration `profile_iter` - bench.py times that launch with them;
                                  NULL: none */
  int profile_iter;            /* 0 .. n_iters - 1; negative: the last iteration */
} vipe_ba_params;
#define VIPE_BA_OPT_ONE_CHAIN 1           /* band solve: eliminate the pose chain from one end (default: both ends at once) */
#define VIPE_BA_OPT_GENERAL_ACCUMULATE 2  /* accumulate: the walk + Schur kernel pair for every source-frame degree */

int64_t vipe_dense_ba_workspace_bytes(const vipe_ba_params* p);
int vipe_dense_ba(const vipe_ba_params* p, float* d_poses, float* d_disps, const float* d_disps_sens,
                  float* d_intrinsics, float* d_rig, const float* d_target, const float* d_weight,
                  const float* d_disp_damping, const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj,
                  const int64_t* d_qj, const int64_t* d_di, void* d_workspace, int64_t workspace_bytes,
                  int* d_info, void* stream);

/* slam_ext.ba with the DROID signature (slam.cpp:31, geom_kernels.cu:1273-1404; dormant in the reference):
 * single pinhole intrinsics[4] at 1/8 scale, targets/weights [E,2,ht,wd], eta [t1-t0... see file], poses/disps
 * updated in place; dx [t1-t0,6], dz [K,ht*wd] written for the last iteration. */
int64_t vipe_ba_workspace_bytes(int n_poses, int ht, int wd, int E);
int vipe_ba(float* d_poses, float* d_disps, const float* d_intrinsics, const float* d_disps_sens,
            const float* d_targets, const float* d_weights, const float* d_eta, const int64_t* d_ii,
            const int64_t* d_jj, int n_poses, int ht, int wd, int E, int n_eta, int t0, int t1, int iterations,
            float lm, float ep, int motion_only, float* d_dx, float* d_dz, void* d_workspace,
            int64_t workspace_bytes, void* stream);

/* frame_distance: replaces frame_distance_cuda, geom_kernels.cu:1406-1434 (kernel :521-676).
 * poses [NV,7], disps [NV,ht,wd], intrinsics [V,4] (pinhole, 1/8 scale), pi,pj,qi,qj,di [M] int64, dist [M]. */
int vipe_frame_distance(const float* d_poses, const float* d_disps, const float* d_intrinsics, const int64_t* d_pi,
                        const int64_t* d_pj, const int64_t* d_qi, const int64_t* d_qj, const int64_t* d_di,
                        float* d_dist, int M, int ht, int wd, float beta, void* stream);

/* [fused] GraphBuffer.frame_distance_dense_disp (vipe/slam/components/buffer.py:550-593, geom.py:335-343) in one launch:
 * candidate m = (keyframe pi[m], view qi[m]) -> (pj[m], qj[m]); the per-view poses R_q^-1 G_p and the pinhole intrinsics at
 * 1 / intr_factor scale are formed in the kernel (the reference expands the poses of ALL frames and rescales the
 * intrinsics on the host side, then calls frame_distance twice and averages).  poses [N,7], rig [V,7], disps [N*V,ht,wd],
 * intrinsics [V, intr_dim] at full resolution (intr_dim 4 pinhole, 5 MEI), dist [M]; bidirectional: 0.5 (d_ij + d_ji). */
int vipe_frame_distance_rig(const float* d_poses, const float* d_rig, const float* d_disps, const float* d_intrinsics,
                            int intr_dim, float intr_factor, const int64_t* d_pi, const int64_t* d_qi, const int64_t* d_pj,
                            const int64_t* d_qj, float* d_dist, int M, int n_views, int ht, int wd, float beta,
                            int bidirectional, void* stream);

/* depth_filter: replaces depth_filter_cuda, geom_kernels.cu:1462-1486 (kernel :678-793).
 * poses [n,7], disps [n,ht,wd], intrinsics [4], inds [num] int64, thresh [num] f32, counter [num,ht,wd] f32. */
int vipe_depth_filter(const float* d_poses, const float* d_disps, const float* d_intrinsics, const int64_t* d_inds,
                      const float* d_thresh, float* d_counter, int n, int num, int ht, int wd, void* stream);

/* projmap: replaces projmap_cuda, geom_kernels.cu:1436-1460 (kernel :434-519). coords [E,ht,wd,3], valid [E,ht,wd,1]. */
int vipe_projmap(const float* d_poses, const float* d_disps, const float* d_intrinsics, const int64_t* d_ii,
                 const int64_t* d_jj, float* d_coords, float* d_valid, int E, int ht, int wd, void* stream);

/* iproj: replaces iproj_cuda, geom_kernels.cu:1488-1507 (kernel :795-861). points [n,ht,wd,3]. */
int vipe_iproj(const float* d_poses, const float* d_disps, const float* d_intrinsics, float* d_points, int n, int ht,
               int wd, void* stream);

/* ---------------------------------------------------------------------------------------------
 * scatter_ext  (csrc/scatter_ext/scatter.cpp:232-238, cuda/scatter_cuda.cu:57-131)
 * src viewed as [outer, src_dim, inner], out as [outer, out_dim, inner], index broadcast to src's shape
 * (int64, same layout as src).  reduce: 0 sum, 1 mul, 2 mean (sum only - caller divides), 3 min, 4 max.
 * For min/max, d_arg_out [outer,out_dim,inner] int64 receives the arg index (src_dim where none).
 * ------------------------------------------------------------------------------------------- */
int vipe_scatter(const void* d_src, const int64_t* d_index, void* d_out, int64_t* d_arg_out, int64_t outer,
                 int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype, void* stream);
/* the same with index [src_dim]: one slot per ROW of the scattered dimension (what scatter_mean's callers pass,
 * droid_net.py:420-421) - no int64 copy of src's shape.  Rows with a slot outside [0, out_dim) are skipped by both. */
int vipe_scatter_rows(const void* d_src, const int64_t* d_index, void* d_out, int64_t* d_arg_out, int64_t outer,
                      int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype, void* stream);

/* The same reductions over HOST memory (float32 / float64), for CPU tensors handed to the Python-level scatter API
 * (vipe/ext/scatter.py:24-63 accepts them through torch.scatter_add_).  Sequential and deterministic; min / max ties
 * resolve to the last source row.  Index values are range-checked (VIPE_EINVAL). */
int vipe_scatter_host(const void* h_src, const int64_t* h_index, void* h_out, int64_t* h_arg_out, int64_t outer,
                      int64_t src_dim, int64_t inner, int64_t out_dim, int reduce, int dtype);

/* [fused] segmented mean of per-edge feature maps onto source nodes (GraphAgg, droid_net.py:420-421):
 * src [E,inner] f16, ix [E] int64 sorted or not, out [n_out,inner] f16 = mean over edges with ix == k. */
int vipe_scatter_mean_rows_f16(const void* d_src, const int64_t* d_ix, void* d_out, int E, int n_out, int64_t inner,
                               void* stream);

/* [fused] deterministic segmented mean (no atomics): out[k] = mean over q in [rowptr[k], rowptr[k+1]) of
 * src[order[q], :, coff:coff+C]; src [E, rows_per_item, src_ctot] f16, out [n_out, rows_per_item, C] f16. */
int vipe_segment_mean_nhwc_f16(const void* d_src, int src_ctot, int src_coff, const int* d_order, const int* d_rowptr,
                               void* d_out, int n_out, int64_t rows_per_item, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * corr_ext  (csrc/corr_ext/correlation_sampler.cpp:82-85, correlation_cuda_kernel.cu:216-330)
 * in1,in2 [B,C,H,W]; out [B,patchH,patchW,oH,oW]; dtype f16/f32.
 * ------------------------------------------------------------------------------------------- */
int vipe_corr_sampler_forward(const void* d_in1, const void* d_in2, void* d_out, int B, int C, int H, int W, int kH,
                              int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                              int dil_patchH, int dil_patchW, int dH, int dW, int dtype, void* stream);
int vipe_corr_sampler_backward(const void* d_in1, const void* d_in2, const void* d_grad_out, void* d_grad1,
                               void* d_grad2, int B, int C, int H, int W, int kH, int kW, int patchH, int patchW,
                               int padH, int padW, int dilH, int dilW, int dil_patchH, int dil_patchW, int dH,
                               int dW, int dtype, void* stream);

/* The same two operators over HOST memory (float32), for CPU tensors: the reference dispatches those to its CPU
 * implementation (correlation_sampler.cpp:44-58, correlation_cpu.cpp).  grad1 / grad2 are accumulated into. */
int vipe_corr_sampler_forward_host(const float* h_in1, const float* h_in2, float* h_out, int B, int C, int H, int W, int kH,
                                   int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW, int dil_patchH,
                                   int dil_patchW, int dH, int dW);
int vipe_corr_sampler_backward_host(const float* h_in1, const float* h_in2, const float* h_grad_out, float* h_grad1,
                                    float* h_grad2, int B, int C, int H, int W, int kH, int kW, int patchH, int patchW,
                                    int padH, int padW, int dilH, int dilW, int dil_patchH, int dil_patchW, int dH, int dW);

/* ---------------------------------------------------------------------------------------------
 * utils_ext.nearest_neighbours (csrc/utils_ext/knn.cu:27-67, utils_bind.cpp:23-24): exact k nearest neighbours of every
 * query point among the tree points, squared L2 distance, points of 1..3 coordinates (missing ones count as 0, as the
 * reference's zero padding).  query [M,qdim] f32, tree [N,tdim] f32 -> dist [M,knn] f32 ascending, idx [M,knn] int32.
 * Brute force with the tree staged through LDS (the reference builds a kd-tree per call); 1 <= knn <= 8, knn <= N.
 * Equal distances resolve to the lower index (the reference's kd-tree order is unspecified).  Used by
 * SLAMMap.project_map(infill=True) (interface.py:126-139) - outside the update iteration.
 * ------------------------------------------------------------------------------------------- */
int vipe_nearest_neighbours(const float* d_query, int qdim, const float* d_tree, int tdim, int64_t M, int64_t N, int knn,
                            float* d_dist, int* d_idx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * [fused] flow-update operator convolutions (UpdateModule, droid_net.py:432-499): NHWC fp16
 * implicit-GEMM convolution on MFMA, fp32 accumulate, fused bias + activation.
 *   x [B,H,W,Cin_total] f16, reads channels [cin_off, cin_off+Cin); w packed [KH*KW, Cin, Cout] f16
 *   (vipe_conv_pack_weights from the OIHW checkpoint layout); bias [Cout] f32;
 *   y [B,H,W,Cout_total] f16, writes channels [cout_off, cout_off+Cout).
 *   act: 0 none, 1 relu, 2 sigmoid, 3 tanh.  Stride 1, "same" padding.
 *   extra [B,Cout] f32 (optional, NULL = none): per-image additive term (the *_glo 1x1 of the GRU).
 * ------------------------------------------------------------------------------------------- */
#define VIPE_ACT_NONE 0
#define VIPE_ACT_RELU 1
#define VIPE_ACT_SIGMOID 2
#define VIPE_ACT_TANH 3
int vipe_conv_pack_weights(const void* d_w_oihw, void* d_w_packed, int Cout, int Cin, int KH, int KW, int src_dtype,
                           void* stream);
int vipe_conv2d_nhwc_f16(const void* d_x, const void* d_w_packed, const float* d_bias, const float* d_extra,
                         void* d_y, int B, int H, int W, int Cin, int cin_total, int cin_off, int Cout,
                         int cout_total, int cout_off, int KH, int KW, int act, void* stream);

/* padded sizes of the packed weight tensor [k_pad/64][cout_pad][64] (any pointer may be NULL) */
int vipe_conv_packed_dims(int Cout, int Cin, int KH, int KW, int* cout_pad, int* cin_pad, int* k_pad);

/* [fused] the flow-update operator's fused convolutions (droid_net.py:373-499).  Input channels [0,split)
 * come from x0, [split,Cin) from x1 (the concatenations of the reference are never materialised).  mode:
 *   0 plain   y = act(conv + bias + extra[image])
 *   1 GLO     fout[image,c] += sum_pixels sigmoid(conv+bias)[c] * net[c]          (ConvGRU global context, :392-393)
 *   2 ZR      Cout = 256: y[:, c] = z = sigmoid(.), c < 128;  y2[:, c-128] = sigmoid(.) * net   (:395-397)
 *   3 Q       Cout = 128: y = (1 - z) * net + z * tanh(conv + bias + extra)                   (:397-399)
 *   4 HEADS   Cout = 4: fout[pixel] = (delta_x, delta_y, sigmoid(w_x), sigmoid(w_y)) as float (:486-490)
 *   5 ETA     Cout = 1: fout[pixel] = 0.01 * softplus(conv + bias)                            (:410,429)
 *   6 PARTIAL fout[pixel, y_coff + c] (f32, row pitch y_ctot floats) = accinit + conv: the raw fp32 accumulators, no
 *             bias / extra / activation - a partial sum over THESE input channels that a later launch over the other
 *             channels starts from (d_accinit of that launch, mode | VIPE_CONV_ACCINIT_F32).  Cout % 4 == 0; same shape
 *             support as d_accinit.  Used to compute the hidden-state part of the z|r gates of the NEXT update
 *             iteration on a side stream while the dense BA of the current one leaves the chip idle.
 * mode may be OR-ed with VIPE_CONV_ACCINIT_F32: d_accinit is then float32 [B*H*W, ai_ctot] instead of fp16.
 */
#define VIPE_CONV_PARTIAL 6
#define VIPE_CONV_ACCINIT_F32 0x100
/* d_accinit (optional, fp16 [B*H*W, ai_ctot], channels [ai_coff, ai_coff + Cout)): initial value of the accumulators,
 * i.e. a precomputed partial sum over OTHER input channels (convolution is linear in its input channels).  Used to
 * hoist the context-feature part of the GRU gates, constant per edge, out of the update iteration.  Supported for
 * image widths that are multiples of 64 and Cout >= 64; VIPE_EUNSUPPORTED otherwise. */
int vipe_conv2d_fused(const void* d_x0, int x0_ctot, int x0_coff, const void* d_x1, int x1_ctot, int x1_coff,
                      int split, const void* d_w_packed, const float* d_bias, const float* d_extra, int extra_stride,
                      int extra_off, void* d_y, int y_ctot, int y_coff, void* d_y2, int y2_ctot, int y2_coff,
                      const void* d_net, int net_ctot, int net_coff, const void* d_z, float* d_fout,
                      const void* d_accinit, int ai_ctot, int ai_coff, int B, int H, int W, int Cin, int Cout, int KH,
                      int KW, int act, int mode, void* stream);

/* [fused] the whole flow-update operator (UpdateModule.forward, droid_net.py:467-499) sequenced natively: the 13 fused
 * convolutions above + vipe_corr_lookup_conv1x1 + vipe_segment_mean_nhwc_f16 + the pooled-context product in ONE call
 * (issued one by one from the host each launch costs more than most of them run for on a frontend window).
 * All tensors channels-last fp16 unless noted; the weight descriptors are `vipe_conv_pack_weights` outputs + fp32 biases. */
typedef struct {
  const void *corr0_w, *corr2_w, *flow0_w, *flow2_w, *gw_w, *zr_w, *q_w, *zr_s_w, *q_s_w, *heads0_w, *heads2_w, *agg2_w, *eta_w;
  const void *zr_n_w, *zr_x_w; /* z|r gate weights over the hidden state only [256 <- 128] and over (corr | flow) only
                                  [256 <- 192]: the two halves of zr_s_w (vipe_update_gate_state) */
  const float *corr0_b, *corr2_b, *flow0_b, *flow2_b, *gw_b, *zr_b, *q_b, *heads0_b, *heads2_b, *agg2_b, *eta_b;
  const float* glo_wT; /* [128,384] f32: (convz_glo | convr_glo | convq_glo) weights transposed */
  const float* glo_b;  /* [384] */
} vipe_update_weights;
typedef struct {
  int E, H, W, n_src;        /* edges, grid, source nodes (0: no GraphAgg / eta) */
  /* correlation features: either a pyramid to look up (levels[0] != NULL; lookup fused with corr_encoder[0]) ... */
  const void* levels[4];
  const float* coords;       /* [E,H,W,2] */
  const int* slots;          /* optional [E] */
  int h2, w2, pyramid_layout;
  const void* corr;          /* ... or the looked-up features [E,H,W,200] */
  const void* motn;          /* [E,H,W,4] */
  const void* net;           /* [E,H,W,128] hidden state (read) */
  void* net_out;             /* [E,H,W,128] new hidden state (must not alias net: 3x3 halo) */
  void* xbuf;                /* [E,H,W,320]: [inp | corr features | flow features], channels >= 128 overwritten */
  const void* pgate;         /* optional [E,H,W,384]: context-feature part of the gate convolutions (gate-context hoisting) */
  void *c1, *f1, *zb, *rnet; /* scratch [E,H,W,128] */
  void* hbuf;                /* scratch [E,H,W,384] */
  float* dw;                 /* out [E,H,W,4] f32: (delta_x, delta_y, weight_x, weight_y) */
  float *glo, *extra;        /* scratch [E,128], [E,384] f32 */
  const int *order, *rowptr; /* CSR of the edges by source node (vipe_segment_mean_nhwc_f16) */
  void *agg, *a2;            /* scratch [n_src,H,W,128] */
  float* eta;                /* out [n_src,H,W] f32 */
  void* side_stream;         /* optional second stream: the operator's two pairs of independent chains - (lookup + corr
                                encoder) || (flow encoder), and (flow / weight heads) || (GraphAgg mean, agg conv 2, eta) -
                                are issued on `stream` and on this one, forked and joined with events inside the call
                                (a gather-bound kernel next to a matrix-bound one); NULL: everything on `stream` */
  float* pzr;                /* optional [gate_state,H,W,256] f32: hidden-state part of the z|r gates (vipe_update_gate_state) */
  int gate_state;            /* n > 0: `extra` (all edges) and `pzr` (the first n edges) already hold the hidden-state part
                                of the gates for `net` (vipe_update_gate_state ran on it): the operator skips the
                                global-context stage, and the z|r convolution of the first n edges runs over (corr | flow)
                                only, its accumulators starting from pzr; edges [n, E) take the unsplit convolution */
} vipe_update_buffers;
int vipe_update_operator(const vipe_update_weights* weights, const vipe_update_buffers* buffers, void* stream);

/* Everything of the ConvGRU gates that depends on the hidden state ALONE, for hidden state d_net [E,H,W,128]:
 *   extra[E,384] = the three *_glo terms (droid_net.py:392-399; uses buffers->glo as scratch),
 *   pzr[E,H,W,256] (f32) = pgate[..., 0:256] + conv3x3(d_net; W_{z|r}[:, 0:128])   (no bias).
 * Convolution is linear in its input channels, so the next vipe_update_operator call with gate_state = 1 gives the
 * result of the unsplit operator up to fp32 summation order.  The point: d_net is final as soon as the operator has
 * run, while the dense BA that follows it in an update iteration keeps one workgroup busy - issued on a second stream
 * this stage (18 % of the operator's FLOPs) runs in the BA's shadow.  Needs buffers->pgate and buffers->pzr.
 * parts: 1 = the global-context terms (all E edges), 2 = the z|r partial sums (all E edges of the descriptor handed
 * in: a caller that stages only the first n edges passes a descriptor with E = n), 3 = both. */
int vipe_update_gate_state(const vipe_update_weights* weights, const vipe_update_buffers* buffers, const void* d_net,
                           int parts, void* stream);
/* The z|r partial sums (parts = 2) in pieces, with the signature of vipe_overlap_fn: piece k of n covers edges
 * [bounds[k], bounds[k+1]) (bounds: n + 1 ascending ints from 0 to E; NULL: [E k / n, E (k+1) / n)).
 * `user` points to a vipe_gate_state_job. */
typedef struct {
  const vipe_update_weights* weights;
  const vipe_update_buffers* buffers;
  const void* net;
  const int* bounds;
  int n_bounds;      /* entries of bounds (= pieces + 1); a call with another piece count falls back to the even split */
} vipe_gate_state_job;
int vipe_update_gate_state_piece(void* user, int piece, int n_pieces, void* stream);

/* the ConvGRU's three global-context 1x1 convolutions on the pooled vector (droid_net.py:392-399):
 * extra[E,384] = bias + (glo_sum[E,128] / hw) @ wT[128,384]   (all f32) */
int vipe_glo_context(const float* d_glo_sum, const float* d_wT, const float* d_bias, float* d_extra, int E, int hw,
                     void* stream);

/* Per-edge state of a factor graph in stores with spare capacity (vipe_amd/slam/factor_graph.py `_EdgeState`).  The
 * reference concatenates - i.e. copies - every per-edge tensor in `add_factors` (factor_graph.py:147-173) and compacts each
 * with a boolean mask in `rm_factors` (:175-202).  Both are ONE launch here, for all tensors together (at most 8 jobs):
 *   vipe_rows_gather: for every job, dst row (dst_row0 + r) = src row idx[r] (r itself if idx is NULL), r < n_rows; a row
 *     is n_seg segments of seg_bytes bytes every seg_pitch bytes (a channel slice of a channels-last tensor: one segment
 *     per pixel), rows start every src_row_pitch / dst_row_pitch bytes.  Sizes, pitches and addresses multiples of 4 (copied
 *     in the largest of 16 / 8 / 4 bytes they are all multiples of); src != dst.
 *   vipe_gather_nchw_to_nhwc_f16: dst[dst_row0 + r][p][dst_coff + c] = src[frame[r]][c][p] for c < C <= 128, p < P:
 *     `buffer.nets[ii].permute(0, 2, 3, 1)` / `buffer.inps[ii]...` of the new edges' source frames, written straight into
 *     the tail of the channels-last stores (dst rows of dst_row_pitch halves, dst_ctot channels per pixel). */
typedef struct {
  const void* src;
  void* dst;
  const int64_t* idx;
  int64_t src_row_pitch, dst_row_pitch;
  int64_t seg_bytes, seg_pitch;
  int n_seg;
  int n_rows;
  int dst_row0;
} vipe_rows_job;
int vipe_rows_gather(const vipe_rows_job* jobs, int n_jobs, void* stream);
typedef struct {
  const void* src;          /* [N, C, P] f16 */
  const int64_t* frame;     /* [n_rows] */
  void* dst;
  int64_t dst_row_pitch;    /* halves */
  int dst_ctot, dst_coff;
  int dst_row0;
} vipe_nhwc_job;
int vipe_gather_nchw_to_nhwc_f16(const vipe_nhwc_job* jobs, int n_jobs, int n_rows, int C, int P, void* stream);

/* [fused] MotionFilter's dense score (vipe/slam/components/motion_filter.py:103-110): score[v] = mean over pixels of
 * |(half) delta| (masked: sum(|delta| (1 - invalid)) / P / (mean(1 - invalid) + 1e-6)).  dw [n_views, P, 4] f32 = the update
 * operator's (delta_x, delta_y, weight_x, weight_y); invalid [n_views, P] bytes (bool) or NULL; score [n_views] f32. */
int vipe_flow_score(const float* d_dw, const unsigned char* d_invalid, float* d_score, int n_views, int P, void* stream);

/* [fused] tail of FactorGraph.update (factor_graph.py:270-276): target = coords1 + delta, weight = (masked source frame
 * ? 0 : w), damping[du[k]] = eta[k].  coords1 / target / weight [E,ht,wd,2] f32, dw [E,ht,wd,4] f32, mask [E,ht,wd] bytes
 * (optional), eta [n_src,ht,wd], du [n_src] int64, damping [*,ht,wd]. */
int vipe_update_finish(const float* d_coords1, const float* d_dw, const unsigned char* d_mask, float* d_target,
                       float* d_weight, const float* d_eta, const int64_t* d_du, float* d_damping, int E, int n_src, int ht,
                       int wd, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Frame encoders: BasicEncoder fnet / cnet (vipe/slam/networks/droid_net.py:290-370,
 * DroidNet.encode_features / encode_context :510-527), run per frame by MotionFilter.check
 * (vipe/slam/components/motion_filter.py:58-150).  NHWC fp16 activations, fp32 bias, packed fp16 weights:
 *   conv: [k*k][Cin/32][Cout][32] (tap-major, 32-channel chunks);  stem: [7][32][32] with k = tap*4 + c (c = 3 and
 *   taps >= 49 zero).  Instance-norm statistics are [B, C, 2] fp32 (sum, sum of squares of the fp16 outputs),
 *   accumulated by the producing kernel (zero them first) and applied (x - mean) * rstd, eps 1e-5, + ReLU by the
 *   consumer while it loads.
 * ------------------------------------------------------------------------------------------- */
/* [V,3,H,W] fp32 RGB in [0,1] -> [V,H,W,4] fp16 ((x - mean) / std, 4th channel 0) */
int vipe_enc_prep(const float* d_img, void* d_x4, int V, int H, int W, void* stream);
/* 7x7 stride-2 pad-3 stem, 3(+1) -> 32 channels: y [B,H/2,W/2,32] = conv + bias (+ReLU if relu) */
int vipe_enc_stem(const void* d_x4, const void* d_w, const float* d_bias, void* d_y, float* d_out_stats, int B, int H,
                  int W, int relu, void* stream);
/* ksize in {1,3} (pad ksize/2), stride in {1,2}, Cin in {32,64,96,128}, Cout % 32 == 0.
 * d_in_stats != null: the input is relu(instance_norm(x)) formed on load.  d_res != null (NHWC only):
 * y = relu(res + act(conv)).  nchw: y is [B,Cout,Ho,Wo].  tanh_split >= 0: channels < split get tanh, the others
 * ReLU (context net output, droid_net.py:525-527). */
int vipe_enc_conv(const void* d_x, const float* d_in_stats, const void* d_w, const float* d_bias, const void* d_res,
                  void* d_y, float* d_out_stats, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                  int relu, int nchw, int tanh_split, void* stream);
/* residual tail of a normalised block: out = relu(xres + relu(IN(raw))); xres = res, or IN(res) if d_res_stats;
 * d_res == null: out = relu(IN(raw)).  All [B,HW,C] fp16. */
int vipe_enc_finish(const void* d_raw, const float* d_raw_stats, const void* d_res, const float* d_res_stats,
                    void* d_out, int B, int HW, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIPE_AMD_H */
