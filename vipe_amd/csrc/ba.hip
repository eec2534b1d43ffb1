// Dense bundle adjustment on SE3 (+) per-pixel inverse depth (+ focal / distortion), Gauss-Newton with
// Schur reduction and a dense block Cholesky, entirely on the device.
//
// Replaces the reference's LIVE Python solver: GraphBuffer.bundle_adjustment (components/buffer.py:373-525)
// -> Solver.run_inplace (ba/solver.py:117-197) -> DenseDepthFlowTerm / DispSensRegularizationTerm
// (ba/terms.py:94-303) -> geom.iproj_i_proj_j_disp (maths/geom.py:187-298) -> block-sparse products with
// host round trips (maths/matrix.py:66-81) -> scipy spsolve on the CPU (solver.py:33-44).
//
// MI355X design (one GN iteration = 3 launches, no host sync):
//   ba_accum_mfma_kernel (source frames of <= 6 terms) / ba_walk_kernel + ba_schur_kernel (any degree)
//                     grid (pixel tiles, source frames).  One lane owns pixel p of source frame k and walks
//                     ALL terms (edges) whose source is k (device-built CSR), so the per-pixel quantities
//                     C_k, w_k, E_kk stay in registers and are final when the walk ends; Jacobians never
//                     touch memory.  J^T W J blocks are Gram matrices on the matrix cores (v_mfma_f32_16x16x4_f32),
//                     combined in LDS and added to the dense reduced system in fp64.  The Schur complement of
//                     frame k (all member pairs of k) is formed right after the walk.
//   ba_solve_kernel   one workgroup: LM damping, blocked (6-wide) right-looking Cholesky in fp64 with the
//                     rhs carried as an extra matrix row (forward substitution for free), blocked backward
//                     substitution, pose / intrinsics retraction.
//   ba_retract_kernel per pixel dz = (w - sum_a E_ak^T dx_a)/C, d += dz (dz > 10 rejected).
//
// Comments of the form `// @stamp N`, `// @stampk N`, `// @wstampk N`, `// @bstamp N`, `// @bwave N`, `// @astamp N`,
// `// @kstamp N` / `// @kstampc N` mark phase boundaries: scratch/make_ba_stamps.py turns them into s_memtime stores of
// a VARIANT source for cycle measurements from inside the kernels.  This file compiles none of it.
#include <stdlib.h>

#include "term_geom.cuh"

namespace {

constexpr int TILE = 256;   // lanes per workgroup = pixels per tile
constexpr int NWAVE = TILE / WAVE;
constexpr int TCHUNK = 8;   // terms whose transforms are staged in LDS at a time
constexpr int AM_DMAX = 6;  // largest source-frame degree the matrix-core accumulate kernel handles

struct BAWs {
  int* rowptr;      // [nF+1] CSR over source disparity frames
  int* order;       // [M] term ids sorted by source frame (stable)
  int* pose_slot;   // [nP] slot in the reduced system or -1
  int* slot_pose;   // [nP] inverse map
  int* fflags;      // [nF] bit0 source, bit1 disparity free (the sensor-prior test is NOT part of the plan: finish_disp reads sens_sum, refreshed every call)
  int* scratch;     // [2*nP + nF]
  int* info;        // [8] n_free, n_free_disp, chol_fail, n_unknowns
  float* sens_sum;  // [nF]
  float* C;         // [nF,P] damped disparity diagonal
  float* wv;        // [nF,P]
  float* Ekk;       // [nF,6,P]
  float* Ef;        // [nF,2,P]
  float* Et;        // [nF,ntail,P] multi-view rigs: rows of the tail unknowns (per-view intrinsics, rig rotations)
  float* Ej;        // [M,6,P]
  double* S;        // [(nmax+1),(nmax+1)] lower triangle + rhs row
  double* Hd;       // [nmax] undamped diagonal of H (for lambda * diag)
  float* dx;        // [nmax]
  double* Wi;       // [ceil(nmax / 64)][64][64] inverses of the diagonal factor tiles (tiled Cholesky)
  int* krow;        // [nF] DROID mode: row of frame k in the sorted unique set arange(t0,t1) U ii (eta / dz row)
  int ld;           // nmax + 1
};

struct BAArgs {
  vipe_ba_params p;
  float *poses, *disps, *intr, *rig;
  const float *sens, *target, *weight, *eta;
  const int64_t *pi, *qi, *pj, *qj, *di;
  BAWs w;
  int P, nF, D;
  // multi-view rigs (n_views > 1): the tail of the reduced system holds one intrinsics block
  // per view (nintr = V (1 + D) unknowns when optimize_intrinsics) and one rotation block per view >= 1 (6 (V - 1) when
  // optimize_rig_rotation; view 0 is the gauge, buffer.py:506); ntail = both.  Mono: the F <= 2 shared intrinsics.
  int mv, nintr, ntail;
  int force_general; // vipe_ba_params.solver_options & VIPE_BA_OPT_GENERAL_ACCUMULATE
  int band2;         // two-chain band solve for long pose-only chains (off: VIPE_BA_OPT_ONE_CHAIN)
  // DROID semantics of slam_ext.ba (geom_kernels.cu:178-432, 1273-1404; see oracle/droid_ba.py for the list):
  // target / weight [M,2,P], eta [K,P] by krow, per-pixel depth prior, reduced-diagonal damping, poses free iff in
  // [t0,t1), stereo terms, MIN_DEPTH 0.25, pose t0 left out of the disparity back-substitution, dz written to dz_out
  int droid;
  float* dz_out;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

constexpr int RG_VMAX = 8;  // views of a rig the multi-view kernels handle: the per-view tail rows of a pixel live in registers
                            // (instantiations for <= 4 and <= 8 views; the reference's Solver is generic in V, buffer.py:404-506)
inline bool is_multiview(const vipe_ba_params& p) { return p.n_views > 1; }  // a mono rig has no rig unknowns (view 0 is the gauge)
inline int tail_intr(const vipe_ba_params& p) {
  const int F = 1 + (p.camera == VIPE_CAM_MEI ? 1 : 0);
  return p.optimize_intrinsics ? (is_multiview(p) ? p.n_views * F : F) : 0;
}
inline int tail_rig(const vipe_ba_params& p) { return p.optimize_rig_rotation ? 6 * (p.n_views - 1) : 0; }

size_t carve(const vipe_ba_params& p, char* base, BAWs* out) {
  const size_t nP = p.n_poses, nF = (size_t)p.n_poses * p.n_views, P = (size_t)p.ht * p.wd, M = p.M;
  const size_t ntail_max = is_multiview(p) ? (size_t)p.n_views * 2 + 6 * (size_t)(p.n_views - 1) : 2;
  const size_t nmax = 6 * nP + ntail_max;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* ptr = base ? base + off : nullptr;
    off += align_up(bytes);
    return ptr;
  };
  BAWs w;
  w.rowptr = (int*)take(4 * (nF + 1));
  w.order = (int*)take(4 * (M + 1));
  w.pose_slot = (int*)take(4 * nP);
  w.slot_pose = (int*)take(4 * nP);
  w.fflags = (int*)take(4 * nF);
  w.scratch = (int*)take(4 * (2 * nP + nF));
  w.info = (int*)take(4 * 8);
  w.sens_sum = (float*)take(4 * nF);
  w.C = (float*)take(4 * nF * P);
  w.wv = (float*)take(4 * nF * P);
  w.Ekk = (float*)take(4 * nF * 6 * P);
  w.Ef = (float*)take(4 * nF * 2 * P);
  w.Et = (float*)take(is_multiview(p) ? 4 * nF * ntail_max * P : 0);
  w.Ej = (float*)take(4 * (M + 1) * 6 * P);
  w.S = (double*)take(8 * (nmax + 1) * (nmax + 1));
  w.Hd = (double*)take(8 * (nmax + 16));  // + 16 debug stamp slots
  w.dx = (float*)take(4 * nmax);
  w.Wi = (double*)take(8 * 64 * 64 * ((nmax + 63) / 64));
  w.krow = (int*)take(4 * (nF + 1));
  w.ld = (int)(nmax + 1);
  if (out) *out = w;
  return off;
}

// ------------------------------------------------------------------------------------------------ plan

__global__ __launch_bounds__(256) void ba_sens_kernel(const float* __restrict__ sens, float* __restrict__ out, int P,
                                                      int* __restrict__ info) {
  // buffer.py:470-471: frames whose sensor disparity sums to > 0
  const int k = blockIdx.x;
  if (k == 0 && threadIdx.x == 0) {
    info[2] = 0;  // Cholesky failure count of this call (also when the plan is reused)
    info[5] = 0;  // "band solver solved": normally reset by that kernel itself, which a path hint may leave out
  }
  float s = 0.f;
  for (int p = threadIdx.x; p < P; p += blockDim.x) s += sens[(int64_t)k * P + p];
  s = wave_sum(s);
  __shared__ float red[NWAVE];
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[k] = red[0] + red[1] + red[2] + red[3];
}

// exclusive scan of v[0..n) in place with 1024 threads; returns the total (all threads)
__device__ int block_scan_excl(int* v, int n, int* lds /* [1024] */) {
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = t * per;
  int s = 0;
  for (int i = b; i < b + per && i < n; ++i) s += v[i];
  lds[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int x = t >= o ? lds[t - o] : 0;
    __syncthreads();
    lds[t] += x;
    __syncthreads();
  }
  int run = t > 0 ? lds[t - 1] : 0;
  const int total = lds[1023];
  for (int i = b; i < b + per && i < n; ++i) {
    int c = v[i];
    v[i] = run;
    run += c;
  }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(1024) void ba_plan_kernel(BAArgs a) {
  const vipe_ba_params& p = a.p;
  const int t = threadIdx.x;
  const int nP = p.n_poses, nF = a.nF, M = p.M, V = p.n_views;
  int* cnt = a.w.scratch;           // [nF]
  int* is_src = cnt + nF;           // [nP]
  int* used = is_src + nP;          // [nP]
  __shared__ int lds[1024];
  __shared__ int stage[4096];
  for (int i = t; i < nF; i += 1024) cnt[i] = 0;
  for (int i = t; i < nP; i += 1024) { is_src[i] = 0; used[i] = 0; }
  __syncthreads();
  for (int e = t; e < M; e += 1024) {
    atomicAdd(&cnt[(int)a.di[e]], 1);
    is_src[(int)a.pi[e]] = 1;
    used[(int)a.pi[e]] = 1;
    used[(int)a.pj[e]] = 1;
  }
  __syncthreads();
  // frame flags before cnt is turned into offsets
  const bool all_fixed = !(p.t0 < p.t1);
  int nfd_local = 0;
  for (int k = t; k < nF; k += 1024) {
    const int pose = k / V;
    int f = cnt[k] > 0 ? 1 : 0;
    if (a.droid) {
      // disparity frames = unique(arange(t0,t1) U ii) (geom_kernels.cu:1297-1303); bit 3: in the set without terms
      const bool inkx = f || (k >= p.t0 && k < p.t1);
      a.w.krow[k] = inkx ? 1 : 0;
      if (inkx && !p.motion_only) { f |= 2; ++nfd_local; if (!(f & 1)) f |= 8; }
    } else {
      bool dfree = f && !p.motion_only && !(p.limited_disp && (pose < p.t0 || pose >= p.t1));  // buffer.py:490-493
      if (dfree) { f |= 2; ++nfd_local; }
    }
    a.w.fflags[k] = f;
  }
  if (a.droid) {
    __syncthreads();
    block_scan_excl(a.w.krow, nF, lds);
  }
  // rowptr = exclusive scan of counts
  for (int i = t; i < nF; i += 1024) a.w.rowptr[i] = cnt[i];
  __syncthreads();
  const int total = block_scan_excl(a.w.rowptr, nF, lds);
  if (t == 0) a.w.rowptr[nF] = total;
  // pose slots (buffer.py:462-465: fixed iff it is a source pose outside [t0,t1); t0 == t1 fixes all)
  for (int i = t; i < nP; i += 1024) {
    const bool fixed = all_fixed || (is_src[i] && (i < p.t0 || i >= p.t1));
    // DROID: the system has one block per pose of [t0, t1), used or not (SparseBlock(t1 - t0, 6))
    a.w.pose_slot[i] = a.droid ? ((i >= p.t0 && i < p.t1) ? 1 : 0) : ((used[i] && !fixed) ? 1 : 0);
  }
  __syncthreads();
  for (int i = t; i < nP; i += 1024) is_src[i] = a.w.pose_slot[i];  // keep the 0/1 flags
  __syncthreads();
  const int n_free = block_scan_excl(a.w.pose_slot, nP, lds);
  for (int i = t; i < nP; i += 1024) {
    if (is_src[i]) a.w.slot_pose[a.w.pose_slot[i]] = i;
    else a.w.pose_slot[i] = -1;
  }
  // count free disparity frames
  lds[t] = nfd_local;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) lds[t] += lds[t + o];
    __syncthreads();
  }
  if (t == 0) {
    a.w.info[0] = n_free;
    a.w.info[1] = lds[0];
    a.w.info[2] = 0;
    a.w.info[3] = 6 * n_free + a.ntail;
    a.w.info[4] = 0;  // band width of the reduced pose system in 6x6 blocks (filled below)
    a.w.info[6] = 0;  // largest number of terms of one source frame (selects the accumulate kernel)
  }
  // stable counting sort of the terms by source frame: frame k's owner scans the term list in order
  // (cursor kept in cnt[]: reuse cnt as the running write position)
  __syncthreads();
  for (int k = t; k < nF; k += 1024) cnt[k] = a.w.rowptr[k];
  for (int c0 = 0; c0 < M; c0 += 4096) {
    const int nc = min(4096, M - c0);
    __syncthreads();
    for (int i = t; i < nc; i += 1024) stage[i] = (int)a.di[c0 + i];
    __syncthreads();
    for (int k = t; k < nF; k += 1024) {
      if (!(a.w.fflags[k] & 1)) continue;
      int pos = cnt[k];
      for (int i = 0; i < nc; ++i)
        if (stage[i] == k) a.w.order[pos++] = c0 + i;
      cnt[k] = pos;
    }
  }
  // Band of the reduced system: two poses couple (directly through H_ij or through the Schur complement of a
  // source frame) only if they are members {pose of k} + {targets of k's terms} of the same frame k.
  __syncthreads();
  for (int k = t; k < nF; k += 1024) {
    if (!(a.w.fflags[k] & 1)) continue;
    int lo = 1 << 30, hi = -1;
    const int si = a.w.pose_slot[k / V];
    if (si >= 0) { lo = si; hi = si; }
    for (int q = a.w.rowptr[k]; q < a.w.rowptr[k + 1]; ++q) {
      const int sj = a.w.pose_slot[(int)a.pj[a.w.order[q]]];
      if (sj >= 0) { lo = min(lo, sj); hi = max(hi, sj); }
    }
    if (hi >= 0) atomicMax(&a.w.info[4], hi - lo);
    atomicMax(&a.w.info[6], a.w.rowptr[k + 1] - a.w.rowptr[k]);
  }
}

// ------------------------------------------------------------------------------------------------ accumulate

// target / weight of term e at pixel p: live layout [M,P,2], DROID layout [M,2,P] (geom_kernels.cu:304-309)
__device__ __forceinline__ void load_tw(const BAArgs& a, int e, int p, int P, float2& tgt, float2& wg) {
  if (a.droid) {
    const int64_t o = (int64_t)e * 2 * P + p;
    tgt = make_float2(a.target[o], a.target[o + P]);
    wg = make_float2(a.weight[o], a.weight[o + P]);
  } else {
    const int64_t o2 = ((int64_t)e * P + p) * 2;
    tgt = *reinterpret_cast<const float2*>(a.target + o2);
    wg = *reinterpret_cast<const float2*>(a.weight + o2);
  }
}
// validity weight: live z0... target-side z > 0.1 (geom.py:263); DROID !(z < 0.25) (geom_kernels.cu:33,304)
__device__ __forceinline__ float valid_weight(const BAArgs& a, float Z, bool inb) {
  const bool ok = a.droid ? !(Z < 0.25f) : (Z > cam::MIN_DEPTH);
  return (inb && ok) ? a.p.weight_scale : 0.0f;
}
// per-term transforms incl. the DROID stereo term (ii == jj: fixed baseline, no pose blocks; geom_kernels.cu:222-233)
__device__ __forceinline__ void term_setup(const BAArgs& a, int e, TermGeom& g) {
  const int pi = (int)a.pi[e], pj = (int)a.pj[e], qj = (int)a.qj[e];
  term_transforms(a.poses, a.rig, pi, (int)a.qi[e], pj, qj, g.T, g.G, g.Rr);
  g.Ij = cam::load_scaled(a.intr + qj * (4 + a.D), a.D, 1.0f / a.p.intr_factor);
  g.e = e;
  g.merge = (pi == pj);
  if (a.droid && pi == pj) {
    g.merge = 2;  // stereo
    for (int i = 0; i < 9; ++i) g.T.R[i] = g.G.R[i] = (i % 4 == 0) ? 1.0f : 0.0f;
    g.T.t[0] = g.G.t[0] = -0.1f;
    g.T.t[1] = g.T.t[2] = g.G.t[1] = g.G.t[2] = 0.0f;
  }
  g.rig_adj = !(g.Rr.t[0] == 0.f && g.Rr.t[1] == 0.f && g.Rr.t[2] == 0.f && g.Rr.R[0] == 1.f &&
                g.Rr.R[4] == 1.f && g.Rr.R[8] == 1.f);
  g.sj = g.merge ? -1 : a.w.pose_slot[pj];
}
// sensor-depth prior and damping of one pixel's disparity block.  Live: frame-level flag, C += alpha, then the
// damping 1e-7 + (0.2 eta + 1e-7) (terms.py:258-268, buffer.py:482-489).  DROID: per-pixel mask m = sens > 0,
// C += m ? alpha : eta, w -= m alpha (d - sens) (geom_kernels.cu:1359-1369).
__device__ __forceinline__ void finish_disp(const BAArgs& a, int k, int p, int P, int flags, float d, float& C, float& wz) {
  const int64_t kp = (int64_t)k * P + p;
  if (a.droid) {
    const float sv = a.sens[kp];
    if (sv > 0.0f) { C += a.p.alpha; wz -= a.p.alpha * (d - sv); }
    else C += a.eta[(int64_t)a.w.krow[k] * P + p];
  } else {
    if (a.w.sens_sum[k] > 0.0f) {  // frames with sensor depth (buffer.py:470-471); read per call, not part of the plan
      C += a.p.alpha;
      wz -= a.p.alpha * (d - a.sens[kp]);
    }
    C += 1e-7f + (0.2f * a.eta[kp] + 1e-7f);
  }
}

// lower-triangle index table for symmetric 6x6 (21 entries): (r,c), r >= c
__device__ __constant__ int8_t SYM_R[21] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5};
__device__ __constant__ int8_t SYM_C[21] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5};

__device__ __forceinline__ void s_add(const BAWs& w, int row, int col, double v) {
  // lower triangle storage: (row, col) with row >= col
  if (row < col) { int tmp = row; row = col; col = tmp; }
  atomicAdd(&w.S[(int64_t)row * w.ld + col], v);
}

// Reduce vals[0..N) over the workgroup; lane `i` of the first N lanes receives the sum of value i.
template <int N>
__device__ __forceinline__ float block_reduce(float (&vals)[N], float* red /* [NWAVE][N] */) {
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = wave_sum_to_lane63(vals[i]);
  __syncthreads();  // previous consumers of `red` are done
  if (lane_id() == 63) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[wave_id() * N + i] = vals[i];
  }
  __syncthreads();
  float s = 0.f;
  if ((int)threadIdx.x < N) {
#pragma unroll
    for (int wv = 0; wv < NWAVE; ++wv) s += red[wv * N + threadIdx.x];
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ accumulate (MFMA)
//
// Every reduction over pixels runs on the matrix cores instead of DPP + LDS
// trees (the shuffle version spends ~9.5k DPP adds and ~70 workgroup barriers per tile on 1,281 reduced values):
//   * R1, per term: the lane (= pixel) writes the rows sqrt(w_c) * [Jj_c(6); r_c; Jf_c(F)] of its pixel for both
//     residual components c into a wave-private LDS tile [16 rows][2 x 64 pixels]; one chain of 32
//     v_mfma_f32_16x16x4_f32 (A = B = the tile) yields the Gram matrix = H_jj, -v_j, H_jf, H_ff, -v_f of the term
//     over the wave's 64 pixels.  Only the TARGET-side blocks are reduced: J_i = M J_j per term (M = -Adj(G_ij)^T,
//     or I - Adj^T when both ends are views of one pose), so H_ii += M H_jj M^T, H_ij = M H_jj, v_i += M v_j and
//     H_if += M H_jf are formed from the 6x6 sums by a few lanes afterwards.
//   * R2, per source frame: rows sqrt(Q) * [E_i; E_j(terms); E_f; w] (Q = 1/C per pixel) go to the same LDS region
//     [<= 48 rows][64 pixels]; the Schur complement E Q E^T and the reduced rhs E Q w are the lower-triangle tiles of
//     its Gram matrix (16 MFMAs per 16x16 tile pair).
// Waves never wait for each other inside the walk: wave-private LDS, LDS float atomics into workgroup accumulators,
// three workgroup barriers in total.  Exact fp32 products and sums (the f32 MFMA is an fmaf chain).
// Handles source frames with at most AM_DMAX terms (the radius-3 neighbourhood graph: 6); ba_plan_kernel publishes
// the largest degree in info[6] and exactly one of the two accumulate kernels runs.
constexpr int AM_ROWS = 48;             // R2 row capacity: 6 (AM_DMAX + 1) + F + 1 <= 48
constexpr int AM_P1 = 130;              // float pitch of the R1 tile [16][128]   (= 2 mod 32: conflict-free b32 reads)
constexpr int AM_P2 = 66;               // float pitch of the R2 image [48][64]
constexpr int AM_P2H = 34;              // float pitch of HALF the R2 image [48][32]: the fused kernel reduces the wave's
                                        // 64 pixels in two passes of 32 (34 = 2 + 32 mod 64: lanes (row l16, pixel kq) of an
                                        // MFMA operand read hit 64 distinct banks)
constexpr int AM_WBUF = 16 * AM_P1;     // 2080 floats per wave: the R1 tile (R2 half image: 48 * 34 = 1632)
static_assert(AM_ROWS * AM_P2H <= AM_WBUF, "R2 half image must fit the wave buffer");
constexpr int AM_SP = AM_ROWS + 1;

struct TermGeomM {
  TermGeom g;
  float Mi[36];  // J_i = Mi J_j
};

constexpr size_t accum_mfma_lds() {
  return sizeof(float) * (NWAVE * AM_WBUF + AM_ROWS * AM_SP + AM_DMAX * 256 + 64) + AM_DMAX * sizeof(TermGeomM);
}

typedef float float4m __attribute__((ext_vector_type(4)));

template <int CAM, int F>
__global__ __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu(3, 3))) void ba_accum_mfma_kernel(BAArgs a) {
  constexpr int FF = F > 0 ? F : 1;
  constexpr int RPT = F > 0 ? 16 : 8;  // R1 rows per term: 6 J, r, F Jf (padded)
  constexpr int TPT = 16 / RPT;        // terms per R1 tile
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  if (a.force_general || w.info[6] > AM_DMAX) return;  // force_general: the walk + Schur pair takes every graph (vipe_ba_params.solver_options)
  const int k = blockIdx.y;
  const int beg = w.rowptr[k], end = w.rowptr[k + 1];
  if (beg == end) return;
  const int deg = end - beg;
  const int P = a.P, V = prm.n_views, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p_raw = blockIdx.x * TILE + tid;
  const bool inb = p_raw < P;
  const int p = inb ? p_raw : P - 1;
  const int flags = w.fflags[k];
  const bool dfree = flags & 2;
  const int pose_i = k / V, qi = k % V;
  const int si = w.pose_slot[pose_i];
  const bool fi = si >= 0;
  const int n_free = w.info[0], nrow = w.info[3];
  const int foff = 6 * n_free;

  // @astamp 0
  extern __shared__ __align__(16) float am_smem[];
  float* wbuf = am_smem + wave * AM_WBUF;       // wave-private
  float* accS = am_smem + NWAVE * AM_WBUF;      // [48][49] Schur Gram accumulators
  float* acc1 = accS + AM_ROWS * AM_SP;         // [AM_DMAX][16][16] per-term Gram accumulators
  float* accI = acc1 + AM_DMAX * 256;           // [64] frame level: H_ii 36, v_i 6, H_if 6F, H_ff 3, v_f F
  TermGeomM* tg = reinterpret_cast<TermGeomM*>(accI + 64);

  for (int i = tid; i < AM_ROWS * AM_SP + AM_DMAX * 256 + 64; i += TILE) accS[i] = 0.0f;
  if (tid < deg) {
    TermGeomM m;
    TermGeom& g = m.g;
    term_setup(a, w.order[beg + tid], g);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float ec[6] = {0, 0, 0, 0, 0, 0}, col[6];
      ec[c] = 1.0f;
      adjT_apply(g.G, ec, col);
#pragma unroll
      for (int r = 0; r < 6; ++r) m.Mi[r * 6 + c] = (g.merge == 1 && r == c ? 1.0f : 0.0f) - col[r];
    }
    tg[tid] = m;
  }
  __syncthreads();
  // @astamp 1

  const cam::Intr Ii = cam::load_scaled(a.intr + qi * (4 + a.D), a.D, 1.0f / prm.intr_factor);
  const float u = (float)(p % prm.wd), v = (float)(p / prm.wd);
  const float d = a.disps[(int64_t)k * P + p];
  float X0, Y0, dX0[FF], dY0[FF];
  cam::iproj<CAM, F>(Ii, u, v, X0, Y0, dX0, dY0);
  float C = 0.f, wz = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0}, Efr[FF] = {};
  const int l16 = lane & 15, kq = lane >> 4;

  // target / weight of the next tile's terms are fetched while the current tile is computed (the walk is otherwise
  // a chain of dependent global-load latencies: measured 5 us per tile)
  float2 nx_t[TPT], nx_w[TPT];
  auto prefetch = [&](int t0) {
#pragma unroll
    for (int uu = 0; uu < TPT; ++uu) {
      const int t = min(t0 + uu, deg - 1);
      load_tw(a, tg[t].g.e, p, P, nx_t[uu], nx_w[uu]);
    }
  };
  prefetch(0);
  for (int t0 = 0; t0 < deg; t0 += TPT) {
    float2 cur_t[TPT], cur_w[TPT];
#pragma unroll
    for (int uu = 0; uu < TPT; ++uu) { cur_t[uu] = nx_t[uu]; cur_w[uu] = nx_w[uu]; }
    if (t0 + TPT < deg) prefetch(t0 + TPT);
#pragma unroll
    for (int uu = 0; uu < TPT; ++uu) {
      const int t = t0 + uu;
      if (t >= deg) break;  // workgroup-uniform
      const TermGeom& G = tg[t].g;
      const int e = G.e;
      const float X = G.T.R[0] * X0 + G.T.R[1] * Y0 + G.T.R[2] + G.T.t[0] * d;
      const float Y = G.T.R[3] * X0 + G.T.R[4] * Y0 + G.T.R[5] + G.T.t[1] * d;
      const float Z = G.T.R[6] * X0 + G.T.R[7] * Y0 + G.T.R[8] + G.T.t[2] * d;
      float x, y, Jp[2][3], Jfj[2][FF];
      cam::proj<CAM, true, F>(G.Ij, X, Y, Z, x, y, Jp, Jfj);
      const float2 tgt = cur_t[uu], wg = cur_w[uu];
      const float val = valid_weight(a, Z, inb);  // geom.py:263, buffer.py:413
      const float wd2[2] = {val * wg.x, val * wg.y};                     // weights of the disparity system
      const float wc[2] = {G.merge == 2 ? 0.0f : wd2[0], G.merge == 2 ? 0.0f : wd2[1]};  // ... of the pose blocks
      const float rc[2] = {x - tgt.x, y - tgt.y};
      float Ja[3][6] = {{d, 0, 0, 0, Z, -Y}, {0, d, 0, -Z, 0, X}, {0, 0, d, Y, -X, 0}};
      if (G.rig_adj) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float tmp[6];
          adjT_apply(G.Rr, Ja[r], tmp);
#pragma unroll
          for (int q = 0; q < 6; ++q) Ja[r][q] = tmp[q];
        }
      }
      const bool fj = G.sj >= 0;
      float Ejv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float Jj[6], Ji[6], Jf[FF];
#pragma unroll
        for (int q = 0; q < 6; ++q) Jj[q] = Jp[c][0] * Ja[0][q] + Jp[c][1] * Ja[1][q] + Jp[c][2] * Ja[2][q];
        const float Jz = Jp[c][0] * G.T.t[0] + Jp[c][1] * G.T.t[1] + Jp[c][2] * G.T.t[2];
        if constexpr (F > 0) {
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const float ax = G.T.R[0] * dX0[f] + G.T.R[1] * dY0[f];
            const float ay = G.T.R[3] * dX0[f] + G.T.R[4] * dY0[f];
            const float az = G.T.R[6] * dX0[f] + G.T.R[7] * dY0[f];
            Jf[f] = (Jp[c][0] * ax + Jp[c][1] * ay + Jp[c][2] * az + Jfj[c][f]) * (1.0f / prm.intr_factor);
          }
        }
        // R1 rows of this term and component
        const float sw = __builtin_amdgcn_sqrtf(wc[c]);  // v_sqrt_f32 (1 ulp): only splits w between the Gram factors
        float* col = wbuf + (uu * RPT) * AM_P1 + c * 64 + lane;
#pragma unroll
        for (int q = 0; q < 6; ++q) col[q * AM_P1] = Jj[q] * sw;
        col[6 * AM_P1] = rc[c] * sw;
        if constexpr (F > 0) {
#pragma unroll
          for (int f = 0; f < F; ++f) col[(7 + f) * AM_P1] = Jf[f] * sw;
        }
        // per-pixel disparity quantities
        if (dfree) {
          const float wJz = wc[c] * Jz;
          C += wd2[c] * Jz * Jz;
          wz -= wd2[c] * Jz * rc[c];
          float tmp[6];
          adjT_apply(G.G, Jj, tmp);
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            Ji[q] = (G.merge == 1 ? Jj[q] : 0.0f) - tmp[q];
            Ei[q] += Ji[q] * wJz;
          }
          if constexpr (F > 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) Efr[f] += Jf[f] * wJz;
          }
#pragma unroll
          for (int q = 0; q < 6; ++q) Ejv[q] += Jj[q] * wJz;
        }
      }
      if (dfree && fj && inb) {
#pragma unroll
        for (int q = 0; q < 6; ++q) w.Ej[((int64_t)e * 6 + q) * P + p] = Ejv[q];
      }
    }
    // ---- Gram matrix of the tile over this wave's 64 pixels x 2 components
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float4m g4 = {0.f, 0.f, 0.f, 0.f};
    const float* arow = wbuf + l16 * AM_P1 + kq;
#pragma unroll 8
    for (int s = 0; s < 32; ++s) {
      const float av = arow[4 * s];
      g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, g4, 0, 0, 0);
    }
    // D[row = 4 kq + r][col = l16]: keep the diagonal RPT x RPT blocks
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * kq + r;
      const int tb = row / RPT;
      if (l16 / RPT == tb && t0 + tb < deg) atomicAdd(&acc1[(t0 + tb) * 256 + (row % RPT) * 16 + (l16 % RPT)], g4[r]);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // @astamp 2
  // ---- finish the disparity block of this pixel: sensor prior, damping (terms.py:258-268, buffer.py:482-489)
  float sq = 0.0f;
  const int NR = 6 * (deg + 1) + F + 1;  // R2 rows: pose i, targets, intrinsics, w
  if (dfree) {
    const int64_t kp = (int64_t)k * P + p;
    finish_disp(a, k, p, P, flags, d, C, wz);
    if (inb) {
      w.C[kp] = C;
      w.wv[kp] = wz;
#pragma unroll
      for (int q = 0; q < 6; ++q) w.Ekk[((int64_t)k * 6 + q) * P + p] = Ei[q];
      if constexpr (F > 0) {
#pragma unroll
        for (int f = 0; f < F; ++f) w.Ef[((int64_t)k * 2 + f) * P + p] = Efr[f];
      }
      sq = __builtin_amdgcn_rsqf(C);  // sqrt(Q), Q = 1 / C
    }
    // R2 rows, scaled by sqrt(Q).  The wave's 64 pixels are reduced in two passes of 32 (lanes 0-31, then 32-63, put
    // their rows into the [48][32] image; the Gram chains continue across the passes): the wave buffer then is the
    // 8 KiB of the R1 tile instead of 12.4 KiB, three workgroups instead of two fit a CU and the 576 workgroups of the
    // 48-keyframe graph are resident at once instead of in two rounds.
    // all E_j rows of this pixel in flight at once (a per-term loop serialises one L2 round trip per term)
    float ej[AM_DMAX][6];
#pragma unroll
    for (int t = 0; t < AM_DMAX; ++t) {
      const bool on = t < deg && tg[min(t, deg - 1)].g.sj >= 0 && inb;
      const int64_t eb = (int64_t)tg[min(t, deg - 1)].g.e * 6 * P + p;
#pragma unroll
      for (int q = 0; q < 6; ++q) ej[t][q] = on ? w.Ej[eb + (int64_t)q * P] : 0.0f;
    }
    constexpr int NPAIR = 6;  // lower-triangle 16 x 16 tile pairs of up to 48 rows
    float4m g4[NPAIR];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) g4[i] = float4m{0.f, 0.f, 0.f, 0.f};
    const int RT = (NR + 15) >> 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((lane >> 5) == half) {
        float* col = wbuf + (lane & 31);
#pragma unroll
        for (int q = 0; q < 6; ++q) col[q * AM_P2H] = fi ? Ei[q] * sq : 0.0f;
#pragma unroll
        for (int t = 0; t < AM_DMAX; ++t) {
          if (t < deg) {
#pragma unroll
            for (int q = 0; q < 6; ++q) col[(6 * (t + 1) + q) * AM_P2H] = ej[t][q] * sq;
          }
        }
        if constexpr (F > 0) {
#pragma unroll
          for (int f = 0; f < F; ++f) col[(6 * (deg + 1) + f) * AM_P2H] = Efr[f] * sq;
        }
        col[(NR - 1) * AM_P2H] = wz * sq;
        for (int r = NR; r < ((NR + 15) & ~15); ++r) col[r * AM_P2H] = 0.0f;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = 0; tb <= ta; ++tb) {
          if (ta < RT) {  // wave-uniform
            const float* ar = wbuf + (16 * ta + l16) * AM_P2H + kq;
            const float* br = wbuf + (16 * tb + l16) * AM_P2H + kq;
            float4m acc = g4[ta * (ta + 1) / 2 + tb];
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4 * s], br[4 * s], acc, 0, 0, 0);
            g4[ta * (ta + 1) / 2 + tb] = acc;
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the second half overwrites the image the chains above have read
    }
#pragma unroll
    for (int ta = 0; ta < 3; ++ta)
#pragma unroll
      for (int tb = 0; tb <= ta; ++tb) {
        if (ta < RT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * ta + 4 * kq + r, cc = 16 * tb + l16;
            if (row < NR && cc <= row) atomicAdd(&accS[row * AM_SP + cc], g4[ta * (ta + 1) / 2 + tb][r]);
          }
        }
      }
  }
  // @astamp 3
  __syncthreads();
  // @astamp 4

  // ---- per-term blocks from the Gram sums (one wave per term)
  for (int t = wave; t < deg; t += NWAVE) {
    const TermGeomM& TG = tg[t];
    const float* Gm = acc1 + t * 256;  // [16][16]: rows/cols 0..5 J, 6 r, 7.. Jf
    const float* Mi = TG.Mi;
    const int sj = TG.g.sj;
    const bool fj = sj >= 0;
    const int bj = 6 * sj, bi = 6 * si;
    float* T1 = wbuf;  // [6][6 + F] scratch: Mi Hjj | Mi Hjf
    if (lane < 36) {
      const int r = lane / 6, c = lane % 6;
      const float hjj = Gm[r * 16 + c];
      if (fj && r >= c) {
        s_add(w, bj + r, bj + c, (double)hjj);
        if (r == c) atomicAdd(&w.Hd[bj + r], (double)hjj);
      }
      float t1 = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) t1 += Mi[r * 6 + q] * Gm[q * 16 + c];
      T1[r * 8 + c] = t1;
      if (fi && fj) s_add(w, bi + r, bj + c, (double)t1);  // H_ij = Mi H_jj
    } else if (lane < 42) {
      const int q = lane - 36;
      const float vjn = Gm[q * 16 + 6];  // sum w J_q r  (v_j = -that)
      if (fj) atomicAdd(&w.S[(int64_t)nrow * w.ld + bj + q], -(double)vjn);
      if (fi) {
        float vin = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) vin += Mi[q * 6 + c] * Gm[c * 16 + 6];
        atomicAdd(&accI[36 + q], -vin);
      }
    } else if (F > 0 && lane < 42 + 6 * F) {
      const int q = (lane - 42) / FF, f = (lane - 42) % FF;
      const float hjf = Gm[q * 16 + 7 + f];
      if (fj) s_add(w, foff + f, bj + q, (double)hjf);
      if (fi) {
        float hif = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) hif += Mi[q * 6 + c] * Gm[c * 16 + 7 + f];
        atomicAdd(&accI[42 + q * FF + f], hif);
      }
    } else if (F > 0 && lane < 42 + 6 * F + F * F) {
      const int i2 = lane - 42 - 6 * F, f = i2 / FF, f2 = i2 % FF;
      if (f >= f2) atomicAdd(&accI[42 + 6 * FF + f * FF + f2], Gm[(7 + f) * 16 + 7 + f2]);
    } else if (F > 0 && lane < 42 + 6 * F + F * F + F) {
      const int f = lane - 42 - 6 * F - F * F;
      atomicAdd(&accI[42 + 6 * FF + FF * FF + f], -Gm[(7 + f) * 16 + 6]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (fi && lane < 36) {
      const int r = lane / 6, c = lane % 6;
      float hii = 0.f;
#pragma unroll
      for (int q = 0; q < 6; ++q) hii += T1[r * 8 + q] * Mi[c * 6 + q];
      atomicAdd(&accI[r * 6 + c], hii);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // @astamp 5

  // ---- frame level: H_ii, v_i, H_if, H_ff, v_f
  {
    const int bi = 6 * si;
    if (tid < 36) {
      const int r = tid / 6, c = tid % 6;
      if (fi && r >= c) {
        const double s = (double)accI[r * 6 + c];
        s_add(w, bi + r, bi + c, s);
        if (r == c) atomicAdd(&w.Hd[bi + r], s);
      }
    } else if (tid < 42) {
      if (fi) atomicAdd(&w.S[(int64_t)nrow * w.ld + bi + (tid - 36)], (double)accI[tid]);
    } else if (F > 0 && tid < 42 + 6 * F) {
      const int q = (tid - 42) / FF, f = (tid - 42) % FF;
      if (fi) s_add(w, foff + f, bi + q, (double)accI[tid]);
    } else if (F > 0 && tid < 42 + 6 * F + F * F) {
      const int i2 = tid - 42 - 6 * F, f = i2 / FF, f2 = i2 % FF;
      if (f >= f2) {
        const double s = (double)accI[tid];
        s_add(w, foff + f, foff + f2, s);
        if (f == f2) atomicAdd(&w.Hd[foff + f], s);
      }
    } else if (F > 0 && tid < 42 + 6 * F + F * F + F) {
      atomicAdd(&w.S[(int64_t)nrow * w.ld + foff + (tid - 42 - 6 * F - F * F)], (double)accI[tid]);
    }
  }
  // @astamp 6
  // ---- Schur complement of frame k: S -= E Q E^T, g -= E Q w   (solver.py:176-178)
  if (dfree) {
    auto gmap = [&](int row) -> int {
      if (row >= 6 * (deg + 1)) return foff + (row - 6 * (deg + 1));
      const int m = row / 6, q = row % 6;
      const int sl = m == 0 ? si : tg[m - 1].g.sj;
      return sl >= 0 ? 6 * sl + q : -1;
    };
    for (int i = tid; i < NR * NR; i += TILE) {
      const int row = i / NR, cc = i % NR;
      if (cc > row || cc == NR - 1) continue;
      const int gc = gmap(cc);
      if (gc < 0) continue;
      const float val = accS[row * AM_SP + cc];
      if (row == NR - 1) {
        atomicAdd(&w.S[(int64_t)nrow * w.ld + gc], -(double)val);
      } else {
        const int gr = gmap(row);
        if (gr >= 0) s_add(w, gr, gc, -(double)val);
      }
    }
  }
  // @astamp 7
}

// ---- general accumulate (any number of terms per source frame): the same walk with matrix-core Gram reductions, the
// terms staged 8 at a time, WITHOUT the Schur complement - that is formed afterwards by ba_schur_kernel from the E rows
// this kernel leaves in the workspace (E_kk, E_j, E_f, w, C).  LDS per workgroup is independent of the degree (42 KB).
constexpr int WK_CH = 8;    // rig walk: terms per chunk
constexpr int WK_CH1 = 12;  // mono walk: the keyframe frontend's source frames have 6-12 terms - with 8 per chunk a 9-term
                            // frame paid a second chunk's set-up (10k cycles of dependent loads, stamps) for one term: 17k of 66k
constexpr size_t walk_lds() {
  return sizeof(float) * (NWAVE * 16 * AM_P1 + WK_CH1 * 256 + 64) + WK_CH1 * sizeof(TermGeomM);
}

template <int CAM, int F>
__global__ __launch_bounds__(TILE) void ba_walk_kernel(BAArgs a) {
  constexpr int WBUF = 16 * AM_P1;
  constexpr int FF = F > 0 ? F : 1;
  constexpr int RPT = F > 0 ? 16 : 8;  // R1 rows per term: 6 J, r, F Jf (padded)
  constexpr int TPT = 16 / RPT;        // terms per R1 tile
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  if (!a.force_general && w.info[6] <= AM_DMAX) return;  // low-degree graphs: the fused kernel
  const int k = blockIdx.y;
  const int beg = w.rowptr[k], end = w.rowptr[k + 1];
  if (beg == end) return;
  const int deg_all = end - beg;
  const int P = a.P, V = prm.n_views, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p_raw = blockIdx.x * TILE + tid;
  const bool inb = p_raw < P;
  const int p = inb ? p_raw : P - 1;
  const int flags = w.fflags[k];
  const bool dfree = flags & 2;
  const int pose_i = k / V, qi = k % V;
  const int si = w.pose_slot[pose_i];
  const bool fi = si >= 0;
  const int n_free = w.info[0], nrow = w.info[3];
  const int foff = 6 * n_free;

  extern __shared__ __align__(16) float am_smem[];
  float* wbuf = am_smem + wave * WBUF;        // wave-private R1 tile
  float* acc1 = am_smem + NWAVE * WBUF;       // [WK_CH1][16][16] per-term Gram accumulators of the current chunk
  float* accI = acc1 + WK_CH1 * 256;          // [64] frame level: H_ii 36, v_i 6, H_if 6F, H_ff 3, v_f F
  TermGeomM* tg = reinterpret_cast<TermGeomM*>(accI + 64);
  // @kstamp 0
  if (tid < 64) accI[tid] = 0.0f;

  const cam::Intr Ii = cam::load_scaled(a.intr + qi * (4 + a.D), a.D, 1.0f / prm.intr_factor);
  const float u = (float)(p % prm.wd), v = (float)(p / prm.wd);
  const float d = a.disps[(int64_t)k * P + p];
  float X0, Y0, dX0[FF], dY0[FF];
  cam::iproj<CAM, F>(Ii, u, v, X0, Y0, dX0, dY0);
  float C = 0.f, wz = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0}, Efr[FF] = {};
  const int l16 = lane & 15, kq = lane >> 4;

  for (int cb = 0; cb < deg_all; cb += WK_CH1) {
    const int deg = min(WK_CH1, deg_all - cb);  // terms of this chunk
    // @kstampc 1
    __syncthreads();                            // the previous chunk's flush is done with acc1 / tg
    for (int i = tid; i < WK_CH1 * 256; i += TILE) acc1[i] = 0.0f;
    if (tid < deg) {
      TermGeomM m;
      TermGeom& g = m.g;
      term_setup(a, w.order[beg + cb + tid], g);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        float ec[6] = {0, 0, 0, 0, 0, 0}, col[6];
        ec[c] = 1.0f;
        adjT_apply(g.G, ec, col);
#pragma unroll
        for (int r = 0; r < 6; ++r) m.Mi[r * 6 + c] = (g.merge == 1 && r == c ? 1.0f : 0.0f) - col[r];
      }
      tg[tid] = m;
    }
    __syncthreads();
    // @kstampc 2

    // target / weight of the next tile's terms are fetched while the current tile is computed (the walk is otherwise
    // a chain of dependent global-load latencies: measured 5 us per tile)
    float2 nx_t[TPT], nx_w[TPT];
    auto prefetch = [&](int t0) {
#pragma unroll
      for (int uu = 0; uu < TPT; ++uu) {
        const int t = min(t0 + uu, deg - 1);
        load_tw(a, tg[t].g.e, p, P, nx_t[uu], nx_w[uu]);
      }
    };
    prefetch(0);
    for (int t0 = 0; t0 < deg; t0 += TPT) {
      float2 cur_t[TPT], cur_w[TPT];
#pragma unroll
      for (int uu = 0; uu < TPT; ++uu) { cur_t[uu] = nx_t[uu]; cur_w[uu] = nx_w[uu]; }
      if (t0 + TPT < deg) prefetch(t0 + TPT);
#pragma unroll
      for (int uu = 0; uu < TPT; ++uu) {
        const int t = t0 + uu;
        if (t >= deg) break;  // workgroup-uniform
        const TermGeom& G = tg[t].g;
        const int e = G.e;
        const float X = G.T.R[0] * X0 + G.T.R[1] * Y0 + G.T.R[2] + G.T.t[0] * d;
        const float Y = G.T.R[3] * X0 + G.T.R[4] * Y0 + G.T.R[5] + G.T.t[1] * d;
        const float Z = G.T.R[6] * X0 + G.T.R[7] * Y0 + G.T.R[8] + G.T.t[2] * d;
        float x, y, Jp[2][3], Jfj[2][FF];
        cam::proj<CAM, true, F>(G.Ij, X, Y, Z, x, y, Jp, Jfj);
        const float2 tgt = cur_t[uu], wg = cur_w[uu];
        const float val = valid_weight(a, Z, inb);  // geom.py:263, buffer.py:413
        const float wd2[2] = {val * wg.x, val * wg.y};                     // weights of the disparity system
        const float wc[2] = {G.merge == 2 ? 0.0f : wd2[0], G.merge == 2 ? 0.0f : wd2[1]};  // ... of the pose blocks
        const float rc[2] = {x - tgt.x, y - tgt.y};
        float Ja[3][6] = {{d, 0, 0, 0, Z, -Y}, {0, d, 0, -Z, 0, X}, {0, 0, d, Y, -X, 0}};
        if (G.rig_adj) {
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            float tmp[6];
            adjT_apply(G.Rr, Ja[r], tmp);
#pragma unroll
            for (int q = 0; q < 6; ++q) Ja[r][q] = tmp[q];
          }
        }
        const bool fj = G.sj >= 0;
        float Ejv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float Jj[6], Ji[6], Jf[FF];
#pragma unroll
          for (int q = 0; q < 6; ++q) Jj[q] = Jp[c][0] * Ja[0][q] + Jp[c][1] * Ja[1][q] + Jp[c][2] * Ja[2][q];
          const float Jz = Jp[c][0] * G.T.t[0] + Jp[c][1] * G.T.t[1] + Jp[c][2] * G.T.t[2];
          if constexpr (F > 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) {
              const float ax = G.T.R[0] * dX0[f] + G.T.R[1] * dY0[f];
              const float ay = G.T.R[3] * dX0[f] + G.T.R[4] * dY0[f];
              const float az = G.T.R[6] * dX0[f] + G.T.R[7] * dY0[f];
              Jf[f] = (Jp[c][0] * ax + Jp[c][1] * ay + Jp[c][2] * az + Jfj[c][f]) * (1.0f / prm.intr_factor);
            }
          }
          // R1 rows of this term and component
          const float sw = __builtin_amdgcn_sqrtf(wc[c]);  // v_sqrt_f32 (1 ulp): only splits w between the Gram factors
          float* col = wbuf + (uu * RPT) * AM_P1 + c * 64 + lane;
#pragma unroll
          for (int q = 0; q < 6; ++q) col[q * AM_P1] = Jj[q] * sw;
          col[6 * AM_P1] = rc[c] * sw;
          if constexpr (F > 0) {
#pragma unroll
            for (int f = 0; f < F; ++f) col[(7 + f) * AM_P1] = Jf[f] * sw;
          }
          // per-pixel disparity quantities
          if (dfree) {
            const float wJz = wc[c] * Jz;
            C += wd2[c] * Jz * Jz;
            wz -= wd2[c] * Jz * rc[c];
            float tmp[6];
            adjT_apply(G.G, Jj, tmp);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              Ji[q] = (G.merge == 1 ? Jj[q] : 0.0f) - tmp[q];
              Ei[q] += Ji[q] * wJz;
            }
            if constexpr (F > 0) {
#pragma unroll
              for (int f = 0; f < F; ++f) Efr[f] += Jf[f] * wJz;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) Ejv[q] += Jj[q] * wJz;
          }
        }
        if (dfree && fj && inb) {
#pragma unroll
          for (int q = 0; q < 6; ++q) w.Ej[((int64_t)e * 6 + q) * P + p] = Ejv[q];
        }
      }
      // ---- Gram matrix of the tile over this wave's 64 pixels x 2 components
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4m g4 = {0.f, 0.f, 0.f, 0.f};
      const float* arow = wbuf + l16 * AM_P1 + kq;
#pragma unroll 8
      for (int s = 0; s < 32; ++s) {
        const float av = arow[4 * s];
        g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, g4, 0, 0, 0);
      }
      // D[row = 4 kq + r][col = l16]: keep the diagonal RPT x RPT blocks
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        const int tb = row / RPT;
        if (l16 / RPT == tb && t0 + tb < deg) atomicAdd(&acc1[(t0 + tb) * 256 + (row % RPT) * 16 + (l16 % RPT)], g4[r]);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // @kstampc 3

    __syncthreads();
    // @kstampc 4
    // ---- per-term blocks from the Gram sums (one wave per term)
    for (int t = wave; t < deg; t += NWAVE) {
      const TermGeomM& TG = tg[t];
      const float* Gm = acc1 + t * 256;  // [16][16]: rows/cols 0..5 J, 6 r, 7.. Jf
      const float* Mi = TG.Mi;
      const int sj = TG.g.sj;
      const bool fj = sj >= 0;
      const int bj = 6 * sj, bi = 6 * si;
      float* T1 = wbuf;  // [6][6 + F] scratch: Mi Hjj | Mi Hjf
      if (lane < 36) {
        const int r = lane / 6, c = lane % 6;
        const float hjj = Gm[r * 16 + c];
        if (fj && r >= c) {
          s_add(w, bj + r, bj + c, (double)hjj);
          if (r == c) atomicAdd(&w.Hd[bj + r], (double)hjj);
        }
        float t1 = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) t1 += Mi[r * 6 + q] * Gm[q * 16 + c];
        T1[r * 8 + c] = t1;
        if (fi && fj) s_add(w, bi + r, bj + c, (double)t1);  // H_ij = Mi H_jj
      } else if (lane < 42) {
        const int q = lane - 36;
        const float vjn = Gm[q * 16 + 6];  // sum w J_q r  (v_j = -that)
        if (fj) atomicAdd(&w.S[(int64_t)nrow * w.ld + bj + q], -(double)vjn);
        if (fi) {
          float vin = 0.f;
#pragma unroll
          for (int c = 0; c < 6; ++c) vin += Mi[q * 6 + c] * Gm[c * 16 + 6];
          atomicAdd(&accI[36 + q], -vin);
        }
      } else if (F > 0 && lane < 42 + 6 * F) {
        const int q = (lane - 42) / FF, f = (lane - 42) % FF;
        const float hjf = Gm[q * 16 + 7 + f];
        if (fj) s_add(w, foff + f, bj + q, (double)hjf);
        if (fi) {
          float hif = 0.f;
#pragma unroll
          for (int c = 0; c < 6; ++c) hif += Mi[q * 6 + c] * Gm[c * 16 + 7 + f];
          atomicAdd(&accI[42 + q * FF + f], hif);
        }
      } else if (F > 0 && lane < 42 + 6 * F + F * F) {
        const int i2 = lane - 42 - 6 * F, f = i2 / FF, f2 = i2 % FF;
        if (f >= f2) atomicAdd(&accI[42 + 6 * FF + f * FF + f2], Gm[(7 + f) * 16 + 7 + f2]);
      } else if (F > 0 && lane < 42 + 6 * F + F * F + F) {
        const int f = lane - 42 - 6 * F - F * F;
        atomicAdd(&accI[42 + 6 * FF + FF * FF + f], -Gm[(7 + f) * 16 + 6]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (fi && lane < 36) {
        const int r = lane / 6, c = lane % 6;
        float hii = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) hii += T1[r * 8 + q] * Mi[c * 6 + q];
        atomicAdd(&accI[r * 6 + c], hii);
      }
      __builtin_amdgcn_wave_barrier();
    }
    // @kstampc 5

  }
  __syncthreads();
  // @kstamp 20

  // ---- frame level: H_ii, v_i, H_if, H_ff, v_f
  {
    const int bi = 6 * si;
    if (tid < 36) {
      const int r = tid / 6, c = tid % 6;
      if (fi && r >= c) {
        const double s = (double)accI[r * 6 + c];
        s_add(w, bi + r, bi + c, s);
        if (r == c) atomicAdd(&w.Hd[bi + r], s);
      }
    } else if (tid < 42) {
      if (fi) atomicAdd(&w.S[(int64_t)nrow * w.ld + bi + (tid - 36)], (double)accI[tid]);
    } else if (F > 0 && tid < 42 + 6 * F) {
      const int q = (tid - 42) / FF, f = (tid - 42) % FF;
      if (fi) s_add(w, foff + f, bi + q, (double)accI[tid]);
    } else if (F > 0 && tid < 42 + 6 * F + F * F) {
      const int i2 = tid - 42 - 6 * F, f = i2 / FF, f2 = i2 % FF;
      if (f >= f2) {
        const double s = (double)accI[tid];
        s_add(w, foff + f, foff + f2, s);
        if (f == f2) atomicAdd(&w.Hd[foff + f], s);
      }
    } else if (F > 0 && tid < 42 + 6 * F + F * F + F) {
      atomicAdd(&w.S[(int64_t)nrow * w.ld + foff + (tid - 42 - 6 * F - F * F)], (double)accI[tid]);
    }
  }
  // ---- finish the disparity block of this pixel: sensor prior, damping (terms.py:258-268, buffer.py:482-489)
  if (dfree) {
    const int64_t kp = (int64_t)k * P + p;
    finish_disp(a, k, p, P, flags, d, C, wz);
    if (inb) {
      w.C[kp] = C;
      w.wv[kp] = wz;
#pragma unroll
      for (int q = 0; q < 6; ++q) w.Ekk[((int64_t)k * 6 + q) * P + p] = Ei[q];
      if constexpr (F > 0) {
#pragma unroll
        for (int f = 0; f < F; ++f) w.Ef[((int64_t)k * 2 + f) * P + p] = Efr[f];
      }
    }
  }
  // @kstamp 21
}

// ---- multi-view rigs (n_views > 1, optionally the rig-rotation group): the general walk with per-term LOCAL variable
// blocks.  Every Jacobian of a term is a linear image of 6 + 2F "base" rows the walk forms per pixel:
//     base = [Jj (6: d r / d pose_j), JfA (F: intrinsics of the source view qi, plus the target's when qj == qi),
//             JfB (F: intrinsics of the target view qj when qj != qi)]
//     pose_i = Mi0 Jj   (Mi0 = -Adj(G_ij)^T, geom.py:277)      pose_j = Jj
//     rig_qi = -pose_i, rig_qj = -pose_j (geom.py:292-294)       intr_qi = JfA, intr_qj = JfB (terms.py:224-227)
// so ONE Gram matrix of [base; r] per term (matrix cores, as in ba_walk_kernel) gives every block of J^T W J through a
// small map Lm [28 local columns x 10 base rows] and a table gcol[28] of reduced-system columns (-1: fixed).  Local
// columns that land on the same unknown (cross-view self edges: pose_i = pose_j; qi == qj: one rig block) are summed by
// adding ALL ordered pairs (a, b) with gcol[a] >= gcol[b] - exactly (Ja + Jb)^T W (Ja + Jb), what the reference's block
// coalescing produces (matrix.py:124-177).  The E rows of the tail unknowns are accumulated per pixel in registers and
// left in the workspace (Et) for ba_schur_kernel / ba_retract_kernel.
constexpr int RG_NB = 10;  // base rows: Jj 0..5, JfA 6..7, JfB 8..9  (Gram tile rows 0..5, 7..8, 9..10; tile row 6 = r)
constexpr int RG_NL = 28;  // local columns: pose_i 0..5, pose_j 6..11, intr A 12..13, intr B 14..15, rig A 16..21, rig B 22..27
struct TermGeomR {
  TermGeom g;
  int qj, same_view;
  int gcol[RG_NL];
  float Lm[RG_NL][RG_NB];
};
constexpr size_t walk_rig_lds() {
  return sizeof(float) * (NWAVE * 16 * AM_P1 + WK_CH * 256) + WK_CH * sizeof(TermGeomR);
}
__device__ __forceinline__ int rg_tile_row(int b) { return b < 6 ? b : b + 1; }

template <int CAM, int VM>
__global__ __launch_bounds__(TILE) void ba_walk_rig_kernel(BAArgs a) {
  constexpr int WBUF = 16 * AM_P1;
  constexpr int F = CAM == VIPE_CAM_MEI ? 2 : 1;
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  const int k = blockIdx.y;
  const int beg = w.rowptr[k], end = w.rowptr[k + 1];
  if (beg == end) return;
  const int deg_all = end - beg;
  const int P = a.P, V = prm.n_views, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p_raw = blockIdx.x * TILE + tid;
  const bool inb = p_raw < P;
  const int p = inb ? p_raw : P - 1;
  const int flags = w.fflags[k];
  const bool dfree = flags & 2;
  const int pose_i = k / V, qi = k % V;
  const int si = w.pose_slot[pose_i];
  const int n_free = w.info[0], nrow = w.info[3];
  const int foff = 6 * n_free, roff = foff + a.nintr;
  const bool oi = prm.optimize_intrinsics, orr = prm.optimize_rig_rotation;

  extern __shared__ __align__(16) float am_smem[];
  float* wbuf = am_smem + wave * WBUF;   // wave-private R1 tile; reused as scratch by the flush
  float* acc1 = am_smem + NWAVE * WBUF;  // [WK_CH][16][16] per-term Gram accumulators of the current chunk
  TermGeomR* tg = reinterpret_cast<TermGeomR*>(acc1 + WK_CH * 256);

  const cam::Intr Ii = cam::load_scaled(a.intr + qi * (4 + a.D), a.D, 1.0f / prm.intr_factor);
  const float u = (float)(p % prm.wd), v = (float)(p / prm.wd);
  const float d = a.disps[(int64_t)k * P + p];
  float X0, Y0, dX0[F], dY0[F];
  cam::iproj<CAM, F>(Ii, u, v, X0, Y0, dX0, dY0);
  float C = 0.f, wz = 0.f, Ei[6] = {0, 0, 0, 0, 0, 0};
  float EfA[F] = {}, ErA[6] = {0, 0, 0, 0, 0, 0};      // tail rows of the source view qi
  float Efv[VM][F] = {}, Erv[VM][6] = {};              // ... of the target views (selected by predicate)
  const int l16 = lane & 15, kq = lane >> 4;

  for (int cb = 0; cb < deg_all; cb += WK_CH) {
    const int deg = min(WK_CH, deg_all - cb);
    __syncthreads();  // the previous chunk's flush is done with acc1 / tg
    for (int i = tid; i < WK_CH * 256; i += TILE) acc1[i] = 0.0f;
    if (tid < deg) {
      TermGeomR& m = tg[tid];
      const int e = w.order[beg + cb + tid];
      const int pi = (int)a.pi[e], pj = (int)a.pj[e], qj = (int)a.qj[e];
      term_transforms(a.poses, a.rig, pi, qi, pj, qj, m.g.T, m.g.G, m.g.Rr);
      m.g.Ij = cam::load_scaled(a.intr + qj * (4 + a.D), a.D, 1.0f / prm.intr_factor);
      m.g.e = e;
      m.g.merge = (pi == pj);
      m.g.rig_adj = !(m.g.Rr.t[0] == 0.f && m.g.Rr.t[1] == 0.f && m.g.Rr.t[2] == 0.f && m.g.Rr.R[0] == 1.f &&
                      m.g.Rr.R[4] == 1.f && m.g.Rr.R[8] == 1.f);
      m.g.sj = m.g.merge ? -1 : w.pose_slot[pj];  // E_j row of the Schur stack: absent when merged into pose i
      m.qj = qj;
      m.same_view = (qj == qi);
      const int sjj = w.pose_slot[pj];
      for (int c = 0; c < RG_NL; ++c)
        for (int b = 0; b < RG_NB; ++b) m.Lm[c][b] = 0.0f;
      for (int c = 0; c < 6; ++c) {
        float ec[6] = {0, 0, 0, 0, 0, 0}, col[6];
        ec[c] = 1.0f;
        adjT_apply(m.g.G, ec, col);
        for (int r = 0; r < 6; ++r) {
          m.Lm[r][c] = -col[r];       // pose_i = Mi0 Jj
          m.Lm[16 + r][c] = col[r];   // rig of view qi = -pose_i
        }
        m.Lm[6 + c][c] = 1.0f;        // pose_j
        m.Lm[22 + c][c] = -1.0f;      // rig of view qj = -pose_j
      }
      for (int f = 0; f < 2; ++f) {
        m.Lm[12 + f][6 + f] = 1.0f;
        m.Lm[14 + f][8 + f] = 1.0f;
      }
      for (int q = 0; q < 6; ++q) {
        m.gcol[q] = si >= 0 ? 6 * si + q : -1;
        m.gcol[6 + q] = sjj >= 0 ? 6 * sjj + q : -1;
        m.gcol[16 + q] = (orr && qi >= 1) ? roff + 6 * (qi - 1) + q : -1;
        m.gcol[22 + q] = (orr && qj >= 1) ? roff + 6 * (qj - 1) + q : -1;
      }
      for (int f = 0; f < 2; ++f) {
        m.gcol[12 + f] = (oi && f < F) ? foff + qi * F + f : -1;
        m.gcol[14 + f] = (oi && f < F && qj != qi) ? foff + qj * F + f : -1;
      }
    }
    __syncthreads();

    for (int t = 0; t < deg; ++t) {
      const TermGeom& G = tg[t].g;
      const int e = G.e, qj = tg[t].qj;
      const bool same = tg[t].same_view;
      const float X = G.T.R[0] * X0 + G.T.R[1] * Y0 + G.T.R[2] + G.T.t[0] * d;
      const float Y = G.T.R[3] * X0 + G.T.R[4] * Y0 + G.T.R[5] + G.T.t[1] * d;
      const float Z = G.T.R[6] * X0 + G.T.R[7] * Y0 + G.T.R[8] + G.T.t[2] * d;
      float x, y, Jp[2][3], Jfj[2][F];
      cam::proj<CAM, true, F>(G.Ij, X, Y, Z, x, y, Jp, Jfj);
      float2 tgt, wg;
      load_tw(a, e, p, P, tgt, wg);
      const float val = valid_weight(a, Z, inb);  // geom.py:263, buffer.py:413
      const float wc[2] = {val * wg.x, val * wg.y};
      const float rc[2] = {x - tgt.x, y - tgt.y};
      float Ja[3][6] = {{d, 0, 0, 0, Z, -Y}, {0, d, 0, -Z, 0, X}, {0, 0, d, Y, -X, 0}};
      if (G.rig_adj) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float tmp[6];
          adjT_apply(G.Rr, Ja[r], tmp);
#pragma unroll
          for (int q = 0; q < 6; ++q) Ja[r][q] = tmp[q];
        }
      }
      float Ejv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float Jj[6], JfA[F], JfB[F];
#pragma unroll
        for (int q = 0; q < 6; ++q) Jj[q] = Jp[c][0] * Ja[0][q] + Jp[c][1] * Ja[1][q] + Jp[c][2] * Ja[2][q];
        const float Jz = Jp[c][0] * G.T.t[0] + Jp[c][1] * G.T.t[1] + Jp[c][2] * G.T.t[2];
#pragma unroll
        for (int f = 0; f < F; ++f) {
          // Jfi = Jp . (R_T dX0/df) (geom.py:286-288), Jfj from the target camera; J_scale 1/8 (terms.py:224-227)
          const float ax = G.T.R[0] * dX0[f] + G.T.R[1] * dY0[f];
          const float ay = G.T.R[3] * dX0[f] + G.T.R[4] * dY0[f];
          const float az = G.T.R[6] * dX0[f] + G.T.R[7] * dY0[f];
          const float ji = (Jp[c][0] * ax + Jp[c][1] * ay + Jp[c][2] * az) * (1.0f / prm.intr_factor);
          const float jj = Jfj[c][f] * (1.0f / prm.intr_factor);
          JfA[f] = same ? ji + jj : ji;
          JfB[f] = same ? 0.0f : jj;
        }
        const float sw = __builtin_amdgcn_sqrtf(wc[c]);
        float* col = wbuf + c * 64 + lane;
#pragma unroll
        for (int q = 0; q < 6; ++q) col[q * AM_P1] = Jj[q] * sw;
        col[6 * AM_P1] = rc[c] * sw;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          col[(7 + f) * AM_P1] = f < F ? JfA[f < F ? f : 0] * sw : 0.0f;
          col[(9 + f) * AM_P1] = f < F ? JfB[f < F ? f : 0] * sw : 0.0f;
        }
#pragma unroll
        for (int r = 11; r < 16; ++r) col[r * AM_P1] = 0.0f;
        if (dfree) {
          const float wJz = wc[c] * Jz;
          C += wc[c] * Jz * Jz;
          wz -= wc[c] * Jz * rc[c];
          float tmp[6];
          adjT_apply(G.G, Jj, tmp);  // pose_i Jacobian before merging: -tmp
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            Ei[q] += ((G.merge ? Jj[q] : 0.0f) - tmp[q]) * wJz;
            ErA[q] += tmp[q] * wJz;          // rig of view qi: -(pose_i Jacobian)
            Ejv[q] += Jj[q] * wJz;
          }
#pragma unroll
          for (int f = 0; f < F; ++f) EfA[f] += JfA[f] * wJz;
#pragma unroll
          for (int vv = 0; vv < VM; ++vv) {
            if (vv == qj) {
#pragma unroll
              for (int f = 0; f < F; ++f) Efv[vv][f] += JfB[f] * wJz;
#pragma unroll
              for (int q = 0; q < 6; ++q) Erv[vv][q] -= Jj[q] * wJz;  // rig of view qj: -(pose_j Jacobian)
            }
          }
        }
      }
      if (dfree && G.sj >= 0 && inb) {
#pragma unroll
        for (int q = 0; q < 6; ++q) w.Ej[((int64_t)e * 6 + q) * P + p] = Ejv[q];
      }
      // ---- Gram matrix of [base; r] over this wave's 64 pixels x 2 components
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4m g4 = {0.f, 0.f, 0.f, 0.f};
      const float* arow = wbuf + l16 * AM_P1 + kq;
#pragma unroll 8
      for (int s2 = 0; s2 < 32; ++s2) {
        const float av = arow[4 * s2];
        g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, g4, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&acc1[t * 256 + (4 * kq + r) * 16 + l16], g4[r]);
      __builtin_amdgcn_wave_barrier();
    }

    __syncthreads();
    // ---- per-term blocks from the Gram sums (one wave per term): T1 = Lm Gb, H = T1 Lm^T, v = -Lm g
    for (int t = wave; t < deg; t += NWAVE) {
      const TermGeomR& TG = tg[t];
      const float* Gm = acc1 + t * 256;
      float* T1 = wbuf;  // [RG_NL][RG_NB]
      for (int i = lane; i < RG_NL * RG_NB; i += 64) {
        const int c = i / RG_NB, b = i % RG_NB;
        float acc = 0.f;
#pragma unroll
        for (int b2 = 0; b2 < RG_NB; ++b2) acc += TG.Lm[c][b2] * Gm[rg_tile_row(b2) * 16 + rg_tile_row(b)];
        T1[i] = acc;
      }
      if (lane < RG_NL && TG.gcol[lane] >= 0) {
        float acc = 0.f;
#pragma unroll
        for (int b2 = 0; b2 < RG_NB; ++b2) acc += TG.Lm[lane][b2] * Gm[rg_tile_row(b2) * 16 + 6];
        atomicAdd(&w.S[(int64_t)nrow * w.ld + TG.gcol[lane]], -(double)acc);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < RG_NL * RG_NL; i += 64) {
        const int ca = i / RG_NL, cb2 = i % RG_NL;
        const int ga = TG.gcol[ca], gb = TG.gcol[cb2];
        if (ga < 0 || gb < 0 || ga < gb) continue;
        float acc = 0.f;
#pragma unroll
        for (int b = 0; b < RG_NB; ++b) acc += T1[ca * RG_NB + b] * TG.Lm[cb2][b];
        atomicAdd(&w.S[(int64_t)ga * w.ld + gb], (double)acc);
        if (ga == gb) atomicAdd(&w.Hd[ga], (double)acc);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }

  // ---- finish the disparity block of this pixel and leave the E rows for the Schur / back-substitution kernels
  if (dfree) {
    const int64_t kp = (int64_t)k * P + p;
    finish_disp(a, k, p, P, flags, d, C, wz);
    if (inb) {
      w.C[kp] = C;
      w.wv[kp] = wz;
#pragma unroll
      for (int q = 0; q < 6; ++q) w.Ekk[((int64_t)k * 6 + q) * P + p] = Ei[q];
      float* et = w.Et + (int64_t)k * a.ntail * P + p;
      if (oi) {
#pragma unroll
        for (int vv = 0; vv < VM; ++vv) {
          if (vv < V) {
#pragma unroll
            for (int f = 0; f < F; ++f) et[(int64_t)(vv * F + f) * P] = Efv[vv][f] + (vv == qi ? EfA[f] : 0.0f);
          }
        }
      }
      if (orr) {
#pragma unroll
        for (int vv = 1; vv < VM; ++vv) {
          if (vv < V) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
              et[(int64_t)(a.nintr + 6 * (vv - 1) + q) * P] = Erv[vv][q] + (vv == qi ? ErA[q] : 0.0f);
          }
        }
      }
    }
  }
}

// Schur complement of one source frame from the E rows in the workspace (general path, after ba_walk_kernel):
// rows = sqrt(Q) * [E_kk (pose i); E_j of every term; E_f; w], Gram over all P pixels, one workgroup per 16 x 16 tile
// pair of the lower triangle.  Each wave takes every fourth 64-pixel chunk: the two row tiles are staged in LDS
// (coalesced 256-byte row segments), 16 v_mfma_f32_16x16x4_f32 per chunk, partial tiles summed through LDS.
constexpr int SC_GRID = 96;  // tile pairs processed in parallel per frame (the kernel strides over the rest; idle blocks exit)

template <int F>
__global__ __launch_bounds__(TILE) void ba_schur_kernel(BAArgs a) {
  const BAWs& w = a.w;
  if (!a.mv && !a.force_general && w.info[6] <= AM_DMAX) return;
  const int k = blockIdx.y;
  const int flags = w.fflags[k];
  const int beg = w.rowptr[k], end = w.rowptr[k + 1];
  if (!(flags & 2) || beg == end) return;
  const int deg = end - beg, P = a.P, V = a.p.n_views;
  const int NT = a.mv ? a.ntail : F;  // tail rows: per-view intrinsics + rig rotations (Et), or the shared intrinsics (Ef)
  const int NR = 6 * (deg + 1) + NT + 1, RT = (NR + 15) >> 4, npairs = RT * (RT + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kq = lane >> 4;
  const int si = w.pose_slot[k / V];
  const int n_free = w.info[0], nrow = w.info[3], foff = 6 * n_free;
  __shared__ float tile[NWAVE][2][16 * AM_P2];
  __shared__ float red[NWAVE][256];
  __shared__ const float* rowp[32];
  __shared__ float rowm[32];  // 1 for a live row, 0 for an absent one (its pointer then aims at valid memory: loads stay unconditional)
  __shared__ int rowg[32];
  // source row r of the stacked E matrix: pointer to its P values (or null) and its index in the reduced system
  auto resolve = [&](int r, const float*& ptr, int& g) {
    ptr = nullptr; g = -1;
    if (r >= NR) return;
    if (r == NR - 1) { ptr = w.wv + (int64_t)k * P; g = -2; return; }
    if (r >= 6 * (deg + 1)) {
      const int f = r - 6 * (deg + 1);
      ptr = a.mv ? w.Et + ((int64_t)k * NT + f) * P : w.Ef + ((int64_t)k * 2 + f) * P;
      g = foff + f;
      return;
    }
    const int m = r / 6, q = r % 6;
    if (m == 0) {
      if (si >= 0) { ptr = w.Ekk + ((int64_t)k * 6 + q) * P; g = 6 * si + q; }
    } else {
      const int e = w.order[beg + m - 1];
      const int pj = (int)a.pj[e];
      const int sj = ((int)a.pi[e] == pj) ? -1 : w.pose_slot[pj];
      if (sj >= 0) { ptr = w.Ej + (int64_t)e * 6 * P + (int64_t)q * P; g = 6 * sj + q; }
    }
  };
  const float* Ck = w.C + (int64_t)k * P;
  const int nchunks = (P + 63) / 64;
  // A frame with few tile pairs (the keyframe frontend: <= 28) leaves most of the grid's blocks without one: the pixel
  // range of every pair is then cut into `slices` pieces, one workgroup each (the partial Grams meet in the fp64 atomics
  // below) - a workgroup's 12 dependent chunk iterations per wave were the kernel's whole duration (42 us)
  const int slices = max(1, min((int)gridDim.x / npairs, nchunks / NWAVE));
  for (int item = blockIdx.x; item < npairs * slices; item += gridDim.x) {
    const int pid = item / slices, slc = item % slices;
    const int ch0 = (int)((int64_t)nchunks * slc / slices), ch1 = (int)((int64_t)nchunks * (slc + 1) / slices);
    int ta = 0;
    while ((ta + 1) * (ta + 2) / 2 <= pid) ++ta;
    const int tb = pid - ta * (ta + 1) / 2;
    __syncthreads();
    if (tid < 32) {
      const float* ptr; int g;
      resolve(16 * (tid < 16 ? ta : tb) + (tid & 15), ptr, g);
      rowp[tid] = ptr ? ptr : (w.C + (int64_t)k * P); rowm[tid] = ptr ? 1.0f : 0.0f; rowg[tid] = g;
    }
    __syncthreads();
    float4m g4 = {0.f, 0.f, 0.f, 0.f};
    // the rows of the next chunk are fetched while the current chunk's MFMAs run
    float va[16], vb[16], na[16], nb[16];
    auto fetch = [&](int ch, float (&da)[16], float (&db)[16]) {
      const int px = ch * 64 + lane;
      const bool ok = px < P;
      const int pxc = ok ? px : 0;
      const float sq = ok ? __builtin_amdgcn_rsqf(Ck[pxc]) : 0.0f;
      // all 32 row loads are unconditional (absent rows read a valid dummy row and are scaled by 0), so they are in
      // flight together instead of one L2 round trip per guarded load
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        da[r] = rowp[r][pxc];
        db[r] = rowp[16 + r][pxc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        da[r] *= sq * rowm[r];
        db[r] *= sq * rowm[16 + r];
      }
    };
    if (ch0 + wave < ch1) fetch(ch0 + wave, na, nb);
    for (int ch = ch0 + wave; ch < ch1; ch += NWAVE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { va[r] = na[r]; vb[r] = nb[r]; }
      if (ch + NWAVE < ch1) fetch(ch + NWAVE, na, nb);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tile[wave][0][r * AM_P2 + lane] = va[r];
        tile[wave][1][r * AM_P2 + lane] = vb[r];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float* ar = &tile[wave][0][l16 * AM_P2 + kq];
      const float* br = &tile[wave][1][l16 * AM_P2 + kq];
#pragma unroll 8
      for (int s2 = 0; s2 < 16; ++s2) g4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[4 * s2], br[4 * s2], g4, 0, 0, 0);
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * kq + r) * 16 + l16] = g4[r];
    __syncthreads();
    {
      const float val = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      const int ra = tid >> 4, cb = tid & 15;
      const int row = 16 * ta + ra, cc = 16 * tb + cb;
      const int gr = rowg[ra], gc = rowg[16 + cb];
      if (row < NR && cc <= row && cc != NR - 1 && gc >= 0) {
        if (row == NR - 1) atomicAdd(&w.S[(int64_t)nrow * w.ld + gc], -(double)val);
        else if (gr >= 0) s_add(w, gr, gc, -(double)val);
      }
    }
  }
}

// 1/sqrt(x) in fp64: hardware estimate + 2 Newton steps (avoids the long sqrt / divide sequences on the
// factorisation's critical path)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  r = r * __builtin_fma(-hx * r, r, 1.5);
  r = r * __builtin_fma(-hx * r, r, 1.5);
  return r;
}

// ------------------------------------------------------------------------------------------------ solve (LDS band)

// Retraction shared by both solve kernels: poses X <- Exp(dx) X (retractor.py:27-29), intrinsics (retractor.py:50-62)
__device__ __forceinline__ void apply_retraction(const BAArgs& a, int t, int nthreads, int n_free) {
  const BAWs& w = a.w;
  for (int sl = t; sl < n_free; sl += nthreads) {
    const int pidx = w.slot_pose[sl];
    float xi[6];
    for (int q = 0; q < 6; ++q) xi[q] = w.dx[6 * sl + q];
    lie::SE3<float> X(a.poses + 7 * pidx);
    (lie::SE3<float>::exp(xi) * X).store(a.poses + 7 * pidx);
  }
  if (a.mv) {
    // one intrinsics block per view (retractor.py:50-62 with len(dx) == V) and one rotation-only step per view >= 1
    // (retractor.py:32-37: the translation part of the tangent is zeroed, X <- Exp([0, phi]) X)
    const int F = 1 + a.D, V = a.p.n_views;
    if (a.p.optimize_intrinsics && t < V) {
      float* I = a.intr + t * (4 + a.D);
      const float df = w.dx[6 * n_free + t * F];
      I[0] += df; I[1] += df;
      if (F > 1) I[4] += 0.01f * w.dx[6 * n_free + t * F + 1];
    }
    if (a.p.optimize_rig_rotation && t >= 1 && t < V) {
      float xi[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int q = 3; q < 6; ++q) xi[q] = w.dx[6 * n_free + a.nintr + 6 * (t - 1) + q];
      lie::SE3<float> X(a.rig + 7 * t);
      (lie::SE3<float>::exp(xi) * X).store(a.rig + 7 * t);
    }
  } else if (a.p.optimize_intrinsics && t == 0) {
    const int F = 1 + a.D;
    const float df = w.dx[6 * n_free];
    for (int vq = 0; vq < a.p.n_views; ++vq) {
      float* I = a.intr + vq * (4 + a.D);
      if (I[0] > 0) { I[0] += df; I[1] += df; if (F > 1) I[4] += 0.01f * w.dx[6 * n_free + 1]; }
    }
  }
}

// When the reduced system is banded (sliding-window / neighbourhood graphs: two poses couple only through a shared
// source frame) and its band fits the 160 KB of LDS, the whole factorisation runs out of LDS: every dependent step
// then costs an LDS round trip (~100 cycles) instead of an L2 round trip (~2000), which is what bounds this
// latency-critical kernel.  Storage: pose row r keeps columns [6 (r/6 - bandblk), 6 (r/6) + 5] at pitch WB + 1
// doubles (so the LDS address of (row j0 + 6 + i, column j0 + m) advances by a constant per block step and every
// thread's operand addresses are computed once); the dense tail rows (intrinsics, then the rhs) are kept full
// length.  Block size 6 = one pose.  Per block step: panel (one row per lane), barrier, trailing update (one
// (row, row) pair per lane, descriptors fixed over the steps) while lane 0 of wave 0 - whose wave owns the 21
// entries of the NEXT diagonal block - already factors that block (look-ahead), barrier.  The back substitution is
// run by wave 0 alone: 8 lanes per column, DPP reductions, the 6x6 triangular solve replicated in every lane with
// stored reciprocal pivots - no workgroup barrier and no division on the dependent chain.
// Sets info[5] = 1 when it solved the system (the global-memory kernel launched after it then exits).
constexpr int BAND_T = 512;
constexpr int BAND_UPT = 6;  // trailing-update pairs per thread: dense 12-pose windows (PB = 66: 2277 pairs) still fit

#define VIPE_DPP_F64(v, ctrl)                                                                                    \
  __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(                         \
                                  0, (int)(__builtin_bit_cast(unsigned long long, v) >> 32), ctrl, 0xf, 0xf, true) \
                              << 32) |                                                                           \
                                 (unsigned)__builtin_amdgcn_update_dpp(                                          \
                                     0, (int)__builtin_bit_cast(unsigned long long, v), ctrl, 0xf, 0xf, true))

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// UPT_: trailing-update pair slots per thread, TWO: two band columns per lane (bands wider than one wave).  The
// neighbourhood graphs of the headline configuration run the <2, false> instantiation (fewer slots to walk per step, one
// pipelined load pass); the frontend's dense windows the <BAND_UPT, true> one.
template <int UPT_, bool TWO>
__device__ __forceinline__ void band_solve_body(const BAArgs& a, int lds_doubles, unsigned char* smem_raw) {
  const int t = threadIdx.x;
  double* const L = reinterpret_cast<double*>(smem_raw);
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  const int n = w.info[3], n_free = w.info[0], bandblk = w.info[4];
  const int ld = w.ld;
  const int npr = 6 * n_free, ntail = n - npr + 1;  // tail rows: intrinsics rows, then the rhs row
  const int F = ntail - 1;
  const int PB = 6 * bandblk, WB = PB + 6, WBP = WB + 1, KS = 6 * WBP;
  const int npp = PB * (PB + 1) / 2, ntp = ntail * PB, ntt = F == 0 ? 0 : (F == 1 ? 2 : 5);
  const int npair = npp + ntp + ntt;
  // LDS carve: band [npr][WBP], tail [ntail][n + 1], rdall [npr], blk[6][7], rd[6], flags
  const int TL0 = npr * WBP;
  const int RD0 = TL0 + ntail * (n + 1);
  const int need = RD0 + npr + 64;
  if (t == 0) w.info[5] = 0;
  if (n == 0 || need > lds_doubles || npair > 21 + UPT_ * (BAND_T - 64) || PB + ntail > BAND_T || WBP > (TWO ? 128 : 64) || F > 2) {
    return;
  }
  double* const Tl = L + TL0;
  double* const rdall = L + RD0;          // 1 / L[r][r] of the pose rows
  double* const blk = rdall + npr;        // 6x7: the current diagonal factor block
  double* const rd = blk + 42;            // its reciprocal pivots
  int* const failp = reinterpret_cast<int*>(rd + 6);
  const double* S = w.S;
  auto bofs = [&](int r, int c) { return r * WBP + c - 6 * (r / 6) + PB; };
  auto tref = [&](int q, int c) -> double& { return Tl[q * (n + 1) + c]; };

  // ---- load (with LM damping on the diagonal, matrix.py:179-186)
  if (t == 0) *failp = 0;
  {
    // one band row (WB <= 64 doubles, contiguous in S) per wave and iteration; unrolled so that a dozen row loads are
    // in flight per wave (the loop is otherwise one L2 round trip per row)
    // Sixteen rows per pass with every load of the pass - matrix entries AND the damping diagonal - issued from clamped
    // addresses before the first use: as `load; if (diagonal) load Hd; store` the loop was one memory round trip per row
    // (the second load depends on a branch on the first; stamps: 34.7k cycles to load the headline's two images).
    // The row is wave-uniform: its block arithmetic runs on the scalar unit (as per-entry divisions it was the loop's bulk).
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6), ln = t & 63;
    constexpr int CH = 16, NC = TWO ? 2 : 1;
    const bool dr = a.droid;
    const double dep = (double)prm.pose_ep, ddm = (double)prm.pose_damping;
    for (int r0 = wv; r0 < npr; r0 += CH * (BAND_T / 64)) {
      double sv[CH][NC], hv[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int r = min(r0 + i * (BAND_T / 64), npr - 1), rb6 = 6 * (r / 6);
        hv[i] = dr ? 0.0 : w.Hd[r];
#pragma unroll
        for (int h = 0; h < NC; ++h) sv[i][h] = S[(int64_t)r * ld + min(max(rb6 - PB + ln + 64 * h, 0), r)];
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int r = r0 + i * (BAND_T / 64);
        if (r < npr) {
          const int rb6 = 6 * (r / 6), rm = r - rb6;
#pragma unroll
          for (int h = 0; h < NC; ++h) {
            const int l2 = ln + 64 * h;
            double v = (l2 < WB && rb6 + l2 >= PB && l2 <= rm + PB) ? sv[i][h] : 0.0;
            if (l2 == rm + PB) v += dep + ddm * (dr ? v : hv[i]);  // DROID: geom_kernels.cu:1176
            if (l2 < WBP) L[r * WBP + l2] = v;
          }
        }
      }
    }
  }
  for (int idx = t; idx < ntail * (n + 1); idx += BAND_T) {
    const int q = idx / (n + 1), c = idx % (n + 1), r = npr + q;
    double v = 0.0;
    if (c <= r && c < n) {
      v = S[(int64_t)r * ld + c];
      if (c == r) v += 1e-6 + 1e-6 * w.Hd[r];
    }
    Tl[idx] = v;
  }

  // ---- per-thread operand descriptors, fixed over the block steps (offsets in doubles from L at step 0 + stride)
  // panel: thread pr < PB + ntail owns one row below the diagonal block
  int prow_off = 0, prow_str = 0, prow_ia = -1;  // prow_ia >= 0: pose row offset (valid while 6 kb + 6 + ia < npr)
  const bool has_prow = t < PB + ntail;
  if (has_prow) {
    if (t < PB) { prow_ia = t; prow_off = (6 + t) * WBP - 6 - 6 * (t / 6) + PB; prow_str = KS; }
    else { prow_off = TL0 + (t - PB) * (n + 1); prow_str = 6; }
  }
  // update: pair index t + BAND_T * slot (slot < UPT) -> one (a, b) pair, b <= a: pose-pose, tail-pose, tail-tail
  constexpr int UPT = UPT_;
  int uA[UPT], uB[UPT], uD[UPT], sA[UPT], sB[UPT], sD[UPT], u_ia[UPT], u_ib[UPT];
  bool has_pair[UPT];
#pragma unroll
  for (int sl = 0; sl < UPT; ++sl) {
    // pairs 0..20 (the next diagonal block) are lanes 0..20 of wave 0 and nothing else runs there: that wave goes
    // straight on to factor the next block; the other pairs are spread over waves 1..7
    const int pid = t < 64 ? (sl == 0 && t < 21 ? t : npair) : 21 + (t - 64) + (BAND_T - 64) * sl;
    has_pair[sl] = pid < npair;
    uA[sl] = uB[sl] = uD[sl] = sA[sl] = sB[sl] = sD[sl] = 0;
    u_ia[sl] = u_ib[sl] = -1;
    if (!has_pair[sl]) continue;
    if (pid < npp) {
      int ia = (int)((sqrtf(8.0f * (float)pid + 1.0f) - 1.0f) * 0.5f);
      while ((ia + 1) * (ia + 2) / 2 <= pid) ++ia;
      while (ia * (ia + 1) / 2 > pid) --ia;
      const int ib = pid - ia * (ia + 1) / 2;
      u_ia[sl] = ia; u_ib[sl] = ib;
      uA[sl] = (6 + ia) * WBP - 6 - 6 * (ia / 6) + PB; sA[sl] = KS;
      uB[sl] = (6 + ib) * WBP - 6 - 6 * (ib / 6) + PB; sB[sl] = KS;
      uD[sl] = (6 + ia) * WBP + ib - 6 * (ia / 6) + PB; sD[sl] = KS;
    } else if (pid < npp + ntp) {
      const int u = pid - npp, q = u / PB, ib = u % PB;
      u_ib[sl] = ib;
      uA[sl] = TL0 + q * (n + 1); sA[sl] = 6;
      uB[sl] = (6 + ib) * WBP - 6 - 6 * (ib / 6) + PB; sB[sl] = KS;
      uD[sl] = TL0 + q * (n + 1) + 6 + ib; sD[sl] = 6;
    } else {
      // (q, q2), q2 <= q, q2 < ntail - 1 (the rhs row has no column): F = 1: (0,0) (1,0); F = 2: (0,0) (1,0) (1,1) (2,0) (2,1)
      const int v = pid - npp - ntp;
      int q, q2;
      if (F == 1) { q = v; q2 = 0; }
      else { q = v == 0 ? 0 : (v <= 2 ? 1 : 2); q2 = v == 0 ? 0 : (v <= 2 ? v - 1 : v - 3); }
      uA[sl] = TL0 + q * (n + 1); sA[sl] = 6;
      uB[sl] = TL0 + q2 * (n + 1); sB[sl] = 6;
      uD[sl] = TL0 + q * (n + 1) + npr + q2; sD[sl] = 0;
    }
  }
  // 6x6 diagonal block of pose kb: factor in registers, publish L (band), blk, rd, rdall
  auto factor_diag = [&](int kb) {
    const int j0 = 6 * kb;
    double* Dk = L + bofs(j0, j0);  // row i of the block at Dk + i * WBP
    double A[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) A[i][j] = Dk[i * WBP + j];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = A[j][j];
#pragma unroll
      for (int m = 0; m < j; ++m) d = __builtin_fma(-(A[j][m]), A[j][m], d);
      if (!(d > 0.0)) { *failp = 1; d = 1.0; }
      const double rl = rsqrt_nr(d);
      A[j][j] = d * rl;
      rd[j] = rl;
      rdall[j0 + j] = rl;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double sacc = A[i][j];
#pragma unroll
        for (int m = 0; m < j; ++m) sacc = __builtin_fma(-(A[i][m]), A[j][m], sacc);
        A[i][j] = sacc * rl;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) { Dk[i * WBP + j] = A[i][j]; blk[i * 7 + j] = A[i][j]; }
  };
  __syncthreads();
  if (t == 0 && n_free > 0) factor_diag(0);
  __syncthreads();

  // ---- factorisation, one pose block (6 columns) per step
  for (int kb = 0; kb < n_free; ++kb) {
    const int j0 = 6 * kb;
    // panel: x = a Lkk^-T for every row below the block (pose rows inside the band, all tail rows)
    if (has_prow && (prow_ia < 0 || j0 + 6 + prow_ia < npr)) {
      double* row = L + prow_off + kb * prow_str;
      double x[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double sacc = row[j];
#pragma unroll
        for (int m = 0; m < j; ++m) sacc = __builtin_fma(-(x[m]), blk[j * 7 + m], sacc);
        x[j] = sacc * rd[j];
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) row[j] = x[j];
    }
    __syncthreads();
    // trailing update, one (a, b) pair per thread
#pragma unroll
    for (int sl = 0; sl < UPT; ++sl) {
      if (has_pair[sl] && (u_ia[sl] < 0 || j0 + 6 + u_ia[sl] < npr) && (u_ib[sl] < 0 || j0 + 6 + u_ib[sl] < npr)) {
        const double* pa = L + uA[sl] + kb * sA[sl];
        const double* pb = L + uB[sl] + kb * sB[sl];
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) sacc = __builtin_fma(pa[m], pb[m], sacc);
        L[uD[sl] + kb * sD[sl]] -= sacc;
      }
    }
    // look-ahead: the 21 entries of the next diagonal block are pairs 0..20, all in wave 0
    if (t < 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (t == 0 && kb + 1 < n_free) factor_diag(kb + 1);
    }
    __syncthreads();
  }
  // ---- tail columns (intrinsics unknowns), unblocked
  for (int f = 0; f < F; ++f) {
    const int cf = npr + f;
    if (t == 0) {
      double d = tref(f, cf);
      if (!(d > 0.0)) { *failp = 1; d = 1.0; }
      const double rl = rsqrt_nr(d);
      tref(f, cf) = d * rl;
      for (int q = f + 1; q < ntail; ++q) tref(q, cf) *= rl;
      for (int q = f + 1; q < ntail; ++q)
        for (int q2 = f + 1; q2 <= q && q2 < ntail - 1; ++q2) tref(q, npr + q2) -= tref(q, cf) * tref(q2, cf);
    }
    __syncthreads();
  }
  // ---- back substitution L^T x = y; y = rhs row (tail row F), solved in place by wave 0
  double* y = &tref(F, 0);
  if (t < 64) {
    // column-oriented: once x of block kb is known (6x6 triangular solve, replicated in every lane from broadcast
    // LDS reads, reciprocal pivots), lane c subtracts its contribution from y of band row 6 kb - PB + c right away,
    // so no reduction and no cross-lane traffic sits on the dependent chain
    if (t == 0) {
      for (int f = F - 1; f >= 0; --f) {
        double sacc = y[npr + f];
        for (int q = f + 1; q < F; ++q) sacc -= tref(q, npr + f) * y[npr + q];
        y[npr + f] = sacc / tref(f, npr + f);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (F > 0) {
      for (int r = t; r < npr; r += 64) {
        double sacc = y[r];
        for (int f = 0; f < F; ++f) sacc -= tref(f, r) * y[npr + f];
        y[r] = sacc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    for (int kb = n_free - 1; kb >= 0; --kb) {
      const int j0 = 6 * kb;
      const double* Dk = L + bofs(j0, j0);
      // operands that do not depend on the running y: block factor, reciprocal pivots, this lane's column of the
      // six block rows (entries L[j0 + j][6 kb - PB + t])
      double Lk[6][6], rp[6], lc[6], lc2[6];
      const int rt = j0 - PB + t, rt2 = rt + 64;  // second column for bands wider than one wave (PB > 64)
      const bool upd = t < PB && rt >= 0;
      const bool upd2 = TWO && t + 64 < PB && rt2 >= 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        rp[i] = rdall[j0 + i];
        lc[i] = upd ? L[(j0 + i) * WBP + t] : 0.0;
        if constexpr (TWO) lc2[i] = upd2 ? L[(j0 + i) * WBP + t + 64] : 0.0;
        else lc2[i] = 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) Lk[i][j] = Dk[i * WBP + j];
      }
      double x[6];
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        double sacc = y[j0 + j];
#pragma unroll
        for (int m = 5; m > j; --m) sacc = __builtin_fma(-(Lk[m][j]), x[m], sacc);
        x[j] = sacc * rp[j];
      }
      if (t < 6) {
        double xo = x[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) xo = t == j ? x[j] : xo;
        y[j0 + t] = xo;
      }
      if (upd) {
        double sacc = lc[0] * x[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) sacc = __builtin_fma(lc[j], x[j], sacc);
        y[rt] -= sacc;
      }
      if (TWO && upd2) {
        double sacc = lc2[0] * x[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) sacc = __builtin_fma(lc2[j], x[j], sacc);
        y[rt2] -= sacc;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  const bool bad = *failp != 0;
  if (t == 0) {
    if (bad) w.info[2] += 1;
    w.info[5] = 1;
  }
  for (int dd = t; dd < n; dd += BAND_T) {
    double x = y[dd];
    if (bad || !(x == x)) x = 0.0;
    w.dx[dd] = (float)x;
  }
  __syncthreads();
  apply_retraction(a, t, BAND_T, n_free);
}

// Two-chain ("burn at both ends") form of the band solve for pose-only systems (no intrinsics columns): the sequential
// pivot chain, not arithmetic, bounds the kernel (47 dependent block steps at N = 48), and a block-banded SPD matrix can be
// eliminated from BOTH ends at once.  Blocks 0..a-1 (chain A, natural order) and blocks nb-1..a+bandblk (chain B,
// REVERSED order - the mirrored matrix is banded too, its lower triangle being the transposed upper one) are factorised
// concurrently by the two halves of the workgroup, each on its own LDS band image with the same code as above; the
// bandblk separator blocks in the middle are ordinary band rows at the end of both images, so both eliminations leave
// their Schur contributions in them.  Chain B's are then added to chain A's image, chain A factors the separator (bandblk
// more steps), and the back substitution runs separator first, then both chains at once.  Dependent block steps:
// max(a, b) + bandblk forward, bandblk + max(a, b) back, instead of nb + nb.
constexpr int B2_T = BAND_T;  // threads per chain: the kernel is launched with 2 * BAND_T threads, the one-chain forms use the first BAND_T

__device__ __forceinline__ bool band2_solve_body(const BAArgs& a, int lds_doubles, unsigned char* smem_raw) {
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  // image B's threads are rotated by one wave: its critical wave (tl < 64: diagonal pairs, look-ahead factor, back
  // substitution) is then hardware wave 9 - another SIMD than image A's wave 0 (waves go to SIMDs round robin); with both
  // chains' critical waves on one SIMD each ran at the pace of two
  const int t = threadIdx.x, g = t >= B2_T ? 1 : 0, tl = g ? ((t - 64) & (B2_T - 1)) : t;
  const int n = w.info[3], nb = w.info[0], bandblk = w.info[4];
  const int ld = w.ld;
  if (n != 6 * nb || bandblk < 1 || nb < 4 * bandblk + 4) return false;
  const int PB = 6 * bandblk, WB = PB + 6, WBP = WB + 1, KS = 6 * WBP;
  if (PB + 7 > 64) return false;
  const int ca = (nb - bandblk) / 2, cb = nb - bandblk - ca;  // chain lengths in blocks (cb >= ca)
  const int nf = (g ? cb : ca) + bandblk, chain = g ? cb : ca;  // blocks of this image, of its chain
  const int nfmax = cb + bandblk;
  const int npr = 6 * nf, nprmax = 6 * nfmax;
  const int npp = PB * (PB + 1) / 2, npair = npp + PB;
  if (npair > 21 + 2 * (B2_T - 64)) return false;
  // LDS carve per image: band [nprmax][WBP], rhs row [nprmax + 1], rdall [nprmax], blk 42, rd 6; then one flag
  const int IMG = nprmax * WBP + (nprmax + 1) + nprmax + 48;
  if (2 * IMG + 8 > lds_doubles) return false;
  double* const L = reinterpret_cast<double*>(smem_raw) + g * IMG;
  double* const y = L + nprmax * WBP;  // the image's right-hand side (a tail row of the factorisation)
  double* const rdall = y + nprmax + 1;
  double* const blk = rdall + nprmax;
  double* const rd = blk + 42;
  int* const failp = reinterpret_cast<int*>(reinterpret_cast<double*>(smem_raw) + 2 * IMG);
  double* const LA = reinterpret_cast<double*>(smem_raw);
  double* const LB = LA + IMG;
  const double* S = w.S;
  auto gblk = [&](int l) { return g ? nb - 1 - l : l; };  // local block -> global block
  // @bstamp 0
  // @bwave 100
  if (t == 0) { *failp = 0; w.info[5] = 0; }

  // ---- load both images (LM damping on the diagonal, matrix.py:179-186).  Image B: mirrored; its separator square and
  // separator right-hand side start from zero (they only collect chain B's contributions)
  {
    // Sixteen rows per pass, every load of the pass in flight before the first use (see band_solve_body), and NO per-entry
    // index arithmetic: the row is wave-uniform (scalar unit), what depends on the lane is a constant of the lane.  The
    // first form of this loop spent its time issuing integer divisions - four waves per SIMD, 14k .. 38k cycles by wave
    // (stamps).  Image A reads its band rows as they lie in S.  Image B is the MIRRORED matrix (its band row is a column
    // of S): it walks the rows of S as well and scatters every entry to its mirrored position (zero-filled first).
    const int wv = __builtin_amdgcn_readfirstlane(tl >> 6), ln = tl & 63;
    const int lq = ln / 6, lm = ln - 6 * lq;
    constexpr int CH = 20;  // the headline's images: 19 rows per wave, one pass = one memory round trip
    const bool dr = a.droid;
    const double dep = (double)prm.pose_ep, ddm = (double)prm.pose_damping;
    if (g)
      for (int idx = tl; idx < npr * WBP; idx += B2_T) L[idx] = 0.0;
    __syncthreads();
    if (!g) {
      for (int r0 = wv; r0 < npr; r0 += CH * (B2_T / 64)) {
        double sv[CH], hv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int r = min(r0 + i * (B2_T / 64), npr - 1), rb6 = 6 * (r / 6);
          sv[i] = S[(int64_t)r * ld + min(max(rb6 - PB + ln, 0), r)];
          hv[i] = dr ? 0.0 : w.Hd[r];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int r = r0 + i * (B2_T / 64);
          if (r < npr) {
            const int rb6 = 6 * (r / 6), rm = r - rb6;
            double v = (ln < WB && rb6 + ln >= PB && ln <= rm + PB) ? sv[i] : 0.0;
            if (ln == rm + PB) v += dep + ddm * (dr ? v : hv[i]);
            if (ln < WBP) L[r * WBP + ln] = v;
          }
        }
      }
    } else {
      const int G0 = 6 * (nb - nf);  // first global row of image B's blocks
      // target offset of lane ln's entry of global row Cg (block bcl = nb - 1 - Cg / 6 locally, cm = Cg % 6):
      //   own block (ln >= PB): (6 bcl + cm) WBP + lm + PB - inside a diagonal block the mirrored matrix keeps the row /
      //   column order, so the entry goes to the transposed position; left of it: local row block bcl + bandblk - lq
      const bool own = ln >= PB;
      const int K1 = 6 * (bandblk - lq);
      const int lconst = own ? lm : (K1 + lm) * WBP - K1;
      for (int r0 = wv; r0 < npr; r0 += CH * (B2_T / 64)) {
        double sv[CH], hv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int Cg = G0 + min(r0 + i * (B2_T / 64), npr - 1), cb6 = 6 * (Cg / 6);
          sv[i] = S[(int64_t)Cg * ld + min(max(cb6 - PB + ln, G0), Cg)];
          hv[i] = dr ? 0.0 : w.Hd[Cg];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int rr = r0 + i * (B2_T / 64);
          const int Cg = G0 + rr, cq = Cg / 6, cb6 = 6 * cq, cm = Cg - cb6, bcl = nb - 1 - cq;
          // rows of the separator: their own square starts from zero (the zero fill), their chain columns are loaded
          if (rr < npr && ln < WB && cb6 - PB + ln >= G0 && ln <= cm + PB) {
            const bool sepsq = bcl >= chain && (own || bcl + bandblk - lq >= chain);
            const int A1 = 6 * bcl * WBP + PB;
            double v = sepsq ? 0.0 : sv[i];
            if (ln == cm + PB && !sepsq) v += dep + ddm * (dr ? v : hv[i]);
            L[(own ? A1 + cm * WBP : A1 + cm) + lconst] = v;
          }
        }
      }
    }
    // @bwave 120
    // @bstamp 8
    for (int c = tl; c <= npr; c += B2_T) {
      double v = 0.0;
      if (c < npr && !(g && c / 6 >= chain)) v = S[(int64_t)n * ld + 6 * gblk(c / 6) + c % 6];
      y[c] = v;
    }
  }
  // @bstamp 9
  // ---- per-thread operand descriptors (as in band_solve_body, one tail row = the right-hand side)
  int prow_off = 0, prow_str = 0, prow_ia = -1;
  const bool has_prow = tl < PB + 1;
  if (has_prow) {
    if (tl < PB) { prow_ia = tl; prow_off = (6 + tl) * WBP - 6 - 6 * (tl / 6) + PB; prow_str = KS; }
    else { prow_off = nprmax * WBP; prow_str = 6; }
  }
  constexpr int UPT = 2;
  int uA[UPT], uB[UPT], uD[UPT], sA[UPT], sB[UPT], sD[UPT], u_ia[UPT], u_ib[UPT];
  bool has_pair[UPT];
#pragma unroll
  for (int sl = 0; sl < UPT; ++sl) {
    const int pid = tl < 64 ? (sl == 0 && tl < 21 ? tl : npair) : 21 + (tl - 64) + (B2_T - 64) * sl;
    has_pair[sl] = pid < npair;
    uA[sl] = uB[sl] = uD[sl] = sA[sl] = sB[sl] = sD[sl] = 0;
    u_ia[sl] = u_ib[sl] = -1;
    if (!has_pair[sl]) continue;
    if (pid < npp) {
      int ia = (int)((sqrtf(8.0f * (float)pid + 1.0f) - 1.0f) * 0.5f);
      while ((ia + 1) * (ia + 2) / 2 <= pid) ++ia;
      while (ia * (ia + 1) / 2 > pid) --ia;
      const int ib = pid - ia * (ia + 1) / 2;
      u_ia[sl] = ia; u_ib[sl] = ib;
      uA[sl] = (6 + ia) * WBP - 6 - 6 * (ia / 6) + PB; sA[sl] = KS;
      uB[sl] = (6 + ib) * WBP - 6 - 6 * (ib / 6) + PB; sB[sl] = KS;
      uD[sl] = (6 + ia) * WBP + ib - 6 * (ia / 6) + PB; sD[sl] = KS;
    } else {
      const int ib = pid - npp;
      u_ib[sl] = ib;
      uA[sl] = nprmax * WBP; sA[sl] = 6;
      uB[sl] = (6 + ib) * WBP - 6 - 6 * (ib / 6) + PB; sB[sl] = KS;
      uD[sl] = nprmax * WBP + 6 + ib; sD[sl] = 6;
    }
  }
  // @bstamp 10
  // @bwave 140
  auto bofs = [&](int r, int c) { return r * WBP + c - 6 * (r / 6) + PB; };
  auto factor_diag = [&](int kb) {
    const int j0 = 6 * kb;
    double* Dk = L + bofs(j0, j0);
    double A[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) A[i][j] = Dk[i * WBP + j];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double d = A[j][j];
#pragma unroll
      for (int m = 0; m < j; ++m) d = __builtin_fma(-(A[j][m]), A[j][m], d);
      if (!(d > 0.0)) { *failp = 1; d = 1.0; }
      const double rl = rsqrt_nr(d);
      A[j][j] = d * rl;
      rd[j] = rl;
      rdall[j0 + j] = rl;
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        double sacc = A[i][j];
#pragma unroll
        for (int m = 0; m < j; ++m) sacc = __builtin_fma(-(A[i][m]), A[j][m], sacc);
        A[i][j] = sacc * rl;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) { Dk[i * WBP + j] = A[i][j]; blk[i * 7 + j] = A[i][j]; }
  };
  // one block step of this image (the barriers are the caller's): panel, then trailing update + look-ahead
  auto panel = [&](int kb) {
    const int j0 = 6 * kb;
    if (has_prow && (prow_ia < 0 || j0 + 6 + prow_ia < npr)) {
      double* row = L + prow_off + kb * prow_str;
      double x[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double sacc = row[j];
#pragma unroll
        for (int m = 0; m < j; ++m) sacc = __builtin_fma(-(x[m]), blk[j * 7 + m], sacc);
        x[j] = sacc * rd[j];
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) row[j] = x[j];
    }
  };
  auto trailing = [&](int kb, int kend) {
    const int j0 = 6 * kb;
#pragma unroll
    for (int sl = 0; sl < UPT; ++sl) {
      if (has_pair[sl] && (u_ia[sl] < 0 || j0 + 6 + u_ia[sl] < npr) && (u_ib[sl] < 0 || j0 + 6 + u_ib[sl] < npr)) {
        const double* pa = L + uA[sl] + kb * sA[sl];
        const double* pb = L + uB[sl] + kb * sB[sl];
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) sacc = __builtin_fma(pa[m], pb[m], sacc);
        L[uD[sl] + kb * sD[sl]] -= sacc;
      }
    }
    if (tl < 64) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (tl == 0 && kb + 1 < kend) factor_diag(kb + 1);
    }
  };
  __syncthreads();
  // @bstamp 1
  if (tl == 0) factor_diag(0);
  __syncthreads();
  // @bstamp 2
  // ---- both chains, one block per step
  for (int kb = 0; kb < cb; ++kb) {
    const bool mine = kb < chain;
    if (mine) panel(kb);
    __syncthreads();
    if (mine) trailing(kb, chain);  // the look-ahead stops at the chain's end: the separator is not final yet
    __syncthreads();
  }
  // @bstamp 3
  // ---- chain B's contributions to the separator square and right-hand side go to image A (mirrored back)
  for (int idx = t; idx < PB * PB + PB; idx += 2 * B2_T) {
    if (idx < PB * PB) {
      const int rB = idx / PB, cB = idx % PB;  // separator-local row / column in image B's order, rB >= cB
      if (rB >= cB) {
        const int sr = rB / 6, sc = cB / 6, i = rB % 6, j = cB % 6;
        const double v = LB[bofs(6 * cb + rB, 6 * cb + cB)];
        const int br = bandblk - 1 - sr, bc = bandblk - 1 - sc;  // separator blocks in image A's order (br <= bc)
        if (sr == sc) LA[bofs(6 * (ca + br) + i, 6 * (ca + br) + j)] += v;
        else LA[bofs(6 * (ca + bc) + j, 6 * (ca + br) + i)] += v;  // the transposed position
      }
    } else {
      const int q = idx - PB * PB, sq = q / 6, i = q % 6;
      LA[nprmax * WBP + 6 * (ca + bandblk - 1 - sq) + i] += LB[nprmax * WBP + 6 * cb + q];
    }
  }
  __syncthreads();
  // ---- chain A goes on through the separator
  if (g == 0 && tl == 0) factor_diag(ca);
  __syncthreads();
  for (int kb = ca; kb < ca + bandblk; ++kb) {
    if (g == 0) panel(kb);
    __syncthreads();
    if (g == 0) trailing(kb, ca + bandblk);
    __syncthreads();
  }
  // @bstamp 4
  // ---- back substitution, column oriented as in band_solve_body (one wave per image): separator first (image A) ...
  auto backsub_block = [&](int kb, bool known, int colmax) {
    // x of block kb (solved here, or `known`: already in y), then y[band columns < colmax] -= L[block rows][column] x
    const int j0 = 6 * kb;
    const double* Dk = L + bofs(j0, j0);
    double Lk[6][6], rp[6], lc[6];
    const int rt = j0 - PB + tl;
    const bool upd = tl < PB && rt >= 0 && rt < colmax;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      rp[i] = rdall[j0 + i];
      lc[i] = upd ? L[(j0 + i) * WBP + tl] : 0.0;
#pragma unroll
      for (int j = 0; j < i; ++j) Lk[i][j] = Dk[i * WBP + j];
    }
    double x[6];
    if (known) {
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = y[j0 + j];
    } else {
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        double sacc = y[j0 + j];
#pragma unroll
        for (int m = 5; m > j; --m) sacc = __builtin_fma(-(Lk[m][j]), x[m], sacc);
        x[j] = sacc * rp[j];
      }
      if (tl < 6) {
        double xo = x[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) xo = tl == j ? x[j] : xo;
        y[j0 + tl] = xo;
      }
    }
    if (upd) {
      double sacc = lc[0] * x[0];
#pragma unroll
      for (int j = 1; j < 6; ++j) sacc = __builtin_fma(lc[j], x[j], sacc);
      y[rt] -= sacc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  if (g == 0 && tl < 64)
    for (int kb = ca + bandblk - 1; kb >= ca; --kb) backsub_block(kb, false, npr);
  __syncthreads();
  // @bstamp 5
  if (t < PB) LB[nprmax * WBP + 6 * (cb + bandblk - 1 - t / 6) + t % 6] = LA[nprmax * WBP + 6 * ca + t];  // x of the separator, mirrored
  __syncthreads();
  // ... then both chains at once; in image B the separator rows only hand their (known) x down to the chain's columns
  if (tl < 64) {
    if (g)
      for (int kb = cb + bandblk - 1; kb >= cb; --kb) backsub_block(kb, true, 6 * cb);
    for (int kb = chain - 1; kb >= 0; --kb) backsub_block(kb, false, npr);
  }
  __syncthreads();
  // @bstamp 6
  const bool bad = *failp != 0;
  if (t == 0) {
    if (bad) w.info[2] += 1;
    w.info[5] = 1;
  }
  for (int r = tl; r < (g ? 6 * cb : npr); r += B2_T) {  // image A: chain + separator, image B: its chain
    double x = y[r];
    if (bad || !(x == x)) x = 0.0;
    w.dx[6 * gblk(r / 6) + r % 6] = (float)x;
  }
  __syncthreads();
  apply_retraction(a, t, 2 * B2_T, nb);
  // @bstamp 7
  return true;
}

__global__ __launch_bounds__(2 * BAND_T) void ba_solve_band_kernel(BAArgs a, int lds_doubles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const BAWs& w = a.w;
  const int n = w.info[3], n_free = w.info[0], bandblk = w.info[4];
  const int ntail = n - 6 * n_free + 1, F = ntail - 1;
  const int PB = 6 * bandblk;
  const int npair = PB * (PB + 1) / 2 + ntail * PB + (F == 0 ? 0 : (F == 1 ? 2 : 5));
  const bool wide = PB + 7 > 64 || npair > 21 + 2 * (BAND_T - 64);
  // long pose-only chains: both ends at once, one chain per half of the workgroup (a uniform decision made before any
  // barrier; VIPE_BA_BAND2=0 in the environment keeps the one-chain form for A/B - passed down as a flag)
  if (!wide && F == 0 && a.band2 && band2_solve_body(a, lds_doubles, smem_raw)) return;
  if (threadIdx.x >= BAND_T) return;  // the one-chain forms are written for BAND_T threads
  if (wide) band_solve_body<BAND_UPT, true>(a, lds_doubles, smem_raw);
  else band_solve_body<2, false>(a, lds_doubles, smem_raw);
}

// ------------------------------------------------------------------------------------------------ solve (LDS dense)
//
// Dense windows that the band solver cannot hold (the keyframe frontend: up to ~25 free poses, every pair coupled through
// proximity and inactive edges), n + 1 <= 160 rows (26 free poses).  One workgroup of 8 waves, fp64, 6-column block steps; row n of the
// matrix is the right-hand side, so the forward substitution falls out of the factorisation.
//   * The trailing matrix lives in REGISTERS of waves 1..7: 16 x 16 tiles of the lower triangle in the accumulator
//     layout of v_mfma_f64_16x16x4_f64 (negated, so that the update is a plain multiply-accumulate); the update of a
//     block step is two matrix instructions per live tile whose operand fragments come from a panel buffer at addresses
//     that never change (no index arithmetic in the loop).
//   * Wave 0 is the CHAIN wave: it owns the dependent chain and nothing else - factor the 6 x 6 diagonal block in
//     registers (row per lane, pivots by v_readlane), solve the six panel rows of the NEXT diagonal block itself,
//     subtract their product from a preview of that block which the tile waves extracted one step earlier, factor it.
//     The tile waves' panel / update / extract run beside it; two workgroup barriers per step.  The wave that shares
//     the chain wave's SIMD (read from HW_ID) stays idle: fp64 matrix instructions and the chain's fp64 arithmetic use
//     the same pipe, and with a tile wave next to it the chain ran 2.5 times slower (stamps).
//   * LDS holds what has left the registers: the factor (packed rows, for the back substitution), the panel buffer
//     (two parities, rows of 8 doubles: 6 panel columns + 2 zeros = the K = 8 of two matrix instructions), previews.
// A single wave issues at most one instruction every ~4 cycles and a dependent fp64 operation takes ~35: the phases were
// sized by in-kernel cycle stamps (scratch/make_ba_stamps.py) - before this form the chain (2 850 cycles per block) ran
// in sequence with the panel, the update and the extraction (7 700 per step).
// Sets info[5] = 2 when it solved the system.
constexpr int DN_T = 512;
constexpr int DN_PP = 20;     // panel buffer: doubles per row = 2 parities x 8 + 4 (20 l16 mod 32 takes 8 values 4 apart: with kq the 64
                              // lanes of an operand-fragment read cover the 32 8-byte slots of the bank window twice - a pitch of 16 is 8-way conflicted)
constexpr int DN_SLOTS = 10;  // 6 tile waves x 10 >= 55 tiles (n + 1 <= 160 rows: 10 tile rows; 11 slots spill)
typedef double double4c __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(DN_T) void ba_solve_dense_kernel(BAArgs a, int lds_doubles) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  double* const L = reinterpret_cast<double*>(smem_raw);
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  const int t = threadIdx.x;
  const int n = w.info[3], n_free = w.info[0];
  const int npr = 6 * n_free, F = n - npr;
  const int NP = ((n + 1) * (n + 2) / 2 + 1) & ~1;  // packed size incl. the rhs row (even: what follows is 16-byte aligned)
  const int NTR = (n + 16) >> 4, NTL = NTR * (NTR + 1) / 2;  // tile rows covering rows 0..n; tiles of the triangle
  const int LDS_NEED = NP + 64 + 2 * 36 + 36 + (n + 8) + 6 * (n + 6) + DN_PP * 16 * NTR;
  // info[5] == 1: the band solver (which resets the flag whenever it runs) solved THIS iteration.  A 2 can only be this
  // kernel's own mark from the previous Gauss-Newton iteration of the call (ba_sens_kernel clears the flag per call).
  if (n == 0 || w.info[5] == 1 || a.mv || LDS_NEED > lds_doubles || F > 2 || NTL > 6 * DN_SLOTS) return;
  double* const blk = L + NP;      // 6x7: the current diagonal factor block
  double* const rd = blk + 42;     // its reciprocal pivots
  int* const failp = reinterpret_cast<int*>(rd + 6);
  double* const dnext = rd + 8;    // [2][36] previews of the next diagonal block (entries (i, j), j <= i)
  double* const xbuf = dnext + 72; // [6][6] the chain wave's own panel rows
  double* const rdall = xbuf + 36; // [n] reciprocal pivots of every column (back substitution)
  double* const Linv = rdall + ((n + 8) & ~1);  // [blocks][6][6] inverses of the diagonal factor blocks, zero above the diagonal
  double* const Pbuf = Linv + 6 * (n + 6);      // [16 NTR rows][DN_PP]: panel rows (raw, then solved), parity p at column 8 p
  auto off = [](int r) { return r * (r + 1) / 2; };
  const double* S = w.S;
  const int ld = w.ld;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, kq = lane >> 4;
  const int nblk = n_free + (F > 0 ? 1 : 0);
  // @stamp 0
  if (t == 0) *failp = 0;
  // roles: wave 0 = chain; waves on its SIMD = idle (barriers only); the others = tile waves, ranked.  Should the
  // hardware place fewer than six waves on the other SIMDs, the idle ones become tile waves after all (slower, correct).
  int* const simd_of = reinterpret_cast<int*>(xbuf);  // 8 ints, before xbuf's first use
  if (lane == 0) simd_of[wave] = (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3);  // HW_ID.SIMD_ID
  __syncthreads();
  int twave = -1, ntw = 0;  // this wave's rank among the tile waves; their number
  {
    int others = 0;
    for (int v = 1; v < DN_T / 64; ++v) others += simd_of[v] != simd_of[0];
    const bool use_partners = others < 6;
    for (int v = 1; v < DN_T / 64; ++v) {
      const bool tw = use_partners || simd_of[v] != simd_of[0];
      if (v == wave && tw) twave = ntw;
      ntw += tw;
    }
    if (twave >= 6) twave = -1;  // six tile waves carry all the slots
  }
  twave = __builtin_amdgcn_readfirstlane(twave);
  __syncthreads();
  for (int i = t; i < DN_PP * 16 * NTR; i += DN_T) Pbuf[i] = 0.0;  // columns 6, 7 stay zero; rows beyond n too
  auto damped = [&](int r, double v) {  // LM damping on the diagonal (matrix.py:179-186)
    const bool pose = r < npr;
    return v + (pose ? (double)prm.pose_ep : 1e-6) + (pose ? (double)prm.pose_damping : 1e-6) * (a.droid ? v : w.Hd[r]);
  };

  // ---- tile waves: tile tau = I (I + 1) / 2 + J of the lower triangle -> tile wave tau % 6, slot tau / 6.  Per slot and
  //      lane: T (four entries: rows 16 I + kq + 4 r4, column 16 J + l16, NEGATED), the row / column this lane extracts
  //      and the LDS byte addresses of its two operand fragments in the panel buffer
  double4c T[DN_SLOTS];
  int colv[DN_SLOTS], rowv[DN_SLOTS];
#pragma unroll
  for (int sl = 0; sl < DN_SLOTS; ++sl) {
    T[sl] = double4c{0.0, 0.0, 0.0, 0.0};
    colv[sl] = rowv[sl] = -(1 << 20);
  }
  if (twave >= 0) {
#pragma unroll
    for (int sl = 0; sl < DN_SLOTS; ++sl) {
      const int tau = twave + 6 * sl;
      int ti = (int)((sqrtf(8.0f * (float)tau + 1.0f) - 1.0f) * 0.5f);
      while ((ti + 1) * (ti + 2) / 2 <= tau) ++ti;
      while (ti * (ti + 1) / 2 > tau) --ti;
      const int tj = tau - ti * (ti + 1) / 2;
      const bool ok = tau < NTL;
      colv[sl] = ok ? 16 * tj + l16 : -(1 << 20);  // an absent tile is never live and never intersects a column block
      rowv[sl] = ok ? 16 * ti + kq : -(1 << 20);
      // unconditional loads from clamped positions (all of a lane's loads in flight at once: a load under a branch
      // whose condition needs the previous load costs a memory round trip each), selected afterwards
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = 16 * ti + kq + 4 * r4, col = 16 * tj + l16;
        const int rr = min(row, n), cc = min(col, min(rr, n - 1));
        T[sl][r4] = S[(int64_t)rr * ld + cc];
      }
    }
    const bool dr = a.droid;
#pragma unroll
    for (int sl = 0; sl < DN_SLOTS; ++sl) {
      double hd[4];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) hd[r4] = dr ? 0.0 : w.Hd[min(max(rowv[sl] + 4 * r4, 0), n - 1)];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int row = rowv[sl] + 4 * r4, col = colv[sl];
        double v = T[sl][r4];
        if (col == row) {
          const bool pose = row < npr;
          v += (pose ? (double)prm.pose_ep : 1e-6) + (pose ? (double)prm.pose_damping : 1e-6) * (dr ? v : hd[r4]);
        }
        T[sl][r4] = (row <= n && col < n && col <= row && col >= 0) ? -v : 0.0;
      }
    }
  }
  // which of this wave's slots hold tile column J / tile row I: bit masks, lane J < 16 holds the column mask of J and
  // lane 16 + I the row mask of I (a block step finds the two or three slots that meet its columns with two v_readlane
  // instead of testing every slot); nsl = slots in use (the unrolled slot loops leave at the first unused one)
  int maskv = 0, nsl = 0;
#pragma unroll
  for (int sl = 0; sl < DN_SLOTS; ++sl) {
    const int tj = __builtin_amdgcn_readfirstlane(colv[sl]) >> 4, ti = __builtin_amdgcn_readfirstlane(rowv[sl]) >> 4;
    if (tj >= 0) {
      maskv |= ((lane == tj) || (lane == 16 + ti)) ? (1 << sl) : 0;
      nsl = sl + 1;
    }
  }
  // the chain wave's diagonal block: lane i = row i (lanes >= 6 run along on row 5 and store nothing)
  const int ic = lane < 6 ? lane : 5;
  double A[6];
  if (wave == 0) {
    const int bw0 = min(6, n);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      double v = (c == ic) ? 1.0 : 0.0;
      if (ic < bw0 && c <= ic) {
        v = S[(int64_t)ic * ld + c];
        if (c == ic) v = damped(ic, v);
      }
      A[c] = v;
    }
  }
  // factor the block held in A (bw columns; identity beyond); publish L, blk, rd, rdall.
  // DIVISION-FREE elimination: a pivot step multiplies the remaining rows by the pivot p instead of dividing the pivot
  // column by it, a_im <- (a_im p - a_ic a_mc) 2^-e with 2^e the binade of p (an exact rescale that keeps the running
  // scale s in (2^-6, 1]): the dependent chain per pivot is v_readlane -> fused multiply-add -> ldexp instead of
  // reciprocal square root + two Newton steps + multiply + fused multiply-add (9 dependent fp64 operations of ~35 cycles
  // each: 2 850 cycles per block by the stamps).  The factor follows at the end, all six columns at once:
  // L_ic = a_ic / sqrt(p_c s_c), 1 / L_cc = s_c / sqrt(p_c s_c), one reciprocal square root per LANE.  Same stability as
  // the Cholesky recurrence (it is the LDL^T elimination with exactly rescaled rows).
  auto factor_diag = [&](int kb) {
    const int j0 = 6 * kb, bw = min(6, n - j0);
    double sc = 1.0, p_own = 1.0, s_own = 1.0;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      // a non-positive pivot marks the solve as failed (its step is then zero) and the arithmetic just runs on
      const double pv = readlane_f64(A[c], c);
      bad |= (c < bw) & !(pv > 0.0);
      p_own = (ic == c) ? pv : p_own;
      s_own = (ic == c) ? sc : s_own;
      if (c + 1 < 6) {
        // pv = ps 2^e with ps in [0.5, 1): rows are multiplied by ps and the pivot column is scaled by 2^-e ONCE, so that
        // an entry's update is one multiply and one fused multiply-add
        const unsigned long long pb = __builtin_bit_cast(unsigned long long, pv);
        const int e = (int)((pb >> 52) & 0x7ff) - 1022;
        const double ps = __builtin_bit_cast(double, (pb & 0x800fffffffffffffull) | (1022ull << 52));
        const double own_sq = A[c] * A[c];              // ready before the pivot arrives
        const double colc = __builtin_ldexp(A[c], -e);  // this lane's entry of the pivot column, scaled
#pragma unroll
        for (int m = c + 1; m < 6; ++m) {
          // lane c + 1 forms its next pivot from its own entry: no lane hand-off on the dependent chain
          const double prod = (m == c + 1 && ic == c + 1) ? __builtin_ldexp(own_sq, -e) : A[c] * readlane_f64(colc, m);
          A[m] = __builtin_fma(A[m], ps, -prod);
        }
        sc = sc * ps;
      }
    }
    const double rho = rsqrt_nr(p_own * s_own);  // lane c: 1 / sqrt(p_c s_c)
    const double rdv = s_own * rho;              // 1 / L_cc
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double rc = readlane_f64(rho, c);
      A[c] = ic == c ? p_own * rho : (ic > c ? A[c] * rc : 0.0);
    }
    if (lane == 0 && bad) *failp = 1;
    if (lane < 6) {
      rd[lane] = rdv;
      if (lane < bw) rdall[j0 + lane] = rdv;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        if (c <= lane) {
          if (lane < bw) L[off(j0 + lane) + j0 + c] = A[c];
          blk[lane * 7 + c] = A[c];
        }
      }
    }
  };
  // forward substitution of one panel row against the published factor block
  auto solve_row = [&](const double (&raw)[6], int bw, double (&x)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double sacc = j < bw ? raw[j] : 0.0;
#pragma unroll
      for (int m = 0; m < j; ++m) sacc = __builtin_fma(-x[m], blk[j * 7 + m], sacc);
      x[j] = sacc * rd[j];
    }
  };
  // tile waves: columns [c0, c0 + cw) of the trailing matrix -> panel buffer Pb (= Pbuf + 8 parity; every row of the
  // tiles that hold them: rows that are not panel rows any more only ever meet finished entries); the block
  // [p0, p0 + pw)^2 that follows -> preview buffer pv
  auto extract = [&](double* Pb, int c0, int cw, int p0, int pw, double* pv) {
    const int cm = __builtin_amdgcn_readlane(maskv, c0 >> 4) | __builtin_amdgcn_readlane(maskv, (c0 + cw - 1) >> 4);
    int pm = 0;
    if (pw > 0) {
      const int ja = p0 >> 4, jb = (p0 + pw - 1) >> 4;
      pm = (__builtin_amdgcn_readlane(maskv, ja) | __builtin_amdgcn_readlane(maskv, jb)) &
           (__builtin_amdgcn_readlane(maskv, 16 + ja) | __builtin_amdgcn_readlane(maskv, 16 + jb));
    }
    if ((cm | pm) == 0) return;
#pragma unroll
    for (int sl = 0; sl < DN_SLOTS; ++sl) {
      if ((cm >> sl) & 1) {
        if ((unsigned)(colv[sl] - c0) < (unsigned)cw) {
          double* dst = Pb + rowv[sl] * DN_PP + (colv[sl] - c0);
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) dst[r4 * 4 * DN_PP] = -T[sl][r4];
        }
      }
      if ((pm >> sl) & 1) {
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int pr = rowv[sl] + 4 * r4 - p0, pc = colv[sl] - p0;
          if ((unsigned)pr < (unsigned)pw && pc >= 0 && pc <= pr) pv[pr * 6 + pc] = -T[sl][r4];
        }
      }
    }
  };

  if (twave >= 0) extract(Pbuf, 0, min(6, n), min(6, n), min(6, n - min(6, n)), dnext + 36);
  if (wave == 0) factor_diag(0);
  __syncthreads();
  // @stamp 1
  // one block step of each role; the panel buffer's parity is kb & 1.  Two workgroup barriers per step in BOTH loops (the
  // hardware barrier counts arrivals, whatever the code address): separate loops keep the tile registers out of the
  // chain wave's code and the chain's out of the tile waves'
  if (wave == 0) {
    for (int kb = 0; kb < nblk; ++kb) {
      double* const Pb = Pbuf + 8 * (kb & 1);
      const int j0 = 6 * kb, bw = min(6, n - j0), R0 = j0 + bw, nbw = min(6, n - R0);
      double x[6];
      // @stampk 0
      // the rows of the next diagonal block: solved here, kept in registers, published for the update
      if (nbw > 0) {
        const int r = min(R0 + ic, n);
        const double* src = Pb + r * DN_PP;
        double raw[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) raw[j] = src[j];
        solve_row(raw, bw, x);
        if (lane < nbw) {
          double* dst = Pb + r * DN_PP;
          double* lrow = L + off(r) + j0;
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            dst[j] = j < bw ? x[j] : 0.0;
            if (j < bw) lrow[j] = x[j];
            xbuf[lane * 6 + j] = j < bw ? x[j] : 0.0;
          }
        }
      }
      // @stampk 1
      __syncthreads();
      // @stampk 2
      if (nbw > 0) {
        // next diagonal block = its preview (state before this step) - P P^T of its six panel rows, then its factor
        const double* pv = dnext + ((kb + 1) & 1) * 36;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          double v = (c == ic) ? 1.0 : 0.0;
          if (ic < nbw && c <= ic) {
            double s0 = pv[ic * 6 + c], s1 = 0.0;
#pragma unroll
            for (int m = 0; m < 6; m += 2) {
              s0 = __builtin_fma(-x[m], xbuf[c * 6 + m], s0);
              s1 = __builtin_fma(-x[m + 1], xbuf[c * 6 + m + 1], s1);
            }
            v = s0 + s1;
          }
          A[c] = v;
        }
        // @stampk 3
        factor_diag(kb + 1);
      }
      // @stampk 4
      __syncthreads();
      // @stampk 5
    }
  } else {
    for (int kb = 0; kb < nblk; ++kb) {
      double* const Pb = Pbuf + 8 * (kb & 1);
      const int j0 = 6 * kb, bw = min(6, n - j0), R0 = j0 + bw, nbw = min(6, n - R0), R1 = R0 + nbw;
      // @wstampk 0
      if (twave >= 0) {
        const int r = R1 + 64 * twave + lane;
        if (r <= n) {
          double* row = Pb + r * DN_PP;
          double raw[6], x[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) raw[j] = row[j];
          solve_row(raw, bw, x);
          double* lrow = L + off(r) + j0;
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            row[j] = j < bw ? x[j] : 0.0;
            if (j < bw) lrow[j] = x[j];
          }
        }
      }
      // @wstampk 1
      __syncthreads();
      // @wstampk 2
      // trailing update T' += P P^T (live tiles: some column >= R0), then the next column block and the preview after it.
      // One inline-asm block per slot - skip test, the four operand reads, both matrix instructions - so that the
      // compiler sees T[sl] modified IN PLACE on every path: through the builtin under a branch it kept the skipped and
      // the updated accumulator in two register sets (four 64-bit moves per slot and step, twice the registers).
      const unsigned pb_u = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)(Pb + kq);
      const bool upd = twave >= 0 && R0 < n;
#pragma unroll
      for (int sl = 0; sl < DN_SLOTS; ++sl) {
        const int live = __builtin_amdgcn_readfirstlane((int)(upd && sl < nsl && (colv[sl] | 15) >= R0));
        const unsigned aa = pb_u + (unsigned)(((rowv[sl] & ~15) + l16) * (DN_PP * 8));
        const unsigned ba = pb_u + (unsigned)(((colv[sl] & ~15) + l16) * (DN_PP * 8));
        double fa0, fa1, fb0, fb1;
        asm volatile(
            "s_cmp_eq_u32 %7, 0\n\t"
            "s_cbranch_scc1 1f\n\t"
            "ds_read_b64 %1, %5\n\t"
            "ds_read_b64 %3, %6\n\t"
            "ds_read_b64 %2, %5 offset:32\n\t"
            "ds_read_b64 %4, %6 offset:32\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_mfma_f64_16x16x4_f64 %0, %1, %3, %0\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_mfma_f64_16x16x4_f64 %0, %2, %4, %0\n"
            "1:"
            : "+v"(T[sl]), "=&v"(fa0), "=&v"(fa1), "=&v"(fb0), "=&v"(fb1)
            : "v"(aa), "v"(ba), "s"(live)
            : "scc", "memory");
      }
      // the compiler does not see matrix instructions inside inline asm: cover the result hazard of the last one (the
      // extraction below reads T with vector instructions) by hand
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      // @wstampk 3
      if (upd) extract(Pbuf + 8 * ((kb + 1) & 1), R0, nbw, R1, min(6, n - R1), dnext + (kb & 1) * 36);
      // @wstampk 4
      __syncthreads();
      // @wstampk 5
    }
  }
  // @stamp 2
  // ---- back substitution L^T x = y (y = row n).  First every diagonal block is replaced by its INVERSE (thread = one
  //      column of one block; all blocks at once), so that a block's six unknowns are six independent dot products
  //      instead of a twelve-step substitution chain.  Then wave 0 alone, y in REGISTERS (lane c holds y[c], y[c + 64],
  //      y[c + 128]): per block the six y entries come by v_readlane, every lane forms all six x (uniform values, the
  //      inverse block by broadcast reads) and subtracts its columns' contributions; the next block's operands are
  //      fetched while the current one is on the chain.  The version that kept y in LDS spent 2 570 cycles per block
  //      (LDS round trip of y behind the 39 in-order prefetch reads) where the dependent chain is ~500.
  const double* y = L + off(n);
  __syncthreads();
  {
    const int kb = t / 6, q = t - 6 * kb;
    if (kb < nblk) {
      const int j0 = 6 * kb, bw = min(6, n - j0);
      double z[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        // column q of the inverse: z_q = 1 / L_qq, z_i = -(sum_{q <= m < i} L_im z_m) / L_ii; zero rows beyond the matrix
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m < i; ++m) sacc = __builtin_fma((i < bw) ? L[off(j0 + i) + j0 + m] : 0.0, (m >= q) ? z[m] : 0.0, sacc);
        const double ri = (i < bw) ? rdall[j0 + i] : 0.0;
        z[i] = (q < bw) ? ((i == q) ? ri : ((i > q) ? -sacc * ri : 0.0)) : 0.0;
        Linv[kb * 36 + 6 * i + q] = z[i];
      }
    }
  }
  __syncthreads();
  // @stamp 3
  const bool bad = *failp != 0;
  if (t < 64) {
    // y in registers: column c = 48 q + lane (lanes 0..47; 48 = 8 blocks, so a block never straddles two registers).
    // Per block: its six y entries by v_readlane, x_q in the block's own lanes (lane 6 kbl + q holds column q of the
    // inverse), x back to every lane by v_readlane, then each lane subtracts its columns' contributions.  Operands are
    // fetched a block ahead with UNCONDITIONAL loads (what lies right of the block in a packed row is discarded by a
    // select on the result): a predicated load costs a branch, and a spilled operand a scratch round trip.
    constexpr int YR = 4;
    double yv[YR];
#pragma unroll
    for (int q = 0; q < YR; ++q) yv[q] = (t < 48 && 48 * q + t < n) ? y[48 * q + t] : 0.0;
    const int qmax = (nblk - 1) >> 3;
    auto run_reg = [&](auto QC, int kb_hi) {  // blocks kb_hi .. 8 Q, register Q
      constexpr int Q = decltype(QC)::value;
      double col[2][6], lc[2][Q + 1][6];
      auto fetch = [&](int kb, double (&cv)[6], double (&lv)[Q + 1][6]) {
        const int j0 = 6 * kb;
        const double* ci = Linv + kb * 36 + (t - 6 * (kb - 8 * Q));  // column (lane - first lane of the block)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          cv[m] = ci[6 * m];
          const double* row = L + off(j0 + m) + t;
#pragma unroll
          for (int q = 0; q <= Q; ++q) lv[q][m] = row[48 * q];
        }
      };
      auto solve = [&](int kb, const double (&cv)[6], const double (&lv)[Q + 1][6]) {
        const int kbl = kb - 8 * Q, j0 = 6 * kb, l0 = 6 * kbl;
        double v[6], x[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) v[m] = readlane_f64(yv[Q], l0 + m);
        double s0 = cv[0] * v[0], s1 = cv[1] * v[1];
        s0 = __builtin_fma(cv[2], v[2], s0);
        s1 = __builtin_fma(cv[3], v[3], s1);
        s0 = __builtin_fma(cv[4], v[4], s0);
        s1 = __builtin_fma(cv[5], v[5], s1);
        const double xq = s0 + s1;  // x of column (lane - l0) in the block's lanes
#pragma unroll
        for (int m = 0; m < 6; ++m) x[m] = readlane_f64(xq, l0 + m);
#pragma unroll
        for (int q = 0; q <= Q; ++q) {
          double d0 = lv[q][0] * x[0], d1 = lv[q][1] * x[1];
          d0 = __builtin_fma(lv[q][2], x[2], d0);
          d1 = __builtin_fma(lv[q][3], x[3], d1);
          d0 = __builtin_fma(lv[q][4], x[4], d0);
          d1 = __builtin_fma(lv[q][5], x[5], d1);
          const double nv = yv[q] - (d0 + d1);
          yv[q] = (t < 48 && 48 * q + t < j0) ? nv : yv[q];  // columns left of the block
        }
        yv[Q] = (t >= l0 && t < l0 + 6) ? xq : yv[Q];  // the block's own entries become x
      };
      int kb = kb_hi;
      fetch(kb, col[0], lc[0]);
      while (true) {  // two blocks per trip: the operand buffers are named at compile time
        if (kb > 8 * Q) fetch(kb - 1, col[1], lc[1]);
        solve(kb, col[0], lc[0]);
        if (--kb < 8 * Q) break;
        if (kb > 8 * Q) fetch(kb - 1, col[0], lc[0]);
        solve(kb, col[1], lc[1]);
        if (--kb < 8 * Q) break;
      }
    };
    if (qmax >= 3) run_reg(std::integral_constant<int, 3>{}, nblk - 1);
    if (qmax >= 2) run_reg(std::integral_constant<int, 2>{}, qmax == 2 ? nblk - 1 : 23);
    if (qmax >= 1) run_reg(std::integral_constant<int, 1>{}, qmax == 1 ? nblk - 1 : 15);
    run_reg(std::integral_constant<int, 0>{}, qmax == 0 ? nblk - 1 : 7);
    if (t < 48) {
#pragma unroll
      for (int q = 0; q < YR; ++q) {
        const int dd = 48 * q + t;
        if (dd < n) {
          double x = yv[q];
          if (bad || !(x == x)) x = 0.0;
          w.dx[dd] = (float)x;
        }
      }
    }
    if (t == 0) {
      if (bad) w.info[2] += 1;
      w.info[5] = 2;
    }
  }
  // @stamp 4
  __syncthreads();
  apply_retraction(a, t, DN_T, n_free);
  // @stamp 5
}

// ------------------------------------------------------------------------------------------------ solve

// Dense Cholesky solve of the reduced system by ONE workgroup (16 waves), fp64.
//   S: lower triangle, row-major, ld; row n holds the rhs, so the forward substitution y = L^-1 g falls out of
//   the factorisation as the last panel row.  Right-looking, NB = 24 columns per step:
//     1. diagonal block: wave 0, one matrix row per lane in registers, column-by-column (Crout) with the
//        finished columns published to LDS;
//     2. panel: one row per thread, X Lkk^T = A by forward substitution against Lkk in LDS; the solved panel is
//        kept TRANSPOSED in LDS (PT[j][row]) so the update below reads it without bank conflicts;
//     3. trailing update A22 -= P P^T: 1x4 register tiles per thread, panel from LDS, S read-modify-write in
//        32-byte row segments.
//   Then blocked backward substitution L^T x = y and the pose / intrinsics retraction.
constexpr int NB = 12;
constexpr int CT_MIN_N = 256;  // larger systems take the tiled, chip-wide factorisation further down
constexpr int SOLVE_T = 512;  // 8 waves: up to 256 VGPRs per lane, no spills in the register-resident phases

struct SolveLds {
  double Lkk[NB][NB + 1];
  double rdiag[NB];  // 1 / L[j][j]
  double xk[NB];
  int fail;
};



__global__ __launch_bounds__(SOLVE_T) void ba_solve_kernel(BAArgs a, int panel_cap) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  SolveLds& sh = *reinterpret_cast<SolveLds*>(smem_raw);
  double* PT = reinterpret_cast<double*>(smem_raw + ((sizeof(SolveLds) + 15) / 16) * 16);  // [NB][panel_cap]
  const vipe_ba_params& prm = a.p;
  const BAWs& w = a.w;
  const int t = threadIdx.x;
  const int n = w.info[3], n_free = w.info[0];
  const int ld = w.ld;
  double* S = w.S;
  if (t == 0) sh.fail = 0;
  if (n == 0 || w.info[5] != 0 || n > CT_MIN_N) return;  // an LDS solver (band: 1, dense: 2) took the system; large ones: tiled
  // LM damping on the diagonal: += ep + lambda * diag(H)  (matrix.py:179-186)
  for (int dd = t; dd < n; dd += SOLVE_T) {
    // poses: the caller's (lambda, ep); intrinsics 1e-6 / 1e-6; rig rotations 1e-4 / 1e-4 (buffer.py:466,498,503)
    const bool pose = dd < 6 * n_free, rigrow = a.mv && dd >= 6 * n_free + a.nintr;
    const double ep = pose ? (double)prm.pose_ep : (rigrow ? 1e-4 : 1e-6);
    const double lam = pose ? (double)prm.pose_damping : (rigrow ? 1e-4 : 1e-6);
    S[(int64_t)dd * ld + dd] += ep + lam * (a.droid ? S[(int64_t)dd * ld + dd] : w.Hd[dd]);
  }
  __syncthreads();
  const bool use_lds_panel = (n + 1) <= panel_cap;
  const int npose_rows = 6 * n_free;
  // band of the pose part in 6x6 blocks (plan kernel); the scalar fallback path treats the system as dense
  const int bandblk = use_lds_panel ? w.info[4] : n;

  for (int k0 = 0; k0 < n; k0 += NB) {
    const int bw = min(NB, n - k0);
    // ---- 1. diagonal block (wave 0)
    if (t < WAVE) {
      double row[NB];
      const int r = t;
      if (r < bw) {
#pragma unroll
        for (int c = 0; c < NB; ++c) row[c] = (c <= r && c < bw) ? S[(int64_t)(k0 + r) * ld + k0 + c] : 0.0;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j < bw) {
          double sacc = 0.0;
          if (r >= j && r < bw) {
            sacc = row[j];
#pragma unroll
            for (int m = 0; m < NB; ++m)
              if (m < j) sacc -= row[m] * sh.Lkk[j][m];
          }
          // pivot from lane j
          double piv = __shfl(sacc, j, WAVE);
          if (!(piv > 0.0)) {
            if (r == 0) sh.fail = 1;
            piv = 1.0;
          }
          const double rl = rsqrt_nr(piv);
          if (r >= j && r < bw) {
            row[j] = (r == j) ? piv * rl : sacc * rl;
            sh.Lkk[r][j] = row[j];
            if (r == j) sh.rdiag[j] = rl;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
      }
      if (r < bw) {
#pragma unroll
        for (int c = 0; c < NB; ++c)
          if (c <= r) S[(int64_t)(k0 + r) * ld + k0 + c] = row[c];
      }
    }
    __syncthreads();
    // ---- 2. panel rows r0..n (row n = rhs)
    // Rows below the block that can be nonzero in these columns: the band [r0, e1) of pose rows plus the dense
    // tail [t0, n] (intrinsics rows and the rhs row).  Compact panel index pr -> global row prow(pr).
    const int r0 = k0 + bw;
    const int e1 = r0 < npose_rows ? min(npose_rows, 6 * ((k0 + bw - 1) / 6 + bandblk + 1)) : r0;
    const int t0 = max(r0, npose_rows);
    const int nb1 = max(e1 - r0, 0);
    const int m = nb1 + (n - t0 + 1);
    auto prow = [&](int pr) { return pr < nb1 ? r0 + pr : t0 + (pr - nb1); };
    for (int pr = t; pr < m; pr += SOLVE_T) {
      double x[NB];
      double* grow = S + (int64_t)prow(pr) * ld + k0;
#pragma unroll
      for (int j = 0; j < NB; ++j) x[j] = j < bw ? grow[j] : 0.0;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j < bw) {
          double sacc = x[j];
#pragma unroll
          for (int q = 0; q < NB; ++q)
            if (q < j) sacc -= x[q] * sh.Lkk[j][q];
          x[j] = sacc * sh.rdiag[j];
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j < bw) grow[j] = x[j];
        if (use_lds_panel) PT[j * panel_cap + pr] = j < bw ? x[j] : 0.0;
      }
    }
    __syncthreads();
    // ---- 3. trailing update A22 -= P P^T on the fp64 matrix cores (v_mfma_f64_16x16x4_f64): 16x16 tiles of the
    //         lower triangle, one tile per wave at a time, K = 24 = 6 MFMAs; operands straight from the
    //         transposed panel in LDS (lane l: A[row l&15][k l>>4], B[k l>>4][col l&15]).
    if (use_lds_panel) {
      typedef double double4v __attribute__((ext_vector_type(4)));
      const int wv = t >> 6, ln = t & 63;
      const int nt = (m + 15) >> 4;
      const int ntiles = nt * (nt + 1) / 2;
      for (int q = wv; q < ntiles; q += SOLVE_T / 64) {
        int ti = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
        while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
        while (ti * (ti + 1) / 2 > q) --ti;
        const int tj = q - ti * (ti + 1) / 2;
        double4v c = {0.0, 0.0, 0.0, 0.0};
        const int ar = 16 * ti + (ln & 15), bc = 16 * tj + (ln & 15), kq = ln >> 4;
#pragma unroll
        for (int s4 = 0; s4 < NB / 4; ++s4) {
          const double av = PT[(4 * s4 + kq) * panel_cap + ar];
          const double bv = PT[(4 * s4 + kq) * panel_cap + bc];
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
        }
        const int cc = 16 * tj + (ln & 15);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int rr = 16 * ti + (ln >> 4) + 4 * r4;
          if (rr < m && cc <= rr && cc <= m - 2) S[(int64_t)prow(rr) * ld + prow(cc)] -= c[r4];
        }
      }
    } else {
      const int tx = t & 31, ty = t >> 5;  // 32 x 32 threads, each a 1 x 4 tile
      for (int rr = ty; rr < m; rr += SOLVE_T / 32) {
        const int cmax = min(rr, m - 2);  // inclusive
        for (int c4 = tx * 4; c4 <= cmax; c4 += 128) {
          double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
          if (use_lds_panel) {
#pragma unroll 4
            for (int j = 0; j < bw; ++j) {
              const double pr_ = PT[j * panel_cap + rr];
              const double* pc = PT + j * panel_cap + c4;
              acc0 += pr_ * pc[0]; acc1 += pr_ * pc[1]; acc2 += pr_ * pc[2]; acc3 += pr_ * pc[3];
            }
          } else {
            const double* prw = S + (int64_t)(r0 + rr) * ld + k0;
            for (int j = 0; j < bw; ++j) {
              const double pr_ = prw[j];
              acc0 += pr_ * S[(int64_t)(r0 + c4) * ld + k0 + j];
              if (c4 + 1 <= cmax) acc1 += pr_ * S[(int64_t)(r0 + c4 + 1) * ld + k0 + j];
              if (c4 + 2 <= cmax) acc2 += pr_ * S[(int64_t)(r0 + c4 + 2) * ld + k0 + j];
              if (c4 + 3 <= cmax) acc3 += pr_ * S[(int64_t)(r0 + c4 + 3) * ld + k0 + j];
            }
          }
          double* dst = S + (int64_t)(r0 + rr) * ld + r0 + c4;
          dst[0] -= acc0;
          if (c4 + 1 <= cmax) dst[1] -= acc1;
          if (c4 + 2 <= cmax) dst[2] -= acc2;
          if (c4 + 3 <= cmax) dst[3] -= acc3;
        }
      }
    }
    __syncthreads();
  }

  // ---- backward substitution L^T x = y (y = row n).
  // (i) invert every diagonal block Lkk (lower triangular) in parallel, one wave per block, lane c = column c of
  //     the inverse by forward substitution; the inverse overwrites the STRICT UPPER part + a side array is not
  //     needed: it is written to the (unused) upper triangle of S at the block's position, transposed, i.e.
  //     S[k0+c][k0+j] (j > c) := Linv[j][c], and the inverse's diagonal to Hd (no longer needed).
  {
    const int nblk = (n + NB - 1) / NB;
    const int wv = t >> 6, ln = t & 63;
    for (int blk = wv; blk < nblk; blk += SOLVE_T / 64) {
      const int k0 = blk * NB, bw = min(NB, n - k0);
      if (ln < bw) {
        const int c = ln;
        double z[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (j < bw && j >= c) {
            double sacc = (j == c) ? 1.0 : 0.0;
#pragma unroll
            for (int q = 0; q < NB; ++q)
              if (q < j) sacc -= S[(int64_t)(k0 + j) * ld + k0 + q] * z[q];  // z[q] == 0 for q < c
            z[j] = sacc / S[(int64_t)(k0 + j) * ld + k0 + j];
          } else {
            z[j] = 0.0;
          }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          if (j < bw && j > c) S[(int64_t)(k0 + c) * ld + k0 + j] = z[j];  // upper triangle: Linv[j][c]
          if (j == c) w.Hd[k0 + c] = z[j];
        }
      }
    }
  }
  __syncthreads();
  // (ii) blocks from the last to the first: x_k = Lkk^-T y_k (a 24x24 mat-vec, lane j: sum_m Linv[m][j] y[m]),
  //      then y_c -= sum_m L[k0+m][c] x_k[m] for every earlier column c (coalesced row reads).
  double* yrow = S + (int64_t)n * ld;
  for (int k0 = ((n - 1) / NB) * NB; k0 >= 0; k0 -= NB) {
    const int bw = min(NB, n - k0);
    if (t < WAVE) {
      if (t < bw) sh.xk[t] = yrow[k0 + t];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      double xj = 0.0;
      if (t < bw) {
        // Linv[m][j] for m > j is stored at S[k0+j][k0+m]; Linv[j][j] in Hd
        xj = w.Hd[k0 + t] * sh.xk[t];
        for (int mq = t + 1; mq < bw; ++mq) xj += S[(int64_t)(k0 + t) * ld + k0 + mq] * sh.xk[mq];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (t < bw) {
        sh.xk[t] = xj;
        yrow[k0 + t] = xj;
      }
    }
    __syncthreads();
    // rows of this block are zero left of the band (pose rows only; tail rows are dense)
    const int c_lo = (k0 + bw <= npose_rows) ? max(0, 6 * (k0 / 6 - bandblk)) : 0;
    for (int c = c_lo + t; c < k0; c += SOLVE_T) {
      double sacc = 0.0;
      for (int q = 0; q < bw; ++q) sacc += S[(int64_t)(k0 + q) * ld + c] * sh.xk[q];
      yrow[c] -= sacc;
    }
    __syncthreads();
  }
  const bool bad = sh.fail != 0;
  if (t == 0 && bad) w.info[2] += 1;
  for (int dd = t; dd < n; dd += SOLVE_T) {
    double x = yrow[dd];
    if (bad || !(x == x)) x = 0.0;  // zero step on a failed factorisation
    w.dx[dd] = (float)x;
  }
  __syncthreads();
  apply_retraction(a, t, SOLVE_T, n_free);
}

// ------------------------------------------------------------------------------------------------ solve (tiled, chip wide)
//
// The global BA's reduced systems (n = 600 ... 1200+ unknowns, dense: every keyframe pair may couple) are bound in the
// single-workgroup kernel above by ~n/12 block steps of dependent fp64 chains and L2 round trips on ONE CU (2.1 ms at
// n = 1200).  Here the factorisation is tiled 64 x 64 and spread over the chip, three launches per tile column k:
//   chol_potrf_kernel  one workgroup: wave 0 factors the diagonal tile in registers (lane = row, right-looking: pivot
//                      by readlane, rsqrt from an fp32 seed + one fp64 Newton step, rank-1 update with the column
//                      broadcast lane by lane), then the workgroup inverts the factor (16 x 16 diagonal blocks by
//                      substitution, off-diagonal blocks level by level) and leaves L^-1 in the workspace;
//   chol_trsm_kernel   one workgroup per tile row below: X = A L^-T as a 64^3 product on the fp64 matrix cores;
//   chol_syrk_kernel   one workgroup per tile pair (i >= j > k): A_ij -= X_i X_j^T, same tiles, same cores.
// Row n (the rhs) rides along as a row of the last tile row, so the forward substitution is part of the factorisation;
// chol_backsub_kernel (one workgroup) then solves L^T x = y tile column by tile column with the stored inverses and
// retracts.  LM damping is added by potrf when it loads its tile (the trailing updates only subtract from later tiles,
// so the order is immaterial).  A dependent fp64 operation costs ~40 cycles on this part: the pivot chain alone is
// ~0.15 us per column - the floor of any Cholesky here - which is why the diagonal tile stays in one wave's registers.
constexpr int CT = 64;

__device__ __forceinline__ double rsqrt_seeded(double x) {
  // branch free (the callers are long fully unrolled blocks): pivots of a damped normal matrix are far inside the float
  // range; should one not be, the seed is clamped and the two Newton steps still converge from within a factor 2^64
  const float xf = fminf(fmaxf((float)x, 1e-30f), 1e30f);
  double r = (double)__builtin_amdgcn_rsqf(xf);  // 23 bits
  const double hx = 0.5 * x;
  r = r * __builtin_fma(-hx * r, r, 1.5);          // ~45 bits
  r = r * __builtin_fma(-hx * r, r, 1.5);          // full fp64
  return r;
}

__device__ __forceinline__ bool chol_active(const BAArgs& a, int& n) {
  n = a.w.info[3];
  return n > CT_MIN_N && a.w.info[5] == 0;
}

__global__ __launch_bounds__(256) void chol_potrf_kernel(BAArgs a, int k) {
  int n;
  if (!chol_active(a, n)) return;
  const int c0 = CT * k;
  if (c0 >= n) return;
  const int bw = min(CT, n - c0);
  const BAWs& w = a.w;
  const vipe_ba_params& prm = a.p;
  const int ld = w.ld, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  double* S = w.S;
  __shared__ double Ls[CT][CT + 1];   // the factor tile (lower), identity beyond bw
  __shared__ double Li[CT][CT + 1];   // its inverse (lower)
  __shared__ int fail;
  if (t == 0) fail = 0;
  if (k == 0 && t == 0) w.info[7] = 0;  // failure flag of this factorisation
  __syncthreads();
  // the tile travels global <-> LDS with all 256 threads (row segments, coalesced), LM damping (matrix.py:179-186: poses
  // (lambda, ep); intrinsics 1e-6; rig rotations 1e-4) added on the way in.  The last tile column of a system with
  // n % 64 != 0 shares its tile row with the rhs (row n = c0 + bw): row bw of the tile carries it through the
  // factorisation as one more row below the diagonal (its own "diagonal" entry is a dummy 1).
  const bool has_rhs = bw < CT && c0 + bw == n;
  {
    const int n_free = w.info[0];
    double v[16], hd[16];  // all loads of the thread in flight before the first LDS store
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      v[q] = (r < bw && c <= r) ? S[(int64_t)(c0 + r) * ld + c0 + c] : (c == r ? 1.0 : 0.0);
      if (has_rhs && r == bw && c < bw) v[q] = S[(int64_t)n * ld + c0 + c];
      hd[q] = (r < bw && c == r && !a.droid) ? w.Hd[c0 + r] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      if (r < bw && c == r) {
        const int g = c0 + r;
        const bool pose = g < 6 * n_free, rigrow = a.mv && g >= 6 * n_free + a.nintr;
        const double ep = pose ? (double)prm.pose_ep : (rigrow ? 1e-4 : 1e-6);
        const double lam = pose ? (double)prm.pose_damping : (rigrow ? 1e-4 : 1e-6);
        v[q] += ep + lam * (a.droid ? v[q] : hd[q]);
      }
      Ls[r][c] = v[q];
    }
  }
  __syncthreads();
  // Blocked right-looking factorisation, four panels of 16 columns.  Panel: wave 0, one tile row per lane, the 16 panel
  // entries of the row in registers; per column the pivot by readlane (compile-time lane), rsqrt from an fp32 seed + two
  // fp64 Newton steps, the finished column published to LDS and read back as BROADCAST reads for the rank-1 update of
  // the remaining panel columns only (<= 14 fused multiply-adds per lane instead of <= 62 over the whole tile row: the
  // dependent pivot chain, ~0.15 us per column, is what is left).  Trailing update: all four waves, 16 x 16 tiles of the
  // lower triangle, A_ij -= P_i P_j^T on the fp64 matrix cores straight in LDS.
  {
    double* colb = &Li[0][0];  // scratch: two column buffers of 64 doubles (Li is not in use yet)
    bool bad = false;
    const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int pnl = 0; pnl < 4; ++pnl) {
      const int p0 = 16 * pnl;
      if (wave == 0) {
        const int r = lane;
        double ar[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) ar[c] = Ls[r][p0 + c];
        double d = readlane_f64(ar[0], p0);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int j = p0 + jj;
          const bool okp = d > 0.0;
          bad |= (j < bw) & !okp;
          d = okp ? d : 1.0;
          const double rl = rsqrt_seeded(d);
          const double lj = r == j ? d * rl : (r > j ? ar[jj] * rl : 0.0);
          ar[jj] = lj;
          // the NEXT pivot only needs lane j + 1's own entry of this column: form it ahead of the LDS round trip
          if (jj + 1 < 16) {
            ar[jj + 1] = __builtin_fma(-lj, readlane_f64(lj, j + 1), ar[jj + 1]);
            d = readlane_f64(ar[jj + 1], j + 1);
          }
          if (jj + 2 < 16) {
            double* cb = colb + (jj & 1) * CT;
            cb[r] = lj;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = jj + 2; c < 16; ++c) ar[c] = __builtin_fma(-lj, cb[p0 + c], ar[c]);
          }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) Ls[r][p0 + c] = ar[c];  // rows above the diagonal hold zeros in finished columns
      }
      __syncthreads();
      if (pnl < 3) {
        // tiles (ti, tj), pnl < tj <= ti <= 3, numbered ti (ti + 1) / 2 + tj relative to pnl + 1
        const int nt = (3 - pnl) * (4 - pnl) / 2;
        for (int e = wave; e < nt; e += 4) {
          int ti = 0;
          while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
          const int tj = e - ti * (ti + 1) / 2;
          const int R = 16 * (pnl + 1 + ti), C = 16 * (pnl + 1 + tj);
          double4c acc;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) acc[r4] = Ls[R + kq + 4 * r4][C + l16];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ls[R + l16][p0 + 4 * s4 + kq], Ls[C + l16][p0 + 4 * s4 + kq], acc, 0, 0, 0);
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) Ls[R + kq + 4 * r4][C + l16] = acc[r4];
        }
        __syncthreads();
      }
    }
    if (wave == 0 && bad) fail = 1;
    // the rhs row leaves for the workspace; the tile itself keeps the factor only (identity beyond bw, zeros above the
    // diagonal - the trailing updates of the diagonal 16 x 16 tiles wrote there)
    if (has_rhs && t < bw) S[(int64_t)n * ld + c0 + t] = Ls[bw][t];
    __syncthreads();
    for (int i = t; i < CT * CT; i += 256) {
      const int r = i >> 6, c = i & 63;
      if (r >= bw || c > r) Ls[r][c] = c == r ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  for (int i = t; i < CT * CT; i += 256) {
    const int r = i >> 6, c = i & 63;
    if (r < bw && c <= r) S[(int64_t)(c0 + r) * ld + c0 + c] = Ls[r][c];
  }
  // ---- inverse of the factor tile.  (1) the four 16 x 16 diagonal blocks, one thread per column: forward substitution,
  //      column oriented - as soon as x[i] is known every later row's partial sum takes its term, so the dependent chain
  //      per step is one multiply and one fused multiply-add (a row-oriented sum is a chain of i of them)
  __syncthreads();  // (the tile store above read Ls; Li's first rows served as column buffers)
  if (t < 64) {
    const int b = t >> 4, cc = t & 15, o = 16 * b;
    double sv[16], x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sv[i] = i == cc ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] = i >= cc ? sv[i] / Ls[o + i][o + i] : 0.0;
#pragma unroll
      for (int m = i + 1; m < 16; ++m) sv[m] = __builtin_fma(-Ls[o + m][o + i], x[i], sv[m]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Li[o + i][o + cc] = x[i];
  }
  for (int i = t; i < CT * CT; i += 256) {  // zero the strictly upper part and the off-diagonal blocks (filled below)
    const int r = i >> 6, c = i & 63;
    if ((r >> 4) != (c >> 4)) Li[r][c] = 0.0;
  }
  __syncthreads();
  // (2) off-diagonal blocks by distance d = 1..3: Linv(i,j) = -Dinv_i * sum_{m=j}^{i-1} L(i,m) Linv(m,j), one wave per
  //     block, both products on the fp64 matrix cores (T travels through LDS between them: D layout -> B operand)
  __shared__ double Tm[3][16][17];
  {
    const int l16 = lane & 15, kq = lane >> 4;
    for (int d = 1; d < 4; ++d) {
      const int bj = wave, bi = bj + d;  // blocks (bi, bj), bj = 0 .. 3 - d
      if (bi < 4) {
        double4c acc = {0.0, 0.0, 0.0, 0.0};
        for (int m = bj; m < bi; ++m)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ls[16 * bi + l16][16 * m + 4 * s4 + kq], Li[16 * m + 4 * s4 + kq][16 * bj + l16], acc, 0, 0, 0);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) Tm[bj][kq + 4 * r4][l16] = acc[r4];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double4c acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Li[16 * bi + l16][16 * bi + 4 * s4 + kq], Tm[bj][4 * s4 + kq][l16], acc2, 0, 0, 0);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) Li[16 * bi + kq + 4 * r4][16 * bj + l16] = acc2[r4];
      }
      __syncthreads();
    }
  }
  double* Wk = w.Wi + (int64_t)k * CT * CT;
  for (int i = t; i < CT * CT; i += 256) Wk[i] = Li[i >> 6][i & 63];
  if (t == 0 && fail) w.info[7] = 1;
}

// X = A L^-T for the tile rows below tile k; grid = tile rows (exits beyond the matrix)
__global__ __launch_bounds__(256) void chol_trsm_kernel(BAArgs a, int k) {
  int n;
  if (!chol_active(a, n)) return;
  const int c0 = CT * k, R0 = CT * (k + 1 + blockIdx.x);
  if (c0 >= n || R0 > n) return;
  const BAWs& w = a.w;
  const int ld = w.ld, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bw = min(CT, n - c0), nr = min(CT, n + 1 - R0);  // rows R0 .. R0 + nr - 1 (row n = rhs)
  __shared__ double As[CT][CT + 2];  // pitch 66 doubles: the 16 x 4 operand fragments of a wave spread over all banks
  __shared__ double Ls[CT][CT + 2];
  const double* Wk = w.Wi + (int64_t)k * CT * CT;
  {
    // all 32 loads of a thread in flight before the first LDS store (a load -> store loop is one L2 round trip per
    // iteration: most of this kernel's 10 us)
    double va[16], vl[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      va[q] = (r < nr && c < bw) ? w.S[(int64_t)(R0 + r) * ld + c0 + c] : 0.0;
      vl[q] = Wk[i];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      As[r][c] = va[q];
      Ls[r][c] = vl[q];
    }
  }
  __syncthreads();
  // wave w: rows 16 w .. 16 w + 15; X[r][c] = sum_m A[r][m] Linv[c][m]
  const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int tc = 0; tc < 4; ++tc) {
    double4c acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(As[16 * wave + l16][4 * s4 + kq], Ls[16 * tc + l16][4 * s4 + kq], acc, 0, 0, 0);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int rr = 16 * wave + kq + 4 * r4, cc = 16 * tc + l16;
      if (rr < nr && cc < bw) w.S[(int64_t)(R0 + rr) * ld + c0 + cc] = acc[r4];
    }
  }
}

// A_ij -= X_i X_j^T for all tile pairs k < j <= i; grid.x = pairs of the largest possible matrix (extra blocks exit)
__global__ __launch_bounds__(256) void chol_syrk_kernel(BAArgs a, int k) {
  int n;
  if (!chol_active(a, n)) return;
  const int c0 = CT * k;
  if (c0 >= n) return;
  int ti = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
  while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ++ti;
  while (ti * (ti + 1) / 2 > (int)blockIdx.x) --ti;
  const int tj = blockIdx.x - ti * (ti + 1) / 2;
  const int Ri = CT * (k + 1 + ti), Rj = CT * (k + 1 + tj);
  if (Ri > n || Rj >= n) return;  // row tile must hold a row <= n, column tile a column < n
  const BAWs& w = a.w;
  const int ld = w.ld, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bw = min(CT, n - c0), nri = min(CT, n + 1 - Ri), ncj = min(CT, n - Rj);
  __shared__ double Xi[CT][CT + 2];
  __shared__ double Xj[CT][CT + 2];
  const int l16 = lane & 15, kq = lane >> 4;
  // every load of the thread - the two operand tiles and the 16 entries of A_ij it will update - is in flight before
  // the first dependent instruction (load -> LDS store loops and the read-modify-write at the end were one L2 round
  // trip per iteration each)
  double aold[4][4];
  {
    double vi[16], vj[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      vi[q] = (r < nri && c < bw) ? w.S[(int64_t)(Ri + r) * ld + c0 + c] : 0.0;
      vj[q] = (r < ncj && c < bw) ? w.S[(int64_t)(Rj + r) * ld + c0 + c] : 0.0;
    }
#pragma unroll
    for (int tc = 0; tc < 4; ++tc)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int rr = 16 * wave + kq + 4 * r4, cc = 16 * tc + l16;
        aold[tc][r4] = (rr < nri && cc < ncj && Rj + cc <= Ri + rr) ? w.S[(int64_t)(Ri + rr) * ld + Rj + cc] : 0.0;
      }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = t + 256 * q, r = i >> 6, c = i & 63;
      Xi[r][c] = vi[q];
      Xj[r][c] = vj[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int tc = 0; tc < 4; ++tc) {
    double4c acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xi[16 * wave + l16][4 * s4 + kq], Xj[16 * tc + l16][4 * s4 + kq], acc, 0, 0, 0);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int rr = 16 * wave + kq + 4 * r4, cc = 16 * tc + l16;
      if (rr < nri && cc < ncj && Rj + cc <= Ri + rr) w.S[(int64_t)(Ri + rr) * ld + Rj + cc] = aold[tc][r4] - acc[r4];
    }
  }
}

__global__ __launch_bounds__(512) void chol_backsub_kernel(BAArgs a) {
  int n;
  if (!chol_active(a, n)) return;
  const BAWs& w = a.w;
  const int ld = w.ld, t = threadIdx.x, n_free = w.info[0];
  __shared__ double Li[2][CT][CT + 1];
  __shared__ double xk[CT];
  extern __shared__ __align__(16) double ys[];  // [n]: the running right-hand side stays in LDS
  for (int i = t; i < n; i += 512) ys[i] = w.S[(int64_t)n * ld + i];
  const int Tc = (n + CT - 1) / CT;
  auto load_tile = [&](int k, int b) {
    const double* Wk = w.Wi + (int64_t)k * CT * CT;
    double v[8];  // all eight loads in flight before the first LDS store
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = Wk[t + 512 * q];
#pragma unroll
    for (int q = 0; q < 8; ++q) Li[b][(t + 512 * q) >> 6][(t + 512 * q) & 63] = v[q];
  };
  load_tile(Tc - 1, (Tc - 1) & 1);
  __syncthreads();
  for (int k = Tc - 1; k >= 0; --k) {
    const int c0 = CT * k, bw = min(CT, n - c0), b = k & 1;
    // x = L^-T y: x[c] = sum_{m >= c} Linv[m][c] y[m]; 8 lanes per column, combined by DPP-free shuffles
    {
      const int c = t >> 3, part = t & 7;
      double sacc = 0.0;
      if (c < bw)
        for (int m = c + part; m < bw; m += 8) sacc = __builtin_fma(Li[b][m][c], ys[c0 + m], sacc);
      sacc += __shfl_xor(sacc, 1, 8);
      sacc += __shfl_xor(sacc, 2, 8);
      sacc += __shfl_xor(sacc, 4, 8);
      if (part == 0 && c < bw) xk[c] = sacc;
    }
    if (k > 0) load_tile(k - 1, b ^ 1);  // next tile's inverse: independent of x
    __syncthreads();
    if (t < bw) ys[c0 + t] = xk[t];
    for (int c = t; c < c0; c += 512) {  // y[c] -= sum_r L[c0 + r][c] x[r]
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const double* col = w.S + (int64_t)c0 * ld + c;
      int r = 0;
      for (; r + 15 < bw; r += 16) {  // sixteen rows of the column in flight per pass (each is its own cache line)
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = col[(int64_t)(r + q) * ld];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
          s0 = __builtin_fma(v[q], xk[r + q], s0);
          s1 = __builtin_fma(v[q + 1], xk[r + q + 1], s1);
          s2 = __builtin_fma(v[q + 2], xk[r + q + 2], s2);
          s3 = __builtin_fma(v[q + 3], xk[r + q + 3], s3);
        }
      }
      for (; r < bw; ++r) s0 = __builtin_fma(col[(int64_t)r * ld], xk[r], s0);
      ys[c] -= (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
  }
  const bool bad = w.info[7] != 0;
  if (t == 0 && bad) w.info[2] += 1;
  for (int dd = t; dd < n; dd += 512) {
    double x = ys[dd];
    if (bad || !(x == x)) x = 0.0;  // zero step on a failed factorisation
    w.dx[dd] = (float)x;
  }
  __syncthreads();
  apply_retraction(a, t, 512, n_free);
}

// host side: the launches of one tiled solve, sized for the largest system the workspace can hold (blocks beyond the
// actual n exit at once; n itself lives on the device)
inline void launch_tiled_cholesky(const BAArgs& a, hipStream_t s) {
  const int nmax = a.w.ld - 1;
  if (nmax <= CT_MIN_N || nmax > 8000) return;  // (the back substitution keeps the rhs, up to 8000 doubles, in LDS)
  const int T = (nmax + 1 + CT - 1) / CT;  // tile rows incl. the rhs row
  const int Tc = (nmax + CT - 1) / CT;
  for (int k = 0; k < Tc; ++k) {
    chol_potrf_kernel<<<1, 256, 0, s>>>(a, k);
    const int m = T - 1 - k;
    if (m > 0) {
      chol_trsm_kernel<<<m, 256, 0, s>>>(a, k);
      chol_syrk_kernel<<<m * (m + 1) / 2, 256, 0, s>>>(a, k);
    }
  }
  chol_backsub_kernel<<<1, 512, sizeof(double) * (size_t)(nmax + 8), s>>>(a);
}

// ------------------------------------------------------------------------------------------------ retract

template <int F>
__global__ __launch_bounds__(TILE) void ba_retract_kernel(BAArgs a) {
  const BAWs& w = a.w;
  {
    // The reduced system has been solved (this kernel only reads dx): clear S and Hd for the next accumulation here,
    // spread over the whole grid, instead of two memset launches per Gauss-Newton iteration.
    const int64_t nthr = (int64_t)gridDim.x * gridDim.y * TILE;
    const int64_t gid = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * TILE + threadIdx.x;
    const int64_t ns = (int64_t)w.ld * w.ld;
    for (int64_t i = gid; i < ns; i += nthr) w.S[i] = 0.0;
    for (int64_t i = gid; i < (int64_t)w.ld - 1; i += nthr) w.Hd[i] = 0.0;
  }
  const int k = blockIdx.y;
  const int flags = w.fflags[k];
  if (!(flags & 2)) return;
  const int P = a.P, V = a.p.n_views;
  const int p = blockIdx.x * TILE + threadIdx.x;
  if (p >= P) return;
  const int64_t kp = (int64_t)k * P + p;
  if (flags & 8) {
    // DROID: frame of [t0, t1) without terms - only the depth prior acts on it (geom_kernels.cu:1359-1369)
    float C = 0.f, wz = 0.f;
    finish_disp(a, k, p, P, flags, a.disps[kp], C, wz);
    const float dz = wz / C;
    a.disps[kp] += dz;
    if (a.dz_out) a.dz_out[(int64_t)w.krow[k] * P + p] = dz;
    return;
  }
  const int beg = w.rowptr[k], end = w.rowptr[k + 1];
  float rhs = w.wv[kp];
  const int si = w.pose_slot[k / V];
  const int n_free = w.info[0];
  // DROID leaves pose slot 0 out of the back-substitution (EvT6x1_kernel: idx <= 0 returns, geom_kernels.cu:1085)
  const int smin = a.droid ? 1 : 0;
  if (si >= smin) {
#pragma unroll
    for (int q = 0; q < 6; ++q) rhs -= w.Ekk[((int64_t)k * 6 + q) * P + p] * w.dx[6 * si + q];
  }
  for (int c = beg; c < end; ++c) {
    const int e = w.order[c];
    const int pj = (int)a.pj[e];
    const int sj = ((int)a.pi[e] == pj) ? -1 : w.pose_slot[pj];
    if (sj < smin) continue;
#pragma unroll
    for (int q = 0; q < 6; ++q) rhs -= w.Ej[((int64_t)e * 6 + q) * P + p] * w.dx[6 * sj + q];
  }
  if constexpr (F > 0) {
#pragma unroll
    for (int f = 0; f < F; ++f) rhs -= w.Ef[((int64_t)k * 2 + f) * P + p] * w.dx[6 * n_free + f];
  }
  if (a.mv) {
    for (int f = 0; f < a.ntail; ++f) rhs -= w.Et[((int64_t)k * a.ntail + f) * P + p] * w.dx[6 * n_free + f];
  }
  float dz = rhs / w.C[kp];
  if (!a.droid && dz > 10.0f) dz = 0.0f;  // retractor.py:41
  a.disps[kp] += dz;
  if (a.dz_out) a.dz_out[(int64_t)w.krow[k] * P + p] = dz;
}

__global__ void clamp_min_kernel(float* __restrict__ x, int64_t n, float lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = fmaxf(x[i], lo);  // NaN stays NaN? fmaxf(NaN, lo) = lo; torch.clamp keeps NaN - disparities are finite here
}

// Event `after the accumulate kernels of this iteration` on the BA stream; see overlap_piece
hipEvent_t overlap_event() {
  static thread_local hipEvent_t ev[64] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  hipEvent_t& e = ev[d & 63];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
  return e;
}

// vipe_overlap_fn protocol: piece `it` goes to the caller's stream behind the event after this iteration's accumulate
// kernels, i.e. it becomes eligible together with the single-workgroup solve
int overlap_piece(const vipe_ba_params& p, hipEvent_t ev, int piece, int n_pieces, bool gated) {
  hipStream_t side = (hipStream_t)p.overlap_stream;
  // (Rounds 2-3 put a 4 us wall-clock delay kernel behind the event so that the solve kernel would win the race for a free
  // CU.  Round 4 A/B on the headline, three runs each: 256.5 / 256.1 / 256.7 it/s with it, 257.5 / 256.5 / 257.9 without,
  // 254.4 / 253.6 / 254.8 with 16 us - the two-chain band solve and the staged share of 0.5 left nothing for it to fix.
  // Removed: the ordering is the event alone.)
  if (gated && ev && hipStreamWaitEvent(side, ev, 0) != hipSuccess) return VIPE_EINVAL;
  return p.overlap_fn(p.overlap_user, piece, n_pieces, p.overlap_stream);
}

template <int CAM, int F>
int run_iters(const BAArgs& a, hipStream_t s, int* pieces_done) {
  const int tiles = (a.P + TILE - 1) / TILE;
  const size_t sbytes = sizeof(double) * (size_t)a.w.ld * a.w.ld;
  const size_t nmax = (size_t)a.w.ld - 1;
  // LDS panel of the solve kernel: [NB][panel_cap] doubles next to the fixed part, up to ~150 KB
  const size_t fixed = ((sizeof(SolveLds) + 15) / 16) * 16;
  int panel_cap = (int)std::min<size_t>(nmax + 1, (150 * 1024 - fixed) / (NB * sizeof(double)));
  panel_cap = (panel_cap + 3) & ~3;
  const size_t solve_lds = fixed + (size_t)NB * panel_cap * sizeof(double);
  const size_t band_lds = 158 * 1024;
  static std::atomic<uint64_t> attr_set{0};  // bit d: set on device d
  vipe_once_per_device(attr_set, [] {
    (void)hipFuncSetAttribute((const void*)ba_solve_band_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ba_solve_dense_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)chol_backsub_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  });
  static std::atomic<uint64_t> tmpl_set{0};  // one per <CAM, F> instantiation of this function
  vipe_once_per_device(tmpl_set, [] {
    (void)hipFuncSetAttribute((const void*)ba_accum_mfma_kernel<CAM, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)accum_mfma_lds());
    (void)hipFuncSetAttribute((const void*)ba_walk_kernel<CAM, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_lds());
    (void)hipFuncSetAttribute((const void*)ba_walk_rig_kernel<CAM, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_rig_lds());
    (void)hipFuncSetAttribute((const void*)ba_walk_rig_kernel<CAM, RG_VMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)walk_rig_lds());
  });
  // S / Hd start each accumulation zeroed: by ba_retract_kernel of the previous iteration when it runs (not motion_only),
  // also across calls when the caller vouches for the workspace (reuse_plan: same key, hence the same motion_only), else
  // by memsets
  const bool retract_clears = !a.p.motion_only;
  const bool overlap = a.p.overlap_stream && a.p.overlap_fn;
  hipEvent_t ev = overlap ? overlap_event() : nullptr;
  for (int it = 0; it < a.p.n_iters; ++it) {
    if (!(retract_clears && (it > 0 || a.p.reuse_plan))) {
      hipError_t e1 = hipMemsetAsync(a.w.S, 0, sbytes, s);
      hipError_t e2 = hipMemsetAsync(a.w.Hd, 0, sizeof(double) * nmax, s);
      if (e1 != hipSuccess || e2 != hipSuccess) return (int)(e1 != hipSuccess ? e1 : e2);
    }
    if (a.mv) {
      // multi-view rigs: local-block walk, Schur Gram over the stacked E rows, global-memory Cholesky with the dense tail
      if (a.p.n_views <= 4) ba_walk_rig_kernel<CAM, 4><<<dim3(tiles, a.nF), TILE, walk_rig_lds(), s>>>(a);
      else ba_walk_rig_kernel<CAM, RG_VMAX><<<dim3(tiles, a.nF), TILE, walk_rig_lds(), s>>>(a);
      ba_schur_kernel<0><<<dim3(SC_GRID, a.nF), TILE, 0, s>>>(a);
      if (ev && hipEventRecord(ev, s) != hipSuccess) return VIPE_EINVAL;
      ba_solve_kernel<<<1, SOLVE_T, solve_lds, s>>>(a, panel_cap);
      launch_tiled_cholesky(a, s);
      if (overlap) {
        const int rc = overlap_piece(a.p, ev, it, a.p.n_iters, true);
        if (rc != VIPE_OK) return rc;
        ++*pieces_done;
      }
      if (!a.p.motion_only) ba_retract_kernel<0><<<dim3(tiles, a.nF), TILE, 0, s>>>(a);
      continue;
    }
    // path_hint (vipe_ba_params): what the caller learnt from an earlier call with this plan; 0 launches everything
    const int hint = a.force_general ? 0 : a.p.path_hint;
    const bool prof = a.p.profile_ev0 && a.p.profile_ev1 &&
                      it == (a.p.profile_iter < 0 || a.p.profile_iter >= a.p.n_iters ? a.p.n_iters - 1 : a.p.profile_iter);
    if (prof && hipEventRecord((hipEvent_t)a.p.profile_ev0, s) != hipSuccess) return VIPE_EINVAL;
    if (!(hint & 2)) ba_accum_mfma_kernel<CAM, F><<<dim3(tiles, a.nF), TILE, accum_mfma_lds(), s>>>(a);
    if (!(hint & 1)) {
      ba_walk_kernel<CAM, F><<<dim3(tiles, a.nF), TILE, walk_lds(), s>>>(a);
      ba_schur_kernel<F><<<dim3(SC_GRID, a.nF), TILE, 0, s>>>(a);
    }
    if (prof && hipEventRecord((hipEvent_t)a.p.profile_ev1, s) != hipSuccess) return VIPE_EINVAL;
    if (ev && hipEventRecord(ev, s) != hipSuccess) return VIPE_EINVAL;
    if (!(hint & 8)) ba_solve_band_kernel<<<1, 2 * BAND_T, band_lds, s>>>(a, (int)(band_lds / sizeof(double)));
    if (!(hint & 16)) ba_solve_dense_kernel<<<1, DN_T, band_lds, s>>>(a, (int)(band_lds / sizeof(double)));
    if (!(hint & 4)) {
      ba_solve_kernel<<<1, SOLVE_T, solve_lds, s>>>(a, panel_cap);
      launch_tiled_cholesky(a, s);
    }
    if (overlap) {
      const int rc = overlap_piece(a.p, ev, it, a.p.n_iters, true);
      if (rc != VIPE_OK) return rc;
      ++*pieces_done;
    }
    if (!a.p.motion_only) ba_retract_kernel<F><<<dim3(tiles, a.nF), TILE, 0, s>>>(a);
  }
  return vipe_launch_status();
}

}  // namespace

VIPE_EXPORT int64_t vipe_dense_ba_workspace_bytes(const vipe_ba_params* p) {
  if (!p || p->n_poses <= 0 || p->n_views <= 0 || p->ht <= 0 || p->wd <= 0 || p->M < 0) return VIPE_EINVAL;
  return (int64_t)carve(*p, nullptr, nullptr);
}

VIPE_EXPORT int vipe_dense_ba(const vipe_ba_params* p, float* d_poses, float* d_disps, const float* d_disps_sens,
                              float* d_intrinsics, float* d_rig, const float* d_target, const float* d_weight,
                              const float* d_disp_damping, const int64_t* d_pi, const int64_t* d_qi,
                              const int64_t* d_pj, const int64_t* d_qj, const int64_t* d_di, void* d_workspace,
                              int64_t workspace_bytes, int* d_info, void* stream) {
  VIPE_CHECK_ARG(p && d_poses && d_disps && d_disps_sens && d_intrinsics && d_rig && d_disp_damping && d_workspace);
  VIPE_CHECK_ARG(p->n_poses > 0 && p->n_views > 0 && p->ht > 0 && p->wd > 0 && p->M >= 0 && p->n_iters >= 0);
  VIPE_CHECK_ARG(p->t0 <= p->t1 && p->intr_factor > 0);
  VIPE_CHECK_ARG(p->camera == VIPE_CAM_PINHOLE || p->camera == VIPE_CAM_MEI);
  VIPE_CHECK_ARG(p->M == 0 || (d_target && d_weight && d_pi && d_qi && d_pj && d_qj && d_di));
  if (is_multiview(*p) && p->n_views > RG_VMAX) return VIPE_EUNSUPPORTED;  // rigs of up to 8 cameras
  if ((int64_t)p->n_poses * p->n_views > 65535) return VIPE_EINVAL;
  BAArgs a;
  a.p = *p;
  if ((int64_t)carve(*p, (char*)d_workspace, &a.w) > workspace_bytes) return VIPE_ENOSPACE;
  a.poses = d_poses; a.disps = d_disps; a.intr = d_intrinsics; a.rig = d_rig;
  a.sens = d_disps_sens; a.target = d_target; a.weight = d_weight; a.eta = d_disp_damping;
  a.pi = d_pi; a.qi = d_qi; a.pj = d_pj; a.qj = d_qj; a.di = d_di;
  a.P = p->ht * p->wd;
  a.nF = p->n_poses * p->n_views;
  a.D = p->camera == VIPE_CAM_MEI ? 1 : 0;
  a.mv = is_multiview(*p) ? 1 : 0;
  a.nintr = tail_intr(*p);
  a.ntail = a.nintr + tail_rig(*p);
  a.force_general = (p->solver_options & VIPE_BA_OPT_GENERAL_ACCUMULATE) != 0;
  a.band2 = !(p->solver_options & VIPE_BA_OPT_ONE_CHAIN);
  a.droid = 0;
  a.dz_out = nullptr;
  hipStream_t s = as_stream(stream);
  int rc = VIPE_OK, pieces_done = 0;
  if (p->M > 0 && p->n_iters > 0) {
    ba_sens_kernel<<<a.nF, 256, 0, s>>>(d_disps_sens, a.w.sens_sum, a.P, a.w.info);
    if (!p->reuse_plan) ba_plan_kernel<<<1, 1024, 0, s>>>(a);
    const int F = p->optimize_intrinsics ? 1 + a.D : 0;
    if (p->camera == VIPE_CAM_PINHOLE) rc = F ? run_iters<VIPE_CAM_PINHOLE, 1>(a, s, &pieces_done) : run_iters<VIPE_CAM_PINHOLE, 0>(a, s, &pieces_done);
    else rc = F ? run_iters<VIPE_CAM_MEI, 2>(a, s, &pieces_done) : run_iters<VIPE_CAM_MEI, 0>(a, s, &pieces_done);
    if (rc != VIPE_OK) return rc;
    if (d_info) {
      hipError_t e = hipMemcpyAsync(d_info, a.w.info, 8 * sizeof(int), hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) return (int)e;
    }
  }
  // vipe_overlap_fn: exactly max(n_iters, 1) pieces, also when there was nothing to optimise
  if (p->overlap_stream && p->overlap_fn) {
    const int n_pieces = p->n_iters > 0 ? p->n_iters : 1;
    for (; pieces_done < n_pieces; ++pieces_done) {
      rc = overlap_piece(*p, nullptr, pieces_done, n_pieces, false);
      if (rc != VIPE_OK) return rc;
    }
  }
  // buffer.py:525: disps.clamp_(min=1e-3) over the whole buffer handed in
  const int64_t nd = (int64_t)a.nF * a.P;
  clamp_min_kernel<<<(int)std::min<int64_t>((nd + 255) / 256, 2048), 256, 0, s>>>(d_disps, nd, 1e-3f);
  return vipe_launch_status();
}

// ---- slam_ext.ba with the DROID signature (dormant in the reference; geom_kernels.cu:1273-1404).  Same kernels as the
// live dense BA with the DROID semantics switched on (BAArgs::droid; oracle/droid_ba.py lists the differences).
namespace {
size_t droid_extra_bytes(int E) { return align_up(8 * (size_t)(E + 1)) + align_up(7 * 4); }
vipe_ba_params droid_params(int n_poses, int ht, int wd, int E, int t0, int t1, int iterations, float lm, float ep,
                            int motion_only) {
  vipe_ba_params p = {};
  p.n_poses = n_poses; p.n_views = 1; p.ht = ht; p.wd = wd; p.M = E; p.t0 = t0; p.t1 = t1; p.n_iters = iterations;
  p.pose_damping = lm; p.pose_ep = ep; p.motion_only = motion_only; p.limited_disp = 0; p.optimize_intrinsics = 0;
  p.optimize_rig_rotation = 0; p.camera = VIPE_CAM_PINHOLE; p.alpha = 0.05f; p.weight_scale = 0.001f; p.intr_factor = 1.0f;
  p.reuse_plan = 0; p.path_hint = 0;
  return p;
}
}  // namespace

VIPE_EXPORT int64_t vipe_ba_workspace_bytes(int n_poses, int ht, int wd, int E) {
  if (n_poses <= 0 || ht <= 0 || wd <= 0 || E < 0) return VIPE_EINVAL;
  const vipe_ba_params p = droid_params(n_poses, ht, wd, E, 0, 0, 0, 0.f, 0.f, 0);
  return (int64_t)(carve(p, nullptr, nullptr) + droid_extra_bytes(E));
}

VIPE_EXPORT int vipe_ba(float* d_poses, float* d_disps, const float* d_intrinsics, const float* d_disps_sens,
                        const float* d_targets, const float* d_weights, const float* d_eta, const int64_t* d_ii,
                        const int64_t* d_jj, int n_poses, int ht, int wd, int E, int n_eta, int t0, int t1,
                        int iterations, float lm, float ep, int motion_only, float* d_dx, float* d_dz,
                        void* d_workspace, int64_t workspace_bytes, void* stream) {
  VIPE_CHECK_ARG(d_poses && d_disps && d_intrinsics && d_disps_sens && d_eta && d_workspace && d_dx && d_dz);
  VIPE_CHECK_ARG(n_poses > 0 && ht > 0 && wd > 0 && E >= 0 && iterations >= 0 && n_eta >= 0);
  VIPE_CHECK_ARG(0 <= t0 && t0 <= t1 && t1 <= n_poses);
  VIPE_CHECK_ARG(E == 0 || (d_targets && d_weights && d_ii && d_jj));
  if (n_poses > 65535) return VIPE_EINVAL;
  BAArgs a;
  a.p = droid_params(n_poses, ht, wd, E, t0, t1, iterations, lm, ep, motion_only);
  const size_t base = carve(a.p, (char*)d_workspace, &a.w);
  if ((int64_t)(base + droid_extra_bytes(E)) > workspace_bytes) return VIPE_ENOSPACE;
  int64_t* zeros = (int64_t*)((char*)d_workspace + base);
  float* rig = (float*)((char*)d_workspace + base + align_up(8 * (size_t)(E + 1)));
  hipStream_t s = as_stream(stream);
  const float rig_id[7] = {0, 0, 0, 0, 0, 0, 1};
  hipError_t e0 = hipMemsetAsync(zeros, 0, 8 * (size_t)(E + 1), s);
  hipError_t e1 = hipMemcpyAsync(rig, rig_id, sizeof(rig_id), hipMemcpyHostToDevice, s);
  const int64_t P = (int64_t)ht * wd;
  hipError_t e2 = hipMemsetAsync(d_dx, 0, sizeof(float) * 6 * (size_t)(t1 - t0), s);
  hipError_t e3 = hipMemsetAsync(d_dz, 0, sizeof(float) * (size_t)n_eta * P, s);
  if (e0 != hipSuccess || e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return VIPE_EINVAL;
  a.poses = d_poses; a.disps = d_disps; a.intr = (float*)d_intrinsics; a.rig = rig;
  a.sens = d_disps_sens; a.target = d_targets; a.weight = d_weights; a.eta = d_eta;
  a.pi = d_ii; a.qi = zeros; a.pj = d_jj; a.qj = zeros; a.di = d_ii;
  a.P = ht * wd; a.nF = n_poses; a.D = 0;
  a.mv = 0; a.nintr = 0; a.ntail = 0;
  a.force_general = 0;
  a.band2 = 1;
  a.droid = 1;
  a.dz_out = d_dz;
  if (iterations == 0 || t1 == t0) return VIPE_OK;
  ba_sens_kernel<<<a.nF, 256, 0, s>>>(d_disps_sens, a.w.sens_sum, a.P, a.w.info);
  ba_plan_kernel<<<1, 1024, 0, s>>>(a);
  int pieces_done = 0;
  const int rc = run_iters<VIPE_CAM_PINHOLE, 0>(a, s, &pieces_done);
  if (rc != VIPE_OK) return rc;
  hipError_t e4 = hipMemcpyAsync(d_dx, a.w.dx, sizeof(float) * 6 * (size_t)(t1 - t0), hipMemcpyDeviceToDevice, s);
  return e4 == hipSuccess ? vipe_launch_status() : (int)e4;
}
