// The flow-update operator (UpdateModule.forward, vipe/slam/networks/droid_net.py:467-499) as ONE library call.
//
// The operator is 13 fused convolutions + a segmented mean + two small reductions (csrc/conv_mfma.hip, corr_lookup.hip,
// aux_ops.hip).  Issued one by one from Python each launch costs 10-15 us of argument marshalling, more than most of
// these kernels run for on a frontend window (~48 edges): the keyframe frontend was launch bound (GPU idle 15-45 % of
// the time, profiles/r02_video_*).  vipe_update_operator sequences them natively from two descriptor structs the host
// fills once per operator / per edge set; vipe_update_finish is the tail of FactorGraph.update (factor_graph.py:
// 270-276: target = coords1 + delta, weight with masked frames zeroed, damping[source frames] = eta) as one kernel
// instead of five elementwise launches.
#include "common.cuh"

namespace {

// extra[e, :] = glo_b + (glo_sum[e, :] / HW) @ glo_w   ([E,128] x [128,384]): the three *_glo 1x1 convolutions of the
// ConvGRU applied to the pooled context vector (droid_net.py:392-399)
__global__ __launch_bounds__(384) void glo_gemm_kernel(const float* __restrict__ glo, const float* __restrict__ wT /* [128][384] */,
                                                       const float* __restrict__ b, float* __restrict__ extra, float inv_hw) {
  __shared__ float g[128];
  const int e = blockIdx.x, t = threadIdx.x;
  if (t < 128) g[t] = glo[(int64_t)e * 128 + t] * inv_hw;
  __syncthreads();
  float acc = b[t];
#pragma unroll 8
  for (int k = 0; k < 128; ++k) acc = __builtin_fmaf(g[k], wT[k * 384 + t], acc);
  extra[(int64_t)e * 384 + t] = acc;
}

__global__ __launch_bounds__(256) void update_finish_kernel(const float* __restrict__ coords1, const float* __restrict__ dw,
                                                            const unsigned char* __restrict__ mask, float* __restrict__ target,
                                                            float* __restrict__ weight, const float* __restrict__ eta,
                                                            const int64_t* __restrict__ du, float* __restrict__ damping,
                                                            int64_t n_px, int64_t n_eta, int P) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_px) {
    const float2 c = reinterpret_cast<const float2*>(coords1)[i];
    const float4 d = reinterpret_cast<const float4*>(dw)[i];
    const bool m = mask && mask[i];
    reinterpret_cast<float2*>(target)[i] = make_float2(c.x + d.x, c.y + d.y);
    reinterpret_cast<float2*>(weight)[i] = m ? make_float2(0.f, 0.f) : make_float2(d.z, d.w);
  }
  if (i < n_eta) damping[du[i / P] * P + i % P] = eta[i];
}

}  // namespace

namespace {
// four events per (thread, device) for the fork / join of the operator's two-stream form
hipEvent_t* operator_events() {
  static thread_local hipEvent_t ev[64][4] = {};
  int d = 0;
  (void)hipGetDevice(&d);
  hipEvent_t* e = ev[d & 63];
  for (int i = 0; i < 4; ++i)
    if (!e[i] && hipEventCreateWithFlags(&e[i], hipEventDisableTiming) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

extern "C" {

VIPE_EXPORT int vipe_update_operator(const vipe_update_weights* wt, const vipe_update_buffers* b, void* stream) {
  VIPE_CHECK_ARG(wt && b && b->E >= 0 && b->H > 0 && b->W > 0);
  if (b->E == 0) return VIPE_OK;
  const int E = b->E, H = b->H, W = b->W;
  hipStream_t s = as_stream(stream);
  int rc;
#define RUN(call)              \
  do {                         \
    rc = (call);               \
    if (rc != VIPE_OK) return rc; \
  } while (0)
  // two-stream form: s2 carries the flow encoder next to the lookup + correlation encoder, and the heads next to the
  // GraphAgg chain
  hipStream_t s2 = (hipStream_t)b->side_stream;
  hipEvent_t* ev = s2 ? operator_events() : nullptr;
  if (s2 && !ev) return VIPE_EINVAL;
  void* const stream2 = s2 ? b->side_stream : stream;
#define FORK(i)                                                                                              \
  do {                                                                                                       \
    if (s2 && (hipEventRecord(ev[i], s) != hipSuccess || hipStreamWaitEvent(s2, ev[i], 0) != hipSuccess)) return VIPE_EINVAL; \
  } while (0)
#define JOIN(i)                                                                                              \
  do {                                                                                                       \
    if (s2 && (hipEventRecord(ev[i], s2) != hipSuccess || hipStreamWaitEvent(s, ev[i], 0) != hipSuccess)) return VIPE_EINVAL; \
  } while (0)
  FORK(0);
  // encoders (droid_net.py:481-482)
  if (b->levels[0]) {
    RUN(vipe_corr_lookup_conv1x1(b->levels, b->coords, wt->corr0_w, wt->corr0_b, b->c1, 128, 0, E, H, W, b->h2, b->w2, 128,
                                 VIPE_ACT_RELU, b->slots, b->pyramid_layout, stream));
  } else {
    VIPE_CHECK_ARG(b->corr);
    RUN(vipe_conv2d_fused(b->corr, 200, 0, nullptr, 0, 0, 200, wt->corr0_w, wt->corr0_b, nullptr, 0, 0, b->c1, 128, 0, nullptr,
                          0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 200, 128, 1, 1, VIPE_ACT_RELU, 0, stream));
  }
  RUN(vipe_conv2d_fused(b->c1, 128, 0, nullptr, 0, 0, 128, wt->corr2_w, wt->corr2_b, nullptr, 0, 0, b->xbuf, 320, 128, nullptr,
                        0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 128, 128, 3, 3, VIPE_ACT_RELU, 0, stream));
  RUN(vipe_conv2d_fused(b->motn, 4, 0, nullptr, 0, 0, 4, wt->flow0_w, wt->flow0_b, nullptr, 0, 0, b->f1, 128, 0, nullptr, 0, 0,
                        nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 4, 128, 7, 7, VIPE_ACT_RELU, 0, stream2));
  RUN(vipe_conv2d_fused(b->f1, 128, 0, nullptr, 0, 0, 128, wt->flow2_w, wt->flow2_b, nullptr, 0, 0, b->xbuf, 320, 256, nullptr,
                        0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 128, 64, 3, 3, VIPE_ACT_RELU, 0, stream2));
  JOIN(1);
  const int Es = b->gate_state;  // leading edges with the hidden-state part of z|r computed ahead (vipe_update_gate_state)
  const bool staged = Es != 0;
  if (staged) VIPE_CHECK_ARG(Es > 0 && Es <= E && b->pgate && b->pzr && wt->zr_x_w);
  if (!staged) {
    // global context (droid_net.py:392-393) and its three 1x1s
    if (hipMemsetAsync(b->glo, 0, sizeof(float) * 128 * (size_t)E, s) != hipSuccess) return VIPE_EINVAL;
    RUN(vipe_conv2d_fused(b->net, 128, 0, nullptr, 0, 0, 128, wt->gw_w, wt->gw_b, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
                          b->net, 128, 0, nullptr, b->glo, nullptr, 0, 0, E, H, W, 128, 128, 1, 1, VIPE_ACT_NONE, 1, stream));
    glo_gemm_kernel<<<E, 384, 0, s>>>(b->glo, wt->glo_wT, wt->glo_b, b->extra, 1.0f / (float)(H * W));
  }
  // gates (droid_net.py:395-399); with pgate the context-feature part is the accumulators' initial value
  if (staged) {
    RUN(vipe_conv2d_fused(b->xbuf, 320, 128, nullptr, 0, 0, 192, wt->zr_x_w, wt->zr_b, b->extra, 384, 0, b->zb, 128, 0, b->rnet,
                          128, 0, b->net, 128, 0, nullptr, nullptr, b->pzr, 256, 0, Es, H, W, 192, 256, 3, 3, VIPE_ACT_NONE,
                          2 | VIPE_CONV_ACCINIT_F32, stream));
    if (Es < E) {
      const int64_t px = (int64_t)Es * H * W;  // first pixel of the unstaged edges
      auto h16 = [&](const void* p, int c) { return (const void*)((const char*)p + px * c * 2); };
      RUN(vipe_conv2d_fused(h16(b->net, 128), 128, 0, h16(b->xbuf, 320), 320, 128, 128, wt->zr_s_w, wt->zr_b,
                            b->extra + (int64_t)Es * 384, 384, 0, (void*)h16(b->zb, 128), 128, 0, (void*)h16(b->rnet, 128), 128, 0,
                            h16(b->net, 128), 128, 0, nullptr, nullptr, h16(b->pgate, 384), 384, 0, E - Es, H, W, 320, 256, 3, 3,
                            VIPE_ACT_NONE, 2, stream));
    }
    RUN(vipe_conv2d_fused(b->rnet, 128, 0, b->xbuf, 320, 128, 128, wt->q_s_w, wt->q_b, b->extra, 384, 256, b->net_out, 128, 0,
                          nullptr, 0, 0, b->net, 128, 0, b->zb, nullptr, b->pgate, 384, 256, E, H, W, 320, 128, 3, 3,
                          VIPE_ACT_NONE, 3, stream));
  } else if (b->pgate) {
    RUN(vipe_conv2d_fused(b->net, 128, 0, b->xbuf, 320, 128, 128, wt->zr_s_w, wt->zr_b, b->extra, 384, 0, b->zb, 128, 0, b->rnet,
                          128, 0, b->net, 128, 0, nullptr, nullptr, b->pgate, 384, 0, E, H, W, 320, 256, 3, 3, VIPE_ACT_NONE, 2,
                          stream));
    RUN(vipe_conv2d_fused(b->rnet, 128, 0, b->xbuf, 320, 128, 128, wt->q_s_w, wt->q_b, b->extra, 384, 256, b->net_out, 128, 0,
                          nullptr, 0, 0, b->net, 128, 0, b->zb, nullptr, b->pgate, 384, 256, E, H, W, 320, 128, 3, 3,
                          VIPE_ACT_NONE, 3, stream));
  } else {
    RUN(vipe_conv2d_fused(b->net, 128, 0, b->xbuf, 320, 0, 128, wt->zr_w, wt->zr_b, b->extra, 384, 0, b->zb, 128, 0, b->rnet, 128,
                          0, b->net, 128, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 448, 256, 3, 3, VIPE_ACT_NONE, 2, stream));
    RUN(vipe_conv2d_fused(b->rnet, 128, 0, b->xbuf, 320, 0, 128, wt->q_w, wt->q_b, b->extra, 384, 256, b->net_out, 128, 0, nullptr,
                          0, 0, b->net, 128, 0, b->zb, nullptr, nullptr, 0, 0, E, H, W, 448, 128, 3, 3, VIPE_ACT_NONE, 3, stream));
  }
  // heads + first aggregation conv on net' (droid_net.py:486-487, 418)
  RUN(vipe_conv2d_fused(b->net_out, 128, 0, nullptr, 0, 0, 128, wt->heads0_w, wt->heads0_b, nullptr, 0, 0, b->hbuf, 384, 0,
                        nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, E, H, W, 128, 384, 3, 3, VIPE_ACT_RELU, 0,
                        stream));
  const bool two_tails = s2 && b->n_src > 0;
  if (two_tails) FORK(2);
  RUN(vipe_conv2d_fused(b->hbuf, 384, 0, nullptr, 0, 0, 256, wt->heads2_w, wt->heads2_b, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0,
                        0, nullptr, 0, 0, nullptr, b->dw, nullptr, 0, 0, E, H, W, 256, 4, 3, 3, VIPE_ACT_NONE, 4,
                        two_tails ? stream2 : stream));
  if (b->n_src > 0) {
    // scatter_mean over the edges of each source node (droid_net.py:420-421), agg conv 2, eta (:422-429)
    VIPE_CHECK_ARG(b->order && b->rowptr && b->agg && b->a2 && b->eta);
    RUN(vipe_segment_mean_nhwc_f16(b->hbuf, 384, 256, b->order, b->rowptr, b->agg, b->n_src, (int64_t)H * W, 128, stream));
    RUN(vipe_conv2d_fused(b->agg, 128, 0, nullptr, 0, 0, 128, wt->agg2_w, wt->agg2_b, nullptr, 0, 0, b->a2, 128, 0, nullptr, 0, 0,
                          nullptr, 0, 0, nullptr, nullptr, nullptr, 0, 0, b->n_src, H, W, 128, 128, 3, 3, VIPE_ACT_RELU, 0,
                          stream));
    RUN(vipe_conv2d_fused(b->a2, 128, 0, nullptr, 0, 0, 128, wt->eta_w, wt->eta_b, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
                          nullptr, 0, 0, nullptr, b->eta, nullptr, 0, 0, b->n_src, H, W, 128, 1, 3, 3, VIPE_ACT_NONE, 5, stream));
  }
  if (two_tails) JOIN(3);
#undef FORK
#undef JOIN
#undef RUN
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_update_gate_state(const vipe_update_weights* wt, const vipe_update_buffers* b, const void* d_net,
                                       int parts, void* stream) {
  VIPE_CHECK_ARG(wt && b && b->E >= 0 && b->H > 0 && b->W > 0 && parts >= 1 && parts <= 3);
  if (b->E == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_net && (!(parts & 1) || (b->glo && b->extra)) && (!(parts & 2) || (b->pgate && b->pzr && wt->zr_n_w)));
  const int E = b->E, H = b->H, W = b->W;
  hipStream_t s = as_stream(stream);
  int rc;
  if (parts & 1) {
    if (hipMemsetAsync(b->glo, 0, sizeof(float) * 128 * (size_t)E, s) != hipSuccess) return VIPE_EINVAL;
    rc = vipe_conv2d_fused(d_net, 128, 0, nullptr, 0, 0, 128, wt->gw_w, wt->gw_b, nullptr, 0, 0, nullptr, 0, 0, nullptr, 0, 0,
                           d_net, 128, 0, nullptr, b->glo, nullptr, 0, 0, E, H, W, 128, 128, 1, 1, VIPE_ACT_NONE, 1, stream);
    if (rc != VIPE_OK) return rc;
    glo_gemm_kernel<<<E, 384, 0, s>>>(b->glo, wt->glo_wT, wt->glo_b, b->extra, 1.0f / (float)(H * W));
  }
  if (parts & 2) {
    rc = vipe_conv2d_fused(d_net, 128, 0, nullptr, 0, 0, 128, wt->zr_n_w, wt->zr_b, nullptr, 0, 0, nullptr, 256, 0, nullptr, 0,
                           0, nullptr, 0, 0, nullptr, b->pzr, b->pgate, 384, 0, E, H, W, 128, 256, 3, 3, VIPE_ACT_NONE,
                           VIPE_CONV_PARTIAL, stream);
    if (rc != VIPE_OK) return rc;
  }
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_update_gate_state_piece(void* user, int piece, int n_pieces, void* stream) {
  const vipe_gate_state_job* j = (const vipe_gate_state_job*)user;
  VIPE_CHECK_ARG(j && j->weights && j->buffers && j->net && n_pieces > 0 && piece >= 0 && piece < n_pieces);
  const vipe_update_buffers* b = j->buffers;
  VIPE_CHECK_ARG(b->E >= 0 && b->H > 0 && b->W > 0 && b->pgate && b->pzr);
  int64_t e0 = (int64_t)b->E * piece / n_pieces, e1 = (int64_t)b->E * (piece + 1) / n_pieces;
  if (j->bounds && j->n_bounds == n_pieces + 1) {
    e0 = j->bounds[piece];
    e1 = j->bounds[piece + 1];
    VIPE_CHECK_ARG(j->bounds[0] == 0 && j->bounds[n_pieces] == b->E && e0 >= 0 && e0 <= e1 && e1 <= b->E);
  }
  if (e1 <= e0) return VIPE_OK;
  const int64_t px = (int64_t)b->H * b->W;
  vipe_update_buffers sub = *b;
  sub.E = (int)(e1 - e0);
  sub.pgate = (const char*)b->pgate + e0 * px * 384 * 2;
  sub.pzr = b->pzr + e0 * px * 256;
  return vipe_update_gate_state(j->weights, &sub, (const char*)j->net + e0 * px * 128 * 2, 2, stream);
}

VIPE_EXPORT int vipe_glo_context(const float* d_glo_sum, const float* d_wT, const float* d_bias, float* d_extra, int E, int hw,
                                 void* stream) {
  VIPE_CHECK_ARG(E >= 0 && hw > 0);
  if (E == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_glo_sum && d_wT && d_bias && d_extra);
  glo_gemm_kernel<<<E, 384, 0, as_stream(stream)>>>(d_glo_sum, d_wT, d_bias, d_extra, 1.0f / (float)hw);
  return vipe_launch_status();
}

VIPE_EXPORT int vipe_update_finish(const float* d_coords1, const float* d_dw, const unsigned char* d_mask, float* d_target,
                                   float* d_weight, const float* d_eta, const int64_t* d_du, float* d_damping, int E,
                                   int n_src, int ht, int wd, void* stream) {
  VIPE_CHECK_ARG(E >= 0 && n_src >= 0 && ht > 0 && wd > 0);
  if (E == 0) return VIPE_OK;
  VIPE_CHECK_ARG(d_coords1 && d_dw && d_target && d_weight && (n_src == 0 || (d_eta && d_du && d_damping)));
  const int P = ht * wd;
  const int64_t n_px = (int64_t)E * P, n_eta = (int64_t)n_src * P;
  const int64_t n = n_px > n_eta ? n_px : n_eta;
  update_finish_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(d_coords1, d_dw, d_mask, d_target, d_weight,
                                                                                  d_eta, d_du, d_damping, n_px, n_eta, P);
  return vipe_launch_status();
}

}  // extern "C"
