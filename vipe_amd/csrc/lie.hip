// lietorch_ext: batched Lie-group ops, one group element per lane, grid-stride (device) or a plain
// loop (host) over the SAME closed forms (lie_math.h).  Replaces csrc/lietorch_ext/lietorch_gpu.cu:24-299
// and lietorch_cpu.cpp of the reference for SO3 (group_id 1), RxSO3 (2), SE3 (3) and Sim3 (4), float32/float64.
// Forward AND backward passes, projector, Jinv.
#include "common.cuh"
#include "lie_math.h"

namespace {

using namespace lie;

// ---- per-element functors: in0 [n,A], in1 [n,B] (optional), out [n,C]; `bc` = rows per element of in0
template <typename G, typename S>
struct OpExp {  // SE3, RxSO3, Sim3: exp takes the tangent row
  static constexpr int A = G::K, B = 0, C = G::N;
  static LIE_HD void run(const S* a, const S*, S* o) { G::exp(a).store(o); }
};
template <typename S>
struct OpExp<SO3<S>, S> {
  static constexpr int A = 3, B = 0, C = 4;
  static LIE_HD void run(const S* a, const S*, S* o) { SO3<S>::exp(Vec3<S>{a[0], a[1], a[2]}).store(o); }
};

template <typename G, typename S>
struct OpLog {
  static constexpr int A = G::N, B = 0, C = G::K;
  static LIE_HD void run(const S* x, const S*, S* o) { G(x).log(o); }
};
template <typename S>
struct OpLog<SO3<S>, S> {
  static constexpr int A = 4, B = 0, C = 3;
  static LIE_HD void run(const S* x, const S*, S* o) {
    Vec3<S> v = SO3<S>(x).log();
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
  }
};

template <typename G, typename S>
struct OpInv {
  static constexpr int A = G::N, B = 0, C = G::N;
  static LIE_HD void run(const S* x, const S*, S* o) { G(x).inv().store(o); }
};
template <typename G, typename S>
struct OpMul {
  static constexpr int A = G::N, B = G::N, C = G::N;
  static LIE_HD void run(const S* x, const S* y, S* o) { (G(x) * G(y)).store(o); }
};

template <typename G, typename S>
struct OpAdj {  // RxSO3 / Sim3: dense K x K adjoint (rxso3.h:110, sim3.h:101)
  static constexpr int A = G::N, B = G::K, C = G::K;
  static LIE_HD void run(const S* x, const S* a, S* o) {
    S Ad[G::K][G::K];
    Jac<G, S>::Adj(G(x), Ad);
    for (int i = 0; i < G::K; ++i) {
      S acc = 0;
      for (int j = 0; j < G::K; ++j) acc += Ad[i][j] * a[j];
      o[i] = acc;
    }
  }
};
template <typename S>
struct OpAdj<SE3<S>, S> {
  static constexpr int A = 7, B = 6, C = 6;
  static LIE_HD void run(const S* x, const S* a, S* o) { SE3<S>(x).adj(a, o); }
};
template <typename S>
struct OpAdj<SO3<S>, S> {
  static constexpr int A = 4, B = 3, C = 3;
  static LIE_HD void run(const S* x, const S* a, S* o) {
    Vec3<S> v = SO3<S>(x).matrix() * Vec3<S>{a[0], a[1], a[2]};
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
  }
};
template <typename G, typename S>
struct OpAdjT {
  static constexpr int A = G::N, B = G::K, C = G::K;
  static LIE_HD void run(const S* x, const S* a, S* o) {
    S Ad[G::K][G::K];
    Jac<G, S>::Adj(G(x), Ad);
    for (int j = 0; j < G::K; ++j) {
      S acc = 0;
      for (int i = 0; i < G::K; ++i) acc += Ad[i][j] * a[i];
      o[j] = acc;
    }
  }
};
template <typename S>
struct OpAdjT<SE3<S>, S> {
  static constexpr int A = 7, B = 6, C = 6;
  static LIE_HD void run(const S* x, const S* a, S* o) { SE3<S>(x).adjT(a, o); }
};
template <typename S>
struct OpAdjT<SO3<S>, S> {
  static constexpr int A = 4, B = 3, C = 3;
  static LIE_HD void run(const S* x, const S* a, S* o) {
    Vec3<S> v = SO3<S>(x).matrix().tmul(Vec3<S>{a[0], a[1], a[2]});
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
  }
};

template <typename G, typename S>
struct OpAct {
  static constexpr int A = G::N, B = 3, C = 3;
  static LIE_HD void run(const S* x, const S* p, S* o) {
    Vec3<S> v = G(x).act(Vec3<S>{p[0], p[1], p[2]});
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
  }
};
template <typename G, typename S>
struct OpAct4 {
  static constexpr int A = G::N, B = 4, C = 4;
  static LIE_HD void run(const S* x, const S* p, S* o) { G(x).act4(p, o); }
};
template <typename S>
struct OpAct4<SO3<S>, S> {
  static constexpr int A = 4, B = 4, C = 4;
  static LIE_HD void run(const S* x, const S* p, S* o) {
    Vec3<S> v = SO3<S>(x).act(Vec3<S>{p[0], p[1], p[2]});
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = p[3];
  }
};

template <typename G, typename S>
struct OpMatrix {
  static constexpr int A = G::N, B = 0, C = 16;
  static LIE_HD void run(const S* x, const S*, S* o) {
    S T[4][4];
    Jac<G, S>::matrix4(G(x), T);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) o[4 * i + j] = T[i][j];
  }
};
template <typename S>
struct OpMatrix<SE3<S>, S> {
  static constexpr int A = 7, B = 0, C = 16;
  static LIE_HD void run(const S* x, const S*, S* o) {
    SE3<S> X(x);
    Mat3<S> R = X.r.matrix();
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[4 * i + j] = R.m[i][j]; }
    o[3] = X.t.x; o[7] = X.t.y; o[11] = X.t.z;
    o[12] = o[13] = o[14] = 0; o[15] = 1;
  }
};
template <typename S>
struct OpMatrix<SO3<S>, S> {
  static constexpr int A = 4, B = 0, C = 16;
  static LIE_HD void run(const S* x, const S*, S* o) {
    Mat3<S> R = SO3<S>(x).matrix();
    for (int i = 0; i < 16; ++i) o[i] = 0;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[4 * i + j] = R.m[i][j]; }
    o[15] = 1;
  }
};

// rows_per_elem > 1: in0 (the group element) is shared by `rows_per_elem` consecutive rows of in1/out
template <typename Op, typename S>
__global__ __launch_bounds__(256) void lie_kernel(const S* __restrict__ in0, const S* __restrict__ in1,
                                                   S* __restrict__ out, int64_t n, int64_t rows_per_elem) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    S a[Op::A], b[Op::B > 0 ? Op::B : 1], o[Op::C];
    const int64_t e = rows_per_elem > 1 ? i / rows_per_elem : i;
#pragma unroll
    for (int k = 0; k < Op::A; ++k) a[k] = in0[e * Op::A + k];
#pragma unroll
    for (int k = 0; k < Op::B; ++k) b[k] = in1[i * Op::B + k];
    Op::run(a, b, o);
#pragma unroll
    for (int k = 0; k < Op::C; ++k) out[i * Op::C + k] = o[k];
  }
}

template <typename Op, typename S>
int run_op(const void* in0, const void* in1, void* out, int64_t n, int64_t rpe, int on_device, hipStream_t s) {
  if (n == 0) return VIPE_OK;
  if (!in0 || !out || (Op::B > 0 && !in1)) return VIPE_EINVAL;
  if (on_device) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    lie_kernel<Op, S><<<(int)blocks, 256, 0, s>>>((const S*)in0, (const S*)in1, (S*)out, n, rpe);
    return vipe_launch_status();
  }
  const S* a = (const S*)in0;
  const S* b = (const S*)in1;
  S* o = (S*)out;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t e = rpe > 1 ? i / rpe : i;
    Op::run(a + e * Op::A, Op::B > 0 ? b + i * Op::B : nullptr, o + i * Op::C);
  }
  return VIPE_OK;
}

template <template <typename, typename> class Op>
int dispatch(int gid, const void* in0, const void* in1, void* out, int64_t n, int64_t rpe, int dtype, int on_device,
             void* stream) {
  hipStream_t s = as_stream(stream);
  if (n < 0) return VIPE_EINVAL;
  if (dtype == VIPE_F32) {
    if (gid == 3) return run_op<Op<SE3<float>, float>, float>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 1) return run_op<Op<SO3<float>, float>, float>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 4) return run_op<Op<Sim3<float>, float>, float>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 2) return run_op<Op<RxSO3<float>, float>, float>(in0, in1, out, n, rpe, on_device, s);
  } else if (dtype == VIPE_F64) {
    if (gid == 3) return run_op<Op<SE3<double>, double>, double>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 1) return run_op<Op<SO3<double>, double>, double>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 4) return run_op<Op<Sim3<double>, double>, double>(in0, in1, out, n, rpe, on_device, s);
    if (gid == 2) return run_op<Op<RxSO3<double>, double>, double>(in0, in1, out, n, rpe, on_device, s);
  } else {
    return VIPE_EINVAL;
  }
  return VIPE_EINVAL;
}

}  // namespace

VIPE_EXPORT int vipe_lie_expm(int g, const void* a, void* X, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpExp>(g, a, nullptr, X, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_logm(int g, const void* X, void* a, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpLog>(g, X, nullptr, a, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_inv(int g, const void* X, void* Y, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpInv>(g, X, nullptr, Y, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_mul(int g, const void* X, const void* Y, void* Z, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpMul>(g, X, Y, Z, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_adj(int g, const void* X, const void* a, void* b, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpAdj>(g, X, a, b, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_adjT(int g, const void* X, const void* a, void* b, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpAdjT>(g, X, a, b, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_act(int g, const void* X, const void* p, void* q, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpAct>(g, X, p, q, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_act4(int g, const void* X, const void* p, void* q, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpAct4>(g, X, p, q, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_as_matrix(int g, const void* X, void* T, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpMatrix>(g, X, nullptr, T, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_adjT_bcast(int g, const void* X, const void* a, void* b, int64_t n_elem, int64_t rpe, int dt,
                                    void* st) {
  if (rpe < 1) return VIPE_EINVAL;
  return dispatch<OpAdjT>(g, X, a, b, n_elem * rpe, rpe, dt, 1, st);
}
VIPE_EXPORT int vipe_lie_act4_bcast(int g, const void* X, const void* p, void* q, int64_t n_elem, int64_t rpe, int dt,
                                    void* st) {
  if (rpe < 1) return VIPE_EINVAL;
  return dispatch<OpAct4>(g, X, p, q, n_elem * rpe, rpe, dt, 1, st);
}

// ---- backward passes (training / optimisation utilities; the SLAM system itself runs under torch.no_grad)

namespace {

// ---- backward passes, projector, Jinv (lietorch_gpu.cu:36-275): per-element functors with up to three inputs
// (grad, in0, in1) and two outputs.  Gradients of group elements live in the first K of N columns (the rest stays 0).
template <typename G, typename S, int K>
LIE_HD void rowvec_mat(const S* v, const S (&M)[K][K], S* o, S sign = S(1)) {
  for (int j = 0; j < K; ++j) {
    S acc = 0;
    for (int i = 0; i < K; ++i) acc += v[i] * M[i][j];
    o[j] = sign * acc;
  }
}

template <typename G, typename S>
struct BwExp {  // da = dX Jl(a)
  static constexpr int GW = G::N, A = G::K, B = 0, C0 = G::K, C1 = 0;
  static LIE_HD void run(const S* g, const S* a, const S*, S* o0, S*) {
    S J[G::K][G::K];
    Jac<G, S>::left_jacobian(a, J);
    rowvec_mat<G, S, G::K>(g, J, o0);
  }
};
template <typename G, typename S>
struct BwLog {  // dX = da Jl^-1(log X)
  static constexpr int GW = G::K, A = G::N, B = 0, C0 = G::N, C1 = 0;
  static LIE_HD void run(const S* g, const S* x, const S*, S* o0, S*) {
    S a[G::K], J[G::K][G::K];
    Jac<G, S>::log(G(x), a);
    Jac<G, S>::left_jacobian_inverse(a, J);
    for (int k = G::K; k < G::N; ++k) o0[k] = 0;
    rowvec_mat<G, S, G::K>(g, J, o0);
  }
};
template <typename G, typename S>
struct BwInv {  // dX = -dY Adj(X^-1)
  static constexpr int GW = G::N, A = G::N, B = 0, C0 = G::N, C1 = 0;
  static LIE_HD void run(const S* g, const S* x, const S*, S* o0, S*) {
    S Ad[G::K][G::K];
    Jac<G, S>::Adj(G(x).inv(), Ad);
    for (int k = G::K; k < G::N; ++k) o0[k] = 0;
    rowvec_mat<G, S, G::K>(g, Ad, o0, S(-1));
  }
};
template <typename G, typename S>
struct BwMul {  // dX = dZ, dY = dZ Adj(X)
  static constexpr int GW = G::N, A = G::N, B = G::N, C0 = G::N, C1 = G::N;
  static LIE_HD void run(const S* g, const S* x, const S*, S* o0, S* o1) {
    S Ad[G::K][G::K];
    Jac<G, S>::Adj(G(x), Ad);
    for (int k = 0; k < G::N; ++k) { o0[k] = k < G::K ? g[k] : S(0); o1[k] = 0; }
    rowvec_mat<G, S, G::K>(g, Ad, o1);
  }
};
template <typename G, typename S>
struct BwAdj {  // b = Adj(X) a: da = db Adj(X), dX = -db adj(b)
  static constexpr int GW = G::K, A = G::N, B = G::K, C0 = G::N, C1 = G::K;
  static LIE_HD void run(const S* g, const S* x, const S* a, S* o0, S* o1) {
    S Ad[G::K][G::K], ad[G::K][G::K], b[G::K];
    Jac<G, S>::Adj(G(x), Ad);
    for (int i = 0; i < G::K; ++i) {
      S acc = 0;
      for (int j = 0; j < G::K; ++j) acc += Ad[i][j] * a[j];
      b[i] = acc;
    }
    Jac<G, S>::adj(b, ad);
    rowvec_mat<G, S, G::K>(g, Ad, o1);
    for (int k = G::K; k < G::N; ++k) o0[k] = 0;
    rowvec_mat<G, S, G::K>(g, ad, o0, S(-1));
  }
};
template <typename G, typename S>
struct BwAdjT {  // b = Adj(X)^T a: da = Adj(X) db, dX = -a^T adj(Adj(X) db)
  static constexpr int GW = G::K, A = G::N, B = G::K, C0 = G::N, C1 = G::K;
  static LIE_HD void run(const S* g, const S* x, const S* a, S* o0, S* o1) {
    S Ad[G::K][G::K], ad[G::K][G::K];
    Jac<G, S>::Adj(G(x), Ad);
    for (int i = 0; i < G::K; ++i) {
      S acc = 0;
      for (int j = 0; j < G::K; ++j) acc += Ad[i][j] * g[j];
      o1[i] = acc;
    }
    Jac<G, S>::adj(o1, ad);
    for (int k = G::K; k < G::N; ++k) o0[k] = 0;
    rowvec_mat<G, S, G::K>(a, ad, o0, S(-1));
  }
};
template <typename G, typename S>
struct BwAct {  // q = X p: dp = dq R, dX = dq act_jacobian(q)
  static constexpr int GW = 3, A = G::N, B = 3, C0 = G::N, C1 = 3;
  static LIE_HD void run(const S* g, const S* x, const S* p, S* o0, S* o1) {
    G X(x);
    S T[4][4], J[3][G::K];
    Jac<G, S>::matrix4(X, T);
    for (int j = 0; j < 3; ++j) o1[j] = g[0] * T[0][j] + g[1] * T[1][j] + g[2] * T[2][j];
    Jac<G, S>::act_jacobian(X.act(Vec3<S>{p[0], p[1], p[2]}), J);
    for (int k = 0; k < G::N; ++k) o0[k] = 0;
    for (int j = 0; j < G::K; ++j) o0[j] = g[0] * J[0][j] + g[1] * J[1][j] + g[2] * J[2][j];
  }
};
template <typename G, typename S>
struct BwAct4 {  // q = X.act4(p): dp = dq T, dX = dq act4_jacobian(q)
  static constexpr int GW = 4, A = G::N, B = 4, C0 = G::N, C1 = 4;
  static LIE_HD void run(const S* g, const S* x, const S* p, S* o0, S* o1) {
    G X(x);
    S T[4][4], J[4][G::K], q[4];
    Jac<G, S>::matrix4(X, T);
    for (int j = 0; j < 4; ++j) o1[j] = g[0] * T[0][j] + g[1] * T[1][j] + g[2] * T[2][j] + g[3] * T[3][j];
    OpAct4<G, S>::run(x, p, q);
    Jac<G, S>::act4_jacobian(q, J);
    for (int k = 0; k < G::N; ++k) o0[k] = 0;
    for (int j = 0; j < G::K; ++j) o0[j] = g[0] * J[0][j] + g[1] * J[1][j] + g[2] * J[2][j] + g[3] * J[3][j];
  }
};
template <typename G, typename S>
struct OpProjector {
  static constexpr int A = G::N, B = 0, C = G::N * G::N;
  static LIE_HD void run(const S* x, const S*, S* o) {
    S P[G::N][G::N];
    Jac<G, S>::projector(G(x), P);
    for (int i = 0; i < G::N; ++i)
      for (int j = 0; j < G::N; ++j) o[i * G::N + j] = P[i][j];
  }
};
template <typename G, typename S>
struct OpJinv {  // b = Jl^-1(log X) a
  static constexpr int A = G::N, B = G::K, C = G::K;
  static LIE_HD void run(const S* x, const S* a, S* o) {
    S l[G::K], J[G::K][G::K];
    Jac<G, S>::log(G(x), l);
    Jac<G, S>::left_jacobian_inverse(l, J);
    for (int i = 0; i < G::K; ++i) {
      S acc = 0;
      for (int j = 0; j < G::K; ++j) acc += J[i][j] * a[j];
      o[i] = acc;
    }
  }
};

template <typename Op, typename S>
__global__ __launch_bounds__(256) void lie_bwd_kernel(const S* __restrict__ grad, const S* __restrict__ in0,
                                                       const S* __restrict__ in1, S* __restrict__ out0,
                                                       S* __restrict__ out1, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    S g[Op::GW], a[Op::A], b[Op::B > 0 ? Op::B : 1], o0[Op::C0], o1[Op::C1 > 0 ? Op::C1 : 1];
#pragma unroll
    for (int k = 0; k < Op::GW; ++k) g[k] = grad[i * Op::GW + k];
#pragma unroll
    for (int k = 0; k < Op::A; ++k) a[k] = in0[i * Op::A + k];
#pragma unroll
    for (int k = 0; k < Op::B; ++k) b[k] = in1[i * Op::B + k];
    Op::run(g, a, b, o0, o1);
#pragma unroll
    for (int k = 0; k < Op::C0; ++k) out0[i * Op::C0 + k] = o0[k];
#pragma unroll
    for (int k = 0; k < Op::C1; ++k) out1[i * Op::C1 + k] = o1[k];
  }
}

template <typename Op, typename S>
int run_bwd(const void* grad, const void* in0, const void* in1, void* out0, void* out1, int64_t n, int on_device,
            hipStream_t s) {
  if (n == 0) return VIPE_OK;
  if (!grad || !in0 || !out0 || (Op::B > 0 && !in1) || (Op::C1 > 0 && !out1)) return VIPE_EINVAL;
  if (on_device) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    lie_bwd_kernel<Op, S><<<(int)blocks, 256, 0, s>>>((const S*)grad, (const S*)in0, (const S*)in1, (S*)out0, (S*)out1, n);
    return vipe_launch_status();
  }
  const S *g = (const S*)grad, *a = (const S*)in0, *b = (const S*)in1;
  S *o0 = (S*)out0, *o1 = (S*)out1, dummy[8];
  for (int64_t i = 0; i < n; ++i)
    Op::run(g + i * Op::GW, a + i * Op::A, Op::B > 0 ? b + i * Op::B : nullptr, o0 + i * Op::C0,
            Op::C1 > 0 ? o1 + i * Op::C1 : dummy);
  return VIPE_OK;
}

template <template <typename, typename> class Op>
int dispatch_bwd(int gid, const void* grad, const void* in0, const void* in1, void* out0, void* out1, int64_t n,
                 int dtype, int on_device, void* stream) {
  hipStream_t s = as_stream(stream);
  if (n < 0) return VIPE_EINVAL;
  if (dtype == VIPE_F32) {
    if (gid == 3) return run_bwd<Op<SE3<float>, float>, float>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 1) return run_bwd<Op<SO3<float>, float>, float>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 4) return run_bwd<Op<Sim3<float>, float>, float>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 2) return run_bwd<Op<RxSO3<float>, float>, float>(grad, in0, in1, out0, out1, n, on_device, s);
  } else if (dtype == VIPE_F64) {
    if (gid == 3) return run_bwd<Op<SE3<double>, double>, double>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 1) return run_bwd<Op<SO3<double>, double>, double>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 4) return run_bwd<Op<Sim3<double>, double>, double>(grad, in0, in1, out0, out1, n, on_device, s);
    if (gid == 2) return run_bwd<Op<RxSO3<double>, double>, double>(grad, in0, in1, out0, out1, n, on_device, s);
  } else {
    return VIPE_EINVAL;
  }
  return VIPE_EINVAL;
}

}  // namespace

VIPE_EXPORT int vipe_lie_projector(int g, const void* X, void* P, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpProjector>(g, X, nullptr, P, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_jinv(int g, const void* X, const void* a, void* b, int64_t n, int dt, int dev, void* st) {
  return dispatch<OpJinv>(g, X, a, b, n, 1, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_expm_backward(int g, const void* grad, const void* a, void* da, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwExp>(g, grad, a, nullptr, da, nullptr, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_logm_backward(int g, const void* grad, const void* X, void* dX, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwLog>(g, grad, X, nullptr, dX, nullptr, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_inv_backward(int g, const void* grad, const void* X, void* dX, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwInv>(g, grad, X, nullptr, dX, nullptr, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_mul_backward(int g, const void* grad, const void* X, const void* Y, void* dX, void* dY, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwMul>(g, grad, X, Y, dX, dY, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_adj_backward(int g, const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwAdj>(g, grad, X, a, dX, da, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_adjT_backward(int g, const void* grad, const void* X, const void* a, void* dX, void* da, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwAdjT>(g, grad, X, a, dX, da, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_act_backward(int g, const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwAct>(g, grad, X, p, dX, dp, n, dt, dev, st);
}
VIPE_EXPORT int vipe_lie_act4_backward(int g, const void* grad, const void* X, const void* p, void* dX, void* dp, int64_t n, int dt, int dev, void* st) {
  return dispatch_bwd<BwAct4>(g, grad, X, p, dX, dp, n, dt, dev, st);
}

// ---- [fused] what the keyframe frontend does to the NEXT frame's slot after every keyframe (frontend.py:70-76, 118-122,
// 147-151): constant-velocity pose  poses[t1] = Exp(0.5 Log(G_{t1-1} G_{t1-2}^-1)) G_{t1-1}  (`init_pose`; skipped when the
// caller supplies poses) and  disps[t1, v] = mean(disps[t1-n_mean .. t1-1, v])  (n_mean = 1 per keyframe, 4 after the
// initialisation).  The reference issues inv, mul, log, scale, exp, mul on the poses and mean + fill per view: 9+ launches
// of a few microseconds each; here one.  Same group formulas as vipe_lie_* (lie_math.h).  grid = n_views blocks.
namespace {
__global__ __launch_bounds__(256) void frontend_next_frame_kernel(float* __restrict__ poses, float* __restrict__ disps, int t1,
                                                                  int V, int P, int n_mean, int init_pose) {
  const int v = blockIdx.x;
  if (v == 0 && threadIdx.x == 0 && init_pose) {
    const SE3<float> p1(poses + 7 * (t1 - 2)), p2(poses + 7 * (t1 - 1));
    float w[6];
    (p2 * p1.inv()).log(w);
    for (int i = 0; i < 6; ++i) w[i] *= 0.5f;
    (SE3<float>::exp(w) * p2).store(poses + 7 * t1);
  }
  __shared__ float red[4];
  float s = 0.0f;
  for (int m = 1; m <= n_mean; ++m) {
    const float* d = disps + ((int64_t)(t1 - m) * V + v) * P;
    for (int k = threadIdx.x; k < P; k += 256) s += d[k];
  }
  s = wave_sum(s);
  if (lane_id() == 0) red[wave_id()] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)((int64_t)P * n_mean);
  float* o = disps + ((int64_t)t1 * V + v) * P;
  for (int k = threadIdx.x; k < P; k += 256) o[k] = mean;
}
}  // namespace

VIPE_EXPORT int vipe_frontend_next_frame(float* d_poses, float* d_disps, int t1, int n_views, int P, int n_mean, int init_pose,
                                         void* stream) {
  VIPE_CHECK_ARG(d_poses && d_disps && n_views > 0 && P > 0 && n_mean >= 1 && t1 >= n_mean && (!init_pose || t1 >= 2));
  frontend_next_frame_kernel<<<n_views, 256, 0, as_stream(stream)>>>(d_poses, d_disps, t1, n_views, P, n_mean, init_pose);
  return vipe_launch_status();
}
