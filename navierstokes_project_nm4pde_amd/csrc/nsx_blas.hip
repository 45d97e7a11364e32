// nsx_blas.hip — fused BLAS-1 for the Krylov drivers (the Epetra_Vector operations behind deal.II's
// SolverGMRES / SolverCG and the sadd/add/scale calls of reference Preconditioners.hpp:176,195,202-203,281,294-309,386,406,492-515).
//
// Scalars never visit the host inside an orthogonalisation sweep: every reduction leaves its result in the device
// array h->scal[slot]; consumers take their coefficients as SRef = c * scal[num] / scal[den] read on the device.
// Reductions are deterministic (fixed grid, fixed-order partial sums; no atomics) so that runs are bitwise repeatable.
// Multi-GPU: the finalise step is followed by an RCCL all-reduce of the slot (comm_allreduce_scalars).
#include "nsx_internal.hpp"

namespace nsx {

struct SRef {
  double c;
  int num, den;
};
__device__ __forceinline__ double sval(const double *__restrict__ scal, SRef r) {
  double v = r.c;
  if (r.num >= 0) v *= scal[r.num];
  if (r.den >= 0) v /= scal[r.den];
  return v;
}

constexpr int RED_BLOCKS = 1024;
constexpr int RED_STRIDE = 1024;  // partial slots per reduction

__device__ __forceinline__ double block_sum_256(double v, double *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[k];
  }
  return t;  // valid in thread 0
}

// d (+)= ... ; partial[b] = sum_i d_i * w_i over the block's fixed slice
enum { OP_DOT = 0, OP_ADD_AND_DOT = 1 };
template <int OP>
__global__ __launch_bounds__(256) void k_reduce(int n, double *__restrict__ d, SRef a, const double *__restrict__ v,
                                                const double *__restrict__ w, const double *__restrict__ scal,
                                                double *__restrict__ partial) {
  __shared__ double sh[4];
  double acc = 0.0;
  const double alpha = OP == OP_ADD_AND_DOT ? sval(scal, a) : 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double di = d[i];
    if (OP == OP_ADD_AND_DOT) {
      di += alpha * v[i];
      d[i] = di;
    }
    acc += di * (w == d ? di : w[i]);
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void k_finalize(int nb, const double *__restrict__ partial, double *__restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) *out = t;
}

static int red_blocks(int n) { return std::max(1, std::min(RED_BLOCKS, cdiv(n, 2048))); }

static void finalize(nsx_handle *h, int nb, int slot) {
  if (nb > 1) hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, h->stream, nb, h->red_partial.p + (size_t)slot * RED_STRIDE, h->scal.p + slot);
  comm_allreduce_scalars(h, slot, 1);
}

void v_dot(nsx_handle *h, int n, const double *a, const double *b, int slot) {
  LaunchScope ls(h, "dot", 16.0 * n);
  const int nb = red_blocks(n);
  double *out = nb > 1 ? h->red_partial.p + (size_t)slot * RED_STRIDE : h->scal.p + slot;
  hipLaunchKernelGGL((k_reduce<OP_DOT>), dim3(nb), dim3(256), 0, h->stream, n, const_cast<double *>(a), SRef{0, -1, -1}, nullptr, b,
                     h->scal.p, out);
  finalize(h, nb, slot);
}

void v_add_and_dot(nsx_handle *h, int n, double *d, double a, int aslot, const double *v, const double *w, int slot) {
  LaunchScope ls(h, "add_and_dot", (w == d ? 24.0 : 32.0) * n);
  const int nb = red_blocks(n);
  double *out = nb > 1 ? h->red_partial.p + (size_t)slot * RED_STRIDE : h->scal.p + slot;
  hipLaunchKernelGGL((k_reduce<OP_ADD_AND_DOT>), dim3(nb), dim3(256), 0, h->stream, n, d, SRef{a, aslot, -1}, v, w, h->scal.p, out);
  finalize(h, nb, slot);
}

// ---- element-wise
__global__ void k_axpby(int n, double *__restrict__ d, SRef s, SRef a, const double *__restrict__ v, const double *__restrict__ scal, int mode) {
  // mode 0: d = s d + a v ; mode 1: d = a v ; mode 2: d = s d
  const double sv = mode == 1 ? 0.0 : sval(scal, s), av = mode == 2 ? 0.0 : sval(scal, a);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    if (mode == 0) d[i] = sv * d[i] + av * v[i];
    else if (mode == 1) d[i] = av * v[i];
    else d[i] = sv * d[i];
  }
}
__global__ void k_scale_vec(int n, double *__restrict__ d, const double *__restrict__ f) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] *= f[i];
}
struct MultiArgs {
  const double *v[32];
  double c[32];
  int k;
};
__global__ void k_axpy_multi(int n, double *__restrict__ x, MultiArgs m) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double s = x[i];
    for (int j = 0; j < m.k; ++j) s += m.c[j] * m.v[j][i];  // same order as the reference's x.add(h(i), tmp_vectors[i]) loop
    x[i] = s;
  }
}
// CG update (SolverCG): x += alpha d ; g += alpha h ; partial(g.g), alpha = scal[gh] / scal[dh]
__global__ __launch_bounds__(256) void k_cg_update(int n, double *__restrict__ x, const double *__restrict__ dvec, double *__restrict__ g,
                                                   const double *__restrict__ hvec, SRef a, const double *__restrict__ scal,
                                                   double *__restrict__ partial) {
  __shared__ double sh[4];
  const double alpha = sval(scal, a);
  double acc = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    x[i] += alpha * dvec[i];
    const double gi = g[i] + alpha * hvec[i];
    g[i] = gi;
    acc += gi * gi;
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

static int ew_blocks(int n) { return std::max(1, std::min(2048, cdiv(n, 256))); }

static void axpby(nsx_handle *h, int n, double *d, SRef s, SRef a, const double *v, int mode, double bytes_per) {
  LaunchScope ls(h, "axpby", bytes_per * n);
  hipLaunchKernelGGL(k_axpby, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, d, s, a, v, h->scal.p, mode);
}

void v_copy(nsx_handle *h, int n, double *d, const double *s) {
  if (d != s && n) HIP_CHECK(hipMemcpyAsync(d, s, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
}
void v_zero(nsx_handle *h, int n, double *d) {
  if (n) HIP_CHECK(hipMemsetAsync(d, 0, (size_t)n * sizeof(double), h->stream));
}
void v_add(nsx_handle *h, int n, double *d, double a, const double *v) { axpby(h, n, d, SRef{1, -1, -1}, SRef{a, -1, -1}, v, 0, 24); }
void v_add_dev(nsx_handle *h, int n, double *d, double a, int slot, const double *v) {
  axpby(h, n, d, SRef{1, -1, -1}, SRef{a, slot, -1}, v, 0, 24);
}
void v_sadd(nsx_handle *h, int n, double *d, double s, double a, const double *v) { axpby(h, n, d, SRef{s, -1, -1}, SRef{a, -1, -1}, v, 0, 24); }
void v_scale(nsx_handle *h, int n, double *d, double a) { axpby(h, n, d, SRef{a, -1, -1}, SRef{0, -1, -1}, nullptr, 2, 16); }
void v_scale_dev_inv(nsx_handle *h, int n, double *d, int slot) { axpby(h, n, d, SRef{1, -1, slot}, SRef{0, -1, -1}, nullptr, 2, 16); }
void v_scale_vec(nsx_handle *h, int n, double *d, const double *f) {
  LaunchScope ls(h, "scale_vec", 24.0 * n);
  hipLaunchKernelGGL(k_scale_vec, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, d, f);
}
void v_axpy_multi(nsx_handle *h, int n, double *x, int k, double *const *vs, const double *coef) {
  for (int j0 = 0; j0 < k; j0 += 32) {
    MultiArgs m;
    m.k = std::min(32, k - j0);
    for (int j = 0; j < m.k; ++j) {
      m.v[j] = vs[j0 + j];
      m.c[j] = coef[j0 + j];
    }
    LaunchScope ls(h, "axpy_multi", 8.0 * n * (2 + m.k));
    hipLaunchKernelGGL(k_axpy_multi, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, x, m);
  }
}

// SolverCG helpers
void cg_update(nsx_handle *h, int n, double *x, const double *d, double *g, const double *hv, int gh_slot, int dh_slot, int res_slot) {
  LaunchScope ls(h, "cg_update", 48.0 * n);
  const int nb = red_blocks(n);
  double *out = nb > 1 ? h->red_partial.p + (size_t)res_slot * RED_STRIDE : h->scal.p + res_slot;
  hipLaunchKernelGGL(k_cg_update, dim3(nb), dim3(256), 0, h->stream, n, x, d, g, hv, SRef{1, gh_slot, dh_slot}, h->scal.p, out);
  finalize(h, nb, res_slot);
}
// d = (scal[num]/scal[den]) d - h
void cg_direction(nsx_handle *h, int n, double *d, const double *hv, int num_slot, int den_slot) {
  axpby(h, n, d, SRef{1, num_slot, den_slot}, SRef{-1, -1, -1}, hv, 0, 24);
}

double read_scalar(nsx_handle *h, int slot) {
  double v;
  read_scalars(h, slot, 1, &v);
  return v;
}
void read_scalars(nsx_handle *h, int slot0, int count, double *out) {
  HIP_CHECK(hipMemcpyAsync(h->scal_host + slot0, h->scal.p + slot0, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < count; ++i) out[i] = h->scal_host[slot0 + i];
}
void write_scalar(nsx_handle *h, int slot, double v) {
  h->scal_host[slot] = v;
  HIP_CHECK(hipMemcpyAsync(h->scal.p + slot, h->scal_host + slot, sizeof(double), hipMemcpyHostToDevice, h->stream));
}

}  // namespace nsx
