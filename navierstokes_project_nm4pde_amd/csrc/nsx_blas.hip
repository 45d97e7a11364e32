// nsx_blas.hip — fused BLAS-1 for the Krylov drivers (the Epetra_Vector operations behind deal.II's
// SolverGMRES / SolverCG and the sadd/add/scale calls of reference Preconditioners.hpp:176,195,202-203,281,294-309,386,406,492-515).
//
// Scalars never visit the host inside an orthogonalisation sweep.  A reduction leaves per-block partial sums in
// h->red_partial[slot][0..nb); the CONSUMER kernel (the next add_and_dot / axpy / CG update) sums them itself in a fixed
// order while it starts up, so a dot product costs one launch, not two (a separate 1-block finalise kernel measured
// 4.6 us x 13 000 launches per step, profiles/r01).  Coefficients are SRef = c * value(num) / value(den) evaluated on the
// device.  Everything is deterministic: fixed grids, fixed-order sums, no atomics.
// Multi-GPU: a reduction is finalised at once and all-reduced over RCCL (comm_allreduce_scalars), consumers then read scal[].
#include "nsx_internal.hpp"

namespace nsx {

constexpr int RED_BLOCKS = 512;
constexpr int RED_STRIDE = 1024;  // partial slots reserved per scalar

struct SRef {
  double c;
  int num, den;        // slots, -1 = none
  int num_nb, den_nb;  // number of valid partials (0 = scal[slot] is final)
};

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// value of one slot, cooperatively by the block (blockDim.x >= 64); sh: one double of LDS
__device__ __forceinline__ double slot_value(const double *__restrict__ scal, const double *__restrict__ partial, int slot, int nb,
                                             double *sh) {
  if (nb == 0) return scal[slot];
  if (threadIdx.x < 64) {
    // nb <= RED_BLOCKS = 512: eight independent loads per lane, one L2 round trip
    const double *p = partial + (size_t)slot * RED_STRIDE;
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = threadIdx.x + 64 * k;
      t[k] = i < nb ? p[i] : 0.0;
    }
    double a = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    a = wave_sum_all(a);
    if (threadIdx.x == 0) *sh = a;
  }
  __syncthreads();
  const double v = *sh;
  __syncthreads();
  return v;
}
__device__ __forceinline__ double sval(const double *__restrict__ scal, const double *__restrict__ partial, SRef r, double *sh) {
  double v = r.c;
  if (r.num >= 0) v *= slot_value(scal, partial, r.num, r.num_nb, sh);
  if (r.den >= 0) v /= slot_value(scal, partial, r.den, r.den_nb, sh);
  return v;
}

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

// d (+)= alpha v ; partial[b] = sum_i d_i * w_i over the block's fixed slice
enum { OP_DOT = 0, OP_ADD_AND_DOT = 1 };
template <int OP>
__global__ __launch_bounds__(256) void k_reduce(int n, int split, int gap, double *__restrict__ d, SRef a, const double *__restrict__ v,
                                                const double *__restrict__ w, const double *__restrict__ scal,
                                                const double *__restrict__ partial_in, double *__restrict__ partial) {
  __shared__ double sh[5];
  constexpr int U = 4;
  const bool self = (w == d);
  const int stride = gridDim.x * 256;
  // first batch of loads is issued BEFORE the coefficient is known: the partial-sum prologue (an L2 round trip, a wave
  // reduction and two barriers) then overlaps with the HBM latency of the stream instead of preceding it
  double dv[U], vv[U], wv[U];
  int idx[U];
#pragma unroll
  for (int k = 0; k < U; ++k) {
    const int i0 = blockIdx.x * 256 + threadIdx.x + k * stride;
    const bool ok = i0 < n;
    const int i = ok ? i0 + (i0 >= split ? gap : 0) : 0;
    idx[k] = ok ? i : -1;
    dv[k] = ok ? d[i] : 0.0;
    vv[k] = (ok && OP == OP_ADD_AND_DOT) ? v[i] : 0.0;
    wv[k] = (ok && !self) ? w[i] : 0.0;
  }
  const double alpha = OP == OP_ADD_AND_DOT ? sval(scal, partial_in, a, sh + 4) : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < U; ++k) {
    double di = dv[k];
    if (OP == OP_ADD_AND_DOT) {
      di += alpha * vv[k];
      if (idx[k] >= 0) d[idx[k]] = di;
    }
    acc += di * (self ? di : wv[k]);
  }
#pragma unroll 4
  for (int i0 = blockIdx.x * 256 + threadIdx.x + U * stride; i0 < n; i0 += stride) {
    const int i = i0 + (i0 >= split ? gap : 0);
    double di = d[i];
    if (OP == OP_ADD_AND_DOT) {
      di += alpha * v[i];
      d[i] = di;
    }
    acc += di * (self ? di : w[i]);
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

struct NbArgs {
  int nb[64];
};
__global__ __launch_bounds__(256) void k_finalize(int slot0, NbArgs nbs, const double *__restrict__ partial, double *__restrict__ scal) {
  __shared__ double sh;
  const int slot = slot0 + blockIdx.x;
  const int nb = nbs.nb[blockIdx.x];
  if (nb == 0) return;
  const double v = slot_value(scal, partial, slot, nb, &sh);
  if (threadIdx.x == 0) scal[slot] = v;
}

// finalise a range of slots AND publish them to mapped host memory; the block that finishes last raises the flag
__global__ __launch_bounds__(256) void k_publish(int slot0, int count, NbArgs nbs, const double *__restrict__ partial,
                                                 double *__restrict__ scal, double *pub_vals, unsigned long long *pub_flag,
                                                 unsigned long long seq, unsigned int *counter) {
  __shared__ double sh;
  const int slot = slot0 + blockIdx.x;
  const int nb = nbs.nb[blockIdx.x];
  const double v = slot_value(scal, partial, slot, nb, &sh);
  if (threadIdx.x == 0) {
    if (nb) scal[slot] = v;
    __hip_atomic_store(pub_vals + slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    const unsigned int ticket = atomicAdd(counter, 1u);
    if (ticket == (unsigned int)count - 1) {
      *counter = 0;
      __threadfence_system();
      __hip_atomic_store(pub_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

static int red_blocks(int n) { return std::max(1, std::min(RED_BLOCKS, cdiv(n, 1024))); }

static SRef sref(nsx_handle *h, double c, int num, int den) {
  return SRef{c, num, den, num >= 0 ? h->slot_nb[num] : 0, den >= 0 ? h->slot_nb[den] : 0};
}

// make scal[slot0 .. slot0+count) final (one launch for the whole range)
void finalize_slots(nsx_handle *h, int slot0, int count) {
  bool any = false;
  for (int i = 0; i < count; ++i) any = any || h->slot_nb[slot0 + i] > 0;
  if (!any) return;
  NbArgs args;
  if (count > 64) NSX_THROW(NSX_ERR_ARG, "internal: finalize_slots range too long");
  for (int i = 0; i < count; ++i) args.nb[i] = h->slot_nb[slot0 + i];
  hipLaunchKernelGGL(k_finalize, dim3(count), dim3(256), 0, h->stream, slot0, args, h->red_partial.p, h->scal.p);
  for (int i = 0; i < count; ++i) h->slot_nb[slot0 + i] = 0;
}

static void after_reduction(nsx_handle *h, int slot, int nb) {
  h->slot_nb[slot] = nb > 1 ? nb : 0;
  if (h->comm) {  // global sum needed before anybody consumes the value
    finalize_slots(h, slot, 1);
    comm_allreduce_scalars(h, slot, 1);
  }
}
static double *red_out(nsx_handle *h, int slot, int nb) {
  return nb > 1 ? h->red_partial.p + (size_t)slot * RED_STRIDE : h->scal.p + slot;
}

void v_dot(nsx_handle *h, Span sp, const double *a, const double *b, int slot) {
  const int n = sp.n;
  LaunchScope ls(h, "dot", (a == b ? 8.0 : 16.0) * n);
  const int nb = red_blocks(n);
  hipLaunchKernelGGL((k_reduce<OP_DOT>), dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, const_cast<double *>(a), SRef{0, -1, -1, 0, 0}, nullptr, b,
                     h->scal.p, h->red_partial.p, red_out(h, slot, nb));
  after_reduction(h, slot, nb);
}

void v_add_and_dot(nsx_handle *h, Span sp, double *d, double a, int aslot, const double *v, const double *w, int slot) {
  const int n = sp.n;
  LaunchScope ls(h, "add_and_dot", (w == d ? 24.0 : 32.0) * n);
  const int nb = red_blocks(n);
  hipLaunchKernelGGL((k_reduce<OP_ADD_AND_DOT>), dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, d, sref(h, a, aslot, -1), v, w, h->scal.p,
                     h->red_partial.p, red_out(h, slot, nb));
  after_reduction(h, slot, nb);
}

// ---- element-wise
__global__ __launch_bounds__(256) void k_axpby(int n, int split, int gap, double *__restrict__ d, SRef s, SRef a, const double *__restrict__ v,
                                               const double *__restrict__ scal, const double *__restrict__ partial, int mode) {
  // mode 0: d = s d + a v ; mode 1: d = a v ; mode 2: d = s d
  __shared__ double sh;
  const double sv = mode == 1 ? 0.0 : sval(scal, partial, s, &sh), av = mode == 2 ? 0.0 : sval(scal, partial, a, &sh);
#pragma unroll 4
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += gridDim.x * 256) {
    const int i = i0 + (i0 >= split ? gap : 0);
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
__global__ void k_axpy_multi(int n, int split, int gap, double *__restrict__ x, MultiArgs m) {
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += gridDim.x * 256) {
    const int i = i0 + (i0 >= split ? gap : 0);
    double s = x[i];
    for (int j = 0; j < m.k; ++j) s += m.c[j] * m.v[j][i];  // same order as the reference's x.add(h(i), tmp_vectors[i]) loop
    x[i] = s;
  }
}
// CG update (SolverCG): x += alpha d ; g += alpha h ; partial(g.g), alpha = value(gh) / value(dh)
__global__ __launch_bounds__(256) void k_cg_update(int n, double *__restrict__ x, const double *__restrict__ dvec, double *__restrict__ g,
                                                   const double *__restrict__ hvec, SRef a, const double *__restrict__ scal,
                                                   const double *__restrict__ partial_in, double *__restrict__ partial) {
  __shared__ double sh[5];
  const double alpha = sval(scal, partial_in, a, sh + 4);
  double acc = 0.0;
#pragma unroll 2
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    x[i] += alpha * dvec[i];
    const double gi = g[i] + alpha * hvec[i];
    g[i] = gi;
    acc += gi * gi;
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

static int ew_blocks(int n) { return std::max(1, std::min(2048, cdiv(n, 512))); }

static void axpby(nsx_handle *h, Span sp, double *d, SRef s, SRef a, const double *v, int mode, double bytes_per) {
  const int n = sp.n;
  LaunchScope ls(h, "axpby", bytes_per * n);
  hipLaunchKernelGGL(k_axpby, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, sp.split, sp.gap, d, s, a, v, h->scal.p, h->red_partial.p, mode);
}

void v_copy(nsx_handle *h, int n, double *d, const double *s) {
  if (d != s && n) HIP_CHECK(hipMemcpyAsync(d, s, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
}
void v_zero(nsx_handle *h, int n, double *d) {
  if (n) HIP_CHECK(hipMemsetAsync(d, 0, (size_t)n * sizeof(double), h->stream));
}
void v_add(nsx_handle *h, Span n, double *d, double a, const double *v) { axpby(h, n, d, SRef{1, -1, -1, 0, 0}, SRef{a, -1, -1, 0, 0}, v, 0, 24); }
void v_add_dev(nsx_handle *h, Span n, double *d, double a, int slot, const double *v) {
  axpby(h, n, d, SRef{1, -1, -1, 0, 0}, sref(h, a, slot, -1), v, 0, 24);
}
void v_sadd(nsx_handle *h, Span n, double *d, double s, double a, const double *v) {
  axpby(h, n, d, SRef{s, -1, -1, 0, 0}, SRef{a, -1, -1, 0, 0}, v, 0, 24);
}
void v_scale(nsx_handle *h, Span n, double *d, double a) { axpby(h, n, d, SRef{a, -1, -1, 0, 0}, SRef{0, -1, -1, 0, 0}, nullptr, 2, 16); }
void v_scale_dev_inv(nsx_handle *h, Span n, double *d, int slot) { axpby(h, n, d, sref(h, 1, -1, slot), SRef{0, -1, -1, 0, 0}, nullptr, 2, 16); }
void v_scale_vec(nsx_handle *h, int n, double *d, const double *f) {
  LaunchScope ls(h, "scale_vec", 24.0 * n);
  hipLaunchKernelGGL(k_scale_vec, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, d, f);
}
void v_axpy_multi(nsx_handle *h, Span sp, double *x, int k, double *const *vs, const double *coef) {
  const int n = sp.n;
  for (int j0 = 0; j0 < k; j0 += 32) {
    MultiArgs m;
    m.k = std::min(32, k - j0);
    for (int j = 0; j < m.k; ++j) {
      m.v[j] = vs[j0 + j];
      m.c[j] = coef[j0 + j];
    }
    LaunchScope ls(h, "axpy_multi", 8.0 * n * (2 + m.k));
    hipLaunchKernelGGL(k_axpy_multi, dim3(ew_blocks(n)), dim3(256), 0, h->stream, n, sp.split, sp.gap, x, m);
  }
}

// SolverCG helpers
void cg_update(nsx_handle *h, int n, double *x, const double *d, double *g, const double *hv, int gh_slot, int dh_slot, int res_slot) {
  LaunchScope ls(h, "cg_update", 48.0 * n);
  const int nb = red_blocks(n);
  hipLaunchKernelGGL(k_cg_update, dim3(nb), dim3(256), 0, h->stream, n, x, d, g, hv, sref(h, 1, gh_slot, dh_slot), h->scal.p,
                     h->red_partial.p, red_out(h, res_slot, nb));
  after_reduction(h, res_slot, nb);
}
// d = (value(num)/value(den)) d - h
void cg_direction(nsx_handle *h, int n, double *d, const double *hv, int num_slot, int den_slot) {
  axpby(h, n, d, sref(h, 1, num_slot, den_slot), SRef{-1, -1, -1, 0, 0}, hv, 0, 24);
}

double read_scalar(nsx_handle *h, int slot) {
  double v;
  read_scalars(h, slot, 1, &v);
  return v;
}
void read_scalars(nsx_handle *h, int slot0, int count, double *out) {
  if (count > 64) NSX_THROW(NSX_ERR_ARG, "internal: read_scalars range too long");
  NbArgs args;
  for (int i = 0; i < count; ++i) {
    args.nb[i] = h->slot_nb[slot0 + i];
    h->slot_nb[slot0 + i] = 0;
  }
  const unsigned long long seq = ++h->pub_seq;
  unsigned long long *flag_dev = (unsigned long long *)(h->pub_dev + N_SLOTS);
  volatile unsigned long long *flag_host = (volatile unsigned long long *)(h->pub_host + N_SLOTS);
  hipLaunchKernelGGL(k_publish, dim3(count), dim3(256), 0, h->stream, slot0, count, args, h->red_partial.p, h->scal.p, h->pub_dev, flag_dev, seq,
                     h->pub_counter.p);
  // poll the sequence number (bounded: fall back to a stream synchronisation, which also surfaces launch errors)
  unsigned long long spins = 0;
  while (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) != seq) {
    if (++spins > 200000000ull) {
      HIP_CHECK(hipStreamSynchronize(h->stream));
      if (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) != seq) NSX_THROW(NSX_ERR_HIP, "scalar publication never arrived");
      break;
    }
  }
  for (int i = 0; i < count; ++i) out[i] = h->pub_host[slot0 + i];
}
void write_scalar(nsx_handle *h, int slot, double v) {
  HIP_CHECK(hipStreamSynchronize(h->stream));
  h->scal_host[slot] = v;
  h->slot_nb[slot] = 0;
  HIP_CHECK(hipMemcpyAsync(h->scal.p + slot, h->scal_host + slot, sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

}  // namespace nsx
