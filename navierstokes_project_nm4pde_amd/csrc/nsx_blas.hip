// nsx_blas.hip — fused BLAS-1 for the Krylov drivers (the Epetra_Vector operations behind deal.II's
// SolverGMRES / SolverCG and the sadd/add/scale calls of reference Preconditioners.hpp:176,195,202-203,281,294-309,386,406,492-515).
//
// Scalars never visit the host inside an orthogonalisation sweep.  A reduction leaves per-block partial sums in
// h->red_partial[slot][0..nb); the CONSUMER kernel (the next add_and_dot / axpy / CG update) sums them itself in a fixed
// order while it starts up, so a dot product costs one launch, not two (a separate 1-block finalise kernel measured
// 4.6 us x 13 000 launches per step, profiles/r01).  Coefficients are SRef = c * value(num) / value(den) evaluated on the
// device.  Everything is deterministic: fixed grids, fixed-order sums, no atomics.
// Multi-GPU: every rank leaves the same number of partial sums, they are all-reduced element-wise over RCCL
// (comm_allreduce_partials) and consumed exactly as on one GPU.
#include "nsx_grid.hpp"
#include "nsx_ilu_lanes.hpp"

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

// Number of per-block partial sums of a reduction over n entries.  With a communicator every rank must use the SAME count
// (the partial sums are all-reduced element by element, see after_reduction), whatever its local size: a fixed grid.
constexpr int DIST_RED_BLOCKS = 256;
static int red_blocks(nsx_handle *h, int n) { return h->comm ? DIST_RED_BLOCKS : std::max(1, std::min(RED_BLOCKS, cdiv(n, 1024))); }

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

// Hold back / release the all-reduces of finished reductions (distributed runs).  On release, slots whose partial-sum regions
// are adjacent go out as one collective.
void defer_reductions(nsx_handle *h, bool on) {
  h->defer_red = on;
  if (on || h->pending_red.empty()) return;
  std::sort(h->pending_red.begin(), h->pending_red.end());
  size_t k = 0;
  while (k < h->pending_red.size()) {
    size_t e = k + 1;
    while (e < h->pending_red.size() && h->pending_red[e] == h->pending_red[e - 1] + 1) ++e;
    const int first = h->pending_red[k], count = (int)(e - k);
    comm_allreduce_partials(h, h->red_partial.p + (size_t)first * RED_STRIDE, (count - 1) * RED_STRIDE + DIST_RED_BLOCKS);
    k = e;
  }
  h->pending_red.clear();
}

void after_reduction(nsx_handle *h, int slot, int nb) {
  h->slot_nb[slot] = nb > 1 ? nb : 0;
  if (!h->comm) return;
  if (h->defer_red && nb == DIST_RED_BLOCKS) {
    h->pending_red.push_back(slot);
    return;
  }
  // global sum needed before anybody consumes the value
  if (nb == DIST_RED_BLOCKS) {
    // all-reduce the partial sums themselves (2 KB instead of 8 B costs the same latency) and let the consumer add
    // them up as on one GPU: no finalising launch in front of the collective
    comm_allreduce_partials(h, h->red_partial.p + (size_t)slot * RED_STRIDE, nb);
  } else {
    finalize_slots(h, slot, 1);
    comm_allreduce_scalars(h, slot, 1);
  }
}
double *red_out(nsx_handle *h, int slot, int nb) {
  return nb > 1 ? h->red_partial.p + (size_t)slot * RED_STRIDE : h->scal.p + slot;
}

void v_dot(nsx_handle *h, Span sp, const double *a, const double *b, int slot) {
  const int n = sp.n;
  LaunchScope ls(h, "dot", (a == b ? 8.0 : 16.0) * n);
  const int nb = red_blocks(h, n);
  hipLaunchKernelGGL((k_reduce<OP_DOT>), dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, const_cast<double *>(a), SRef{0, -1, -1, 0, 0}, nullptr, b,
                     h->scal.p, h->red_partial.p, red_out(h, slot, nb));
  after_reduction(h, slot, nb);
}

void v_add_and_dot(nsx_handle *h, Span sp, double *d, double a, int aslot, const double *v, const double *w, int slot) {
  const int n = sp.n;
  LaunchScope ls(h, "add_and_dot", (w == d ? 24.0 : 32.0) * n);
  const int nb = red_blocks(h, n);
  hipLaunchKernelGGL((k_reduce<OP_ADD_AND_DOT>), dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, d, sref(h, a, aslot, -1), v, w, h->scal.p,
                     h->red_partial.p, red_out(h, slot, nb));
  after_reduction(h, slot, nb);
}

// ---- modified Gram-Schmidt sweep in ONE persistent launch -----------------------------------------------------------
// deal.II's SolverGMRES orthogonalises the new Krylov vector w with the chain  h(0) = w.v_0 ;
// h(i+1) = w.add_and_dot(-h(i), v_i, v_{i+1}) ; |w|^2 = w.add_and_dot(-h(dim-1), v_{dim-1}, w): dim+1 dependent global
// reductions.  As separate launches every link streams w (read + write) and two basis vectors, 32 B per entry; here the
// grid is co-resident, every thread keeps its entries of w and of the current v_i in registers for the whole sweep, and a
// link costs ONE read of the next basis vector (8 B per entry, prefetched before the wait) plus a grid-wide exchange of
// the partial sums (nsx_grid.hpp: mailboxes, no atomics on shared counters): workgroup 0 waits for all mailboxes, adds them
// in a fixed order and publishes the total, which every workgroup picks up.  (Letting every workgroup read all mailboxes
// itself — one hop instead of two — measured slower: 60 against 50 us per sweep, the polling traffic gets in its own way.)
// The arithmetic of each entry is that of the chain (w += (-h) v_i), sums are fixed-order, so results do not depend on
// timing.  Every wait is bounded by a wall-clock timeout: a grid that is not co-resident (another stream or process holds
// compute units) ends without touching w, and the host falls back to the launch-per-link chain (v_mgs).
constexpr int MGS_MAX_WG = 512;
constexpr int MGS_STEPS = 32;     // >= max_n_tmp_vectors + 1
constexpr size_t MGS_REGION = (size_t)MGS_STEPS * MGS_MAX_WG + MGS_STEPS;  // words per mailbox region (+ the totals)
// after the two mailbox regions: one word per workgroup = sequence number of the last sweep whose part of w it wrote back
// (distinct addresses: a shared counter would serialise 512 atomics at the end of every sweep)
constexpr size_t MGS_TAIL = MGS_MAX_WG;

struct MgsArgs {
  const double *v[MGS_STEPS];
};

template <int E>
__global__ __launch_bounds__(256) void k_mgs(int n, int split, int gap, double *__restrict__ w, MgsArgs V, int dim,
                                             unsigned long long *box, unsigned long long *box_next, int reset_wg, int reset_steps,
                                             double *__restrict__ scal_out, int *err_host, unsigned long long *tail, int normalize, int consider,
                                             double *pub_vals, unsigned long long *pub_flag, unsigned long long seq, int drop_wg) {
  __shared__ double sh[2][4];  // two buffers: a wave may start the next sum while a slower one still reads this one
  __shared__ unsigned long long bc;
  __shared__ int s_err;  // raised by any thread whose wait timed out; read after the next barrier
  __shared__ double tots[MGS_STEPS];
  const int nwg = gridDim.x, wg = blockIdx.x, T = nwg * 256, t = wg * 256 + threadIdx.x;
  if (threadIdx.x == 0) s_err = 0;
  unsigned long long *total = box + (size_t)MGS_STEPS * MGS_MAX_WG, *total_next = box_next + (size_t)MGS_STEPS * MGS_MAX_WG;
  // leave the other region empty for the next launch (stream order makes this visible to it): its last user filled
  // reset_steps rows of reset_wg mailboxes, possibly more than this grid has workgroups
  for (int q = t; q < reset_steps * reset_wg; q += T) box_next[(size_t)(q / reset_wg) * MGS_MAX_WG + q % reset_wg] = GX_EMPTY;
  if (wg == 0 && threadIdx.x < MGS_STEPS) total_next[threadIdx.x] = GX_EMPTY;
  double wv[E], vc[E], vn[E];
  int idx[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int i0 = t + k * T;
    idx[k] = i0 < n ? i0 + (i0 >= split ? gap : 0) : -1;
    wv[k] = idx[k] >= 0 ? w[idx[k]] : 0.0;
    vc[k] = idx[k] >= 0 ? ld_stream<1>(V.v[0] + idx[k]) : 0.0;
    vn[k] = 0.0;
  }
  // consider: SolverGMRES' re-orthogonalisation test (every 5th inner iteration) needs |w| BEFORE the sweep: one more link
  // in front (mailbox row dim + 1), and the decision whether w may be normalised is taken here exactly as the host takes it
  double norm0_sq = 0.0;
  int lerr = 0;
  bool dead = false;
  for (int s = consider ? -1 : 0; s <= dim; ++s) {
    const bool pre = s < 0;
    const int ri = pre ? dim + 1 : s;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) acc += wv[k] * ((!pre && s < dim) ? vc[k] : wv[k]);
    const double part = gx_block_sum(acc, sh[0]);
    unsigned long long *row = box + (size_t)ri * MGS_MAX_WG;
    if (threadIdx.x == 0 && wg != drop_wg) gx_post(row + wg, part);  // drop_wg >= 0: fault injection (NSX_GX_DROP_WG), a workgroup that never arrives
    if (wg == 0) {
      double a = 0.0;
      for (int q = threadIdx.x; q < nwg; q += 256) a += gx_wait_value(row + q, &lerr);
      if (lerr) s_err = 1;
      const double tot = gx_block_sum(a, sh[1]);
      // a total built on a timed-out mailbox must never go out: the others then time out as well and nobody writes w
      if (threadIdx.x == 0 && !s_err) {
        scal_out[ri] = tot;
        gx_post(total + ri, tot);
        tots[ri] = tot;
      }
    }
    // The next basis vector is fetched while the sums are exchanged.  Its loads are issued AFTER this workgroup's partial sum
    // (and, in workgroup 0, the total) has gone out and after the first poll: issued in front, they queue ahead of the exchange's
    // own traffic (tools/exchange_bench.hip: 3.2 against 3.0 us per link at 512 workgroups x 8 loads per thread)
    unsigned long long first = GX_EMPTY;
    if (threadIdx.x == 0) first = gx_load(total + ri);
    if (!pre && s + 1 < dim) {
      const double *__restrict__ vp = V.v[s + 1];
#pragma unroll
      for (int k = 0; k < E; ++k) vn[k] = idx[k] >= 0 ? ld_stream<1>(vp + idx[k]) : 0.0;
    }
    if (threadIdx.x == 0) {
      bc = first != GX_EMPTY ? first : gx_wait(total + ri, &lerr);
      if (lerr) s_err = 1;
    }
    __syncthreads();
    dead = s_err != 0;
    const double hs = __longlong_as_double((long long)bc);
    if (dead) break;
    if (pre) {
      norm0_sq = hs;
      continue;
    }
    if (s < dim) {
      const double alpha = -1.0 * hs;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        wv[k] += alpha * vc[k];
        vc[k] = vn[k];
      }
    } else if (normalize) {  // vv *= 1. / s with s = sqrt(|vv|^2), skipped for s == 0 (SolverGMRES)
      const double nrm = sqrt(hs);
      // no normalisation if the test asks for a second sweep: s <= 10 |vv_start| sqrt(eps), sqrt(eps) = 2^-26
      const bool second_sweep = consider && !(nrm > 10. * sqrt(norm0_sq) * 1.4901161193847656e-08);
      if (nrm != 0.0 && !second_sweep) {
        const double inv = 1. / nrm;
#pragma unroll
        for (int k = 0; k < E; ++k) wv[k] = inv * wv[k];
      }
    }
  }
  if (dead) {
    // w stays as it was.  Tell the host (mapped word) and, from workgroup 0, wake it up
    if (threadIdx.x == 0) {
      __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (wg == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  // hand the coefficients to the host: values, then the flag it is polling, both in fine-grained mapped host memory.  The
  // stores are acknowledged (vmcnt) before the flag goes out; a system-scope release would also write back the whole L2.
  if (wg == 0) {
    __syncthreads();
    if ((int)threadIdx.x <= dim + (consider ? 1 : 0)) __hip_atomic_store(pub_vals + threadIdx.x, tots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) tail[wg] = seq;  // this workgroup commits its part of w
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (idx[k] >= 0) w[idx[k]] = wv[k];
}

// ---- the same sweep with M links per grid-wide exchange ---------------------------------------------------------------
// The chain's coefficients are h_j = v_j . w_j with w_{j+1} = w_j - h_j v_j.  For the links j0 .. j0+M-1 of a block, by
// linearity of the dot product (no orthogonality of the basis is assumed),
//     h_j = v_j . (w_{j0} - sum_{j0 <= i < j} h_i v_i) = r_j - sum_{j0 <= i < j} (v_i . v_j) h_i ,   r_j = v_j . w_{j0} :
// the M numbers r_j and the M (M-1) / 2 numbers v_i . v_j are sums over the SAME registers (the block's M basis vectors and
// w_{j0} are held by the thread), so they travel in ONE exchange, workgroup 0 solves the unit lower-triangular M x M system
// and hands out h_{j0..j0+M-1}, and every thread applies w += (-h_j) v_j for j ascending exactly as the chain does.  The
// entries of w see the chain's operations in the chain's order; the coefficients differ from the chain's by the rounding
// of the dot products only (identical in exact arithmetic, whatever the basis).  A sweep of `dim` links costs
// ceil(dim / M) + 1 exchanges instead of dim + 1 (tools/exchange_bench.hip: 2.4 - 3.7 us each).  M = 1 is k_mgs.
// Mailboxes: value-major, box[(x * NV + v) * nwg + wg] for exchange x, so workgroup 0 reads them coalesced.
constexpr int MGS_BLK_TOT = 64;  // words reserved for the totals of a region ((M + 1) per exchange)
constexpr size_t MGS_BLK_REGION = 57344 + MGS_BLK_TOT;  // (ceil(28 / M) + 1) * NV * 512 words for M <= 5, + the totals

#ifdef NSX_MGS_TRACE  // development only (tools/mgs_bench.hip): wall-clock stamps of workgroup 0 and of the last workgroup
__device__ unsigned long long *g_mgs_trace = nullptr;
#define MGS_STAMP()                                                                                         \
  do {                                                                                                      \
    if (g_mgs_trace && threadIdx.x == 0 && (wg == 0 || wg == nwg - 1) && n_stamp < 64)                       \
      g_mgs_trace[(wg == 0 ? 0 : 64) + n_stamp++] = wall_clock64();                                          \
  } while (0)
#else
#define MGS_STAMP() \
  do {              \
  } while (0)
#endif

template <int E, int M, bool PF>
__global__ __launch_bounds__(256) void k_mgs_blk(int n, int split, int gap, double *__restrict__ w, MgsArgs V, int dim, unsigned long long *box,
                                                 unsigned long long *box_next, int reset_words, double *__restrict__ scal_out, int *err_host,
                                                 unsigned long long *tail, int normalize, int consider, double *pub_vals,
                                                 unsigned long long *pub_flag, unsigned long long seq, int drop_wg) {
  constexpr int NP = M * (M - 1) / 2, NV = M + NP + 1;  // r_0..r_{M-1}, pairs (i < j) at M + j (j - 1) / 2 + i, |w|^2 before the sweep
  __shared__ double sh[4][NV];
  __shared__ double bc[M + 1];
  __shared__ int s_err;
  __shared__ double tots[MGS_STEPS + 2];
  const int nwg = gridDim.x, wg = blockIdx.x, T = nwg * 256, t = wg * 256 + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  [[maybe_unused]] int n_stamp = 0;
  MGS_STAMP();
  if (threadIdx.x == 0) s_err = 0;
  unsigned long long *total = box + (MGS_BLK_REGION - MGS_BLK_TOT), *total_next = box_next + (MGS_BLK_REGION - MGS_BLK_TOT);
  for (int q = t; q < reset_words; q += T) box_next[q] = GX_EMPTY;
  if (wg == 0 && threadIdx.x < MGS_BLK_TOT) total_next[threadIdx.x] = GX_EMPTY;
  double wv[E], vb[M][E], vn[PF ? M : 1][PF ? E : 1];
  int idx[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int i0 = t + k * T;
    idx[k] = i0 < n ? i0 + (i0 >= split ? gap : 0) : -1;
    wv[k] = idx[k] >= 0 ? w[idx[k]] : 0.0;
  }
  auto load_block = [&](double (&dst)[M][E], int j0) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      if (j0 + i < dim) {
        const double *__restrict__ vp = V.v[j0 + i];
#pragma unroll
        for (int k = 0; k < E; ++k) dst[i][k] = idx[k] >= 0 ? ld_stream<1>(vp + idx[k]) : 0.0;
      } else {
#pragma unroll
        for (int k = 0; k < E; ++k) dst[i][k] = 0.0;
      }
    }
  };
  load_block(vb, 0);
  const int nblk = (dim + M - 1) / M;
  double norm0_sq = 0.0;
  int lerr = 0;
  bool dead = false;
  for (int x = 0; x <= nblk; ++x) {
    const bool last = x == nblk;
    const int j0 = x * M, mb = last ? 0 : (dim - j0 < M ? dim - j0 : M);
    const bool pre = x == 0 && consider;
    // which of the NV values this exchange carries (wave-uniform)
    unsigned int used = last ? 1u : ((1u << mb) - 1u) | (((1u << (mb * (mb - 1) / 2)) - 1u) << M) | (pre ? 1u << (NV - 1) : 0u);
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    if (last) {
#pragma unroll
      for (int k = 0; k < E; ++k) acc[0] += wv[k] * wv[k];
    } else {
#pragma unroll
      for (int i = 0; i < M; ++i)
        if (i < mb) {
#pragma unroll
          for (int k = 0; k < E; ++k) acc[i] += wv[k] * vb[i][k];
#pragma unroll
          for (int i2 = 0; i2 < i; ++i2) {
#pragma unroll
            for (int k = 0; k < E; ++k) acc[M + i * (i - 1) / 2 + i2] += vb[i2][k] * vb[i][k];
          }
        }
      if (pre) {
#pragma unroll
        for (int k = 0; k < E; ++k) acc[NV - 1] += wv[k] * wv[k];
      }
    }
    // fixed-order sums over the workgroup, one barrier for all values
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (used >> v & 1u) {
        const double s = gx_wave_sum(acc[v]);
        if (lane == 0) sh[wave][v] = s;
      }
    __syncthreads();
    MGS_STAMP();  // local sums done (the block's loads have arrived)
    unsigned long long *xbox = box + (size_t)x * NV * nwg;
    if (threadIdx.x < NV && (used >> threadIdx.x & 1u) && wg != drop_wg) {  // drop_wg: fault injection, see k_mgs
      const int v = threadIdx.x;
      gx_post(xbox + (size_t)v * nwg + wg, (sh[0][v] + sh[1][v]) + (sh[2][v] + sh[3][v]));
    }
    unsigned long long *xtot = total + x * (M + 1);
    if (wg == 0) {
      __syncthreads();  // sh is reused below
      double a[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v] = 0.0;
      for (int q = threadIdx.x; q < nwg; q += 256) {
        unsigned long long b[NV], t0 = 0;
        for (unsigned int spin = 1;; ++spin) {
          bool all = true;
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (used >> v & 1u) b[v] = gx_load(xbox + (size_t)v * nwg + q);
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (used >> v & 1u) all = all && b[v] != GX_EMPTY;
          if (all) break;
          __builtin_amdgcn_s_sleep(1);
          if ((spin & 255u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > GX_TIMEOUT_TICKS) {
              lerr = 1;
              break;
            }
          }
        }
        if (lerr) break;
#pragma unroll
        for (int v = 0; v < NV; ++v)
          if (used >> v & 1u) a[v] += __longlong_as_double((long long)b[v]);
      }
      if (lerr) s_err = 1;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (used >> v & 1u) {
          const double s = gx_wave_sum(a[v]);
          if (lane == 0) sh[wave][v] = s;
        }
      __syncthreads();
      // a coefficient built on a timed-out mailbox must never go out: the others then time out as well and nobody writes w
      if (threadIdx.x == 0 && !s_err) {
        double tv[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) tv[v] = (used >> v & 1u) ? (sh[0][v] + sh[1][v]) + (sh[2][v] + sh[3][v]) : 0.0;
        if (last) {
          scal_out[dim] = tv[0];
          gx_post(xtot, tv[0]);
          tots[dim] = tv[0];
        } else {
          double hc[M];
#pragma unroll
          for (int j = 0; j < M; ++j) {
            double s = tv[j];
#pragma unroll
            for (int i = 0; i < j; ++i) s -= tv[M + j * (j - 1) / 2 + i] * hc[i];
            hc[j] = s;
            if (j < mb) {
              scal_out[j0 + j] = s;
              gx_post(xtot + j, s);
              tots[j0 + j] = s;
            }
          }
          if (pre) {
            scal_out[dim + 1] = tv[NV - 1];
            gx_post(xtot + M, tv[NV - 1]);
            tots[dim + 1] = tv[NV - 1];
          }
        }
      }
    }
    MGS_STAMP();  // posted (workgroup 0: coefficients out)
    // the next block of basis vectors is fetched while the sums are exchanged (behind this workgroup's post and first poll,
    // see k_mgs)
    const int nw = last ? 1 : mb + (pre ? 1 : 0);  // words to pick up: h of the block (+ |w|^2 before the sweep in word M)
    unsigned long long first = GX_EMPTY;
    const int myword = (int)threadIdx.x < (last ? 1 : mb) ? (int)threadIdx.x : M;
    if ((int)threadIdx.x < nw) first = gx_load(xtot + myword);
    if constexpr (PF) {
      if (!last && x + 1 < nblk) load_block(vn, j0 + M);
    }
    if ((int)threadIdx.x < nw) {
      const unsigned long long b = first != GX_EMPTY ? first : gx_wait(xtot + myword, &lerr);
      if (lerr) s_err = 1;
      bc[myword] = __longlong_as_double((long long)b);
    }
    __syncthreads();
    MGS_STAMP();  // coefficients picked up
    dead = s_err != 0;
    if (dead) break;
    if (last) {
      if (normalize) {  // vv *= 1. / s with s = sqrt(|vv|^2), skipped for s == 0 (SolverGMRES)
        const double nrm = sqrt(bc[0]);
        // no normalisation if the test asks for a second sweep: s <= 10 |vv_start| sqrt(eps), sqrt(eps) = 2^-26
        const bool second_sweep = consider && !(nrm > 10. * sqrt(norm0_sq) * 1.4901161193847656e-08);
        if (nrm != 0.0 && !second_sweep) {
          const double inv = 1. / nrm;
#pragma unroll
          for (int k = 0; k < E; ++k) wv[k] = inv * wv[k];
        }
      }
    } else {
      if (pre) norm0_sq = bc[M];
#pragma unroll
      for (int i = 0; i < M; ++i)
        if (i < mb) {
          const double alpha = -1.0 * bc[i];
#pragma unroll
          for (int k = 0; k < E; ++k) wv[k] += alpha * vb[i][k];
        }
      if (x + 1 < nblk) {
        if constexpr (PF) {
#pragma unroll
          for (int i = 0; i < M; ++i)
#pragma unroll
            for (int k = 0; k < E; ++k) vb[i][k] = vn[i][k];
        } else {
          load_block(vb, j0 + M);
        }
      }
    }
    __syncthreads();  // bc and sh are rewritten by the next exchange
  }
  if (dead) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (wg == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (wg == 0) {
    if ((int)threadIdx.x <= dim + (consider ? 1 : 0)) __hip_atomic_store(pub_vals + threadIdx.x, tots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) tail[wg] = seq;  // this workgroup commits its part of w
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (idx[k] >= 0) w[idx[k]] = wv[k];
  MGS_STAMP();
}

// ---- the whole sweep in ONE grid-wide exchange ---------------------------------------------------------------------------
// The linearity that k_mgs_blk uses for M links of the chain holds for all of them:  h_j = r_j - sum_{i<j} G_ji h_i  with
// r_j = v_j . w (the vector as it ENTERS the sweep) and G the Gram matrix of the basis (no orthogonality assumed).  G is kept
// on the device per GMRES nesting level: the sweep that meets v_{dim-1} for the first time computes its row g_i = v_{dim-1} . v_i
// beside the r_j (the same registers), all older rows were computed by the earlier sweeps of the cycle.  So ONE exchange carries
// r_0..r_{dim-1}, g_0..g_{dim-1} and |w|^2; every workgroup then solves the same unit-lower-triangular system from the same
// totals, applies  w += (-h_j) v_j  for j ascending (the chain's operations on every entry, in the chain's order) from the
// basis vectors it still holds in registers, and gets |w|^2 AFTER the sweep without another exchange from
//     |w - sum_j h_j v_j|^2 = |w|^2 - 2 h.r + h^T G h
// (a difference of numbers of size |w|^2: taken when the sweep leaves more than 1 % of the norm, i.e. its rounding error stays
// below 1e-12 |w'|^2; otherwise a second exchange sums |w'|^2 itself).  A sweep of `dim` links costs one exchange (two when
// the formula is refused) instead of dim / 2 + 1, and every basis vector is still read exactly once as long as the thread can
// keep the block (dim <= DMAX); beyond, the oldest dim - DMAX vectors are streamed twice (dots, then update).
// The exchange is two hops like the others: value v is summed over the workgroups' mailboxes by workgroup v % nwg (the 2 dim + 1
// sums are spread over the grid instead of queueing in workgroup 0), the totals are picked up by everybody.
// Mailboxes: box[v * nwg + wg], totals behind them at box[MGS_ONE_VALS * MGS_MAX_WG + v].
// Basis vectors beyond the block kept in registers are read TWICE by a sweep (dot pass, update pass).  The register-resident ones
// are streamed with non-temporal loads (-DNSX_NT & 1: they must not evict F and the ILU factors from the 256-MiB Infinity Cache,
// profiles/r02_cache_policy_and_links.txt); the older ones take the default policy, so that the update pass finds in that cache what
// the dot pass brought in (-DNSX_MGS_OLD_NT=1: non-temporal as well, rounds 2-3).
#ifndef NSX_MGS_OLD_NT
#define NSX_MGS_OLD_NT 0
#endif
__device__ __forceinline__ double ld_twice(const double *p) {
  if constexpr (NSX_MGS_OLD_NT != 0) return ld_stream<1>(p);
  else return *p;
}
constexpr int MGS_ONE_VALS = 2 * (MGS_STEPS - 2) + 2;  // r_j, g_j (j < 30), |w|^2 before, |w|^2 after (second exchange)
static_assert((size_t)MGS_ONE_VALS * MGS_MAX_WG + MGS_ONE_VALS <= MGS_BLK_REGION, "the one-exchange sweep shares the mailbox regions of k_mgs_blk");

// ---- the same sweep in a DISTRIBUTED run (DIST): the single exchange also crosses the ranks -------------------------------------
// The reducers leave the LOCAL totals in ext.vals (one word each) and count themselves in at ext.arrive.  On the communication
// stream the host has enqueued, right behind this launch:  k_ext_wait (spins until the count is complete)  ->  ncclAllReduce of
// ext.vals over the ranks  ->  k_ext_release (stores this sweep's sequence number in ext.flag).  Every workgroup waits for that
// flag instead of for the grid totals and goes on with the GLOBAL sums: the ten newest basis vectors stay in registers across
// the collective, where the two-pass sweep (k_ls_dots / k_ls_update) reads the basis twice and pays two more launches.
// Failure is agreed on by all ranks: a reducer whose mailbox wait timed out (or k_ext_wait, if the count never completes) raises
// (All hand-offs are RELAXED agent-scope atomics behind an explicit s_waitcnt, like the mailboxes of nsx_grid.hpp: an acquire or a
// release at agent scope makes the compiler invalidate / write back the XCD's whole L2 around the access -- polled by 448 workgroups
// that doubled the time of the sweep's first phase: profiles/r04_ext_collective_timeline.txt.)
// ext.vals[MGS_EXT_FAIL], the collective SUMS that word, and a non-zero sum makes every rank's grid end without touching w; the
// hosts then all redo the sweep with the two-pass path.  The wait for the flag is bounded ABOVE the other time-outs (mailboxes 2 s,
// k_ext_wait 4 s: a rank that fails locally needs both before its collective goes out): 8 s.  A flag wait that still times out is a
// verdict of ONE rank -- its peers' grids may have gone on with the global sums -- so it must not change this rank's collective
// sequence: the host waits for the (late) collective, finishes the sweep from the global sums it delivered (v_mgs: same coefficients,
// same updates, no further collective) and raises ext.vals[MGS_EXT_LEAVE] in its NEXT sweep; that word is summed like the failure
// word, and a non-zero sum takes every rank to the two-pass sweep together.  When the Gram formula for |w'|^2 is refused (the sweep
// removed > 99 % of the norm: decided from the global sums, i.e. alike on every rank) the grid leaves the LOCAL sum of |w'|^2 in
// ext.norm_out, does not normalise and reports "norm pending": the host all-reduces that word (the second collective).
constexpr int MGS_EXT_VALS = 64, MGS_EXT_FAIL = 63, MGS_EXT_LEAVE = 62;
constexpr unsigned long long GX_EXT_TIMEOUT_TICKS = 800000000ull;  // 8 s at 100 MHz: above the mailbox wait (2 s) + k_ext_wait (4 s) a failing peer needs before its collective goes out
static_assert(GX_EXT_TIMEOUT_TICKS > 3 * GX_TIMEOUT_TICKS, "the flag wait must outlast a peer's mailbox wait + k_ext_wait");
struct MgsExt {
  double *vals;              // [MGS_EXT_VALS] this sweep's buffer: local totals, then (after the collective) the global ones
  double *vals_other;        // the other buffer: its failure word is cleared for the next sweep
  unsigned int *arrive;      // reducers that have delivered, cumulative over all sweeps
  unsigned long long *flag;  // sequence number of the last sweep whose collective is complete
  unsigned long long *abort_seq;  // sequence number of the last sweep a workgroup gave up on: the verdict of the WHOLE grid (see the wait)
  double *norm_out;          // local |w'|^2 when the formula is refused
  int leave;                 // 1: this rank asks all ranks to leave the persistent sweep (an earlier flag wait of its own timed out)
};
__device__ __forceinline__ double ext_ld(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ext_st(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int E, int DMAX, bool DIST>
__global__ __launch_bounds__(256) void k_mgs_one(int n, int split, int gap, double *__restrict__ w, MgsArgs V, int dim, double *__restrict__ gram,
                                                 unsigned long long *box, unsigned long long *box_next, int reset_words, double *__restrict__ scal_out,
                                                 int *err_host, unsigned long long *tail, int normalize, int consider, double *pub_vals,
                                                 unsigned long long *pub_flag, unsigned long long seq, int drop_wg, double norm_guard, MgsExt ext) {
  __shared__ double sh[4][MGS_ONE_VALS];   // per-wave sums of every value
  __shared__ double tot[MGS_ONE_VALS];     // grid totals
  __shared__ double G[MGS_STEPS][MGS_STEPS + 1], hc[MGS_STEPS];
  __shared__ double s_norm2;
  __shared__ int s_err;
  const int nwg = gridDim.x, wg = blockIdx.x, T = nwg * 256, t = wg * 256 + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_err = 0;
  [[maybe_unused]] int n_stamp = 0;  // development only (tools/mgs_bench.hip): the stamps are empty without NSX_MGS_TRACE
  MGS_STAMP();  // start
  unsigned long long *total = box + (size_t)MGS_ONE_VALS * MGS_MAX_WG, *total_next = box_next + (size_t)MGS_ONE_VALS * MGS_MAX_WG;
  for (int q = t; q < reset_words; q += T) box_next[q] = GX_EMPTY;
  if (wg == 0 && threadIdx.x < MGS_ONE_VALS) total_next[threadIdx.x] = GX_EMPTY;
  if constexpr (DIST) {
    if (wg == 0 && threadIdx.x < MGS_EXT_VALS) ext_st(ext.vals_other + threadIdx.x, 0.0);  // the next sweep's buffer (its failure word in particular)
  }
  const int nvals = 2 * dim + 1;  // r_j at j, g_j at dim + j, |w|^2 at 2 dim
  // the older rows of the Gram matrix (written by the earlier sweeps of this cycle) are requested first and parked in registers:
  // read behind the exchange they were a trip through memory on the critical path of every workgroup
  // (only where the registers are there: the E = 10 instantiation would lose its second wave per SIMD, and with it the resident grid)
  constexpr bool PRE = E <= 8;
  constexpr int GPRE = PRE ? (MGS_STEPS * MGS_STEPS + 255) / 256 : 1;
  double gpre[GPRE];
  if constexpr (PRE) {
#pragma unroll
    for (int k = 0; k < GPRE; ++k) {
      const int q = threadIdx.x + 256 * k, r_ = q / MGS_STEPS, c_ = q % MGS_STEPS;
      gpre[k] = (r_ < dim - 1 && c_ <= r_) ? gram[r_ * 32 + c_] : 0.0;
    }
  }
  double wv[E], vb[DMAX][E];
  int idx[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int i0 = t + k * T;
    idx[k] = i0 < n ? i0 + (i0 >= split ? gap : 0) : -1;
    wv[k] = idx[k] >= 0 ? w[idx[k]] : 0.0;
  }
  // the block the thread keeps: the LAST min(dim, DMAX) basis vectors (the newest one, whose Gram row is due, is among them)
  const int j_keep = dim > DMAX ? dim - DMAX : 0;
#pragma unroll
  for (int i = 0; i < DMAX; ++i) {
    const double *__restrict__ vp = j_keep + i < dim ? V.v[j_keep + i] : nullptr;
#pragma unroll
    for (int k = 0; k < E; ++k) vb[i][k] = (vp && idx[k] >= 0) ? ld_stream<1>(vp + idx[k]) : 0.0;
  }
  // v_{dim-1} in registers of its own (static index): the second operand of the Gram row
  double vl[E];
#pragma unroll
  for (int k = 0; k < E; ++k) vl[k] = 0.0;
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (j_keep + i == dim - 1) {
#pragma unroll
      for (int k = 0; k < E; ++k) vl[k] = vb[i][k];
    }
  auto wave_post = [&](int v, double a) {  // this wave's sum of value v
    const double s_ = gx_wave_sum(a);
    if (lane == 0) sh[wave][v] = s_;
  };
  {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) a += wv[k] * wv[k];
    wave_post(2 * dim, a);
  }
  for (int j = 0; j < j_keep; ++j) {  // older vectors: streamed, not kept (dim > DMAX only)
    const double *__restrict__ vp = V.v[j];
    double ar = 0.0, ag = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const double x_ = idx[k] >= 0 ? ld_twice(vp + idx[k]) : 0.0;
      ar += wv[k] * x_;
      ag += vl[k] * x_;
    }
    wave_post(j, ar);
    wave_post(dim + j, ag);
  }
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (j_keep + i < dim) {
      double ar = 0.0, ag = 0.0;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        ar += wv[k] * vb[i][k];
        ag += vl[k] * vb[i][k];
      }
      wave_post(j_keep + i, ar);
      wave_post(dim + j_keep + i, ag);
    }
  if constexpr (PRE) {
#pragma unroll
    for (int k = 0; k < GPRE; ++k) {
      const int q = threadIdx.x + 256 * k, r_ = q / MGS_STEPS, c_ = q % MGS_STEPS;
      if (r_ < dim - 1 && c_ <= r_) G[r_][c_] = gpre[k];
    }
  }
  __syncthreads();
  MGS_STAMP();  // loads arrived, local sums done
  // ---- hop 1: mailboxes; value v is summed by workgroup v % nwg
  int lerr = 0;
  for (int v = threadIdx.x; v < nvals; v += 256)
    if (wg != drop_wg) gx_post(box + (size_t)v * nwg + wg, (sh[0][v] + sh[1][v]) + (sh[2][v] + sh[3][v]));  // drop_wg: fault injection, see k_mgs
  for (int v = wg; v < nvals; v += nwg) {
    double a = 0.0;
    for (int q = threadIdx.x; q < nwg; q += 256) a += gx_wait_value(box + (size_t)v * nwg + q, &lerr);
    if (lerr) s_err = 1;
    __syncthreads();  // sh is reused by the block sum below (and s_err must be seen)
    const double s_ = gx_wave_sum(a);
    if (lane == 0) sh[wave][0] = s_;
    __syncthreads();
    if constexpr (DIST) {
      if (threadIdx.x == 0) {
        if (!s_err) ext_st(ext.vals + v, (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]));
        else ext_st(ext.vals + MGS_EXT_FAIL, 1.0);  // summed over the ranks: everybody learns of it
        if (v == 0 && ext.leave) ext_st(ext.vals + MGS_EXT_LEAVE, 1.0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ext.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // counted in either way: the collective must go out
      }
    } else {
      if (threadIdx.x == 0 && !s_err) gx_post(total + v, (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]));  // a total built on a timed-out mailbox never goes out
    }
    __syncthreads();
  }
  MGS_STAMP();  // posted, and (reducers) totals out
  // ---- hop 2: everybody picks up the totals
  if constexpr (DIST) {
    if (threadIdx.x == 0) {  // the collective of this sweep is complete once the flag carries its sequence number
      // A grid that gives up must give up as a whole.  The likely reason for a collective that does not come is that its kernel finds
      // no place on the device WHILE this grid holds it (RCCL's generic kernel: 256 threads x 264 VGPRs; measured with a self-addressed
      // send / receive, NSX_EXT_SELF_P2P): then the first workgroup that leaves makes room, the collective runs, and the workgroups
      // still waiting would see the flag and go on to write w.  So the first one to time out records the sweep in abort_seq BEFORE it
      // leaves, and a workgroup that sees the flag looks there before it believes it.
      unsigned long long t0 = 0;
      for (unsigned int spin = 1; __hip_atomic_load(ext.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if ((spin & 255u) == 0) {
          if (__hip_atomic_load(ext.abort_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) {
            s_err = 1;
            break;
          }
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > GX_EXT_TIMEOUT_TICKS) {
            __hip_atomic_store(ext.abort_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_err = 1;
            break;
          }
        }
      }
      if (!s_err && __hip_atomic_load(ext.abort_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq) s_err = 1;
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nvals; v += 256) tot[v] = ext_ld(ext.vals + v);
    if (threadIdx.x == 0 && (ext_ld(ext.vals + MGS_EXT_FAIL) != 0.0 || ext_ld(ext.vals + MGS_EXT_LEAVE) != 0.0)) s_err = 1;  // some rank's grid was not complete, or a rank asks everybody to leave
  } else {
    for (int v = threadIdx.x; v < nvals; v += 256) {
      tot[v] = gx_wait_value(total + v, &lerr);
      if (lerr) s_err = 1;
    }
  }
  __syncthreads();
  MGS_STAMP();  // totals picked up
  bool dead = s_err != 0;
  double xo[E];  // entries of the next streamed (older) vector of the update below
#pragma unroll
  for (int k = 0; k < E; ++k) xo[k] = (!dead && j_keep > 0 && idx[k] >= 0) ? ld_twice(V.v[0] + idx[k]) : 0.0;
  if (!dead) {
    // Gram matrix of the basis: older rows parked in G before the exchange, the new row from this exchange; then h by forward substitution and the
    // norm after the sweep, one wave, lane j = link j
    if constexpr (!PRE) {
      for (int q = threadIdx.x; q < (dim - 1) * MGS_STEPS; q += 256) {
        const int r_ = q / MGS_STEPS, c_ = q % MGS_STEPS;
        if (c_ <= r_) G[r_][c_] = gram[r_ * 32 + c_];
      }
    }
    if ((int)threadIdx.x < dim) G[dim - 1][threadIdx.x] = tot[dim + threadIdx.x];
    __syncthreads();
    if (wave == 0) {
      double hj = 0.0;
      const int col = lane < dim ? lane : 0;
      double g_cur = G[0][col], t_cur = tot[0];  // the LDS operands of link j + 1 are requested while link j is summed
      for (int j = 0; j < dim; ++j) {
        const int jn = j + 1 < dim ? j + 1 : j;
        const double g_next = G[jn][col], t_next = tot[jn];
        // s = sum_{i<j} G_ji h_i over the lanes i < j
        double part = (lane < j) ? g_cur * hj : 0.0;
        part = gx_wave_sum(part);
        if (lane == j) hj = t_cur - part;
        g_cur = g_next;
        t_cur = t_next;
      }
      if (lane < dim) hc[lane] = hj;
      // |w'|^2 = |w|^2 - 2 h.r + h^T G h   (G symmetric: row lane against all columns)
      double quad = 0.0;
      if (lane < dim) {
        double row = 0.0;
        for (int i = 0; i < dim; ++i) row += (i <= lane ? G[lane][i] : G[i][lane]) * __shfl(hj, i, 64);
        quad = hj * (row - 2.0 * tot[lane]);
      } else {
        for (int i = 0; i < dim; ++i) (void)__shfl(hj, i, 64);
      }
      quad = gx_wave_sum(quad);
      if (lane == 0) s_norm2 = tot[2 * dim] + quad;
    }
    __syncthreads();
    MGS_STAMP();  // coefficients solved
    // w += (-h_j) v_j, j ascending: the streamed (older) vectors first, then the kept block.  The entries of vector j + 1 are
    // requested before those of vector j are used (one vector ahead; the first one in front of the coefficient solve): taken one
    // after the other every older vector cost a full trip through memory, 4 us each at dim 14 (profiles/r03_mgs_one_timeline.txt).
    // Same operations on the same operands in the same order.
    for (int j = 0; j < j_keep; ++j) {
      double xn[E];
#pragma unroll
      for (int k = 0; k < E; ++k) xn[k] = (j + 1 < j_keep && idx[k] >= 0) ? ld_twice(V.v[j + 1] + idx[k]) : 0.0;
      const double alpha = -1.0 * hc[j];
#pragma unroll
      for (int k = 0; k < E; ++k)
        if (idx[k] >= 0) wv[k] += alpha * xo[k];
#pragma unroll
      for (int k = 0; k < E; ++k) xo[k] = xn[k];
    }
#pragma unroll
    for (int i = 0; i < DMAX; ++i)
      if (j_keep + i < dim) {
        const double alpha = -1.0 * hc[j_keep + i];
#pragma unroll
        for (int k = 0; k < E; ++k) wv[k] += alpha * vb[i][k];
      }
    // the formula is a difference of numbers of size |w|^2: refuse it when less than 1 % of the norm is left
    double norm2 = s_norm2;
    const double w2 = tot[2 * dim];
    bool norm_pending = false;
    if (!(norm2 > norm_guard * w2)) {  // uniform over the grid (same totals everywhere): a second exchange sums |w'|^2 itself
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < E; ++k) a += wv[k] * wv[k];
      const double s_ = gx_wave_sum(a);
      __syncthreads();
      if (lane == 0) sh[wave][0] = s_;
      __syncthreads();
      const int v = 2 * dim + 1;
      if (threadIdx.x == 0 && wg != drop_wg) gx_post(box + (size_t)v * nwg + wg, (sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0]));
      if (wg == v % nwg) {
        double b = 0.0;
        for (int q = threadIdx.x; q < nwg; q += 256) b += gx_wait_value(box + (size_t)v * nwg + q, &lerr);
        if (lerr) s_err = 1;
        __syncthreads();
        const double sb = gx_wave_sum(b);
        if (lane == 0) sh[wave][1] = sb;
        __syncthreads();
        if constexpr (DIST) {
          // the local sum for the host's collective; a timed-out mailbox makes it NaN, which the host turns into the two-pass redo
          if (threadIdx.x == 0) *ext.norm_out = s_err ? __longlong_as_double(0x7ff8000000000000ll) : (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]);
        } else {
          if (threadIdx.x == 0 && !s_err) gx_post(total + v, (sh[0][1] + sh[1][1]) + (sh[2][1] + sh[3][1]));
        }
      }
      if constexpr (DIST) {
        norm_pending = true;  // nobody waits: the sum travels through the host's all-reduce behind this launch
      } else {
        if (threadIdx.x == 0) {
          const double x_ = gx_wait_value(total + v, &lerr);
          if (lerr) s_err = 1;
          s_norm2 = x_;
        }
        __syncthreads();
        dead = s_err != 0;
        norm2 = s_norm2;
      }
    }
    if (!dead && normalize && !norm_pending) {  // vv *= 1. / s with s = sqrt(|vv|^2), skipped for s == 0 (SolverGMRES)
      const double nrm = sqrt(norm2);
      const bool second_sweep = consider && !(nrm > 10. * sqrt(w2) * 1.4901161193847656e-08);
      if (nrm != 0.0 && !second_sweep) {
        const double inv = 1. / nrm;
#pragma unroll
        for (int k = 0; k < E; ++k) wv[k] = inv * wv[k];
      }
    }
  }
  if (dead) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (wg == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (wg == 0) {
    // the new Gram row for the later sweeps of this cycle, the coefficients for the device and the host
    if ((int)threadIdx.x < dim) {
      gram[(dim - 1) * 32 + threadIdx.x] = tot[dim + threadIdx.x];
      scal_out[threadIdx.x] = hc[threadIdx.x];
      __hip_atomic_store(pub_vals + threadIdx.x, hc[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
      scal_out[dim] = s_norm2;
      scal_out[dim + 1] = tot[2 * dim];
      __hip_atomic_store(pub_vals + dim, s_norm2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(pub_vals + dim + 1, tot[2 * dim], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if constexpr (DIST) {  // 1: |w'|^2 is still rank-local (ext.norm_out) and w is not normalised
        const double norm2_ = s_norm2, w2_ = tot[2 * dim];
        __hip_atomic_store(pub_vals + dim + 2, !(norm2_ > norm_guard * w2_) ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (threadIdx.x == 0) tail[wg] = seq;  // this workgroup commits its part of w
  MGS_STAMP();  // update done
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (idx[k] >= 0) w[idx[k]] = wv[k];
  MGS_STAMP();  // stores issued
}

// ---- the triangular solves of the preconditioner AND the sweep in ONE launch (round 5) ------------------------------------------
// An inner GMRES iteration on F is  p = F v_k (SpMV)  ->  z = (LU)^-1 p (k_ilu_solve_lanes: one wave per ~8 rank blocks, their rows
// in LDS)  ->  sweep of z against the basis (k_mgs_one).  The last two hand z to each other through HBM, and while the sweeping wave
// of the solve works through its ~120 ticks the device moves nothing, while the sweep then spends its first 11-17 us loading basis
// vectors with every wave stalled.  Here workgroup b of the persistent grid IS wave b of the solve's schedule (same stream, same ticks,
// same arithmetic: z is bit for bit the separate kernel's): its four waves load the right-hand side rows into LDS, wave 0 runs the two
// sweeps while waves 1-3 request their entries of the basis vectors, and after a barrier every thread takes its entries of z from
// LDS -- z never travels through memory -- and the kernel goes on as k_mgs_one: one grid exchange, the coefficients from the Gram
// matrix, update, norm, normalisation.  Entry -> thread: entry e of the workgroup's rows (LDS order) belongs to thread e % 256, so a
// workgroup needs at most 256 * E / NCOMP rows (853 with E = 10; the bench layout's largest wave has 747).  Sums are fixed-order but
// taken in another grouping than k_mgs_one's (entries are dealt by rank block, not striped over the vector): same history class, other
// last bits.  No static __shared__ object: the solve's stream holds absolute LDS addresses (nsx_ilu_lanes.hpp), its rows sit at address 0.
struct IluMgsArgs {
  const int32_t *row_ptr, *rows, *slab_ptr;
  const uint32_t *meta;
  const double *val, *dinv, *rhs;
  int ilu_doubles;  // LDS doubles reserved for the solve's rows (+ 64 scratch rows), the sweep's arrays follow
  unsigned long long *trace;  // development (NSX_ILU_MGS_TRACE): 16 wall-clock stamps per wave, or null
};
#define IM_STAMP(k)                                                                                                   \
  do {                                                                                                                \
    if (I.trace && lane == 0) I.trace[((size_t)wg * 4 + wave) * 16 + (k)] = wall_clock64();                             \
  } while (0)
template <int NCOMP, int EI, int PF, int E, int DMAX>
__global__ __launch_bounds__(256) void k_ilu_mgs(int n, double *__restrict__ w, MgsArgs V, int dim, double *__restrict__ gram, unsigned long long *box,
                                                 unsigned long long *box_next, int reset_words, double *__restrict__ scal_out, int *err_host, unsigned long long *tail,
                                                 int normalize, int consider, double *pub_vals, unsigned long long *pub_flag, unsigned long long seq, int drop_wg,
                                                 double norm_guard, IluMgsArgs I) {
  extern __shared__ double xs[];
  double *sh = xs + I.ilu_doubles;                 // [4][MGS_ONE_VALS]
  double *tot = sh + 4 * MGS_ONE_VALS;             // [MGS_ONE_VALS]
  double *G = tot + MGS_ONE_VALS;                  // [MGS_STEPS][MGS_STEPS + 1]
  double *hc = G + MGS_STEPS * (MGS_STEPS + 1);    // [MGS_STEPS]
  double *s_norm2 = hc + MGS_STEPS;
  int *s_err = (int *)(s_norm2 + 1);
  double *stage = s_norm2 + 2;                     // [DMAX][E][64]: the sweeping wave's entries of the basis
  const int nwg = gridDim.x, wg = blockIdx.x, T = nwg * 256, t = wg * 256 + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool lds_ok = (uint32_t)(uintptr_t)(lds_f64 *)xs == 0u;  // uniform over the grid: everybody leaves, nobody waits
  if (!lds_ok) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (wg == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (threadIdx.x == 0) *s_err = 0;
  IM_STAMP(0);  // start
  unsigned long long *total = box + (size_t)MGS_ONE_VALS * MGS_MAX_WG, *total_next = box_next + (size_t)MGS_ONE_VALS * MGS_MAX_WG;
  for (int q = t; q < reset_words; q += T) box_next[q] = GX_EMPTY;
  if (wg == 0 && threadIdx.x < MGS_ONE_VALS) total_next[threadIdx.x] = GX_EMPTY;
  const int nvals = 2 * dim + 1;
  // ---- the workgroup's rows: wave `wg` of the solve's schedule
  const int rb = I.row_ptr[wg], nr = I.row_ptr[wg + 1] - rb, ne = nr * NCOMP;
  const int s0 = I.slab_ptr[2 * wg], s1 = I.slab_ptr[2 * wg + 1], s2 = I.slab_ptr[2 * wg + 2];
  int idx[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const int e = (int)threadIdx.x + 256 * k;
    const int r_ = e < ne ? I.rows[rb + e / NCOMP] : -1;
    idx[k] = r_ >= 0 ? r_ * NCOMP + e % NCOMP : -1;
  }
  {
    double y[E];
#pragma unroll
    for (int k = 0; k < E; ++k) y[k] = idx[k] >= 0 ? I.rhs[idx[k]] : 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k)
      if (idx[k] >= 0) xs[(int)threadIdx.x + 256 * k] = y[k];
  }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) xs[(nr + lane) * NCOMP + c] = 0.0;  // the scratch rows of the idle slots
  }
  __syncthreads();
  IM_STAMP(1);  // right-hand side rows in LDS
  // which wave sweeps.  (Giving the two workgroups of a CU different sweeping waves -- different SIMDs -- measured nothing; neither did
  // delaying the other waves' basis requests by 4 - 12 us or pacing them one vector per 0.5 - 1 us: 45.9 - 46.8 us per launch throughout.)
  constexpr int iw = 0;
  if (wave == iw) {  // the two sweeps of the triangular solve, exactly k_ilu_solve_lanes'
    const uint32_t scratch = (uint32_t)(nr + lane) * (8u * NCOMP);
    LaneSlot<EI> A[PF];
    lane_load<EI, PF>(A, s0, I.meta, I.val, (unsigned)lane);
    lane_sweep<NCOMP, EI, PF>(A, s0, s1, I.meta, I.val, (unsigned)lane, scratch);  // y = L^{-1} b
    IM_STAMP(2);  // forward sweep done
    lane_load<EI, PF>(A, s1, I.meta, I.val, (unsigned)lane);
    for (int base = 0; base < nr; base += 64 * 4) {  // y *= D^{-1}
      double d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = base + 64 * k + lane;
        d[k] = q < nr ? I.dinv[rb + q] : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int q = base + 64 * k + lane;
        if (q < nr) {
#pragma unroll
          for (int c = 0; c < NCOMP; ++c) xs[q * NCOMP + c] *= d[k];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    lane_sweep<NCOMP, EI, PF>(A, s1, s2, I.meta, I.val, (unsigned)lane, scratch);  // x = U^{-1} y
    IM_STAMP(3);  // backward sweep done
  }
  // ---- the basis.  The three waves that do not sweep get here at once: their requests fly while the fourth sweeps.  The sweeping
  // wave's OWN entries of the basis would be requested behind its sweeps and arrive 6 - 7 us later with the whole workgroup waiting at
  // the barrier (profiles/r05_ilu_mgs_timeline.txt): the other three fetch them as well, into LDS (`stage`, [vector][k][lane]).
  double wv[E], vb[DMAX][E];
  const int j_keep = dim > DMAX ? dim - DMAX : 0;
  if (wave != iw) {
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
      const double *__restrict__ vp = j_keep + i < dim ? V.v[j_keep + i] : nullptr;
#pragma unroll
      for (int k = 0; k < E; ++k) vb[i][k] = (vp && idx[k] >= 0) ? ld_stream<1>(vp + idx[k]) : 0.0;
    }
    constexpr int HB = (64 * E + 191) / 192;
    const int hid = (wave < iw ? wave : wave - 1) * 64 + lane;  // 0 .. 191
    int gi[HB];
#pragma unroll
    for (int m = 0; m < HB; ++m) {
      const int q = hid + 192 * m, e = 64 * iw + (q & 63) + 256 * (q >> 6);
      const int r_ = (q < 64 * E && e < ne) ? I.rows[rb + e / NCOMP] : -1;
      gi[m] = r_ >= 0 ? r_ * NCOMP + e % NCOMP : -1;
    }
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
      const double *__restrict__ vp = j_keep + i < dim ? V.v[j_keep + i] : nullptr;
      double tmp[HB];
#pragma unroll
      for (int m = 0; m < HB; ++m) tmp[m] = (vp && gi[m] >= 0) ? ld_stream<1>(vp + gi[m]) : 0.0;
#pragma unroll
      for (int m = 0; m < HB; ++m)
        if (hid + 192 * m < 64 * E) stage[i * (64 * E) + hid + 192 * m] = tmp[m];
    }
  }
  if (I.trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IM_STAMP(4);  // this wave's requests have arrived
  }
  __syncthreads();  // z is complete in LDS, and so is the sweeping wave's part of the basis
  IM_STAMP(5);
  if (wave == iw) {
#pragma unroll
    for (int i = 0; i < DMAX; ++i)
#pragma unroll
      for (int k = 0; k < E; ++k) vb[i][k] = stage[(i * E + k) * 64 + lane];
  }
#pragma unroll
  for (int k = 0; k < E; ++k) wv[k] = idx[k] >= 0 ? xs[(int)threadIdx.x + 256 * k] : 0.0;
  double vl[E];
#pragma unroll
  for (int k = 0; k < E; ++k) vl[k] = 0.0;
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (j_keep + i == dim - 1) {
#pragma unroll
      for (int k = 0; k < E; ++k) vl[k] = vb[i][k];
    }
  auto wave_post = [&](int v, double a) {
    const double s_ = gx_wave_sum(a);
    if (lane == 0) sh[wave * MGS_ONE_VALS + v] = s_;
  };
  {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) a += wv[k] * wv[k];
    wave_post(2 * dim, a);
  }
  for (int j = 0; j < j_keep; ++j) {  // older vectors: streamed, not kept (dim > DMAX only)
    const double *__restrict__ vp = V.v[j];
    double ar = 0.0, ag = 0.0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const double x_ = idx[k] >= 0 ? ld_twice(vp + idx[k]) : 0.0;
      ar += wv[k] * x_;
      ag += vl[k] * x_;
    }
    wave_post(j, ar);
    wave_post(dim + j, ag);
  }
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (j_keep + i < dim) {
      double ar = 0.0, ag = 0.0;
#pragma unroll
      for (int k = 0; k < E; ++k) {
        ar += wv[k] * vb[i][k];
        ag += vl[k] * vb[i][k];
      }
      wave_post(j_keep + i, ar);
      wave_post(dim + j_keep + i, ag);
    }
  __syncthreads();
  IM_STAMP(6);  // local sums done
  // ---- hop 1: mailboxes; value v is summed by workgroup v % nwg
  int lerr = 0;
  for (int v = threadIdx.x; v < nvals; v += 256)
    if (wg != drop_wg) gx_post(box + (size_t)v * nwg + wg, (sh[v] + sh[MGS_ONE_VALS + v]) + (sh[2 * MGS_ONE_VALS + v] + sh[3 * MGS_ONE_VALS + v]));
  for (int v = wg; v < nvals; v += nwg) {
    double a = 0.0;
    for (int q = threadIdx.x; q < nwg; q += 256) a += gx_wait_value(box + (size_t)v * nwg + q, &lerr);
    if (lerr) *s_err = 1;
    __syncthreads();
    const double s_ = gx_wave_sum(a);
    if (lane == 0) sh[wave * MGS_ONE_VALS] = s_;
    __syncthreads();
    if (threadIdx.x == 0 && !*s_err) gx_post(total + v, (sh[0] + sh[MGS_ONE_VALS]) + (sh[2 * MGS_ONE_VALS] + sh[3 * MGS_ONE_VALS]));
    __syncthreads();
  }
  // ---- hop 2: everybody picks up the totals
  for (int v = threadIdx.x; v < nvals; v += 256) {
    tot[v] = gx_wait_value(total + v, &lerr);
    if (lerr) *s_err = 1;
  }
  __syncthreads();
  IM_STAMP(7);  // totals picked up
  bool dead = *s_err != 0;
  double xo[E];
#pragma unroll
  for (int k = 0; k < E; ++k) xo[k] = (!dead && j_keep > 0 && idx[k] >= 0) ? ld_twice(V.v[0] + idx[k]) : 0.0;
  if (!dead) {
    for (int q = threadIdx.x; q < (dim - 1) * MGS_STEPS; q += 256) {
      const int r_ = q / MGS_STEPS, c_ = q % MGS_STEPS;
      if (c_ <= r_) G[r_ * (MGS_STEPS + 1) + c_] = gram[r_ * 32 + c_];
    }
    if ((int)threadIdx.x < dim) G[(dim - 1) * (MGS_STEPS + 1) + threadIdx.x] = tot[dim + threadIdx.x];
    __syncthreads();
    if (wave == 0) {
      double hj = 0.0;
      const int col = lane < dim ? lane : 0;
      double g_cur = G[col], t_cur = tot[0];
      for (int j = 0; j < dim; ++j) {
        const int jn = j + 1 < dim ? j + 1 : j;
        const double g_next = G[jn * (MGS_STEPS + 1) + col], t_next = tot[jn];
        double part = (lane < j) ? g_cur * hj : 0.0;
        part = gx_wave_sum(part);
        if (lane == j) hj = t_cur - part;
        g_cur = g_next;
        t_cur = t_next;
      }
      if (lane < dim) hc[lane] = hj;
      double quad = 0.0;
      if (lane < dim) {
        double row = 0.0;
        for (int i = 0; i < dim; ++i) row += (i <= lane ? G[lane * (MGS_STEPS + 1) + i] : G[i * (MGS_STEPS + 1) + lane]) * __shfl(hj, i, 64);
        quad = hj * (row - 2.0 * tot[lane]);
      } else {
        for (int i = 0; i < dim; ++i) (void)__shfl(hj, i, 64);
      }
      quad = gx_wave_sum(quad);
      if (lane == 0) *s_norm2 = tot[2 * dim] + quad;
    }
    __syncthreads();
    for (int j = 0; j < j_keep; ++j) {
      double xn[E];
#pragma unroll
      for (int k = 0; k < E; ++k) xn[k] = (j + 1 < j_keep && idx[k] >= 0) ? ld_twice(V.v[j + 1] + idx[k]) : 0.0;
      const double alpha = -1.0 * hc[j];
#pragma unroll
      for (int k = 0; k < E; ++k)
        if (idx[k] >= 0) wv[k] += alpha * xo[k];
#pragma unroll
      for (int k = 0; k < E; ++k) xo[k] = xn[k];
    }
#pragma unroll
    for (int i = 0; i < DMAX; ++i)
      if (j_keep + i < dim) {
        const double alpha = -1.0 * hc[j_keep + i];
#pragma unroll
        for (int k = 0; k < E; ++k) wv[k] += alpha * vb[i][k];
      }
    double norm2 = *s_norm2;
    const double w2 = tot[2 * dim];
    if (!(norm2 > norm_guard * w2)) {  // uniform over the grid: a second exchange sums |w'|^2 itself
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < E; ++k) a += wv[k] * wv[k];
      const double s_ = gx_wave_sum(a);
      __syncthreads();
      if (lane == 0) sh[wave * MGS_ONE_VALS] = s_;
      __syncthreads();
      const int v = 2 * dim + 1;
      if (threadIdx.x == 0 && wg != drop_wg) gx_post(box + (size_t)v * nwg + wg, (sh[0] + sh[MGS_ONE_VALS]) + (sh[2 * MGS_ONE_VALS] + sh[3 * MGS_ONE_VALS]));
      if (wg == v % nwg) {
        double b = 0.0;
        for (int q = threadIdx.x; q < nwg; q += 256) b += gx_wait_value(box + (size_t)v * nwg + q, &lerr);
        if (lerr) *s_err = 1;
        __syncthreads();
        const double sb = gx_wave_sum(b);
        if (lane == 0) sh[wave * MGS_ONE_VALS + 1] = sb;
        __syncthreads();
        if (threadIdx.x == 0 && !*s_err) gx_post(total + v, (sh[1] + sh[MGS_ONE_VALS + 1]) + (sh[2 * MGS_ONE_VALS + 1] + sh[3 * MGS_ONE_VALS + 1]));
      }
      if (threadIdx.x == 0) {
        const double x_ = gx_wait_value(total + v, &lerr);
        if (lerr) *s_err = 1;
        *s_norm2 = x_;
      }
      __syncthreads();
      dead = *s_err != 0;
      norm2 = *s_norm2;
    }
    if (!dead && normalize) {
      const double nrm = sqrt(norm2);
      const bool second_sweep = consider && !(nrm > 10. * sqrt(w2) * 1.4901161193847656e-08);
      if (nrm != 0.0 && !second_sweep) {
        const double inv = 1. / nrm;
#pragma unroll
        for (int k = 0; k < E; ++k) wv[k] = inv * wv[k];
      }
    }
  }
  if (dead) {
    if (threadIdx.x == 0) {
      __hip_atomic_store(err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (wg == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (wg == 0) {
    if ((int)threadIdx.x < dim) {
      gram[(dim - 1) * 32 + threadIdx.x] = tot[dim + threadIdx.x];
      scal_out[threadIdx.x] = hc[threadIdx.x];
      __hip_atomic_store(pub_vals + threadIdx.x, hc[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) {
      scal_out[dim] = *s_norm2;
      scal_out[dim + 1] = tot[2 * dim];
      __hip_atomic_store(pub_vals + dim, *s_norm2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(pub_vals + dim + 1, tot[2 * dim], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  IM_STAMP(8);  // coefficients, update, norm done
  if (threadIdx.x == 0) tail[wg] = seq;
#pragma unroll
  for (int k = 0; k < E; ++k)
    if (idx[k] >= 0) w[idx[k]] = wv[k];
}
// LDS of the fused kernel: the solve's rows + scratch rows, then sh, tot, G, hc, |w'|^2, the error word
static size_t ilu_mgs_lds_doubles(int ilu_doubles, int e) {
  const int dmax = e <= 8 ? 10 : e == 9 ? 9 : 8;
  return (size_t)ilu_doubles + 4 * MGS_ONE_VALS + MGS_ONE_VALS + MGS_STEPS * (MGS_STEPS + 1) + MGS_STEPS + 2 + (size_t)dmax * e * 64;
}

// entries per thread x basis vectors kept in registers: 8 x 10, 10 x 8, 12 x 6 (round 4: 1.28 M velocity dofs per GPU -- the 10.6 M-DoF mesh
// on 8 GPUs -- need 11.1 entries per thread of the 448-workgroup grid a distributed sweep may use; one GPU: vectors up to 1.57 M entries)
static const void *mgs_one_fn(int e, bool dist = false) {
  if (dist) return e <= 8 ? (const void *)k_mgs_one<8, 10, true> : e <= 10 ? (const void *)k_mgs_one<10, 8, true> : (const void *)k_mgs_one<12, 6, true>;
  return e <= 8 ? (const void *)k_mgs_one<8, 10, false> : e <= 10 ? (const void *)k_mgs_one<10, 8, false> : (const void *)k_mgs_one<12, 6, false>;
}

template <int M>
static const void *mgs_blk_fn(int e) {
  return e <= 8 ? (const void *)k_mgs_blk<8, M, true> : e <= 10 ? (const void *)k_mgs_blk<10, M, true> : (const void *)k_mgs_blk<20, M, false>;
}
static const void *mgs_fn(int m, int e) {
  switch (m) {
    case 0: return mgs_one_fn(e);
    case 2: return mgs_blk_fn<2>(e);
    case 3: return mgs_blk_fn<3>(e);
    case 4: return mgs_blk_fn<4>(e);
    case 5: return mgs_blk_fn<5>(e);
    default: return e <= 10 ? (const void *)k_mgs<10> : (const void *)k_mgs<20>;
  }
}

static void mgs_setup(nsx_handle *h) {
  if (h->mgs_box.p || h->mgs_disabled) return;
  h->mgs_max_wg = 0;
  if (getenv("NSX_MGS") && atoi(getenv("NSX_MGS")) == 0) {
    h->mgs_disabled = true;
    return;
  }
  // links per exchange: 0 = all of them (k_mgs_one, the default), 1 = deal.II's chain link by link (k_mgs), 2..5 = k_mgs_blk
  h->mgs_links = getenv("NSX_MGS_LINKS") ? std::max(0, std::min(5, atoi(getenv("NSX_MGS_LINKS")))) : 0;
  int cus = 0;
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->prm.device));
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  h->mgs_box.alloc(2 * region + MGS_TAIL);
  HIP_CHECK(hipMemsetAsync(h->mgs_box.p, 0xff, 2 * region * sizeof(unsigned long long), h->stream));
  HIP_CHECK(hipMemsetAsync(h->mgs_box.p + 2 * region, 0, MGS_TAIL * sizeof(unsigned long long), h->stream));
  const int es[3] = {8, 10, h->mgs_links == 0 ? 12 : 20};
  for (int k = 0; k < 3; ++k) {
    int per_cu = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mgs_fn(h->mgs_links, es[k]), 256, 0));
    h->mgs_max_wg_e[k] = std::min(MGS_MAX_WG, per_cu * cus);
    if (getenv("NSX_MGS_MAXWG")) h->mgs_max_wg_e[k] = std::max(1, std::min(h->mgs_max_wg_e[k], atoi(getenv("NSX_MGS_MAXWG"))));
    if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] mgs sweep (%d links per exchange, %d entries per thread): %d CUs x %d resident workgroups\n", h->mgs_links, es[k], cus, per_cu);
  }
  h->mgs_max_wg = h->mgs_max_wg_e[1];
  // distributed instantiations: the collective's own kernels (RCCL's all-reduce, the two one-thread kernels around it) must find a
  // place on the device WHILE the grid is resident and waiting for them: an eighth of the slots (at least 32) stays free
  const int es_one[3] = {8, 10, 12};
  for (int k = 0; k < 3; ++k) {
    int per_cu = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mgs_one_fn(es_one[k], true), 256, 0));
    const int slots = per_cu * cus;
    h->mgs_max_wg_dist[k] = std::max(0, std::min(MGS_MAX_WG, slots - std::max(32, slots / 8)));
    // on a compute stream that leaves one CU per XCD to the collective's kernel: every shader engine counts as the one that lost a CU
    h->mgs_dist_cap_reserved[k] = std::max(0, std::min(MGS_MAX_WG, per_cu * (cus - 32)));
    if (getenv("NSX_MGS_MAXWG")) {  // (tests: a small mesh then needs the instantiations a large one does)
      h->mgs_max_wg_dist[k] = std::max(1, std::min(h->mgs_max_wg_dist[k], atoi(getenv("NSX_MGS_MAXWG"))));
      h->mgs_dist_cap_reserved[k] = std::max(1, std::min(h->mgs_dist_cap_reserved[k], atoi(getenv("NSX_MGS_MAXWG"))));
    }
  }
  h->mgs_ext_vals.alloc(2 * MGS_EXT_VALS);
  h->mgs_ext_vals.zero(h->stream);
  h->mgs_ext_words.alloc(3);
  h->mgs_ext_words.zero(h->stream);
  h->mgs_ext_expected = 0;
}

void wait_published(nsx_handle *h, unsigned long long seq) {
  volatile unsigned long long *flag_host = (volatile unsigned long long *)(h->pub_host + N_SLOTS);
  unsigned long long spins = 0;
  // sequence numbers only grow: a later publication that has already landed also proves this one did
  while (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) < seq) {
    if (++spins > 200000000ull) {  // bounded: fall back to a stream synchronisation, which also surfaces launch errors
      HIP_CHECK(hipStreamSynchronize(h->stream));
      if (__atomic_load_n(flag_host, __ATOMIC_ACQUIRE) < seq) NSX_THROW(NSX_ERR_HIP, "scalar publication never arrived");
      break;
    }
  }
}

// A persistent sweep ended on a timeout (its grid was not co-resident).  Put the handle back into a usable state: wait for the
// stragglers, empty the mailboxes, clear the error words and use the launch-per-link chain from now on.  Returns how many
// workgroups had already written their part of w (0: w is untouched and the sweep can simply be redone by the chain).
static unsigned int mgs_recover(nsx_handle *h, unsigned long long failed_seq) {
  HIP_CHECK(hipStreamSynchronize(h->stream));
  if (h->comm_stream) HIP_CHECK(hipStreamSynchronize(h->comm_stream));
  if (h->mgs_ext_vals.p) {
    h->mgs_ext_vals.zero(h->stream);
    h->mgs_ext_words.zero(h->stream);
    h->mgs_ext_expected = 0;
  }
  h->mgs_max_wg_dist[0] = h->mgs_max_wg_dist[1] = h->mgs_max_wg_dist[2] = 0;
  h->mgs_dist_fit.clear();
  std::vector<unsigned long long> tail(MGS_TAIL, 0);
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  HIP_CHECK(hipMemcpy(tail.data(), h->mgs_box.p + 2 * region, MGS_TAIL * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_CHECK(hipMemsetAsync(h->mgs_box.p, 0xff, 2 * region * sizeof(unsigned long long), h->stream));
  HIP_CHECK(hipMemsetAsync(h->mgs_box.p + 2 * region, 0, MGS_TAIL * sizeof(unsigned long long), h->stream));
  *(volatile int *)(h->pub_host + N_SLOTS + 2) = 0;
  h->mgs_used_wg[0] = h->mgs_used_wg[1] = h->mgs_used_steps[0] = h->mgs_used_steps[1] = 0;
  h->mgs_max_wg = h->mgs_max_wg_e[0] = h->mgs_max_wg_e[1] = h->mgs_max_wg_e[2] = 0;
  h->mgs_disabled = true;
  h->n_persistent_fallbacks++;
  unsigned int committed = 0;
  for (unsigned long long v : tail) committed += v == failed_seq;
  fprintf(stderr, "[nsx] warning: the persistent Gram-Schmidt sweep timed out (grid not co-resident): this handle uses one launch per link from now on\n");
  return committed;
}

static void mgs_chain(nsx_handle *h, Span sp, double *w, int dim, double *const *vs, int slot0, double *out, bool consider) {
  if (consider) v_dot(h, sp, w, w, slot0 + dim + 1);
  v_dot(h, sp, w, vs[0], slot0);
  for (int i = 1; i < dim; ++i) v_add_and_dot(h, sp, w, -1.0, slot0 + i - 1, vs[i - 1], vs[i], slot0 + i);
  v_add_and_dot(h, sp, w, -1.0, slot0 + dim - 1, vs[dim - 1], w, slot0 + dim);
  read_scalars(h, slot0, dim + 1 + (consider ? 1 : 0), out);
}

// ---- the sweep with TWO collectives (distributed runs) ------------------------------------------------------------------
// With a communicator every link of the chain is a launch plus an all-reduce (dim + 1 collectives per sweep, the reference
// pays one MPI_Allreduce per link as well).  The same linearity that k_mgs_blk uses for M links holds for all of them:
//     h_j = v_j . w - sum_{i < j} (v_i . v_j) h_i ,
// so one pass computes r_j = v_j . w for every j and the new row of the basis' Gram matrix (v_{dim-1} . v_i, i < dim - 1; the
// older rows were computed by the earlier sweeps of this GMRES cycle and are kept on the device), ONE all-reduce sums them over
// the ranks, every rank solves the same unit lower-triangular system, and a second pass applies w += (-h_j) v_j for j ascending
// (the chain's operations on every entry, in the chain's order) and leaves the partial sums of |w|^2 for the second all-reduce.
// No orthogonality of the basis is assumed: in exact arithmetic the coefficients ARE the chain's.
constexpr int LS_C = 8;       // basis vectors per pass of the dot kernel
constexpr int LS_VALS = 64;   // r_j at j, Gram row at 32 + i, |w|^2 before the sweep at 63
constexpr int LS_BLOCKS = 2048;  // most workgroups of the dot kernel (partial sums per value)
constexpr int N_TMP_MAX = 32;
enum { S_LS_NORM = 7 };          // scalar slot of the explicit |w|^2 of the distributed sweep (free in nsx_solve.hip's table)

__global__ __launch_bounds__(256) void k_ls_dots(int n, int split, int gap, const double *__restrict__ w, MgsArgs V, int dim, double *__restrict__ partial) {
  __shared__ double sh[4][2 * LS_C + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double *__restrict__ vl = V.v[dim - 1];
  for (int c0 = 0; c0 < dim; c0 += LS_C) {
    double ar[LS_C], ag[LS_C], aw = 0.0;
#pragma unroll
    for (int k = 0; k < LS_C; ++k) ar[k] = ag[k] = 0.0;
    // (two entries per thread and pass, 2 x (LS_C + 2) loads in flight, measured SLOWER at 10.2 M entries: 228 against 214 us)
    for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += gridDim.x * 256) {
      const int i = i0 + (i0 >= split ? gap : 0);
      const double wi = w[i], li = vl[i];
#pragma unroll
      for (int k = 0; k < LS_C; ++k)
        if (c0 + k < dim) {
          const double vk = V.v[c0 + k][i];
          ar[k] += wi * vk;
          ag[k] += li * vk;
        }
      aw += wi * wi;
    }
#pragma unroll
    for (int k = 0; k < LS_C; ++k) {
      const double a = gx_wave_sum(ar[k]), b = gx_wave_sum(ag[k]);
      if (lane == 0) sh[wave][k] = a, sh[wave][LS_C + k] = b;
    }
    {
      const double a = gx_wave_sum(aw);
      if (lane == 0) sh[wave][2 * LS_C] = a;
    }
    __syncthreads();
    if (threadIdx.x < 2 * LS_C + 1) {
      const int q = threadIdx.x;
      const double tot = (sh[0][q] + sh[1][q]) + (sh[2][q] + sh[3][q]);
      const int j = c0 + (q < LS_C ? q : q - LS_C);
      if (q < LS_C) {
        if (j < dim) partial[(size_t)j * LS_BLOCKS + blockIdx.x] = tot;
      } else if (q < 2 * LS_C) {
        if (j < dim) partial[(size_t)(32 + j) * LS_BLOCKS + blockIdx.x] = tot;  // j = dim - 1: the diagonal |v_{dim-1}|^2
      } else if (c0 == 0) {
        partial[(size_t)63 * LS_BLOCKS + blockIdx.x] = tot;
      }
    }
    __syncthreads();
  }
}
// vals[v] = fixed-order sum of the LS_BLOCKS partial sums of value v (one workgroup per value; unused values become 0)
__global__ __launch_bounds__(256) void k_ls_finalize(int dim, int nblk, const double *__restrict__ partial, double *__restrict__ vals) {
  __shared__ double sh[4];
  const int v = blockIdx.x;
  const bool used = v < dim || (v >= 32 && v < 32 + dim) || v == 63;
  double a = 0.0;
  if (used)
    for (int q = threadIdx.x; q < nblk; q += 256) a += partial[(size_t)v * LS_BLOCKS + q];
  const double t = gx_block_sum(a, sh);
  if (threadIdx.x == 0) vals[v] = t;
}
// every rank, from the same all-reduced numbers: the new Gram row (diagonal included), then h = (I + L)^-1 r by forward
// substitution, and |w|^2 AFTER the sweep without touching the vectors again:
//     |w - sum_j h_j v_j|^2 = |w|^2 - 2 sum_j h_j r_j + sum_ij h_i G_ij h_j        (G = full Gram matrix of the basis, r_j = v_j . w)
// -- exact algebra, no orthogonality assumed; in floating point a difference of numbers of size |w|^2, so its relative error is
// eps |w|^2 / |w_after|^2: the host takes it when the sweep left more than 1 % of the norm and otherwise pays the second collective
// (scal_out[dim] = |w_after|^2 by the formula, scal_out[dim + 1] = |w|^2 before the sweep).
__global__ __launch_bounds__(64) void k_ls_solve(int dim, const double *__restrict__ vals, double *__restrict__ gram, double *__restrict__ scal_out) {
  __shared__ double G[32][33], hc[32];
  const int t = threadIdx.x;
  for (int i = t; i < dim; i += 64) gram[(dim - 1) * 32 + i] = vals[32 + i];
  __syncthreads();
  for (int q = t; q < dim * 32; q += 64) G[q >> 5][q & 31] = (q & 31) <= (q >> 5) ? gram[q] : 0.0;
  __syncthreads();
  if (t == 0) {
    for (int j = 0; j < dim; ++j) {
      double s = vals[j];
      for (int i = 0; i < j; ++i) s -= G[j][i] * hc[i];
      hc[j] = s;
      scal_out[j] = s;
    }
    double cross = 0.0, quad = 0.0;
    for (int j = 0; j < dim; ++j) {
      cross += hc[j] * vals[j];
      double row = 0.5 * G[j][j] * hc[j];
      for (int i = 0; i < j; ++i) row += G[j][i] * hc[i];
      quad += hc[j] * row;  // half of the symmetric form
    }
    scal_out[dim] = vals[63] - 2.0 * cross + 2.0 * quad;
    scal_out[dim + 1] = vals[63];
  }
}
// w += (-h_j) v_j, j ascending; partial sums of |w|^2
__global__ __launch_bounds__(256) void k_ls_update(int n, int split, int gap, double *__restrict__ w, MgsArgs V, int dim, const double *__restrict__ coef,
                                                   double *__restrict__ partial) {
  __shared__ double hs[32], sh[5];
  if ((int)threadIdx.x < dim) hs[threadIdx.x] = -1.0 * coef[threadIdx.x];
  __syncthreads();
  double acc = 0.0;
  // four entries per thread and pass: the grid is at most 512 workgroups (one partial sum each), so at 10 M entries a thread walks ~80
  // of them, and taken one by one every basis vector was a dependent trip with a single load in flight (262 us per sweep at 10.2 M
  // entries, 3.1 TB/s).  Same operations on every entry in the same order, same order of the squares in the thread's sum: bit-identical.
  constexpr int U = 4;
  const int stride = gridDim.x * 256;
  int i0 = blockIdx.x * 256 + threadIdx.x;
  for (; i0 + (U - 1) * stride < n; i0 += U * stride) {
    int ii[U];
    double wi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = i0 + u * stride;
      ii[u] = q + (q >= split ? gap : 0);
      wi[u] = w[ii[u]];
    }
#pragma unroll 2
    for (int j = 0; j < dim; ++j) {
      const double *__restrict__ vj = V.v[j];
      double x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = vj[ii[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) wi[u] += hs[j] * x[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      w[ii[u]] = wi[u];
      acc += wi[u] * wi[u];
    }
  }
  for (; i0 < n; i0 += stride) {
    const int i = i0 + (i0 >= split ? gap : 0);
    double wi = w[i];
    for (int j = 0; j < dim; ++j) wi += hs[j] * V.v[j][i];
    w[i] = wi;
    acc += wi * wi;
  }
  const double t = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// |w'|^2 = |w|^2 - 2 h.r + h^T G h is a difference of numbers of size |w|^2: its relative error is about eps * dim * |w|^2 / |w'|^2.
// It is accepted when |w'|^2 > guard * |w|^2 (default 1e-2: the norm of the new basis vector is then good to ~1e-13, two orders
// below the tightest tolerance the parity tests solve to); otherwise |w'|^2 is summed over the vector (one more exchange / collective).
static double mgs_norm_guard(const nsx_handle *h) {
  static const double g = getenv("NSX_MGS_NORM_GUARD") ? atof(getenv("NSX_MGS_NORM_GUARD")) : 1e-2;
  return h->mgs_guard_override >= 0.0 ? h->mgs_guard_override : g;  // the override: nsx_gram_schmidt_cycle (tests)
}

static void mgs_lowsync(nsx_handle *h, Span sp, double *w, int dim, double *const *vs, int slot0, double *out, bool consider, double *gram) {
  if (!h->ls_partial.p) {
    h->ls_partial.alloc((size_t)LS_VALS * LS_BLOCKS);
    h->ls_vals.alloc(LS_VALS);
  }
  MgsArgs V;
  for (int i = 0; i < MGS_STEPS; ++i) V.v[i] = i < dim ? vs[i] : nullptr;
  const int n = sp.n;
  {
    LaunchScope ls(h, "mgs_dots", 8.0 * n * (dim + 2.0 * cdiv(dim, LS_C)));
    const int nblk = std::max(1, std::min(LS_BLOCKS, cdiv(n, 1024)));  // depends on the local size only: the values are final before they travel
    hipLaunchKernelGGL(k_ls_dots, dim3(nblk), dim3(256), 0, h->stream, n, sp.split, sp.gap, w, V, dim, h->ls_partial.p);
    hipLaunchKernelGGL(k_ls_finalize, dim3(LS_VALS), dim3(256), 0, h->stream, dim, nblk, h->ls_partial.p, h->ls_vals.p);
  }
  comm_allreduce_partials(h, h->ls_vals.p, LS_VALS);  // THE collective of the sweep: every r_j, the Gram row and |w|^2 before the sweep
  hipLaunchKernelGGL(k_ls_solve, dim3(1), dim3(64), 0, h->stream, dim, h->ls_vals.p, gram, h->scal.p + slot0);
  for (int i = 0; i <= dim + 1; ++i) h->slot_nb[slot0 + i] = 0;
  const int nb = red_blocks(h, n);
  {
    LaunchScope ls(h, "mgs_update", 8.0 * n * (dim + 2));
    // the update also leaves the partial sums of |w|^2 (slot S_LS_NORM), in case the formula cannot be trusted
    hipLaunchKernelGGL(k_ls_update, dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, w, V, dim, h->scal.p + slot0, red_out(h, S_LS_NORM, nb));
  }
  double tmp[N_TMP_MAX + 2];
  read_scalars(h, slot0, dim + 2, tmp);
  // one collective: |w_after|^2 from the Gram algebra, unless the sweep removed more than 99 % of the norm (cancellation)
  const bool by_formula = h->ls_mode >= 2 && tmp[dim] > mgs_norm_guard(h) * tmp[dim + 1];
  if (!by_formula) {
    after_reduction(h, S_LS_NORM, nb);  // collective 2: |w|^2 summed over the vector
    tmp[dim] = read_scalar(h, S_LS_NORM);
  }
  for (int i = 0; i <= dim; ++i) out[i] = tmp[i];
  if (consider) out[dim + 1] = tmp[dim + 1];
}

// the all-reduced sums of a persistent sweep (r_j at j, Gram row at dim + j, |w|^2 at 2 dim) in the layout of k_ls_solve
__global__ void k_ext_to_ls(int dim, const double *__restrict__ ext_vals, double *__restrict__ ls_vals) {
  const int t = threadIdx.x;  // 64 threads
  double v = 0.0;
  if (t < dim) v = ext_vals[t];
  else if (t >= 32 && t < 32 + dim) v = ext_vals[dim + (t - 32)];
  else if (t == 63) v = ext_vals[2 * dim];
  ls_vals[t] = v;
}

// which vector of the solve a sweep works on: the same answer on every rank, whatever its local sizes (0 velocity, 1 pressure,
// 2 block vector, 3 anything else) -- the key under which choices made by all ranks together are remembered
static int mgs_role(const nsx_handle *h, Span sp) { return sp.split < sp.n ? 2 : sp.n == h->n_u ? 0 : sp.n == h->n_p ? 1 : 3; }

// out[0..dim) = h(i), out[dim] = |w|^2 after the sweep.  Returns true when w was also normalised (only if asked to).
// fused kernel table: NCOMP x (E, DMAX)
static const void *ilu_mgs_fn(int ncomp, int e) {
  if (ncomp == 3) return e <= 8 ? (const void *)k_ilu_mgs<3, 2, 8, 8, 10> : e == 9 ? (const void *)k_ilu_mgs<3, 2, 8, 9, 9> : (const void *)k_ilu_mgs<3, 2, 8, 10, 8>;
  return e <= 8 ? (const void *)k_ilu_mgs<2, 2, 8, 8, 10> : e == 9 ? (const void *)k_ilu_mgs<2, 2, 8, 9, 9> : (const void *)k_ilu_mgs<2, 2, 8, 10, 8>;
}
// May the triangular solves of the velocity ILU(0) and the sweep behind them run as ONE launch (k_ilu_mgs)?  One GPU, the one-exchange
// sweep, the lane-owner stream with two entries per tick, a wave's rows within 256 x 8 or 256 x 10 entries, the grid resident.
// Returns the entries per thread (8 / 10) or 0.
static int ilu_mgs_entries(nsx_handle *h, Span sp, int dim, const double *gram) {
  // Opt-in (NSX_ILU_MGS=1; read per call: the tests switch it inside one process).  Measured at the bench size: 45.7 us per launch
  // against 26.7 + 25.7 for the two separate kernels, 3.05 against 3.18 ms per outer iteration (-4 %) -- and a re-rolled iteration
  // history (the sweep's sums are grouped by rank block): both sampled windows of the chaotic GMRES(28) sequence came out with MORE
  // restart steps (driver window 25.4 against 21.9 outer iterations per step, 325 steps 30.4 against 28.0), i.e. slower per time step.
  // The default therefore stays with the separate kernels and rounds 3-4's history.
  const bool wanted = getenv("NSX_ILU_MGS") && atoi(getenv("NSX_ILU_MGS")) == 1;
  const IluSchedule &s = h->schedF;
  if (!wanted || h->comm || h->mgs_disabled || !h->mgs_box.p || h->mgs_links != 0 || !gram || dim + 2 > MGS_STEPS) return 0;
  if (sp.n != h->n_u || sp.split != sp.n || sp.gap != 0 || (h->dim != 2 && h->dim != 3)) return 0;
  if (!s.packed_ok || s.levelled || s.stream_ncomp != h->dim || s.stream_epl != 2 || s.n_waves < 1 || s.n_waves > MGS_MAX_WG) return 0;
  if (getenv("NSX_PF") && atoi(getenv("NSX_PF")) != 8) return 0;
  // entries per thread x basis vectors kept in registers: 8 x 10, 9 x 9 (the bench layout: 747 rows in its largest wave), 10 x 8
  const int entries = s.max_wave_rows * h->dim, e = entries <= 256 * 8 ? 8 : entries <= 256 * 9 ? 9 : entries <= 256 * 10 ? 10 : 0;
  if (!e) return 0;
  const int k = e - 8;
  if (h->ilu_mgs_cap[k] < 0 || h->ilu_mgs_cap_rows != s.max_wave_rows) {  // resident-grid limit with this schedule's LDS request
    if (h->ilu_mgs_cap_rows != s.max_wave_rows) h->ilu_mgs_cap[0] = h->ilu_mgs_cap[1] = h->ilu_mgs_cap[2] = -1;
    h->ilu_mgs_cap_rows = s.max_wave_rows;
    int cus = 0, per_cu = 0;
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->prm.device));
    const size_t shm = ilu_mgs_lds_doubles((s.max_wave_rows + 64) * h->dim, e) * sizeof(double);
    if (shm > 80 * 1024) h->ilu_mgs_cap[k] = 0;
    else {
      if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute(ilu_mgs_fn(h->dim, e), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));  // (160 KB per CU on gfx950; the runtime's default limit per workgroup is 64 KB)
      HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ilu_mgs_fn(h->dim, e), 256, shm));
      h->ilu_mgs_cap[k] = std::min(MGS_MAX_WG, per_cu * cus);
      if (getenv("NSX_MGS_MAXWG")) h->ilu_mgs_cap[k] = std::min(h->ilu_mgs_cap[k], atoi(getenv("NSX_MGS_MAXWG")));
    }
    if (getenv("NSX_DEBUG"))
      fprintf(stderr, "[nsx] triangular solves + sweep in one launch (%d entries per thread, %zu B of LDS): %d resident workgroups for %d waves of the solve\n", e, shm,
              h->ilu_mgs_cap[k], s.n_waves);
  }
  return s.n_waves <= h->ilu_mgs_cap[k] ? e : 0;
}

bool v_mgs(nsx_handle *h, Span sp, double *w, int dim, double *const *vs, int slot0, bool normalize, double *out,
           const std::function<void()> *after_launch, bool consider, double *gram, const double *ilu_rhs) {
  const int n = sp.n;
  // ilu_rhs: w = (LU)^-1 ilu_rhs (the velocity ILU(0) of the last initialisation) comes FIRST -- inside the sweep's launch when
  // that is possible (k_ilu_mgs), as the separate kernel otherwise
  int fused_e = 0;
  if (ilu_rhs) {
    if (!h->comm) mgs_setup(h);
    fused_e = ilu_mgs_entries(h, sp, dim, gram);
    if (!fused_e) ilu_solve(h, h->gA, h->schedF, h->luF.p, ilu_rhs, w, h->dim, "ilu_solve_F");
  }
  // distributed run: the persistent sweep with the collective inside its exchange (k_mgs_one<.., true>) needs stream collectives
  // (RCCL), the Gram cache and room on the device; NSX_MGS_DIST=0 keeps the two-pass sweep (mgs_lowsync)
  if (h->comm && h->mgs_dist_state < 0) {  // decided once per handle, by all ranks together (comm_streams_concurrent)
    const bool wanted = !(getenv("NSX_MGS_DIST") && atoi(getenv("NSX_MGS_DIST")) == 0);
    h->mgs_dist_state = (wanted && comm_on_stream(h) && comm_streams_concurrent(h)) ? 1 : 0;
  }
  bool dist = h->comm && h->mgs_dist_state == 1 && !h->mgs_disabled && gram;
  if (!h->comm || dist) mgs_setup(h);
  dist = dist && h->mgs_links == 0 && 2 * dim + 1 < MGS_EXT_LEAVE;
  // entries per thread: the smallest instantiation (8, 10, 20) whose resident grid covers the vector
  int nwg = 1, per_thread = 1 << 30, e_inst = 0;
  const int n_inst = 3;  // the one-exchange sweep keeps a block of basis vectors in registers: 8, 10 or 12 entries per thread
  for (int k = 0; k < n_inst && h->mgs_max_wg; ++k) {
    const int es[3] = {8, 10, h->mgs_links == 0 ? 12 : 20};
    // with the collective inside the grid only the 8-entry instantiation: it holds 246 VGPRs (248 allocated), so a CU that carries
    // ONE of its workgroups keeps 264 registers per SIMD lane free -- exactly what a wave of RCCL's generic kernel needs (264; 256
    // threads, 19.7 KB of LDS) -- and the grid limit leaves such CUs (mgs_setup).  The 10- and 12-entry instantiations allocate 256:
    // RCCL's kernel finds no place beside them and the sweep times out (measured with a self-addressed send / receive in front of the
    // collective, tools/r04_self_p2p.sh: level 5, 8 entries: 23.0 -> 30.9 us per sweep, no fallback; level 7, 10 entries: time-out)
    // ... unless the compute stream leaves one CU per XCD to the communication stream (comm_reserve_cus, below): then any instantiation
    if (dist && es[k] != 8 && !h->cu_reserved) break;
    const int cap = dist ? (h->cu_reserved ? h->mgs_dist_cap_reserved[k] : h->mgs_max_wg_dist[k]) : h->mgs_max_wg_e[k];
    if (cap <= 0) continue;
    nwg = std::max(1, std::min(cap, cdiv(n, 256 * 4)));
    per_thread = cdiv(n, (int64_t)nwg * 256);
    e_inst = es[k];
    if (per_thread <= es[k]) break;
  }
  int per_thread_max = (dist && !h->cu_reserved) ? 8 : h->mgs_links == 0 ? 12 : 20;
  const int role = mgs_role(h, sp);
  if (dist && !h->cu_reserved && !h->cu_reserve_failed && h->mgs_dist_fit.find(role) == h->mgs_dist_fit.end()) {
    // too long for the 8-entry grid somewhere, but not for the larger instantiations on a masked compute stream?  NSX_COMM_CU_RESERVE:
    // 0 never, 1 (default) when that is what keeps the collective inside the grid, 2 always.  Every condition below is an answer all
    // ranks gave together, so every rank takes the same steps -- including the outcome of the reservation itself: if it fails anywhere
    // (no masked stream, the probe) every rank goes back to its plain streams and nobody asks again on this communicator.
    static const int reserve = getenv("NSX_COMM_CU_RESERVE") ? atoi(getenv("NSX_COMM_CU_RESERVE")) : 1;
    const bool fits8 = e_inst != 0 && per_thread <= 8;
    bool fits_reserved = false;
    for (int k = 0; k < 3 && !fits_reserved; ++k)
      fits_reserved = h->mgs_dist_cap_reserved[k] > 0 &&
                      cdiv(n, (int64_t)std::max(1, std::min(h->mgs_dist_cap_reserved[k], cdiv(n, 256 * 4))) * 256) <= (k == 0 ? 8 : k == 1 ? 10 : 12);
    const bool all8 = reserve > 0 ? comm_agree_all(h, fits8) : true;
    if (reserve > 0 && (reserve > 1 || !all8) && comm_agree_all(h, fits_reserved)) {
      const bool mine = comm_reserve_cus(h);
      if (!comm_agree_all(h, mine)) {
        if (mine) comm_release_cus(h);
        h->cu_reserve_failed = true;
      }
      if (h->cu_reserved) {  // choose the instantiation again, with the masked stream's limits
        per_thread_max = 12;
        nwg = 1, per_thread = 1 << 30, e_inst = 0;
        for (int k = 0; k < 3; ++k) {
          const int es[3] = {8, 10, 12};
          const int cap = h->mgs_dist_cap_reserved[k];
          if (cap <= 0) continue;
          nwg = std::max(1, std::min(cap, cdiv(n, 256 * 4)));
          per_thread = cdiv(n, (int64_t)nwg * 256);
          e_inst = es[k];
          if (per_thread <= es[k]) break;
        }
      }
    }
  }
  if (dist) {
    // does the resident grid hold the vector -- on EVERY rank?  (local lengths differ; a rank on the two-pass sweep and a rank on the
    // persistent one would all-reduce differently laid-out buffers.)  Agreed once per role of the vector in the solve.
    auto it = h->mgs_dist_fit.find(role);
    if (it == h->mgs_dist_fit.end()) it = h->mgs_dist_fit.emplace(role, comm_agree_all(h, e_inst != 0 && per_thread <= per_thread_max) ? 1 : 0).first;
    if (!it->second || dim + 2 > MGS_STEPS) dist = false;
  }
  if (fused_e) {
    nwg = h->schedF.n_waves;
    e_inst = fused_e;
    per_thread = fused_e;
  }
  if (!fused_e && ((h->comm && !dist) || h->mgs_max_wg == 0 || dim + 2 > MGS_STEPS || per_thread > per_thread_max || (h->mgs_links == 0 && !gram))) {
    // distributed solve: two collectives per sweep (mgs_lowsync); NSX_MGS_LOWSYNC=0: one launch + all-reduce per link, as the
    // reference's MPI run does.  Without a Gram cache (or too many vectors for it) the chain as well.
    if (h->ls_mode < 0) h->ls_mode = getenv("NSX_MGS_LOWSYNC") ? atoi(getenv("NSX_MGS_LOWSYNC")) : 2;  // read once per handle: 0 chain, 1 two collectives, 2 one
    // (one GPU, vector too long for the persistent sweep: the same two passes read the basis twice instead of four times)
    const bool too_long = !h->comm && !h->mgs_disabled && per_thread > per_thread_max;
    if ((h->comm || too_long) && h->ls_mode && gram && dim <= 31) mgs_lowsync(h, sp, w, dim, vs, slot0, out, consider, gram);
    else mgs_chain(h, sp, w, dim, vs, slot0, out, consider);
    return false;
  }
  const unsigned long long seq = ++h->pub_seq;
  double *ext_vals_this = nullptr;  // distributed: the buffer this sweep's collective works on
  h->mgs_last_e = e_inst;
  h->mgs_last_fused = fused_e ? 1 : 0;
  h->mgs_fused_launches += fused_e ? 1 : 0;
  h->mgs_last_nwg = nwg;
  h->mgs_last_dist = dist ? 1 : 0;
  h->mgs_max_e_seen = std::max(h->mgs_max_e_seen, e_inst);
  {
    LaunchScope ls(h, fused_e ? "ilu_mgs" : "mgs_sweep",
                   8.0 * n * (dim + 2) + (fused_e ? 12.0 * (double)h->schedF.in_block_nnz + (double)h->N2 * (4 + 8.0 * h->dim) - 8.0 * n : 0.0));
    MgsArgs V;
    for (int i = 0; i < dim; ++i) V.v[i] = vs[i];
    for (int i = dim; i < MGS_STEPS; ++i) V.v[i] = nullptr;
    const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
    const int M = h->mgs_links;
    int n_ = n, split = sp.split, gap = sp.gap, dim_ = dim, norm_ = normalize ? 1 : 0, consider_ = consider ? 1 : 0;
    unsigned long long *box = h->mgs_box.p + (size_t)h->mgs_parity * region, *box_next = h->mgs_box.p + (size_t)(1 - h->mgs_parity) * region;
    double *sout = h->scal.p + slot0, *pub_vals = h->pub_dev + slot0;
    unsigned long long *pub_flag = (unsigned long long *)(h->pub_dev + N_SLOTS), seq_ = seq;
    int *err = (int *)(h->pub_dev + N_SLOTS + 2);  // mapped host word
    unsigned long long *tail = h->mgs_box.p + 2 * region;
    int reset_wg = h->mgs_used_wg[1 - h->mgs_parity], reset_steps = h->mgs_used_steps[1 - h->mgs_parity];
    int drop_wg = h->gx_drop_wg;
    // Co-residency: the grid never exceeds what the device holds at once (mgs_setup), the stream is in-order and normally nothing
    // else runs on the device, so a plain launch places every workgroup at once.  (hipLaunchCooperativeKernel was an option until
    // round 3: it adds a launch-time size check and ~20 us of cross-queue synchronisation per launch but no residency guarantee
    // beyond that — /opt/skills/guides/MI355X_MICROARCH.md, "Residency and cooperative launch" — and was removed.)  What makes the
    // sweep safe is the bounded wait: should a workgroup be missing (another stream or process holds compute units), the kernel
    // ends without writing w and the sweep is redone by the launch-per-link chain below.
    const void *fn = fused_e ? ilu_mgs_fn(h->dim, fused_e) : dist ? mgs_one_fn(e_inst, true) : mgs_fn(M, e_inst);
    if (fused_e) {
      const IluSchedule &s_ = h->schedF;
      int reset_words = reset_wg * reset_steps;
      double *gram_ = gram;
      double guard_ = mgs_norm_guard(h);
      IluMgsArgs I{s_.pk_row_ptr.p, s_.pk_rows.p, s_.pk_slab_ptr.p, reinterpret_cast<const uint32_t *>(s_.pk_meta.p), s_.pk_val.p, s_.pk_dinv.p, ilu_rhs,
                   (s_.max_wave_rows + 64) * h->dim, nullptr};
      // development (tools/r05_ilu_mgs_trace.sh): wall-clock stamps of every wave of ONE launch (the NSX_ILU_MGS_TRACE_CALL-th, default 3000)
      DevBuf<unsigned long long> trace_buf;
      const char *trace_path = getenv("NSX_ILU_MGS_TRACE");
      const bool traced = trace_path && h->mgs_fused_launches == (getenv("NSX_ILU_MGS_TRACE_CALL") ? atoi(getenv("NSX_ILU_MGS_TRACE_CALL")) : 3000);
      if (traced) {
        trace_buf.alloc((size_t)nwg * 4 * 16);
        trace_buf.zero(h->stream);
        I.trace = trace_buf.p;
      }
      const size_t shm = ilu_mgs_lds_doubles(I.ilu_doubles, fused_e) * sizeof(double);
      void *args[] = {&n_, &w, &V, &dim_, &gram_, &box, &box_next, &reset_words, &sout, &err, &tail, &norm_, &consider_, &pub_vals, &pub_flag, &seq_, &drop_wg, &guard_, &I};
      HIP_CHECK(hipLaunchKernel(fn, dim3(nwg), dim3(256), args, shm, h->stream));
      if (traced) {
        std::vector<unsigned long long> tr((size_t)nwg * 4 * 16);
        trace_buf.download(tr.data(), tr.size(), h->stream);
        if (FILE *f = fopen(trace_path, "w")) {
          fprintf(f, "# k_ilu_mgs launch %d: dim %d, %d workgroups, %d entries per thread; per wave: workgroup wave rows, then stamps 0..8 in ticks of 10 ns relative to the grid's first stamp\n"
                     "# 0 start, 1 rhs rows in LDS, 2 forward sweep done, 3 backward sweep done (sweeping wave only), 4 basis entries arrived, 5 z complete (barrier), 6 local sums, 7 totals picked up, 8 update done\n",
                  h->mgs_fused_launches, dim, nwg, fused_e);
          unsigned long long t0 = ~0ull;
          for (size_t q = 0; q < tr.size(); q += 16)
            if (tr[q]) t0 = std::min(t0, tr[q]);
          std::vector<int32_t> rp((size_t)nwg + 1);
          s_.pk_row_ptr.download(rp.data(), rp.size(), h->stream);
          for (int b = 0; b < nwg; ++b)
            for (int wv_ = 0; wv_ < 4; ++wv_) {
              fprintf(f, "%d %d %d", b, wv_, rp[b + 1] - rp[b]);
              for (int k = 0; k < 9; ++k) {
                const unsigned long long v = tr[((size_t)b * 4 + wv_) * 16 + k];
                fprintf(f, " %lld", v ? (long long)(v - t0) : -1ll);
              }
              fprintf(f, "\n");
            }
          fclose(f);
        }
      }
      h->mgs_used_wg[h->mgs_parity] = 1;
      h->mgs_used_steps[h->mgs_parity] = (2 * dim + 2) * nwg;
    } else if (M == 0) {
      // k_mgs_one: mailboxes box[v * nwg + wg] for the 2 dim + 1 values of the single exchange (+ 1 for the explicit norm)
      int reset_words = reset_wg * reset_steps;
      double *gram_ = gram;
      double guard_ = mgs_norm_guard(h);
      MgsExt ext{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
      if (dist) {
        ext.leave = h->mgs_leave_req ? 1 : 0;
        ext.vals = h->mgs_ext_vals.p + (size_t)h->mgs_ext_parity * MGS_EXT_VALS;
        ext.vals_other = h->mgs_ext_vals.p + (size_t)(1 - h->mgs_ext_parity) * MGS_EXT_VALS;
        ext.flag = h->mgs_ext_words.p;
        ext.arrive = (unsigned int *)(h->mgs_ext_words.p + 1);
        ext.abort_seq = h->mgs_ext_words.p + 2;
        ext.norm_out = h->scal.p + S_LS_NORM;
        ext_vals_this = ext.vals;
      }
      void *args[] = {&n_, &split, &gap, &w, &V, &dim_, &gram_, &box, &box_next, &reset_words, &sout, &err, &tail, &norm_, &consider_, &pub_vals, &pub_flag, &seq_, &drop_wg, &guard_, &ext};
      HIP_CHECK(hipLaunchKernel(fn, dim3(nwg), dim3(256), args, 0, h->stream));
      if (dist) {
        // the collective of this sweep, on the communication stream: it starts when the grid's 2 dim + 1 reducers have delivered
        h->mgs_ext_expected += (unsigned int)(2 * dim + 1);
        comm_ext_allreduce(h, ext.vals, MGS_EXT_VALS, MGS_EXT_FAIL, ext.arrive, h->mgs_ext_expected, ext.flag, seq);
        h->mgs_ext_parity ^= 1;
        h->slot_nb[S_LS_NORM] = 0;
      }
      h->mgs_used_wg[h->mgs_parity] = 1;
      h->mgs_used_steps[h->mgs_parity] = (2 * dim + 2) * nwg;
    } else if (M == 1) {
      void *args[] = {&n_, &split, &gap, &w, &V, &dim_, &box, &box_next, &reset_wg, &reset_steps, &sout, &err, &tail, &norm_, &consider_, &pub_vals, &pub_flag, &seq_, &drop_wg};
      HIP_CHECK(hipLaunchKernel(fn, dim3(nwg), dim3(256), args, 0, h->stream));
      h->mgs_used_wg[h->mgs_parity] = nwg;
      h->mgs_used_steps[h->mgs_parity] = dim + 2;
    } else {
      // k_mgs_blk: the region is a flat array of (exchanges x values x workgroups) words; "steps" counts words, "wg" is 1
      int reset_words = reset_wg * reset_steps;
      void *args[] = {&n_, &split, &gap, &w, &V, &dim_, &box, &box_next, &reset_words, &sout, &err, &tail, &norm_, &consider_, &pub_vals, &pub_flag, &seq_, &drop_wg};
      HIP_CHECK(hipLaunchKernel(fn, dim3(nwg), dim3(256), args, 0, h->stream));
      const int nv = M + M * (M - 1) / 2 + 1;
      h->mgs_used_wg[h->mgs_parity] = 1;
      h->mgs_used_steps[h->mgs_parity] = (cdiv(dim, M) + 1) * nv * nwg;
    }
    h->mgs_used_wg[1 - h->mgs_parity] = h->mgs_used_steps[1 - h->mgs_parity] = 0;
    h->mgs_parity ^= 1;
    for (int i = 0; i <= dim + 1; ++i) h->slot_nb[slot0 + i] = 0;
  }
  // w is final (and normalised) once the kernel has run: work that only depends on it may be enqueued before the host
  // has the coefficients
  const bool ran_ahead = normalize && !consider && after_launch;
  if (ran_ahead) (*after_launch)();
  wait_published(h, seq);
  if (*(volatile int *)(h->pub_host + N_SLOTS + 2)) {
    if (dist) {
      // Whose verdict was it?  Wait for the collective (the grid is gone, so its kernels find room whatever kept them) and look at the
      // words it summed: a failure or a leave request is known to every rank alike -- all redo the sweep in two passes.  Neither: only
      // THIS rank's grid gave up on the flag; its peers may be past this sweep already.
      HIP_CHECK(hipStreamSynchronize(h->stream));
      HIP_CHECK(hipStreamSynchronize(h->comm_stream));
      double words[2] = {0.0, 0.0};
      HIP_CHECK(hipMemcpy(words, ext_vals_this + MGS_EXT_LEAVE, 2 * sizeof(double), hipMemcpyDeviceToHost));
      if (words[0] == 0.0 && words[1] == 0.0) {
        const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
        std::vector<unsigned long long> tail(MGS_TAIL, 0);
        HIP_CHECK(hipMemcpy(tail.data(), h->mgs_box.p + 2 * region, MGS_TAIL * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned int committed = 0;
        for (unsigned long long v : tail) committed += v == seq;
        if (committed != 0) NSX_THROW(NSX_ERR_HIP, "Gram-Schmidt sweep: %u workgroups had written w when another one gave up on the collective", committed);
        if (++h->mgs_local_timeouts > 2)
          NSX_THROW(NSX_ERR_COMM, "Gram-Schmidt sweep: the collective inside the persistent grid did not arrive within %.0f s for the third time on rank %d", 1e-8 * (double)GX_EXT_TIMEOUT_TICKS, h->rank);
        fprintf(stderr, "[nsx] warning: rank %d: the collective inside the Gram-Schmidt sweep came too late for its grid (%.0f s): this sweep is finished from the sums it delivered, "
                        "and all ranks are asked to use the two-pass sweep from the next one on\n", h->rank, 1e-8 * (double)GX_EXT_TIMEOUT_TICKS);
        // mailboxes and error word as a new handle's; the words of the collective protocol (arrival count, flag, abort word) stay: they are cumulative
        HIP_CHECK(hipMemsetAsync(h->mgs_box.p, 0xff, 2 * region * sizeof(unsigned long long), h->stream));
        HIP_CHECK(hipMemsetAsync(h->mgs_box.p + 2 * region, 0, MGS_TAIL * sizeof(unsigned long long), h->stream));
        *(volatile int *)(h->pub_host + N_SLOTS + 2) = 0;
        h->mgs_used_wg[0] = h->mgs_used_wg[1] = h->mgs_used_steps[0] = h->mgs_used_steps[1] = 0;
        h->n_persistent_fallbacks++;
        h->mgs_leave_req = true;
        // the sweep itself, from the global sums: coefficients and |w'|^2 as every grid computed them (k_ls_solve evaluates the same formula
        // in another grouping: a decision on its threshold's knife edge could differ from the peers' in the last bit), then the updates
        if (!h->ls_partial.p) {
          h->ls_partial.alloc((size_t)LS_VALS * LS_BLOCKS);
          h->ls_vals.alloc(LS_VALS);
        }
        MgsArgs V;
        for (int i = 0; i < MGS_STEPS; ++i) V.v[i] = i < dim ? vs[i] : nullptr;
        hipLaunchKernelGGL(k_ext_to_ls, dim3(1), dim3(64), 0, h->stream, dim, ext_vals_this, h->ls_vals.p);
        hipLaunchKernelGGL(k_ls_solve, dim3(1), dim3(64), 0, h->stream, dim, h->ls_vals.p, gram, h->scal.p + slot0);
        for (int i = 0; i <= dim + 1; ++i) h->slot_nb[slot0 + i] = 0;
        const int nb = red_blocks(h, n);
        hipLaunchKernelGGL(k_ls_update, dim3(nb), dim3(256), 0, h->stream, n, sp.split, sp.gap, w, V, dim, h->scal.p + slot0, red_out(h, S_LS_NORM, nb));
        h->slot_nb[S_LS_NORM] = nb > 1 ? nb : 0;
        double tmp[N_TMP_MAX + 2];
        read_scalars(h, slot0, dim + 2, tmp);
        if (!(tmp[dim] > mgs_norm_guard(h) * tmp[dim + 1])) {  // the peers' grids refused the formula as well: the sweep's second collective
          finalize_slots(h, S_LS_NORM, 1);
          comm_allreduce_scalars(h, S_LS_NORM, 1);
          tmp[dim] = read_scalar(h, S_LS_NORM);
        }
        h->slot_nb[S_LS_NORM] = 0;
        for (int i = 0; i <= dim; ++i) out[i] = tmp[i];
        if (consider) out[dim + 1] = tmp[dim + 1];
        if (ran_ahead) h->mgs_redo_ahead = true;
        return false;
      }
    }
    const unsigned int committed = mgs_recover(h, seq);
    if (committed != 0) NSX_THROW(NSX_ERR_HIP, "Gram-Schmidt sweep: %u workgroups had written w when another one timed out", committed);
    // what after_launch enqueued (the next operator application) used the unfinished w: its result is a temporary that the
    // caller recomputes when told that w was not normalised here
    // (distributed: the failure word travelled through the collective, so every rank is here and redoes the sweep in two passes)
    if (dist) {
      h->mgs_leave_req = false;
      if (h->ls_mode < 0) h->ls_mode = getenv("NSX_MGS_LOWSYNC") ? atoi(getenv("NSX_MGS_LOWSYNC")) : 2;
      if (h->ls_mode && dim <= 31) mgs_lowsync(h, sp, w, dim, vs, slot0, out, consider, gram);
      else mgs_chain(h, sp, w, dim, vs, slot0, out, consider);
    } else {
      if (fused_e) ilu_solve(h, h->gA, h->schedF, h->luF.p, ilu_rhs, w, h->dim, "ilu_solve_F");  // the fused launch ended without writing w: z first, then the chain
      mgs_chain(h, sp, w, dim, vs, slot0, out, consider);
    }
    if (ran_ahead) h->mgs_redo_ahead = true;
    return false;
  }
  for (int i = 0; i <= dim + ((consider || h->mgs_links == 0) ? 1 : 0); ++i) out[i] = h->pub_host[slot0 + i];
  if (dist && h->pub_host[slot0 + dim + 2] != 0.0) {
    // the Gram formula for |w'|^2 was refused (alike on every rank): the grid left its local sum in the scalar slot and did not
    // normalise; the second collective of the sweep sums it over the ranks.  (A NaN -- a mailbox of that sum timed out somewhere --
    // is replaced by a plain dot product: w itself is complete.)
    comm_allreduce_scalars(h, S_LS_NORM, 1);
    double nrm2 = read_scalar(h, S_LS_NORM);
    if (nrm2 != nrm2) {
      v_dot(h, sp, w, w, S_LS_NORM);
      nrm2 = read_scalar(h, S_LS_NORM);
    }
    out[dim] = nrm2;
    if (ran_ahead) h->mgs_redo_ahead = true;  // what was enqueued behind the launch used the unnormalised w
    return false;
  }
  if (!normalize) return false;
  // the kernel's own decision, recomputed from the same two numbers
  return !consider || std::sqrt(out[dim]) > 10. * std::sqrt(out[dim + 1]) * 1.4901161193847656e-08;
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
  const int nb = red_blocks(h, n);
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
// Enqueue the publication of scal[slot0 .. slot0+count) and return its sequence number; the host may enqueue more work
// before it waits for the values (collect_published).  No other publication may be enqueued in between.
unsigned long long publish_scalars(nsx_handle *h, int slot0, int count) {
  if (count > 64) NSX_THROW(NSX_ERR_ARG, "internal: read_scalars range too long");
  NbArgs args;
  for (int i = 0; i < count; ++i) {
    args.nb[i] = h->slot_nb[slot0 + i];
    h->slot_nb[slot0 + i] = 0;
  }
  const unsigned long long seq = ++h->pub_seq;
  unsigned long long *flag_dev = (unsigned long long *)(h->pub_dev + N_SLOTS);
  hipLaunchKernelGGL(k_publish, dim3(count), dim3(256), 0, h->stream, slot0, count, args, h->red_partial.p, h->scal.p, h->pub_dev, flag_dev, seq,
                     h->pub_counter.p);
  return seq;
}
void collect_published(nsx_handle *h, unsigned long long seq, int slot0, int count, double *out) {
  wait_published(h, seq);
  for (int i = 0; i < count; ++i) out[i] = h->pub_host[slot0 + i];
}
void read_scalars(nsx_handle *h, int slot0, int count, double *out) { collect_published(h, publish_scalars(h, slot0, count), slot0, count, out); }
void write_scalar(nsx_handle *h, int slot, double v) {
  HIP_CHECK(hipStreamSynchronize(h->stream));
  h->scal_host[slot] = v;
  h->slot_nb[slot] = 0;
  HIP_CHECK(hipMemcpyAsync(h->scal.p + slot, h->scal_host + slot, sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

int cg_dirty_words(nsx_handle *h);
// diagnostics (nsx_persistent_state): words of the region the next sweep would use that are not empty.  A healthy handle keeps
// that region empty (every launch clears the other region for its successor); after a time-out mgs_recover clears both.
static int mgs_dirty_words(nsx_handle *h) {
  if (!h->mgs_box.p) return 0;
  const size_t region = std::max(MGS_REGION, MGS_BLK_REGION);
  std::vector<unsigned long long> w(region);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  HIP_CHECK(hipMemcpy(w.data(), h->mgs_box.p + (size_t)h->mgs_parity * region, region * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  int dirty = 0;
  for (unsigned long long v : w) dirty += v != GX_EMPTY;
  return dirty;
}

}  // namespace nsx

// One GMRES cycle's worth of orthogonalisation on the caller's vectors, through the very sweep the solvers use (v_mgs with the
// basis' Gram matrix kept on the device): vectors[0] is normalised, vectors[k] is swept against the k vectors in front of it and
// normalised.  norm_guard >= 0 replaces the threshold below which the Gram formula for |w'|^2 is refused (0: always the formula,
// 1e300: always the explicitly summed norm); < 0 keeps the handle's.  For the tests of that formula (tests/test_gpu_errors.py).
extern "C" int nsx_gram_schmidt_cycle(nsx_handle *h, int n, int m, double *vectors, double norm_guard, double *coeffs, double *norms2) {
  if (!h || !vectors || !coeffs || !norms2 || n < 1 || m < 1 || m > 30) return NSX_ERR_ARG;
  try {
    HIP_CHECK(hipSetDevice(h->prm.device));
    std::vector<nsx::DevBuf<double>> v(m);
    for (int k = 0; k < m; ++k) v[k].upload(vectors + (size_t)k * n, n, h->stream);
    if (!h->ls_gram.p) {
      h->ls_gram.alloc(4 * 1024);
      h->ls_gram.zero(h->stream);
    }
    const double keep = h->mgs_guard_override;
    h->mgs_guard_override = norm_guard;
    try {
      const nsx::Span sp(n);
      nsx::v_dot(h, sp, v[0].p, v[0].p, 40);
      norms2[0] = nsx::read_scalar(h, 40);
      nsx::v_scale(h, sp, v[0].p, 1.0 / std::sqrt(norms2[0]));
      for (int k = 1; k < m; ++k) {
        double *vs[32], out[34];
        for (int i = 0; i < k; ++i) vs[i] = v[i].p;
        const bool normalized = nsx::v_mgs(h, sp, v[k].p, k, vs, 8, true, out, nullptr, false, h->ls_gram.p);
        for (int i = 0; i < k; ++i) coeffs[(size_t)k * m + i] = out[i];
        norms2[k] = out[k];
        if (!normalized && out[k] > 0.0) nsx::v_scale(h, sp, v[k].p, 1.0 / std::sqrt(out[k]));
      }
    } catch (...) {
      h->mgs_guard_override = keep;
      throw;
    }
    h->mgs_guard_override = keep;
    for (int k = 0; k < m; ++k) v[k].download(vectors + (size_t)k * n, n, h->stream);
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

extern "C" int nsx_persistent_state(nsx_handle *h, int state[4]) {
  if (!h || !state) return NSX_ERR_ARG;
  try {
    HIP_CHECK(hipSetDevice(h->prm.device));
    state[0] = h->mgs_box.p != nullptr && !h->mgs_disabled && h->mgs_max_wg > 0;
    state[1] = h->cg_box.p != nullptr && !h->cg_disabled && h->cg_max_wg > 0;
    state[2] = h->n_persistent_fallbacks;
    state[3] = nsx::mgs_dirty_words(h) + nsx::cg_dirty_words(h);
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

// Which code paths this handle's products and solves take (tests, bench.py's rehearsal log): see include/nsx.h
extern "C" int nsx_path_info(nsx_handle *h, int info[32]) {
  if (!h || !info) return NSX_ERR_ARG;
  for (int k = 0; k < 32; ++k) info[k] = 0;
  if (!h->have_mesh) return NSX_OK;
  try {
    HIP_CHECK(hipSetDevice(h->prm.device));
    const nsx::SpmvBlocked &b = h->blkA;
    info[0] = nsx::blocked_usable(h) ? 1 : 0;
    info[1] = b.n_chunks;
    info[2] = b.n_chunks_if;
    info[3] = h->mgs_last_e;
    info[4] = h->mgs_last_nwg;
    info[5] = h->mgs_last_dist;
    info[6] = h->mgs_max_e_seen;
    info[7] = h->cu_reserved;
    info[8] = h->cg_last_path;
    info[9] = h->schedS.n_blocks;
    info[10] = (int)h->haloU.nbr.size();
    info[11] = h->haloU.nbr.empty() ? 0 : h->haloU.send_ptr[h->haloU.nbr.size()];
    info[12] = h->N2_loc - h->N2;
    info[13] = h->schedS.dense ? 1 : 0;
    info[14] = h->mgs_disabled ? 1 : 0;
    info[15] = h->n_persistent_fallbacks;
    // what the distributed sweep WOULD run with an RCCL communicator (a rehearsal over host callbacks runs the two-pass sweep): the
    // instantiation for the velocity and the block vector on plain streams (only the 8-entry grid leaves RCCL's kernel room) and on
    // masked ones; 0 = the resident grid does not hold the vector
    if (!h->mgs_disabled) {
      nsx::mgs_setup(h);
      auto fit = [&](int n, const int *caps, int n_inst) {
        for (int k = 0; k < n_inst; ++k) {
          const int es[3] = {8, 10, 12};
          if (caps[k] <= 0) continue;
          const int nwg = std::max(1, std::min(caps[k], nsx::cdiv(n, 256 * 4)));
          if (nsx::cdiv(n, (int64_t)nwg * 256) <= es[k]) return es[k];
        }
        return 0;
      };
      info[16] = fit(h->n_u, h->mgs_max_wg_dist, 1);
      info[17] = fit(h->n_u, h->mgs_dist_cap_reserved, 3);
      info[18] = fit(h->n_u + h->n_p, h->mgs_max_wg_dist, 1);
      info[19] = fit(h->n_u + h->n_p, h->mgs_dist_cap_reserved, 3);
      info[21] = fit(h->n_u, h->mgs_max_wg_e, 3);  // one GPU, no communicator
    }
    info[20] = nsx::cdiv(std::max(1, h->schedS.n_blocks), 1024);  // Schur blocks per entry of a partial-sum array of the two-launch CG (1: no fold launch)
    info[22] = h->N2;
    info[23] = h->NP;
    info[24] = h->mgs_last_fused;
    info[25] = h->mgs_fused_launches;
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}
