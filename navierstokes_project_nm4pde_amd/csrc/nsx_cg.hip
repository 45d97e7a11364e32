// nsx_cg.hip — the inner CG solve on the Schur complement as ONE persistent launch.
//
//   SolverCG<Vector>(solver_control_S).solve(negative_S_tilde, yp, tmp, preconditioner_S)
//                                  reference Navier-Stokes/include/Preconditioners.hpp:179-182 (SIMPLE), :388-390 (Yosida), :500-502 (aYosida)
//   with solver_control_S(maxiter, 1e-2 * tmp.l2_norm())  (:179, :388, :500) and preconditioner_S = ILU(0) per rank (:148, :362, :471).
//
// negative_S_tilde has ~46 k rows at 1 M DoF: as separate launches an iteration is five kernels of ~10 us each, all of them
// latency bound (nsx_solve.hip: cg()).  Here workgroup b owns Schur ILU block b for the whole solve: its rows of x, g and d
// stay in LDS, its block of the preconditioner is the explicit inverse P_b = U^-1 D^-1 L^-1 (k_ilu_invert), and an iteration
// costs TWO grid-wide exchanges (nsx_grid.hpp):
//     A:  h_i = sum_j S_ij d_j  with d_j = beta d_j(old) - h_j(old) evaluated while gathering (the neighbours' d of this
//         iteration is never waited for: both operands were complete before the previous exchange) ; partial d.h
//     -- exchange 1: d.h --
//     B:  alpha = g.h / d.h ; x += alpha d ; g += alpha h ; h = P_b g (block local) ; partials g.g and g.h
//     -- exchange 2: g.g, g.h --      res = sqrt(|g.g|), SolverControl::check, beta = g.h / g.h(old)
// The arithmetic of every entry is SolverCG's (same recurrences, same operands); sums are fixed-order, so results do not
// depend on timing.  d is double-buffered in global memory (written with write-through stores, gathered with L1-bypassing
// loads), so a workgroup never overwrites a value a neighbour may still be reading.
#include "nsx_grid.hpp"

namespace nsx {

constexpr int CG_MAXB = 256;    // rows of one Schur block: one thread per row in the update phases
constexpr int CG_MAX_WG = 1024;
constexpr int CG_NV = 3;        // values per exchange
constexpr int CG_RING = 4;      // mailbox rows in flight; row (e + 2) % 4 is emptied at exchange e
constexpr size_t CG_REGION = (size_t)CG_RING * CG_NV * CG_MAX_WG + (size_t)CG_RING * CG_NV;
enum { S_CGP = 100 };           // scalar slots of the publication: steps, last residual, status, tolerance

__device__ __forceinline__ double ld_agent(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int W>
__device__ __forceinline__ double cg_group_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  if (W >= 8) v += dpp_f64<0x141>(v);
  if (W >= 16) v += dpp_f64<0x140>(v);
  return v;
}

// grid-wide fixed-order sums of NV values; returns false when a wait timed out (the grid is then abandoned)
template <int NV>
__device__ __forceinline__ bool cg_exchange(const double (&part)[NV], double (&tot)[NV], unsigned long long *box, int e, int nwg, double (*sh)[4],
                                            double *bc) {
  // every data store of this workgroup (h, d: write-through) is acknowledged before its partial sums go out
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int wg = blockIdx.x, tid = threadIdx.x;
  unsigned long long *row = box + (size_t)(e % CG_RING) * CG_NV * CG_MAX_WG, *row2 = box + (size_t)((e + 2) % CG_RING) * CG_NV * CG_MAX_WG;
  unsigned long long *total = box + (size_t)CG_RING * CG_NV * CG_MAX_WG + (size_t)(e % CG_RING) * CG_NV;
  unsigned long long *total2 = box + (size_t)CG_RING * CG_NV * CG_MAX_WG + (size_t)((e + 2) % CG_RING) * CG_NV;
  double bs[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) bs[v] = gx_block_sum(part[v], sh[v & 1]);
  int lerr = 0;
  if (tid == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) gx_post(row + (size_t)v * CG_MAX_WG + wg, bs[v]);
#pragma unroll
    for (int v = 0; v < CG_NV; ++v) gx_clear(row2 + (size_t)v * CG_MAX_WG + wg);
  }
  if (wg == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      double a = 0.0;
      for (int s = tid; s < nwg; s += 256) a += gx_wait_value(row + (size_t)v * CG_MAX_WG + s, &lerr);
      const double t = gx_block_sum(a, sh[(v + NV) & 1]);
      if (tid == 0) gx_post(total + v, t);
    }
    if (tid < CG_NV) gx_clear(total2 + tid);
  }
  if (tid == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) bc[v] = gx_wait_value(total + v, &lerr);
  }
  const int dead = __syncthreads_or(lerr);
#pragma unroll
  for (int v = 0; v < NV; ++v) tot[v] = bc[v];
  return dead == 0;
}

__global__ __launch_bounds__(256) void k_cg_schur(int n_blocks, const int32_t *__restrict__ bptr, const int32_t *__restrict__ rp,
                                                  const int32_t *__restrict__ ci, const double *__restrict__ sv,
                                                  const int64_t *__restrict__ dn_off, const double *__restrict__ P, const double *__restrict__ b,
                                                  double *x, double *D0, double *D1, double *H, double rtol, int maxiter,
                                                  unsigned long long *box, unsigned long long *box_other, double *pub_vals,
                                                  unsigned long long *pub_flag, unsigned long long seq, int *err_dev) {
  __shared__ double gs[CG_MAXB], hs[CG_MAXB], ds[CG_MAXB], hvs[CG_MAXB], xs[CG_MAXB];
  __shared__ double sh[2][4], bc[CG_NV];
  const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x;
  // leave the other region empty for the next launch (stream order makes this visible to it)
  for (size_t q = (size_t)wg * 256 + tid; q < CG_REGION; q += (size_t)nwg * 256) box_other[q] = GX_EMPTY;
  const int r0 = bptr[wg], nb = bptr[wg + 1] - r0;
  const double *Pb = P + dn_off[wg];
  const int grp = tid >> 4, lane = tid & 15;
  const bool own = tid < nb;
  int e = 0;

  // h = P_b g on the block (16 lanes per row), result in hs[] and (write-through) in H
  auto apply_P = [&]() {
    for (int q = grp; q < nb; q += 16) {
      const double *prow = Pb + (size_t)q * nb;
      double a0 = 0.0, a1 = 0.0;
      int j = lane;
      for (; j + 16 < nb; j += 32) {
        a0 += prow[j] * gs[j];
        a1 += prow[j + 16] * gs[j + 16];
      }
      if (j < nb) a0 += prow[j] * gs[j];
      const double acc = cg_group_sum<16>(a0 + a1);
      if (lane == 0) hs[q] = acc;
    }
    __syncthreads();
    if (own) st_agent(H + r0 + tid, hs[tid]);
  };

  // ---- g = A x - b ; h = P g ; sums g.g, b.b, g.h
  double bi = 0.0;
  if (own) {
    xs[tid] = x[r0 + tid];
    bi = b[r0 + tid];
  }
  for (int q = grp; q < nb; q += 16) {
    const int i = r0 + q;
    double acc = 0.0;
    for (int k = rp[i] + lane; k < rp[i + 1]; k += 16) acc += sv[k] * x[ci[k]];
    acc = cg_group_sum<16>(acc);
    if (lane == 0) gs[q] = acc;
  }
  __syncthreads();
  if (own) gs[tid] = gs[tid] - bi;
  __syncthreads();
  apply_P();
  double tot3[3];
  {
    const double part[3] = {own ? gs[tid] * gs[tid] : 0.0, own ? bi * bi : 0.0, own ? gs[tid] * hs[tid] : 0.0};
    if (!cg_exchange<3>(part, tot3, box, e++, nwg, sh, bc)) {
      if (tid == 0) __hip_atomic_store(err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wg == 0 && tid == 0) {
        __hip_atomic_store(pub_vals + 2, 3.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  const double tol = rtol * sqrt(tot3[1]);  // solver_control_S(maxiter, 1e-2 * tmp.l2_norm())
  double res = sqrt(tot3[0]), gh = tot3[2], beta = 0.0;
  int it = 0;
  // SolverControl::check: 0 iterate, 1 success, 2 failure
  int conv = res <= tol ? 1 : ((it >= maxiter || res != res) ? 2 : 0);
  bool dead = false;
  while (conv == 0) {
    ++it;
    const double *Dp = (it & 1) ? D0 : D1;
    double *Dc = (it & 1) ? D1 : D0;
    // ---- A: h = A d, d = -h(old) in the first iteration, beta d(old) - h(old) afterwards
    for (int q = grp; q < nb; q += 16) {
      const int i = r0 + q;
      double acc = 0.0;
      if (it == 1) {
        for (int k = rp[i] + lane; k < rp[i + 1]; k += 16) acc += sv[k] * (-ld_agent(H + ci[k]));
      } else {
        for (int k = rp[i] + lane; k < rp[i + 1]; k += 16) {
          const int j = ci[k];
          acc += sv[k] * __builtin_fma(beta, ld_agent(Dp + j), -ld_agent(H + j));  // the owner's own expression, bit for bit
        }
      }
      acc = cg_group_sum<16>(acc);
      if (lane == 0) hvs[q] = acc;
    }
    if (own) {
      const double dcur = it == 1 ? -hs[tid] : __builtin_fma(beta, ds[tid], -hs[tid]);  // d = beta d - h (SolverCG: d.sadd(beta, -1., h))
      ds[tid] = dcur;
      st_agent(Dc + r0 + tid, dcur);
    }
    __syncthreads();
    double tot1[1];
    {
      const double part[1] = {own ? ds[tid] * hvs[tid] : 0.0};
      if (!cg_exchange<1>(part, tot1, box, e++, nwg, sh, bc)) {
        dead = true;
        break;
      }
    }
    // ---- B: alpha = g.h / d.h ; x += alpha d ; g += alpha h ; h = P g
    const double alpha = gh / tot1[0];
    if (own) {
      xs[tid] += alpha * ds[tid];
      gs[tid] = gs[tid] + alpha * hvs[tid];
    }
    __syncthreads();
    apply_P();
    double tot2[2];
    {
      const double part[2] = {own ? gs[tid] * gs[tid] : 0.0, own ? gs[tid] * hs[tid] : 0.0};
      if (!cg_exchange<2>(part, tot2, box, e++, nwg, sh, bc)) {
        dead = true;
        break;
      }
    }
    res = sqrt(fabs(tot2[0]));
    conv = res <= tol ? 1 : ((it >= maxiter || res != res) ? 2 : 0);
    beta = tot2[1] / gh;
    gh = tot2[1];
  }
  if (dead && tid == 0) __hip_atomic_store(err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (!dead && own) x[r0 + tid] = xs[tid];
  if (wg == 0 && tid == 0) {
    __hip_atomic_store(pub_vals + 0, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub_vals + 1, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub_vals + 2, dead ? 3.0 : (conv == 1 ? 0.0 : 1.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub_vals + 3, tol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static void cg_setup(nsx_handle *h) {
  if (h->cg_box.p || h->cg_disabled) return;
  h->cg_max_wg = 0;
  if (getenv("NSX_CG_PERSISTENT") && atoi(getenv("NSX_CG_PERSISTENT")) == 0) {
    h->cg_disabled = true;
    return;
  }
  int cus = 0, per_cu = 0;
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->prm.device));
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_schur, 256, 0));
  h->cg_box.alloc(2 * CG_REGION + 2);
  HIP_CHECK(hipMemsetAsync(h->cg_box.p, 0xff, 2 * CG_REGION * sizeof(unsigned long long), h->stream));
  HIP_CHECK(hipMemsetAsync(h->cg_box.p + 2 * CG_REGION, 0, 2 * sizeof(unsigned long long), h->stream));
  h->cg_max_wg = std::min(CG_MAX_WG, per_cu * cus);
  if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] persistent Schur CG: %d CUs x %d resident workgroups, grid <= %d\n", cus, per_cu, h->cg_max_wg);
}

// CG on negative_S_tilde with the explicit block inverses as preconditioner, one launch.  Returns false when the launch-per-
// operation path has to be used instead (distributed run, blocks too large or too many, persistent kernels disabled).
bool cg_schur_persistent(nsx_handle *h, double *x, const double *b, double rtol, int maxiter, int *steps, double *last, int *status) {
  const IluSchedule &s = h->schedS;
  if (h->comm || !s.dense || s.max_rows > CG_MAXB) return false;
  cg_setup(h);
  if (h->cg_max_wg == 0 || s.n_blocks > h->cg_max_wg) return false;
  const int n = h->n_p;
  if ((int)h->cg_vec.n < 3 * n) h->cg_vec.alloc((size_t)3 * n);
  double *D0 = h->cg_vec.p, *D1 = D0 + n, *H = D1 + n;
  const unsigned long long seq = ++h->pub_seq;
  unsigned long long *box = h->cg_box.p + (size_t)h->cg_parity * CG_REGION, *box_other = h->cg_box.p + (size_t)(1 - h->cg_parity) * CG_REGION;
  int *err_dev = (int *)(h->cg_box.p + 2 * CG_REGION);
  double *pub_vals = h->pub_dev + S_CGP;
  unsigned long long *pub_flag = (unsigned long long *)(h->pub_dev + N_SLOTS);
  ProfEntry *pe = nullptr;
  {
    LaunchScope ls(h, "cg_S", 0.0);
    pe = ls.e;
    hipLaunchKernelGGL(k_cg_schur, dim3(s.n_blocks), dim3(256), 0, h->stream, s.n_blocks, s.block_ptr.p, h->gS.rowptr.p, h->gS.colind.p, h->vSchur.p,
                       s.dn_off.p, s.dn_P.p, b, x, D0, D1, H, rtol, maxiter, box, box_other, pub_vals, pub_flag, seq, err_dev);
  }
  h->cg_parity ^= 1;
  wait_published(h, seq);
  const int st = (int)h->pub_host[S_CGP + 2];
  if (st == 3) {
    // a workgroup never arrived (the grid was not co-resident: something else holds compute units).  x is untouched; clean
    // up and leave the persistent path for good on this handle
    HIP_CHECK(hipStreamSynchronize(h->stream));
    HIP_CHECK(hipMemsetAsync(h->cg_box.p, 0xff, 2 * CG_REGION * sizeof(unsigned long long), h->stream));
    HIP_CHECK(hipMemsetAsync(h->cg_box.p + 2 * CG_REGION, 0, 2 * sizeof(unsigned long long), h->stream));
    h->cg_max_wg = 0;
    h->cg_disabled = true;
    if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] persistent Schur CG timed out: falling back to one launch per operation\n");
    return false;
  }
  *steps = (int)h->pub_host[S_CGP];
  *last = h->pub_host[S_CGP + 1];
  *status = st;
  // algorithmic bytes: per iteration the matrix (12 B / entry) and the block inverses once, plus the vectors
  if (pe) pe->bytes += (double)(*steps + 1) * (12.0 * h->gS.nnz() + 8.0 * (double)s.dn_entries + 48.0 * n);
  return true;
}

}  // namespace nsx
