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

constexpr int CG_THREADS = 256;  // 256 or 512 (two halves of 256: each streams the slabs of every other 16-row round of the block)
constexpr int CG_NW = CG_THREADS / 64, CG_NG = CG_THREADS / 16;  // waves, 16-lane row groups
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

// Grid-wide fixed-order sums of NV values in two halves, so that the caller can issue loads between them:
//   cg_post:     the workgroup's partial sums go out (one barrier for all NV values); workgroup 0 also waits for every
//                mailbox and publishes the totals
//   cg_collect:  wait for the totals.  Returns false when a wait timed out (the grid is then abandoned).
// sh: [4][CG_NV * 8] doubles (two stages x exchange parity), s_err: shared flag raised by any thread whose wait timed out.
struct CgBox {
  unsigned long long *row, *row2, *total, *total2;
};
__device__ __forceinline__ CgBox cg_box_of(unsigned long long *box, int e) {
  CgBox b;
  b.row = box + (size_t)(e % CG_RING) * CG_NV * CG_MAX_WG;
  b.row2 = box + (size_t)((e + 2) % CG_RING) * CG_NV * CG_MAX_WG;
  b.total = box + (size_t)CG_RING * CG_NV * CG_MAX_WG + (size_t)(e % CG_RING) * CG_NV;
  b.total2 = box + (size_t)CG_RING * CG_NV * CG_MAX_WG + (size_t)((e + 2) % CG_RING) * CG_NV;
  return b;
}
__device__ __forceinline__ double cg_sum8(const double *w) {  // the per-wave sums of the block in a fixed order
  if (CG_NW == 4) return (w[0] + w[1]) + (w[2] + w[3]);
  return ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
}

template <int NV>
__device__ __forceinline__ void cg_post(const double (&part)[NV], unsigned long long *box, int e, int nwg, double (*sh)[CG_NV * 8], int *s_err, bool drop = false) {
  const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const CgBox bx = cg_box_of(box, e);
  double *buf = sh[2 * (e & 1)], *buf0 = sh[2 * (e & 1) + 1];
  double ws[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ws[v] = gx_wave_sum(part[v]);
  // every data store of this wave (h, d: write-through) is acknowledged before the barrier behind which the sums go out
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((tid & 63) == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) buf[v * 8 + wave] = ws[v];
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (!drop) gx_post(bx.row + (size_t)v * CG_MAX_WG + wg, cg_sum8(buf + v * 8));  // drop: fault injection (NSX_GX_DROP_WG)
#pragma unroll
    for (int v = 0; v < CG_NV; ++v) gx_clear(bx.row2 + (size_t)v * CG_MAX_WG + wg);
  }
  if (wg == 0) {
    int lerr = 0;
    double a[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) a[v] = 0.0;
    for (int q = tid; q < nwg; q += CG_THREADS) {
      double m[NV];
      gx_wait_n<NV>(bx.row + q, CG_MAX_WG, m, &lerr);  // the NV words of mailbox q polled together
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v] += m[v];
    }
    if (lerr) *s_err = 1;
#pragma unroll
    for (int v = 0; v < NV; ++v) ws[v] = gx_wave_sum(a[v]);
    if ((tid & 63) == 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) buf0[v * 8 + wave] = ws[v];
    }
    __syncthreads();
    if (tid == 0 && !*s_err) {  // a total built on a timed-out mailbox never goes out
#pragma unroll
      for (int v = 0; v < NV; ++v) gx_post(bx.total + v, cg_sum8(buf0 + v * 8));
    }
    if (tid < CG_NV) gx_clear(bx.total2 + tid);
  }
}
template <int NV>
__device__ __forceinline__ bool cg_collect(double (&tot)[NV], unsigned long long *box, int e, double *bc, int *s_err) {
  if (threadIdx.x == 0) {
    int lerr = 0;
    double m[NV];
    gx_wait_n<NV>(cg_box_of(box, e).total, 1, m, &lerr);
#pragma unroll
    for (int v = 0; v < NV; ++v) bc[v] = m[v];
    if (lerr) *s_err = 1;
  }
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) tot[v] = bc[v];
  return *s_err == 0;
}
// ---- packed operator stream ------------------------------------------------------------------------------------------
// A latency-bound kernel must not chase pointers: rowptr -> colind -> x is three dependent trips through memory per row.
// At setup (build_cg_plan) the rows of every Schur block are laid out as slabs of 256 slots — slab (round r, chunk c) holds
// entry 16c + lane of row 16r + grp for thread (grp, lane) — so every value load has an address that depends on the loop
// counter alone and many slabs are in flight at once.  Columns are 16-bit indices into the block's list of unique columns,
// whose d entries are staged in LDS once per iteration (~400 gathers per block instead of ~7000).
constexpr int CG_PF = 8;         // slabs per register set (two sets in flight)
constexpr int CG_MAX_UCOLS = 1024;
constexpr int CG_LPOOL = 6656;     // entries of one block's rows held in LDS by the LRES variant (8 + 2 bytes each)

__global__ void k_cg_pack(int64_t n_slots, const int32_t *__restrict__ src, const double *__restrict__ sv, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_slots) out[i] = src[i] >= 0 ? sv[src[i]] : 0.0;
}

// hv_q = sum_j S_qj xst[lidx]  for the rows of this block; 16 lanes per row, result in hvs[].  The first register set is
// loaded by cg_spmv_prefetch, which the caller issues early (the values do not depend on the exchange in front of the product).
#define NSX_CG_LOAD(V, L, S0)                                   \
  _Pragma("unroll") for (int k = 0; k < CG_PF; ++k) {           \
    const int s_ = (S0) + k;                                    \
    const bool ok_ = s_ < s1;                                   \
    V[k] = ok_ ? sval[(size_t)s_ * 256 + tid] : 0.0;            \
    L[k] = ok_ ? (int)slidx[(size_t)s_ * 256 + tid] : 0;        \
  }
__device__ __forceinline__ void cg_spmv_prefetch(int s0, int s1, const double *__restrict__ sval, const uint16_t *__restrict__ slidx, int tid,
                                                 double (&va)[CG_PF], int (&la)[CG_PF]) {
  NSX_CG_LOAD(va, la, s0)
}
__device__ __forceinline__ void cg_block_spmv(int s0, int s1, const double *__restrict__ sval, const uint16_t *__restrict__ slidx,
                                              const int32_t *__restrict__ sinfo, const double *xst, double *hvs, int tid /* 0..255 within the half */,
                                              double (&va)[CG_PF], int (&la)[CG_PF]) {
  const int grp = tid >> 4, lane = tid & 15;
  double vb[CG_PF];
  int lb[CG_PF];
  double acc = 0.0;
#define NSX_CG_USE(V, L, S0)                                    \
  _Pragma("unroll") for (int k = 0; k < CG_PF; ++k) {           \
    const int s_ = (S0) + k;                                    \
    if (s_ < s1) {                                              \
      acc += V[k] * xst[L[k]];                                  \
      const int inf = sinfo[s_];                                \
      if (inf & 0x8000) {                                       \
        const double r_ = cg_group_sum<16>(acc);                \
        if (lane == 0) hvs[(inf & 0x7fff) * 16 + grp] = r_;     \
        acc = 0.0;                                              \
      }                                                         \
    }                                                           \
  }
  for (int sb = s0; sb < s1; sb += 2 * CG_PF) {
    NSX_CG_LOAD(vb, lb, sb + CG_PF)
    NSX_CG_USE(va, la, sb)
    NSX_CG_LOAD(va, la, sb + 2 * CG_PF)
    NSX_CG_USE(vb, lb, sb + CG_PF)
  }
#undef NSX_CG_USE
}
#undef NSX_CG_LOAD

// RPG  0: the block inverses are streamed in every iteration; 6 / 8: blocks of at most 96 / 128 rows, inverses in registers.
// LRES true (RPG > 0 only): the block's rows of negative_S_tilde stay in LDS for the whole solve as well (values + 16-bit local
//      columns, CSR order, at most CG_LPOOL entries: 65 KB of the CU's 160 KB, two workgroups per CU) -- an iteration then reads
//      nothing from HBM but the neighbours' d and h.  Same lanes, same entries per lane, same sums as the slab stream: bit-identical.
template <int RPG, bool LRES>
__global__ __launch_bounds__(CG_THREADS) void k_cg_schur(const int32_t *__restrict__ bptr, const int32_t *__restrict__ u_ptr,
                                                  const int32_t *__restrict__ u_cols, const int32_t *__restrict__ s_ptr,
                                                  const double *__restrict__ sval, const uint16_t *__restrict__ slidx,
                                                  const int32_t *__restrict__ sinfo, const int64_t *__restrict__ dn_off,
                                                  const double *__restrict__ P, const double *__restrict__ b, double *x, double *D0, double *D1,
                                                  double *H, double rtol, int maxiter, unsigned long long *box, unsigned long long *box_other,
                                                  double *pub_vals, unsigned long long *pub_flag, unsigned long long seq, int *err_dev, int drop_wg,
                                                  const int32_t *__restrict__ a_rowptr, const double *__restrict__ a_val,
                                                  const uint16_t *__restrict__ a_lidx) {
  static_assert(!LRES || RPG > 0, "the LDS-resident operator needs the register-resident block inverses");
  constexpr int MAXB = RPG > 0 ? 16 * RPG : CG_MAXB;  // rows of the largest block this instantiation serves
  __shared__ double gs[MAXB], hs[MAXB], ds[MAXB], hvs[MAXB], xs[MAXB], xst[CG_MAX_UCOLS];
  __shared__ double sh[4][CG_NV * 8], bc[CG_NV];
  __shared__ int s_err;
  __shared__ double lval[LRES ? CG_LPOOL : 1];
  __shared__ uint16_t lidx[LRES ? CG_LPOOL : 2];
  __shared__ int lrow[LRES ? MAXB + 1 : 1];
  const int wg = blockIdx.x, tid = threadIdx.x, nwg = gridDim.x;
  if (tid == 0) s_err = 0;
  // leave the other region empty for the next launch (stream order makes this visible to it)
  for (size_t q = (size_t)wg * CG_THREADS + tid; q < CG_REGION; q += (size_t)nwg * CG_THREADS) box_other[q] = GX_EMPTY;
  const int r0 = bptr[wg], nb = bptr[wg + 1] - r0;
  const int u0 = u_ptr[wg], nu = u_ptr[wg + 1] - u0;
  // slabs of this block: the even 16-row rounds first (half 0 of the workgroup), then the odd ones (half 1)
  const int half = tid >> 8, ht = tid & 255;
  const int s0 = s_ptr[3 * wg + half], s1 = s_ptr[3 * wg + (CG_THREADS == 512 ? half + 1 : 2)];
  const double *Pb = P + dn_off[wg];
  const int grp = tid >> 4, lane = tid & 15;  // CG_NG row groups of 16 lanes in the dense product
  const bool own = tid < nb;
  int e = 0;
  // the block's unique columns stay in registers for the whole solve (nu <= CG_MAX_UCOLS)
  constexpr int NU = CG_MAX_UCOLS / CG_THREADS;
  int ucol[NU];
#pragma unroll
  for (int k = 0; k < NU; ++k) ucol[k] = tid + CG_THREADS * k < nu ? u_cols[u0 + tid + CG_THREADS * k] : -1;
  if (tid < MAXB) gs[tid] = 0.0;  // rows beyond nb stay 0: the dense product reads gs[] unguarded

  // h = P_b g on the block, 16 lanes per row (32 rows at a time); result in hs[] and (write-through) in H.  Blocks of up to
  // 128 rows (the usual case) keep the loads of two rows in flight, and the first two rows are fetched by prefetch_P() while
  // the exchange in front of the product is still in the air (P does not depend on its result); larger blocks use one
  // register set of twice the width.
  double pa[8], pb[8];
  auto fetch = [&](int q, double(&pv)[8]) {
    const bool live = q < nb;
    const double *prow = Pb + (size_t)(live ? q : 0) * nb;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = lane + 16 * c;
      pv[c] = (live && j < nb) ? prow[j] : 0.0;
    }
  };
  // RPG > 0: every block has at most 16 * RPG rows and its explicit inverse P_b stays in REGISTERS for the whole solve (row group
  // g holds rows g, g + 16, ...: RPG rows x RPG values per lane, 72 VGPRs at 96 rows), loaded once; the iteration then streams the
  // operator slabs only.  P_b and negative_S_tilde change once per time step at most, a solve runs ~20 iterations and a step ~25
  // solves: re-reading the 34 MB of inverses in every iteration was half of the kernel's traffic.
  constexpr bool PRES = RPG > 0;
  double pr[PRES ? RPG : 1][PRES ? RPG : 1];
  if constexpr (PRES) {
#pragma unroll
    for (int r = 0; r < RPG; ++r) {
      const int q = grp + CG_NG * r;
      const double *prow = Pb + (size_t)(q < nb ? q : 0) * nb;
#pragma unroll
      for (int c = 0; c < RPG; ++c) {
        const int j = lane + 16 * c;
        pr[r][c] = (q < nb && j < nb) ? prow[j] : 0.0;
      }
    }
  }
  auto prefetch_P = [&]() {
    if constexpr (!PRES) {
      if (nb <= 128) {
        fetch(grp, pa);
        fetch(grp + CG_NG, pb);
      }
    }
  };
  auto apply_P = [&]() {  // prefetch_P() has been called
    if constexpr (PRES) {
#pragma unroll
      for (int r = 0; r < RPG; ++r) {
        const int q = grp + CG_NG * r;
        if (q < nb) {
          double acc = 0.0;
#pragma unroll
          for (int c = 0; c < RPG; ++c) acc += pr[r][c] * gs[lane + 16 * c];
          acc = cg_group_sum<16>(acc);
          if (lane == 0) hs[q] = acc;
        }
      }
    } else if (nb <= 128) {
      auto use = [&](int q, const double(&pv)[8]) {
        if (q >= nb) return;
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc += pv[c] * gs[lane + 16 * c];
        acc = cg_group_sum<16>(acc);
        if (lane == 0) hs[q] = acc;
      };
      for (int q = grp; q < nb; q += 2 * CG_NG) {
        use(q, pa);
        fetch(q + 2 * CG_NG, pa);
        use(q + CG_NG, pb);
        fetch(q + 3 * CG_NG, pb);
      }
    } else {
      for (int q = grp; q < nb; q += CG_NG) {
        const double *prow = Pb + (size_t)q * nb;
        double pv[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int j = lane + 16 * c;
          pv[c] = j < nb ? prow[j] : 0.0;
        }
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc += pv[c] * gs[lane + 16 * c];
        acc = cg_group_sum<16>(acc);
        if (lane == 0) hs[q] = acc;
      }
    }
    __syncthreads();
    if (own) st_agent(H + r0 + tid, hs[tid]);
  };

  double va[LRES ? 1 : CG_PF];
  int la[LRES ? 1 : CG_PF];
  if constexpr (LRES) {
    const int p0 = a_rowptr[r0], nz = a_rowptr[r0 + nb] - p0;  // nz <= CG_LPOOL (checked on the host)
    for (int q = tid; q < nz; q += CG_THREADS) {
      lval[q] = a_val[p0 + q];
      lidx[q] = a_lidx[p0 + q];
    }
    if (tid <= nb) lrow[tid] = a_rowptr[r0 + tid] - p0;
  }
  // hvs = (negative_S_tilde restricted to the block's rows) * xst: row q by lane group q % 16, entry e by lane e % 16 -- from LDS
  // (LRES) or from the slab stream, whose first register set the caller has requested with prefetch_A()
  auto prefetch_A = [&]() {
    if constexpr (!LRES) cg_spmv_prefetch(s0, s1, sval, slidx, ht, va, la);
  };
  auto block_spmv = [&]() {
    if constexpr (LRES) {
#pragma unroll
      for (int r = 0; r < RPG; ++r) {
        const int q = grp + CG_NG * r;
        if (q < nb) {
          const int e1 = lrow[q + 1];
          double acc = 0.0;
          for (int e = lrow[q] + lane; e < e1; e += 16) acc += lval[e] * xst[lidx[e]];
          acc = cg_group_sum<16>(acc);
          if (lane == 0) hvs[q] = acc;
        }
      }
    } else {
      cg_block_spmv(s0, s1, sval, slidx, sinfo, xst, hvs, ht, va, la);
    }
  };

  // ---- g = A x - b ; h = P g ; sums g.g, b.b, g.h
  double bi = 0.0;
  if (own) {
    xs[tid] = x[r0 + tid];
    bi = b[r0 + tid];
  }
#pragma unroll
  for (int k = 0; k < NU; ++k)
    if (ucol[k] >= 0) xst[tid + CG_THREADS * k] = x[ucol[k]];
  prefetch_A();
  __syncthreads();
  block_spmv();
  __syncthreads();
  prefetch_P();
  if (own) gs[tid] = hvs[tid] - bi;
  __syncthreads();
  apply_P();
  double tot3[3];
  {
    const double part[3] = {own ? gs[tid] * gs[tid] : 0.0, own ? bi * bi : 0.0, own ? gs[tid] * hs[tid] : 0.0};
    cg_post<3>(part, box, e, nwg, sh, &s_err, wg == drop_wg);
    prefetch_A();  // the operator values of the first iteration ride on the exchange
    if (!cg_collect<3>(tot3, box, e++, bc, &s_err)) {
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
    // ---- A: h = A d, d = -h(old) in the first iteration, beta d(old) - h(old) afterwards, evaluated for the block's columns
#pragma unroll
    for (int k = 0; k < NU; ++k)
      if (ucol[k] >= 0) {
        const int j = ucol[k];
        xst[tid + CG_THREADS * k] = it == 1 ? -ld_agent(H + j) : __builtin_fma(beta, ld_agent(Dp + j), -ld_agent(H + j));  // the owner's expression
      }
    if (own) {
      const double dcur = it == 1 ? -hs[tid] : __builtin_fma(beta, ds[tid], -hs[tid]);  // d = beta d - h (SolverCG: d.sadd(beta, -1., h))
      ds[tid] = dcur;
      st_agent(Dc + r0 + tid, dcur);
    }
    __syncthreads();
    block_spmv();
    __syncthreads();
    double tot1[1];
    {
      const double part[1] = {own ? ds[tid] * hvs[tid] : 0.0};
      cg_post<1>(part, box, e, nwg, sh, &s_err);
      prefetch_P();  // rides on the exchange's round trips
      if (!cg_collect<1>(tot1, box, e++, bc, &s_err)) {
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
      cg_post<2>(part, box, e, nwg, sh, &s_err);
      prefetch_A();  // next iteration's operator values (wasted once, in the last iteration)
      if (!cg_collect<2>(tot2, box, e++, bc, &s_err)) {
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

// host: slab layout of negative_S_tilde per Schur ILU block (see above).  Built with the ILU schedules; values are
// refreshed by cg_pack_values after every numeric Schur product.
void build_cg_plan(nsx_handle *h) {
  CgPlan &pl = h->cgplan;
  pl.ok = false;
  const IluSchedule &s = h->schedS;
  if (!s.dense || s.max_rows > CG_MAXB) return;  // (distributed handles use the plan through cg_schur_fused: columns may be ghosts)
  const Csr &g = h->gS.host;
  const std::vector<int32_t> &bptr = s.block_ptr_h;
  const int nb = s.n_blocks;
  std::vector<int32_t> u_ptr(nb + 1, 0), u_cols, s_ptr(3 * (size_t)nb + 1, 0), s_info, s_src, tmp;
  std::vector<uint16_t> s_lidx, a_lidx((size_t)g.nnz());
  pl.max_block_nnz = 0;
  for (int b = 0; b < nb; ++b) {
    const int r0 = bptr[b], r1 = bptr[b + 1], n = r1 - r0;
    tmp.assign(g.colind.begin() + g.rowptr[r0], g.colind.begin() + g.rowptr[r1]);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if ((int)tmp.size() > CG_MAX_UCOLS) return;  // plan not applicable: the launch-per-operation solver stays in charge
    u_cols.insert(u_cols.end(), tmp.begin(), tmp.end());
    u_ptr[b + 1] = (int32_t)u_cols.size();
    pl.max_block_nnz = std::max(pl.max_block_nnz, g.rowptr[r1] - g.rowptr[r0]);
    for (int p = g.rowptr[r0]; p < g.rowptr[r1]; ++p) a_lidx[p] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), g.colind[p]) - tmp.begin());
    for (int half = 0; half < 2; ++half) {  // the even rounds (first half of the workgroup), then the odd ones
      s_ptr[3 * (size_t)b + half] = (int32_t)s_info.size();
      for (int r = half; r * 16 < n; r += 2) {
        int maxlen = 0;
        for (int q = 16 * r; q < std::min(n, 16 * r + 16); ++q) maxlen = std::max(maxlen, g.rowptr[r0 + q + 1] - g.rowptr[r0 + q]);
        const int chunks = std::max(1, (maxlen + 15) / 16);
        for (int c = 0; c < chunks; ++c) {
          s_info.push_back(r | (c == chunks - 1 ? 0x8000 : 0));
          for (int t = 0; t < 256; ++t) {
            const int q = 16 * r + (t >> 4), e = 16 * c + (t & 15);
            int32_t src = -1;
            uint16_t li = 0;
            if (q < n && g.rowptr[r0 + q] + e < g.rowptr[r0 + q + 1]) {
              src = g.rowptr[r0 + q] + e;
              li = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), g.colind[src]) - tmp.begin());
            }
            s_src.push_back(src);
            s_lidx.push_back(li);
          }
        }
      }
    }
    s_ptr[3 * (size_t)b + 2] = (int32_t)s_info.size();
  }
  pl.n_slots = (int64_t)s_src.size();
  pl.u_ptr.upload(u_ptr, h->stream);
  pl.u_cols.upload(u_cols, h->stream);
  pl.s_ptr.upload(s_ptr, h->stream);
  pl.s_info.upload(s_info, h->stream);
  pl.s_src.upload(s_src, h->stream);
  pl.s_lidx.upload(s_lidx, h->stream);
  pl.a_lidx.upload(a_lidx, h->stream);
  pl.s_val.alloc((size_t)pl.n_slots);
  pl.ok = true;
  pl.values_current = false;
  if (getenv("NSX_DEBUG"))
    fprintf(stderr, "[nsx] persistent Schur CG plan: %d blocks, %lld slabs (fill %.2f), unique columns per block avg %.0f\n", nb,
            (long long)s_info.size(), (double)g.nnz() / (double)std::max<int64_t>(1, pl.n_slots), (double)u_cols.size() / std::max(1, nb));
}

// after schur_numeric: the packed copy of the values
void cg_pack_values(nsx_handle *h) {
  CgPlan &pl = h->cgplan;
  if (!pl.ok) return;
  LaunchScope ls(h, "cg_pack", 20.0 * (double)pl.n_slots);
  hipLaunchKernelGGL(k_cg_pack, dim3(cdiv(pl.n_slots, 256)), dim3(256), 0, h->stream, pl.n_slots, pl.s_src.p, h->vSchur.p, pl.s_val.p);
  pl.values_current = true;
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
  int per_cu_res[2] = {0, 0};
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_schur<0, false>, CG_THREADS, 0));
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_res[0], k_cg_schur<6, false>, CG_THREADS, 0));
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_res[1], k_cg_schur<8, false>, CG_THREADS, 0));
  h->cg_max_wg_res[0] = std::min(CG_MAX_WG, per_cu_res[0] * cus);
  h->cg_max_wg_res[1] = std::min(CG_MAX_WG, per_cu_res[1] * cus);
  int per_cu_lres[2] = {0, 0};
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_lres[0], k_cg_schur<6, true>, CG_THREADS, 0));
  HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_lres[1], k_cg_schur<8, true>, CG_THREADS, 0));
  h->cg_max_wg_lres[0] = std::min(CG_MAX_WG, per_cu_lres[0] * cus);
  h->cg_max_wg_lres[1] = std::min(CG_MAX_WG, per_cu_lres[1] * cus);
  h->cg_box.alloc(2 * CG_REGION + 2);
  HIP_CHECK(hipMemsetAsync(h->cg_box.p, 0xff, 2 * CG_REGION * sizeof(unsigned long long), h->stream));
  HIP_CHECK(hipMemsetAsync(h->cg_box.p + 2 * CG_REGION, 0, 2 * sizeof(unsigned long long), h->stream));
  h->cg_max_wg = std::min(CG_MAX_WG, per_cu * cus);
  if (getenv("NSX_DEBUG"))
    fprintf(stderr, "[nsx] persistent Schur CG: %d CUs x %d resident workgroups, grid <= %d (block inverses in registers: <= 96 rows x %d, <= 128 rows x %d; operator in LDS as well: x %d, x %d)\n", cus, per_cu,
            h->cg_max_wg, per_cu_res[0], per_cu_res[1], per_cu_lres[0], per_cu_lres[1]);
}

// CG on negative_S_tilde with the explicit block inverses as preconditioner, one launch.  Returns false when the launch-per-
// operation path has to be used instead (distributed run, blocks too large or too many, persistent kernels disabled).
bool cg_schur_persistent(nsx_handle *h, double *x, const double *b, double rtol, int maxiter, int *steps, double *last, int *status) {
  const IluSchedule &s = h->schedS;
  const CgPlan &pl = h->cgplan;
  if (h->comm || !s.dense || s.max_rows > CG_MAXB || !pl.ok || !pl.values_current) return false;
  cg_setup(h);
  if (h->cg_max_wg == 0 || s.n_blocks > h->cg_max_wg) return false;
  const bool pres_ok = !(getenv("NSX_CG_PRES") && atoi(getenv("NSX_CG_PRES")) == 0);  // read per solve: the tests switch it
  // rows per group of the register-resident variant: the smallest that holds the largest block, if its grid is resident
  const int rpg = !pres_ok ? 0 : (s.max_rows <= 96 && s.n_blocks <= h->cg_max_wg_res[0]) ? 6 : (s.max_rows <= 128 && s.n_blocks <= h->cg_max_wg_res[1]) ? 8 : 0;
  const bool pres = rpg > 0;
  const bool lres_ok = !(getenv("NSX_CG_LRES") && atoi(getenv("NSX_CG_LRES")) == 0);
  // ... and the block's rows of the operator in LDS, if every block fits the pool and that grid is resident too
  const bool lres = lres_ok && pres && pl.max_block_nnz <= CG_LPOOL && s.n_blocks <= h->cg_max_wg_lres[rpg == 6 ? 0 : 1];
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
#define NSX_CG_GO(PRES_, LRES_)                                                                                                                       \
  hipLaunchKernelGGL((k_cg_schur<PRES_, LRES_>), dim3(s.n_blocks), dim3(CG_THREADS), 0, h->stream, s.block_ptr.p, pl.u_ptr.p, pl.u_cols.p, pl.s_ptr.p,  \
                     pl.s_val.p, pl.s_lidx.p, pl.s_info.p, s.dn_off.p, s.dn_P.p, b, x, D0, D1, H, rtol, maxiter, box, box_other, pub_vals, pub_flag, seq, \
                     err_dev, h->gx_drop_wg, h->gS.rowptr.p, h->vSchur.p, pl.a_lidx.p)
    if (rpg == 6 && lres) NSX_CG_GO(6, true); else if (rpg == 8 && lres) NSX_CG_GO(8, true);
    else if (rpg == 6) NSX_CG_GO(6, false); else if (rpg == 8) NSX_CG_GO(8, false); else NSX_CG_GO(0, false);
#undef NSX_CG_GO
    if (getenv("NSX_DEBUG") && (h->cg_resident != pres || h->cg_lds_resident != lres || !h->cg_variant_said)) {
      fprintf(stderr, "[nsx] Schur CG variant: block inverses in registers %d (rows per lane group %d), operator in LDS %d (largest block: %d entries)\n", (int)pres, rpg,
              (int)lres, (int)pl.max_block_nnz);
      h->cg_variant_said = true;
    }
    h->cg_resident = pres;
    h->cg_lds_resident = lres;
  }
  h->cg_parity ^= 1;
  h->cg_last_path = 2;
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
    h->n_persistent_fallbacks++;
    fprintf(stderr, "[nsx] warning: the persistent Schur CG timed out (grid not co-resident): this handle uses one launch per operation from now on\n");
    return false;
  }
  *steps = (int)h->pub_host[S_CGP];
  *last = h->pub_host[S_CGP + 1];
  *status = st;
  // algorithmic bytes: per iteration the matrix (12 B / entry) and the block inverses once, plus the vectors
  // (block inverses resident in registers: read once per solve)
  // (operator resident in LDS: read once per solve, too)
  if (pe) pe->bytes += (lres ? 1.0 : (double)(*steps + 1)) * 10.0 * h->gS.nnz() + (double)(*steps + 1) * 48.0 * n + (pres ? 1.0 : (double)(*steps + 1)) * 8.0 * (double)s.dn_entries;
  return true;
}


// ---- the same solve in TWO launches per iteration: distributed runs, and one GPU with more Schur blocks than resident workgroups ----
// With a communicator the solve cannot stay in one launch: the neighbours' d comes through a ghost exchange and the two sums of an
// iteration through collectives the host enqueues.  The launch-per-operation solver (nsx_solve.hip: cg) pays five to six kernels
// per iteration; here the persistent kernel's two phases are cut at its two exchanges and nothing else:
//     k_cgd_A(it):  beta = g.h / g.h(old) from the all-reduced partial sums ; d = beta d(old) - h evaluated while gathering the
//                   block's columns (ghost columns: the owner computed the same expression in its pack kernel and sent it) ;
//                   hv = S d on the block's rows (slab stream) ; partial d.hv ; workgroup 0 publishes |g|^2 of the iteration before
//     -- all-reduce of the d.hv partials --
//     k_cgd_B(it):  alpha = g.h / d.hv ; x += alpha d ; g += alpha hv ; h = P_b g (explicit block inverse) ; partials g.g, g.h
//     -- all-reduce of the g.g / g.h partials (one collective) --
// Same lanes, same entries per lane and same block-local sums as k_cg_schur; the grid-wide sums are formed from one partial sum
// per Schur block in block order (every rank pads to CGD_PARTS entries, the collective adds element-wise, every workgroup adds
// the CGD_PARTS numbers in the same fixed order), so the result does not depend on timing -- and differs from the one-GPU kernel's
// by the rounding of those sums.  The host waits for the residual published by k_cgd_A(it + 1), i.e. the next direction and product
// are enqueued speculatively (wasted once per solve, by every rank alike: the collective sequences stay matched).
// More than CGD_PARTS Schur blocks on a rank (the 10.6 M-DoF mesh on one GPU has 4 833, on two 2 417): the kernels leave their per-block
// partial sums in a raw array and a small launch (k_cgd_fold) adds them in groups of ceil(n_blocks / CGD_PARTS) consecutive blocks, in
// block order, into the CGD_PARTS-entry array everything else works with -- the collective, the consumers and their fixed-order sums
// stay as they are, whatever the block count (up to CGD_FOLD_MAX per group: 8 192 blocks, ~17 M DoF per GPU).
constexpr int CGD_PARTS = 1024, CGD_FOLD_MAX = 8;
enum { CGD_DH = 0, CGD_P0 = CGD_PARTS, CGD_BB = 3 * CGD_PARTS, CGD_P1 = 4 * CGD_PARTS, CGD_TOTAL = 6 * CGD_PARTS };  // P0 | BB contiguous: one collective after the set-up kernel

// fixed-order sum of CGD_PARTS numbers by the 256 threads of the block (every thread gets it); sh: 4 doubles of its own per call site
__device__ __forceinline__ double cgd_sum(const double *__restrict__ p, double *sh) {
  double a = 0.0;
#pragma unroll
  for (int k = 0; k < CGD_PARTS / 256; ++k) a += p[threadIdx.x + 256 * k];
  return gx_block_sum(a, sh);
}
// The collectives add the ranks' partial-sum arrays element by element IN PLACE: entries beyond this rank's own block count hold the
// other ranks' sums afterwards, and would be added again the next time the array goes out.  The last workgroup clears them (n_arrays
// consecutive arrays of CGD_PARTS entries) in every launch that fills the array.
__device__ __forceinline__ void cgd_clear_tail(double *arrays, int n_arrays) {
  if (blockIdx.x != gridDim.x - 1 || (int)gridDim.x > CGD_PARTS) return;  // (more blocks than entries: the fold kernel writes every entry)
  for (int a = 0; a < n_arrays; ++a)
    for (int q = (int)gridDim.x + (int)threadIdx.x; q < CGD_PARTS; q += CG_THREADS) arrays[(size_t)a * CGD_PARTS + q] = 0.0;
}
// dst[a][g] = raw[a][g * grp] + ... + raw[a][g * grp + grp - 1] (block order; groups beyond the last block: 0), a < gridDim.x / 4
__global__ __launch_bounds__(256) void k_cgd_fold(int nblk, int grp, const double *__restrict__ raw, double *__restrict__ dst) {
  const int a = blockIdx.x >> 2, g = (blockIdx.x & 3) * 256 + threadIdx.x;
  const double *r = raw + (size_t)a * nblk;
  double s_ = 0.0;
  for (int k = 0; k < grp; ++k) {
    const int q = g * grp + k;
    if (q < nblk) s_ += r[q];
  }
  dst[(size_t)a * CGD_PARTS + g] = s_;
}
__device__ __forceinline__ void cgd_dense_apply(const double *__restrict__ Pb, int nb, const double *gs, double *hs, int tid) {
  const int grp = tid >> 4, lane = tid & 15;
  for (int q = grp; q < nb; q += CG_NG) {
    const double *prow = Pb + (size_t)q * nb;
    double pv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int j = lane + 16 * c;
      pv[c] = j < nb ? prow[j] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) acc += pv[c] * gs[lane + 16 * c];
    acc = cg_group_sum<16>(acc);
    if (lane == 0) hs[q] = acc;
  }
}

// set-up: g = S x - b ; h = P g ; partials g.g, g.h (P0) and b.b (BB)
__global__ __launch_bounds__(CG_THREADS) void k_cgd_init(const int32_t *__restrict__ bptr, const int32_t *__restrict__ u_ptr, const int32_t *__restrict__ u_cols,
                                                   const int32_t *__restrict__ s_ptr, const double *__restrict__ sval, const uint16_t *__restrict__ slidx,
                                                   const int32_t *__restrict__ sinfo, const int64_t *__restrict__ dn_off, const double *__restrict__ P,
                                                   const double *__restrict__ b, const double *__restrict__ x, double *__restrict__ G, double *__restrict__ H,
                                                   double *__restrict__ out, int ostride) {  // out: parts + CGD_P0 (stride CGD_PARTS), or the raw array in front of k_cgd_fold
  __shared__ double gs[CG_MAXB], hs[CG_MAXB], hvs[CG_MAXB], xst[CG_MAX_UCOLS], sh[3][4];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int r0 = bptr[wg], nb = bptr[wg + 1] - r0, u0 = u_ptr[wg], nu = u_ptr[wg + 1] - u0;
  const int s0 = s_ptr[3 * wg], s1 = s_ptr[3 * wg + 2];
  const bool own = tid < nb;
  double va[CG_PF];
  int la[CG_PF];
  cg_spmv_prefetch(s0, s1, sval, slidx, tid, va, la);
  for (int k = tid; k < nu; k += CG_THREADS) xst[k] = x[u_cols[u0 + k]];
  gs[tid] = 0.0;
  const double bi = own ? b[r0 + tid] : 0.0;
  __syncthreads();
  cg_block_spmv(s0, s1, sval, slidx, sinfo, xst, hvs, tid, va, la);
  __syncthreads();
  if (own) {
    gs[tid] = hvs[tid] - bi;
    G[r0 + tid] = gs[tid];
  }
  __syncthreads();
  cgd_dense_apply(P + dn_off[wg], nb, gs, hs, tid);
  __syncthreads();
  if (own) H[r0 + tid] = hs[tid];
  const double gg = gx_block_sum(own ? gs[tid] * gs[tid] : 0.0, sh[0]);
  const double gh = gx_block_sum(own ? gs[tid] * hs[tid] : 0.0, sh[1]);
  const double bb = gx_block_sum(bi * bi, sh[2]);
  if (tid == 0) {
    out[wg] = gg;
    out[(size_t)ostride + wg] = gh;
    out[2 * (size_t)ostride + wg] = bb;
  }
  cgd_clear_tail(out, 3);
}

// the direction of iteration `it` for the nodes a neighbour needs (the communication stream's pack kernel): the owner's expression
__global__ __launch_bounds__(256) void k_cgd_pack(int n_send, const int32_t *__restrict__ idx, int it, const double *__restrict__ parts, const double *__restrict__ H,
                                                  const double *__restrict__ Dp, double *__restrict__ buf) {
  __shared__ double sh[2][4];
  double beta = 0.0;
  if (it >= 2) {
    const double *cur = parts + (((it - 1) & 1) ? CGD_P1 : CGD_P0), *old = parts + (((it - 1) & 1) ? CGD_P0 : CGD_P1);
    beta = cgd_sum(cur + CGD_PARTS, sh[0]) / cgd_sum(old + CGD_PARTS, sh[1]);
  }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n_send) {
    const int j = idx[k];
    buf[k] = it == 1 ? -H[j] : __builtin_fma(beta, Dp[j], -H[j]);
  }
}

__global__ __launch_bounds__(CG_THREADS) void k_cgd_A(const int32_t *__restrict__ bptr, const int32_t *__restrict__ u_ptr, const int32_t *__restrict__ u_cols,
                                                const int32_t *__restrict__ s_ptr, const double *__restrict__ sval, const uint16_t *__restrict__ slidx,
                                                const int32_t *__restrict__ sinfo, int n_own, int it, const double *__restrict__ parts, const double *__restrict__ H,
                                                const double *__restrict__ Dp, double *Dc, double *__restrict__ Hv, double *__restrict__ part_dh, double *pub_vals,
                                                unsigned long long *pub_flag, unsigned long long seq) {
  __shared__ double ds[CG_MAXB], hvs[CG_MAXB], xst[CG_MAX_UCOLS], sh[4][4];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int r0 = bptr[wg], nb = bptr[wg + 1] - r0, u0 = u_ptr[wg], nu = u_ptr[wg + 1] - u0;
  const int s0 = s_ptr[3 * wg], s1 = s_ptr[3 * wg + 2];
  const bool own = tid < nb;
  double va[CG_PF];
  int la[CG_PF];
  cg_spmv_prefetch(s0, s1, sval, slidx, tid, va, la);
  // the sums of the iteration before (all-reduced partial sums): |g|^2 and g.h ; beta = g.h / g.h(old)
  const double *cur = parts + (((it - 1) & 1) ? CGD_P1 : CGD_P0), *old = parts + (((it - 1) & 1) ? CGD_P0 : CGD_P1);
  const double gg = cgd_sum(cur, sh[0]), gh = cgd_sum(cur + CGD_PARTS, sh[1]);
  const double gh_old = it >= 2 ? cgd_sum(old + CGD_PARTS, sh[2]) : 1.0;
  const double beta = it >= 2 ? gh / gh_old : 0.0;
  if (wg == 0) {  // the residual of iteration it - 1 (and |b|^2 behind the set-up kernel) for the host's SolverControl::check
    const double bb = it == 1 ? cgd_sum(parts + CGD_BB, sh[3]) : 0.0;
    if (tid == 0) {
      __hip_atomic_store(pub_vals + 0, gg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(pub_vals + 1, bb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(pub_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  for (int k = tid; k < nu; k += CG_THREADS) {
    const int j = u_cols[u0 + k];
    xst[k] = j >= n_own ? Dc[j] : (it == 1 ? -H[j] : __builtin_fma(beta, Dp[j], -H[j]));  // ghost column: received ; owned: the owner's expression
  }
  if (own) {
    const double dcur = it == 1 ? -H[r0 + tid] : __builtin_fma(beta, Dp[r0 + tid], -H[r0 + tid]);  // d = beta d - h (SolverCG: d.sadd(beta, -1., h))
    ds[tid] = dcur;
    Dc[r0 + tid] = dcur;
  }
  __syncthreads();
  cg_block_spmv(s0, s1, sval, slidx, sinfo, xst, hvs, tid, va, la);
  __syncthreads();
  if (own) Hv[r0 + tid] = hvs[tid];
  const double dh = gx_block_sum(own ? ds[tid] * hvs[tid] : 0.0, sh[3]);
  if (tid == 0) part_dh[wg] = dh;
  cgd_clear_tail(part_dh, 1);
}

__global__ __launch_bounds__(CG_THREADS) void k_cgd_B(const int32_t *__restrict__ bptr, const int64_t *__restrict__ dn_off, const double *__restrict__ P, int it,
                                                double *__restrict__ parts, double *__restrict__ x, double *__restrict__ G, double *__restrict__ H,
                                                const double *__restrict__ Dc, const double *__restrict__ Hv, double *__restrict__ out, int ostride) {
  __shared__ double gs[CG_MAXB], hs[CG_MAXB], sh[4][4];
  const int wg = blockIdx.x, tid = threadIdx.x;
  const int r0 = bptr[wg], nb = bptr[wg + 1] - r0;
  const bool own = tid < nb;
  const double *prev = parts + (((it - 1) & 1) ? CGD_P1 : CGD_P0);
  double *mine = out;  // parts + ((it & 1) ? CGD_P1 : CGD_P0) with stride CGD_PARTS, or the raw array in front of k_cgd_fold
  const double dh = cgd_sum(parts + CGD_DH, sh[0]), gh = cgd_sum(prev + CGD_PARTS, sh[1]);
  const double alpha = gh / dh;
  gs[tid] = 0.0;
  __syncthreads();
  if (own) {
    x[r0 + tid] += alpha * Dc[r0 + tid];
    const double g = G[r0 + tid] + alpha * Hv[r0 + tid];
    gs[tid] = g;
    G[r0 + tid] = g;
  }
  __syncthreads();
  cgd_dense_apply(P + dn_off[wg], nb, gs, hs, tid);
  __syncthreads();
  if (own) H[r0 + tid] = hs[tid];
  const double gg = gx_block_sum(own ? gs[tid] * gs[tid] : 0.0, sh[2]);
  const double ghn = gx_block_sum(own ? gs[tid] * hs[tid] : 0.0, sh[3]);
  if (tid == 0) {
    mine[wg] = gg;
    mine[(size_t)ostride + wg] = ghn;
  }
  cgd_clear_tail(mine, 2);
}

bool cg_schur_fused(nsx_handle *h, double *x, const double *b, double rtol, int maxiter, int *steps, double *last, int *status) {
  const IluSchedule &s = h->schedS;
  const CgPlan &pl = h->cgplan;
  const bool wanted = !(getenv("NSX_CG_FUSED") && atoi(getenv("NSX_CG_FUSED")) == 0);  // read per solve: the tests switch it inside one process
  if (!wanted) return false;  // (one GPU: reached when the persistent kernel cannot run -- more Schur blocks than resident workgroups, i.e. beyond ~1.16 M DoF)
  // whether THIS rank's Schur blocks fit the kernels (dense inverses, <= 256 rows, <= 1024 unique columns per block) depends on its
  // own part of the mesh: the ranks agree once per set of schedules, or one of them would run the launch-per-operation solver's
  // collectives against the others' (found by the 2-process test with 6 virtual ranks per GPU: one rank had a 260-row block)
  if (h->cgd_agreed < 0)
    h->cgd_agreed = comm_agree_all(h, s.dense && s.max_rows <= CG_MAXB && pl.ok && s.n_blocks <= CGD_PARTS * CGD_FOLD_MAX && CG_THREADS == 256) ? 1 : 0;
  if (!h->cgd_agreed) return false;
  if (!pl.values_current) cg_pack_values(h);  // (as the persistent variant's fall-back to the launch-per-operation solver used to allow: a stale packed copy is refreshed, not an error)
  const int n = h->n_p, len = h->len_p, nblk = s.n_blocks;
  // more blocks than entries of a partial-sum array: raw sums + fold (see CGD_PARTS)
  const bool fold = nblk > CGD_PARTS;
  const int fold_grp = cdiv(nblk, CGD_PARTS);
  if (fold && (int)h->cgd_raw.n < 3 * nblk) h->cgd_raw.alloc((size_t)3 * nblk);
  auto fold_into = [&](double *dst, int n_arrays) {
    if (fold) hipLaunchKernelGGL(k_cgd_fold, dim3(4 * n_arrays), dim3(256), 0, h->stream, nblk, fold_grp, h->cgd_raw.p, dst);
  };
  h->cg_last_path = 3;
  if ((int)h->cgd_vec.n < 3 * n + 2 * len) {
    h->cgd_vec.alloc((size_t)3 * n + 2 * len);
    h->cgd_vec.zero(h->stream);
  }
  if (!h->cgd_parts.p) {
    h->cgd_parts.alloc(CGD_TOTAL);
    h->cgd_parts.zero(h->stream);  // entries beyond a rank's blocks stay 0 for good: the collectives add them element-wise
  }
  double *G = h->cgd_vec.p, *H = G + n, *Hv = H + n, *D[2] = {Hv + n, Hv + n + len}, *parts = h->cgd_parts.p;
  double *pub_vals = h->pub_dev + S_CGP;
  unsigned long long *pub_flag = (unsigned long long *)(h->pub_dev + N_SLOTS);
  comm_halo_p(h, x);  // the ghost entries of the initial guess
  {
    LaunchScope ls(h, "cgd_init", 12.0 * h->gS.nnz() + 8.0 * (double)s.dn_entries + 40.0 * n);
    hipLaunchKernelGGL(k_cgd_init, dim3(nblk), dim3(CG_THREADS), 0, h->stream, s.block_ptr.p, pl.u_ptr.p, pl.u_cols.p, pl.s_ptr.p, pl.s_val.p, pl.s_lidx.p,
                       pl.s_info.p, s.dn_off.p, s.dn_P.p, b, x, G, H, fold ? h->cgd_raw.p : parts + CGD_P0, fold ? nblk : CGD_PARTS);
    fold_into(parts + CGD_P0, 3);
  }
  comm_allreduce_partials(h, parts + CGD_P0, 3 * CGD_PARTS);
  int it = 0, conv = 0;
  double tol = 0.0, res = 0.0;
  static const bool trace = getenv("NSX_TRACE") != nullptr;
  for (;;) {
    const int nx = it + 1;
    if (trace && (it < 3 || it % 200 == 0)) fprintf(stderr, "[nsx trace] rank %d: Schur CG iteration %d, residual %.3e, tolerance %.3e\n", h->rank, it, res, tol);
    double *Dp = D[(nx + 1) & 1], *Dc = D[nx & 1];
    // ghost entries of the next direction: the owners evaluate the same expression for the nodes their neighbours need
    if (h->dist && !h->haloP.nbr.empty()) {
      const std::function<void(hipStream_t, double *, const int32_t *, int)> packer = [&](hipStream_t st, double *buf, const int32_t *idx, int n_send) {
        if (n_send) hipLaunchKernelGGL(k_cgd_pack, dim3(cdiv(n_send, 256)), dim3(256), 0, st, n_send, idx, nx, parts, H, Dp, buf);
      };
      comm_halo_begin(h, h->haloP, Dc, 1, &packer);
      comm_halo_finish(h, h->haloP, Dc, 1);
    }
    const unsigned long long seq = ++h->pub_seq;
    {
      LaunchScope ls(h, "cgd_A", 10.0 * h->gS.nnz() + 40.0 * n);
      hipLaunchKernelGGL(k_cgd_A, dim3(nblk), dim3(CG_THREADS), 0, h->stream, s.block_ptr.p, pl.u_ptr.p, pl.u_cols.p, pl.s_ptr.p, pl.s_val.p, pl.s_lidx.p, pl.s_info.p,
                         n, nx, parts, H, Dp, Dc, Hv, fold ? h->cgd_raw.p : parts + CGD_DH, pub_vals, pub_flag, seq);
      fold_into(parts + CGD_DH, 1);
    }
    comm_allreduce_partials(h, parts + CGD_DH, CGD_PARTS);
    wait_published(h, seq);
    if (it == 0) tol = rtol * std::sqrt(h->pub_host[S_CGP + 1]);  // solver_control_S(maxiter, 1e-2 * tmp.l2_norm())
    res = std::sqrt(std::fabs(h->pub_host[S_CGP]));
    conv = res <= tol ? 1 : ((it >= maxiter || res != res) ? 2 : 0);  // SolverControl::check
    if (conv != 0) break;
    it = nx;
    {
      LaunchScope ls(h, "cgd_B", 8.0 * (double)s.dn_entries + 56.0 * n);
      double *mine = parts + ((it & 1) ? CGD_P1 : CGD_P0);
      hipLaunchKernelGGL(k_cgd_B, dim3(nblk), dim3(CG_THREADS), 0, h->stream, s.block_ptr.p, s.dn_off.p, s.dn_P.p, it, parts, x, G, H, Dc, Hv, fold ? h->cgd_raw.p : mine,
                         fold ? nblk : CGD_PARTS);
      fold_into(mine, 2);
    }
    comm_allreduce_partials(h, parts + ((it & 1) ? CGD_P1 : CGD_P0), 2 * CGD_PARTS);
  }
  *steps = it;
  *last = res;
  *status = conv == 1 ? 0 : 1;
  return true;
}

// diagnostics (nsx_persistent_state): mailbox words of both regions that are not empty while nothing is in flight
int cg_dirty_words(nsx_handle *h) {
  if (!h->cg_box.p) return 0;
  std::vector<unsigned long long> w(2 * CG_REGION);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  HIP_CHECK(hipMemcpy(w.data(), h->cg_box.p, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  // the region the next launch will use must be all empty; the other one holds the last solve's sums (its successor clears it)
  const size_t base = (size_t)h->cg_parity * CG_REGION;
  int dirty = 0;
  for (size_t q = 0; q < CG_REGION; ++q) dirty += w[base + q] != GX_EMPTY;
  return dirty;
}

}  // namespace nsx
