// nsx_assemble.hip — element assembly and Dirichlet rows on gfx950.
//
//   NavierStokes::assemble            reference Navier-Stokes/src/NavierStokes3D.cpp:163-356
//   NavierStokes::assemble_time_step  reference Navier-Stokes/src/NavierStokes3D.cpp:361-544
//   (2D: NavierStokes2D.cpp:164-527; convergence: Convergence3D.cpp:187-581)
//
// Design (MI355X-first, not the reference's serial cell loop):
//   * one CELL PER LANE (64 cells per wave): the P2/P1 simplex element is a 10x10 (6x6) scalar block — far too
//     small for a workgroup, and every velocity-velocity term is delta_cd (x) scalar, so only the scalar block
//     is computed (the reference loops over 34x34 x n_q, NS3D.cpp:442-463).  Shape tables are read with
//     wave-uniform addresses (scalar loads), u_n is gathered once per cell, all loops are fully unrolled.
//   * local matrices go to HBM as SoA planes [entry][cell] (coalesced 512-B stores per wave), then a
//     DETERMINISTIC gather sums, for every CSR entry, its contributions in ascending cell order and adds the
//     static part: F = (M/dt + nu K) + C(u_n).  No atomics, bitwise reproducible, same summation order as the
//     reference's cell loop.  This fuses system_matrix.add(-1,C_old) / add(+1,C_new) (NS3D.cpp:388,512) away.
//   * Dirichlet rows: MatrixTools::apply_boundary_values (NS3D.cpp:353,541) as a row-list kernel.
// Roofline: HBM.  Algorithmic bytes per cell (3D): 4*10 ids + 8*10 geometry + 8*30 gather of u_n + 8*100 local matrix.
#include "nsx_internal.hpp"

namespace nsx {

// ------------------------------------------------------------------ per-step cell kernel
// C_loc[a][b] = conv_scale * sum_q (w_q . grad N_b) N_a JxW_q  (+ 0.5 (div w)_q N_a N_b JxW_q if TEMAM)   (NS3D.cpp:456; NS2D.cpp:444-446)
// Work in reference coordinates: Ut_a = J^{-1} U_a, so J^{-1} w_q = sum_a Ut_a N_a(q), (w_q . grad N_b) = (J^{-1} w_q) . grad_hat N_b,
// div w = sum_a grad_hat N_a . Ut_a.  Tables: tN[q][a], tdN[q][a][k], and transposed copies tNT[a][q], tdNT[b][q][k] so that the
// wave-uniform (scalar) loads of one loop body are contiguous.  The b loop is kept rolled and its table pointers are made opaque
// per iteration: otherwise hipcc hoists all 560 table values into SGPRs and spills them into 256 VGPRs (occupancy 1).
template <int DIM, int NP2, int NQ, bool TEMAM>
__global__ __launch_bounds__(64) void k_cell_convection(int n_active, int n_cells, const int32_t *__restrict__ cell_n2,
                                                        const double *__restrict__ geo, const double *__restrict__ tN,
                                                        const double *__restrict__ tdN, const double *__restrict__ tNT,
                                                        const double *__restrict__ tdNT, const double *__restrict__ tw,
                                                        const double *__restrict__ sol, double conv_scale,
                                                        double *__restrict__ cellbuf) {
  const int cell = blockIdx.x * 64 + threadIdx.x;
  if (cell >= n_active) return;  // n_active = cells touching an owned node; n_cells = SoA plane stride
  double Ji[DIM][DIM];
#pragma unroll
  for (int k = 0; k < DIM; ++k)
#pragma unroll
    for (int d = 0; d < DIM; ++d) Ji[k][d] = geo[(size_t)(k * DIM + d) * n_cells + cell];
  const double adet = geo[(size_t)(DIM * DIM) * n_cells + cell];
  double Ut[NP2][DIM];
#pragma unroll
  for (int a = 0; a < NP2; ++a) {
    const int node = cell_n2[(size_t)a * n_cells + cell];
    double u[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) u[c] = sol[(size_t)node * DIM + c];
#pragma unroll
    for (int k = 0; k < DIM; ++k) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) s += Ji[k][d] * u[d];
      Ut[a][k] = s;
    }
  }
  double what[NQ][DIM];  // conv_scale * JxW_q * J^{-1} w_q
  double tq[NQ];         // 0.5 (div w)_q JxW_q
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double jxw = adet * tw[q];
    double wt[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) wt[k] = 0.0;
#pragma unroll
    for (int a = 0; a < NP2; ++a) {
      const double n = tN[q * NP2 + a];
#pragma unroll
      for (int k = 0; k < DIM; ++k) wt[k] += Ut[a][k] * n;
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) what[q][k] = wt[k] * (jxw * conv_scale);
    if (TEMAM) {
      double div = 0.0;
#pragma unroll
      for (int a = 0; a < NP2; ++a)
#pragma unroll
        for (int k = 0; k < DIM; ++k) div += tdN[(q * NP2 + a) * DIM + k] * Ut[a][k];
      tq[q] = 0.5 * div * jxw;
    }
  }
#pragma unroll 1
  for (int b = 0; b < NP2; ++b) {
    const double *pN = tNT, *pdNb = tdNT + b * NQ * DIM, *pNb = tNT + b * NQ;
    asm volatile("" : "+s"(pN), "+s"(pdNb), "+s"(pNb));
    double g[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) s += what[q][k] * pdNb[q * DIM + k];
      if (TEMAM) s += pNb[q] * tq[q];
      g[q] = s;
    }
    double *out = cellbuf + (size_t)b * n_cells + cell;
#pragma unroll
    for (int a = 0; a < NP2; ++a) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) s += pN[a * NQ + q] * g[q];
      out[(size_t)(a * NP2) * n_cells] = s;
    }
  }
}

// ------------------------------------------------------------------ first-step cell kernel (runs once per run)
// which: 0 mass/dt (NS3D.cpp:249), 1 nu*stiffness (:246), 2 D[a][v][c] = int psi_v d_c N_a (:258,:261), 3 pressure mass/nu (:264)
template <int DIM>
__global__ __launch_bounds__(64) void k_cell_static(int which, int n_cells, int np2, int np1, int nq,
                                                    const double *__restrict__ geo, const double *__restrict__ tN,
                                                    const double *__restrict__ tdN, const double *__restrict__ tN1,
                                                    const double *__restrict__ tw, double nu, double inv_dt,
                                                    double *__restrict__ cellbuf) {
  const int cell = blockIdx.x * 64 + threadIdx.x;
  if (cell >= n_cells) return;
  double Ji[DIM][DIM];
  for (int k = 0; k < DIM; ++k)
    for (int d = 0; d < DIM; ++d) Ji[k][d] = geo[(size_t)(k * DIM + d) * n_cells + cell];
  const double adet = geo[(size_t)(DIM * DIM) * n_cells + cell];
  if (which == 0) {
    for (int a = 0; a < np2; ++a)
      for (int b = 0; b < np2; ++b) {
        double s = 0.0;
        for (int q = 0; q < nq; ++q) s += tN[q * np2 + a] * tN[q * np2 + b] * inv_dt * (adet * tw[q]);
        cellbuf[(size_t)(a * np2 + b) * n_cells + cell] = s;
      }
  } else if (which == 1) {
    for (int a = 0; a < np2; ++a)
      for (int b = 0; b < np2; ++b) {
        double s = 0.0;
        for (int q = 0; q < nq; ++q) {
          double gg = 0.0;
          for (int d = 0; d < DIM; ++d) {
            double ga = 0.0, gb = 0.0;
            for (int k = 0; k < DIM; ++k) {
              ga += Ji[k][d] * tdN[(q * np2 + a) * DIM + k];
              gb += Ji[k][d] * tdN[(q * np2 + b) * DIM + k];
            }
            gg += ga * gb;
          }
          s += nu * gg * (adet * tw[q]);
        }
        cellbuf[(size_t)(a * np2 + b) * n_cells + cell] = s;
      }
  } else if (which == 2) {
    for (int a = 0; a < np2; ++a)
      for (int v = 0; v < np1; ++v)
        for (int c = 0; c < DIM; ++c) {
          double s = 0.0;
          for (int q = 0; q < nq; ++q) {
            double ga = 0.0;
            for (int k = 0; k < DIM; ++k) ga += Ji[k][c] * tdN[(q * np2 + a) * DIM + k];
            s += tN1[q * np1 + v] * ga * (adet * tw[q]);
          }
          cellbuf[(size_t)((a * np1 + v) * DIM + c) * n_cells + cell] = s;
        }
  } else {
    for (int v = 0; v < np1; ++v)
      for (int u = 0; u < np1; ++u) {
        double s = 0.0;
        for (int q = 0; q < nq; ++q) s += tN1[q * np1 + v] * tN1[q * np1 + u] / nu * (adet * tw[q]);
        cellbuf[(size_t)(v * np1 + u) * n_cells + cell] = s;
      }
  }
}

// ------------------------------------------------------------------ deterministic gather
// out[e][c] = (base ? base[e][c] : 0) + sign * sum_k buf[src[k] + c*n_cells];  out2 (optional) gets the sum alone.
template <int NCOMP>
__global__ __launch_bounds__(256) void k_gather(int64_t n_out, const int32_t *__restrict__ ptr, const int32_t *__restrict__ src,
                                                const double *__restrict__ buf, int64_t comp_stride, double sign,
                                                const double *__restrict__ base, double *__restrict__ out,
                                                double *__restrict__ out2) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_out) return;
  double s[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) s[c] = 0.0;
  // four contributions per pass: their offsets are requested together, then their values, then they are added in the order of the
  // list (one at a time every contribution was two trips through memory behind the one before it)
  const int k1 = ptr[e + 1];
  for (int k0 = ptr[e]; k0 < k1; k0 += 4) {
    int64_t o[4];
    double v[4][NCOMP];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = k0 + u < k1 ? (int64_t)src[k0 + u] : -1;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) v[u][c] = o[u] >= 0 ? buf[o[u] + c * comp_stride] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (o[u] >= 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) s[c] += v[u][c];
      }
  }
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    const double v = sign * s[c];
    if (out2) out2[e * NCOMP + c] = v;
    out[e * NCOMP + c] = base ? base[e * NCOMP + c] + v : v;
  }
}

__global__ void k_add3(int64_t n, const double *a, const double *b, const double *c, double *out, double *out_ab) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double ab = a[i] + b[i];
  if (out_ab) out_ab[i] = ab;
  out[i] = ab + c[i];
}

__global__ void k_fill(int64_t n, double *d, double v) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = v;
}

// ------------------------------------------------------------------ Dirichlet
// first non-zero |diagonal| of each rank's local range (deal.II apply_boundary_values for Trilinos matrices)
__global__ void k_dbar(int n_ranks, const int32_t *__restrict__ rank_u, const int32_t *__restrict__ diag,
                       const double *__restrict__ F, double *__restrict__ dbar) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_ranks) return;
  double d = 1.0;
  for (int i = rank_u[r]; i < rank_u[r + 1]; ++i) {
    const double v = F[diag[i]];
    if (v != 0.0) {
      d = fabs(v);
      break;
    }
  }
  dbar[r] = d;
}

// one wave per (dof, value) pair of the boundary_values map
template <int DIM>
__global__ __launch_bounds__(64) void k_dirichlet(int n, const int32_t *__restrict__ dofs, const double *__restrict__ vals,
                                                  int n_ranks, const int32_t *__restrict__ rank_u, const double *__restrict__ dbar,
                                                  const int32_t *__restrict__ A_rp, const int32_t *__restrict__ A_ci,
                                                  double *__restrict__ F, const int32_t *__restrict__ G_rp, double *__restrict__ G,
                                                  double *__restrict__ rhs, double *__restrict__ sol, double *__restrict__ mask) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int dof = dofs[k], node = dof / DIM, c = dof % DIM;
  int lo = 0, hi = n_ranks;  // rank of the node
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rank_u[mid] <= node) lo = mid; else hi = mid;
  }
  const double d = dbar[lo];
  if (c == 0)
    for (int p = A_rp[node] + threadIdx.x; p < A_rp[node + 1]; p += 64) F[p] = (A_ci[p] == node) ? d : 0.0;
  for (int p = G_rp[node] + threadIdx.x; p < G_rp[node + 1]; p += 64) G[(size_t)p * DIM + c] = 0.0;
  if (threadIdx.x == 0) {
    rhs[dof] = vals[k] * d;
    sol[dof] = vals[k];
    mask[dof] = 0.0;
  }
}

__global__ void k_add_rhs(int n, const int32_t *dofs, const double *vals, double *rhs) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) rhs[dofs[k]] += vals[k];  // the caller's list has unique dofs
}

// ------------------------------------------------------------------ host drivers
template <int DIM, int NP2, int NQ>
static void launch_conv(nsx_handle *h, bool temam, double conv_scale) {
  const int grid = cdiv(h->n_cells1, 64);
  const double bytes = (double)h->n_cells1 * (4.0 * NP2 + 8.0 * (DIM * DIM + 1) + 8.0 * DIM * NP2 + 8.0 * NP2 * NP2);
  LaunchScope ls(h, "cell_convection", bytes);
  if (temam)
    hipLaunchKernelGGL((k_cell_convection<DIM, NP2, NQ, true>), dim3(grid), dim3(64), 0, h->stream, h->n_cells1, h->n_cells,
                       h->cell_n2.p, h->geo.p, h->tab_N2.p, h->tab_dN2.p, h->tab_N2T.p, h->tab_dN2T.p, h->tab_w.p, h->sol.p, conv_scale,
                       h->cellbuf.p);
  else
    hipLaunchKernelGGL((k_cell_convection<DIM, NP2, NQ, false>), dim3(grid), dim3(64), 0, h->stream, h->n_cells1, h->n_cells,
                       h->cell_n2.p, h->geo.p, h->tab_N2.p, h->tab_dN2.p, h->tab_N2T.p, h->tab_dN2T.p, h->tab_w.p, h->sol.p, conv_scale,
                       h->cellbuf.p);
}

static void dispatch_conv(nsx_handle *h, bool temam, double conv_scale) {
  const int nq = h->n_q;
  if (h->dim == 2) {
    switch (nq) {
      case 3: return launch_conv<2, 6, 3>(h, temam, conv_scale);
      case 4: return launch_conv<2, 6, 4>(h, temam, conv_scale);
      case 6: return launch_conv<2, 6, 6>(h, temam, conv_scale);
      case 7: return launch_conv<2, 6, 7>(h, temam, conv_scale);
      case 12: return launch_conv<2, 6, 12>(h, temam, conv_scale);
    }
  } else {
    switch (nq) {
      case 4: return launch_conv<3, 10, 4>(h, temam, conv_scale);
      case 10: return launch_conv<3, 10, 10>(h, temam, conv_scale);
      case 11: return launch_conv<3, 10, 11>(h, temam, conv_scale);
      case 14: return launch_conv<3, 10, 14>(h, temam, conv_scale);
      case 15: return launch_conv<3, 10, 15>(h, temam, conv_scale);
      case 24: return launch_conv<3, 10, 24>(h, temam, conv_scale);
    }
  }
  NSX_THROW(NSX_ERR_UNSUPPORTED, "no cell kernel instantiated for dim=%d n_q=%d (see dispatch_conv in nsx_assemble.hip)", h->dim, nq);
}

template <int NCOMP>
static void gather(nsx_handle *h, const char *name, const GatherMap &gm, const double *buf, double sign, const double *base,
                   double *out, double *out2) {
  const double bytes = 4.0 * (gm.n_out + 1) + (4.0 + 8.0 * NCOMP) * gm.n_src + 8.0 * gm.n_out * NCOMP * (1 + (base != nullptr) + (out2 != nullptr));
  LaunchScope ls(h, name, bytes);
  hipLaunchKernelGGL((k_gather<NCOMP>), dim3(cdiv(gm.n_out, 256)), dim3(256), 0, h->stream, gm.n_out, gm.ptr.p, gm.src.p, buf,
                     (int64_t)h->n_cells, sign, base, out, out2);
}

static void launch_static(nsx_handle *h, int which) {
  const int grid = cdiv(h->n_cells, 64);
  LaunchScope ls(h, "cell_static", 0);
  if (h->dim == 2)
    hipLaunchKernelGGL((k_cell_static<2>), dim3(grid), dim3(64), 0, h->stream, which, h->n_cells, h->np2, h->np1, h->n_q, h->geo.p,
                       h->tab_N2.p, h->tab_dN2.p, h->tab_N1.p, h->tab_w.p, h->prm.nu, 1.0 / h->prm.deltat, h->cellbuf.p);
  else
    hipLaunchKernelGGL((k_cell_static<3>), dim3(grid), dim3(64), 0, h->stream, which, h->n_cells, h->np2, h->np1, h->n_q, h->geo.p,
                       h->tab_N2.p, h->tab_dN2.p, h->tab_N1.p, h->tab_w.p, h->prm.nu, 1.0 / h->prm.deltat, h->cellbuf.p);
}

void run_assemble(nsx_handle *h, bool first, int flags) {
  if (!h->have_mesh) NSX_THROW(NSX_ERR_ARG, "nsx_set_mesh first");
  if (!first && !h->assembled) NSX_THROW(NSX_ERR_ARG, "nsx_assemble (first step) must precede nsx_assemble_time_step");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int dim = h->dim;
  const int64_t nA = h->gA.nnz();
  if (first) {
    // static operators: mass/dt, nu*stiffness, +-B, pressure mass
    launch_static(h, 0);
    gather<1>(h, "gather_static", h->gmA, h->cellbuf.p, 1.0, nullptr, h->vMass.p, nullptr);
    launch_static(h, 1);
    gather<1>(h, "gather_static", h->gmA, h->cellbuf.p, 1.0, nullptr, h->vStiff.p, nullptr);
    launch_static(h, 2);
    if (dim == 2) {
      gather<2>(h, "gather_static", h->gmG, h->cellbuf.p, -1.0, nullptr, h->vG.p, nullptr);
      gather<2>(h, "gather_static", h->gmB, h->cellbuf.p, 1.0, nullptr, h->vB.p, nullptr);
    } else {
      gather<3>(h, "gather_static", h->gmG, h->cellbuf.p, -1.0, nullptr, h->vG.p, nullptr);
      gather<3>(h, "gather_static", h->gmB, h->cellbuf.p, 1.0, nullptr, h->vB.p, nullptr);
    }
    launch_static(h, 3);
    gather<1>(h, "gather_static", h->gmPM, h->cellbuf.p, 1.0, nullptr, h->vPM.p, nullptr);
    hipLaunchKernelGGL(k_fill, dim3(cdiv(h->len_u, 256)), dim3(256), 0, h->stream, (int64_t)h->len_u, h->dirmask.p, 1.0);
    // S0 = M + K is formed below together with F (k_add3)
  }
  const double conv_scale = (first && (flags & NSX_DOUBLE_CONVECTION)) ? 2.0 : 1.0;
  dispatch_conv(h, (flags & NSX_TEMAM) != 0, conv_scale);
  if (first) {
    gather<1>(h, "gather_convection", h->gmA, h->cellbuf.p, 1.0, nullptr, h->vConv.p, nullptr);
    // system(0,0) = mass + convection + stiffness (NS3D.cpp:322-324); keep S0 = mass + stiffness for later steps
    LaunchScope ls(h, "sum_static", 8.0 * nA * 5);
    hipLaunchKernelGGL(k_add3, dim3(cdiv(nA, 256)), dim3(256), 0, h->stream, nA, h->vMass.p, h->vStiff.p, h->vConv.p, h->vF.p, h->vS0.p);
  } else {
    // F = S0 + C_new  (system -= C_old ; system += C_new, NS3D.cpp:388,512) and convection_matrix = C_new
    gather<1>(h, "gather_convection", h->gmA, h->cellbuf.p, 1.0, h->vS0.p, h->vF.p, h->vConv.p);
  }
  // rhs (NS3D.cpp:269,459): rhs_i = sum_q (u_n . phi_i) JxW / dt = (mass_matrix * u_n)_i since u_n = sum_j U_j phi_j;
  // one SpMV with the stored M/dt replaces a second per-cell scatter.  Pressure part is zero (NS3D.cpp:195,396).
  spmv_F(h, h->vMass.p, h->sol.p, h->rhs.p);
  HIP_CHECK(hipMemsetAsync(h->rhs.p + h->off_p, 0, (size_t)h->n_p * sizeof(double), h->stream));
  HIP_CHECK(hipGetLastError());
  h->assembled = true;
  h->prec_ready = false;
  if (first) h->schur_valid = false;  // block(1,0) was reassembled
}

void run_dirichlet(nsx_handle *h, int n_in, const int32_t *dofs_in, const double *vals_in) {
  if (!h->assembled) NSX_THROW(NSX_ERR_ARG, "assemble before applying boundary values");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int dim = h->dim;
  if (n_in < 0 || (n_in > 0 && (!dofs_in || !vals_in))) NSX_THROW(NSX_ERR_ARG, "bad boundary value arrays");
  // the map uses global dofs; a rank applies the entries it owns (MatrixTools::apply_boundary_values does the same per rank)
  std::vector<int32_t> dofs;
  std::vector<double> vals;
  dofs.reserve((size_t)std::max(0, n_in));
  vals.reserve((size_t)std::max(0, n_in));
  const int32_t lo = dim * h->goff_u, hi = lo + h->n_u;
  for (int k = 0; k < n_in; ++k) {
    if (dofs_in[k] < 0 || dofs_in[k] >= h->n_u_glob) NSX_THROW(NSX_ERR_UNSUPPORTED, "only velocity dofs can be constrained (dof %d)", dofs_in[k]);
    if (k > 0 && dofs_in[k] <= dofs_in[k - 1]) NSX_THROW(NSX_ERR_ARG, "boundary map must be sorted by dof (std::map order)");
    if (dofs_in[k] >= lo && dofs_in[k] < hi) {
      const int32_t l = dofs_in[k] - lo;  // caller-local dof -> the node's place in the internal layout, same component
      dofs.push_back(dim * node_to_internal(h, l / dim) + l % dim);
      vals.push_back(vals_in[k]);
    }
  }
  const int n = (int)dofs.size();
  // validate the map once per distinct dof list: whole nodes only
  if (h->bc_cache != dofs) {
    if (n % dim) NSX_THROW(NSX_ERR_UNSUPPORTED, "boundary map must constrain all %d velocity components of a node", dim);
    for (int k = 0; k < n; ++k)
      if (k % dim == 0 ? dofs[k] % dim != 0 : dofs[k] != dofs[k - 1] + 1)
        NSX_THROW(NSX_ERR_UNSUPPORTED, "boundary map must constrain all %d velocity components of a node (dof %d)", dim, dofs[k] + lo);
    h->bc_cache = dofs;
    h->bc_dofs.alloc(n);
    h->bc_vals.alloc(n);
    if (n) HIP_CHECK(hipMemcpyAsync(h->bc_dofs.p, dofs.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  }
  if (n) HIP_CHECK(hipMemcpyAsync(h->bc_vals.p, vals.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int nr = (int)h->rank_u_h.size() - 1;
  {
    LaunchScope ls(h, "dirichlet", 0);
    hipLaunchKernelGGL(k_dbar, dim3(cdiv(nr, 64)), dim3(64), 0, h->stream, nr, h->rank_u.p, h->gA.diag.p, h->vF.p, h->dbar.p);
    if (n) {
      if (dim == 2)
        hipLaunchKernelGGL((k_dirichlet<2>), dim3(n), dim3(64), 0, h->stream, n, h->bc_dofs.p, h->bc_vals.p, nr, h->rank_u.p, h->dbar.p,
                           h->gA.rowptr.p, h->gA.colind.p, h->vF.p, h->gG.rowptr.p, h->vG.p, h->rhs.p, h->sol.p, h->dirmask.p);
      else
        hipLaunchKernelGGL((k_dirichlet<3>), dim3(n), dim3(64), 0, h->stream, n, h->bc_dofs.p, h->bc_vals.p, nr, h->rank_u.p, h->dbar.p,
                           h->gA.rowptr.p, h->gA.colind.p, h->vF.p, h->gG.rowptr.p, h->vG.p, h->rhs.p, h->sol.p, h->dirmask.p);
    }
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(h->stream));  // host arrays are borrowed for the call only
  comm_halo_u(h, h->dirmask.p);                // the Schur weights need the mask of ghost dofs too
  h->prec_ready = false;
}

}  // namespace nsx

extern "C" {

#define NSX_API_BODY(h_, ...)                  \
  if (!(h_)) return NSX_ERR_ARG;               \
  try {                                        \
    __VA_ARGS__;                               \
  } catch (const nsx::Error &e) {              \
    (h_)->err = e.msg;                         \
    return e.code;                             \
  } catch (const std::exception &e) {          \
    (h_)->err = e.what();                      \
    return NSX_ERR_ARG;                        \
  }                                            \
  return NSX_OK;

int nsx_assemble(nsx_handle *h, int flags) {
  // the first assembly is the run's set-up step: the ILU schedules for the final rank tables are built here, not inside
  // the first timed preconditioner initialisation
  NSX_API_BODY(h, {
    if (h->have_mesh) nsx::ensure_schedules(h);
    nsx::run_assemble(h, true, flags);
  })
}
int nsx_assemble_time_step(nsx_handle *h, int flags) { NSX_API_BODY(h, nsx::run_assemble(h, false, flags)) }
int nsx_apply_boundary_values(nsx_handle *h, int n, const int32_t *dofs, const double *vals) {
  NSX_API_BODY(h, nsx::run_dirichlet(h, n, dofs, vals))
}
int nsx_add_rhs(nsx_handle *h, int n, const int32_t *dofs, const double *vals) {
  NSX_API_BODY(h, {
    if (!h->assembled) NSX_THROW(NSX_ERR_ARG, "assemble first");
    if (n < 0 || (n && (!dofs || !vals))) NSX_THROW(NSX_ERR_ARG, "bad arrays");
    HIP_CHECK(hipSetDevice(h->prm.device));
    std::vector<int32_t> ld;  // global dofs -> local owned positions
    std::vector<double> lv;
    for (int k = 0; k < n; ++k) {
      const int32_t g = dofs[k];
      if (g < h->n_u_glob) {
        if (g >= h->dim * h->goff_u && g < h->dim * h->goff_u + h->n_u) {
          const int32_t l = g - h->dim * h->goff_u;
          ld.push_back(h->dim * nsx::node_to_internal(h, l / h->dim) + l % h->dim);
          lv.push_back(vals[k]);
        }
      } else if (g - h->n_u_glob >= h->goff_p && g - h->n_u_glob < h->goff_p + h->n_p) {
        ld.push_back(h->off_p + nsx::pnode_to_internal(h, g - h->n_u_glob - h->goff_p));
        lv.push_back(vals[k]);
      }
    }
    const int m = (int)ld.size();
    nsx::DevBuf<int32_t> d;
    nsx::DevBuf<double> v;
    d.upload(ld.data(), m, h->stream);
    v.upload(lv.data(), m, h->stream);
    if (m) hipLaunchKernelGGL(nsx::k_add_rhs, dim3(nsx::cdiv(m, 256)), dim3(256), 0, h->stream, m, d.p, v.p, h->rhs.p);
    HIP_CHECK(hipStreamSynchronize(h->stream));
  })
}

}  // extern "C"
