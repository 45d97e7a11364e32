// nsx_sparse.hip — sparse kernels of the saddle-point solve on gfx950.
//
//   block SpMV            system_matrix.vmult inside SolverGMRES    reference NavierStokes3D.cpp:574
//   F / B / B_T vmult     Preconditioners.hpp:175,201,280,304,385,398,496,507,510
//   S = B diag(v) B_T     B->mmult(negative_S, *B_T, v)              Preconditioners.hpp:144,248,358,468
//   ILU(0) factor/solve   TrilinosWrappers::PreconditionILU          Preconditioners.hpp:147-148,215-216 (Ifpack, overlap 0)
//
// Layout exploited: F(0,0) = delta_cd (x) A with a scalar P2 x P2 operator A, so one 12-byte CSR entry serves the
// dim interleaved velocity components (the reference streams dim^2 entries, 2/3 of them explicit zeros, SURVEY A-struct).
// All kernels are HBM/L2-bandwidth bound; reductions inside a row use wavefront shuffles (64-wide waves, sub-groups of
// 8/16/32/64 lanes per row chosen from the average row length).
#include <atomic>

#include "nsx_internal.hpp"
#include "nsx_ilu_lanes.hpp"

namespace nsx {

template <int LW>
__device__ __forceinline__ double lane_group_sum(double v) {
  if (LW >= 2) v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  if (LW >= 4) v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  if (LW >= 8) v += dpp_f64<0x141>(v);   // row_half_mirror
  if (LW >= 16) v += dpp_f64<0x140>(v);  // row_mirror
  if (LW >= 32) v += __shfl_xor(v, 16, 64);
  if (LW >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// sum over the W lanes of a row group, result in every lane: DPP (pure VALU) up to 16 lanes, then cross-row shuffles
template <int W>
__device__ __forceinline__ double group_sum(double v) {
  return lane_group_sum<W>(v);
}

// ------------------------------------------------------------------ SpMV
// y[i][c] = sum_j A[i,j] x[j][c]  (+ sum_k G[i,k][c] xp[k]);  W lanes per row.
template <int DIM, int W, bool WITH_G>
__global__ __launch_bounds__(256) void k_spmv_vel(int n_rows, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                  const double *__restrict__ av, const double *__restrict__ x,
                                                  const int32_t *__restrict__ grp, const int32_t *__restrict__ gci,
                                                  const double *__restrict__ gv, const double *__restrict__ xp,
                                                  double *__restrict__ y, const int32_t *__restrict__ rows) {
  // rows: optional list of the rows to compute (n_rows of them): interior / interface split of a distributed product
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so give XCD k the k-th contiguous eighth of the
  // rows: the x entries a row gathers are then shared inside one 4-MiB L2 instead of being fetched into all eight
  // (grid is a multiple of 8; speed only, any placement is correct).
  const int bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int slot = (bid * 256 + threadIdx.x) / W, lane = threadIdx.x % W;
  if (slot >= n_rows) return;  // whole groups exit together
  const int row = rows ? rows[slot] : slot;
  double acc[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) acc[c] = 0.0;
  const int e = rp[row + 1];
  for (int p = rp[row] + lane; p < e; p += W) {
    const double a = av[p];
    const double *xj = x + (size_t)ci[p] * DIM;
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc[c] += a * xj[c];
  }
  if (WITH_G) {
    const int ge = grp[row + 1];
    for (int p = grp[row] + lane; p < ge; p += W) {
      const double xk = xp[gci[p]];
#pragma unroll
      for (int c = 0; c < DIM; ++c) acc[c] += gv[(size_t)p * DIM + c] * xk;
    }
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) acc[c] = group_sum<W>(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) y[(size_t)row * DIM + c] = acc[c];
  }
}

// y_u[i][c] (+)= sum_k G[i,k][c] xp[k]
template <int DIM, int W>
__global__ __launch_bounds__(256) void k_spmv_G(int n_rows, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                const double *__restrict__ gv, const double *__restrict__ xp,
                                                double *__restrict__ y, int accumulate, const int32_t *__restrict__ rows) {
  const int slot = (blockIdx.x * 256 + threadIdx.x) / W, lane = threadIdx.x % W;
  if (slot >= n_rows) return;
  const int row = rows ? rows[slot] : slot;
  double acc[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) acc[c] = 0.0;
  const int e = rp[row + 1];
  for (int p = rp[row] + lane; p < e; p += W) {
    const double xk = xp[ci[p]];
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc[c] += gv[(size_t)p * DIM + c] * xk;
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c) acc[c] = group_sum<W>(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double *o = y + (size_t)row * DIM + c;
      *o = accumulate ? *o + acc[c] : acc[c];
    }
  }
}

// y_p[i] = sum_j sum_c B[i,j][c] xu[j][c]
template <int DIM, int W>
__global__ __launch_bounds__(256) void k_spmv_B(int n_rows, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                const double *__restrict__ bv, const double *__restrict__ xu,
                                                double *__restrict__ y, const int32_t *__restrict__ rows) {
  const int slot = (blockIdx.x * 256 + threadIdx.x) / W, lane = threadIdx.x % W;
  if (slot >= n_rows) return;
  const int row = rows ? rows[slot] : slot;
  double acc = 0.0;
  const int e = rp[row + 1];
  for (int p = rp[row] + lane; p < e; p += W) {
    const double *xj = xu + (size_t)ci[p] * DIM;
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc += bv[(size_t)p * DIM + c] * xj[c];
  }
  acc = group_sum<W>(acc);
  if (lane == 0) y[row] = acc;
}

template <int W>
__global__ __launch_bounds__(256) void k_spmv_csr(int n_rows, const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                  const double *__restrict__ v, const double *__restrict__ x,
                                                  double *__restrict__ y, const int32_t *__restrict__ rows) {
  const int slot = (blockIdx.x * 256 + threadIdx.x) / W, lane = threadIdx.x % W;
  if (slot >= n_rows) return;
  const int row = rows ? rows[slot] : slot;
  double acc = 0.0;
  const int e = rp[row + 1];
  for (int p = rp[row] + lane; p < e; p += W) acc += v[p] * x[ci[p]];
  acc = group_sum<W>(acc);
  if (lane == 0) y[row] = acc;
}

#ifdef NSX_SPMV_TRACE  // development only (tools/spmv_trace.sh): wall-clock stamps of every workgroup of ONE launch of the LDS-staged SpMV
__device__ unsigned long long *g_spmv_trace = nullptr;
__device__ int g_spmv_dbg = 0;  // what the traced launch leaves out: 1 the x gathers, 2 gathers from consecutive addresses, 3 the matrix stream, 4 all of the staging
#define SPMV_DBG (g_spmv_trace ? g_spmv_dbg : 0)
#define SPMV_STAMP(k)                                                                              \
  do {                                                                                             \
    if (g_spmv_trace && threadIdx.x == 0) g_spmv_trace[(size_t)blockIdx.x * 4 + (k)] = wall_clock64(); \
  } while (0)
#else
#define SPMV_DBG 0
#define SPMV_STAMP(k) \
  do {                \
  } while (0)
#endif

// LDS-staged SpMV: y[i][c] = sum_j A[i,j] x[j][c] with the chunk's x entries gathered ONCE into LDS (SpmvBlocked).
// The per-non-zero stream is 8 B value + 2 B local column; the 24-B gathers of the plain kernel (10 M per product,
// bound by the per-CU address rate, not by bytes) become ~1.5 M staged gathers + LDS reads.
template <int DIM, int W>
__global__ __launch_bounds__(256) void k_spmv_blocked(int n_rows, const int32_t *__restrict__ desc, const int32_t *__restrict__ rp, const uint16_t *__restrict__ lidx,
                                                      const double *__restrict__ av, const int32_t *__restrict__ ucols, const double *__restrict__ x,
                                                      double *__restrict__ y) {
  extern __shared__ double xs[];
  __shared__ int rps[449];
  // launch order (SpmvBlocked::desc): workgroups are dealt round-robin over the 8 XCDs and XCD k takes a contiguous range of chunks, so
  // the x entries its chunks stage (neighbouring chunks share most of them) stay in ONE 4-MiB L2 instead of being fetched into all
  // eight (x is 8 MB at 1 M DoF); inside an XCD the chunks with the most non-zeros start first.  Speed only.
  const int4 d = reinterpret_cast<const int4 *>(desc)[blockIdx.x];
  const int c0 = d.x, nu = d.y, r0 = d.z, r1 = d.w;
  if (r1 <= r0) return;
  SPMV_STAMP(0);
  [[maybe_unused]] const int dbg = SPMV_DBG;
  for (int t = threadIdx.x; t <= r1 - r0; t += 256) rps[t] = rp[r0 + t];
  for (int t = threadIdx.x; t < nu && dbg != 4; t += 256) {
    const int u = dbg == 2 ? (c0 + t) % n_rows : ucols[c0 + t];
    const double *xj = x + (size_t)u * DIM;
#pragma unroll
    for (int c = 0; c < DIM; ++c) xs[t * DIM + c] = dbg == 1 ? (double)u : xj[c];
  }
  __syncthreads();
  SPMV_STAMP(1);
  constexpr int G = 256 / W, U = 4;  // rows in flight per pass, loads per lane kept in flight
  const int grp = threadIdx.x / W, lane = threadIdx.x % W;
  // software pipeline over the rows of this lane group: the loads of the next row are issued before the current row is
  // reduced, so a wave always has 2*U value loads + 2*U index loads outstanding instead of one dependent chain per row
  double a[U], na[U];
  int l[U], nl[U];
  int row = r0 + grp;
  auto fetch = [&](int rw, double (&va)[U], int (&vl)[U]) {
    const bool live = rw < r1;
    const int p0 = live ? rps[rw - r0] + lane : 0, e = live ? rps[rw - r0 + 1] : 0;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int q = p0 + k * W;
      const bool ok = q < e;
      va[k] = ok ? ld_stream<2>(av + q) : 0.0;
      vl[k] = ok ? (int)ld_stream<2>(lidx + q) : 0;
    }
  };
  auto consume = [&](int rw, const double (&va)[U], const int (&vl)[U]) {
    if (rw >= r1) return;
    double acc[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc[c] = 0.0;
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const double *xj = xs + vl[k] * DIM;
#pragma unroll
      for (int c = 0; c < DIM; ++c) acc[c] += va[k] * xj[c];
    }
    const int e = rps[rw - r0 + 1];
    for (int p = rps[rw - r0] + lane + U * W; p < e; p += W) {  // rows longer than U*W entries (rare)
      const double av_ = av[p];
      const double *xj = xs + (int)lidx[p] * DIM;
#pragma unroll
      for (int c = 0; c < DIM; ++c) acc[c] += av_ * xj[c];
    }
#pragma unroll
    for (int c = 0; c < DIM; ++c) acc[c] = group_sum<W>(acc[c]);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) y[(size_t)rw * DIM + c] = acc[c];
    }
  };
  // two register sets, no moves: while one row is reduced the loads of the next two are in flight
  if (SPMV_DBG == 3) row = r1;
  fetch(row, a, l);
  fetch(row + G, na, nl);
  for (; row < r1; row += 2 * G) {
    consume(row, a, l);
    fetch(row + 2 * G, a, l);
    consume(row + G, na, nl);
    fetch(row + 3 * G, na, nl);
  }
#ifdef NSX_SPMV_TRACE
  __syncthreads();
  SPMV_STAMP(2);
  if (g_spmv_trace && threadIdx.x == 0) {
    unsigned xcc, id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    g_spmv_trace[(size_t)blockIdx.x * 4 + 3] = ((unsigned long long)xcc << 32) | id;
  }
#endif
}

static double bytes_vel(nsx_handle *h, bool with_g) {
  double b = 12.0 * h->gA.nnz() + (double)h->N2 * (4 + 8.0 * h->dim * 2);
  if (with_g) b += (4.0 + 8.0 * h->dim) * h->gG.nnz() + 4.0 * h->N2 + 8.0 * h->NP;
  return b;
}

// ---- launches (rows == nullptr: all n rows)
static void launch_vel(nsx_handle *h, bool with_g, const double *vals, const double *x, const double *xp, double *y, const int32_t *rows, int n) {
  if (n <= 0) return;
  static const int Wsel = getenv("NSX_SPMV_W") ? atoi(getenv("NSX_SPMV_W")) : 16;
#define NSX_SPMV(D, W_, G_)                                                                                                            \
  hipLaunchKernelGGL((k_spmv_vel<D, W_, G_>), dim3((cdiv((int64_t)n * W_, 256) + 7) & ~7), dim3(256), 0, h->stream, n, h->gA.rowptr.p, \
                     h->gA.colind.p, vals, x, h->gG.rowptr.p, h->gG.colind.p, h->vG.p, xp, y, rows)
  if (with_g) {
    if (h->dim == 2) NSX_SPMV(2, 16, true); else NSX_SPMV(3, 16, true);
  } else if (h->dim == 2) {
    if (Wsel == 8) NSX_SPMV(2, 8, false); else if (Wsel == 32) NSX_SPMV(2, 32, false); else NSX_SPMV(2, 16, false);
  } else {
    if (Wsel == 8) NSX_SPMV(3, 8, false); else if (Wsel == 32) NSX_SPMV(3, 32, false); else NSX_SPMV(3, 16, false);
  }
#undef NSX_SPMV
}
static void launch_B(nsx_handle *h, const double *xu, double *yp, const int32_t *rows, int n) {
  if (n <= 0) return;
  if (h->dim == 2)
    hipLaunchKernelGGL((k_spmv_B<2, 32>), dim3(cdiv((int64_t)n * 32, 256)), dim3(256), 0, h->stream, n, h->gB.rowptr.p, h->gB.colind.p, h->vB.p, xu, yp, rows);
  else
    hipLaunchKernelGGL((k_spmv_B<3, 64>), dim3(cdiv((int64_t)n * 64, 256)), dim3(256), 0, h->stream, n, h->gB.rowptr.p, h->gB.colind.p, h->vB.p, xu, yp, rows);
}
static void launch_G(nsx_handle *h, const double *xp, double *yu, bool accumulate, const int32_t *rows, int n) {
  if (n <= 0) return;
  const int W = 8, grid = cdiv((int64_t)n * W, 256);
  if (h->dim == 2)
    hipLaunchKernelGGL((k_spmv_G<2, W>), dim3(grid), dim3(256), 0, h->stream, n, h->gG.rowptr.p, h->gG.colind.p, h->vG.p, xp, yu, (int)accumulate, rows);
  else
    hipLaunchKernelGGL((k_spmv_G<3, W>), dim3(grid), dim3(256), 0, h->stream, n, h->gG.rowptr.p, h->gG.colind.p, h->vG.p, xp, yu, (int)accumulate, rows);
}
static void launch_S(nsx_handle *h, const double *x, double *y, const int32_t *rows, int n) {
  if (n <= 0) return;
  const int W = 32;
  hipLaunchKernelGGL((k_spmv_csr<W>), dim3(cdiv((int64_t)n * W, 256)), dim3(256), 0, h->stream, n, h->gS.rowptr.p, h->gS.colind.p, h->vSchur.p, x, y, rows);
}

// Distributed products: the rows whose columns are all owned are computed while the ghosts of the input are still on
// their way (second stream, comm_halo_begin / finish); the rows on the partition interface follow once they have arrived.
// This is the Epetra_Import + local multiply of every vmult with the import hidden behind the interior rows.
// y_u = A x_u through the LDS-staged kernel; false if the handle has no chunk table for it
bool blocked_usable(const nsx_handle *h) {
  static const bool blocked = !(getenv("NSX_SPMV_BLOCKED") && atoi(getenv("NSX_SPMV_BLOCKED")) == 0);
  const SpmvBlocked &b = h->blkA;
  return blocked && b.n_chunks > 0 && b.max_rows <= 448 && (size_t)b.max_ucols * h->dim * sizeof(double) <= 64 * 1024;
}
// part 0: the table of a one-GPU handle / the chunks of a distributed handle whose staged columns are all owned; part 1: the chunks that
// stage a ghost column (distributed handles only; an empty table launches nothing)
static bool launch_blocked(nsx_handle *h, const double *vals, const double *x, double *y, int part = 0) {
  const SpmvBlocked &b = h->blkA;
  if (!blocked_usable(h)) return false;
  const size_t shm = (size_t)b.max_ucols * h->dim * sizeof(double);
  const int grid = part ? b.grid_if : b.grid;
  const int32_t *desc = part ? b.desc_if.p : b.desc.p;
  if (grid == 0) return true;
#define NSX_BLK(D) \
  hipLaunchKernelGGL((k_spmv_blocked<D, 16>), dim3(grid), dim3(256), shm, h->stream, h->N2, desc, h->gA.rowptr.p, b.lidx.p, vals, b.ucols.p, x, y)
#ifdef NSX_SPMV_TRACE
  static int n_call = 0;
  const bool traced = getenv("NSX_SPMV_TRACE_OUT") && ++n_call == (getenv("NSX_SPMV_TRACE_CALL") ? atoi(getenv("NSX_SPMV_TRACE_CALL")) : 5000);
  for (int mode = 0; traced && part == 0 && mode <= 4; ++mode) {  // every variant once, traced; the untraced launch below leaves the right y behind
    unsigned long long *tr = nullptr;
    (void)hipStreamSynchronize(h->stream);
    (void)hipMalloc(&tr, (size_t)grid * 4 * sizeof(unsigned long long));
    (void)hipMemset(tr, 0, (size_t)grid * 4 * sizeof(unsigned long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_trace), &tr, sizeof(tr));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_dbg), &mode, sizeof(mode));
    if (h->dim == 2) NSX_BLK(2); else NSX_BLK(3);
    (void)hipStreamSynchronize(h->stream);
    std::vector<unsigned long long> host((size_t)grid * 4);
    (void)hipMemcpy(host.data(), tr, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long *none = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_spmv_trace), &none, sizeof(none));
    (void)hipFree(tr);
    const std::string path = std::string(getenv("NSX_SPMV_TRACE_OUT")) + "_mode" + std::to_string(mode) + ".txt";
    if (FILE *f = fopen(path.c_str(), "w")) {
      fprintf(f, "# block chunk rows ucols t_start t_staged t_end xcc hw_id   (wall_clock64 ticks of 10 ns)\n");
      std::vector<int32_t> ds((size_t)grid * 4);
      (void)hipMemcpy(ds.data(), b.desc.p, ds.size() * 4, hipMemcpyDeviceToHost);
      for (int blk = 0; blk < grid; ++blk) {
        if (b.order[blk] < 0) continue;
        fprintf(f, "%d %d %d %d %llu %llu %llu %llu %llu\n", blk, b.order[blk], ds[4 * (size_t)blk + 3] - ds[4 * (size_t)blk + 2], ds[4 * (size_t)blk + 1],
                host[4 * (size_t)blk], host[4 * (size_t)blk + 1], host[4 * (size_t)blk + 2], host[4 * (size_t)blk + 3] >> 32, host[4 * (size_t)blk + 3] & 0xffffffffull);
      }
      fclose(f);
    }
  }
#endif
  if (h->dim == 2) NSX_BLK(2); else NSX_BLK(3);
#undef NSX_BLK
  return true;
}

// algorithmic bytes of a ghost exchange as LaunchScope counts them (pack read + buffer write, both directions)
static double bytes_halo(const HaloPlan &p, int ncomp) {
  const int nn = (int)p.nbr.size();
  return nn ? 16.0 * (p.send_ptr[nn] + p.recv_ptr[nn]) * ncomp : 0.0;
}

void spmv_F(nsx_handle *h, const double *vals, const double *x, double *y) {
  if (h->dist) {
    // Scopes: "spmv_F" = the rows that need no ghost (they run while the exchange is in flight), "spmv_F_if" = the rows behind it,
    // "halo_u_wait" = what the compute stream waits for the exchange once the first launch is through (its exposed part).
    // (Round 5 tried the second launch on a SECOND compute stream behind the exchange's event, beside the first launch -- a launch of a few
    // hundred chunks costs a 12 - 18 us floor whatever it holds -- and measured it without an exchange in between, one rank's share alone
    // on the card: 1.09 M DoF / 8: +15 us per product (26 - 28 against 23 - 26 ms for a solve of 128 products), / 2: no difference.  The
    // two extra cross-stream event hops cost more than the floor they hide.  Removed.)
    double *xx = const_cast<double *>(x);
    const bool blk = blocked_usable(h);
    const double frac_if = blk ? h->blkA.frac_if : (double)h->splitA.n_interface / std::max(1, h->N2);
    comm_halo_begin(h, h->haloU, xx, h->dim);
    {
      LaunchScope ls(h, "spmv_F", bytes_vel(h, false) * (1.0 - frac_if));
      if (blk) launch_blocked(h, vals, x, y, 0);
      else launch_vel(h, false, vals, x, nullptr, y, h->splitA.interior.p, h->splitA.n_interior);
    }
    {
      LaunchScope ls(h, "halo_u_wait", bytes_halo(h->haloU, h->dim));
      comm_halo_finish(h, h->haloU, xx, h->dim);
    }
    {
      LaunchScope ls(h, "spmv_F_if", bytes_vel(h, false) * frac_if);
      if (blk) launch_blocked(h, vals, x, y, 1);
      else launch_vel(h, false, vals, x, nullptr, y, h->splitA.interface.p, h->splitA.n_interface);
    }
    return;
  }
  LaunchScope ls(h, "spmv_F", bytes_vel(h, false));
  if (!launch_blocked(h, vals, x, y)) launch_vel(h, false, vals, x, nullptr, y, nullptr, h->N2);
}

static double bytes_B(nsx_handle *h) { return (4.0 + 8.0 * h->dim) * h->gB.nnz() + 12.0 * h->NP + 8.0 * h->dim * h->N2; }
void spmv_B(nsx_handle *h, const double *xu, double *yp) {
  LaunchScope ls(h, "spmv_B", bytes_B(h));
  if (h->dist) {
    double *xx = const_cast<double *>(xu);
    comm_halo_begin(h, h->haloU, xx, h->dim);
    launch_B(h, xu, yp, h->splitB.interior.p, h->splitB.n_interior);
    comm_halo_finish(h, h->haloU, xx, h->dim);
    launch_B(h, xu, yp, h->splitB.interface.p, h->splitB.n_interface);
    return;
  }
  launch_B(h, xu, yp, nullptr, h->NP);
}

void spmv_G(nsx_handle *h, const double *xp, double *yu, bool accumulate) {
  LaunchScope ls(h, "spmv_G", (4.0 + 8.0 * h->dim) * h->gG.nnz() + (double)h->N2 * (4 + 8.0 * h->dim) + 8.0 * h->NP);
  if (h->dist) {
    double *xx = const_cast<double *>(xp);
    comm_halo_begin(h, h->haloP, xx, 1);
    launch_G(h, xp, yu, accumulate, h->splitG.interior.p, h->splitG.n_interior);
    comm_halo_finish(h, h->haloP, xx, 1);
    launch_G(h, xp, yu, accumulate, h->splitG.interface.p, h->splitG.n_interface);
    return;
  }
  launch_G(h, xp, yu, accumulate, nullptr, h->N2);
}

// BlockSparseMatrix::vmult: y_u = F x_u + block(0,1) x_p ; y_p = block(1,0) x_u  (block (1,1) has an empty pattern)
void spmv_saddle(nsx_handle *h, const double *x, double *y) {
  double *xx = const_cast<double *>(x);
  if (h->dist) {
    comm_halo_begin(h, h->haloU, xx, h->dim);
    comm_halo_begin(h, h->haloP, xx + h->off_p, 1);
  }
  const bool blk_dist = h->dist && blocked_usable(h);
  {
    LaunchScope ls(h, "spmv_saddle_u", bytes_vel(h, true));
    if (blk_dist) launch_blocked(h, h->vF.p, x, y, 0);  // F x_u on the chunks without a ghost column; the rest and += block(0,1) x_p behind the exchange
    else if (h->dist) launch_vel(h, true, h->vF.p, x, x + h->off_p, y, h->splitVel.interior.p, h->splitVel.n_interior);
    else if (launch_blocked(h, h->vF.p, x, y)) launch_G(h, x + h->off_p, y, true, nullptr, h->N2);  // F x_u staged through LDS, then += block(0,1) x_p
    else launch_vel(h, true, h->vF.p, x, x + h->off_p, y, nullptr, h->N2);
  }
  {
    LaunchScope ls(h, "spmv_B", bytes_B(h));
    if (h->dist) launch_B(h, x, y + h->off_p, h->splitB.interior.p, h->splitB.n_interior);
    else launch_B(h, x, y + h->off_p, nullptr, h->NP);
  }
  if (h->dist) {
    {
      LaunchScope ls(h, "halo_up_wait", bytes_halo(h->haloU, h->dim) + bytes_halo(h->haloP, 1));
      comm_halo_finish(h, h->haloU, xx, h->dim);
      comm_halo_finish(h, h->haloP, xx + h->off_p, 1);
    }
    LaunchScope ls(h, "spmv_saddle_if", 0);
    if (blk_dist) {
      launch_blocked(h, h->vF.p, x, y, 1);
      launch_G(h, x + h->off_p, y, true, nullptr, h->N2);
    } else {
      launch_vel(h, true, h->vF.p, x, x + h->off_p, y, h->splitVel.interface.p, h->splitVel.n_interface);
    }
    launch_B(h, x, y + h->off_p, h->splitB.interface.p, h->splitB.n_interface);
  }
}

void spmv_S(nsx_handle *h, const double *x, double *y) {
  LaunchScope ls(h, "spmv_S", 12.0 * h->gS.nnz() + 20.0 * h->NP);
  if (h->dist) {
    double *xx = const_cast<double *>(x);
    comm_halo_begin(h, h->haloP, xx, 1);
    launch_S(h, x, y, h->splitS.interior.p, h->splitS.n_interior);
    comm_halo_finish(h, h->haloP, xx, 1);
    launch_S(h, x, y, h->splitS.interface.p, h->splitS.n_interface);
    return;
  }
  launch_S(h, x, y, nullptr, h->NP);
}

// interior / interface rows of the distributed operators (owned rows; a column >= the owned count is a ghost)
static void split_rows(nsx_handle *h, RowSplit &sp, int n_rows, const Csr &a, int own_a, const Csr *b, int own_b) {
  std::vector<int32_t> in, out;
  for (int i = 0; i < n_rows; ++i) {
    bool ghost = false;
    for (int k = a.rowptr[i]; k < a.rowptr[i + 1] && !ghost; ++k) ghost = a.colind[k] >= own_a;
    if (b)
      for (int k = b->rowptr[i]; k < b->rowptr[i + 1] && !ghost; ++k) ghost = b->colind[k] >= own_b;
    (ghost ? out : in).push_back(i);
  }
  sp.n_interior = (int)in.size();
  sp.n_interface = (int)out.size();
  sp.interior.upload(in, h->stream);
  sp.interface.upload(out, h->stream);
}
void build_row_splits(nsx_handle *h) {
  if (!h->dist) return;
  split_rows(h, h->splitA, h->N2, h->gA.host, h->N2, nullptr, 0);
  split_rows(h, h->splitVel, h->N2, h->gA.host, h->N2, &h->gG.host, h->NP);
  split_rows(h, h->splitG, h->N2, h->gG.host, h->NP, nullptr, 0);
  split_rows(h, h->splitB, h->NP, h->gB.host, h->N2, nullptr, 0);
  split_rows(h, h->splitS, h->NP, h->gS.host, h->NP, nullptr, 0);
  if (getenv("NSX_DEBUG"))
    fprintf(stderr, "[nsx] rank %d: interface rows F %d / %d, block(1,0) %d / %d, S %d / %d\n", h->rank, h->splitA.n_interface, h->N2, h->splitB.n_interface,
            h->NP, h->splitS.n_interface, h->NP);
}

// ------------------------------------------------------------------ diagonals
__global__ void k_extract_diag(int n_nodes, int dim, const int32_t *__restrict__ diag, const double *__restrict__ v, double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes * dim) return;
  d[i] = v[diag[i / dim]];
}
__global__ void k_abs_rowsum(int n_nodes, int dim, const int32_t *__restrict__ rp, const double *__restrict__ v, double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  double s = 0.0;
  for (int p = rp[i]; p < rp[i + 1]; ++p) s += fabs(v[p]);
  for (int c = 0; c < dim; ++c) d[(size_t)i * dim + c] = s;
}
// F->diag_element(i) for every velocity dof (Prec.hpp:137,241,353,449): the scalar diagonal replicated on dim components
void extract_diag(nsx_handle *h, const DevCsr &g, const double *vals, double *d) {
  LaunchScope ls(h, "extract_diag", 0);
  hipLaunchKernelGGL(k_extract_diag, dim3(cdiv(h->n_u, 256)), dim3(256), 0, h->stream, g.n_rows(), h->dim, g.diag.p, vals, d);
}
// sum_j |M_ij| over the row (Prec.hpp:456-465); cross-component slots of the reference's row are zero
void abs_rowsum(nsx_handle *h, const DevCsr &g, const double *vals, double *d) {
  LaunchScope ls(h, "abs_rowsum", 0);
  hipLaunchKernelGGL(k_abs_rowsum, dim3(cdiv(g.n_rows(), 256)), dim3(256), 0, h->stream, g.n_rows(), h->dim, g.rowptr.p, vals, d);
}

// ------------------------------------------------------------------ S = B diag(v) B_T  (numeric phase on the static pattern)
// block(0,1) = -block(1,0)^T with the Dirichlet rows cleared (NS3D.cpp:258,261 + apply_boundary_values), so
//   S_ij = sum_{k in row_i(B) ^ row_j(B)} sum_c B[i,k][c] * w[k][c] * B[j,k][c],   w = -v * dirichlet_mask.
// One wave per row i: row i (columns + dim weighted values) is staged in LDS, each lane takes one S entry (i,j),
// walks row j of B and merges its (sorted) columns with the LDS copy.
template <int DIM>
__global__ __launch_bounds__(64) void k_schur(int n_rows, const int32_t *__restrict__ srp, const int32_t *__restrict__ sci,
                                              const int32_t *__restrict__ brp, const int32_t *__restrict__ bci,
                                              const double *__restrict__ bv, const double *__restrict__ w, int max_row,
                                              double *__restrict__ sv) {
  extern __shared__ double smem[];
  double *rv = smem;                                  // [max_row][DIM]
  int32_t *rc = (int32_t *)(smem + (size_t)max_row * DIM);  // [max_row]
  const int i = blockIdx.x;
  if (i >= n_rows) return;
  const int b0 = brp[i], nb = brp[i + 1] - b0;
  for (int t = threadIdx.x; t < nb; t += 64) {
    const int k = bci[b0 + t];
    rc[t] = k;
#pragma unroll
    for (int c = 0; c < DIM; ++c) rv[t * DIM + c] = bv[(size_t)(b0 + t) * DIM + c] * w[(size_t)k * DIM + c];
  }
  __syncthreads();
  for (int e = srp[i] + threadIdx.x; e < srp[i + 1]; e += 64) {
    const int j = sci[e];
    double acc = 0.0;
    // both rows are sorted by column: one pass over row j with a cursor into row i's LDS copy (round 5; a binary search per entry of
    // row j before: 1.24 ms per step).  The products enter the sum in row j's order as before: bit-identical.
    // both rows are sorted by column: one pass over row j with a cursor into row i's LDS copy (round 5; a binary search per entry of
    // row j before: 1.24 ms per step, now 0.93).  The products enter the sum in row j's order as before: bit-identical.  What bounds it
    // now is the gather rate: every lane walks another row j, so a load instruction touches 64 different sectors (430 M lane-loads per
    // product); fetching eight entries per trip changed nothing (1.02 ms).  A wave per (i, j) pair would read row j coalesced, but
    // its sum would need the lanes' products in row order -- another rounding, another iteration history.
    int q = 0;
    for (int p = brp[j]; p < brp[j + 1]; ++p) {
      const int k = bci[p];
      while (q < nb && rc[q] < k) ++q;
      if (q < nb && rc[q] == k) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc += rv[q * DIM + c] * bv[(size_t)p * DIM + c];
      }
    }
    sv[e] = acc;
  }
}

void schur_numeric(nsx_handle *h, const double *w) {
  comm_halo_u(h, w);  // weights of ghost velocity dofs
  int max_row = 0;
  const Csr &B = h->gB.host;
  for (int i = 0; i < B.n_rows; ++i) max_row = std::max(max_row, B.rowptr[i + 1] - B.rowptr[i]);
  const size_t shm = (size_t)max_row * (h->dim * 8 + 4) + 8;
  if (shm > 160 * 1024) NSX_THROW(NSX_ERR_UNSUPPORTED, "row of block(1,0) too long for the Schur kernel (%d)", max_row);
  LaunchScope ls(h, "schur_numeric", 0);
  if (h->dim == 2)
    hipLaunchKernelGGL((k_schur<2>), dim3(h->NP), dim3(64), shm, h->stream, h->NP, h->gS.rowptr.p, h->gS.colind.p, h->gB.rowptr.p,
                       h->gB.colind.p, h->vB.p, w, max_row, h->vSchur.p);
  else
    hipLaunchKernelGGL((k_schur<3>), dim3(h->NP), dim3(64), shm, h->stream, h->NP, h->gS.rowptr.p, h->gS.colind.p, h->gB.rowptr.p,
                       h->gB.colind.p, h->vB.p, w, max_row, h->vSchur.p);
}

// ------------------------------------------------------------------ block-Jacobi ILU(0)
// One workgroup per rank block, rows visited level by level (levels precomputed on the static pattern, exact:
// the arithmetic of every row is that of the sequential IKJ sweep).  Storage = Ifpack_ILU's: strict lower = L,
// diagonal = 1/d, strict upper = U/d.  Entries whose column lies outside the block are dropped (Ifpack_LocalFilter).
constexpr int ILU_WAVES = 4;
constexpr int ILU_MAXROW = 1024;

__global__ __launch_bounds__(ILU_WAVES * 64) void k_ilu_factor(const int32_t *__restrict__ bptr, const int32_t *__restrict__ lvl_off,
                                                              const int32_t *__restrict__ lvl_ptr, const int32_t *__restrict__ lvl_rows,
                                                              const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                              const int32_t *__restrict__ diag, const double *__restrict__ a,
                                                              double *__restrict__ lu, const int32_t *__restrict__ slot_of,
                                                              double *__restrict__ pk_val, double *__restrict__ pk_dinv,
                                                              int *__restrict__ err, const int32_t *__restrict__ dinv_slot) {
  __shared__ double wv[ILU_WAVES][ILU_MAXROW];
  const int blk = blockIdx.x, wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int r0 = bptr[blk], r1 = bptr[blk + 1];
  volatile double *w = wv[wave];  // volatile: lanes of the wave exchange values through this row buffer
  for (int lv = lvl_off[blk]; lv < lvl_off[blk + 1]; ++lv) {
    const int l0 = lvl_ptr[lv], cnt = lvl_ptr[lv + 1] - l0;
    for (int r = wave; r < cnt; r += ILU_WAVES) {
      const int i = lvl_rows[l0 + r];
      const int p0 = rp[i], n = rp[i + 1] - p0, dpos = diag[i] - p0;
      if (n > ILU_MAXROW) {
        if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        continue;
      }
      for (int t = lane; t < n; t += 64) {
        const int j = ci[p0 + t];
        w[t] = (j >= r0 && j < r1) ? a[p0 + t] : 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = 0; t < dpos; ++t) {  // L part, ascending columns (wave-uniform loop)
        const int j = ci[p0 + t];
        if (j < r0) continue;
        const int dj = diag[j];
        const double mult = w[t];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) w[t] = mult * lu[dj];  // InV[jj] *= DV[j]
        const int ue = rp[j + 1];
        for (int q = dj + 1 + lane; q < ue; q += 64) {  // scaled U row of j
          const int c = ci[q];
          if (c >= r1) break;
          int lo = t + 1, hi = n - 1;  // columns of row i are sorted; c > j
          while (lo <= hi) {
            const int mid = (lo + hi) >> 1, cm = ci[p0 + mid];
            if (cm < c) lo = mid + 1; else if (cm > c) hi = mid - 1; else { w[mid] -= mult * lu[q]; break; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      const double d = w[dpos];
      const double dinv = 1.0 / d;
      if (lane == 0 && !(fabs(d) > 0.0)) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (int t = lane; t < n; t += 64) {
        const int j = ci[p0 + t];
        double v = w[t];
        if (t == dpos) v = dinv;
        else if (t > dpos) v = (j < r1) ? v * dinv : 0.0;
        lu[p0 + t] = v;
        if (slot_of) {
          const int sl = slot_of[p0 + t];
          if (sl >= 0) pk_val[sl] = -v;  // the stream adds value * x[col]: it stores -L and -U/d
          if (t == dpos) pk_dinv[dinv_slot[i]] = dinv;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();  // rows of the next level read the U rows written here (same CU: L1 is coherent for the workgroup)
  }
}

// ---- explicit block inverses -----------------------------------------------------------------------------------------
// P_b = U^-1 D^-1 L^-1 of one block (Ifpack storage: strict lower = L, diagonal = 1/d, strict upper = U/d), dense row-major.
// Thread j owns column j: forward sweep Y e_j = L^-1 e_j row by row, then the backward sweep in place.  Columns are
// independent, so there is no synchronisation; for a fixed row all threads read the same factor entries (broadcast)
// and consecutive entries of a row of P (coalesced).
// The same with the block's matrix in LDS while it is built (round 5): a thread owns a column and walks the rows; every row reads
// entries of its column that the same thread wrote a few rows earlier -- through global memory that is a store -> load round trip
// through L2 per row (0.76 ms per step for 511 blocks of <= 96 rows), through LDS a few hundred cycles.  Same operations on the same
// operands in the same order: bit-identical (tests/test_gpu_errors.py).
__global__ __launch_bounds__(128) void k_ilu_invert_lds(const int32_t *__restrict__ bptr, const int64_t *__restrict__ off,
                                                        const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                        const int32_t *__restrict__ diag, const double *__restrict__ lu, double *__restrict__ P) {
  extern __shared__ double Ps[];  // [n][n] row-major: column j of all rows by thread j, consecutive threads on consecutive banks
  __shared__ int s_rp[113], s_dg[112];  // the rows' entry ranges and diagonal positions: one trip for the block instead of one per row
  const int blk = blockIdx.x, r0 = bptr[blk], n = bptr[blk + 1] - r0;
  const int j = threadIdx.x;
  for (int t = threadIdx.x; t <= n; t += 128) s_rp[t] = rp[r0 + t];
  for (int t = threadIdx.x; t < n; t += 128) s_dg[t] = diag[r0 + t];
  __syncthreads();
  if (j < n) {
    for (int i = 0; i < n; ++i) {  // Y = L^-1 (unit lower)
      const int pe = s_dg[i];
      double acc = i == j ? 1.0 : 0.0;
      for (int p = s_rp[i]; p < pe; ++p) {
        const int k = ci[p] - r0;
        if (k >= 0) acc -= lu[p] * Ps[k * n + j];
      }
      Ps[i * n + j] = acc;
    }
    for (int i = n - 1; i >= 0; --i) {  // P = U^-1 (D^-1 Y), in place from the last row up
      const int pd = s_dg[i], pe = s_rp[i + 1];
      double acc = lu[pd] * Ps[i * n + j];
      for (int p = pd + 1; p < pe; ++p) {
        const int k = ci[p] - r0;
        if (k < n) acc -= lu[p] * Ps[k * n + j];
      }
      Ps[i * n + j] = acc;
    }
  }
  __syncthreads();
  double *Pb = P + off[blk];
  for (int q = threadIdx.x; q < n * n; q += 128) Pb[q] = Ps[q];
}

__global__ __launch_bounds__(256) void k_ilu_invert(const int32_t *__restrict__ bptr, const int64_t *__restrict__ off,
                                                    const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                    const int32_t *__restrict__ diag, const double *__restrict__ lu, double *P) {
  const int blk = blockIdx.x, r0 = bptr[blk], n = bptr[blk + 1] - r0;
  double *Pb = P + off[blk];
  for (int j = threadIdx.x; j < n; j += 256) {
    for (int i = 0; i < n; ++i) {  // Y = L^-1 (unit lower)
      const int row = r0 + i, pe = diag[row];
      double acc = i == j ? 1.0 : 0.0;
      for (int p = rp[row]; p < pe; ++p) {
        const int k = ci[p] - r0;
        if (k >= 0) acc -= lu[p] * Pb[(size_t)k * n + j];
      }
      Pb[(size_t)i * n + j] = acc;
    }
    for (int i = n - 1; i >= 0; --i) {  // P = U^-1 (D^-1 Y), in place from the last row up
      const int row = r0 + i, pd = diag[row], pe = rp[row + 1];
      double acc = lu[pd] * Pb[(size_t)i * n + j];
      for (int p = pd + 1; p < pe; ++p) {
        const int k = ci[p] - r0;
        if (k < n) acc -= lu[p] * Pb[(size_t)k * n + j];
      }
      Pb[(size_t)i * n + j] = acc;
    }
  }
}

// x_b = P_b b_b: 16 lanes per row, 64 row groups per workgroup (a ~100-row block is done in two rounds: the kernel is
// bound by load latency, not bytes, so the block gets as many loads in flight as the CU allows); optional b . x partial
constexpr int DENSE_THREADS = 1024;
__global__ __launch_bounds__(DENSE_THREADS) void k_ilu_apply_dense(const int32_t *__restrict__ bptr, const int64_t *__restrict__ off,
                                                                   const double *__restrict__ P, const double *b, double *x,
                                                                   double *__restrict__ dot_partial) {
  extern __shared__ double bs[];
  __shared__ double sh[DENSE_THREADS / 64];
  const int blk = blockIdx.x, r0 = bptr[blk], n = bptr[blk + 1] - r0;
  const double *Pb = P + off[blk];
  for (int t = threadIdx.x; t < n; t += DENSE_THREADS) bs[t] = b[r0 + t];
  __syncthreads();
  const int grp = threadIdx.x >> 4, lane = threadIdx.x & 15;
  double dot = 0.0;
  for (int i = grp; i < n; i += DENSE_THREADS / 16) {
    const double *row = Pb + (size_t)i * n;
    double a0 = 0.0, a1 = 0.0;
    int j = lane;
    for (; j + 16 < n; j += 32) {  // two independent chains, loads issued back to back
      a0 += row[j] * bs[j];
      a1 += row[j + 16] * bs[j + 16];
    }
    if (j < n) a0 += row[j] * bs[j];
    const double acc = lane_group_sum<16>(a0 + a1);
    if (lane == 0) {
      x[r0 + i] = acc;
      dot += bs[i] * acc;
    }
  }
  if (dot_partial) {
    dot = lane_group_sum<64>(dot);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < DENSE_THREADS / 64; ++k) t += sh[k];
      dot_partial[blk] = t;
    }
  }
}

// Same factorisation for blocks of at most ILU_DENSE_ROWS rows: every wave keeps its row as a DENSE vector over the block's
// columns in LDS (pos[c] = entry index of column c in the row, -1 outside the pattern), so an update is one LDS lookup
// instead of a binary search through global memory.  Same operations on the same entries in the same order.
constexpr int ILU_DENSE_ROWS = 512;
__global__ __launch_bounds__(ILU_WAVES * 64) void k_ilu_factor_small(const int32_t *__restrict__ bptr, const int32_t *__restrict__ lvl_off,
                                                                    const int32_t *__restrict__ lvl_ptr, const int32_t *__restrict__ lvl_rows,
                                                                    const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                                    const int32_t *__restrict__ diag, const double *__restrict__ a,
                                                                    double *__restrict__ lu, const int32_t *__restrict__ slot_of,
                                                                    double *__restrict__ pk_val, double *__restrict__ pk_dinv,
                                                                    int *__restrict__ err, const int32_t *__restrict__ dinv_slot) {
  __shared__ double wv[ILU_WAVES][ILU_DENSE_ROWS];
  __shared__ short posv[ILU_WAVES][ILU_DENSE_ROWS];
  const int blk = blockIdx.x, wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int r0 = bptr[blk], r1 = bptr[blk + 1], nb = r1 - r0;
  volatile double *w = wv[wave];  // indexed by block-local column
  volatile short *pos = posv[wave];
  for (int lv = lvl_off[blk]; lv < lvl_off[blk + 1]; ++lv) {
    const int l0 = lvl_ptr[lv], cnt = lvl_ptr[lv + 1] - l0;
    for (int r = wave; r < cnt; r += ILU_WAVES) {
      const int i = lvl_rows[l0 + r];
      const int p0 = rp[i], n = rp[i + 1] - p0, dpos = diag[i] - p0;
      for (int t = lane; t < nb; t += 64) pos[t] = -1;
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < n; t += 64) {
        const int c = ci[p0 + t] - r0;
        if (c >= 0 && c < nb) {
          pos[c] = (short)t;
          w[c] = a[p0 + t];
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = 0; t < dpos; ++t) {  // L part, ascending columns (wave-uniform loop)
        const int j = ci[p0 + t];
        if (j < r0) continue;
        const int dj = diag[j];
        const double mult = w[j - r0];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) w[j - r0] = mult * lu[dj];  // InV[jj] *= DV[j]
        const int ue = rp[j + 1];
        for (int q = dj + 1 + lane; q < ue; q += 64) {  // scaled U row of j
          const int c = ci[q] - r0;
          if (c >= nb) break;
          if (pos[c] >= 0) w[c] -= mult * lu[q];
        }
        __builtin_amdgcn_wave_barrier();
      }
      const double d = w[i - r0];
      const double dinv = 1.0 / d;
      if (lane == 0 && !(fabs(d) > 0.0)) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      for (int t = lane; t < n; t += 64) {
        const int j = ci[p0 + t], c = j - r0;
        const bool in = c >= 0 && c < nb;
        double v = in ? w[c] : 0.0;
        if (t == dpos) v = dinv;
        else if (t > dpos) v = in ? v * dinv : 0.0;
        lu[p0 + t] = v;
        if (slot_of) {
          const int sl = slot_of[p0 + t];
          if (sl >= 0) pk_val[sl] = -v;  // the stream adds value * x[col]: it stores -L and -U/d
          if (t == dpos) pk_dinv[dinv_slot[i]] = dinv;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();  // rows of the next level read the U rows written here (same CU: L1 is coherent for the workgroup)
  }
}

// Same factorisation with the in-block part of the whole block staged in LDS (values + 16-bit local columns in the compact
// numbering of IluSchedule::in_cptr, row starts, diagonal positions, the level lists): the elimination of a row then makes no
// trip through global memory at all -- k_ilu_factor_small chases ci -> diag -> lu -> ci/lu of the pivot row through L2 for every
// L entry.  Same operations on the same entries in the same order; rows of a level are dealt to the waves, a barrier per level.
// Dynamic LDS: ilu_factor_lds_bytes(max_rows, max_block_nnz).
constexpr uint16_t ILU_NOPOS = 0xffffu;
__host__ __device__ inline size_t ilu_factor_lds_bytes(int max_rows, int max_nz) {
  const size_t nz = ((size_t)max_nz + 3) & ~(size_t)3;
  return nz * 10 + (size_t)(max_rows + 4) * 2 * (4 + ILU_WAVES);
}
__global__ __launch_bounds__(ILU_WAVES * 64) void k_ilu_factor_lds(const int32_t *__restrict__ order, const int32_t *__restrict__ bptr, const int32_t *__restrict__ lvl_off,
                                                                  const int32_t *__restrict__ lvl_ptr, const int32_t *__restrict__ lvl_rows,
                                                                  const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                                  const int32_t *__restrict__ diag, const int32_t *__restrict__ in_lo,
                                                                  const int32_t *__restrict__ cptr, const int32_t *__restrict__ cpos,
                                                                  const double *__restrict__ a, double *__restrict__ lu,
                                                                  const int32_t *__restrict__ slot_of, double *__restrict__ pk_val,
                                                                  double *__restrict__ pk_dinv, int *__restrict__ err,
                                                                  const int32_t *__restrict__ dinv_slot, int max_rows, int max_nz) {
  extern __shared__ double lds_f[];
  const int nzp = (max_nz + 3) & ~3, mr = max_rows + 4;
  volatile double *lv = lds_f;                       // [nzp] the block's in-block entries, compact numbering
  volatile uint16_t *lc = (uint16_t *)(lds_f + nzp);  // [nzp] block-local column
  uint16_t *lrp = (uint16_t *)lc + nzp;              // [mr] compact row starts
  uint16_t *ldg = lrp + mr;                          // [mr] compact position of the diagonal
  uint16_t *llv = ldg + mr;                          // [mr] block-local rows in level order
  uint16_t *lvp = llv + mr;                          // [mr] level starts inside llv
  volatile uint16_t *posv = lvp + mr;                // [ILU_WAVES][mr] column -> compact position in the wave's current row
  const int blk = order[blockIdx.x], tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
  const int r0 = bptr[blk], nb = bptr[blk + 1] - r0;
  if (nb == 0) return;
  const int c0 = cptr[r0];
  const int L0 = lvl_off[blk], nlev = lvl_off[blk + 1] - L0, q0 = lvl_ptr[L0];
  for (int q = tid; q <= nb; q += ILU_WAVES * 64) lrp[q] = (uint16_t)(cptr[r0 + q] - c0);
  for (int q = tid; q < nb; q += ILU_WAVES * 64) {
    ldg[q] = (uint16_t)(cptr[r0 + q] - c0 + diag[r0 + q] - in_lo[r0 + q]);
    llv[q] = (uint16_t)(lvl_rows[q0 + q] - r0);  // the level lists of a block are contiguous and hold every row once
  }
  for (int q = tid; q <= nlev; q += ILU_WAVES * 64) lvp[q] = (uint16_t)(lvl_ptr[L0 + q] - q0);
  for (int q = tid; q < ILU_WAVES * mr; q += ILU_WAVES * 64) posv[q] = ILU_NOPOS;
  // staging and write-back run flat over the block's entries (compact number -> CSR position through cpos): every load address
  // depends on the loop counter or on one earlier load, so the requests of several passes are in flight together
  const int nzb = cptr[r0 + nb] - c0;
  {
    double *lvn = lds_f;
    uint16_t *lcn = (uint16_t *)(lds_f + nzp);
    const int P1 = rp[r0 + nb];
    for (int p = rp[r0] + tid; p < P1; p += ILU_WAVES * 64) {  // entries outside the block are not part of the rank's factor
      const int c = ci[p] - r0;
      if (c < 0 || c >= nb) lu[p] = 0.0;
    }
#pragma unroll 4
    for (int e = tid; e < nzb; e += ILU_WAVES * 64) {
      const int p = cpos[c0 + e];
      lvn[e] = a[p];
      lcn[e] = (uint16_t)(ci[p] - r0);
    }
  }
  __syncthreads();
  volatile uint16_t *pos = posv + wave * mr;
  for (int l = 0; l < nlev; ++l) {
    const int a1 = lvp[l + 1];
    for (int r = lvp[l] + wave; r < a1; r += ILU_WAVES) {
      const int i = llv[r];
      const int s0 = lrp[i], s1 = lrp[i + 1], dg = ldg[i];
      for (int e = s0 + lane; e < s1; e += 64) pos[lc[e]] = (uint16_t)e;
      __builtin_amdgcn_wave_barrier();
      // L part, ascending columns (wave-uniform loop).  Everything an entry needs except its own multiplier is a constant of the
      // pattern or belongs to a finished row -- pivot, its diagonal, this lane's entry of its scaled U row and where that lands in
      // row i -- and is requested one entry ahead: the dependent chain per L entry is multiplier -> update, two LDS trips.
      struct Ahead {
        int ue, q;
        uint16_t pc;
        double uq, dinvk;
      };
      auto ahead = [&](int e) {
        Ahead n;
        const int k = lc[e], dk = ldg[k];
        n.ue = lrp[k + 1];
        n.q = dk + 1 + lane;
        const bool has = n.q < n.ue;
        n.pc = has ? pos[lc[has ? n.q : dk]] : ILU_NOPOS;
        n.uq = lv[has ? n.q : dk];
        n.dinvk = lv[dk];
        return n;
      };
      Ahead cur = ahead(s0 < dg ? s0 : dg);
      for (int e = s0; e < dg; ++e) {
        const Ahead nxt = ahead(e + 1 < dg ? e + 1 : dg);  // (a dummy request behind the last entry: row i's own diagonal)
        const double mult = lv[e];
        if (lane == 0) lv[e] = mult * cur.dinvk;  // InV[jj] *= DV[j]
        if (cur.pc != ILU_NOPOS) lv[cur.pc] -= mult * cur.uq;
        for (int q = cur.q + 64; q < cur.ue; q += 64) {  // scaled U rows of more than 64 entries
          const uint16_t pc = pos[lc[q]];
          if (pc != ILU_NOPOS) lv[pc] -= mult * lv[q];
        }
        cur = nxt;
      }
      __builtin_amdgcn_wave_barrier();
      const double d = lv[dg];
      const double dinv = 1.0 / d;
      if (lane == 0 && !(fabs(d) > 0.0)) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_wave_barrier();
      for (int e = dg + lane; e < s1; e += 64) lv[e] = e == dg ? dinv : lv[e] * dinv;
      for (int e = s0 + lane; e < s1; e += 64) pos[lc[e]] = ILU_NOPOS;
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  {
    const double *lvn = lds_f;
#pragma unroll 4
    for (int e = tid; e < nzb; e += ILU_WAVES * 64) {
      const int p = cpos[c0 + e];
      const double v = lvn[e];
      lu[p] = v;
      if (slot_of) {
        const int sl = slot_of[p];
        if (sl >= 0) pk_val[sl] = -v;  // the stream adds value * x[col]: it stores -L and -U/d
      }
    }
    if (slot_of)
      for (int q = tid; q < nb; q += ILU_WAVES * 64) pk_dinv[dinv_slot[r0 + q]] = lvn[ldg[q]];
  }
}

// ---- levelled path (IluSchedule::levelled): blocks of any size, one launch per dependency level ------------------------
// Factorisation: one wave per row of the level, same row arithmetic as k_ilu_factor (sorted-column binary search in global
// memory); rows of earlier levels are complete because the previous launch has finished.
__global__ __launch_bounds__(ILU_WAVES * 64) void k_ilu_factor_level(int n_lvl_rows, const int32_t *__restrict__ rows,
                                                                    const int32_t *__restrict__ in_lo, const int32_t *__restrict__ in_hi,
                                                                    const int32_t *__restrict__ rp, const int32_t *__restrict__ ci,
                                                                    const int32_t *__restrict__ diag, const double *__restrict__ a,
                                                                    double *lu, int *__restrict__ err) {
  __shared__ double wv[ILU_WAVES][ILU_MAXROW];
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const int r = blockIdx.x * ILU_WAVES + wave;
  if (r >= n_lvl_rows) return;  // whole waves exit; no workgroup barrier below
  const int i = rows[r];
  volatile double *w = wv[wave];
  const int p0 = rp[i], n = rp[i + 1] - p0, dpos = diag[i] - p0, lo = in_lo[i] - p0, hi = in_hi[i] - p0;
  if (n > ILU_MAXROW) {
    if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  for (int t = lane; t < n; t += 64) w[t] = (t >= lo && t < hi) ? a[p0 + t] : 0.0;
  __builtin_amdgcn_wave_barrier();
  for (int t = lo; t < dpos; ++t) {  // L part, ascending columns (wave-uniform loop)
    const int j = ci[p0 + t];
    const int dj = diag[j];
    const double mult = w[t];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) w[t] = mult * lu[dj];  // InV[jj] *= DV[j]
    const int ue = in_hi[j];
    for (int q = dj + 1 + lane; q < ue; q += 64) {  // scaled U row of j
      const int c = ci[q];
      int l2 = t + 1, h2 = hi - 1;  // columns of row i are sorted; c > j
      while (l2 <= h2) {
        const int mid = (l2 + h2) >> 1, cm = ci[p0 + mid];
        if (cm < c) l2 = mid + 1; else if (cm > c) h2 = mid - 1; else { w[mid] -= mult * lu[q]; break; }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const double d = w[dpos];
  const double dinv = 1.0 / d;
  if (lane == 0 && !(fabs(d) > 0.0)) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int t = lane; t < n; t += 64) {
    double v = w[t];
    if (t == dpos) v = dinv;
    else if (t > dpos) v = t < hi ? v * dinv : 0.0;
    lu[p0 + t] = v;
  }
}

// Solve: LW lanes per row of the level, x in global memory (in place: x holds b on entry of the forward sweep).
// forward:  x_i -= sum_{j<i in block} L_ij x_j ;  backward: x_i = x_i / d_i - sum_{j>i in block} (U_ij/d_i) x_j
// (the D^-1 scaling of Ifpack's ApplyInverse is folded into the backward visit of each row: its x_j are final by then)
template <int NCOMP, int LW, bool FWD>
__global__ __launch_bounds__(256) void k_ilu_solve_level(int n_lvl_rows, const int4 *__restrict__ rec, const int32_t *__restrict__ ci,
                                                         const double *__restrict__ lu, double *x) {
  const int r = (blockIdx.x * 256 + threadIdx.x) / LW, lane = threadIdx.x % LW;
  if (r >= n_lvl_rows) return;
  const int4 q = rec[r];  // row, first entry, end of entries, diagonal: one trip (the records of a level are contiguous)
  const int i = q.x, pb = q.y, pe = q.z;
  // requested with the entries, not behind the reduction: the row's own value and (backward) its 1 / d
  double xi0[NCOMP];
  double *xi = x + (size_t)i * NCOMP;
  double dinv = 1.0;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) xi0[c] = xi[c];
    if (!FWD) dinv = lu[q.w];
  }
  double acc[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) acc[c] = 0.0;
  for (int p = pb + lane; p < pe; p += LW) {
    const double l = lu[p];
    const double *xj = x + (size_t)ci[p] * NCOMP;
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) acc[c] += l * xj[c];
  }
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) acc[c] = lane_group_sum<LW>(acc[c]);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) xi[c] = FWD ? xi0[c] - acc[c] : xi0[c] * dinv - acc[c];
  }
}

template <int NCOMP>
static void ilu_solve_levelled(nsx_handle *h, const DevCsr &g, const IluSchedule &s, const double *lu, const double *b, double *x) {
  constexpr int LW = 8;
  v_copy(h, g.n_rows() * NCOMP, x, b);
  const int nf = (int)s.gl_f_ptr_h.size() - 1, nbk = (int)s.gl_b_ptr_h.size() - 1;
  for (int l = 1; l < nf; ++l) {  // level 0 of the forward sweep has no in-block L entries: y = b there
    const int n = s.gl_f_ptr_h[l + 1] - s.gl_f_ptr_h[l];
    if (n > 0)
      hipLaunchKernelGGL((k_ilu_solve_level<NCOMP, LW, true>), dim3(cdiv((int64_t)n * LW, 256)), dim3(256), 0, h->stream, n,
                         reinterpret_cast<const int4 *>(s.gl_f_rec.p) + s.gl_f_ptr_h[l], g.colind.p, lu, x);
  }
  for (int l = 0; l < nbk; ++l) {
    const int n = s.gl_b_ptr_h[l + 1] - s.gl_b_ptr_h[l];
    if (n > 0)
      hipLaunchKernelGGL((k_ilu_solve_level<NCOMP, LW, false>), dim3(cdiv((int64_t)n * LW, 256)), dim3(256), 0, h->stream, n,
                         reinterpret_cast<const int4 *>(s.gl_b_rec.p) + s.gl_b_ptr_h[l], g.colind.p, lu, x);
  }
}

// The factorisation kernels report a failure (row too long, zero pivot) through a word in mapped host memory, written only
// when something is wrong; ilu_check() looks at it after the caller's next synchronisation (no stream sync, no copy here).
void ilu_check(nsx_handle *h) {
  volatile int *err = (volatile int *)(h->pub_host + N_SLOTS + 3);
  const int herr = *err;
  if (!herr) return;
  *err = 0;
  if (herr == 1) NSX_THROW(NSX_ERR_UNSUPPORTED, "ILU: a row has more than %d entries", ILU_MAXROW);
  if (herr == 3) NSX_THROW(NSX_ERR_HIP, "ILU: the lane-owner triangular solve found its LDS array away from address 0 and did not run (internal: a static __shared__ object in k_ilu_solve_lanes?)");
  NSX_THROW(NSX_ERR_NUMERIC, "ILU: zero pivot");
}

void ilu_factor(nsx_handle *h, const DevCsr &g, IluSchedule &s, const double *vals, double *lu, const char *name) {
  int *err = (int *)(h->pub_dev + N_SLOTS + 3);
  {
    LaunchScope ls(h, name, 20.0 * g.nnz() + 12.0 * g.n_rows());
    static const bool small_ok = !(getenv("NSX_ILU_SMALL") && atoi(getenv("NSX_ILU_SMALL")) == 0);
    const bool lds_ok = !(getenv("NSX_ILU_FACTOR_LDS") && atoi(getenv("NSX_ILU_FACTOR_LDS")) == 0);  // read per call: the tests switch it
    if (s.levelled) {
      for (size_t l = 0; l + 1 < s.gl_f_ptr_h.size(); ++l) {
        const int n = s.gl_f_ptr_h[l + 1] - s.gl_f_ptr_h[l];
        if (n > 0)
          hipLaunchKernelGGL(k_ilu_factor_level, dim3(cdiv(n, ILU_WAVES)), dim3(ILU_WAVES * 64), 0, h->stream, n, s.gl_f_rows.p + s.gl_f_ptr_h[l],
                             s.in_lo.p, s.in_hi.p, g.rowptr.p, g.colind.p, g.diag.p, vals, lu, err);
      }
    } else if (lds_ok && s.max_rows <= 4096 && s.max_block_nnz < 65535 && ilu_factor_lds_bytes(s.max_rows, s.max_block_nnz) <= 80 * 1024) {
      // the whole block in LDS: at most 80 KB, so that at least two workgroups share a CU
      const size_t lds = ilu_factor_lds_bytes(s.max_rows, s.max_block_nnz);
      static std::atomic<size_t> lds_allowed{64 * 1024};  // per process: the attribute belongs to the function, not to a handle
      if (lds > lds_allowed.load()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_ilu_factor_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        lds_allowed.store(80 * 1024);
      }
      hipLaunchKernelGGL(k_ilu_factor_lds, dim3(s.n_blocks), dim3(ILU_WAVES * 64), lds, h->stream, s.fac_order.p, s.block_ptr.p, s.blk_lvl_off.p, s.fwd_lvl_ptr.p,
                         s.fwd_rows.p, g.rowptr.p, g.colind.p, g.diag.p, s.in_lo.p, s.in_cptr.p, s.in_cpos.p, vals, lu,
                         s.packed_ok ? s.pk_slot_of.p : nullptr, s.pk_val.p, s.pk_dinv.p, err, s.pk_dinv_slot.p, s.max_rows, (int)s.max_block_nnz);
    } else if (small_ok && s.max_rows <= ILU_DENSE_ROWS)
      hipLaunchKernelGGL(k_ilu_factor_small, dim3(s.n_blocks), dim3(ILU_WAVES * 64), 0, h->stream, s.block_ptr.p, s.blk_lvl_off.p,
                         s.fwd_lvl_ptr.p, s.fwd_rows.p, g.rowptr.p, g.colind.p, g.diag.p, vals, lu, s.packed_ok ? s.pk_slot_of.p : nullptr,
                         s.pk_val.p, s.pk_dinv.p, err, s.pk_dinv_slot.p);
    else
      hipLaunchKernelGGL(k_ilu_factor, dim3(s.n_blocks), dim3(ILU_WAVES * 64), 0, h->stream, s.block_ptr.p, s.blk_lvl_off.p, s.fwd_lvl_ptr.p,
                         s.fwd_rows.p, g.rowptr.p, g.colind.p, g.diag.p, vals, lu, s.packed_ok ? s.pk_slot_of.p : nullptr, s.pk_val.p,
                         s.pk_dinv.p, err, s.pk_dinv_slot.p);
  }
  if (s.dense) {
    LaunchScope ls(h, "ilu_invert", 8.0 * (double)s.dn_entries + 12.0 * g.nnz());
    // blocks of up to 112 rows (98 KB of LDS: one workgroup per CU and half) keep the n x n matrix in LDS while it is built
    const bool inv_lds = !(getenv("NSX_ILU_INVERT_LDS") && atoi(getenv("NSX_ILU_INVERT_LDS")) == 0);  // read per call: the tests switch it inside one process
    const size_t shm = (size_t)s.max_rows * s.max_rows * sizeof(double);
    if (inv_lds && s.max_rows <= 112) {
      static bool attr_set = false;
      if (!attr_set && shm > 64 * 1024) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_ilu_invert_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 112 * (int)sizeof(double)));
        attr_set = true;
      }
      hipLaunchKernelGGL(k_ilu_invert_lds, dim3(s.n_blocks), dim3(128), shm, h->stream, s.block_ptr.p, s.dn_off.p, g.rowptr.p, g.colind.p, g.diag.p, lu, s.dn_P.p);
    } else {
      hipLaunchKernelGGL(k_ilu_invert, dim3(s.n_blocks), dim3(256), 0, h->stream, s.block_ptr.p, s.dn_off.p, g.rowptr.p, g.colind.p, g.diag.p, lu,
                         s.dn_P.p);
    }
  }
}

// x = U^{-1} D^{-1} L^{-1} b per block (Ifpack_ILU::ApplyInverse), NCOMP right-hand sides interleaved.
// LW lanes cooperate on one row; the block's part of x lives in LDS when it fits (USE_LDS), else in x itself.
template <int NCOMP, int LW, bool USE_LDS>
__global__ __launch_bounds__(256) void k_ilu_solve(const int32_t *__restrict__ bptr, const int32_t *__restrict__ offF,
                                                   const int32_t *__restrict__ ptrF, const int32_t *__restrict__ rowsF,
                                                   const int32_t *__restrict__ offB, const int32_t *__restrict__ ptrB,
                                                   const int32_t *__restrict__ rowsB, const int32_t *__restrict__ rp,
                                                   const int32_t *__restrict__ ci, const int32_t *__restrict__ diag,
                                                   const double *__restrict__ lu, const double *__restrict__ b, double *__restrict__ x) {
  extern __shared__ double xs_[];
  const int blk = blockIdx.x;
  const int r0 = bptr[blk], r1 = bptr[blk + 1], nloc = r1 - r0;
  double *xs = USE_LDS ? xs_ : x + (size_t)r0 * NCOMP;
  for (int t = threadIdx.x; t < nloc * NCOMP; t += 256) xs[t] = b[(size_t)r0 * NCOMP + t];
  __syncthreads();
  const int grp = threadIdx.x / LW, lane = threadIdx.x % LW, ngrp = 256 / LW;
  for (int lv = offF[blk]; lv < offF[blk + 1]; ++lv) {  // forward: y_i = b_i - sum_{j<i} L_ij y_j
    const int l0 = ptrF[lv], cnt = ptrF[lv + 1] - l0;
    for (int r = grp; r < cnt; r += ngrp) {
      const int i = rowsF[l0 + r];
      double acc[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = 0.0;
      const int e = diag[i];
      for (int p = rp[i] + lane; p < e; p += LW) {
        const int j = ci[p];
        if (j >= r0) {
          const double l = lu[p];
#pragma unroll
          for (int c = 0; c < NCOMP; ++c) acc[c] += l * xs[(j - r0) * NCOMP + c];
        }
      }
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = group_sum<LW>(acc[c]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) xs[(i - r0) * NCOMP + c] -= acc[c];
      }
    }
    __syncthreads();
  }
  for (int t = threadIdx.x; t < nloc; t += 256) {  // y *= D^{-1}
    const double dinv = lu[diag[r0 + t]];
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) xs[t * NCOMP + c] *= dinv;
  }
  __syncthreads();
  for (int lv = offB[blk]; lv < offB[blk + 1]; ++lv) {  // backward: x_i = y_i - sum_{j>i} U_ij x_j
    const int l0 = ptrB[lv], cnt = ptrB[lv + 1] - l0;
    for (int r = grp; r < cnt; r += ngrp) {
      const int i = rowsB[l0 + r];
      double acc[NCOMP];
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = 0.0;
      const int e = rp[i + 1];
      for (int p = diag[i] + 1 + lane; p < e; p += LW) {
        const int j = ci[p];
        if (j < r1) {
          const double u = lu[p];
#pragma unroll
          for (int c = 0; c < NCOMP; ++c) acc[c] += u * xs[(j - r0) * NCOMP + c];
        }
      }
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) acc[c] = group_sum<LW>(acc[c]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) xs[(i - r0) * NCOMP + c] -= acc[c];
      }
    }
    __syncthreads();
  }
  if (USE_LDS)
    for (int t = threadIdx.x; t < nloc * NCOMP; t += 256) x[(size_t)r0 * NCOMP + t] = xs[t];
}

// ---- lane-owner stream: device side in nsx_ilu_lanes.hpp (shared with tools/ilu_lanes_bench.hip), schedule in host/ilu_stream.hpp
constexpr int LANES_K = 12;  // rows per lane and memory trip in the load / scale / store passes of k_ilu_solve_lanes (768 rows: one trip for a wave of 8 blocks)

template <int NCOMP, int E, int PF>
__global__ __launch_bounds__(64) void k_ilu_solve_lanes(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ rows,
                                                        const int32_t *__restrict__ slab_ptr, const uint32_t *__restrict__ meta,
                                                        const double *__restrict__ val, const double *__restrict__ dinv, const double *b, double *x,
                                                        double *__restrict__ dot_partial, int *err_host, int force_guard) {
  extern __shared__ double xs[];  // the only LDS of this kernel: the stream's addresses are absolute (nsx_ilu_lanes.hpp)
  // The stream's 16-bit fields are absolute LDS byte addresses: the dynamic array must start at address 0, i.e. the kernel (and
  // every helper inlined into it) must own no static __shared__ object.  Should that ever change, x is NOT written -- so the word
  // ilu_check() reads is raised (code 3) and the API call that ran this solve fails instead of handing back stale memory
  // (force_guard: NSX_ILU_LDS_GUARD_TEST, tests/test_gpu_errors.py).
  if ((uint32_t)(uintptr_t)(lds_f64 *)xs != 0u || force_guard) {
    if (threadIdx.x == 0) __hip_atomic_store(err_host, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const int w = blockIdx.x;
  const unsigned lane = threadIdx.x;
  const int s0 = slab_ptr[2 * w], s1 = slab_ptr[2 * w + 1], s2 = slab_ptr[2 * w + 2];
  const int rb = row_ptr[w], nr = row_ptr[w + 1] - rb;
  // The wave is alone on its SIMD: nothing hides a trip to memory but the wave's own other requests.  So everything whose address
  // is known at once is requested at once, in front of everything else: the first slabs of the forward sweep, the row ids and the
  // inverse pivots of the wave's first 64 * LANES_K rows (both kept in registers to the end: the scaling between the sweeps and
  // the store at the end then need no further trip); then the right-hand side (one dependent trip).  Waves with more rows go
  // through the same three passes chunk by chunk for the rest.
  constexpr int K = LANES_K;
  LaneSlot<E> A[PF];
  lane_load<E, PF>(A, s0, meta, val, lane);
  int idx0[K];
  double d0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = 64 * k + (int)lane;
    idx0[k] = t < nr ? rows[rb + t] : -1;
    d0[k] = t < nr ? dinv[rb + t] : 0.0;
  }
  {
    double v[K][NCOMP];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) v[k][c] = idx0[k] >= 0 ? b[(size_t)idx0[k] * NCOMP + c] : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (idx0[k] >= 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) xs[(64 * k + (int)lane) * NCOMP + c] = v[k][c];
      }
  }
  for (int base = 64 * K; base < nr; base += 64 * K) {
    int idx[K];
    double v[K][NCOMP];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int t = base + 64 * k + (int)lane;
      idx[k] = t < nr ? rows[rb + t] : -1;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) v[k][c] = idx[k] >= 0 ? b[(size_t)idx[k] * NCOMP + c] : 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (idx[k] >= 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) xs[(base + 64 * k + (int)lane) * NCOMP + c] = v[k][c];
      }
  }
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) xs[(nr + lane) * NCOMP + c] = 0.0;  // the scratch rows of the idle slots
  const uint32_t scratch = (uint32_t)(nr + lane) * (8u * NCOMP);
  __builtin_amdgcn_wave_barrier();
  lane_sweep<NCOMP, E, PF>(A, s0, s1, meta, val, lane, scratch);  // y = L^{-1} b
  lane_load<E, PF>(A, s1, meta, val, lane);                       // (the backward sweep's first slabs fly during the scaling)
#pragma unroll
  for (int k = 0; k < K; ++k) {                                   // y *= D^{-1} (inverse pivots stored in the wave's row order)
    const int t = 64 * k + (int)lane;
    if (t < nr) {
#pragma unroll
      for (int c = 0; c < NCOMP; ++c) xs[t * NCOMP + c] *= d0[k];
    }
  }
  for (int base = 64 * K; base < nr; base += 64 * K) {
    double d[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int t = base + 64 * k + (int)lane;
      d[k] = t < nr ? dinv[rb + t] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int t = base + 64 * k + (int)lane;
      if (t < nr) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) xs[t * NCOMP + c] *= d[k];
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  lane_sweep<NCOMP, E, PF>(A, s1, s2, meta, val, lane, scratch);  // x = U^{-1} y
  double dot = 0.0;  // b . x over this wave's rows (CG's g.h right after the preconditioner, Prec.hpp:388 / SolverCG)
  for (int base = 0; base < nr; base += 64 * K) {
    int idx[K];
    double bv[K][NCOMP];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int t = base + 64 * k + (int)lane;
      idx[k] = base == 0 ? idx0[k] : (t < nr ? rows[rb + t] : -1);
    }
    if (dot_partial) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) bv[k][c] = idx[k] >= 0 ? b[(size_t)idx[k] * NCOMP + c] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (idx[k] >= 0) {
#pragma unroll
        for (int c = 0; c < NCOMP; ++c) {
          const double xv = xs[(base + 64 * k + (int)lane) * NCOMP + c];
          if (dot_partial) dot += bv[k][c] * xv;
          x[(size_t)idx[k] * NCOMP + c] = xv;
        }
      }
  }
  if (dot_partial) {
    dot = lane_group_sum<64>(dot);
    if (lane == 0) dot_partial[w] = dot;
  }
}

template <int NCOMP>
static void launch_lanes(nsx_handle *h, const IluSchedule &s, const double *b, double *x, double *dot_partial) {
  const size_t shm = (size_t)(s.max_wave_rows + 64) * NCOMP * sizeof(double);
  static const int pf = getenv("NSX_PF") ? atoi(getenv("NSX_PF")) : 8;
  int *err = (int *)(h->pub_dev + N_SLOTS + 3);  // the mapped word ilu_check() reads
  const int force_guard = getenv("NSX_ILU_LDS_GUARD_TEST") ? atoi(getenv("NSX_ILU_LDS_GUARD_TEST")) : 0;  // fault injection (tests), read per launch
#define NSX_GO(E_, PF_)                                                                                                                        \
  hipLaunchKernelGGL((k_ilu_solve_lanes<NCOMP, E_, PF_>), dim3(s.n_waves), dim3(64), shm, h->stream, s.pk_row_ptr.p, s.pk_rows.p, s.pk_slab_ptr.p, \
                     reinterpret_cast<const uint32_t *>(s.pk_meta.p), s.pk_val.p, s.pk_dinv.p, b, x, dot_partial, err, force_guard)
#define NSX_GO_E(E_)  \
  if (pf == 4) NSX_GO(E_, 4); else NSX_GO(E_, 8)
  switch (s.stream_epl) {
    case 1: NSX_GO_E(1); break;
    case 2: NSX_GO_E(2); break;
    case 3: NSX_GO_E(3); break;
    case 4: NSX_GO_E(4); break;
    default: NSX_THROW(NSX_ERR_ARG, "internal: %d entries per tick", s.stream_epl);
  }
#undef NSX_GO_E
#undef NSX_GO
}

template <int NCOMP>
static void launch_ilu_solve(nsx_handle *h, const DevCsr &g, const IluSchedule &s, const double *lu, const double *b, double *x) {
  const size_t shm = (size_t)s.max_rows * NCOMP * sizeof(double);
  constexpr int LW = 8;
#define NSX_ILU_ARGS s.block_ptr.p, s.blk_lvl_off.p, s.fwd_lvl_ptr.p, s.fwd_rows.p, s.blk_lvl_off_b.p, s.bwd_lvl_ptr.p, s.bwd_rows.p, \
                     g.rowptr.p, g.colind.p, g.diag.p, lu, b, x
  if (shm <= 64 * 1024)
    hipLaunchKernelGGL((k_ilu_solve<NCOMP, LW, true>), dim3(s.n_blocks), dim3(256), shm, h->stream, NSX_ILU_ARGS);
  else
    hipLaunchKernelGGL((k_ilu_solve<NCOMP, LW, false>), dim3(s.n_blocks), dim3(256), 0, h->stream, NSX_ILU_ARGS);
#undef NSX_ILU_ARGS
}

bool ilu_solve(nsx_handle *h, const DevCsr &g, const IluSchedule &s, const double *lu, const double *b, double *x, int ncomp,
               const char *name, int dot_slot) {
  if (s.levelled) {
    LaunchScope ls(h, name, 12.0 * (double)s.in_block_nnz + (double)g.n_rows() * (4 + 16.0 * ncomp));
    if (ncomp == 1) ilu_solve_levelled<1>(h, g, s, lu, b, x);
    else if (ncomp == 2) ilu_solve_levelled<2>(h, g, s, lu, b, x);
    else ilu_solve_levelled<3>(h, g, s, lu, b, x);
    return false;
  }
  const bool packed = s.packed_ok && s.stream_ncomp == ncomp;  // (the schedule already checked that a wave's rows fit 64 KiB of LDS)
  if (s.dense && ncomp == 1 && (size_t)s.max_rows * sizeof(double) <= 48 * 1024) {
    LaunchScope ls(h, name, 8.0 * (double)s.dn_entries + 16.0 * g.n_rows());
    const bool with_dot = dot_slot >= 0 && !h->comm && s.n_blocks >= 2 && s.n_blocks <= 512;  // a communicator needs equal counts on all ranks
    hipLaunchKernelGGL(k_ilu_apply_dense, dim3(s.n_blocks), dim3(DENSE_THREADS), (size_t)s.max_rows * sizeof(double), h->stream, s.block_ptr.p, s.dn_off.p,
                       s.dn_P.p, b, x, with_dot ? red_out(h, dot_slot, s.n_blocks) : nullptr);
    if (with_dot) after_reduction(h, dot_slot, s.n_blocks);
    return with_dot;
  }
  // algorithmic bytes (nnz(L+U) * 12 + n * (4 + 8 + 8) per component set): what ONE application of a per-rank ILU(0)
  // has to read -- the IN-BLOCK entries of the factor (diagonal included; couplings between ranks are dropped by Ifpack's
  // local filter and are never touched) + right-hand side and solution.  The packed stream itself moves 768 B per slab.
  LaunchScope ls(h, name, 12.0 * (double)s.in_block_nnz + (double)g.n_rows() * (4 + 16.0 * ncomp));
  if (packed) {
    const bool with_dot = dot_slot >= 0 && !h->comm && s.n_waves >= 2 && s.n_waves <= 512;  // one partial sum per wave
    double *dp = with_dot ? red_out(h, dot_slot, s.n_waves) : nullptr;
    if (ncomp == 1) launch_lanes<1>(h, s, b, x, dp);
    else if (ncomp == 2) launch_lanes<2>(h, s, b, x, dp);
    else launch_lanes<3>(h, s, b, x, dp);
    if (with_dot) after_reduction(h, dot_slot, s.n_waves);
    return with_dot;
  }
  if (ncomp == 1) launch_ilu_solve<1>(h, g, s, lu, b, x);
  else if (ncomp == 2) launch_ilu_solve<2>(h, g, s, lu, b, x);
  else launch_ilu_solve<3>(h, g, s, lu, b, x);
  return false;
}

}  // namespace nsx
