// nsx_forces.hip — drag / lift on the obstacle by face quadrature.
//
//   NavierStokes::compute_forces   reference Navier-Stokes/src/NavierStokes3D.cpp:744-846 (tangential-gradient formula, :795-826)
//                                  reference Navier-Stokes/src/NavierStokes2D.cpp:752-859 (stress-tensor formula, :816-838)
// One lane per (face, quadrature point); the contributions are written to a small buffer and summed on the host in a
// fixed order (the reference's own loop is a serial cell/face loop followed by an MPI sum, NS3D.cpp:830-831).
#include "nsx_internal.hpp"

namespace nsx {

template <int DIM>
__global__ void k_forces(int n_faces, int nqf, const int32_t *__restrict__ cells, const int32_t *__restrict__ lfaces, int n_cells,
                         int np2, int np1, const int32_t *__restrict__ cell_n2, const int32_t *__restrict__ cell_n1,
                         const double *__restrict__ geo, const double *__restrict__ N2f, const double *__restrict__ dN2f,
                         const double *__restrict__ N1f, const double *__restrict__ wf, const double *__restrict__ sol, int off_p,
                         double nu, double rho, double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_faces * nqf) return;
  const int f = t / nqf, q = t % nqf, cell = cells[f], lf = lfaces[f], tq = lf * nqf + q;
  double Ji[DIM][DIM];
  for (int k = 0; k < DIM; ++k)
    for (int d = 0; d < DIM; ++d) Ji[k][d] = geo[(size_t)(k * DIM + d) * n_cells + cell];
  const double adet = geo[(size_t)(DIM * DIM) * n_cells + cell];
  // outward reference normal of deal.II face lf (scaled so that |J^-T nref| * |det J| * mref is the face measure)
  double nref[3] = {0, 0, 0};
  double mref;
  if (DIM == 3) {
    mref = 0.5;
    if (lf == 0) nref[2] = -1; else if (lf == 1) nref[1] = -1; else if (lf == 2) nref[0] = -1; else nref[0] = nref[1] = nref[2] = 1;
  } else {
    mref = 1.0;
    if (lf == 0) nref[1] = -1; else if (lf == 1) nref[0] = nref[1] = 1; else nref[0] = -1;
  }
  double nv[DIM], len = 0.0;
  for (int d = 0; d < DIM; ++d) {
    double s = 0.0;
    for (int k = 0; k < DIM; ++k) s += Ji[k][d] * nref[k];
    nv[d] = s;
    len += s * s;
  }
  len = sqrt(len);
  const double jxw = wf[q] * adet * len * mref;
  double n[DIM];
  for (int d = 0; d < DIM; ++d) n[d] = -nv[d] / len;  // n = -fe_face_values.normal_vector(q)
  double p = 0.0;
  for (int v = 0; v < np1; ++v) p += N1f[tq * np1 + v] * sol[off_p + cell_n1[(size_t)v * n_cells + cell]];
  double G[DIM][DIM];
  for (int i = 0; i < DIM; ++i)
    for (int j = 0; j < DIM; ++j) G[i][j] = 0.0;
  for (int a = 0; a < np2; ++a) {
    const int node = cell_n2[(size_t)a * n_cells + cell];
    double g[DIM];
    for (int j = 0; j < DIM; ++j) {
      double s = 0.0;
      for (int k = 0; k < DIM; ++k) s += Ji[k][j] * dN2f[(tq * np2 + a) * DIM + k];
      g[j] = s;
    }
    for (int i = 0; i < DIM; ++i) {
      const double u = sol[(size_t)node * DIM + i];
      for (int j = 0; j < DIM; ++j) G[i][j] += u * g[j];
    }
  }
  double drag, lift;
  if (DIM == 3) {  // NS3D.cpp:803-826
    const double nx = n[0], ny = n[1];
    const double tang[3] = {ny, -nx, 0.0};
    const double t2 = ny * ny + nx * nx;
    double s = 0.0;
    for (int i = 0; i < DIM; ++i)
      for (int j = 0; j < DIM; ++j) s += n[i] * G[i][j] * (tang[j] / t2);
    drag = (rho * nu * s * ny - p * nx) * jxw;
    lift = -(rho * nu * s * nx + p * ny) * jxw;
  } else {  // NS2D.cpp:821-838: forces = (nu grad u - p I) n JxW
    double fo[DIM];
    for (int i = 0; i < DIM; ++i) {
      double s = 0.0;
      for (int j = 0; j < DIM; ++j) s += (nu * G[i][j] - (i == j ? p : 0.0)) * n[j];
      fo[i] = s * jxw;
    }
    drag = fo[0];
    lift = fo[1];
  }
  out[2 * t] = drag;
  out[2 * t + 1] = lift;
}

}  // namespace nsx

extern "C" {

int nsx_set_force_faces(nsx_handle *h, int n_faces, const int32_t *cells, const int32_t *local_faces, int n_qf, const double *N2f,
                        const double *dN2f, const double *N1f, const double *wf) {
  if (!h) return NSX_ERR_ARG;
  try {
    if (!h->have_mesh) NSX_THROW(NSX_ERR_ARG, "nsx_set_mesh first");
    if (n_faces < 0 || n_qf < 1 || (n_faces && (!cells || !local_faces)) || !N2f || !dN2f || !N1f || !wf) NSX_THROW(NSX_ERR_ARG, "bad face tables");
    HIP_CHECK(hipSetDevice(h->prm.device));
    const int nfr = h->dim + 1;
    for (int f = 0; f < n_faces; ++f)
      if (cells[f] < 0 || cells[f] >= h->n_cells || local_faces[f] < 0 || local_faces[f] >= nfr) NSX_THROW(NSX_ERR_ARG, "face %d: bad cell / local face", f);
    h->ff_n = n_faces;
    h->ff_nq = n_qf;
    h->ff_cells.upload(cells, n_faces, h->stream);
    h->ff_lf.upload(local_faces, n_faces, h->stream);
    h->ff_N2.upload(N2f, (size_t)nfr * n_qf * h->np2, h->stream);
    h->ff_dN2.upload(dN2f, (size_t)nfr * n_qf * h->np2 * h->dim, h->stream);
    h->ff_N1.upload(N1f, (size_t)nfr * n_qf * h->np1, h->stream);
    h->ff_w.upload(wf, n_qf, h->stream);
    h->ff_out.alloc((size_t)2 * std::max(1, n_faces * n_qf));
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

int nsx_compute_forces(nsx_handle *h, double *drag, double *lift) {
  if (!h || !drag || !lift) return NSX_ERR_ARG;
  try {
    if (h->ff_nq == 0) NSX_THROW(NSX_ERR_ARG, "nsx_set_force_faces first");
    HIP_CHECK(hipSetDevice(h->prm.device));
    const int n = h->ff_n * h->ff_nq;
    double sums[2] = {0.0, 0.0};
    if (n) {
      nsx::LaunchScope ls(h, "forces", 0);
      if (h->dim == 2)
        hipLaunchKernelGGL((nsx::k_forces<2>), dim3(nsx::cdiv(n, 128)), dim3(128), 0, h->stream, h->ff_n, h->ff_nq, h->ff_cells.p, h->ff_lf.p,
                           h->n_cells, h->np2, h->np1, h->cell_n2.p, h->cell_n1.p, h->geo.p, h->ff_N2.p, h->ff_dN2.p, h->ff_N1.p, h->ff_w.p,
                           h->sol.p, h->off_p, h->prm.nu, 1.0, h->ff_out.p);
      else
        hipLaunchKernelGGL((nsx::k_forces<3>), dim3(nsx::cdiv(n, 128)), dim3(128), 0, h->stream, h->ff_n, h->ff_nq, h->ff_cells.p, h->ff_lf.p,
                           h->n_cells, h->np2, h->np1, h->cell_n2.p, h->cell_n1.p, h->geo.p, h->ff_N2.p, h->ff_dN2.p, h->ff_N1.p, h->ff_w.p,
                           h->sol.p, h->off_p, h->prm.nu, 1.0, h->ff_out.p);
      std::vector<double> buf((size_t)2 * n);
      h->ff_out.download(buf.data(), buf.size(), h->stream);
      for (int t = 0; t < n; ++t) {  // fixed order: face by face, point by point
        sums[0] += buf[2 * t];
        sums[1] += buf[2 * t + 1];
      }
    }
    if (h->comm) {  // Utilities::MPI::sum (NS3D.cpp:830-831)
      h->scal_host[0] = sums[0];
      h->scal_host[1] = sums[1];
      HIP_CHECK(hipMemcpyAsync(h->scal.p, h->scal_host, 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
      h->slot_nb[0] = h->slot_nb[1] = 0;
      nsx::comm_allreduce_scalars(h, 0, 2);
      nsx::read_scalars(h, 0, 2, sums);
    }
    *drag = sums[0];
    *lift = sums[1];
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

}  // extern "C"
