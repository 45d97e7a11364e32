// Convergence.hpp — C++ host mirror of the reference's convergence executable class (Ethier-Steinmann manufactured
// solution on the cube [-1,1]^3) on top of the C-ABI (include/nsx.h) and the front-end (include/nsx_host.h).
//
//   Navier-Stokes/include/Convergence3D.hpp:12-300  (ExactSolution :51-148, FunctionH :159-174, class NavierStokes)
//   Navier-Stokes/src/Convergence3D.cpp             (setup :5-184, assemble :187-383, assemble_time_step :396-581,
//                                                    solve_time_step :583-723, solve :726-764, compute_error :766-794)
//   Navier-Stokes/src/main_convergence3D.cpp:5-86   (the loop over meshes and the convergence table)
//
// Reference quirks kept: the convection matrix is assembled twice in the first step
// (Conv.cpp:277,284 -> NSX_DOUBLE_CONVECTION), the Neumann datum is evaluated at t_n, not t_{n+1} (Conv.cpp:747-750),
// Temam's term stays on in every step, and the error is taken against the exact solution at T = 3e-4 while the state
// is at t = 4e-4 (Conv.cpp:774).  The Neumann face integral and the error norms are host-side loops over boundary
// faces / cells, like the Python driver (problem.py: neumann_rhs, velocity_error); the time step runs in libnsx.
#pragma once
#include <array>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/nsx.h"
#include "../../include/nsx_host.h"

namespace nsx {

// ExactSolution (Convergence3D.hpp:51-148): a = pi/4, b = pi/2
struct EthierSteinmann {
  double nu = 1e-2, a = M_PI / 4.0, b = M_PI / 2.0, time = 0.0;
  void set_time(double t) { time = t; }
  std::array<double, 3> velocity(const double *p) const {
    const double e = std::exp(-nu * b * b * time), x = p[0], y = p[1], z = p[2];
    return {-a * e * (std::exp(a * x) * std::sin(a * y + b * z) + std::exp(a * z) * std::cos(a * x + b * y)),
            -a * e * (std::exp(a * y) * std::sin(a * z + b * x) + std::exp(a * x) * std::cos(a * y + b * z)),
            -a * e * (std::exp(a * z) * std::sin(a * x + b * y) + std::exp(a * y) * std::cos(a * z + b * x))};
  }
  double pressure(const double *p) const {
    const double x = p[0], y = p[1], z = p[2], factor = -(a * a * std::exp(-2 * nu * b * b * time)) / 2.0;
    const double t1 = 2.0 * std::sin(a * x + b * y) * std::cos(a * z + b * x) * std::exp(a * (y + z));
    const double t2 = 2.0 * std::sin(a * y + b * z) * std::cos(a * x + b * y) * std::exp(a * (x + z));
    const double t3 = 2.0 * std::sin(a * z + b * x) * std::cos(a * y + b * z) * std::exp(a * (x + y));
    const double t4 = std::exp(2 * a * x) + std::exp(2 * a * y) + std::exp(2 * a * z);
    return factor * (t1 + t2 + t3 + t4);
  }
  // g[i][j] = d u_i / d x_j (Convergence3D.hpp:109-132)
  void gradient(const double *p, double g[3][3]) const {
    const double e = std::exp(-nu * b * b * time), x = p[0], y = p[1], z = p[2];
    const double ex = std::exp(a * x), ey = std::exp(a * y), ez = std::exp(a * z);
    g[0][0] = -a * e * (a * ex * std::sin(a * y + b * z) - a * ez * std::sin(a * x + b * y));
    g[0][1] = -a * e * (a * ex * std::cos(a * y + b * z) - b * ez * std::sin(a * x + b * y));
    g[0][2] = -a * e * (b * ex * std::cos(a * y + b * z) + a * ez * std::cos(a * x + b * y));
    g[1][0] = -a * e * (b * ey * std::cos(a * z + b * x) + a * ex * std::cos(a * y + b * z));
    g[1][1] = -a * e * (a * ey * std::sin(a * z + b * x) - a * ex * std::sin(a * y + b * z));
    g[1][2] = -a * e * (a * ey * std::cos(a * z + b * x) - b * ex * std::sin(a * y + b * z));
    g[2][0] = -a * e * (a * ez * std::cos(a * x + b * y) - b * ey * std::sin(a * z + b * x));
    g[2][1] = -a * e * (b * ez * std::cos(a * x + b * y) + a * ey * std::cos(a * z + b * x));
    g[2][2] = -a * e * (a * ez * std::sin(a * x + b * y) - a * ey * std::sin(a * z + b * x));
  }
  // FunctionH on the face y = -1 (Convergence3D.hpp:159-174)
  std::array<double, 3> neumann_h(const double *p) const {
    const double e = std::exp(-nu * b * b * time), x = p[0], y = p[1], z = p[2];
    return {-nu * a * e * (a * std::exp(a * x) * std::cos(a * y + b * z) - b * std::exp(a * z) * std::sin(a * x + b * y)),
            -nu * a * e * (a * std::exp(a * y) * std::sin(a * z + b * x) - a * std::exp(a * x) * std::sin(a * y + b * z)) - pressure(p),
            -nu * a * e * (b * std::exp(a * z) * std::cos(a * x + b * y) + a * std::exp(a * y) * std::cos(a * z + b * x))};
  }
};

class Convergence3D {
public:
  static constexpr int dim = 3;
  // n: cells per side of the cube (the reference reads mesh-cube-{1,2,5,10}.msh, which are not shipped: SURVEY D7)
  Convergence3D(int n_, const unsigned int &degree_velocity_, const unsigned int &degree_pressure_, const double &T_, const double &deltat_)
      : n(n_), T(T_), deltat(deltat_) {
    if (degree_velocity_ != 2 || degree_pressure_ != 1) throw std::runtime_error("only Taylor-Hood P2/P1 is supported");
    exact_solution.nu = nu;
  }
  ~Convergence3D() {
    if (h) nsx_destroy(h);
    if (face_tables) nsxh_tables_free(face_tables);
    if (dofs) nsxh_dofs_free(dofs);
    if (mesh) nsxh_mesh_free(mesh);
  }

  enum NormType { L2_norm, H1_norm };
  const double nu = 1e-2;            // Convergence3D.hpp:233
  unsigned int preconditioner_type = 0;
  double tol_abs = 1e-4, inner_rtol = 1e-2;  // Conv.cpp:587, Preconditioners.hpp:155 (exposed: the tests tighten them)
  bool verbose = true;
  std::vector<int> gmres_iterations;

  void setup() {  // Conv.cpp:5-184
    out() << "Initializing the mesh" << std::endl;
    mesh = nsxh_mesh_cube(n);
    if (!mesh) throw std::runtime_error("cannot build the cube mesh");
    out() << "  Number of elements = " << nsxh_mesh_n_cells(mesh) << std::endl;
    nsxh_tables *t = nsxh_tables_create(dim, 0, 0);
    dofs = nsxh_distribute_dofs(mesh);
    n_u = nsxh_n_u(dofs);
    n_p = nsxh_n_p(dofs);
    out() << "  Number of DoFs: velocity = " << n_u << " pressure = " << n_p << " total = " << n_u + n_p << std::endl;
    nsx_params p{dim, 0, nu, deltat};
    if (nsx_create(&p, &h)) throw std::runtime_error(nsx_last_error(nullptr));
    ck(nsx_set_tables(h, nsxh_tables_n_q(t), nsxh_tables_n_p2(t), nsxh_tables_n_p1(t), nsxh_tables_N2(t), nsxh_tables_dN2(t),
                      nsxh_tables_N1(t), nsxh_tables_weights(t)));
    nsxh_tables_free(t);
    ck(nsx_set_mesh(h, nsxh_mesh_n_cells(mesh), nsxh_dofs_per_cell(dofs), nsxh_cell_dofs(dofs), nsxh_cell_coords(dofs), n_u, n_p));
    face_tables = nsxh_tables_create(dim, 1, 0);
  }

  void solve() {  // Conv.cpp:726-764
    out() << "===============================================" << std::endl << "Applying the initial condition" << std::endl;
    const double *sp = nsxh_support_points(dofs);
    std::vector<double> u0((size_t)n_u + n_p);
    exact_solution.set_time(0.0);
    for (int i = 0; i < n_u; ++i) u0[i] = exact_solution.velocity(sp + (size_t)i * dim)[i % dim];
    for (int i = 0; i < n_p; ++i) u0[(size_t)n_u + i] = exact_solution.pressure(sp + (size_t)(n_u + i) * dim);
    ck(nsx_set_solution(h, u0.data()));
    unsigned int time_step = 0;
    double time = 0;
    while (time < T - 0.5 * deltat) {
      function_h.nu = nu;
      function_h.set_time(time);  // BEFORE the increment (Conv.cpp:747-750)
      time += deltat;
      ++time_step;
      out() << "n = " << std::setw(3) << time_step << ", t = " << std::setw(5) << time << ":" << std::flush;
      if (time_step == 1) assemble(time);
      else assemble_time_step(time);
      solve_time_step();
    }
  }

  // VectorTools::integrate_difference with the velocity mask + compute_global_error (Conv.cpp:766-794);
  // quadrature: the front-end's conical rule of order 5 (exact far beyond degree fe->degree + 2)
  double compute_error(const NormType &norm_type) {
    nsxh_tables *t = nsxh_tables_create(dim, 2, 5);
    const int nq = nsxh_tables_n_q(t), np2 = nsxh_tables_n_p2(t), nv = dim + 1;
    const double *N = nsxh_tables_N2(t), *dN = nsxh_tables_dN2(t), *w = nsxh_tables_weights(t), *xh = nsxh_tables_points(t);
    std::vector<double> sol((size_t)n_u + n_p);
    ck(nsx_get_solution_ghosted(h, sol.data()));
    exact_solution.set_time(T);  // the error is taken at T although the state is at the last step's time (Conv.cpp:774)
    const int32_t *cd = nsxh_cell_dofs(dofs);
    const double *cc = nsxh_cell_coords(dofs);
    const int dpc = nsxh_dofs_per_cell(dofs), nc = nsxh_mesh_n_cells(mesh);
    double err2 = 0.0;
    for (int c = 0; c < nc; ++c) {
      const double *X = cc + (size_t)c * nv * dim;
      double J[3][3], Ji[3][3];
      for (int d = 0; d < dim; ++d)
        for (int k = 0; k < dim; ++k) J[d][k] = X[(size_t)(k + 1) * dim + d] - X[d];
      const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                         J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det;
      Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
      Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det;
      Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
      Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det;
      Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
      Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det;
      Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
      Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
      for (int q = 0; q < nq; ++q) {
        double xq[3], uh[3] = {0, 0, 0}, gh[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int d = 0; d < dim; ++d) {
          xq[d] = X[d];
          for (int k = 0; k < dim; ++k) xq[d] += J[d][k] * xh[(size_t)q * dim + k];
        }
        for (int a = 0; a < np2; ++a) {
          const int base = a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv);
          double gphys[3];  // d N_a / d x_d = sum_k Ji[k][d] dNhat[k]
          for (int d = 0; d < dim; ++d) {
            gphys[d] = 0;
            for (int k = 0; k < dim; ++k) gphys[d] += Ji[k][d] * dN[((size_t)q * np2 + a) * dim + k];
          }
          for (int i = 0; i < dim; ++i) {
            const double U = sol[cd[(size_t)c * dpc + base + i]];
            uh[i] += N[(size_t)q * np2 + a] * U;
            for (int d = 0; d < dim; ++d) gh[i][d] += gphys[d] * U;
          }
        }
        const auto ue = exact_solution.velocity(xq);
        double e2 = 0;
        for (int i = 0; i < dim; ++i) e2 += (uh[i] - ue[i]) * (uh[i] - ue[i]);
        if (norm_type == H1_norm) {
          double ge[3][3];
          exact_solution.gradient(xq, ge);
          for (int i = 0; i < dim; ++i)
            for (int d = 0; d < dim; ++d) e2 += (gh[i][d] - ge[i][d]) * (gh[i][d] - ge[i][d]);
        }
        err2 += std::fabs(det) * w[q] * e2;
      }
    }
    nsxh_tables_free(t);
    return std::sqrt(err2);
  }

  std::vector<double> get_solution() const {
    std::vector<double> x((size_t)n_u + n_p);
    if (nsx_get_solution(h, x.data())) throw std::runtime_error(nsx_last_error(h));
    return x;
  }

protected:
  void assemble(const double &time) {  // Conv.cpp:187-383
    out() << "===============================================" << std::endl << "Assembling the system" << std::endl;
    ck(nsx_assemble(h, NSX_TEMAM | NSX_DOUBLE_CONVECTION));
    add_neumann();
    apply_dirichlet(time);
  }
  void assemble_time_step(const double &time) {  // Conv.cpp:396-581
    out() << "===============================================" << std::endl << "Assembling the system" << std::endl;
    ck(nsx_assemble_time_step(h, NSX_TEMAM));
    add_neumann();
    apply_dirichlet(time);
  }
  // cell_rhs(i) += scalar_product(h, phi_i) JxW on the faces with boundary id 3 (Conv.cpp:309-331, 506-528)
  void add_neumann() {
    static const int TETF[4][3] = {{0, 1, 2}, {1, 0, 3}, {0, 2, 3}, {2, 1, 3}};
    const int nbf = nsxh_mesh_n_bfaces(mesh), nv = dim + 1, np2 = nsxh_tables_n_p2(face_tables), nqf = nsxh_tables_n_qf(face_tables);
    const int32_t *bf = nsxh_mesh_bfaces(mesh), *ids = nsxh_mesh_bface_ids(mesh), *bc = nsxh_mesh_bface_cells(mesh), *cells = nsxh_mesh_cells(mesh);
    const double *V = nsxh_mesh_vertices(mesh), *N = nsxh_tables_N2(face_tables), *w = nsxh_tables_weights(face_tables),
                 *xh = nsxh_tables_points(face_tables);
    const int32_t *cd = nsxh_cell_dofs(dofs);
    const int dpc = nsxh_dofs_per_cell(dofs);
    std::map<int32_t, double> acc;
    for (int f = 0; f < nbf; ++f) {
      if (ids[f] != 3) continue;
      const int32_t *cv = cells + (size_t)bc[f] * nv;
      int lf = -1;
      for (int k = 0; k < 4 && lf < 0; ++k) {
        int match = 0;
        for (int a = 0; a < 3; ++a)
          for (int q = 0; q < 3; ++q) match += cv[TETF[k][a]] == bf[(size_t)f * 3 + q];
        if (match == 3) lf = k;
      }
      if (lf < 0) throw std::runtime_error("boundary face not found in its cell");
      const double *X0 = V + (size_t)cv[0] * dim;
      const double *A = V + (size_t)cv[TETF[lf][0]] * dim, *B = V + (size_t)cv[TETF[lf][1]] * dim, *C = V + (size_t)cv[TETF[lf][2]] * dim;
      const double e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
      const double cr[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const double area = 0.5 * std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
      for (int q = lf * nqf; q < (lf + 1) * nqf; ++q) {
        double xp[3];
        for (int d = 0; d < dim; ++d) {
          xp[d] = X0[d];
          for (int k = 0; k < dim; ++k) xp[d] += (V[(size_t)cv[k + 1] * dim + d] - X0[d]) * xh[(size_t)q * dim + k];
        }
        const auto hv = function_h.neumann_h(xp);
        for (int a = 0; a < np2; ++a) {
          const int base = a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv);
          for (int c = 0; c < dim; ++c) acc[cd[(size_t)bc[f] * dpc + base + c]] += w[q] * area * N[(size_t)q * np2 + a] * hv[c];
        }
      }
    }
    std::vector<int32_t> d;
    std::vector<double> v;
    for (const auto &kv : acc) {
      d.push_back(kv.first);
      v.push_back(kv.second);
    }
    ck(nsx_add_rhs(h, (int)d.size(), d.data(), v.data()));
  }
  void apply_dirichlet(const double &time) {  // Conv.cpp:363-380: ids 0,1,2,4,5 with the exact solution, id 3 is Neumann
    exact_solution.set_time(time);
    std::map<int32_t, double> boundary_values;
    const double *sp = nsxh_support_points(dofs);
    for (int id : {0, 1, 2, 4, 5}) {
      const int32_t *d;
      const int nd = nsxh_boundary_dofs(dofs, id, &d);
      for (int k = 0; k < nd; ++k) boundary_values[d[k]] = exact_solution.velocity(sp + (size_t)d[k] * dim)[d[k] % dim];
    }
    std::vector<int32_t> bd;
    std::vector<double> bv;
    for (const auto &kv : boundary_values) {
      bd.push_back(kv.first);
      bv.push_back(kv.second);
    }
    ck(nsx_apply_boundary_values(h, (int)bd.size(), bd.data(), bv.data()));
  }
  void solve_time_step() {  // Conv.cpp:583-723
    out() << "===============================================" << std::endl;
    nsx_solve_stats st;
    const int rc = nsx_solve_time_step(h, (int)preconditioner_type, tol_abs, inner_rtol, 100000, 100000, &st);
    if (rc) throw std::runtime_error(std::string("nsx: ") + nsx_last_error(h));
    gmres_iterations.push_back(st.outer_iterations);
    out() << "Result:  " << st.outer_iterations << " GMRES iterations" << std::endl;
  }
  void ck(int rc) const {
    if (rc) throw std::runtime_error(std::string("nsx: ") + nsx_last_error(h));
  }
  std::ostream &out() const {
    static std::ofstream null;
    return verbose ? std::cout : null;
  }

  const int n;
  const double T, deltat;
  EthierSteinmann exact_solution, function_h;
  nsxh_mesh *mesh = nullptr;
  nsxh_dofs *dofs = nullptr;
  nsxh_tables *face_tables = nullptr;
  nsx_handle *h = nullptr;
  int n_u = 0, n_p = 0;
};

}  // namespace nsx
