// NavierStokes.hpp — C++ host mirror of the reference's `NavierStokes` class on top of the C-ABI (include/nsx.h)
// and the repository's own front-end (include/nsx_host.h).
//
// Same member names, argument meaning, call order and console output as the reference
//   Navier-Stokes/include/NavierStokes3D.hpp:10-252, NavierStokes2D.hpp, Convergence3D.hpp
//   Navier-Stokes/src/NavierStokes3D.cpp (setup :2-157, assemble :163-356, assemble_time_step :361-544,
//   solve_time_step :546-640, solve :687-741), NavierStokes2D.cpp, Convergence3D.cpp
// but every loop of the hot path runs in libnsx on the GPU.  Errors surface as C++ exceptions, as in the reference
// (SolverControl::NoConvergence -> nsx::NoConvergence, std::runtime_error for everything else).
#pragma once
#include <cmath>
#include <cstdio>
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

struct NoConvergence : std::runtime_error {  // SolverControl::NoConvergence
  unsigned int last_step;
  double last_residual;
  NoConvergence(unsigned int s, double r) : std::runtime_error("Iterative method reported convergence failure"), last_step(s), last_residual(r) {}
};

// Function<dim>-like functor of the inlet profile (NavierStokes3D.hpp:17-81 / NavierStokes2D.hpp:18-81).
template <int dim>
class InletVelocity {
public:
  explicit InletVelocity(int test_case_ = 2, double u_m_ = dim == 3 ? 9.0 : 1.5) : test_case(test_case_), u_m(u_m_) {}
  void set_time(double t) { time = t; }
  double get_time() const { return time; }
  double value(const double *p, unsigned int component = 0) const {
    if (component != 0 || test_case == 1) return 0.0;
    if (dim == 3) {
      const double v = 16.0 * u_m * p[1] * p[2] * (H - p[2]) * (H - p[1]) / (H * H * H * H);
      return test_case == 3 ? 16.0 * u_m * p[1] * p[2] * (H - p[2]) * (H - p[1]) * std::sin(M_PI * time / 8.0) / (H * H * H * H) : v;
    }
    if (test_case == 2) return 4.0 * u_m * p[1] * (H - p[1]) * std::sin(M_PI * time / 8.0) / (H * H);  // NavierStokes2D.hpp:33-34
    return 4.0 * u_m * p[1] * (H - p[1]) / (H * H);
  }
  double getMeanVelocity() const {  // NavierStokes3D.hpp:64-75, NavierStokes2D.hpp:64-76
    if (test_case == 1) return 0.0;
    const double k = dim == 3 ? 4.0 / 9.0 : 2.0 / 3.0;
    return test_case == 3 ? k * u_m * std::sin(time * M_PI / 8.0) : k * u_m;
  }

protected:
  int test_case;
  double H = 0.41;
  double u_m;
  double time = 0.0;
};

template <int dim>
class NavierStokes {
public:
  // mesh_file_name: a gmsh .msh file, or "level:N" for the built-in cylinder generator (no .msh ships with the reference)
  NavierStokes(const std::string &mesh_file_name_, const unsigned int &degree_velocity_, const unsigned int &degree_pressure_,
               const double &T_, const double &deltat_, const int test_case_ = 2, const int n_ranks_ = 1,
               const double u_m_ = dim == 3 ? 9.0 : 1.5)  // u_m: a hard-coded member in the reference (NavierStokes3D.hpp:80: Re = 400); 2.25 gives Re = 100
      : test_case(test_case_), inlet_velocity(test_case_, u_m_), T(T_), mesh_file_name(mesh_file_name_), degree_velocity(degree_velocity_),
        degree_pressure(degree_pressure_), deltat(deltat_), n_ranks(n_ranks_) {
    if (degree_velocity != 2 || degree_pressure != 1) throw std::runtime_error("only Taylor-Hood P2/P1 is supported");
  }
  ~NavierStokes() {
    if (h) nsx_destroy(h);
    if (dofs) nsxh_dofs_free(dofs);
    if (mesh) nsxh_mesh_free(mesh);
  }

  std::vector<double> time_prec, time_solve;  // NavierStokes3D.hpp:121-122
  std::vector<double> vec_drag, vec_lift, vec_drag_coeff, vec_lift_coeff;  // NavierStokes3D.hpp:116-119 (never filled there)
  std::vector<int> gmres_iterations;
  bool write_output = true;   // output(): VTU + PVTU record every 20 steps (3D) / every step (2D), like the reference
  bool write_csv = true;      // 2D: gmres.csv and coeff_2.csv appended per step (NavierStokes2D.cpp:623-636,679-692)
  unsigned int preconditioner_type = dim == 3 ? 0 : 3;  // NavierStokes3D.cpp:562 / NavierStokes2D.cpp:547
  double nu = 1e-3;                                     // NavierStokes3D.hpp:162
  bool verbose = true;

  void setup() {  // NavierStokes3D.cpp:2-157
    out() << "Initializing the mesh" << std::endl;
    if (mesh_file_name.rfind("level:", 0) == 0)
      mesh = nsxh_mesh_cylinder_level(dim, std::stoi(mesh_file_name.substr(6)));
    else
      mesh = nsxh_mesh_read_msh(mesh_file_name.c_str());
    if (!mesh || nsxh_mesh_dim(mesh) != dim) throw std::runtime_error("cannot read mesh " + mesh_file_name);
    if (n_ranks > 1 && nsxh_mesh_partition(mesh, 1, n_ranks)) throw std::runtime_error("partition failed");
    out() << "  Number of elements = " << nsxh_mesh_n_cells(mesh) << std::endl;
    out() << "Initializing the finite element space" << std::endl;
    nsxh_tables *t = nsxh_tables_create(dim, 0, 0);
    out() << "  DoFs per cell              = " << (dim == 3 ? 34 : 15) << std::endl;
    out() << "  Quadrature points per cell = " << nsxh_tables_n_q(t) << std::endl;
    out() << "Initializing the DoF handler" << std::endl;
    dofs = nsxh_distribute_dofs_ordered(mesh, NSXH_ORDER_COLOUR);  // shallow per-rank ILU(0) dependency graphs
    n_u = nsxh_n_u(dofs);
    n_p = nsxh_n_p(dofs);
    out() << "  Number of DoFs: " << std::endl << "    velocity = " << n_u << std::endl << "    pressure = " << n_p << std::endl
          << "    total    = " << n_u + n_p << std::endl;
    out() << "Initializing the linear system" << std::endl;
    nsx_params p{dim, 0, nu, deltat};
    if (nsx_create(&p, &h)) throw std::runtime_error(nsx_last_error(nullptr));
    ck(nsx_set_tables(h, nsxh_tables_n_q(t), nsxh_tables_n_p2(t), nsxh_tables_n_p1(t), nsxh_tables_N2(t), nsxh_tables_dN2(t),
                      nsxh_tables_N1(t), nsxh_tables_weights(t)));
    nsxh_tables_free(t);
    ck(nsx_set_mesh(h, nsxh_mesh_n_cells(mesh), nsxh_dofs_per_cell(dofs), nsxh_cell_dofs(dofs), nsxh_cell_coords(dofs), n_u, n_p));
    if (n_ranks > 1) ck(nsx_set_ranks(h, nsxh_n_subdomains(dofs), nsxh_owned_u_ptr(dofs), nsxh_owned_p_ptr(dofs)));
    setup_force_faces();
  }

  void solve() {  // NavierStokes3D.cpp:687-741
    out() << "===============================================" << std::endl << "Applying the initial condition" << std::endl;
    std::vector<double> u0((size_t)n_u + n_p, 0.0);  // u_0 = ZeroFunction (NavierStokes3D.hpp:200)
    ck(nsx_set_solution(h, u0.data()));
    output(0, {0.0, 0.0});  // NavierStokes3D.cpp:699, NavierStokes2D.cpp:712
    unsigned int time_step = 0;
    double time = 0;
    while (time < T - 0.5 * deltat) {
      time += deltat;
      ++time_step;
      inlet_velocity.set_time(time);
      out() << "n = " << std::setw(3) << time_step << ", t = " << std::setw(5) << time << ":" << std::flush;
      if (time == deltat) assemble(time);
      else assemble_time_step(time);
      solve_time_step();
      current_time = time;
      // NavierStokes3D.cpp:725-726, as written: an EXACT floating-point comparison of the accumulated time with T - deltat (true or false
      // by the rounding of the additions; the reference's own run decides it the same way, so the mirror does not "repair" it)
      if (dim == 3 && time == T - deltat) compute_pressure_difference();
      // NavierStokes3D.cpp:728-733 (forces only after t = 0.1); NavierStokes2D.cpp:737-741 (every step)
      std::vector<double> coefficients = {0.0, 0.0};
      if (dim == 2 || time > 0.1) {
        coefficients = compute_forces();
        c_D_max = std::max(coefficients[0], c_D_max);
        c_L_min = std::min(coefficients[1], c_L_min);
      }
      if (time_step % (dim == 3 ? 20 : 1) == 0) output(time_step, coefficients);  // NavierStokes3D.cpp:734, NavierStokes2D.cpp:743
    }
    out() << "===============================================" << std::endl
          << "Drag Coefficient Max ----->   " << c_D_max << std::endl
          << std::endl
          << "Lift Coefficient Min ----->   " << c_L_min << std::endl
          << "===============================================" << std::endl;
  }

  // NavierStokes::compute_forces (NavierStokes3D.cpp:744-846 / NavierStokes2D.cpp:752-859)
  std::vector<double> compute_forces() {
    out() << "===============================================" << std::endl << "Computing forces: " << std::endl;
    double drag = 0, lift = 0;
    ck(nsx_compute_forces(h, &drag, &lift));
    out() << "Drag :\t " << drag << " Lift :\t " << lift << std::endl;
    const double mean_v = inlet_velocity.getMeanVelocity();
    const double D = 0.1, H = 0.41, rho = 1.;
    const double den = dim == 3 ? rho * mean_v * mean_v * D * H : mean_v * mean_v * D;
    const double c_d = (2. * drag) / den, c_l = (2. * lift) / den;
    out() << "Coeff:\t " << c_d << " Coeff:\t " << c_l << std::endl;
    vec_drag.push_back(drag);
    vec_lift.push_back(lift);
    vec_drag_coeff.push_back(c_d);
    vec_lift_coeff.push_back(c_l);
    return {c_d, c_l};
  }

  // NavierStokes::output (NavierStokes3D.cpp:643-683; NavierStokes2D.cpp:642-695 also appends the coefficients to coeff_2.csv)
  void output(const unsigned int &time_step, const std::vector<double> &coeff = {0.0, 0.0}) const {
    if (write_output) {
      out() << "===============================================" << std::endl;
      const std::string output_file_name = dim == 3 ? "output-navier-stokes-3D" : "output-navier-stokes-2D";
      const std::vector<double> x = get_solution();
      if (nsxh_write_vtu(dofs, x.data(), dim == 3 ? "./outputConvergence/" : "./output2D_1/", output_file_name.c_str(), time_step))
        throw std::runtime_error("cannot write " + output_file_name);
      out() << "Output written to " << output_file_name << std::endl;
    }
    if (dim == 2 && write_csv) {
      std::ofstream coeff_file("coeff_2.csv", std::ios::app);
      if (coeff_file.is_open()) coeff_file << time_step << "," << coeff[0] << "," << coeff[1] << "\n";
      else out() << "Error: Unable to open coeff.csv for writing." << std::endl;
    }
    if (write_output) out() << "===============================================" << std::endl;
  }

  // NavierStokes::compute_pressure_difference (NavierStokes3D.cpp:849-923)
  double compute_pressure_difference() const {
    const double p_a[3] = {0.45, 0.2, 0.205}, p_e[3] = {0.55, 0.2, 0.205};
    const std::vector<double> x = get_solution();
    double p_diff = 0.0;
    nsxh_pressure_difference(dofs, x.data(), p_a, p_e, &p_diff);
    out() << "Pressure difference (P(A) - P(B)) = " << p_diff << std::endl;
    return p_diff;
  }
  double c_D_max = -999, c_L_min = 999;

  std::vector<double> get_solution() const {
    std::vector<double> x((size_t)n_u + n_p);
    if (nsx_get_solution(h, x.data())) throw std::runtime_error(nsx_last_error(h));
    return x;
  }

protected:
  void assemble(const double &time) {  // NavierStokes3D.cpp:163-356
    out() << "===============================================" << std::endl << "Assembling the system" << std::endl;
    ck(nsx_assemble(h, NSX_TEMAM));
    apply_dirichlet(time);
  }
  void assemble_time_step(const double &time) {  // NavierStokes3D.cpp:361-544 (Temam kept in 2D: NavierStokes2D.cpp:446)
    out() << "===============================================" << std::endl << "Assembling the system" << std::endl;
    ck(nsx_assemble_time_step(h, dim == 2 ? NSX_TEMAM : 0));
    apply_dirichlet(time);
  }
  void apply_dirichlet(const double &time) {  // NavierStokes3D.cpp:327-354, 515-542
    std::map<int32_t, double> boundary_values;
    inlet_velocity.set_time(time);
    const double *sp = nsxh_support_points(dofs);
    const int32_t *d;
    int n = nsxh_boundary_dofs(dofs, 0, &d);
    for (int k = 0; k < n; ++k) boundary_values[d[k]] = inlet_velocity.value(sp + (size_t)d[k] * dim, d[k] % dim);
    for (int id : {2, 3}) {  // zero_function on walls and obstacle; overwrites shared dofs like the second interpolate_boundary_values
      n = nsxh_boundary_dofs(dofs, id, &d);
      for (int k = 0; k < n; ++k) boundary_values[d[k]] = 0.0;
    }
    std::vector<int32_t> bd;
    std::vector<double> bv;
    for (const auto &kv : boundary_values) {
      bd.push_back(kv.first);
      bv.push_back(kv.second);
    }
    ck(nsx_apply_boundary_values(h, (int)bd.size(), bd.data(), bv.data()));
  }
  void solve_time_step() {  // NavierStokes3D.cpp:546-640
    out() << "===============================================" << std::endl;
    if (preconditioner_type > 3) throw std::runtime_error("Invalid preconditioner type");  // NavierStokes3D.cpp:633
    nsx_solve_stats st;
    const unsigned int inner_maxiter = (preconditioner_type == 1 || preconditioner_type == 3) ? 10000 : 100000;  // Preconditioners.hpp:155,259,368,482
    const int rc = nsx_solve_time_step(h, (int)preconditioner_type, 1e-4, 1e-2, 100000, (int)inner_maxiter, &st);
    if (rc == NSX_ERR_NOCONV) throw NoConvergence(st.outer_iterations, st.final_residual);
    ck(rc);
    out() << "Time taken to initialize preconditioner: " << st.t_prec << " seconds" << std::endl;
    out() << "Time taken to solve Navier Stokes problem: " << st.t_solve << " seconds" << std::endl;
    time_prec.push_back(st.t_prec);
    time_solve.push_back(st.t_solve);
    gmres_iterations.push_back(st.outer_iterations);
    out() << "Result:  " << st.outer_iterations << " GMRES iterations" << std::endl;
    if (dim == 2 && write_csv) {  // NavierStokes2D.cpp:622-636
      const int Re = int(0.1 * 1.5 * std::sin(inlet_velocity.get_time() * M_PI / 8.0) / .001);
      std::ofstream coeff_file("gmres.csv", std::ios::app);
      if (coeff_file.is_open()) coeff_file << inlet_velocity.get_time() << ',' << Re << ',' << st.outer_iterations << "\n";
      else out() << "Error: Unable to open coeff.csv for writing." << std::endl;
    }
  }

  void setup_force_faces() {  // faces with boundary id 3 and the face-quadrature tables (FEFaceValues of compute_forces)
    static const int TETF[4][3] = {{0, 1, 2}, {1, 0, 3}, {0, 2, 3}, {2, 1, 3}};
    static const int TRIF[3][2] = {{0, 1}, {1, 2}, {2, 0}};
    const int nbf = nsxh_mesh_n_bfaces(mesh);
    const int32_t *bf = nsxh_mesh_bfaces(mesh), *ids = nsxh_mesh_bface_ids(mesh), *bc = nsxh_mesh_bface_cells(mesh);
    const int32_t *cells = nsxh_mesh_cells(mesh);
    std::vector<int32_t> fc, fl;
    for (int f = 0; f < nbf; ++f) {
      if (ids[f] != 3) continue;
      const int32_t *cv = cells + (size_t)bc[f] * (dim + 1);
      for (int lf = 0; lf <= dim; ++lf) {
        int match = 0;
        for (int k = 0; k < dim; ++k) {
          const int32_t v = cv[dim == 3 ? TETF[lf][k] : TRIF[lf][k]];
          for (int q = 0; q < dim; ++q) match += v == bf[(size_t)f * dim + q];
        }
        if (match == dim) {
          fc.push_back(bc[f]);
          fl.push_back(lf);
          break;
        }
      }
    }
    nsxh_tables *t = nsxh_tables_create(dim, 1, 0);
    ck(nsx_set_force_faces(h, (int)fc.size(), fc.data(), fl.data(), nsxh_tables_n_qf(t), nsxh_tables_N2(t), nsxh_tables_dN2(t),
                           nsxh_tables_N1(t), nsxh_tables_weights(t)));
    nsxh_tables_free(t);
  }

  void ck(int rc) const {
    if (rc) throw std::runtime_error(std::string("nsx: ") + nsx_last_error(h));
  }
  std::ostream &out() const {
    static std::ofstream null;
    return verbose ? std::cout : null;
  }

  unsigned int test_case;
  InletVelocity<dim> inlet_velocity;
  const double T;
  const std::string mesh_file_name;
  const unsigned int degree_velocity, degree_pressure;
  const double deltat;
  const int n_ranks;
  nsxh_mesh *mesh = nullptr;
  nsxh_dofs *dofs = nullptr;
  nsx_handle *h = nullptr;
  int n_u = 0, n_p = 0;
  double current_time = 0.0;
};

}  // namespace nsx
