// main_convergence.cpp — the reference's `convergence` executable on the C++ host mirror
// (reference Navier-Stokes/src/main_convergence3D.cpp:5-86): Ethier-Steinmann on a sequence of cube meshes,
// L2 / H1 velocity errors, convergence.csv and the table with log2 reduction rates.
//   usage: convergence [n_1 n_2 ...]      cells per side of the cube meshes (default 2 4 8; the reference's
//                                         mesh-cube-{1,2,5,10}.msh are not shipped)
//   NSX_CONV_TOL="tol_abs inner_rtol" tightens the solver tolerances (default: the reference's 1e-4 / 1e-2)
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "Convergence.hpp"

int main(int argc, char *argv[]) {
  std::vector<int> sizes;
  for (int i = 1; i < argc; ++i) sizes.push_back(std::atoi(argv[i]));
  if (sizes.empty()) sizes = {2, 4, 8};
  const unsigned int degree_velocity = 2, degree_pressure = 1;
  const double T = 0.0003, deltat = 0.0004;  // main_convergence3D.cpp:34-35
  double tol_abs = 1e-4, inner_rtol = 1e-2;
  if (const char *e = std::getenv("NSX_CONV_TOL")) {
    std::istringstream is(e);
    is >> tol_abs >> inner_rtol;
  }
  try {
    const auto t0 = std::chrono::steady_clock::now();
    std::ofstream convergence_file("convergence.csv");
    convergence_file << "h,eL2,eH1" << std::endl;
    std::vector<double> h_vals, errors_L2, errors_H1;
    for (int n : sizes) {
      nsx::Convergence3D problem(n, degree_velocity, degree_pressure, T, deltat);
      problem.tol_abs = tol_abs;
      problem.inner_rtol = inner_rtol;
      problem.setup();
      problem.solve();
      const double error_L2 = problem.compute_error(nsx::Convergence3D::L2_norm);
      const double error_H1 = problem.compute_error(nsx::Convergence3D::H1_norm);
      h_vals.push_back(2.0 / n);
      errors_L2.push_back(error_L2);
      errors_H1.push_back(error_H1);
      convergence_file << std::setprecision(17) << 2.0 / n << "," << error_L2 << "," << error_H1 << std::endl;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time taken to solve ENTIRE Navier Stokes problem: " << wall << " seconds" << std::endl;
    // ConvergenceTable::evaluate_all_convergence_rates(reduction_rate_log2) + write_text
    std::cout << "h       L2            rate   H1            rate" << std::endl;
    for (size_t i = 0; i < h_vals.size(); ++i) {
      std::cout << std::fixed << std::setprecision(4) << h_vals[i] << "  " << std::scientific << std::setprecision(4) << errors_L2[i] << "  ";
      if (i == 0) std::cout << "-     ";
      else std::cout << std::fixed << std::setprecision(2) << std::log2(errors_L2[i - 1] / errors_L2[i]) << "  ";
      std::cout << std::scientific << std::setprecision(4) << errors_H1[i] << "  ";
      if (i == 0) std::cout << "-" << std::endl;
      else std::cout << std::fixed << std::setprecision(2) << std::log2(errors_H1[i - 1] / errors_H1[i]) << std::endl;
    }
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
