// main_cylinder.cpp — the reference's navier_stokes3D / navier_stokes2D executables on the C++ host mirror
// (reference Navier-Stokes/src/main3D.cpp:4-79, src/main2D.cpp:4-63).  Compiled twice: -DNSX_DIM=3 / -DNSX_DIM=2.
//   usage: navier_stokes{2,3}D [mesh.msh | level:N] [n_steps] [n_ranks]
#include <chrono>

#include "NavierStokes.hpp"

#ifndef NSX_DIM
#define NSX_DIM 3
#endif

int main(int argc, char *argv[]) {
  const std::string mesh_file_name = argc > 1 ? argv[1] : "level:1";  // reference default: ../mesh/Parallelepiped3D.msh (absent)
  const unsigned int degree_velocity = 2, degree_pressure = 1;
  const double deltat = NSX_DIM == 3 ? 2e-4 : 0.01;                  // main3D.cpp:38 / main2D.cpp:22
  double T = NSX_DIM == 3 ? 4.0 : 8.0;                                // main3D.cpp:37 / main2D.cpp:21
  if (argc > 2) T = std::atoi(argv[2]) * deltat;
  const int n_ranks = argc > 3 ? std::atoi(argv[3]) : 1;
  try {
    const auto t0 = std::chrono::steady_clock::now();
    nsx::NavierStokes<NSX_DIM> problem(mesh_file_name, degree_velocity, degree_pressure, T, deltat, 2, n_ranks);
    problem.setup();
    problem.solve();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time taken to solve ENTIRE Navier Stokes problem: " << wall << " seconds" << std::endl;
    // forces_results CSV of main3D.cpp:56-76 would be empty in the reference too; write the timings instead
    std::ofstream csv(NSX_DIM == 3 ? "timings_3D.csv" : "timings_2D.csv");
    csv << "step,gmres_iterations,time_prec,time_solve\n";
    for (size_t i = 0; i < problem.time_prec.size(); ++i)
      csv << i + 1 << ',' << problem.gmres_iterations[i] << ',' << problem.time_prec[i] << ',' << problem.time_solve[i] << '\n';
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
