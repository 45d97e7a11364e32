// main_cylinder.cpp — the reference's navier_stokes3D / navier_stokes2D executables on the C++ host mirror
// (reference Navier-Stokes/src/main3D.cpp:4-79, src/main2D.cpp:4-63).  Compiled twice: -DNSX_DIM=3 / -DNSX_DIM=2.
//   usage: navier_stokes{2,3}D [mesh.msh | level:N] [n_steps] [n_ranks] [write_output 0|1] [u_m]
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
    const double u_m = argc > 5 ? std::atof(argv[5]) : (NSX_DIM == 3 ? 9.0 : 1.5);  // NavierStokes3D.hpp:80 / NavierStokes2D.hpp:80; 2.25: Re = 100 in 3D
    nsx::NavierStokes<NSX_DIM> problem(mesh_file_name, degree_velocity, degree_pressure, T, deltat, 2, n_ranks, u_m);
    if (argc > 4) problem.write_output = problem.write_csv = std::atoi(argv[4]) != 0;
    problem.setup();
    problem.solve();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "Time taken to solve ENTIRE Navier Stokes problem: " << wall << " seconds" << std::endl;
    {  // main3D.cpp:56-76 / main2D.cpp:40-60, with the Lift column holding the lift (the reference prints the lift coefficient twice)
      std::ofstream outputFile(NSX_DIM == 3 ? "forces_results_3D_2case.csv" : "forces_results_2D_2case.csv");
      if (!outputFile.is_open()) {
        std::cerr << "Error opening output file" << std::endl;
        return -1;
      }
      outputFile << "Iteration, Drag, Lift, Coeff Drag, CoeffLift, time prec, time solve" << std::endl;
      const size_t first = problem.time_prec.size() - problem.vec_drag.size();  // 3D: forces start after t = 0.1
      for (size_t ite = 0; ite < problem.vec_drag.size(); ite++)
        outputFile << (first + ite + 1) * deltat << ", " << problem.vec_drag[ite] << ", " << problem.vec_lift[ite] << ", " << problem.vec_drag_coeff[ite]
                   << ", " << problem.vec_lift_coeff[ite] << ", " << problem.time_prec[first + ite] << ", " << problem.time_solve[first + ite] << std::endl;
    }
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
