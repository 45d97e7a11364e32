// Translation unit of the adaptor's syntax check (tests/test_adaptor_header.py): includes the adaptor against the
// declaration-only interfaces of this directory and instantiates every template in it.  Compiled with -fsyntax-only, never linked.
#include <nsx_dealii_adaptor.hpp>

template class nsx::Binding<2>;
template class nsx::Binding<3>;
template class nsx::PreconditionSIMPLENsx<3>;
template class nsx::PreconditionaSIMPLENsx<2>;
template class nsx::PreconditionYosidaNsx<3>;
template class nsx::PreconditionaYosidaNsx<3>;

// the usage block of the header, as a maintainer would write it inside NavierStokes (members reduced to what the block touches)
template <int dim>
struct UsageSketch {
  dealii::DoFHandler<dim> dof_handler;
  dealii::FiniteElement<dim> *fe;
  dealii::Quadrature<dim> *quadrature;
  std::vector<dealii::IndexSet> block_owned_dofs;
  dealii::TrilinosWrappers::MPI::BlockVector solution_owned, solution;
  dealii::TrilinosWrappers::BlockSparseMatrix system_matrix;
  std::vector<double> time_prec, time_solve;
  dealii::ConditionalOStream pcout;
  nsx::Binding<dim> nsx_;
  std::map<dealii::types::global_dof_index, double> boundary_values;
  void setup() { nsx_.setup(dof_handler, *fe, *quadrature, 1e-3, 2e-4, block_owned_dofs, MPI_COMM_WORLD); }
  // the MPI path with an explicit layout request: 512 virtual ranks inside every MPI rank's node range, Schur blocks of <= 96 rows
  void setup_mpi_with_layout() { nsx_.setup(dof_handler, *fe, *quadrature, 1e-3, 2e-4, block_owned_dofs, MPI_COMM_WORLD, -1, 512, 96); }
  // the reference's own layout (one ILU(0) block per MPI rank): no internal layout
  void setup_reference_layout() { nsx_.setup(dof_handler, *fe, *quadrature, 1e-3, 2e-4, block_owned_dofs, MPI_COMM_WORLD, 0, 1); }
  int health() { return nsx_.report_persistent_state(pcout); }
  void paths() { nsx_.report_paths(std::cerr); }   // after the first solve_time_step, on every rank (MPI run with the internal layout: setup_mpi_with_layout)
  void solve_head() { nsx_.write_solution(solution_owned); }
  void assemble() {
    nsx_.assemble(NSX_TEMAM);
    nsx_.apply_boundary_values(boundary_values);
  }
  void assemble_time_step() {
    nsx_.assemble_time_step(0);
    nsx_.apply_boundary_values(boundary_values);
  }
  void solve_time_step() { nsx_.solve_time_step(NSX_PREC_YOSIDA, solution_owned, solution, time_prec, time_solve, pcout); }
  void inspect() { nsx_.export_matrix(0, system_matrix); }
  void with_dealii_gmres() {
    nsx::PreconditionYosidaNsx<dim> yosida(nsx_);
    yosida.initialize(system_matrix.block(0, 0), system_matrix.block(1, 0), system_matrix.block(0, 1), system_matrix.block(0, 0), solution_owned);
    yosida.vmult(solution, solution_owned);
  }
};
template struct UsageSketch<2>;
template struct UsageSketch<3>;
