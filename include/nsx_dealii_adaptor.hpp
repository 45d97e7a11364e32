// nsx_dealii_adaptor.hpp — the reference-side binding of libnsx.
//
// NEVER BUILT OR RUN AGAINST deal.II IN THIS REPOSITORY (UNVERIFIED): it needs deal.II >= 9.3.1 with Trilinos and MPI, none of
// which exist in the build image.  It is written against the public deal.II 9.3-9.5 API and the C-ABI of
// include/nsx.h, to be dropped into the reference as Navier-Stokes/include/nsx_dealii_adaptor.hpp and linked with -lnsx.
// tests/test_adaptor_header.py checks what can be checked without deal.II: that every nsx_* call below exists in include/nsx.h
// with the argument count used here, and that the header and its usage block are valid C++ whose use of the third-party types
// is consistent with the DECLARATION-ONLY interfaces of tests/stubs/ (g++ -fsyntax-only; written from memory of the public
// deal.II / Epetra / MPI headers, so a misremembered deal.II signature is not caught).
//
// Numbering the library relies on (checked by nsx_set_mesh, which fails with NSX_ERR_ARG otherwise): velocity dofs first and
// dof = dim * node + c for the dim components of a P2 node, pressure dofs after them — what DoFRenumbering::component_wise by
// block gives on FESystem(FE_SimplexP(2)^dim, FE_SimplexP(1)) (NavierStokes3D.cpp:62-69).  A DoFHandler renumbered otherwise
// (e.g. Cuthill-McKee across components) is rejected, not silently misread.
//
// What it replaces in lelecaruso/NavierStokes_Project_NM4PDE (file:line of the reference):
//   NavierStokes::assemble(time)            Navier-Stokes/include/NavierStokes3D.hpp:126-127, src/NavierStokes3D.cpp:163-356
//   NavierStokes::assemble_time_step(time)  include/NavierStokes3D.hpp:129-130,               src/NavierStokes3D.cpp:361-544
//   NavierStokes::solve_time_step()         include/NavierStokes3D.hpp:133-134,               src/NavierStokes3D.cpp:546-640
//   Precondition{SIMPLE,aSIMPLE,Yosida,aYosida}::initialize / ::vmult
//                                           include/Preconditioners.hpp:122-126,152-153 / 224-228,255-256 / 336-340,365-366 / 431-435,475-476
// One nsx handle per MPI rank (= per GPU).  With one rank the whole mesh goes through nsx_set_mesh; with several, every rank
// hands over its cells plus two layers of neighbours (nsx_set_mesh_distributed) and the library talks through the run's own
// communicator (nsx_comm_init_callbacks: MPI_Allreduce / MPI_Isend+Irecv on host buffers) or RCCL (nsx_comm_init).
//
// Usage inside the reference (three member bodies, the rest of the class untouched):
//
//   // NavierStokes3D.hpp, private:            nsx::Binding<dim> nsx_;
//   // end of NavierStokes::setup() (:157):    nsx_.setup(dof_handler, *fe, *quadrature, nu, deltat, block_owned_dofs, MPI_COMM_WORLD);
//   // NavierStokes::solve(), right after the initial interpolation `VectorTools::interpolate(dof_handler, u_0, solution_owned);
//   //   solution = solution_owned;` (:696-697):   nsx_.write_solution(solution_owned);    // the device starts from u_0, not from zero
//   void NavierStokes::assemble(const double &time) {            // :163-356
//     nsx_.assemble(NSX_TEMAM);                                   // Temam term in the first step (:255)
//     nsx_.apply_boundary_values(boundary_values_at(time));       // the std::map built exactly as at :327-352
//   }
//   void NavierStokes::assemble_time_step(const double &time) {   // :361-544   (2D / convergence: NSX_TEMAM, NavierStokes2D.cpp:446)
//     nsx_.assemble_time_step(0);
//     nsx_.apply_boundary_values(boundary_values_at(time));       // :515-541
//   }
//   void NavierStokes::solve_time_step() {                        // :546-640
//     nsx_.solve_time_step(NSX_PREC_YOSIDA, solution_owned, solution, time_prec, time_solve, pcout);
//   }
#ifndef NSX_DEALII_ADAPTOR_HPP
#define NSX_DEALII_ADAPTOR_HPP

#include <deal.II/base/conditional_ostream.h>
#include <deal.II/base/index_set.h>
#include <deal.II/base/mpi.h>
#include <deal.II/base/quadrature.h>
#include <deal.II/dofs/dof_handler.h>
#include <deal.II/fe/fe.h>
#include <deal.II/lac/solver_control.h>
#include <deal.II/lac/trilinos_block_sparse_matrix.h>
#include <deal.II/lac/trilinos_parallel_block_vector.h>
#include <deal.II/lac/trilinos_sparse_matrix.h>

#include <Epetra_CrsMatrix.h>
#include <Epetra_Map.h>
#include <mpi.h>
#include <nsx.h>

#include <algorithm>
#include <iostream>
#include <map>
#include <set>
#include <stdexcept>
#include <utility>
#include <vector>

namespace nsx {

using namespace dealii;

inline void ck(nsx_handle *h, int rc) {
  if (rc == NSX_OK) return;
  if (rc == NSX_ERR_ARG && h == nullptr) throw std::runtime_error(nsx_last_error(nullptr));
  throw std::runtime_error(nsx_last_error(h));  // includes "Invalid preconditioner type" (NavierStokes3D.cpp:633)
}

// ---- the run's own communicator behind the library's two collective operations (replaces Epetra_MpiComm traffic) ------
inline int allreduce_cb(void *ctx, double *v, int n) { return MPI_Allreduce(MPI_IN_PLACE, v, n, MPI_DOUBLE, MPI_SUM, *static_cast<MPI_Comm *>(ctx)); }
inline int exchange_cb(void *ctx, int n_nbr, const int *ranks, const double *const *send, const int *n_send, double *const *recv,
                       const int *n_recv) {
  std::vector<MPI_Request> rq(2 * static_cast<std::size_t>(n_nbr));
  for (int k = 0; k < n_nbr; ++k) {
    MPI_Irecv(recv[k], n_recv[k], MPI_DOUBLE, ranks[k], 7, *static_cast<MPI_Comm *>(ctx), &rq[2 * k]);
    MPI_Isend(const_cast<double *>(send[k]), n_send[k], MPI_DOUBLE, ranks[k], 7, *static_cast<MPI_Comm *>(ctx), &rq[2 * k + 1]);
  }
  return MPI_Waitall(2 * n_nbr, rq.data(), MPI_STATUSES_IGNORE);
}

template <int dim>
class Binding {
public:
  nsx_handle *h = nullptr;
  MPI_Comm comm = MPI_COMM_SELF;
  unsigned int rank = 0, world = 1;
  types::global_dof_index n_u = 0, n_p = 0;
  bool state_reported = false;

  Binding() = default;
  Binding(const Binding &) = delete;
  ~Binding() { if (h) nsx_destroy(h); }

  // Called once at the end of NavierStokes::setup() (NavierStokes3D.cpp:157).  block_owned_dofs as built at :71-87.
  // The DoFHandler keeps deal.II's own numbering (distribute_dofs + component_wise, :58-69) and the run keeps its own rank count.
  // virtual_ranks > 1 (default: ~85 velocity nodes per ILU(0) block, what the device kernels are laid out for): libnsx builds
  // that many spatially compact, colour-ordered rank blocks INSIDE this rank's node range behind the C-ABI
  // (nsx_set_internal_layout) -- nothing on the deal.II side is renumbered, every vector and dof list
  // that crosses the boundary stays in the DoFHandler's numbering.  virtual_ranks = 1 keeps the reference's own layout (one
  // Ifpack ILU per MPI rank, Preconditioners.hpp:215-216: exact, and orders of magnitude slower on a GPU).
  void setup(const DoFHandler<dim> &dh, const FiniteElement<dim> &fe, const Quadrature<dim> &quadrature, const double nu, const double deltat,
             const std::vector<IndexSet> &block_owned_dofs, MPI_Comm communicator, const int device = -1, int virtual_ranks = 0,
             const int schur_max_rows = 96) {
    comm = communicator;
    rank = Utilities::MPI::this_mpi_process(comm);
    world = Utilities::MPI::n_mpi_processes(comm);
    n_u = block_owned_dofs[0].size();
    n_p = block_owned_dofs[1].size();
    nsx_params prm{dim, device >= 0 ? device : int(rank % 8), nu, deltat};
    ck(nullptr, nsx_create(&prm, &h));
    set_tables(fe, quadrature);
    if (virtual_ranks == 0) virtual_ranks = std::max<int>(1, int(block_owned_dofs[0].n_elements() / dim / 85));
    // requested in front of the mesh: one set-up pass.  The layout is built from the cell tables handed over below.
    if (virtual_ranks > 1) ck(h, nsx_set_internal_layout(h, virtual_ranks, NSX_ORDER_COLOUR, schur_max_rows));
    // this rank's owned cells: dof indices and vertex coordinates, exactly what the reference's cell loops ask for (:304,494)
    const unsigned int dpc = fe.dofs_per_cell;
    std::vector<int32_t> my_dofs;
    std::vector<double> my_coords;
    std::vector<types::global_dof_index> idx(dpc);
    for (const auto &cell : dh.active_cell_iterators()) {
      if (!cell->is_locally_owned()) continue;
      cell->get_dof_indices(idx);
      my_dofs.insert(my_dofs.end(), idx.begin(), idx.end());
      for (unsigned int v = 0; v <= dim; ++v)
        for (unsigned int d = 0; d < dim; ++d) my_coords.push_back(cell->vertex(v)[d]);
    }
    // owned ranges per block: contiguous after DoFRenumbering::component_wise by block (:67-69)
    const std::pair<int32_t, int32_t> mine_u = contiguous_range(block_owned_dofs[0], 0), mine_p = contiguous_range(block_owned_dofs[1], n_u);
    if (world == 1) {
      ck(h, nsx_set_mesh(h, int(my_dofs.size() / dpc), int(dpc), my_dofs.data(), my_coords.data(), int(n_u), int(n_p)));
    } else {
      setup_distributed(dpc, my_dofs, my_coords, mine_u, mine_p);
      ck(h, nsx_comm_init_callbacks(h, int(rank), int(world), allreduce_cb, exchange_cb, &comm));
    }
  }

  // After the first solve_time_step: did the two single-launch kernels of the inner iteration (Gram-Schmidt sweep, Schur CG) run
  // as such?  A handle whose grid was not co-resident (another process on the device) has fallen back to one launch per operation:
  // correct, several times slower, and worth a line on the console.  Returns the number of fall-backs so far.
  int report_persistent_state(ConditionalOStream &pcout) const {
    int st[4] = {0, 0, 0, 0};
    ck(h, nsx_persistent_state(h, st));
    if (st[2] > 0 || (world == 1 && (!st[0] || !st[1])))
      pcout << "nsx: persistent kernels: sweep " << st[0] << ", Schur CG " << st[1] << ", time-outs " << st[2]
            << " -- running on the launch-per-operation path" << std::endl;
    return st[2];
  }

  // Which paths this rank's handle takes (nsx_path_info): every F->vmult through the LDS-staged SpMV (with MPI: the chunks without a
  // ghost column while the Epetra_Import-equivalent exchange is in flight, the others behind it), the Schur CG's variant (1 one launch
  // per operation, 2 one persistent launch, 3 two launches per iteration), neighbours and ghost nodes.  One line per rank.
  void report_paths(std::ostream &out) const {
    int p[32];
    ck(h, nsx_path_info(h, p));
    out << "nsx rank " << rank << ": LDS-staged SpMV " << p[0] << " (" << p[1] << " chunks, " << p[2] << " behind the ghost exchange), sweep " << p[3]
        << " entries per thread on " << p[4] << " workgroups, Schur CG path " << p[8] << " on " << p[9] << " blocks, " << p[10] << " neighbours, " << p[12]
        << " ghost nodes" << std::endl;
  }

  // ---- the three members -------------------------------------------------------------------------------------------
  void assemble(const int flags) { ck(h, nsx_assemble(h, flags)); }                      // NavierStokes3D.cpp:163-324
  void assemble_time_step(const int flags) { ck(h, nsx_assemble_time_step(h, flags)); }  // :361-512
  // MatrixTools::apply_boundary_values(boundary_values, system_matrix, solution, system_rhs, false)   (:353, :541)
  void apply_boundary_values(const std::map<types::global_dof_index, double> &boundary_values) {
    std::vector<int32_t> d;
    std::vector<double> v;
    d.reserve(boundary_values.size());
    v.reserve(boundary_values.size());
    for (const auto &kv : boundary_values) {
      d.push_back(int32_t(kv.first));
      v.push_back(kv.second);
    }
    ck(h, nsx_apply_boundary_values(h, int(d.size()), d.data(), v.data()));
  }
  // NavierStokes::solve_time_step (:546-640): same outputs — solution_owned, solution, time_prec, time_solve, the console line
  void solve_time_step(const int preconditioner_type, TrilinosWrappers::MPI::BlockVector &solution_owned, TrilinosWrappers::MPI::BlockVector &solution,
                       std::vector<double> &time_prec, std::vector<double> &time_solve, ConditionalOStream &pcout, const double tol_abs = 1e-4,
                       const double inner_rtol = 1e-2, const int maxiter = 100000, const int inner_maxiter = 100000) {
    nsx_solve_stats st;
    const int rc = nsx_solve_time_step(h, preconditioner_type, tol_abs, inner_rtol, maxiter, inner_maxiter, &st);
    if (rc == NSX_ERR_NOCONV) throw SolverControl::NoConvergence(st.outer_iterations, st.final_residual);
    ck(h, rc);
    time_prec.push_back(st.t_prec);    // :572
    time_solve.push_back(st.t_solve);  // :577
    pcout << "Result:  " << st.outer_iterations << " GMRES iterations" << std::endl;  // :636
    if (!state_reported) {
      state_reported = true;
      report_persistent_state(pcout);
    }
    read_solution(solution_owned);
    solution = solution_owned;  // :638 (ghost import on the deal.II side for output() / compute_forces())
  }
  // state in / out: global numbering, every rank touches the entries it owns
  void write_solution(const TrilinosWrappers::MPI::BlockVector &solution_owned) {
    std::vector<double> x(n_u + n_p, 0.0);
    for (const auto i : solution_owned.locally_owned_elements()) x[i] = solution_owned[i];
    ck(h, nsx_set_solution(h, x.data()));
  }
  void read_solution(TrilinosWrappers::MPI::BlockVector &solution_owned) const {
    std::vector<double> x(n_u + n_p, 0.0);
    ck(h, nsx_get_solution(h, x.data()));
    for (const auto i : solution_owned.locally_owned_elements()) solution_owned[i] = x[i];
    solution_owned.compress(VectorOperation::insert);
  }

  // ---- export back into the reference's Trilinos objects (one-rank handles: nsx_export_block) ------------------------
  // which: 0 system_matrix, 1 mass_matrix, 2 convection_matrix, 3 stiffness_matrix, 4 pressure_mass (NavierStokes3D.hpp:230-239)
  void export_matrix(const int which, TrilinosWrappers::BlockSparseMatrix &M) const {
    for (unsigned int br = 0; br < 2; ++br)
      for (unsigned int bc = 0; bc < 2; ++bc) {
        const int block = which == 4 ? 3 : int(2 * br + bc);
        if ((which == 4) != (br == 1 && bc == 1)) continue;  // pressure_mass lives in (1,1), everything else has an empty (1,1)
        Epetra_CrsMatrix &E = const_cast<Epetra_CrsMatrix &>(M.block(br, bc).trilinos_matrix());
        int *rowptr = nullptr, *lcol = nullptr;
        double *vals = nullptr;
        if (E.ExtractCrsDataPointers(rowptr, lcol, vals) != 0) throw std::runtime_error("nsx: Epetra matrix not in optimized storage");
        const int n_rows = E.NumMyRows(), nnz = rowptr[n_rows];
        std::vector<int32_t> gcol(nnz);
        for (int k = 0; k < nnz; ++k) gcol[k] = int32_t(E.ColMap().GID(lcol[k]));  // block-local global column ids
        ck(h, nsx_export_block(h, which, block, n_rows, rowptr, gcol.data(), vals));
      }
  }

private:
  static std::pair<int32_t, int32_t> contiguous_range(const IndexSet &owned, const types::global_dof_index offset) {
    if (owned.n_elements() == 0) return {int32_t(offset), int32_t(offset)};
    if (!owned.is_contiguous()) throw std::runtime_error("nsx: owned dofs of a block are not one contiguous range (component_wise renumbering missing?)");
    return {int32_t(owned.nth_index_in_set(0)), int32_t(owned.nth_index_in_set(0) + owned.n_elements())};
  }

  // FEValues tables on the reference cell, as data: the user's own FE_SimplexP spaces and QGaussSimplex rule (:31-50)
  void set_tables(const FiniteElement<dim> &fe, const Quadrature<dim> &q) {
    const unsigned int n_q = q.size(), n_p2 = (dim == 2 ? 6 : 10), n_p1 = dim + 1;
    const FiniteElement<dim> &fe_u = fe.base_element(0), &fe_p = fe.base_element(1);
    std::vector<double> N2(n_q * n_p2), dN2(n_q * n_p2 * dim), N1(n_q * n_p1), w(q.get_weights());
    for (unsigned int k = 0; k < n_q; ++k) {
      for (unsigned int a = 0; a < n_p2; ++a) {
        N2[k * n_p2 + a] = fe_u.shape_value(a, q.point(k));
        for (unsigned int d = 0; d < dim; ++d) dN2[(k * n_p2 + a) * dim + d] = fe_u.shape_grad(a, q.point(k))[d];
      }
      for (unsigned int v = 0; v < n_p1; ++v) N1[k * n_p1 + v] = fe_p.shape_value(v, q.point(k));
    }
    ck(h, nsx_set_tables(h, int(n_q), int(n_p2), int(n_p1), N2.data(), dN2.data(), N1.data(), w.data()));
  }

  // The MPI path.  A fullydistributed triangulation keeps one ghost layer; the Schur product of an owned pressure row
  // reaches one layer further, so the cell tables of all ranks are gathered once (a set-up cost of ~230 B per cell) and
  // every rank picks: layer 1 = cells touching a P2 node it owns, layer 2 = cells touching a node of layer 1.
  void setup_distributed(const unsigned int dpc, const std::vector<int32_t> &my_dofs, const std::vector<double> &my_coords,
                         const std::pair<int32_t, int32_t> mine_u, const std::pair<int32_t, int32_t> mine_p) {
    const auto all_dofs = Utilities::MPI::all_gather(comm, my_dofs);
    const auto all_coords = Utilities::MPI::all_gather(comm, my_coords);
    const auto all_u = Utilities::MPI::all_gather(comm, mine_u), all_p = Utilities::MPI::all_gather(comm, mine_p);
    std::vector<int32_t> gpu_u_ptr(world + 1, 0), gpu_p_ptr(world + 1, 0);  // node ranges per rank
    for (unsigned int r = 0; r < world; ++r) {
      gpu_u_ptr[r + 1] = all_u[r].second / dim;
      gpu_p_ptr[r + 1] = all_p[r].second - int32_t(n_u);
    }
    gpu_u_ptr[0] = all_u[0].first / dim;
    gpu_p_ptr[0] = all_p[0].first - int32_t(n_u);
    const unsigned int nv = dim + 1, n_p2 = (dim == 2 ? 6 : 10);
    auto p2_node = [&](const int32_t *cd, unsigned int a) { return cd[a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv)] / dim; };
    auto owned_u = [&](int32_t node) { return node >= gpu_u_ptr[rank] && node < gpu_u_ptr[rank + 1]; };
    // pass 1: layer 1 and the set of its P2 nodes; pass 2: layer 2
    std::vector<std::pair<unsigned int, unsigned int>> layer1, layer2;  // (source rank, cell index there)
    std::set<int32_t> nodes1;
    for (unsigned int r = 0; r < world; ++r)
      for (std::size_t c = 0; c * dpc < all_dofs[r].size(); ++c) {
        const int32_t *cd = all_dofs[r].data() + c * dpc;
        bool touch = false;
        for (unsigned int a = 0; a < n_p2 && !touch; ++a) touch = owned_u(p2_node(cd, a));
        if (touch) {
          layer1.emplace_back(r, unsigned(c));
          for (unsigned int a = 0; a < n_p2; ++a) nodes1.insert(p2_node(cd, a));
        }
      }
    for (unsigned int r = 0; r < world; ++r)
      for (std::size_t c = 0; c * dpc < all_dofs[r].size(); ++c) {
        const int32_t *cd = all_dofs[r].data() + c * dpc;
        bool own = false, touch = false;
        for (unsigned int a = 0; a < n_p2; ++a) {
          own = own || owned_u(p2_node(cd, a));
          touch = touch || nodes1.count(p2_node(cd, a)) > 0;
        }
        if (touch && !own) layer2.emplace_back(r, unsigned(c));
      }
    std::vector<int32_t> cell_dofs;
    std::vector<double> cell_coords;
    std::set<int32_t> ghost_u, ghost_p;
    for (const auto *layer : {&layer1, &layer2})
      for (const auto &rc : *layer) {
        const int32_t *cd = all_dofs[rc.first].data() + std::size_t(rc.second) * dpc;
        const double *cc = all_coords[rc.first].data() + std::size_t(rc.second) * nv * dim;
        cell_dofs.insert(cell_dofs.end(), cd, cd + dpc);
        cell_coords.insert(cell_coords.end(), cc, cc + nv * dim);
        for (unsigned int a = 0; a < n_p2; ++a)
          if (!owned_u(p2_node(cd, a))) ghost_u.insert(p2_node(cd, a));
        for (unsigned int v = 0; v < nv; ++v) {
          const int32_t node = cd[(dim + 1) * v + dim] - int32_t(n_u);
          if (node < gpu_p_ptr[rank] || node >= gpu_p_ptr[rank + 1]) ghost_p.insert(node);
        }
      }
    // halo plan: tell every owner which of its nodes this rank needs; what the others need from this rank are the send lists
    auto owner_of = [](const std::vector<int32_t> &ptr, int32_t node) { return unsigned(std::upper_bound(ptr.begin(), ptr.end(), node) - ptr.begin()) - 1; };
    std::map<unsigned int, std::vector<int32_t>> need_u, need_p;
    for (const int32_t g : ghost_u) need_u[owner_of(gpu_u_ptr, g)].push_back(g);
    for (const int32_t g : ghost_p) need_p[owner_of(gpu_p_ptr, g)].push_back(g);
    const auto got_u = Utilities::MPI::some_to_some(comm, need_u), got_p = Utilities::MPI::some_to_some(comm, need_p);
    std::set<unsigned int> nbr_set;
    for (const auto &kv : need_u) nbr_set.insert(kv.first);
    for (const auto &kv : need_p) nbr_set.insert(kv.first);
    for (const auto &kv : got_u) nbr_set.insert(kv.first);
    for (const auto &kv : got_p) nbr_set.insert(kv.first);
    std::vector<int32_t> nbr(nbr_set.begin(), nbr_set.end()), send_u_ptr{0}, send_u_nodes, send_p_ptr{0}, send_p_nodes;
    for (const int32_t r : nbr) {
      const auto iu = got_u.find(unsigned(r)), ip = got_p.find(unsigned(r));
      if (iu != got_u.end()) send_u_nodes.insert(send_u_nodes.end(), iu->second.begin(), iu->second.end());  // ascending: std::set order
      if (ip != got_p.end()) send_p_nodes.insert(send_p_nodes.end(), ip->second.begin(), ip->second.end());
      send_u_ptr.push_back(int32_t(send_u_nodes.size()));
      send_p_ptr.push_back(int32_t(send_p_nodes.size()));
    }
    ck(h, nsx_set_mesh_distributed(h, int(layer1.size() + layer2.size()), int(layer1.size()), int(dpc), cell_dofs.data(), cell_coords.data(), int(n_u),
                                   int(n_p), int(world), int(rank), gpu_u_ptr.data(), gpu_p_ptr.data(), int(nbr.size()), nbr.data(), send_u_ptr.data(),
                                   send_u_nodes.data(), send_p_ptr.data(), send_p_nodes.data()));
    (void)mine_p;
  }
};

// ---- the preconditioner concept (for callers that keep deal.II's own SolverGMRES, NavierStokes3D.cpp:554,574) ----------------
// Same initialize / vmult signatures as the reference's classes; the matrices stay on the device, so the Trilinos arguments
// are ignored.  vmult copies the two block vectors through the host: a correctness path, not the fast one (single rank).
template <int dim, int prec_type>
class PreconditionNsxBase {
public:
  void vmult(TrilinosWrappers::MPI::BlockVector &dst, const TrilinosWrappers::MPI::BlockVector &src) const {
    std::vector<double> s(src.size()), d(dst.size());
    for (types::global_dof_index i = 0; i < s.size(); ++i) {
      s[i] = src[i];
      d[i] = dst[i];  // aSIMPLE reads dst as the initial guess of its inner solve (Preconditioners.hpp:271)
    }
    ck(nsx->h, nsx_prec_vmult(nsx->h, prec_type, inner_rtol, inner_maxiter, d.data(), s.data(), nullptr));
    for (types::global_dof_index i = 0; i < d.size(); ++i) dst[i] = d[i];
    dst.compress(VectorOperation::insert);
  }

protected:
  void init(Binding<dim> &b) {
    nsx = &b;
    ck(nsx->h, nsx_prec_initialize(nsx->h, prec_type));
  }
  Binding<dim> *nsx = nullptr;
  double inner_rtol = 1e-2;     // Preconditioners.hpp:156,260,369,483
  int inner_maxiter = 100000;
};

template <int dim>
class PreconditionSIMPLENsx : public PreconditionNsxBase<dim, NSX_PREC_SIMPLE> {  // Preconditioners.hpp:118-217
public:
  explicit PreconditionSIMPLENsx(Binding<dim> &b) : b_(b) { this->inner_maxiter = 10000; }
  void initialize(const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &,
                  const TrilinosWrappers::MPI::BlockVector &) { this->init(b_); }  // :122-126
private:
  Binding<dim> &b_;
};
template <int dim>
class PreconditionaSIMPLENsx : public PreconditionNsxBase<dim, NSX_PREC_ASIMPLE> {  // Preconditioners.hpp:220-329
public:
  explicit PreconditionaSIMPLENsx(Binding<dim> &b) : b_(b) { this->inner_maxiter = 10000; }
  void initialize(const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &,
                  const TrilinosWrappers::MPI::BlockVector &) { this->init(b_); }  // :224-228
private:
  Binding<dim> &b_;
};
template <int dim>
class PreconditionYosidaNsx : public PreconditionNsxBase<dim, NSX_PREC_YOSIDA> {  // Preconditioners.hpp:332-423
public:
  explicit PreconditionYosidaNsx(Binding<dim> &b) : b_(b) {}
  void initialize(const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &,
                  const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::MPI::BlockVector &) { this->init(b_); }  // :336-340
private:
  Binding<dim> &b_;
};
template <int dim>
class PreconditionaYosidaNsx : public PreconditionNsxBase<dim, NSX_PREC_AYOSIDA> {  // Preconditioners.hpp:427-534
public:
  explicit PreconditionaYosidaNsx(Binding<dim> &b) : b_(b) {}
  void initialize(const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::SparseMatrix &,
                  const TrilinosWrappers::SparseMatrix &, const TrilinosWrappers::MPI::BlockVector &) { this->init(b_); }  // :431-435
private:
  Binding<dim> &b_;
};

}  // namespace nsx
#endif
