/* nsx.h — C-ABI of libnsx.so, the MI355X (gfx950) implementation of the per-time-step hot path of
 * lelecaruso/NavierStokes_Project_NM4PDE:
 *
 *     NavierStokes::assemble(time)            reference Navier-Stokes/src/NavierStokes3D.cpp:163-356
 *     NavierStokes::assemble_time_step(time)  reference Navier-Stokes/src/NavierStokes3D.cpp:361-544
 *     NavierStokes::solve_time_step()         reference Navier-Stokes/src/NavierStokes3D.cpp:546-640
 *     Precondition{Yosida,SIMPLE,aYosida,aSIMPLE}::initialize / ::vmult
 *                                              reference Navier-Stokes/include/Preconditioners.hpp:118-534
 * (2D: src/NavierStokes2D.cpp:164-639; convergence study: src/Convergence3D.cpp:187-680.)
 *
 * The reference has no plugin / FFI layer: the seams a replacement binds to are those member functions
 * (SURVEY.md section 8b).  Every entry point below names the member (file:line) whose work it performs;
 * INTEGRATION.md shows the deal.II-side adaptor that forwards the three members to these calls.
 *
 * Conventions: extern "C"; opaque handle, one per GPU / rank; every call returns 0 (NSX_OK) or a negative
 * nsx_status, with a message available from nsx_last_error(); no exception crosses the boundary; host
 * arrays are borrowed for the duration of the call only; device memory is owned by the handle; calls on
 * one handle are not thread-safe.  Indices are int32 (as Epetra's), values are FP64.
 *
 * Numbering contract (what deal.II's DoFHandler provides after DoFRenumbering::component_wise by block,
 * reference NavierStokes3D.cpp:62-69): velocity dofs first, [0, n_u), the `dim` components of one P2 node
 * consecutive (dof = dim * node + c); pressure dofs after them, dof = n_u + p1_node.  Local dof order on a
 * cell is FESystem's: per vertex the dim velocity components then the pressure, then per line the dim
 * velocity components (34 dofs on a tetrahedron, 15 on a triangle).
 */
#ifndef NSX_H
#define NSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsx_handle nsx_handle;

typedef enum {
  NSX_OK = 0,
  NSX_ERR_ARG = -1,          /* bad argument / call order */
  NSX_ERR_HIP = -2,          /* HIP runtime error (message has the HIP error string) */
  NSX_ERR_UNSUPPORTED = -3,  /* valid input this build has no kernel for (e.g. quadrature size) */
  NSX_ERR_NOCONV = -4,       /* a Krylov solver hit its iteration limit: deal.II would throw SolverControl::NoConvergence */
  NSX_ERR_NUMERIC = -5,      /* NaN / zero pivot */
  NSX_ERR_COMM = -6          /* RCCL error */
} nsx_status;

/* preconditioner_type of NavierStokes::solve_time_step (reference NavierStokes3D.cpp:562-634) */
typedef enum { NSX_PREC_YOSIDA = 0, NSX_PREC_SIMPLE = 1, NSX_PREC_AYOSIDA = 2, NSX_PREC_ASIMPLE = 3 } nsx_prec;

/* assembly variants: which terms the reference's three executables put into the convection matrix */
enum {
  NSX_TEMAM = 1,             /* 0.5 (div u_n) (phi_i . phi_j): NS3D first step only (:255); NS2D / Conv every step (NS2D:446, Conv:490) */
  NSX_DOUBLE_CONVECTION = 2  /* Convergence3D.cpp:277 + :284 add the convective term twice in the first assembly */
};

typedef struct {
  int dim;          /* 2 or 3 */
  int device;       /* HIP device ordinal */
  double nu;        /* kinematic viscosity (reference NavierStokes3D.hpp:162) */
  double deltat;    /* time step (reference NavierStokes3D.hpp:191) */
} nsx_params;

typedef struct {
  int outer_iterations;    /* solver_control.last_step()  (reference NavierStokes3D.cpp:636) */
  int inner_F_iterations;  /* sum over all inner GMRES(F) solves of the step */
  int inner_S_iterations;  /* sum over all inner CG / GMRES(S) solves of the step */
  int n_F_solves, n_S_solves;
  double final_residual;   /* last preconditioned residual norm seen by the outer GMRES */
  double t_prec;           /* seconds, preconditioner initialize (reference NavierStokes3D.cpp:558-572, time_prec) */
  double t_solve;          /* seconds, outer solve               (reference NavierStokes3D.cpp:573-577, time_solve) */
  int status;              /* 0 ok, 1 outer GMRES not converged, 2 an inner solve not converged */
  int persistent_fallbacks; /* persistent kernels (Gram-Schmidt sweep, Schur CG) that timed out on this handle SO FAR because their grid was
                            * not co-resident; the handle then stays on the launch-per-operation path.  0 in a healthy run. */
} nsx_solve_stats;

/* ---- life cycle ---- */
int nsx_create(const nsx_params *params, nsx_handle **out);
int nsx_destroy(nsx_handle *h);
const char *nsx_last_error(const nsx_handle *h); /* h may be NULL: error of the last failed nsx_create */
const char *nsx_version(void);

/* ---- setup: the outputs of NavierStokes::setup() (reference NavierStokes3D.cpp:2-157) ---- */

/* FEValues tables on the reference cell, as data (reference NavierStokes3D.cpp:31-50, 172-175):
 * scalar P2 values N2[n_q][n_p2], reference gradients dN2[n_q][n_p2][dim], scalar P1 values N1[n_q][n_p1],
 * quadrature weights w[n_q] (sum = volume of the reference simplex).  n_p2 = 6/10, n_p1 = 3/4. */
int nsx_set_tables(nsx_handle *h, int n_q, int n_p2, int n_p1, const double *N2, const double *dN2,
                   const double *N1, const double *weights);

/* Mesh + DoF tables: cell->get_dof_indices for every locally owned cell (reference NavierStokes3D.cpp:304,494)
 * and the cell's vertex coordinates (affine map).  Builds the sparsity graphs of the three blocks
 * (reference NavierStokes3D.cpp:109-124) internally.  cell_dofs[n_cells][dofs_per_cell], cell_coords[n_cells][dim+1][dim]. */
int nsx_set_mesh(nsx_handle *h, int n_cells, int dofs_per_cell, const int32_t *cell_dofs, const double *cell_coords,
                 int n_u, int n_p);

/* MPI-rank structure of the run being reproduced: velocity P2-node ranges u_ptr[n_ranks+1] and P1-node
 * ranges p_ptr[n_ranks+1] owned by each rank.  They define (a) the blocks of the per-rank Ifpack ILU(0)
 * (TrilinosWrappers::PreconditionILU, overlap 0: reference Preconditioners.hpp:215-216 and SURVEY D4) and
 * (b) the per-rank diagonal scan of MatrixTools::apply_boundary_values.  Default: one rank.
 * On one GPU this lets the ILU run as n_ranks independent triangular solves, exactly what `mpirun -n n_ranks`
 * of the reference computes. */
int nsx_set_ranks(nsx_handle *h, int n_ranks, const int32_t *u_ptr, const int32_t *p_ptr);
/* Optional: coarser blocks for the Schur-complement ILU (unions of consecutive ranks); default = the ranks. */
int nsx_set_schur_blocks(nsx_handle *h, int n_blocks, const int32_t *p_ptr);

/* The numbering libnsx works in BEHIND this boundary.  The reference numbers its DoFs with distribute_dofs +
 * DoFRenumbering::component_wise on a METIS partition into mpi_size parts (reference NavierStokes3D.cpp:16-19,58-69) and a caller
 * hands exactly that over (nsx_set_mesh, nsx_set_ranks).  The triangular solves of the per-rank ILU(0) want thousands of small,
 * spatially compact rank blocks with a shallow dependency graph inside each; with this call libnsx builds them itself:
 * inside the node range of EVERY rank of the caller the cells are bisected (coordinates of their centroids) into virtual ranks --
 * n_virtual_ranks over the whole handle, dealt to the caller's ranks in proportion to their nodes --, a node belongs to the lowest
 * virtual rank touching it (deal.II's rule), nodes are numbered virtual rank by virtual rank in first-touch order (cell by cell,
 * vertices then lines) and, for NSX_ORDER_COLOUR, sorted by a greedy colouring of the rank's P2 graph (NSX_ORDER_COLOUR_ALL: the
 * pressure nodes as well, on the graph of the Schur complement).  schur_max_rows > 0 merges consecutive virtual ranks into Schur
 * ILU blocks of at most that many pressure rows (on one handle a block may span two ranks of the caller; nothing spans handles);
 * 0: one block per virtual rank.
 * What the library then computes is what the reference computes on n_virtual_ranks MPI ranks with that numbering (per-rank ILU(0),
 * per-rank diagonal of apply_boundary_values).  NOTHING changes at the boundary: every vector, dof list, graph and value array that
 * crosses it stays in the caller's numbering (permuted on the device on the way in and out).
 * Call order: any time before nsx_assemble -- before nsx_set_mesh (one set-up pass) or after it and after nsx_set_ranks (the set-up
 * products are rebuilt; state vectors are reset).  nsx_set_ranks afterwards lays the nodes out again inside the new ranges;
 * nsx_set_schur_blocks is refused while a layout is in force.  n_virtual_ranks = 0 switches the layout off. */
enum { NSX_ORDER_FIRST_TOUCH = 0, NSX_ORDER_COLOUR = 1, NSX_ORDER_COLOUR_ALL = 2 };
int nsx_set_internal_layout(nsx_handle *h, int n_virtual_ranks, int order, int schur_max_rows);
/* info = {layout in force (0/1), ranks the ILU(0) of system(0,0) runs on, Schur ILU blocks, colours of the P2 nodes, colours of the P1 nodes} */
int nsx_layout_info(nsx_handle *h, int info[5]);
/* The layout itself, for tests and tools (any pointer may be NULL): node_perm[i] / pnode_perm[i] = internal (global) number of the
 * i-th P2 / P1 node this handle owns, u_ptr / p_ptr [info[1] + 1] the internal node ranges of the ranks, schur_ptr [info[2] + 1]. */
int nsx_layout_get(nsx_handle *h, int32_t *node_perm, int32_t *pnode_perm, int32_t *u_ptr, int32_t *p_ptr, int32_t *schur_ptr);

/* ---- state ---- */
/* `solution` (ghosted) and `solution_owned` (reference NavierStokes3D.hpp:245-248), length n_u + n_p. */
int nsx_set_solution(nsx_handle *h, const double *solution_owned);  /* also does solution = solution_owned (NavierStokes3D.cpp:696-697) */
int nsx_get_solution(nsx_handle *h, double *solution_owned);
int nsx_get_solution_ghosted(nsx_handle *h, double *solution);
int nsx_get_rhs(nsx_handle *h, double *system_rhs);
int nsx_set_rhs(nsx_handle *h, const double *system_rhs);

/* ---- the hot path ---- */

/* NavierStokes::assemble(time) without its Dirichlet block (reference NavierStokes3D.cpp:163-324):
 * mass/deltat, nu*stiffness, convection(u_n) (+Temam), -B^T / B, pressure mass; system = blocks + M + C + K; rhs.
 * As the run's set-up step it also builds the ILU schedules for the current rank / Schur block tables (host work, once). */
int nsx_assemble(nsx_handle *h, int flags);
/* NavierStokes::assemble_time_step(time) without its Dirichlet block (reference NavierStokes3D.cpp:361-512):
 * new convection(u_n) and rhs; system = system - C_old + C_new. */
int nsx_assemble_time_step(nsx_handle *h, int flags);
/* system_rhs.add(dof_indices, cell_rhs) for terms integrated by the caller (Neumann face term, Convergence3D.cpp:309-331). */
int nsx_add_rhs(nsx_handle *h, int n, const int32_t *dofs, const double *values);
/* MatrixTools::apply_boundary_values(boundary_values, system_matrix, solution, system_rhs, false)
 * (reference NavierStokes3D.cpp:353,541).  (dofs[k], values[k]) is the std::map, sorted by dof; every component
 * of a constrained P2 node must be present (the reference's ComponentMask always selects all of them). */
int nsx_apply_boundary_values(nsx_handle *h, int n, const int32_t *dofs, const double *values);
/* NavierStokes::solve_time_step() (reference NavierStokes3D.cpp:546-640): previous_solution = solution;
 * preconditioner.initialize(...); SolverGMRES(solver_control(maxiter, tol_abs)).solve(system_matrix,
 * solution_owned, system_rhs, preconditioner); solution = solution_owned.
 * Reference values: tol_abs = 1e-4, inner_rtol = 1e-2, maxiter = 100000, inner_maxiter = 100000 (10000 for (a)SIMPLE). */
int nsx_solve_time_step(nsx_handle *h, int prec_type, double tol_abs, double inner_rtol, int maxiter,
                        int inner_maxiter, nsx_solve_stats *stats);

/* Pieces of solve_time_step, exposed for parity tests and for callers that drive deal.II's own SolverGMRES
 * through the preconditioner concept (initialize / vmult, reference Preconditioners.hpp:122-126,152-153). */
int nsx_prec_initialize(nsx_handle *h, int prec_type);
int nsx_prec_vmult(nsx_handle *h, int prec_type, double inner_rtol, int inner_maxiter, double *dst, const double *src,
                   nsx_solve_stats *stats);
int nsx_system_vmult(nsx_handle *h, double *dst, const double *src); /* BlockSparseMatrix::vmult, host vectors n_u+n_p */
/* One PreconditionILU::vmult with the factors of the last initialize: which = 0 (F, length n_u) or 1 (S, length n_p). */
int nsx_ilu_apply(nsx_handle *h, int which, double *dst, const double *src);

/* ---- export in the reference's own layout (Trilinos block CSR with all velocity couplings stored) ---- */
/* which: 0 system_matrix, 1 mass_matrix, 2 convection_matrix, 3 stiffness_matrix, 4 pressure_mass
 * block: 0=(0,0) n_u x n_u, 1=(0,1) n_u x n_p, 2=(1,0) n_p x n_u, 3=(1,1) (pressure_mass only).
 * The caller passes the CSR graph it wants filled (e.g. Epetra's ExtractCrsDataPointers); entries that are
 * structural zeros in the reference (cross-component couplings) are written as 0. */
int nsx_export_block(nsx_handle *h, int which, int block, int n_rows, const int32_t *rowptr, const int32_t *colind,
                     double *values);
/* negative_S_tilde of the last initialize (reference Preconditioners.hpp:144,248,358,468): query nnz, then fetch. */
int nsx_schur_nnz(nsx_handle *h, int64_t *nnz);
int nsx_schur_get(nsx_handle *h, int32_t *rowptr, int32_t *colind, double *values);
/* ILU(0) factors of the last initialize in the compact layout of the scalar velocity graph / the Schur graph
 * (strict lower = L, diagonal = 1/d, strict upper = U/d as Ifpack stores them); graph via nsx_scalar_graph. */
int nsx_scalar_graph_nnz(nsx_handle *h, int which, int64_t *nnz); /* which: 0 velocity scalar P2 graph, 1 Schur graph */
int nsx_scalar_graph(nsx_handle *h, int which, int32_t *rowptr, int32_t *colind);
int nsx_ilu_get(nsx_handle *h, int which, double *values);

/* ---- forces on the obstacle ---- */
/* NavierStokes::compute_forces (reference NavierStokes3D.cpp:744-846, NavierStokes2D.cpp:752-859): drag and lift by
 * face quadrature over the faces with boundary id 3.  cells[f] = position of the face's cell in the cell list given to
 * nsx_set_mesh(_distributed), local_faces[f] = deal.II face number inside that cell.  Face tables: for every face of the
 * reference cell the shape values / reference gradients at its n_qf quadrature points, N2f[(face*n_qf+q)][n_p2] etc.,
 * weights wf[n_qf] summing to 1 (QGaussSimplex<dim-1>(3) in 3D, QGauss<1>(3) in 2D).  Returns the raw forces; the
 * coefficients 2F/(rho U^2 D H) (3D) and 2F/(U^2 D) (2D) are host arithmetic (NavierStokes3D.cpp:835-842).
 * In a multi-process run each rank passes the faces of the cells it owns and the sum is all-reduced. */
int nsx_set_force_faces(nsx_handle *h, int n_faces, const int32_t *cells, const int32_t *local_faces, int n_qf, const double *N2f,
                        const double *dN2f, const double *N1f, const double *wf);
int nsx_compute_forces(nsx_handle *h, double *drag, double *lift);

/* ---- measurement ---- */
/* Per-kernel HIP-event timing of the hot path (bench.py roofline): enable, run, then read name/count/total-ms. */
int nsx_profile_enable(nsx_handle *h, int on);
int nsx_profile_reset(nsx_handle *h);
int nsx_profile_count(nsx_handle *h);
int nsx_profile_get(nsx_handle *h, int i, const char **name, int64_t *launches, double *total_ms, double *bytes_per_launch);

/* State of the two persistent (single-launch, grid-wide-exchange) kernels of this handle, for tests and bench.py:
 * state[0] / state[1] = 1 while the Gram-Schmidt sweep / the Schur-complement CG run as ONE launch (0: never used yet, switched
 * off, or fallen back), state[2] = time-outs so far (= nsx_solve_stats::persistent_fallbacks), state[3] = mailbox words that are
 * not empty in the region the next launch would use (0 on a healthy handle and after a recovered time-out). */
int nsx_persistent_state(nsx_handle *h, int state[4]);

/* Which code paths this handle's products and solves take -- for tests, and for the log a multi-GPU rehearsal writes per rank:
 * info[0] 1 = every F->vmult (reference Preconditioners.hpp:382,405; NavierStokes3D.cpp:574) goes through the LDS-staged SpMV,
 * [1] its chunks, [2] those of them that stage a ghost column (launched behind the ghost exchange; 0 on one GPU),
 * [3] entries per thread of the LAST Gram-Schmidt sweep that ran as one persistent launch (8 / 10 / 12; 0: none yet, or two passes),
 * [4] workgroups of its grid, [5] 1 = with the collective inside, [6] the largest instantiation any sweep has used so far,
 * [7] CUs the compute stream leaves to the communication stream (0 or 8), [8] the last Schur-complement CG: 1 one launch per
 * operation, 2 one persistent launch, 3 two launches per iteration, [9] Schur ILU blocks, [10] neighbours of the ghost exchange,
 * [11] owned P2 nodes sent per exchange, [12] ghost P2 nodes, [13] 1 = explicit inverses of the Schur ILU blocks,
 * [14] 1 = the persistent sweep is switched off or has fallen back, [15] persistent kernels that timed out so far,
 * [16] / [17] entries per thread of the sweep instantiation an RCCL run WOULD use for the velocity vector on plain / on CU-masked
 * streams (0: the resident grid does not hold it: two passes), [18] / [19] the same for the block vector, [20] Schur blocks per
 * entry of a partial-sum array of the two-launch CG (1: no fold launch), [21] the velocity sweep's instantiation on one GPU without a
 * communicator, [22] / [23] P2 / P1 nodes this handle owns, [24] 1 = the last persistent sweep had the triangular solves of the velocity
 * ILU(0) inside its launch (k_ilu_mgs: PreconditionILU::vmult + the orthogonalisation of one inner GMRES iteration, reference
 * Preconditioners.hpp:382,405, as ONE kernel), [25] such launches so far; [26..31] reserved (0). */
int nsx_path_info(nsx_handle *h, int info[32]);

/* SolverGMRES' orthogonalisation (deal.II's modified Gram-Schmidt add_and_dot chain inside every solver.solve of the path: reference
 * NavierStokes3D.cpp:574, Preconditioners.hpp:173,273,288,382,405) on the caller's vectors, through the very sweep kernel the solvers
 * use: vectors[m][n] (row k = vector k) -- vector 0 is normalised, vector k is orthogonalised against vectors 0..k-1 and normalised,
 * in place.  coeffs[k * m + i] = h(i) of sweep k, norms2[k] = |w|^2 after sweep k (norms2[0]: |vector 0|^2 as it came).
 * norm_guard: the sweep takes |w'|^2 from the basis' Gram matrix (|w|^2 - 2 h.r + h^T G h) while more than that fraction of |w|^2
 * is left and sums it explicitly otherwise; 0 = always the formula, 1e300 = always the explicit sum, < 0 = the library's own (1e-2).
 * A test hook for that formula; m <= 30. */
int nsx_gram_schmidt_cycle(nsx_handle *h, int n, int m, double *vectors, double norm_guard, double *coeffs, double *norms2);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) ---- */
/* Replaces the MPI communicator inside Epetra (reference NavierStokes3D.hpp:93-94,102): MPI_Allreduce behind every
 * dot / norm, Epetra_Import behind every vmult and behind `solution = solution_owned` (NavierStokes3D.cpp:638). */
int nsx_comm_unique_id(uint8_t id[128]);                       /* rank 0: ncclGetUniqueId */
int nsx_comm_init(nsx_handle *h, int rank, int world, const uint8_t id[128]);   /* RCCL over xGMI */
/* Bring-your-own communicator (the reference's MPI, gloo in the tests): host-buffer callbacks, return 0 on success.
 * allreduce: in-place sum of `count` doubles over all ranks.  exchange: for each of n neighbours send send[k]
 * (send_count[k] doubles) to rank ranks[k] and receive recv_count[k] doubles from it into recv[k]. */
typedef int (*nsx_allreduce_fn)(void *ctx, double *buf, int count);
typedef int (*nsx_exchange_fn)(void *ctx, int n, const int *ranks, const double *const *send, const int *send_count,
                               double *const *recv, const int *recv_count);
int nsx_comm_init_callbacks(nsx_handle *h, int rank, int world, nsx_allreduce_fn allreduce, nsx_exchange_fn exchange, void *ctx);
/* Collectives issued by this handle since the communicator was set: counts[0] all-reduces (the MPI_Allreduce behind Epetra's
 * Dot / Norm2, reference Preconditioners.hpp:157,179,371,388,403 and every SolverGMRES / SolverCG iteration), counts[1] ghost
 * exchanges (the Epetra_Import of every vmult).  The orthogonalisation of a Krylov vector costs ONE all-reduce (csrc/nsx_blas.hip,
 * mgs_lowsync; a second one only when the sweep removes more than 99 % of the vector's norm) where the reference pays one per link
 * of the add_and_dot chain. */
int nsx_comm_counters(const nsx_handle *h, long long counts[2]);
/* Test hook: the RCCL branch of the ghost exchange (pack kernel, grouped ncclSend / ncclRecv straight into the ghost region -- the
 * Epetra_Import of every vmult --, event, wait of the compute stream) on a 1-rank communicator whose only neighbour is the rank
 * itself: ghost node k of a vector of n_own + n_ghost nodes must receive the ncomp values of owned node (7 k + 3) % n_own.
 * max_err = largest deviation after three exchanges in a row.  (RCCL refuses two ranks on one device: a one-GPU box has no other way
 * to execute that branch.) */
int nsx_comm_self_halo_test(nsx_handle *h, int n_own, int n_ghost, int ncomp, double *max_err);
/* Distributed mesh, replaces nsx_set_mesh for world > 1.  cell_dofs keep the GLOBAL deal.II numbering; gpu_u_ptr /
 * gpu_p_ptr [world+1] are the P2 / P1 node ranges owned by each rank (locally_owned_dofs per block, reference
 * NavierStokes3D.cpp:71-87).  Cells: first the n_cells_layer1 cells that touch an owned P2 node (every owned row is
 * assembled locally, no compress(VectorOperation::add), NavierStokes3D.cpp:314-319), then the cells touching a node
 * of those (rows of block(1,0) for the ghost pressure nodes the Schur product reaches).  Halo plan: for each
 * neighbour (ascending ranks) the global ids of the owned nodes it needs, ascending (include/nsx_host.h:
 * nsxh_rank_view_*).  Afterwards nsx_set_ranks takes GLOBAL node ranges of this rank's sub-blocks; state vectors
 * (nsx_set_solution, nsx_get_*, nsx_apply_boundary_values, nsx_system_vmult) are indexed GLOBALLY, get_* fill only the
 * entries this rank owns. */
int nsx_set_mesh_distributed(nsx_handle *h, int n_cells, int n_cells_layer1, int dofs_per_cell, const int32_t *cell_dofs,
                             const double *cell_coords, int n_u_global, int n_p_global, int world, int rank,
                             const int32_t *gpu_u_ptr, const int32_t *gpu_p_ptr, int n_neighbors, const int32_t *neighbors,
                             const int32_t *send_u_ptr, const int32_t *send_u_nodes, const int32_t *send_p_ptr,
                             const int32_t *send_p_nodes);

#ifdef __cplusplus
}
#endif
#endif
