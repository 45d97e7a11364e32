/* nsx_host.h — C API of the host-side front-end (mesh, DoF handler, FE tables).
 *
 * This is the CPU "setup" stage that the reference performs with deal.II in
 * NavierStokes::setup() (reference Navier-Stokes/src/NavierStokes3D.cpp:2-157):
 * mesh creation + partition (:5-23), FE/quadrature (:27-54), DoF distribution +
 * block renumbering (:58-93) and sparsity (:97-156).  Its outputs are exactly
 * the inputs of the device library (include/nsx.h).  No GPU code here.
 *
 * All arrays returned by nsxh_* accessors are owned by the object they were
 * obtained from and stay valid until that object is freed.
 */
#ifndef NSX_HOST_H
#define NSX_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsxh_mesh nsxh_mesh;
typedef struct nsxh_dofs nsxh_dofs;
typedef struct nsxh_tables nsxh_tables;

/* ---- meshes (reference mesh/ .geo geometries; gmsh is replaced by a block-structured generator) ---- */

/* Channel with a circular obstacle.
 * dim=2: reference mesh/Cylinder2D.geo:6-22 (2.2 x 0.41, circle c=(0.2,0.2) r=0.05);
 * dim=3: reference mesh/Cylinder3D.geo:9-15 (2.5 x 0.41 x 0.41, axis z through (0.5,0.2), R=0.05).
 * Boundary ids as in the .geo files: 0 inlet (x=0), 1 outlet (x=L), 2 walls, 3 obstacle.
 * `m`  = cells per side of the square that carries the O-grid around the cylinder (even, >=2),
 * `nr` = radial layers of the O-grid, `nxu`/`nxd` = cells upstream/downstream of the square,
 * `nyb`/`nyt` = cells below/above the square, `nz` = layers in z (3D only).
 * `grade_x` >= 1 stretches the downstream cells geometrically, `grade_r` >= 1 the radial layers. */
nsxh_mesh *nsxh_mesh_cylinder(int dim, int m, int nr, int nxu, int nxd, int nyb, int nyt, int nz,
                              double grade_x, double grade_r);
/* Preset: one integer "level" scales all counts (level 1 ~ coarse). */
nsxh_mesh *nsxh_mesh_cylinder_level(int dim, int level);
/* Cube [-1,1]^3 with n cells per side x 6 tets (reference mesh/mesh-cube.geo:1-28).
 * Boundary ids 0..5: x=-1:0, x=+1:1, y=+1:2, y=-1:3 (Neumann side, reference Convergence3D.cpp:308), z=-1:4, z=+1:5. */
nsxh_mesh *nsxh_mesh_cube(int n);
/* Generic box [x0,x1]x[y0,y1](x[z0,z1]) split into simplices; boundary ids as for the channel (no obstacle). */
nsxh_mesh *nsxh_mesh_box(int dim, int nx, int ny, int nz, const double *lo, const double *hi);
/* Gmsh MSH 2.2 / 4.1 ASCII reader for triangle / tetrahedron meshes with physical ids (reference NavierStokes3D.cpp:10-14). */
nsxh_mesh *nsxh_mesh_read_msh(const char *path);
void nsxh_mesh_free(nsxh_mesh *);

int nsxh_mesh_dim(const nsxh_mesh *);
int nsxh_mesh_n_vertices(const nsxh_mesh *);
int nsxh_mesh_n_cells(const nsxh_mesh *);
int nsxh_mesh_n_bfaces(const nsxh_mesh *);
const double *nsxh_mesh_vertices(const nsxh_mesh *);   /* [n_vertices][dim] */
const int32_t *nsxh_mesh_cells(const nsxh_mesh *);     /* [n_cells][dim+1], positively oriented */
const int32_t *nsxh_mesh_bfaces(const nsxh_mesh *);    /* [n_bfaces][dim] vertex ids */
const int32_t *nsxh_mesh_bface_ids(const nsxh_mesh *); /* [n_bfaces] boundary id */
const int32_t *nsxh_mesh_bface_cells(const nsxh_mesh *); /* [n_bfaces] adjacent cell */
const int32_t *nsxh_mesh_subdomain(const nsxh_mesh *); /* [n_cells] */

/* Partition cells into n_parts * n_sub subdomains by recursive coordinate bisection of the
 * cell centroids (stands in for GridTools::partition_triangulation / METIS, reference
 * NavierStokes3D.cpp:16): first n_parts "ranks" (GPUs), then n_sub blocks inside each.
 * subdomain id = part * n_sub + sub. */
int nsxh_mesh_partition(nsxh_mesh *, int n_parts, int n_sub);
/* The same bisection with the cuts placed so that the subdomains OWN equal numbers of vertices under deal.II's rule
 * "an interface node belongs to the lowest subdomain id touching it" (equal cell counts give the low ids up to twice the
 * mean).  Equal ILU(0) blocks are what the wave-per-block triangular solve wants; the cell counts then differ. */
int nsxh_mesh_partition_owned(nsxh_mesh *, int n_parts, int n_sub);

/* ---- DoF handler: Taylor-Hood P2/P1 (FESystem(FE_SimplexP(2)^dim, FE_SimplexP(1)), reference NavierStokes3D.cpp:31-36) ---- */

/* distribute_dofs + DoFRenumbering::component_wise(block {0,..,0,1}) (reference :62-69), numbered
 * subdomain by subdomain exactly as an MPI run with one rank per subdomain would be. */
nsxh_dofs *nsxh_distribute_dofs(const nsxh_mesh *);
/* Same, with a choice of the node order INSIDE each subdomain (deal.II leaves it to the cell traversal, the reference
 * applies no renumbering of its own):
 *   NSXH_ORDER_FIRST_TOUCH  cell by cell, vertices then lines (what nsxh_distribute_dofs does);
 *   NSXH_ORDER_COLOUR       velocity nodes sorted by a greedy colouring of the subdomain's P2 graph: the per-rank ILU(0)
 *                           gets a dependency graph as shallow as the number of colours.
 *   NSXH_ORDER_COLOUR_ALL   the same, and the pressure nodes sorted by a greedy colouring of the Schur complement's graph
 *                           (B D^-1 B^T): for few large ranks (mpirun -n 1, one rank per GPU), where the ILU(0) of the Schur
 *                           matrix is applied as sparse sweeps and its depth is what bounds them.
 * Ownership and rank ranges are the same for all; the pressure numbering is the same for the first two. */
enum { NSXH_ORDER_FIRST_TOUCH = 0, NSXH_ORDER_COLOUR = 1, NSXH_ORDER_COLOUR_ALL = 2 };
nsxh_dofs *nsxh_distribute_dofs_ordered(const nsxh_mesh *, int ordering);
int nsxh_n_colours(const nsxh_dofs *);       /* colours of the velocity nodes (NSXH_ORDER_COLOUR, _ALL), 0 otherwise */
int nsxh_n_colours_p(const nsxh_dofs *);     /* colours of the pressure nodes (NSXH_ORDER_COLOUR_ALL), 0 otherwise */
void nsxh_dofs_free(nsxh_dofs *);

int nsxh_dofs_per_cell(const nsxh_dofs *);   /* 15 (2D) / 34 (3D) */
int nsxh_n_nodes_p2(const nsxh_dofs *);      /* scalar P2 nodes; n_u = dim * this */
int nsxh_n_nodes_p1(const nsxh_dofs *);      /* = n_p */
int nsxh_n_u(const nsxh_dofs *);
int nsxh_n_p(const nsxh_dofs *);
const int32_t *nsxh_cell_dofs(const nsxh_dofs *);      /* [n_cells][dofs_per_cell] global dof indices, FESystem local order */
const double *nsxh_cell_coords(const nsxh_dofs *);     /* [n_cells][dim+1][dim] */
const double *nsxh_support_points(const nsxh_dofs *);  /* [n_u+n_p][dim] support point of every dof */
const int32_t *nsxh_node_owner(const nsxh_dofs *);     /* [n_nodes_p2] owning subdomain */
const int32_t *nsxh_pnode_owner(const nsxh_dofs *);    /* [n_nodes_p1] owning subdomain */
const int32_t *nsxh_owned_u_ptr(const nsxh_dofs *);    /* [n_subdomains+1] P2-node ranges owned by each subdomain */
const int32_t *nsxh_owned_p_ptr(const nsxh_dofs *);    /* [n_subdomains+1] P1-node ranges */
int nsxh_n_subdomains(const nsxh_dofs *);

/* Boundary dofs of one boundary id (what VectorTools::interpolate_boundary_values visits,
 * reference NavierStokes3D.cpp:334-351), velocity components only, sorted ascending.
 * Returns the count; *dofs is owned by the dof object. */
int nsxh_boundary_dofs(nsxh_dofs *, int boundary_id, const int32_t **dofs);

/* Block sparsity of the reference (DoFTools::make_sparsity_pattern with all couplings but (p,p),
 * reference NavierStokes3D.cpp:109-124): block 0=(0,0) n_u x n_u, 1=(0,1) n_u x n_p, 2=(1,0) n_p x n_u,
 * 3 = pressure-mass (1,1) n_p x n_p (:127-142). CSR with sorted columns. */
int nsxh_reference_sparsity(nsxh_dofs *, int block, const int32_t **rowptr, const int32_t **colind);

/* ---- per-rank view for multi-GPU runs (one process per GPU; GPU r = subdomains [r*n_sub, (r+1)*n_sub)) ----
 * Cells: layer 1 = every cell touching an owned P2 node (all owned matrix rows assemble without exchange, replacing
 * compress(VectorOperation::add), reference NavierStokes3D.cpp:314-319,506-511), then layer 2 = cells touching a node
 * of layer 1 (completes the block(1,0) rows the Schur product of an owned row reaches).  cell_dofs keep GLOBAL indices.
 * Halo plan (the Epetra_Import of every vmult): for each neighbour the owned P2 / P1 nodes it needs, sorted by global
 * id; it is computed from the replicated serial mesh like the reference's partition (NavierStokes3D.cpp:8-19). */
typedef struct nsxh_rank_view nsxh_rank_view;
nsxh_rank_view *nsxh_rank_view_create(const nsxh_dofs *, int rank, int world);
void nsxh_rank_view_free(nsxh_rank_view *);
int nsxh_rank_view_n_cells(const nsxh_rank_view *);
int nsxh_rank_view_n_cells_layer1(const nsxh_rank_view *);
const int32_t *nsxh_rank_view_cell_ids(const nsxh_rank_view *);      /* [n_cells] global cell ids */
const int32_t *nsxh_rank_view_cell_dofs(const nsxh_rank_view *);     /* [n_cells][dofs_per_cell] global dofs */
const double *nsxh_rank_view_cell_coords(const nsxh_rank_view *);
const int32_t *nsxh_rank_view_gpu_u_ptr(const nsxh_rank_view *);     /* [world+1] P2-node ranges of the GPUs */
const int32_t *nsxh_rank_view_gpu_p_ptr(const nsxh_rank_view *);
int nsxh_rank_view_n_virtual_ranks(const nsxh_rank_view *);          /* n_sub */
const int32_t *nsxh_rank_view_rank_u_ptr(const nsxh_rank_view *);    /* [n_sub+1] global P2-node ranges of this GPU's subdomains */
const int32_t *nsxh_rank_view_rank_p_ptr(const nsxh_rank_view *);
int nsxh_rank_view_n_neighbors(const nsxh_rank_view *);
const int32_t *nsxh_rank_view_neighbors(const nsxh_rank_view *);     /* [n_neighbors] ascending ranks */
const int32_t *nsxh_rank_view_send_u_ptr(const nsxh_rank_view *);    /* [n_neighbors+1] */
const int32_t *nsxh_rank_view_send_u_nodes(const nsxh_rank_view *);  /* global P2 node ids */
const int32_t *nsxh_rank_view_send_p_ptr(const nsxh_rank_view *);
const int32_t *nsxh_rank_view_send_p_nodes(const nsxh_rank_view *);

/* ---- reference-element tables (what FEValues evaluates once per run) ---- */

/* rule: 0 = cell rule QGaussSimplex<dim>(3) stand-in (degree 5: 7 pts in 2D, 14 pts in 3D; see DESIGN.md note Q),
 *       1 = face rule QGaussSimplex<dim-1>(3) stand-in mapped onto every face of the reference cell,
 *       2 = high-order conical rule (error norms; reference Convergence3D.cpp:772). */
nsxh_tables *nsxh_tables_create(int dim, int rule, int order);
void nsxh_tables_free(nsxh_tables *);
int nsxh_tables_n_q(const nsxh_tables *);
int nsxh_tables_n_p2(const nsxh_tables *);  /* 6 / 10 */
int nsxh_tables_n_p1(const nsxh_tables *);  /* 3 / 4  */
const double *nsxh_tables_points(const nsxh_tables *);   /* [n_q][dim] */
const double *nsxh_tables_weights(const nsxh_tables *);  /* [n_q] (sum = reference cell volume) */
const double *nsxh_tables_N2(const nsxh_tables *);       /* [n_q][n_p2] */
const double *nsxh_tables_dN2(const nsxh_tables *);      /* [n_q][n_p2][dim] reference gradients */
const double *nsxh_tables_N1(const nsxh_tables *);       /* [n_q][n_p1] */
const double *nsxh_tables_dN1(const nsxh_tables *);      /* [n_q][n_p1][dim] */
/* face rule only: n_q = n_faces * n_qf; face f uses points [f*n_qf, (f+1)*n_qf) */
int nsxh_tables_n_qf(const nsxh_tables *);

/* ---- post-processing (pure host I/O, the reference's output side) ---- */

/* NavierStokes::output (reference NavierStokes3D.cpp:643-683, NavierStokes2D.cpp:642-677): DataOut with the vector
 * "velocity", the scalar "pressure" and the cell-wise "partitioning", build_patches() with one subdivision (one linear
 * patch per cell with its own vertices), write_vtu_with_pvtu_record(directory, basename, counter, comm, -, 1).
 * Writes <directory>/<basename>_<counter>.0.vtu and <directory>/<basename>_<counter>.pvtu (ASCII data arrays; the
 * directory is created if missing).  solution = ghosted solution in the global dof numbering.  0 on success. */
int nsxh_write_vtu(const nsxh_dofs *, const double *solution, const char *directory, const char *basename, unsigned counter);

/* NavierStokes::compute_pressure_difference (reference NavierStokes3D.cpp:849-923): P1 pressure at two points
 * (VectorTools::point_value): p(a) - p(b).  A point outside the mesh contributes 0 like a rank that does not hold it.
 * Returns the number of points found (0..2). */
int nsxh_pressure_difference(const nsxh_dofs *, const double *solution, const double *point_a, const double *point_b, double *diff);

/* ---- test hooks: the schedule of the packed block-ILU(0) triangular solve (navierstokes_project_nm4pde_amd/host/ilu_stream.hpp,
 * the code the device library runs at set-up; device kernel k_ilu_solve_lanes) ---- */

/* Build the slab stream for a square CSR graph (sorted columns) and a block table: `blocks_per_wave` blocks per wave on average,
 * `ncomp` interleaved right-hand sides, a row readable `gap` ticks after its last tick (the kernel needs 2), `entries_per_tick`
 * (1..4) entries of its row per lane and tick.
 * out = {slabs, most slabs of one wave, in-block entries incl. the diagonal, most LDS rows of one wave, waves, slots in use}.
 * 0; -3 when a wave's rows do not fit 16-bit LDS addresses; -1 on bad input. */
int nsxh_ilu_stream_stats(int n_rows, const int32_t *rowptr, const int32_t *colind, int n_blocks, const int32_t *block_ptr,
                          int blocks_per_wave, int ncomp, int gap, int entries_per_tick, int64_t out[6]);
/* Replay the stream on the host tick by tick exactly as the kernel consumes it (the LDS reads of a tick before the writes of the
 * tick in front of it): x = U^-1 D^-1 L^-1 b per block for factors `lu` in Ifpack's storage on the graph. */
int nsxh_ilu_stream_apply(int n_rows, const int32_t *rowptr, const int32_t *colind, int n_blocks, const int32_t *block_ptr,
                          int blocks_per_wave, int ncomp, int gap, int entries_per_tick, const double *lu, const double *b, double *x);

/* ---- test hook: the internal layout of the device library (navierstokes_project_nm4pde_amd/host/layout.hpp, the code
 * nsx_set_internal_layout runs) for a serial DoF table: cell_dofs / cell_coords as for nsx_set_mesh, the caller's ranks as for
 * nsx_set_ranks.  Outputs: node_perm[n_u/dim], pnode_perm[n_p] (caller node -> internal node), the virtual ranks' node ranges
 * u_ptr / p_ptr [*n_ranks + 1] and the Schur ILU blocks schur_ptr [*n_schur + 1] (capacity n_virtual + n_in_ranks + 1 each),
 * colours[2] (P2, P1).  0 on success. */
int nsxh_internal_layout(int dim, int n_cells, int dofs_per_cell, const int32_t *cell_dofs, const double *cell_coords, int n_u, int n_p,
                         int n_in_ranks, const int32_t *in_u_ptr, const int32_t *in_p_ptr, int n_virtual, int order, int schur_max_rows,
                         int32_t *node_perm, int32_t *pnode_perm, int32_t *n_ranks, int32_t *u_ptr, int32_t *p_ptr, int32_t *n_schur,
                         int32_t *schur_ptr, int32_t *colours);

#ifdef __cplusplus
}
#endif
#endif
