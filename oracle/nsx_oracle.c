/* nsx_oracle.c — CPU restatement of the reference's per-time-step hot path (see nsx_oracle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into or called by the product.  PARITY UNPINNED (header).
 *
 * Reference short names: NS3D, NS2D, Conv, Prec (see header).  deal.II / Trilinos library behaviour that the
 * reference relies on is restated from the published algorithms of deal.II 9.3-9.5 (SolverGMRES with modified
 * Gram-Schmidt + Kelley re-orthogonalisation, SolverCG, MatrixTools::apply_boundary_values for Trilinos block
 * matrices) and Ifpack (Ifpack_ILU level 0, overlap 0); those restatements are marked [lib].
 */
/* Threads: the same file builds twice.  liboracle.so (no -fopenmp: every `#pragma omp` below is ignored) is the serial,
 * reference-shaped restatement the parity tests check against.  liboracle_mt.so (-fopenmp) is the "all host cores"
 * CPU baseline of bench.py: the loops the reference distributes over MPI ranks run on OpenMP threads instead
 * (cells of the assembly loop, rows of every vmult / mmult, the per-rank ILU(0) blocks, Epetra's BLAS-1 with its
 * MPI_Allreduce as an OpenMP reduction).  Same algorithm and data layout; only the summation order of the
 * reductions and of the concurrent `add`s differs (as it does between two MPI runs). */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include "nsx_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#include <malloc.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define N_TMP 30 /* SolverGMRES::AdditionalData::max_n_tmp_vectors default [lib] */

typedef struct {
  int n_rows, n_cols;
  int32_t *rp, *ci;
  double *v;
} csr_t;

struct orc {
  int dim, n_cells, dpc, n_u, n_p, n_q, np2, np1;
  int32_t *cell_dofs;
  double *cell_coords, *N2, *dN2, *N1, *w;
  int32_t *rp[4], *ci[4];
  int nrows[4];
  double nu, dt;
  double *sys[3], *mass[3], *conv[3], *stiff[3], *pmass;
  double *rhs, *sol, *sol_owned, *prev_sol;
  int n_ranks;
  int32_t *rank_u, *rank_p; /* dof units */
  int n_sblocks;
  int32_t *sblock_p;
  int *l_comp, *l_node; /* FESystem local dof -> component / scalar node */
  /* preconditioner state (members of the Precondition* classes, Prec:209-216 etc.) */
  csr_t S;
  double *ilu_F, *ilu_S;
  double *diag_D, *diag_D_inv, *neg_diag_D_inv, *lump_M;
  double *ay_tmp, *ay_tmp2;
  double alpha_simple, alpha_asimple;
  /* compact storage of block (0,0) (orc_set_compact; BASELINE.md section 2 (ii), the "best CPU" variant of bench.py's
   * cpu_baseline): the scalar P2 graph and, per entry, its position in the padded graph of the reference */
  int compact, cs_n;
  int32_t *cs_rp, *cs_ci, *cs_pos, *cs_rank;
  double *cs_F, *cs_ilu;
};

/* threads the library was built for and will use (1 = the serial restatement) */
int orc_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "nsx_oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}
static void *xcalloc(size_t n, size_t s) {
  void *p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "nsx_oracle: out of memory\n");
    abort();
  }
  return p;
}
static void *dup_mem(const void *src, size_t bytes) {
  void *p = xmalloc(bytes);
  memcpy(p, src, bytes);
  return p;
}

/* ------------------------------------------------------------------ BLAS-1 (Epetra_Vector ops) */
static double v_dot(int n, const double *a, const double *b) {
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}
static double v_norm(int n, const double *a) { return sqrt(v_dot(n, a, a)); }
static void v_copy(int n, double *d, const double *s) { memcpy(d, s, (size_t)n * sizeof(double)); }
static void v_zero(int n, double *d) { memset(d, 0, (size_t)n * sizeof(double)); }
static void v_add(int n, double *d, double a, const double *v) { /* d += a v */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) d[i] += a * v[i];
}
static void v_sadd(int n, double *d, double s, double a, const double *v) { /* d = s d + a v */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) d[i] = s * d[i] + a * v[i];
}
static void v_scale(int n, double *d, double a) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) d[i] *= a;
}
static void v_scale_vec(int n, double *d, const double *f) { /* Vector::scale(factors) */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) d[i] *= f[i];
}
static double v_add_and_dot(int n, double *d, double a, const double *v, const double *w) { /* d += a v; return d.w */
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; ++i) {
    d[i] += a * v[i];
    s += d[i] * w[i];
  }
  return s;
}
static int v_all_zero(int n, const double *d) {
  for (int i = 0; i < n; ++i)
    if (d[i] != 0.0) return 0;
  return 1;
}

/* ------------------------------------------------------------------ sparse kernels */
void orc_spmv(int n_rows, const int32_t *rp, const int32_t *ci, const double *v, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_rows; ++i) { /* Epetra_CrsMatrix::Multiply [lib] */
    double s = 0;
    for (int k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
    y[i] = s;
  }
}

static int row_find(const int32_t *rp, const int32_t *ci, int i, int j) {
  int lo = rp[i], hi = rp[i + 1] - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    if (ci[mid] < j)
      lo = mid + 1;
    else if (ci[mid] > j)
      hi = mid - 1;
    else
      return mid;
  }
  return -1;
}

/* C = A * diag(V) * B  — TrilinosWrappers::SparseMatrix::mmult(C, B, V) -> EpetraExt::MatrixMatrix::Multiply [lib]
 * (Prec:144,248,358,468).  Structural product pattern, sorted columns. */
static void csr_free(csr_t *m) {
  free(m->rp);
  free(m->ci);
  free(m->v);
  memset(m, 0, sizeof(*m));
}
static void mmult(csr_t *C, int a_rows, const int32_t *arp, const int32_t *aci, const double *av, int b_cols,
                  const int32_t *brp, const int32_t *bci, const double *bv, const double *V) {
  csr_free(C);
  C->n_rows = a_rows;
  C->n_cols = b_cols;
  C->rp = xcalloc((size_t)a_rows + 1, sizeof(int32_t));
  /* rows are independent: each thread builds a contiguous chunk of rows in buffers of its own (the serial build is the
   * one-chunk case), then the chunks are concatenated */
  int n_chunks = 1;
#ifdef _OPENMP
  n_chunks = omp_get_max_threads();
#endif
  if (n_chunks > a_rows) n_chunks = a_rows > 0 ? a_rows : 1;
  int32_t **cci = xcalloc((size_t)n_chunks, sizeof(*cci));
  double **ccv = xcalloc((size_t)n_chunks, sizeof(*ccv));
  size_t *cnnz = xcalloc((size_t)n_chunks, sizeof(*cnnz));
#pragma omp parallel for schedule(static, 1)
  for (int t = 0; t < n_chunks; ++t) {
    const int i0 = (int)((long long)a_rows * t / n_chunks), i1 = (int)((long long)a_rows * (t + 1) / n_chunks);
    int *mark = xmalloc((size_t)b_cols * sizeof(int));
    double *acc = xcalloc((size_t)b_cols, sizeof(double));
    int *cols = xmalloc((size_t)b_cols * sizeof(int));
    for (int j = 0; j < b_cols; ++j) mark[j] = -1;
    size_t cap = 1024, nnz = 0;
    int32_t *oci = xmalloc(cap * sizeof(int32_t));
    double *ov = xmalloc(cap * sizeof(double));
    for (int i = i0; i < i1; ++i) {
      int cnt = 0;
      for (int ka = arp[i]; ka < arp[i + 1]; ++ka) {
        const int k = aci[ka];
        const double aik = av[ka] * V[k];
        for (int kb = brp[k]; kb < brp[k + 1]; ++kb) {
          const int j = bci[kb];
          if (mark[j] != i) {
            mark[j] = i;
            acc[j] = 0;
            cols[cnt++] = j;
          }
          acc[j] += aik * bv[kb];
        }
      }
      /* sort the row's columns (insertion sort on small rows, qsort-free) */
      for (int a = 1; a < cnt; ++a) {
        int c = cols[a], b = a - 1;
        while (b >= 0 && cols[b] > c) {
          cols[b + 1] = cols[b];
          --b;
        }
        cols[b + 1] = c;
      }
      if (nnz + (size_t)cnt > cap) {
        while (nnz + (size_t)cnt > cap) cap *= 2;
        oci = realloc(oci, cap * sizeof(int32_t));
        ov = realloc(ov, cap * sizeof(double));
        if (!oci || !ov) abort();
      }
      for (int a = 0; a < cnt; ++a) {
        oci[nnz] = cols[a];
        ov[nnz++] = acc[cols[a]];
      }
      C->rp[i + 1] = (int32_t)cnt; /* row length for now */
    }
    free(mark);
    free(acc);
    free(cols);
    cci[t] = oci;
    ccv[t] = ov;
    cnnz[t] = nnz;
  }
  for (int i = 0; i < a_rows; ++i) C->rp[i + 1] += C->rp[i];
  const size_t total = (size_t)C->rp[a_rows];
  C->ci = xmalloc(total * sizeof(int32_t));
  C->v = xmalloc(total * sizeof(double));
  size_t off = 0;
  for (int t = 0; t < n_chunks; ++t) {
    memcpy(C->ci + off, cci[t], cnnz[t] * sizeof(int32_t));
    memcpy(C->v + off, ccv[t], cnnz[t] * sizeof(double));
    off += cnnz[t];
    free(cci[t]);
    free(ccv[t]);
  }
  free(cci);
  free(ccv);
  free(cnnz);
}

/* Ifpack_ILU::Compute, level 0, relax 0, athresh 0, rthresh 1, overlap 0 [lib] — what
 * TrilinosWrappers::PreconditionILU::initialize(A) runs with default AdditionalData (Prec:147-148 etc.).
 * Per rank (block) the local matrix drops off-block columns (Ifpack_LocalFilter).
 * Storage of `out` (same CSR layout as A): strict lower = L (unit diagonal implied), diagonal = 1/d,
 * strict upper = U row scaled by 1/d (unit upper). */
void orc_ilu0_factor(int n_rows, const int32_t *rp, const int32_t *ci, const double *vals, int n_blocks,
                     const int32_t *bptr, double *out) {
  memset(out, 0, (size_t)rp[n_rows] * sizeof(double));
  /* the blocks (MPI ranks) are independent; colflag is indexed by block-local column so every thread owns a small one */
  int max_rows = 0;
  for (int b = 0; b < n_blocks; ++b)
    if (bptr[b + 1] - bptr[b] > max_rows) max_rows = bptr[b + 1] - bptr[b];
#pragma omp parallel
  {
    int *colflag = xmalloc((size_t)(max_rows ? max_rows : 1) * sizeof(int));
    for (int i = 0; i < max_rows; ++i) colflag[i] = -1;
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < n_blocks; ++b) {
      const int r0 = bptr[b], r1 = bptr[b + 1];
      for (int i = r0; i < r1; ++i) {
        int diag = -1;
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
          const int j = ci[k];
          if (j < r0 || j >= r1) continue;
          out[k] = vals[k];
          colflag[j - r0] = k;
          if (j == i) diag = k;
        }
        if (diag < 0) {
          fprintf(stderr, "nsx_oracle: ILU row %d has no diagonal entry\n", i);
          abort();
        }
        for (int kk = rp[i]; kk < rp[i + 1]; ++kk) {
          const int j = ci[kk];
          if (j < r0 || j >= i) continue; /* L part, ascending columns */
          const double multiplier = out[kk];
          /* find diagonal of row j */
          const int dj = row_find(rp, ci, j, j);
          out[kk] *= out[dj]; /* InV[jj] *= DV[j] */
          for (int ku = dj + 1; ku < rp[j + 1]; ++ku) { /* U row of j (already scaled) */
            const int c = ci[ku];
            if (c >= r1) break;
            const int pos = colflag[c - r0];
            if (pos >= 0) out[pos] -= multiplier * out[ku];
          }
        }
        out[diag] = 1.0 / out[diag];
        for (int k = diag + 1; k < rp[i + 1]; ++k) {
          if (ci[k] >= r1) break;
          out[k] *= out[diag];
        }
        for (int k = rp[i]; k < rp[i + 1]; ++k) {
          const int j = ci[k];
          if (j >= r0 && j < r1) colflag[j - r0] = -1;
        }
      }
    }
    free(colflag);
  }
}

/* Ifpack_ILU::ApplyInverse: L solve (unit), D scaling, U solve (unit) [lib]. */
void orc_ilu0_solve(int n_rows, const int32_t *rp, const int32_t *ci, const double *lu, int n_blocks,
                    const int32_t *bptr, const double *b, double *x) {
  (void)n_rows;
#pragma omp parallel for schedule(dynamic, 1)
  for (int blk = 0; blk < n_blocks; ++blk) { /* one block per MPI rank: independent */
    const int r0 = bptr[blk], r1 = bptr[blk + 1];
    for (int i = r0; i < r1; ++i) {
      double s = b[i];
      int k = rp[i];
      for (; k < rp[i + 1] && ci[k] < i; ++k)
        if (ci[k] >= r0) s -= lu[k] * x[ci[k]];
      x[i] = s;
    }
    for (int i = r0; i < r1; ++i) x[i] *= lu[row_find(rp, ci, i, i)];
    for (int i = r1 - 1; i >= r0; --i) {
      double s = x[i];
      for (int k = rp[i + 1] - 1; k >= rp[i] && ci[k] > i; --k)
        if (ci[k] < r1) s -= lu[k] * x[ci[k]];
      x[i] = s;
    }
  }
}

/* ------------------------------------------------------------------ construction */
orc *orc_create(int dim, int n_cells, int dpc, int n_u, int n_p, const int32_t *cell_dofs, const double *cell_coords,
                int n_q, int n_p2, int n_p1, const double *N2, const double *dN2, const double *N1, const double *weights,
                const int32_t *const rowptr[4], const int32_t *const colind[4], double nu, double deltat) {
  /* Krylov temporaries (megabytes each) are allocated and freed in every inner solve, as deal.II's GrowingVectorMemory
   * pool hands them out in the reference; keep them in the heap instead of mmap/munmap + fresh page faults each time */
  mallopt(M_MMAP_THRESHOLD, 1 << 30);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  orc *o = xcalloc(1, sizeof(orc));
  o->dim = dim;
  o->n_cells = n_cells;
  o->dpc = dpc;
  o->n_u = n_u;
  o->n_p = n_p;
  o->n_q = n_q;
  o->np2 = n_p2;
  o->np1 = n_p1;
  o->nu = nu;
  o->dt = deltat;
  o->cell_dofs = dup_mem(cell_dofs, (size_t)n_cells * dpc * sizeof(int32_t));
  o->cell_coords = dup_mem(cell_coords, (size_t)n_cells * (dim + 1) * dim * sizeof(double));
  o->N2 = dup_mem(N2, (size_t)n_q * n_p2 * sizeof(double));
  o->dN2 = dup_mem(dN2, (size_t)n_q * n_p2 * dim * sizeof(double));
  o->N1 = dup_mem(N1, (size_t)n_q * n_p1 * sizeof(double));
  o->w = dup_mem(weights, (size_t)n_q * sizeof(double));
  const int nr[4] = {n_u, n_u, n_p, n_p};
  for (int b = 0; b < 4; ++b) {
    o->nrows[b] = nr[b];
    o->rp[b] = dup_mem(rowptr[b], ((size_t)nr[b] + 1) * sizeof(int32_t));
    o->ci[b] = dup_mem(colind[b], (size_t)rowptr[b][nr[b]] * sizeof(int32_t));
  }
  for (int b = 0; b < 3; ++b) {
    const size_t nnz = (size_t)o->rp[b][nr[b]];
    o->sys[b] = xcalloc(nnz, sizeof(double));
    o->mass[b] = xcalloc(nnz, sizeof(double));
    o->conv[b] = xcalloc(nnz, sizeof(double));
    o->stiff[b] = xcalloc(nnz, sizeof(double));
  }
  o->pmass = xcalloc((size_t)o->rp[3][n_p], sizeof(double));
  const int n = n_u + n_p;
  o->rhs = xcalloc(n, sizeof(double));
  o->sol = xcalloc(n, sizeof(double));
  o->sol_owned = xcalloc(n, sizeof(double));
  o->prev_sol = xcalloc(n, sizeof(double));
  /* FESystem(FE_SimplexP(2)^dim, FE_SimplexP(1)) local numbering (NS3D:31-36) [lib]: per vertex dim velocity
   * components then the pressure; then per line dim velocity components. */
  o->l_comp = xmalloc((size_t)dpc * sizeof(int));
  o->l_node = xmalloc((size_t)dpc * sizeof(int));
  const int nv = dim + 1;
  for (int i = 0; i < dpc; ++i) {
    if (i < nv * (dim + 1)) {
      o->l_comp[i] = i % (dim + 1);
      o->l_node[i] = i / (dim + 1);
    } else {
      o->l_comp[i] = (i - nv * (dim + 1)) % dim;
      o->l_node[i] = nv + (i - nv * (dim + 1)) / dim;
    }
  }
  o->n_ranks = 1;
  o->rank_u = xmalloc(2 * sizeof(int32_t));
  o->rank_p = xmalloc(2 * sizeof(int32_t));
  o->rank_u[0] = 0;
  o->rank_u[1] = n_u;
  o->rank_p[0] = 0;
  o->rank_p[1] = n_p;
  o->n_sblocks = 0;
  o->diag_D = xcalloc(n_u, sizeof(double));
  o->diag_D_inv = xcalloc(n_u, sizeof(double));
  o->neg_diag_D_inv = xcalloc(n_u, sizeof(double));
  o->lump_M = xcalloc(n_u, sizeof(double));
  o->ay_tmp = xcalloc(n_u, sizeof(double));
  o->ay_tmp2 = xcalloc(n_p, sizeof(double));
  o->ilu_F = xcalloc((size_t)o->rp[0][n_u], sizeof(double));
  o->alpha_simple = 0.5;  /* Prec:207 */
  o->alpha_asimple = 1.0; /* Prec:328 */
  return o;
}

void orc_destroy(orc *o) {
  if (!o) return;
  free(o->cell_dofs);
  free(o->cell_coords);
  free(o->N2);
  free(o->dN2);
  free(o->N1);
  free(o->w);
  for (int b = 0; b < 4; ++b) {
    free(o->rp[b]);
    free(o->ci[b]);
  }
  for (int b = 0; b < 3; ++b) {
    free(o->sys[b]);
    free(o->mass[b]);
    free(o->conv[b]);
    free(o->stiff[b]);
  }
  free(o->pmass);
  free(o->rhs);
  free(o->sol);
  free(o->sol_owned);
  free(o->prev_sol);
  free(o->rank_u);
  free(o->rank_p);
  free(o->sblock_p);
  free(o->l_comp);
  free(o->l_node);
  csr_free(&o->S);
  free(o->ilu_F);
  free(o->ilu_S);
  free(o->diag_D);
  free(o->diag_D_inv);
  free(o->neg_diag_D_inv);
  free(o->lump_M);
  free(o->ay_tmp);
  free(o->ay_tmp2);
  free(o->cs_rp);
  free(o->cs_ci);
  free(o->cs_pos);
  free(o->cs_rank);
  free(o->cs_F);
  free(o->cs_ilu);
  free(o);
}

void orc_set_ranks(orc *o, int n_ranks, const int32_t *u_ptr_nodes, const int32_t *p_ptr_nodes) {
  free(o->rank_u);
  free(o->rank_p);
  o->n_ranks = n_ranks;
  o->rank_u = xmalloc(((size_t)n_ranks + 1) * sizeof(int32_t));
  o->rank_p = xmalloc(((size_t)n_ranks + 1) * sizeof(int32_t));
  for (int r = 0; r <= n_ranks; ++r) {
    o->rank_u[r] = o->dim * u_ptr_nodes[r];
    o->rank_p[r] = p_ptr_nodes[r];
  }
}
void orc_set_schur_blocks(orc *o, int n_blocks, const int32_t *p_ptr_nodes) {
  free(o->sblock_p);
  o->n_sblocks = n_blocks;
  o->sblock_p = dup_mem(p_ptr_nodes, ((size_t)n_blocks + 1) * sizeof(int32_t));
}

double *orc_solution(orc *o) { return o->sol; }
double *orc_solution_owned(orc *o) { return o->sol_owned; }
double *orc_rhs(orc *o) { return o->rhs; }
double *orc_matrix_values(orc *o, int which, int block) {
  if (which == 4) return o->pmass;
  if (block < 0 || block > 2) return NULL;
  switch (which) {
    case 0: return o->sys[block];
    case 1: return o->mass[block];
    case 2: return o->conv[block];
    case 3: return o->stiff[block];
  }
  return NULL;
}
int orc_schur(orc *o, const int32_t **rp, const int32_t **ci, const double **v) {
  *rp = o->S.rp;
  *ci = o->S.ci;
  *v = o->S.v;
  return o->S.n_rows;
}
const double *orc_ilu_F(orc *o) { return o->ilu_F; }
const double *orc_ilu_S(orc *o) { return o->ilu_S; }

/* ------------------------------------------------------------------ assembly */
/* BlockSparseMatrix::add(dof_indices, cell_matrix) with elide_zero_values = true [lib] (NS3D:306-311,500). */
static void block_add(orc *o, double *const blk[3], const int32_t *dofs, const double *cm) {
  const int n = o->dpc, n_u = o->n_u;
  for (int i = 0; i < n; ++i) {
    const int gi = dofs[i];
    for (int j = 0; j < n; ++j) {
      const double v = cm[i * n + j];
      if (v == 0.0) continue;
      const int gj = dofs[j];
      int b, r, c;
      if (gi < n_u) {
        r = gi;
        if (gj < n_u) { b = 0; c = gj; } else { b = 1; c = gj - n_u; }
      } else {
        r = gi - n_u;
        if (gj < n_u) { b = 2; c = gj; } else {
          fprintf(stderr, "nsx_oracle: non-zero (p,p) entry has no sparsity slot (NS3D:114-115)\n");
          abort();
        }
      }
      const int pos = row_find(o->rp[b], o->ci[b], r, c);
      if (pos < 0) {
        fprintf(stderr, "nsx_oracle: entry (%d,%d) not in sparsity pattern of block %d\n", r, c, b);
        abort();
      }
#pragma omp atomic
      blk[b][pos] += v; /* concurrent cells share rows: the sum-into of Epetra_FECrsMatrix, any order */
    }
  }
}

/* FEValues::reinit for an affine simplex [lib]: JxW_q and physical gradients of the scalar shape functions. */
static void fe_reinit(const orc *o, int cell, double *JxW, double *gradN2 /*[q][np2][dim]*/) {
  const int dim = o->dim;
  const double *X = o->cell_coords + (size_t)cell * (dim + 1) * dim;
  double J[3][3] = {{0}}, Ji[3][3] = {{0}}, det;
  for (int d = 0; d < dim; ++d)
    for (int k = 0; k < dim; ++k) J[d][k] = X[(k + 1) * dim + d] - X[d];
  if (dim == 2) {
    det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    Ji[0][0] = J[1][1] / det;
    Ji[0][1] = -J[0][1] / det;
    Ji[1][0] = -J[1][0] / det;
    Ji[1][1] = J[0][0] / det;
  } else {
    det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
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
  }
  for (int q = 0; q < o->n_q; ++q) {
    JxW[q] = fabs(det) * o->w[q];
    for (int a = 0; a < o->np2; ++a)
      for (int d = 0; d < dim; ++d) {
        double g = 0;
        for (int k = 0; k < dim; ++k) g += Ji[k][d] * o->dN2[((size_t)q * o->np2 + a) * dim + k]; /* J^{-T} grad_hat */
        gradN2[((size_t)q * o->np2 + a) * dim + d] = g;
      }
  }
}

/* One routine for both NavierStokes::assemble (first != 0) and ::assemble_time_step (first == 0). */
static void assemble_impl(orc *o, int first, int flags) {
  const int dim = o->dim, n = o->dpc, n_q = o->n_q;
  const double nu = o->nu, deltat = o->dt;
  const size_t nnz[3] = {(size_t)o->rp[0][o->n_u], (size_t)o->rp[1][o->n_u], (size_t)o->rp[2][o->n_p]};

  if (first) { /* NS3D:191-196 */
    for (int b = 0; b < 3; ++b) {
      memset(o->sys[b], 0, nnz[b] * sizeof(double));
      memset(o->mass[b], 0, nnz[b] * sizeof(double));
      memset(o->stiff[b], 0, nnz[b] * sizeof(double));
      memset(o->conv[b], 0, nnz[b] * sizeof(double));
    }
    memset(o->pmass, 0, (size_t)o->rp[3][o->n_p] * sizeof(double));
  } else { /* NS3D:388,395 */
    for (int b = 0; b < 3; ++b) {
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < nnz[b]; ++k) o->sys[b][k] += -1. * o->conv[b][k];
    }
    for (int b = 0; b < 3; ++b) memset(o->conv[b], 0, nnz[b] * sizeof(double));
  }
  v_zero(o->n_u + o->n_p, o->rhs); /* NS3D:195,396 */

  /* the cell loop: every MPI rank of the reference walks its own cells; here threads share the loop */
#pragma omp parallel
  {
  double *cell_matrix = xmalloc((size_t)n * n * sizeof(double));
  double *cell_mass = xmalloc((size_t)n * n * sizeof(double));
  double *cell_stiff = xmalloc((size_t)n * n * sizeof(double));
  double *cell_conv = xmalloc((size_t)n * n * sizeof(double));
  double *cell_pmass = xmalloc((size_t)n * n * sizeof(double));
  double *cell_rhs = xmalloc((size_t)n * sizeof(double));
  double *JxW = xmalloc((size_t)n_q * sizeof(double));
  double *gradN2 = xmalloc((size_t)n_q * o->np2 * dim * sizeof(double));
  double *cur_val = xmalloc((size_t)n_q * dim * sizeof(double));
  double *cur_div = xmalloc((size_t)n_q * sizeof(double));
#pragma omp for schedule(static)
  for (int cell = 0; cell < o->n_cells; ++cell) { /* NS3D:208,420 (all cells are "locally owned" here) */
    const int32_t *dofs = o->cell_dofs + (size_t)cell * n;
    fe_reinit(o, cell, JxW, gradN2);
    memset(cell_conv, 0, (size_t)n * n * sizeof(double));
    memset(cell_rhs, 0, (size_t)n * sizeof(double));
    if (first) {
      memset(cell_matrix, 0, (size_t)n * n * sizeof(double));
      memset(cell_mass, 0, (size_t)n * n * sizeof(double));
      memset(cell_stiff, 0, (size_t)n * n * sizeof(double));
      memset(cell_pmass, 0, (size_t)n * n * sizeof(double));
    }
    /* fe_values[velocity].get_function_values / _divergences(solution, ...) (NS3D:224-228,436-440) */
    for (int q = 0; q < n_q; ++q) {
      for (int d = 0; d < dim; ++d) cur_val[q * dim + d] = 0;
      cur_div[q] = 0;
      for (int i = 0; i < n; ++i) {
        const int c = o->l_comp[i];
        if (c >= dim) continue;
        const double ui = o->sol[dofs[i]];
        cur_val[q * dim + c] += ui * o->N2[(size_t)q * o->np2 + o->l_node[i]];
        cur_div[q] += ui * gradN2[((size_t)q * o->np2 + o->l_node[i]) * dim + c];
      }
    }
    for (int q = 0; q < n_q; ++q) { /* NS3D:230-272, 442-463 */
      const double *w = cur_val + (size_t)q * dim;
      for (int i = 0; i < n; ++i) {
        const int ci = o->l_comp[i], ai = o->l_node[i];
        const int i_vel = ci < dim;
        const double phi_i = i_vel ? o->N2[(size_t)q * o->np2 + ai] : 0.0;  /* fe_values[velocity].value(i,q)[ci] */
        const double psi_i = i_vel ? 0.0 : o->N1[(size_t)q * o->np1 + ai]; /* fe_values[pressure].value(i,q) */
        const double *gi = i_vel ? gradN2 + ((size_t)q * o->np2 + ai) * dim : NULL;
        const double div_i = i_vel ? gi[ci] : 0.0;
        for (int j = 0; j < n; ++j) {
          const int cj = o->l_comp[j], aj = o->l_node[j];
          const int j_vel = cj < dim;
          const double phi_j = j_vel ? o->N2[(size_t)q * o->np2 + aj] : 0.0;
          const double psi_j = j_vel ? 0.0 : o->N1[(size_t)q * o->np1 + aj];
          const double *gj = j_vel ? gradN2 + ((size_t)q * o->np2 + aj) * dim : NULL;
          const double div_j = j_vel ? gj[cj] : 0.0;
          const int same = i_vel && j_vel && ci == cj;
          /* scalar_product(value_i, value_j): single non-zero components ci, cj */
          const double vv = same ? phi_i * phi_j : 0.0;
          if (first) {
            double gg = 0.0; /* scalar_product(gradient_i, gradient_j): rows ci, cj */
            if (same)
              for (int d = 0; d < dim; ++d) gg += gi[d] * gj[d];
            cell_stiff[i * n + j] += nu * gg * JxW[q];                /* NS3D:246 */
            cell_mass[i * n + j] += vv / deltat * JxW[q];             /* NS3D:249 */
          }
          double conv = 0.0; /* scalar_product(gradient(j,q) * w, value(i,q)) */
          if (same) {
            double gw = 0.0;
            for (int d = 0; d < dim; ++d) gw += gj[d] * w[d];
            conv = gw * phi_i;
          }
          cell_conv[i * n + j] += conv * JxW[q];                      /* NS3D:252,456 */
          if (flags & ORC_TEMAM)
            cell_conv[i * n + j] += 0.5 * cur_div[q] * vv * JxW[q];   /* NS3D:255; NS2D:446; Conv:490 */
          if (first && (flags & ORC_DOUBLE_CONVECTION))
            cell_conv[i * n + j] += conv * JxW[q];                    /* Conv:284 (second copy of Conv:277) */
          if (first) {
            cell_matrix[i * n + j] -= psi_j * div_i * JxW[q];         /* NS3D:258 */
            cell_matrix[i * n + j] += psi_i * div_j * JxW[q];         /* NS3D:261 */
            cell_pmass[i * n + j] += psi_i * psi_j / nu * JxW[q];     /* NS3D:264 */
          }
        }
        /* scalar_product(current_velocity_values[q], value(i,q)) * JxW / deltat  (NS3D:269,459) */
        cell_rhs[i] += (i_vel ? w[ci] * phi_i : 0.0) * JxW[q] / deltat;
      }
    }
    if (first) {
      block_add(o, o->sys, dofs, cell_matrix);   /* NS3D:306 */
      block_add(o, o->mass, dofs, cell_mass);    /* NS3D:307 */
    }
    block_add(o, o->conv, dofs, cell_conv);      /* NS3D:308,500 */
    if (first) block_add(o, o->stiff, dofs, cell_stiff); /* NS3D:309 */
    for (int i = 0; i < n; ++i) { /* NS3D:310,501 */
#pragma omp atomic
      o->rhs[dofs[i]] += cell_rhs[i];
    }
    if (first) { /* pressure_mass.add (NS3D:311): only (p,p) entries are non-zero */
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const double v = cell_pmass[i * n + j];
          if (v == 0.0) continue;
          const int pos = row_find(o->rp[3], o->ci[3], dofs[i] - o->n_u, dofs[j] - o->n_u);
#pragma omp atomic
          o->pmass[pos] += v;
        }
    }
  }
  free(cell_matrix);
  free(cell_mass);
  free(cell_stiff);
  free(cell_conv);
  free(cell_pmass);
  free(cell_rhs);
  free(JxW);
  free(gradN2);
  free(cur_val);
  free(cur_div);
  } /* omp parallel */
  if (first) { /* NS3D:322-324 */
    for (int b = 0; b < 3; ++b) {
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < nnz[b]; ++k) o->sys[b][k] += 1. * o->mass[b][k];
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < nnz[b]; ++k) o->sys[b][k] += 1. * o->conv[b][k];
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < nnz[b]; ++k) o->sys[b][k] += 1. * o->stiff[b][k];
    }
  } else { /* NS3D:512 */
    for (int b = 0; b < 3; ++b) {
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < nnz[b]; ++k) o->sys[b][k] += 1. * o->conv[b][k];
    }
  }
}

void orc_assemble(orc *o, int flags) { assemble_impl(o, 1, flags); }
void orc_assemble_time_step(orc *o, int flags) { assemble_impl(o, 0, flags); }

void orc_add_rhs(orc *o, int n, const int32_t *dofs, const double *vals) {
  for (int k = 0; k < n; ++k) o->rhs[dofs[k]] += vals[k];
}

/* MatrixTools::apply_boundary_values for TrilinosWrappers::BlockSparseMatrix, eliminate_columns = false [lib]
 * (NS3D:353,541).  Per diagonal block and per rank: first non-zero |diagonal| of the rank's local range
 * (1 if none) becomes the diagonal of every constrained row; the rows are cleared in the diagonal AND the
 * off-diagonal blocks; rhs = diag * value; `solution` (ghosted) = value.  Only velocity dofs are ever
 * constrained (ComponentMask, NS3D:337-338), the pressure block sees an empty map. */
void orc_apply_boundary_values(orc *o, int n, const int32_t *dofs, const double *vals) {
  const int32_t *rp = o->rp[0], *ci = o->ci[0];
  double *F = o->sys[0], *G = o->sys[1];
#pragma omp parallel for schedule(dynamic, 1)
  for (int r = 0; r < o->n_ranks; ++r) { /* every rank handles the constrained rows of its own range */
    const int lo = o->rank_u[r], hi = o->rank_u[r + 1];
    double first_nonzero_diag = 1;
    for (int i = lo; i < hi; ++i) {
      const int d = row_find(rp, ci, i, i);
      if (d >= 0 && F[d] != 0) {
        first_nonzero_diag = fabs(F[d]);
        break;
      }
    }
    for (int k = 0; k < n; ++k) {
      const int i = dofs[k];
      if (i < lo || i >= hi) continue;
      for (int p = rp[i]; p < rp[i + 1]; ++p) F[p] = (ci[p] == i) ? first_nonzero_diag : 0.0; /* clear_rows(rows, diag) */
      for (int p = o->rp[1][i]; p < o->rp[1][i + 1]; ++p) G[p] = 0.0; /* off-diagonal block row cleared */
      o->sol[i] = vals[k];
      o->rhs[i] = vals[k] * first_nonzero_diag;
    }
  }
}

/* ------------------------------------------------------------------ Krylov solvers [lib] */
typedef void (*op_fn)(void *ctx, double *dst, const double *src);

typedef struct {
  int status; /* 0 success, 1 failure */
  int steps;
  double last;
} sc_result;

static int sc_check(int step, double value, double tol, int maxsteps) { /* SolverControl::check: 0 iterate 1 success 2 fail */
  if (value <= tol) return 1;
  if (step >= maxsteps || isnan(value)) return 2;
  return 0;
}

/* SolverGMRES<VectorType>::solve, left preconditioning, default residual, restart N_TMP-2, modified Gram-Schmidt
 * with the Kelley re-orthogonalisation test every 5th inner iteration. */
static sc_result gmres(op_fn A, void *actx, double *x, const double *b, op_fn P, void *pctx, int n, double tol,
                       int maxiter) {
  sc_result res = {1, 0, 0.0};
  double *tmp[N_TMP] = {0};
  double H[N_TMP][N_TMP - 1];
  double gamma[N_TMP], ci_[N_TMP - 1], si_[N_TMP - 1], h[N_TMP];
  int accumulated = 0, state = 0, dimk = 0;
  int re_orth = 0;
  tmp[0] = xcalloc(n, sizeof(double));
  tmp[N_TMP - 1] = xcalloc(n, sizeof(double));
  double *v = tmp[0], *p = tmp[N_TMP - 1];
  do {
    memset(h, 0, sizeof(h));
    A(actx, p, x);
    v_sadd(n, p, -1., 1., b);
    P(pctx, v, p);
    double rho = v_norm(n, v);
    res.last = rho;
    state = sc_check(accumulated, rho, tol, maxiter);
    if (state != 0) break;
    gamma[0] = rho;
    v_scale(n, v, 1. / rho);
    dimk = 0;
    for (int inner = 0; inner < N_TMP - 2 && state == 0; ++inner) {
      ++accumulated;
      if (!tmp[inner + 1]) tmp[inner + 1] = xcalloc(n, sizeof(double));
      double *vv = tmp[inner + 1];
      A(actx, p, tmp[inner]);
      P(pctx, vv, p);
      dimk = inner + 1;
      /* modified_gram_schmidt */
      double norm_vv_start = 0;
      const int consider = (!re_orth) && (inner % 5 == 4);
      if (consider) norm_vv_start = v_norm(n, vv);
      h[0] = v_dot(n, vv, tmp[0]);
      for (int i = 1; i < dimk; ++i) h[i] = v_add_and_dot(n, vv, -h[i - 1], tmp[i - 1], tmp[i]);
      double s = sqrt(v_add_and_dot(n, vv, -h[dimk - 1], tmp[dimk - 1], vv));
      if (consider && !(s > 10. * norm_vv_start * sqrt(2.220446049250313e-16))) re_orth = 1;
      if (re_orth) {
        double htmp = v_dot(n, vv, tmp[0]);
        h[0] += htmp;
        for (int i = 1; i < dimk; ++i) {
          htmp = v_add_and_dot(n, vv, -htmp, tmp[i - 1], tmp[i]);
          h[i] += htmp;
        }
        s = sqrt(v_add_and_dot(n, vv, -htmp, tmp[dimk - 1], vv));
      }
      h[inner + 1] = s;
      if (s != 0) v_scale(n, vv, 1. / s);
      /* givens_rotation(h, gamma, ci, si, inner) */
      for (int i = 0; i < inner; ++i) {
        const double sn = si_[i], cs = ci_[i], dummy = h[i];
        h[i] = cs * dummy + sn * h[i + 1];
        h[i + 1] = -sn * dummy + cs * h[i + 1];
      }
      const double r = 1. / sqrt(h[inner] * h[inner] + h[inner + 1] * h[inner + 1]);
      si_[inner] = h[inner + 1] * r;
      ci_[inner] = h[inner] * r;
      h[inner] = ci_[inner] * h[inner] + si_[inner] * h[inner + 1];
      gamma[inner + 1] = -si_[inner] * gamma[inner];
      gamma[inner] *= ci_[inner];
      for (int i = 0; i < dimk; ++i) H[i][inner] = h[i];
      rho = fabs(gamma[dimk]);
      res.last = rho;
      state = sc_check(accumulated, rho, tol, maxiter);
    }
    /* H1.backward(h, gamma) */
    double y[N_TMP];
    for (int i = dimk - 1; i >= 0; --i) {
      double s = gamma[i];
      for (int j = i + 1; j < dimk; ++j) s -= H[i][j] * y[j];
      y[i] = s / H[i][i];
    }
    for (int i = 0; i < dimk; ++i) v_add(n, x, y[i], tmp[i]);
  } while (state == 0);
  res.status = state == 1 ? 0 : 1;
  res.steps = accumulated;
  for (int i = 0; i < N_TMP; ++i) free(tmp[i]);
  return res;
}

/* SolverCG<VectorType>::solve with a preconditioner. */
static sc_result cg(op_fn A, void *actx, double *x, const double *b, op_fn P, void *pctx, int n, double tol, int maxiter) {
  sc_result res = {1, 0, 0.0};
  double *g = xcalloc(n, sizeof(double)), *d = xcalloc(n, sizeof(double)), *h = xcalloc(n, sizeof(double));
  int it = 0, conv;
  double gh, beta;
  if (!v_all_zero(n, x)) {
    A(actx, g, x);
    v_add(n, g, -1., b);
  } else {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) g[i] = -b[i];
  }
  double r = v_norm(n, g);
  res.last = r;
  conv = sc_check(0, r, tol, maxiter);
  if (conv == 0) {
    P(pctx, h, g);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) d[i] = -h[i];
    gh = v_dot(n, g, h);
    while (conv == 0) {
      it++;
      A(actx, h, d);
      double alpha = v_dot(n, d, h);
      alpha = gh / alpha;
      v_add(n, x, alpha, d);
      r = sqrt(fabs(v_add_and_dot(n, g, alpha, h, g)));
      res.last = r;
      conv = sc_check(it, r, tol, maxiter);
      if (conv != 0) break;
      P(pctx, h, g);
      beta = gh;
      gh = v_dot(n, g, h);
      beta = gh / beta;
      v_sadd(n, d, beta, -1., h);
    }
  }
  res.status = conv == 1 ? 0 : 1;
  res.steps = it;
  free(g);
  free(d);
  free(h);
  return res;
}

/* ------------------------------------------------------------------ compact storage of block (0,0)
 * NOT the reference's layout: the reference stores every coupling of the dim velocity components of two P2 nodes
 * (NS3D:109-119), i.e. dim^2 entries of which dim carry the same scalar value and the others are explicit zeros.  With
 * orc_set_compact(o, 1) the products with system(0,0) and its per-rank ILU(0) run on the scalar P2 x P2 operator applied to
 * the dim interleaved components -- the same algorithm and the same numbers up to rounding (the cross-component fill of the
 * padded ILU(0) is exactly zero), 1/dim^2 of the matrix traffic.  Used only by bench.py's cpu_baseline leg ("best CPU").
 * Assembly, Dirichlet rows, the Schur product and the rectangular blocks keep the reference-shaped layout. */
void orc_set_compact(orc *o, int on) {
  o->compact = on != 0;
  if (!o->compact || o->cs_rp) return;
  const int dim = o->dim, n = o->n_u / dim;
  const int32_t *rp = o->rp[0], *ci = o->ci[0];
  o->cs_n = n;
  o->cs_rp = xcalloc((size_t)n + 1, sizeof(int32_t));
  for (int i = 0; i < n; ++i) {
    int c = 0;
    for (int k = rp[dim * i]; k < rp[dim * i + 1]; ++k) c += ci[k] % dim == 0;
    o->cs_rp[i + 1] = o->cs_rp[i] + c;
  }
  o->cs_ci = xmalloc((size_t)o->cs_rp[n] * sizeof(int32_t));
  o->cs_pos = xmalloc((size_t)o->cs_rp[n] * sizeof(int32_t));
  for (int i = 0; i < n; ++i) {
    int q = o->cs_rp[i];
    for (int k = rp[dim * i]; k < rp[dim * i + 1]; ++k)
      if (ci[k] % dim == 0) {
        o->cs_ci[q] = ci[k] / dim;
        o->cs_pos[q++] = k;
      }
  }
  o->cs_F = xcalloc((size_t)o->cs_rp[n], sizeof(double));
  o->cs_ilu = xcalloc((size_t)o->cs_rp[n], sizeof(double));
}

typedef struct {
  int n, dim; /* n scalar rows, dim interleaved components */
  const int32_t *rp, *ci;
  const double *v;
} csr3_ctx;
static void op_csr3(void *c, double *dst, const double *src) {
  const csr3_ctx *m = c;
  const int dim = m->dim;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m->n; ++i) {
    double s[3] = {0, 0, 0};
    for (int k = m->rp[i]; k < m->rp[i + 1]; ++k) {
      const double a = m->v[k];
      const double *xj = src + (size_t)m->ci[k] * dim;
      for (int d = 0; d < dim; ++d) s[d] += a * xj[d];
    }
    for (int d = 0; d < dim; ++d) dst[(size_t)i * dim + d] = s[d];
  }
}
typedef struct {
  int n, dim;
  const int32_t *rp, *ci;
  const double *lu;
  int nb;
  const int32_t *bptr; /* node units */
} ilu3_ctx;
/* orc_ilu0_solve on the scalar factors for dim interleaved right-hand sides */
static void op_ilu3(void *c, double *x, const double *b) {
  const ilu3_ctx *m = c;
  const int dim = m->dim;
  const int32_t *rp = m->rp, *ci = m->ci;
  const double *lu = m->lu;
#pragma omp parallel for schedule(dynamic, 1)
  for (int blk = 0; blk < m->nb; ++blk) {
    const int r0 = m->bptr[blk], r1 = m->bptr[blk + 1];
    for (int i = r0; i < r1; ++i) {
      double s[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) s[d] = b[(size_t)i * dim + d];
      int k = rp[i];
      for (; k < rp[i + 1] && ci[k] < i; ++k)
        if (ci[k] >= r0)
          for (int d = 0; d < dim; ++d) s[d] -= lu[k] * x[(size_t)ci[k] * dim + d];
      for (int d = 0; d < dim; ++d) x[(size_t)i * dim + d] = s[d];
    }
    for (int i = r0; i < r1; ++i) {
      const double di = lu[row_find(rp, ci, i, i)];
      for (int d = 0; d < dim; ++d) x[(size_t)i * dim + d] *= di;
    }
    for (int i = r1 - 1; i >= r0; --i) {
      double s[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) s[d] = x[(size_t)i * dim + d];
      for (int k = rp[i + 1] - 1; k >= rp[i] && ci[k] > i; --k)
        if (ci[k] < r1)
          for (int d = 0; d < dim; ++d) s[d] -= lu[k] * x[(size_t)ci[k] * dim + d];
      for (int d = 0; d < dim; ++d) x[(size_t)i * dim + d] = s[d];
    }
  }
}
/* system(0,0) in compact form + its ILU(0) per rank: called by orc_prec_initialize when compact */
static void compact_refresh(orc *o) {
  const int n = o->cs_n;
#pragma omp parallel for schedule(static)
  for (int k = 0; k < o->cs_rp[n]; ++k) o->cs_F[k] = o->sys[0][o->cs_pos[k]];
  free(o->cs_rank);
  o->cs_rank = xmalloc(((size_t)o->n_ranks + 1) * sizeof(int32_t));
  for (int r = 0; r <= o->n_ranks; ++r) o->cs_rank[r] = o->rank_u[r] / o->dim;
  orc_ilu0_factor(n, o->cs_rp, o->cs_ci, o->cs_F, o->n_ranks, o->cs_rank, o->cs_ilu);
}

/* ------------------------------------------------------------------ operators */
typedef struct {
  int n;
  const int32_t *rp, *ci;
  const double *v;
} csr_ctx;
static void op_csr(void *c, double *dst, const double *src) {
  csr_ctx *m = c;
  orc_spmv(m->n, m->rp, m->ci, m->v, src, dst);
}
typedef struct {
  int n;
  const int32_t *rp, *ci;
  const double *lu;
  int nb;
  const int32_t *bptr;
} ilu_ctx;
static void op_ilu(void *c, double *dst, const double *src) {
  ilu_ctx *m = c;
  orc_ilu0_solve(m->n, m->rp, m->ci, m->lu, m->nb, m->bptr, src, dst);
}

/* BlockSparseMatrix::vmult: dst_u = F x_u + block(0,1) x_p ; dst_p = block(1,0) x_u (block (1,1) is empty). */
void orc_system_vmult(orc *o, double *dst, const double *src) {
  const int n_u = o->n_u, n_p = o->n_p;
  if (o->compact && o->cs_F) {
    csr3_ctx Fc = {o->cs_n, o->dim, o->cs_rp, o->cs_ci, o->cs_F};
    op_csr3(&Fc, dst, src);
  } else {
    orc_spmv(n_u, o->rp[0], o->ci[0], o->sys[0], src, dst);
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_u; ++i) {
    double s = 0;
    for (int k = o->rp[1][i]; k < o->rp[1][i + 1]; ++k) s += o->sys[1][k] * src[n_u + o->ci[1][k]];
    dst[i] += s;
  }
  orc_spmv(n_p, o->rp[2], o->ci[2], o->sys[2], src, dst + n_u);
}
static void op_system(void *c, double *dst, const double *src) { orc_system_vmult((orc *)c, dst, src); }

/* ------------------------------------------------------------------ preconditioners (Prec:118-534) */
typedef struct {
  orc *o;
  int type;
  double inner_rtol;
  int inner_maxiter;
  orc_stats *st;
} prec_ctx;

static const int32_t *schur_blocks(orc *o, int *nb) {
  if (o->n_sblocks > 0) {
    *nb = o->n_sblocks;
    return o->sblock_p;
  }
  *nb = o->n_ranks;
  return o->rank_p;
}

void orc_prec_initialize(orc *o, int type) {
  const int n_u = o->n_u, n_p = o->n_p;
  const int32_t *rp = o->rp[0], *ci = o->ci[0];
  const double *F = o->sys[0], *M = o->mass[0];
  const double *V = NULL;
  if (type == ORC_YOSIDA) { /* Prec:350-358 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_u; ++i) {
      const double m = M[row_find(rp, ci, i, i)];
      o->diag_D_inv[i] = 1.0 / m;
      o->neg_diag_D_inv[i] = -1.0 / m;
    }
    V = o->neg_diag_D_inv;
  } else if (type == ORC_SIMPLE || type == ORC_ASIMPLE) { /* Prec:135-144, 239-248 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_u; ++i) {
      const double temp = F[row_find(rp, ci, i, i)];
      o->diag_D[i] = temp;
      o->diag_D_inv[i] = 1.0 / temp;
      o->neg_diag_D_inv[i] = -1.0 / temp;
    }
    V = o->neg_diag_D_inv;
  } else { /* aYosida, Prec:447-468 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_u; ++i) {
      const double temp = F[row_find(rp, ci, i, i)];
      o->diag_D[i] = temp;
      o->diag_D_inv[i] = 1.0 / temp;
      double s = 0.0;
      for (int k = rp[i]; k < rp[i + 1]; ++k) s += fabs(M[k]);
      o->lump_M[i] = -1.0 / s;
    }
    V = o->lump_M;
  }
  /* B->mmult(negative_S, *B_T, V): B = block(1,0), B_T = block(0,1) */
  mmult(&o->S, n_p, o->rp[2], o->ci[2], o->sys[2], n_p, o->rp[1], o->ci[1], o->sys[1], V);
  /* preconditioner_F.initialize(*F); preconditioner_S.initialize(negative_S) */
  if (o->compact) compact_refresh(o); /* scalar operator + scalar ILU(0) per rank (bench.py's "best CPU" leg only) */
  else orc_ilu0_factor(n_u, rp, ci, F, o->n_ranks, o->rank_u, o->ilu_F);
  free(o->ilu_S);
  o->ilu_S = xcalloc((size_t)o->S.rp[n_p], sizeof(double));
  int nb;
  const int32_t *bp = schur_blocks(o, &nb);
  orc_ilu0_factor(n_p, o->S.rp, o->S.ci, o->S.v, nb, bp, o->ilu_S);
}

static void count_F(prec_ctx *pc, sc_result r) {
  if (pc->st) {
    pc->st->inner_F_iterations += r.steps;
    pc->st->n_F_solves++;
    if (r.status) pc->st->status = 2;
  }
}
static void count_S(prec_ctx *pc, sc_result r) {
  if (pc->st) {
    pc->st->inner_S_iterations += r.steps;
    pc->st->n_S_solves++;
    if (r.status) pc->st->status = 2;
  }
}

static void prec_vmult(void *c, double *dst, const double *src) {
  prec_ctx *pc = c;
  orc *o = pc->o;
  const int n_u = o->n_u, n_p = o->n_p;
  const double tol = pc->inner_rtol;
  const int maxit = pc->inner_maxiter;
  csr_ctx Fm_padded = {n_u, o->rp[0], o->ci[0], o->sys[0]};
  csr3_ctx Fm_compact = {o->cs_n, o->dim, o->cs_rp, o->cs_ci, o->cs_F};
  ilu3_ctx PF_compact = {o->cs_n, o->dim, o->cs_rp, o->cs_ci, o->cs_ilu, o->n_ranks, o->cs_rank};
  const op_fn opF = o->compact ? op_csr3 : op_csr, opPF = o->compact ? op_ilu3 : op_ilu;
  void *Fm_ = o->compact ? (void *)&Fm_compact : (void *)&Fm_padded;
  csr_ctx Bm = {n_p, o->rp[2], o->ci[2], o->sys[2]};   /* B   = block(1,0) */
  csr_ctx BTm = {n_u, o->rp[1], o->ci[1], o->sys[1]};  /* B_T = block(0,1) */
  csr_ctx Sm = {n_p, o->S.rp, o->S.ci, o->S.v};
  ilu_ctx PF_padded = {n_u, o->rp[0], o->ci[0], o->ilu_F, o->n_ranks, o->rank_u};
  void *PF_ = o->compact ? (void *)&PF_compact : (void *)&PF_padded;
  int nb;
  const int32_t *bp = schur_blocks(o, &nb);
  ilu_ctx PS = {n_p, o->S.rp, o->S.ci, o->ilu_S, nb, bp};
  const double *src_u = src, *src_p = src + n_u;
  double *dst_u = dst, *dst_p = dst + n_u;

  if (pc->type == ORC_YOSIDA) { /* Prec:365-408 */
    double *yu = dup_mem(src_u, (size_t)n_u * sizeof(double));
    double *yp = dup_mem(src_p, (size_t)n_p * sizeof(double));
    double *tmp = dup_mem(src_p, (size_t)n_p * sizeof(double));
    double *tmp2 = dup_mem(src_u, (size_t)n_u * sizeof(double));
    count_F(pc, gmres(opF, Fm_, yu, src_u, opPF, PF_, n_u, tol * v_norm(n_u, src_u), maxit)); /* :382 */
    op_csr(&Bm, tmp, yu);                                                                           /* :385 */
    v_add(n_p, tmp, -1.0, src_p);                                                                   /* :386 */
    count_S(pc, cg(op_csr, &Sm, yp, tmp, op_ilu, &PS, n_p, tol * v_norm(n_p, tmp), maxit));         /* :388-390 */
    v_copy(n_p, dst_p, yp);                                                                         /* :394 */
    op_csr(&BTm, tmp2, dst_p);                                                                      /* :398 */
    double *res = xcalloc(n_u, sizeof(double));                                                     /* :401 */
    v_copy(n_u, dst_u, yu);                                                                         /* :402 */
    count_F(pc, gmres(opF, Fm_, res, tmp2, opPF, PF_, n_u, tol * v_norm(n_u, tmp2), maxit));   /* :403-405 */
    v_sadd(n_u, dst_u, -1., 1., res); /* dst.block(0).sadd(-1,res): dst = -dst + res                   :406 */
    free(yu);
    free(yp);
    free(tmp);
    free(tmp2);
    free(res);
  } else if (pc->type == ORC_SIMPLE) { /* Prec:151-205 */
    double *sol1_u = dup_mem(src_u, (size_t)n_u * sizeof(double));
    double *sol1_p = dup_mem(src_p, (size_t)n_p * sizeof(double));
    double *temp_1 = dup_mem(src_p, (size_t)n_p * sizeof(double));
    count_F(pc, gmres(opF, Fm_, sol1_u, src_u, opPF, PF_, n_u, tol * v_norm(n_u, src_u), maxit)); /* :173 */
    op_csr(&Bm, temp_1, sol1_u);                                                                        /* :175 */
    v_add(n_p, temp_1, -1.0, src_p); /* temp_1 -= src.block(1)                                             :176 */
    count_S(pc, cg(op_csr, &Sm, sol1_p, temp_1, op_ilu, &PS, n_p, tol * v_norm(n_p, temp_1), maxit));   /* :179-182 */
    v_copy(n_p, dst_p, sol1_p);                                                                         /* :194 */
    v_scale(n_p, dst_p, 1. / o->alpha_simple);                                                          /* :195 */
    v_copy(n_u, dst_u, sol1_u);                                                                         /* :199 */
    double *tmp = dup_mem(src_u, (size_t)n_u * sizeof(double));                                         /* :200 */
    op_csr(&BTm, tmp, dst_p);                                                                           /* :201 */
    v_scale_vec(n_u, tmp, o->diag_D_inv);                                                               /* :202 */
    v_add(n_u, dst_u, -1.0, tmp);                                                                       /* :203 */
    free(sol1_u);
    free(sol1_p);
    free(temp_1);
    free(tmp);
  } else if (pc->type == ORC_ASIMPLE) { /* Prec:254-311 */
    double *tmp_u = xcalloc(n_u, sizeof(double)), *tmp_p = xcalloc(n_p, sizeof(double));                 /* :266 */
    count_F(pc, gmres(opF, Fm_, dst_u, src_u, opPF, PF_, n_u, tol * v_norm(n_u, src_u), maxit));    /* :271-273 */
    op_csr(&Bm, dst_p, dst_u);                                                                            /* :280 */
    v_sadd(n_p, dst_p, -1.0, 1.0, src_p); /* dst1.sadd(-1.0, src1): dst1 = -dst1 + src1                      :281 */
    v_copy(n_p, tmp_p, dst_p);                                                                            /* :282 */
    count_S(pc, gmres(op_csr, &Sm, dst_p, tmp_p, op_ilu, &PS, n_p, tol * v_norm(n_p, tmp_p), maxit));     /* :287-289 */
    v_scale_vec(n_u, dst_u, o->diag_D);                                                                   /* :294 */
    v_scale(n_p, dst_p, 1. / o->alpha_asimple);                                                           /* :298 */
    op_csr(&BTm, tmp_u, dst_p);                                                                           /* :304 */
    v_add(n_u, dst_u, -1.0, tmp_u);                                                                       /* :305 */
    v_scale_vec(n_u, dst_u, o->diag_D_inv);                                                               /* :309 */
    free(tmp_u);
    free(tmp_p);
  } else { /* aYosida, Prec:474-517 */
    double *tmp = o->ay_tmp, *tmp2 = o->ay_tmp2;
    double *yu = dup_mem(src_u, (size_t)n_u * sizeof(double));
    double *yp = dup_mem(src_p, (size_t)n_p * sizeof(double));
    v_copy(n_u, tmp, src_u);                /* :491 */
    v_scale_vec(n_u, tmp, o->diag_D_inv);   /* :492 */
    v_copy(n_u, yu, tmp);                   /* :493 */
    op_csr(&Bm, tmp2, tmp);                 /* :496 */
    v_sadd(n_p, yp, -1.0, 1.0, tmp2);       /* yp.sadd(-1.0,tmp2): yp = -yp + tmp2   :497 */
    count_S(pc, cg(op_csr, &Sm, dst_p, yp, op_ilu, &PS, n_p, tol * v_norm(n_p, yp), maxit)); /* :500-502 */
    v_copy(n_p, yp, dst_p);                 /* :504 */
    { /* F->vmult(yu,yu) :507 — Epetra multiplies out of place when source and destination alias [lib] */
      double *t = xmalloc((size_t)n_u * sizeof(double));
      opF(Fm_, t, yu);
      v_copy(n_u, yu, t);
      free(t);
    }
    op_csr(&BTm, tmp, yp);                  /* :510 */
    v_sadd(n_u, yu, -1.0, 1.0, tmp);        /* yu.sadd(-1.0,tmp): yu = -yu + tmp     :511 */
    v_scale_vec(n_u, yu, o->diag_D_inv);    /* :514 */
    v_copy(n_u, dst_u, yu);                 /* :515 */
    free(yu);
    free(yp);
  }
}

void orc_prec_vmult(orc *o, int type, double inner_rtol, int inner_maxiter, double *dst, const double *src, orc_stats *st) {
  prec_ctx pc = {o, type, inner_rtol, inner_maxiter, st};
  prec_vmult(&pc, dst, src);
}

/* NavierStokes::solve_time_step (NS3D:546-640). */
void orc_solve_time_step(orc *o, int type, double tol_abs, double inner_rtol, int maxiter, int inner_maxiter, orc_stats *st) {
  const int n = o->n_u + o->n_p;
  orc_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  v_copy(n, o->prev_sol, o->sol); /* previous_solution = solution  (NS3D:555) */
  double t0 = now_s();
  orc_prec_initialize(o, type);   /* NS3D:568-569 */
  st->t_prec = now_s() - t0;
  t0 = now_s();
  prec_ctx pc = {o, type, inner_rtol, inner_maxiter, st};
  sc_result r = gmres(op_system, o, o->sol_owned, o->rhs, prec_vmult, &pc, n, tol_abs, maxiter); /* NS3D:574 */
  st->t_solve = now_s() - t0;
  st->outer_iterations = r.steps;
  st->final_residual = r.last;
  if (r.status && st->status == 0) st->status = 1;
  v_copy(n, o->sol, o->sol_owned); /* solution = solution_owned  (NS3D:638) */
}

/* NavierStokes::compute_forces (NS3D:744-846 / NS2D:752-859): face loop over boundary id 3 with FEFaceValues.
 * Faces are given as (cell, deal.II local face); face tables as in nsxh_tables rule 1 (weights sum to 1 per face).
 * The geometry (outward normal, face measure) is computed here from the vertex coordinates (cross products), i.e.
 * independently of the J^{-T} n_ref formula the device kernel uses. */
void orc_compute_forces(orc *o, int n_faces, const int32_t *cells, const int32_t *lfaces, int n_qf, const double *N2f,
                        const double *dN2f, const double *N1f, const double *wf, double *drag_out, double *lift_out) {
  const int dim = o->dim, nv = dim + 1, np2 = o->np2, np1 = o->np1;
  static const int TETF[4][3] = {{0, 1, 2}, {1, 0, 3}, {0, 2, 3}, {2, 1, 3}};
  static const int TRIF[3][2] = {{0, 1}, {1, 2}, {2, 0}};
  const double rho = 1.0, nu = o->nu;
  double local_drag = 0.0, local_lift = 0.0;
  for (int f = 0; f < n_faces; ++f) {
    const int cell = cells[f], lf = lfaces[f];
    const double *X = o->cell_coords + (size_t)cell * nv * dim;
    const int32_t *dofs = o->cell_dofs + (size_t)cell * o->dpc;
    double J[3][3] = {{0}}, Ji[3][3] = {{0}}, det;
    for (int d = 0; d < dim; ++d)
      for (int k = 0; k < dim; ++k) J[d][k] = X[(k + 1) * dim + d] - X[d];
    if (dim == 2) {
      det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      Ji[0][0] = J[1][1] / det; Ji[0][1] = -J[0][1] / det; Ji[1][0] = -J[1][0] / det; Ji[1][1] = J[0][0] / det;
    } else {
      det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
            J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
      Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det; Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
      Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
      Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
      Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
    }
    /* outward unit normal and measure of the face from its vertices */
    double nrm[3] = {0, 0, 0}, meas, opp[3] = {0, 0, 0};
    int opp_v = 0;
    if (dim == 3) {
      const double *a = X + 3 * TETF[lf][0], *b = X + 3 * TETF[lf][1], *c = X + 3 * TETF[lf][2];
      const double e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
      nrm[0] = e1[1] * e2[2] - e1[2] * e2[1]; nrm[1] = e1[2] * e2[0] - e1[0] * e2[2]; nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
      const double l = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
      meas = 0.5 * l;
      for (int d = 0; d < 3; ++d) nrm[d] /= l;
      opp_v = 6 - TETF[lf][0] - TETF[lf][1] - TETF[lf][2];
      for (int d = 0; d < 3; ++d) opp[d] = X[3 * opp_v + d] - a[d];
    } else {
      const double *a = X + 2 * TRIF[lf][0], *b = X + 2 * TRIF[lf][1];
      const double e[2] = {b[0] - a[0], b[1] - a[1]};
      meas = sqrt(e[0] * e[0] + e[1] * e[1]);
      nrm[0] = e[1] / meas; nrm[1] = -e[0] / meas;
      opp_v = 3 - TRIF[lf][0] - TRIF[lf][1];
      for (int d = 0; d < 2; ++d) opp[d] = X[2 * opp_v + d] - a[d];
    }
    double s = 0;
    for (int d = 0; d < dim; ++d) s += nrm[d] * opp[d];
    if (s > 0) for (int d = 0; d < dim; ++d) nrm[d] = -nrm[d]; /* outward = away from the opposite vertex */
    for (int q = 0; q < n_qf; ++q) {
      const int tq = lf * n_qf + q;
      const double JxW = wf[q] * meas;
      /* get_function_values(pressure) / get_function_gradients(velocity) on the face (NS3D:795-796) */
      double p = 0.0, G[3][3] = {{0}};
      for (int i = 0; i < o->dpc; ++i) {
        const int c = o->l_comp[i], a = o->l_node[i];
        const double ui = o->sol[dofs[i]];
        if (c == dim) {
          p += ui * N1f[tq * np1 + a];
        } else {
          for (int d = 0; d < dim; ++d) {
            double g = 0;
            for (int k = 0; k < dim; ++k) g += Ji[k][d] * dN2f[(tq * np2 + a) * dim + k];
            G[c][d] += ui * g;
          }
        }
      }
      double n[3] = {-nrm[0], -nrm[1], -nrm[2]}; /* n = -fe_face_values.normal_vector(q) */
      if (dim == 3) { /* NS3D:799-826 */
        const double nx = n[0], ny = n[1];
        const double tangent[3] = {ny, -nx, 0.};
        const double t2 = tangent[0] * tangent[0] + tangent[1] * tangent[1] + tangent[2] * tangent[2];
        double ngt = 0;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) ngt += n[i] * G[i][j] * (tangent[j] / t2);
        local_drag += (rho * nu * ngt * ny - p * nx) * JxW;
        local_lift -= (rho * nu * ngt * nx + p * ny) * JxW;
      } else { /* NS2D:821-838 */
        double forces[2];
        for (int i = 0; i < 2; ++i) {
          double v = 0;
          for (int j = 0; j < 2; ++j) v += (nu * G[i][j] - (i == j ? p : 0.0)) * n[j];
          forces[i] = v * JxW;
        }
        local_drag += forces[0];
        local_lift += forces[1];
      }
    }
  }
  *drag_out = local_drag;
  *lift_out = local_lift;
}
