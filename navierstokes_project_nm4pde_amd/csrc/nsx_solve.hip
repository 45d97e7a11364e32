// nsx_solve.hip — Krylov drivers and the four block preconditioners.
//
//   NavierStokes::solve_time_step          reference Navier-Stokes/src/NavierStokes3D.cpp:546-640
//   PreconditionSIMPLE / aSIMPLE / Yosida / aYosida (initialize + vmult)
//                                           reference Navier-Stokes/include/Preconditioners.hpp:118-217, 220-329, 332-423, 427-534
//   SolverGMRES / SolverCG                  deal.II templates the reference instantiates (NS3D.cpp:554; Prec.hpp:159,180,272,288,372,389,404,501):
//                                           left-preconditioned restarted GMRES (30 temporary vectors => restart 28, modified Gram-Schmidt
//                                           with the Kelley re-orthogonalisation test, stopping on the preconditioned residual) and
//                                           preconditioned CG stopping on the true residual; restated from the published deal.II 9.3-9.5 algorithms.
//
// Host-driven control flow, device-resident vectors and scalars: one host synchronisation per Krylov iteration
// (the Hessenberg column / residual norm), none inside the Gram-Schmidt sweep.
#include <chrono>
#include <cmath>
#include <functional>
#include <memory>

#include "nsx_internal.hpp"

namespace nsx {

void cg_update(nsx_handle *h, int n, double *x, const double *d, double *g, const double *hv, int gh_slot, int dh_slot, int res_slot);
void cg_direction(nsx_handle *h, int n, double *d, const double *hv, int num_slot, int den_slot);
void v_reciprocal(nsx_handle *h, int n, double *d, const double *s, double num);

// ---- pooled temporary vectors (TrilinosWrappers::MPI::Vector temporaries of Prec.hpp:168-171,375-378 and the solvers' own)
struct Tmp {
  nsx_handle *h;
  DevBuf<double> *b;
  Tmp(nsx_handle *h_, size_t n) : h(h_), b(nullptr) {
    for (size_t i = 0; i < h->pool.size(); ++i)
      if (h->pool[i]->n >= n) {
        b = h->pool[i];
        h->pool.erase(h->pool.begin() + i);
        break;
      }
    if (!b) {
      b = new DevBuf<double>();
      b->alloc(n);
      b->zero(h->stream);  // never hand out uninitialised memory (ghost entries are only written by halo exchanges)
    }
  }
  ~Tmp() { h->pool.push_back(b); }
  Tmp(const Tmp &) = delete;
  double *p() const { return b->p; }
};

using Op = std::function<void(double *dst, const double *src)>;

struct SolveResult {
  int status;  // 0 success, 1 failure
  int steps;
  double last;
};

// Distributed runs: hold back the all-reduces of finished reductions for the lifetime of the scope and send them merged when
// it ends (defer_reductions).  On unwinding (an exception between the two calls) nothing is sent any more, but the handle must
// not stay in the deferring state: later solves would consume rank-local sums without an error.
struct DeferScope {
  nsx_handle *h;
  bool active;
  DeferScope(nsx_handle *h_, bool on) : h(h_), active(on) {
    if (active) defer_reductions(h, true);
  }
  void release() {  // normal path: flush (one merged collective)
    if (!active) return;
    active = false;
    defer_reductions(h, false);
  }
  ~DeferScope() {
    if (!active) return;
    h->defer_red = false;
    h->pending_red.clear();
  }
};

static int sc_check(int step, double value, double tol, int maxsteps) {  // SolverControl::check: 0 iterate, 1 success, 2 failure
  if (value <= tol) return 1;
  if (step >= maxsteps || std::isnan(value)) return 2;
  return 0;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr int N_TMP = 30;  // SolverGMRES::AdditionalData::max_n_tmp_vectors
enum { S_H = 8 /* 8..8+N_TMP */, S_NRM = 40, S_H2 = 41 /* re-orthogonalisation coefficients 41..41+N_TMP */, S_DH = 2, S_GH = 3, S_RES = 4, S_GH2 = 5 /* S_RES sits between the two g.h slots: whichever is current, the pair is adjacent */, S_T = 6 };

// SolverGMRES<VectorType>::solve (left preconditioning, default residual).
// zero_new: a freshly created Epetra vector is zero.  That only matters when the preconditioner READS its destination
// (aSIMPLE takes it as the initial guess of its inner solve, Prec.hpp:271), i.e. for the outer solve; the inner solves'
// operators (SpMV, ILU) overwrite every owned entry, so their temporaries are handed out as they come from the pool.
// x_is_zero: the caller has just zeroed x (Prec.hpp:401): the residual b - A x of the first cycle is b without touching the matrix
// (deal.II forms it through vmult + sadd; -1 * (A 0) + b gives the same numbers).
// plain_P: the preconditioner is a plain kernel (ILU), not a solver with host round trips of its own: the operator AND the
// preconditioner of the next iteration are then enqueued behind the Gram-Schmidt sweep, before its coefficients are waited for.
// (Fusing the two into one launch per rank block — four waves for the block's rows of A x, then one wave for the sweeps — was
// tried: 100 us against 35 + 35, the workgroups of the product hold LDS and registers the sweeping waves of other blocks need.)
// P_is_ilu_F: P is the velocity ILU(0) of the last initialisation (the inner solves on F): its triangular solves then run INSIDE the
// launch of the sweep that follows them (v_mgs with ilu_rhs, k_ilu_mgs) where the layout allows it -- the preconditioned vector never
// travels through memory between the two; only the operator of the next iteration is enqueued ahead.
static SolveResult gmres(nsx_handle *h, const Op &A, double *x, const double *b, const Op &P, Span n, int len, double tol, int maxiter,
                         bool zero_new = false, bool x_is_zero = false, bool plain_P = false, bool P_is_ilu_F = false) {
  SolveResult res{1, 0, 0.0};
  std::vector<std::unique_ptr<Tmp>> tmp(N_TMP);
  auto vec = [&](int i) -> double * {
    if (!tmp[i]) {
      tmp[i] = std::make_unique<Tmp>(h, len);
      if (zero_new) v_zero(h, len, tmp[i]->p());
    }
    return tmp[i]->p();
  };
  // distributed runs: the Gram matrix of this solve's basis, kept on the device between its sweeps (mgs_lowsync); one per nesting
  // level (the outer solve's preconditioner runs GMRES solves of its own between two outer sweeps)
  struct Depth {
    nsx_handle *h;
    int d;
    explicit Depth(nsx_handle *h_) : h(h_), d(h_->gmres_depth++) {}
    ~Depth() { h->gmres_depth--; }
  } depth(h);
  double *gram = nullptr;
  if (depth.d < 4) {  // also used on one GPU when a vector is too long for the persistent sweep (v_mgs)
    if (!h->ls_gram.p) {
      h->ls_gram.alloc(4 * 1024);
      h->ls_gram.zero(h->stream);
    }
    gram = h->ls_gram.p + (size_t)depth.d * 1024;
  }
  // (in place needs the lane-owner stream: its kernel loads the right-hand side rows into LDS before it writes anything)
  const bool ilu_conv = P_is_ilu_F && !h->comm && h->schedF.packed_ok && !h->schedF.levelled && h->schedF.stream_ncomp == h->dim &&
                        (getenv("NSX_ILU_MGS") && atoi(getenv("NSX_ILU_MGS")) == 1);  // opt-in: see ilu_mgs_entries (nsx_blas.hip)
  double H[N_TMP][N_TMP - 1];
  double gamma[N_TMP], ci[N_TMP - 1], si[N_TMP - 1], hh[N_TMP + 2], h2[N_TMP + 2];
  int accumulated = 0, state = 0, dim = 0;
  bool re_orth = false;
  double *v = vec(0), *p = vec(N_TMP - 1);
  int ahead = 0;  // what of the next iteration is already enqueued: 0 nothing, 1 the operator (A p), 2 operator and preconditioner
  do {
    ahead = 0;
    if (x_is_zero) {
      P(v, b);  // the residual b - A 0 is b itself: the preconditioner reads it where it is (p is only a temporary)
      x_is_zero = false;
    } else {
      A(p, x);
      v_sadd(h, n, p, -1., 1., b);
      P(v, p);
    }
    v_dot(h, n, v, v, S_NRM);
    double rho = std::sqrt(read_scalar(h, S_NRM));
    res.last = rho;
    state = sc_check(accumulated, rho, tol, maxiter);
    if (state != 0) break;
    gamma[0] = rho;
    v_scale(h, n, v, 1. / rho);
    dim = 0;
    double rho_before = 0.0;  // residual estimate one iteration earlier (0: none yet in this cycle)
    for (int inner = 0; inner < N_TMP - 2 && state == 0; ++inner) {
      ++accumulated;
      static const bool trace = getenv("NSX_TRACE") != nullptr;
      if (trace && h->gmres_depth == 1) fprintf(stderr, "[nsx trace] rank %d: outer iteration %d, residual %.3e\n", h->rank, accumulated, rho);
      double *vv = vec(inner + 1);
      // (already enqueued behind the previous iteration's Gram-Schmidt sweep when `ahead`)
      auto apply_AP = [&](double *dst, const double *src) {  // dst = P (A src)
        A(p, src);
        P(dst, p);
      };
      // ilu_conv: the product A v lands in vv itself and the preconditioner is applied IN PLACE -- by the sweep's own launch (fuse) or,
      // in a re-orthogonalisation cycle, by the separate kernel through p
      const bool fuse = ilu_conv && !re_orth;
      if (ilu_conv) {
        if (ahead == 0) A(vv, vec(inner));
        if (!fuse) {
          v_copy(h, n.n, p, vv);
          P(vv, p);
        }
      } else if (ahead == 0) apply_AP(vv, vec(inner));
      else if (ahead == 1) P(vv, p);
      ahead = 0;
      dim = inner + 1;
      // modified Gram-Schmidt, h(i) = vv . v_i after removing the previous components (add_and_dot chain)
      const bool consider = !re_orth && (inner % 5 == 4);
      double *basis[N_TMP];
      for (int i = 0; i < dim; ++i) basis[i] = vec(i);
      // the sweep can normalise vv itself (vv *= 1./s below) when no second sweep can follow it
      // The sweep leaves the next basis vector complete; A * vv of the NEXT iteration is enqueued right behind it, so the
      // device does not idle while the coefficients travel to the host and the Givens rotations are updated.  Wasted
      // once per solve (the iteration that converges); p is a temporary.
      // ... unless this iteration will probably be the last one: the residual after it is estimated from the last reduction factor
      // (rho * rho / rho_before); a wrong "continues" costs one wasted operator + preconditioner application, a wrong
      // "converges" one exposed host round trip, so the test leans towards not running ahead (NSX_AHEAD_MARGIN, default 2 x tol).
      static const double ahead_margin = getenv("NSX_AHEAD_MARGIN") ? atof(getenv("NSX_AHEAD_MARGIN")) : 2.0;
      const bool likely_last = rho_before > 0.0 && rho * std::min(1.0, rho / rho_before) <= ahead_margin * tol;
      const std::function<void()> next_A = [&]() {
        if (inner + 1 < N_TMP - 2 && !likely_last) {
          static const int ahead_mode = getenv("NSX_AHEAD_MODE") ? atoi(getenv("NSX_AHEAD_MODE")) : 2;
          if (ilu_conv) {  // the next iteration's product goes straight into its own vector; its preconditioner runs inside the next sweep's launch
            A(vec(inner + 2), vv);
            ahead = 1;
          } else if (plain_P && ahead_mode == 2) {  // operator AND preconditioner of the next iteration are plain kernels that depend on vv alone
            apply_AP(vec(inner + 2), vv);
            ahead = 2;
          } else {   // the outer solve's preconditioner runs Krylov solves of its own (host round trips, the same scalar slots): only A
            A(p, vv);
            ahead = 1;
          }
        }
      };
      bool normalized = v_mgs(h, n, vv, dim, basis, S_H, !re_orth, hh, &next_A, consider, gram, fuse ? vv : nullptr);
      if (h->mgs_redo_ahead) {  // the sweep fell back to the launch-per-link chain: A * vv was enqueued on an unfinished vv
        h->mgs_redo_ahead = false;
        ahead = 0;
      }
      double s = std::sqrt(hh[dim]);
      if (consider) {
        const double norm_vv_start = std::sqrt(hh[dim + 1]);  // |vv| before the sweep, computed inside it
        if (!(s > 10. * norm_vv_start * std::sqrt(2.220446049250313e-16))) re_orth = true;
      }
      if (re_orth) {
        normalized = v_mgs(h, n, vv, dim, basis, S_H2, true, h2, nullptr, false, gram);
        for (int i = 0; i < dim; ++i) hh[i] += h2[i];
        s = std::sqrt(h2[dim]);
      }
      hh[inner + 1] = s;
      if (s != 0 && !normalized) v_scale(h, n, vv, 1. / s);
      // givens_rotation(h, gamma, ci, si, inner)
      for (int i = 0; i < inner; ++i) {
        const double sn = si[i], cs = ci[i], dummy = hh[i];
        hh[i] = cs * dummy + sn * hh[i + 1];
        hh[i + 1] = -sn * dummy + cs * hh[i + 1];
      }
      const double r = 1. / std::sqrt(hh[inner] * hh[inner] + hh[inner + 1] * hh[inner + 1]);
      si[inner] = hh[inner + 1] * r;
      ci[inner] = hh[inner] * r;
      hh[inner] = ci[inner] * hh[inner] + si[inner] * hh[inner + 1];
      gamma[inner + 1] = -si[inner] * gamma[inner];
      gamma[inner] *= ci[inner];
      for (int i = 0; i < dim; ++i) H[i][inner] = hh[i];
      rho_before = rho;
      rho = std::fabs(gamma[dim]);
      res.last = rho;
      state = sc_check(accumulated, rho, tol, maxiter);
    }
    double y[N_TMP];
    for (int i = dim - 1; i >= 0; --i) {  // H1.backward(h, gamma)
      double s = gamma[i];
      for (int j = i + 1; j < dim; ++j) s -= H[i][j] * y[j];
      y[i] = s / H[i][i];
    }
    double *vs[N_TMP];
    for (int i = 0; i < dim; ++i) vs[i] = vec(i);
    v_axpy_multi(h, n, x, dim, vs, y);
  } while (state == 0);
  res.status = state == 1 ? 0 : 1;
  res.steps = accumulated;
  return res;
}

// SolverCG<VectorType>::solve with a preconditioner.
// Pdot (optional): dst = P src AND scal[slot] = src . dst in the same launch; returns false if it only applied P.
using OpDot = std::function<bool(double *dst, const double *src, int slot)>;
static SolveResult cg(nsx_handle *h, const Op &A, double *x, const double *b, const Op &P, Span n, int len, double tol, int maxiter,
                      const OpDot *Pdot = nullptr) {
  auto apply_P_dot = [&](double *dst, const double *src, int slot) {  // h = P g ; slot = g . h
    if (Pdot && (*Pdot)(dst, src, slot)) return;
    if (!Pdot) P(dst, src);
    v_dot(h, n, src, dst, slot);
  };
  SolveResult res{1, 0, 0.0};
  Tmp g(h, len), d(h, len), hv(h, len);
  int it = 0;
  // g = A x - b.  deal.II short-cuts to g = -b when x.all_zero(); A*0 - b gives the identical vector, so no device-side test is needed.
  A(g.p(), x);
  v_add(h, n, g.p(), -1., b);
  v_dot(h, n, g.p(), g.p(), S_RES);
  double r = std::sqrt(read_scalar(h, S_RES));
  res.last = r;
  int conv = sc_check(0, r, tol, maxiter);
  if (conv == 0) {
    int gh = S_GH, gh_new = S_GH2;  // ping-pong slots for (g.h) of the current / next iteration
    apply_P_dot(hv.p(), g.p(), gh);
    v_copy(h, n.n, d.p(), hv.p());
    v_scale(h, n, d.p(), -1.);
    while (conv == 0) {
      it++;
      A(hv.p(), d.p());
      v_dot(h, n, d.p(), hv.p(), S_DH);
      // Distributed run: |g|^2 and the g.h of the next iteration are independent sums: their all-reduces are held back and
      // go out as ONE collective over the two adjacent slots (a latency, not a bandwidth, matter: 8 B or 16 KB cost the same).
      const bool batch = h->comm != nullptr;
      DeferScope defer(h, batch);
      cg_update(h, n.n, x, d.p(), g.p(), hv.p(), gh, S_DH, S_RES);  // alpha = gh / (d.h); x += alpha d; g += alpha h; res = |g|
      // The residual travels to the host while the GPU already applies the preconditioner of the NEXT iteration
      // (h = P g and g.h only touch temporaries): the host round trip hides behind that kernel instead of idling the
      // device; the work is wasted once per solve, in the iteration that converges.
      unsigned long long seq = 0;
      if (!batch) seq = publish_scalars(h, S_RES, 1);
      apply_P_dot(hv.p(), g.p(), gh_new);
      if (batch) {
        defer.release();  // flushes: one all-reduce for S_RES and gh_new
        seq = publish_scalars(h, S_RES, 1);
      }
      double res2;
      collect_published(h, seq, S_RES, 1, &res2);
      r = std::sqrt(std::fabs(res2));
      res.last = r;
      conv = sc_check(it, r, tol, maxiter);
      if (conv != 0) break;
      cg_direction(h, n.n, d.p(), hv.p(), gh_new, gh);  // beta = gh_new / gh_old ; d = beta d - h
      std::swap(gh, gh_new);
    }
  }
  res.status = conv == 1 ? 0 : 1;
  res.steps = it;
  return res;
}

static double norm2(nsx_handle *h, Span n, const double *v) {
  v_dot(h, n, v, v, S_T);
  return std::sqrt(read_scalar(h, S_T));
}

// ------------------------------------------------------------------ preconditioners
__global__ void k_same_and_keep(int n, const double *__restrict__ w, double *__restrict__ prev, int *changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long a = (unsigned long long)__double_as_longlong(w[i]), b = (unsigned long long)__double_as_longlong(prev[i]);
  if (a != b) {
    prev[i] = w[i];
    __hip_atomic_store(changed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// Are the inputs of the Schur product the ones its current values were computed from?  (Bitwise comparison of the weight vector
// on the device; the answer comes back through a mapped word behind one stream synchronisation.)
static bool schur_inputs_unchanged(nsx_handle *h, int type) {
  const bool cache = !(getenv("NSX_SCHUR_CACHE") && atoi(getenv("NSX_SCHUR_CACHE")) == 0);  // read per call: bench.py times both schedules on one handle
  const int n = h->len_u;  // owned + ghost weights
  volatile int *flag = (volatile int *)(h->pub_host + N_SLOTS + 4);
  const bool had = h->schur_valid && h->schur_type == type && (int)h->schur_w_prev.n == n;
  if (!had) {
    h->schur_w_prev.alloc(n);
    comm_halo_u(h, h->schur_w.p);  // the comparison below covers what schur_numeric reads, ghosts included
    v_copy(h, n, h->schur_w_prev.p, h->schur_w.p);
    return false;  // schur_valid is set by the caller once the rebuild has gone through
  }
  comm_halo_u(h, h->schur_w.p);
  *flag = 0;
  hipLaunchKernelGGL(k_same_and_keep, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, n, h->schur_w.p, h->schur_w_prev.p, (int *)(h->pub_dev + N_SLOTS + 4));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  // a distributed run rebuilds every time: the ranks would have to agree on the answer first (the rebuild contains collectives)
  return cache && !h->comm && *flag == 0;
}

void prec_initialize(nsx_handle *h, int type) {
  ensure_schedules(h);
  if (!h->assembled) NSX_THROW(NSX_ERR_ARG, "assemble before initialising a preconditioner");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int n_u = h->n_u;
  const double *V = nullptr;
  if (type == NSX_PREC_YOSIDA) {  // Prec.hpp:350-355: D = diag(mass_matrix) (= M/dt)
    extract_diag(h, h->gA, h->vMass.p, h->diag_D.p);
  } else if (type == NSX_PREC_SIMPLE || type == NSX_PREC_ASIMPLE || type == NSX_PREC_AYOSIDA) {  // Prec.hpp:135-140,239-245,447-452
    extract_diag(h, h->gA, h->vF.p, h->diag_D.p);
  } else {
    NSX_THROW(NSX_ERR_ARG, "Invalid preconditioner type");  // std::runtime_error of NS3D.cpp:633
  }
  // diag_D_inv = 1/D, neg_diag_D_inv = -1/D
  v_reciprocal(h, n_u, h->diag_D_inv.p, h->diag_D.p, 1.0);
  v_reciprocal(h, n_u, h->neg_diag_D_inv.p, h->diag_D.p, -1.0);
  V = h->neg_diag_D_inv.p;
  if (type == NSX_PREC_AYOSIDA) {  // Prec.hpp:456-465: lump_M = -1 / sum_j |M_ij|
    abs_rowsum(h, h->gA, h->vMass.p, h->lump_M.p);
    v_reciprocal(h, n_u, h->lump_M.p, h->lump_M.p, -1.0);
    V = h->lump_M.p;
  }
  // negative_S = B * diag(V) * B_T with B_T = -B^T (Dirichlet rows cleared): weights w = -V * mask
  v_copy(h, n_u, h->schur_w.p, V);
  v_scale_vec(h, n_u, h->schur_w.p, h->dirmask.p);
  v_scale(h, n_u, h->schur_w.p, -1.0);
  // preconditioner_F.initialize(*F)   (Prec.hpp:147,250,361,470): F changes every step
  ilu_factor(h, h->gA, h->schedF, h->vF.p, h->luF.p, "ilu_factor_F");
  // negative_S and preconditioner_S.initialize(negative_S)   (Prec.hpp:144-148,248-251,358-362,468-471).  The reference
  // rebuilds both in every step.  Their only inputs are block(1,0) (assembled once) and the weights w; when w is bit for bit
  // the vector of the previous initialisation (Yosida: D = diag(M / deltat) and the Dirichlet mask do not change in time) the
  // product, its ILU(0) factors and the block inverses would come out bit for bit the same, and are kept.
  // The kept values count as valid only once a rebuild has run to its end AND its factorisation reported no failure: a failed or
  // interrupted rebuild must not be reused by the next initialisation with the same weights.  (k_same_and_keep has already
  // overwritten schur_w_prev, so validity cannot be derived from the weights alone.)
  if (!schur_inputs_unchanged(h, type)) {
    h->schur_valid = false;
    h->schur_pending = true;  // confirmed by prec_confirm() behind the caller's synchronisation + ilu_check
    schur_numeric(h, h->schur_w.p);
    cg_pack_values(h);
    ilu_factor(h, h->gS, h->schedS, h->vSchur.p, h->luS.p, "ilu_factor_S");
    h->schur_type = type;
  }
  h->prec_ready = true;
}

// behind the stream synchronisation that follows prec_initialize: factorisation failures surface here (ilu_check throws), and only
// a rebuild that got this far may be kept for later initialisations
void prec_confirm(nsx_handle *h) {
  try {
    ilu_check(h);
  } catch (...) {
    h->schur_valid = false;
    h->schur_pending = false;
    h->prec_ready = false;
    throw;
  }
  if (h->schur_pending) {
    h->schur_pending = false;
    h->schur_valid = true;
  }
}

static void count(nsx_solve_stats *st, bool F, const SolveResult &r) {
  if (!st) return;
  if (F) {
    st->inner_F_iterations += r.steps;
    st->n_F_solves++;
  } else {
    st->inner_S_iterations += r.steps;
    st->n_S_solves++;
  }
  if (r.status) st->status = 2;
}

void prec_vmult(nsx_handle *h, int type, double tol, int maxit, double *dst, const double *src, nsx_solve_stats *st) {
  if (!h->prec_ready) NSX_THROW(NSX_ERR_ARG, "preconditioner not initialised");
  const int n_u = h->n_u, n_p = h->n_p, dim = h->dim, len_u = h->len_u, len_p = h->len_p;
  const double *src_u = src, *src_p = src + h->off_p;
  double *dst_u = dst, *dst_p = dst + h->off_p;
  Op Fm = [h](double *d, const double *s) { spmv_F(h, h->vF.p, s, d); };
  Op Sm = [h](double *d, const double *s) { spmv_S(h, s, d); };
  Op PF = [h, dim](double *d, const double *s) { ilu_solve(h, h->gA, h->schedF, h->luF.p, s, d, dim, "ilu_solve_F"); };
  Op PS = [h](double *d, const double *s) { ilu_solve(h, h->gS, h->schedS, h->luS.p, s, d, 1, "ilu_solve_S"); };
  OpDot PSdot = [h](double *d, const double *s, int slot) { return ilu_solve(h, h->gS, h->schedS, h->luS.p, s, d, 1, "ilu_solve_S", slot); };
  // SolverCG on negative_S_tilde with tolerance tol * |b| (Prec.hpp:179-182,388-390,500-502): one persistent launch where the
  // layout allows it (nsx_cg.hip), else the launch-per-operation solver above
  auto cg_S = [&](double *x, const double *b) {
    int steps = 0, status = 0;
    double last = 0.0;
    if (cg_schur_persistent(h, x, b, tol, maxit, &steps, &last, &status)) return SolveResult{status, steps, last};
    if (cg_schur_fused(h, x, b, tol, maxit, &steps, &last, &status)) return SolveResult{status, steps, last};  // distributed, or too many blocks for a resident grid: two launches per iteration
    h->cg_last_path = 1;
    return cg(h, Sm, x, b, PS, n_p, len_p, tol * norm2(h, n_p, b), maxit, &PSdot);
  };

  if (type == NSX_PREC_YOSIDA) {  // Prec.hpp:365-408
    Tmp yu(h, len_u), yp(h, len_p), tmp(h, len_p), tmp2(h, len_u), res(h, len_u);
    v_copy(h, n_u, yu.p(), src_u);                                                            // :375
    v_copy(h, n_p, yp.p(), src_p);                                                            // :376
    count(st, true, gmres(h, Fm, yu.p(), src_u, PF, n_u, len_u, tol * norm2(h, n_u, src_u), maxit, false, false, true, true)); // :371-382
    spmv_B(h, yu.p(), tmp.p());                                                               // :385
    v_add(h, n_p, tmp.p(), -1.0, src_p);                                                      // :386
    count(st, false, cg_S(yp.p(), tmp.p()));                                                  // :388-390
    v_copy(h, n_p, dst_p, yp.p());                                                            // :394
    spmv_G(h, dst_p, tmp2.p(), false);                                                        // :398
    v_zero(h, n_u, res.p());                                                                  // :401
    v_copy(h, n_u, dst_u, yu.p());                                                            // :402
    count(st, true, gmres(h, Fm, res.p(), tmp2.p(), PF, n_u, len_u, tol * norm2(h, n_u, tmp2.p()), maxit, false, true, true, true));  // :403-405
    v_sadd(h, n_u, dst_u, -1., 1., res.p());  // dst.block(0).sadd(-1,res): dst = -dst + res            :406
  } else if (type == NSX_PREC_SIMPLE) {  // Prec.hpp:151-205
    Tmp sol1_u(h, len_u), sol1_p(h, len_p), temp_1(h, len_p), tmp(h, len_u);
    v_copy(h, n_u, sol1_u.p(), src_u);                                                             // :168
    v_copy(h, n_p, sol1_p.p(), src_p);                                                             // :169
    count(st, true, gmres(h, Fm, sol1_u.p(), src_u, PF, n_u, len_u, tol * norm2(h, n_u, src_u), maxit, false, false, true, true));  // :157-173
    spmv_B(h, sol1_u.p(), temp_1.p());                                                             // :175
    v_add(h, n_p, temp_1.p(), -1.0, src_p);                                                        // :176
    count(st, false, cg_S(sol1_p.p(), temp_1.p()));                                               // :179-182
    v_copy(h, n_p, dst_p, sol1_p.p());                                                             // :194
    v_scale(h, n_p, dst_p, 1. / 0.5);                                                              // :195, alpha = 0.5 (:207)
    v_copy(h, n_u, dst_u, sol1_u.p());                                                             // :199
    spmv_G(h, dst_p, tmp.p(), false);                                                              // :201
    v_scale_vec(h, n_u, tmp.p(), h->diag_D_inv.p);                                                 // :202
    v_add(h, n_u, dst_u, -1.0, tmp.p());                                                           // :203
  } else if (type == NSX_PREC_ASIMPLE) {  // Prec.hpp:254-311 (dst is the caller's vector: its content is the initial guess)
    Tmp tmp_u(h, len_u), tmp_p(h, len_p);
    count(st, true, gmres(h, Fm, dst_u, src_u, PF, n_u, len_u, tol * norm2(h, n_u, src_u), maxit, false, false, true, true));  // :271-273
    spmv_B(h, dst_u, dst_p);                                                                   // :280
    v_sadd(h, n_p, dst_p, -1.0, 1.0, src_p);                                                   // :281
    v_copy(h, n_p, tmp_p.p(), dst_p);                                                          // :282
    count(st, false, gmres(h, Sm, dst_p, tmp_p.p(), PS, n_p, len_p, tol * norm2(h, n_p, tmp_p.p()), maxit));  // :287-289
    v_scale_vec(h, n_u, dst_u, h->diag_D.p);                                                   // :294
    v_scale(h, n_p, dst_p, 1. / 1.0);                                                          // :298, alpha = 1 (:328)
    spmv_G(h, dst_p, tmp_u.p(), false);                                                        // :304
    v_add(h, n_u, dst_u, -1.0, tmp_u.p());                                                     // :305
    v_scale_vec(h, n_u, dst_u, h->diag_D_inv.p);                                               // :309
  } else if (type == NSX_PREC_AYOSIDA) {  // Prec.hpp:474-517
    Tmp tmp(h, len_u), tmp2(h, len_p), yu(h, len_u), yp(h, len_p), t(h, len_u);
    v_copy(h, n_p, yp.p(), src_p);                    // :487
    v_copy(h, n_u, tmp.p(), src_u);                   // :491
    v_scale_vec(h, n_u, tmp.p(), h->diag_D_inv.p);    // :492
    v_copy(h, n_u, yu.p(), tmp.p());                  // :493
    spmv_B(h, tmp.p(), tmp2.p());                     // :496
    v_sadd(h, n_p, yp.p(), -1.0, 1.0, tmp2.p());      // :497
    count(st, false, cg_S(dst_p, yp.p()));          // :500-502
    v_copy(h, n_p, yp.p(), dst_p);                    // :504
    spmv_F(h, h->vF.p, yu.p(), t.p());                // :507 F->vmult(yu,yu): Epetra multiplies out of place when the arguments alias
    v_copy(h, n_u, yu.p(), t.p());
    spmv_G(h, yp.p(), tmp.p(), false);                // :510
    v_sadd(h, n_u, yu.p(), -1.0, 1.0, tmp.p());       // :511
    v_scale_vec(h, n_u, yu.p(), h->diag_D_inv.p);     // :514
    v_copy(h, n_u, dst_u, yu.p());                    // :515
  } else {
    NSX_THROW(NSX_ERR_ARG, "Invalid preconditioner type");
  }
}

void solve_time_step(nsx_handle *h, int type, double tol, double inner_rtol, int maxiter, int inner_maxiter, nsx_solve_stats *st) {
  HIP_CHECK(hipSetDevice(h->prm.device));
  const Span n = blk_span(h);
  nsx_solve_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof(*st));
  v_copy(h, h->len_blk, h->prev_sol.p, h->sol.p);  // previous_solution = solution (NS3D.cpp:555)
  HIP_CHECK(hipStreamSynchronize(h->stream));
  double t0 = now_s();
  h->defer_red = false;  // a solve that unwound in the middle of a batched reduction must not leave the handle deferring
  h->pending_red.clear();
  static const bool trace = getenv("NSX_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[nsx trace] rank %d: solve_time_step: preconditioner set-up\n", h->rank);
  prec_initialize(h, type);  // NS3D.cpp:568-569
  HIP_CHECK(hipStreamSynchronize(h->stream));
  prec_confirm(h);
  if (trace) fprintf(stderr, "[nsx trace] rank %d: solve_time_step: outer solve\n", h->rank);
  st->t_prec = now_s() - t0;
  t0 = now_s();
  Op A = [h](double *d, const double *s) { spmv_saddle(h, s, d); };
  Op P = [h, type, inner_rtol, inner_maxiter, st](double *d, const double *s) { prec_vmult(h, type, inner_rtol, inner_maxiter, d, s, st); };
  SolveResult r = gmres(h, A, h->sol_owned.p, h->rhs.p, P, n, h->len_blk, tol, maxiter, true);  // NS3D.cpp:574
  v_copy(h, h->len_blk, h->sol.p, h->sol_owned.p);  // solution = solution_owned (NS3D.cpp:638): copy + ghost import
  comm_halo_u(h, h->sol.p);
  comm_halo_p(h, h->sol.p + h->off_p);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  ilu_check(h);  // a triangular-solve kernel that refused to run (nsx_sparse.hip: k_ilu_solve_lanes) fails the call
  st->t_solve = now_s() - t0;
  st->outer_iterations = r.steps;
  st->final_residual = r.last;
  st->persistent_fallbacks = h->n_persistent_fallbacks;
  if (r.status && st->status == 0) st->status = 1;
}

// small element-wise helper used only at initialize time
__global__ void k_reciprocal(int n, double *__restrict__ d, const double *__restrict__ s, double num) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = num / s[i];
}
void v_reciprocal(nsx_handle *h, int n, double *d, const double *s, double num) {
  hipLaunchKernelGGL(k_reciprocal, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, n, d, s, num);
}

}  // namespace nsx

#define NSX_API_BODY(h_, ...)                  \
  if (!(h_)) return NSX_ERR_ARG;               \
  try {                                        \
    __VA_ARGS__;                               \
  } catch (const nsx::Error &e) {              \
    (h_)->err = e.msg;                         \
    return e.code;                             \
  } catch (const std::exception &e) {          \
    (h_)->err = e.what();                      \
    return NSX_ERR_ARG;                        \
  }                                            \
  return NSX_OK;

extern "C" {

int nsx_solve_time_step(nsx_handle *h, int prec_type, double tol_abs, double inner_rtol, int maxiter, int inner_maxiter,
                        nsx_solve_stats *stats) {
  NSX_API_BODY(h, {
    if (!h->assembled) NSX_THROW(NSX_ERR_ARG, "assemble before solving");
    nsx::solve_time_step(h, prec_type, tol_abs, inner_rtol, maxiter, inner_maxiter, stats);
    if (stats && stats->status) {
      h->err = stats->status == 1 ? "outer GMRES did not converge (SolverControl::NoConvergence)" : "an inner solve did not converge";
      return NSX_ERR_NOCONV;
    }
  })
}

int nsx_prec_initialize(nsx_handle *h, int prec_type) {
  NSX_API_BODY(h, {
    nsx::prec_initialize(h, prec_type);
    HIP_CHECK(hipStreamSynchronize(h->stream));
    nsx::prec_confirm(h);
  })
}

int nsx_prec_vmult(nsx_handle *h, int prec_type, double inner_rtol, int inner_maxiter, double *dst, const double *src, nsx_solve_stats *stats) {
  NSX_API_BODY(h, {
    if (!dst || !src) NSX_THROW(NSX_ERR_ARG, "null vector");
    if (h->dist) NSX_THROW(NSX_ERR_UNSUPPORTED, "nsx_prec_vmult with host vectors works on a single-process handle only");
    HIP_CHECK(hipSetDevice(h->prm.device));
    const int n = h->n_u + h->n_p;
    nsx::Tmp d(h, n), s(h, n);
    nsx::vec_from_caller(h, s.p(), src, false);
    nsx::vec_from_caller(h, d.p(), dst, false);  // aSIMPLE reads dst as initial guess
    if (stats) memset(stats, 0, sizeof(*stats));
    h->defer_red = false;
    h->pending_red.clear();
    nsx::prec_vmult(h, prec_type, inner_rtol, inner_maxiter, d.p(), s.p(), stats);
    if (stats) stats->persistent_fallbacks = h->n_persistent_fallbacks;
    nsx::vec_to_caller(h, d.p(), dst);
    nsx::ilu_check(h);
  })
}

int nsx_system_vmult(nsx_handle *h, double *dst, const double *src) {
  NSX_API_BODY(h, {
    if (!dst || !src || !h->assembled) NSX_THROW(NSX_ERR_ARG, "null vector / nothing assembled");
    HIP_CHECK(hipSetDevice(h->prm.device));
    nsx::Tmp d(h, h->len_blk), s(h, h->len_blk);
    // host vectors are globally indexed: pick the owned entries (ghosts come through the halo exchange of the product)
    nsx::vec_from_caller(h, s.p(), src, false);
    nsx::spmv_saddle(h, s.p(), d.p());
    nsx::vec_to_caller(h, d.p(), dst);
  })
}

int nsx_ilu_apply(nsx_handle *h, int which, double *dst, const double *src) {
  NSX_API_BODY(h, {
    if (!dst || !src || which < 0 || which > 1) NSX_THROW(NSX_ERR_ARG, "bad arguments");
    if (!h->prec_ready) NSX_THROW(NSX_ERR_ARG, "no factors: call nsx_prec_initialize first");
    if (h->dist) NSX_THROW(NSX_ERR_UNSUPPORTED, "nsx_ilu_apply with host vectors works on a single-process handle only");
    HIP_CHECK(hipSetDevice(h->prm.device));
    const int n = which == 0 ? h->n_u : h->n_p;
    nsx::Tmp d(h, n), s(h, n);
    nsx::part_from_caller(h, which, s.p(), src);
    if (which == 0) nsx::ilu_solve(h, h->gA, h->schedF, h->luF.p, s.p(), d.p(), h->dim, "ilu_solve_F");
    else nsx::ilu_solve(h, h->gS, h->schedS, h->luS.p, s.p(), d.p(), 1, "ilu_solve_S");
    nsx::part_to_caller(h, which, d.p(), dst);
    nsx::ilu_check(h);
  })
}

}  // extern "C"
