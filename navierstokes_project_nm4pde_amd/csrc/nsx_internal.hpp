// nsx_internal.hpp — private state of a libnsx handle (C-ABI: include/nsx.h).
//
// HBM layout (all FP64 values, int32 indices; N2 = scalar P2 nodes, NP = P1 nodes, dim components interleaved):
//   vectors        [n_u + n_p] = [N2][dim] velocity (node-major, components consecutive) then [NP] pressure
//   A-graph        scalar P2 x P2 CSR (rowptr/colind/diag position); value arrays on it:
//                  S0 = M/dt + nu K (static), Mass = M/dt, Stiff = nu K, Conv = C(u_n), F = system(0,0), LU_F
//                  -> the reference stores dim^2 x as many entries (all component couplings, NS3D.cpp:109-119);
//                     every velocity-velocity term is delta_cd (x) scalar, so one scalar operator serves dim components.
//   G-graph        P2 x P1 CSR, dim values per entry: block (0,1) = -int psi_k d_c phi_i   (NS3D.cpp:258)
//   B-graph        P1 x P2 CSR, dim values per entry: block (1,0) = +int psi_k d_c phi_j   (NS3D.cpp:261)
//   S-graph        P1 x P1 CSR = structural product B*G: negative_S_tilde and its ILU(0)   (Prec.hpp:144,358)
//   cell tables    SoA: cell_n2[a][cell], cell_n1[v][cell], geo[k][cell] (J^-1 row-major, then |det J|)
//   gather maps    per CSR entry the list of (local entry, cell) contributions -> deterministic assembly, no atomics
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <functional>
#include <vector>

#include "../../include/nsx.h"
#include "../host/graph.hpp"

// Cache policy (compile time, -DNSX_NT=mask): non-temporal loads for 1 = the Krylov basis in the Gram-Schmidt sweep,
// 2 = the matrix stream of the LDS-staged SpMV, 4 = the factor stream of the packed triangular solve.  An inner GMRES
// iteration on F streams F (126 MB), the ILU factors (130 MB) and the basis (17 - 125 MB) once each: more than the 256-MiB
// Infinity Cache holds, so with default-policy loads every stream evicts the next one's lines.  Measured per mask
// (tools/nt_sweep.sh, us per launch SpMV / triangular solve / sweep): 0: 34.2 / 34.4 / 35.2; 1: 30.6 / 32.8 / 35.7;
// 4: 30.7 / 35.1 / 32.8; 2: 39.5 / 33.4 / 33.0 (the 2-byte index stream does not like nt); 6, 7: worse.
#ifndef NSX_NT
#define NSX_NT 1
#endif
template <int BIT, class T>
__device__ __forceinline__ T ld_stream(const T *p) {
  if constexpr ((NSX_NT & BIT) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}

namespace nsx {

struct Error {
  int code;
  std::string msg;
};

#define NSX_THROW(code_, ...)                              \
  do {                                                     \
    char buf_[512];                                        \
    snprintf(buf_, sizeof(buf_), __VA_ARGS__);             \
    throw ::nsx::Error{(code_), std::string(buf_)};        \
  } while (0)

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) NSX_THROW(NSX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    if (count == n && p) return;
    release();
    n = count;
    HIP_CHECK(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
  }
  void zero(hipStream_t s) {
    if (n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), s));
  }
  void upload(const T *src, size_t count, hipStream_t s) {
    alloc(count);
    if (count) HIP_CHECK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
  void upload(const std::vector<T> &v, hipStream_t s) { upload(v.data(), v.size(), s); }
  void download(T *dst, size_t count, hipStream_t s) const {
    if (count) HIP_CHECK(hipMemcpyAsync(dst, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
};

// CSR graph resident on the device (+ host copy kept for exports / setup products)
struct DevCsr {
  Csr host;
  DevBuf<int32_t> rowptr, colind, diag;  // diag: position of (i,i) (square graphs only)
  int32_t n_rows() const { return host.n_rows; }
  int64_t nnz() const { return host.nnz(); }
};

// Deterministic gather map: out entry e sums src[ptr[e] .. ptr[e+1]) (buffer offsets, ascending cell order).
struct GatherMap {
  DevBuf<int32_t> ptr, src;
  int64_t n_out = 0, n_src = 0;
};

// Block-Jacobi ILU(0) schedule on a square graph: per block the rows sorted by dependency level.
struct IluSchedule {
  int n_blocks = 0, max_rows = 0;
  std::vector<int32_t> block_ptr_h;
  DevBuf<int32_t> block_ptr;             // [n_blocks+1] row ranges
  DevBuf<int32_t> fwd_lvl_ptr, fwd_rows; // per block: levels of the L solve / factorisation (global arrays with offsets)
  DevBuf<int32_t> bwd_lvl_ptr, bwd_rows; // per block: levels of the U solve
  DevBuf<int32_t> blk_lvl_off;           // [n_blocks+1] offsets into fwd_lvl_ptr (levels per block), same for bwd
  DevBuf<int32_t> blk_lvl_off_b;
  int max_levels = 0;
  // packed solve stream (host/ilu_stream.hpp, k_ilu_solve_lanes): one wave per group of blocks, a lane owns a row for as many ticks
  // as the row has in-block entries; slab t = tick t of the wave
  DevBuf<int32_t> in_cptr;      // [n_rows+1] running count of in-block entries (compact numbering of a block's factor)
  DevBuf<int32_t> in_cpos;      // [in-block entries] compact number -> CSR position
  DevBuf<int32_t> fac_order;    // [n_blocks] blocks by descending in-block entries (dispatch order of k_ilu_factor_lds)
  int32_t max_block_nnz = 0;    // in-block entries of the largest block
  int64_t in_block_nnz = 0;     // in-block entries of the factor, diagonal included: what one application has to read (algorithmic bytes)
  int stream_ncomp = 0;         // the stream holds LDS byte addresses: it is built for one number of interleaved right-hand sides
  int stream_epl = 1;           // entries of its row a lane takes per tick (NSX_ILU_EPT)
  int64_t n_slabs = 0;
  int blocks_per_wave = 1, n_waves = 0, max_wave_rows = 0;
  bool packed_ok = false;       // false: a wave's rows do not fit 16-bit LDS addresses (few large blocks): workgroup-per-block kernel instead
  DevBuf<int32_t> pk_row_ptr, pk_rows;  // the rows of every wave in LDS order ([n_waves+1] offsets, global row ids)
  DevBuf<int32_t> pk_dinv_slot;         // position of row i's inverse pivot in pk_dinv (wave order)
  DevBuf<int32_t> pk_slab_ptr;  // [2*n_waves+1]: forward slabs, then backward slabs, per wave
  DevBuf<int32_t> pk_meta;      // [(n_slabs + pad) * 64 * meta words]
  DevBuf<int32_t> pk_slot_of;   // [nnz]: position of every in-block off-diagonal CSR entry in pk_val, -1 otherwise
  DevBuf<double> pk_val;        // [(n_slabs + pad) * 64 * stream_epl]: -L and -U/d in stream order (unused slots stay 0)
  DevBuf<double> pk_dinv;       // [n_rows] inverse pivots in wave order
  // explicit inverses (k_ilu_invert / k_ilu_apply_dense): P_b = (L D U)^-1 of every block as a dense row-major n_b x n_b
  // matrix; the triangular solves become one dependency-free dense product per block
  bool dense = false;
  DevBuf<int64_t> dn_off;       // [n_blocks+1] offsets of the blocks' matrices in dn_P
  DevBuf<double> dn_P;
  int64_t dn_entries = 0;
  // levelled path for blocks too large for one wave / one workgroup (few real MPI ranks: R = 1 is the serial reference,
  // R = 8 one rank per GPU): the dependency levels of ALL blocks merged (blocks are independent, so level l of the
  // schedule is the union of the blocks' level-l rows), one launch per level over all its rows.  in_lo/in_hi: CSR
  // positions bounding the in-block entries of every row (columns are sorted, so they are one contiguous range).
  bool levelled = false;
  std::vector<int32_t> gl_f_ptr_h, gl_b_ptr_h;  // [levels+1] offsets into gl_f_rows / gl_b_rows
  DevBuf<int32_t> gl_f_rows, gl_b_rows, in_lo, in_hi;
  DevBuf<int32_t> gl_f_rec, gl_b_rec;  // [4 * rows] per row in level order: row, first entry, end of entries, diagonal (k_ilu_solve_level)
};

// Ghost exchange plan of one scalar space (the Epetra_Import of every vmult): neighbours in ascending rank order,
// what to pack for each, and where each neighbour's values land in the ghost part of a vector.
struct HaloPlan {
  std::vector<int> nbr;
  std::vector<int32_t> send_ptr, recv_ptr;  // [n_nbr+1], node units
  std::vector<int32_t> send_idx_in;         // owned nodes to pack, caller-local ids (the internal layout maps them)
  DevBuf<int32_t> send_idx;                 // local (owned) node ids to pack, internal numbering
  DevBuf<double> sendbuf;
  std::vector<double> h_send, h_recv;       // host staging for the callback backend
  int n_own = 0;                            // ghosts start at node n_own
  hipEvent_t ev_done = nullptr;             // recorded on the communication stream when the ghosts of the last exchange are in place
  bool in_flight = false;                   // callback backend: packed and copied out, the host exchange itself still to do
  bool self_test = false;                   // nsx_comm_self_halo_test: a 1-rank communicator whose only neighbour is the rank itself
};

// Rows of a distributed operator that can be computed before the halo of its input arrives (all their columns are owned)
// and those that cannot (at least one ghost column): the interior rows run while the exchange is in flight.
struct RowSplit {
  DevBuf<int32_t> interior, interface;
  int n_interior = 0, n_interface = 0;
};

// LDS-staged ("blocked") SpMV schedule on a CSR graph: rows are cut into chunks of consecutive rows; per chunk the
// sorted unique column list and, per entry, the 16-bit position of its column in that list.  A workgroup stages the
// chunk's x entries in LDS once (each x entry is gathered once per chunk instead of once per non-zero).
struct SpmvBlocked {
  int n_chunks = 0, max_ucols = 0, max_rows = 0;
  DevBuf<int32_t> crow;  // [n_chunks+1] row range of every chunk (unions of consecutive rank blocks, or a fixed row count)
  DevBuf<int32_t> cptr, ucols;
  DevBuf<uint16_t> lidx;
  double ucols_total = 0;
  // launch order: desc[4 * block] = {first staged column, staged columns, first row, end row} of the chunk workgroup `block` takes
  // (grid = 8 * blocks per XCD; {0, 0, 0, 0} pads).  Workgroup b runs on XCD b % 8: every XCD gets a CONTIGUOUS range of chunks (their
  // x entries meet in one L2) holding an eighth of the non-zeros, and takes them largest first, so the last workgroups to start are the short ones
  DevBuf<int32_t> desc;
  int grid = 0;
  std::vector<int32_t> order;  // host copy: chunk of every block, -1 = padding
  // distributed handles: `desc` holds the chunks whose staged columns are all owned (launched while the ghost exchange is in flight),
  // `desc_if` those that stage a ghost column (launched behind it); frac_if = their share of the non-zeros
  DevBuf<int32_t> desc_if;
  int grid_if = 0, n_chunks_if = 0;
  double frac_if = 0;
};

// Persistent Schur-complement CG (nsx_cg.hip): negative_S_tilde as slabs of 256 slots per Schur ILU block, 16-bit columns
// into the block's unique-column list.
struct CgPlan {
  bool ok = false, values_current = false;
  int64_t n_slots = 0;
  DevBuf<int32_t> u_ptr, u_cols, s_ptr, s_info, s_src;
  DevBuf<uint16_t> s_lidx;
  DevBuf<double> s_val;
  DevBuf<uint16_t> a_lidx;   // local column of every entry in CSR order (the LDS-resident operator of k_cg_schur<.., true>)
  int32_t max_block_nnz = 0;
};

struct ProfEntry {
  int64_t launches = 0;
  double bytes = 0;  // algorithmic bytes summed over the launches (a scope's size may vary: Gram-Schmidt sweeps)
  double ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct Comm;  // RCCL state (nsx_comm.hip)

enum { N_SLOTS = 128 };  // device scalar slots (reduction results)

}  // namespace nsx

struct nsx_handle {
  nsx_params prm{};
  std::string err;
  hipStream_t stream = nullptr;
  // ---- discretisation
  int dim = 0, n_q = 0, np2 = 0, np1 = 0, dpc = 0;
  int n_cells = 0, N2 = 0, NP = 0, n_u = 0, n_p = 0;  // local sizes (rows owned by this handle)
  int N2_loc = 0, NP_loc = 0;                          // owned + ghost (== N2, NP on one GPU)
  // vector layout [u_owned (n_u) | u_ghost (g_u) | p_owned (n_p) | p_ghost (g_p)]; off_p = n_u + g_u
  int n_cells1 = 0;                                    // layer-1 cells (touch an owned node): the per-step cell loop
  int g_u = 0, g_p = 0, off_p = 0, len_blk = 0, len_u = 0, len_p = 0;
  bool dist = false;
  int rank = 0, world = 1, goff_u = 0, goff_p = 0, n_u_glob = 0, n_p_glob = 0;  // global numbering of this rank's range
  std::vector<int32_t> ghost_u, ghost_p;               // global ids of the ghost nodes (sorted)
  nsx::HaloPlan haloU, haloP;
  nsx::RowSplit splitA, splitVel, splitB, splitG, splitS;  // F alone, F + block(0,1) (the saddle product), block(1,0), block(0,1), S
  hipStream_t comm_stream = nullptr;                       // halo pack / send / receive run here, beside the compute stream
  hipEvent_t ev_ready = nullptr;                           // "the input vector is complete" (compute stream -> communication stream)
  bool have_tables = false, have_mesh = false, assembled = false, prec_ready = false;
  std::vector<double> N2_h, dN2_h, N1_h, w_h;
  nsx::DevBuf<double> tab_N2, tab_dN2, tab_N1, tab_w, tab_N2T, tab_dN2T;  // T: [a][q] / [b][q][k]
  nsx::DevBuf<int32_t> cell_n2, cell_n1;  // SoA [a][cell]
  nsx::DevBuf<double> geo;                // SoA [(dim*dim+1)][cell]
  std::vector<int32_t> cell_n2_h, cell_n1_h;  // scalar connectivity in the INTERNAL numbering (what every product below is built from)
  // ---- internal layout (nsx_set_internal_layout): the caller keeps its own numbering and rank count, libnsx renumbers the owned
  //      nodes behind the boundary (virtual ranks + colour order, host/layout.hpp); perm: caller-local owned node -> internal node
  std::vector<int32_t> cell_n2_in, cell_n1_in;   // connectivity as handed over (caller-local ids: owned < N2 <= ghosts)
  std::vector<double> cell_coords_in;            // [n_cells][dim+1][dim]: a new layout reruns the set-up products
  std::vector<int32_t> in_rank_u_h, in_rank_p_h, in_sblk_h;  // the caller's rank / Schur block tables (local node ids)
  int layout_req_ranks = 0, layout_req_order = 0, layout_req_schur = 0;  // the request (0 ranks: none; may precede nsx_set_mesh)
  bool layout_on = false;
  int layout_colours = 0, layout_colours_p = 0;
  std::vector<int32_t> perm2_h, perm1_h, iperm2_h, iperm1_h;
  nsx::DevBuf<int32_t> perm2_d, perm1_d;
  nsx::DevBuf<double> io_stage;                  // block vector in the caller's order on its way in / out
  nsx::Csr caller_graph[2];                      // scalar velocity / Schur graph in the caller's numbering (exports), built on demand
  std::vector<int32_t> caller_pos[2];            // ... and for each of its entries the position in the internal CSR
  // ---- graphs and values
  nsx::DevCsr gA, gG, gB, gS, gPM;
  nsx::SpmvBlocked blkA;
  nsx::DevBuf<double> vS0, vMass, vStiff, vConv, vF, vG, vB, vPM, vSchur, luF, luS;
  nsx::DevBuf<int32_t> bt_of_g;            // for every G entry (i,k): position of (k,i) in the B graph
  nsx::GatherMap gmA, gmG, gmB, gmPM;
  nsx::DevBuf<double> cellbuf;             // per-cell local matrices, SoA [(entry)][cell]
  // ---- vectors (n_u + n_p)
  nsx::DevBuf<double> sol, sol_owned, prev_sol, rhs;
  // ---- ranks / ILU
  std::vector<int32_t> rank_u_h, rank_p_h, sblk_h;
  nsx::DevBuf<int32_t> rank_u;             // node units
  nsx::IluSchedule schedF, schedS;
  nsx::DevBuf<double> dbar;                // per-rank Dirichlet diagonal
  nsx::DevBuf<int32_t> bc_dofs;
  nsx::DevBuf<double> bc_vals;
  nsx::DevBuf<double> dirmask;             // [n_u] 1 = free, 0 = constrained row of block (0,1)
  std::vector<int32_t> bc_cache;
  // ---- preconditioner vectors
  nsx::DevBuf<double> diag_D, diag_D_inv, neg_diag_D_inv, lump_M, schur_w;
  nsx::DevBuf<double> schur_w_prev;        // weights the current Schur product / its factors were computed from
  bool schur_valid = false, schur_pending = false;  // pending: rebuilt in this initialisation, not yet confirmed (prec_confirm)
  int schur_type = -1;
  // ---- Krylov workspace
  std::vector<nsx::DevBuf<double> *> pool;  // temporary vectors handed out by size
  nsx::DevBuf<double> red_partial;          // reduction partials
  nsx::DevBuf<double> scal;                 // device scalars
  int slot_nb[nsx::N_SLOTS] = {0};          // >0: the slot's value is still spread over that many partial sums
  double *scal_host = nullptr;              // pinned mirror
  // host-visible publication of scalars without a memcpy + stream sync: the publishing kernel writes the values and then
  // a sequence number into fine-grained mapped host memory, the host polls the sequence number
  double *pub_host = nullptr, *pub_dev = nullptr;  // [N_SLOTS] values, then the flag word
  unsigned long long pub_seq = 0;
  nsx::DevBuf<unsigned int> pub_counter;
  // persistent Gram-Schmidt kernel (nsx_blas.hip: k_mgs): two mailbox regions used alternately by successive launches
  nsx::DevBuf<unsigned long long> mgs_box;
  bool sched_dirty = true;  // the ILU schedules do not match the current rank tables yet
  int mgs_used_wg[2] = {0, 0}, mgs_used_steps[2] = {0, 0};  // what the last launch on each region filled
  int mgs_parity = 0, mgs_max_wg = 0;  // mgs_max_wg = 0: the launch-per-link chain is used
  int mgs_max_wg_e[3] = {0, 0, 0};     // resident-grid limit of the 8 / 10 / 12 (link-by-link variants: 20) entries-per-thread instantiations
  int mgs_links = 2;                   // links of the add_and_dot chain per grid-wide exchange (NSX_MGS_LINKS; 1 = k_mgs)
  bool mgs_disabled = false;
  double mgs_guard_override = -1.0;    // >= 0: threshold of the Gram formula for |w'|^2 for the duration of nsx_gram_schmidt_cycle
  int gx_drop_wg = -1;                 // NSX_GX_DROP_WG (fault injection, tests): this workgroup of a persistent grid never posts its sums
  int n_persistent_fallbacks = 0;      // persistent kernels that timed out on this handle (nsx_solve_stats::persistent_fallbacks)
  // distributed sweep with two collectives (mgs_lowsync): partial sums / all-reduced values / one 32 x 32 Gram matrix per GMRES nesting level
  nsx::DevBuf<double> ls_partial, ls_vals, ls_gram;
  int gmres_depth = 0, ls_mode = -1;
  // the persistent sweep of a distributed run (k_mgs_one<.., true>): two value buffers used alternately, arrival counter, release flag
  nsx::DevBuf<double> mgs_ext_vals;
  nsx::DevBuf<unsigned long long> mgs_ext_words;  // [0] release flag, [1] arrival counter (32 bits used), [2] the sweep the grid gave up on
  unsigned int mgs_ext_expected = 0;
  int mgs_ext_parity = 0;
  std::map<int, int> mgs_dist_fit;   // role of the vector in the solve (mgs_role) -> do ALL ranks' resident grids hold their vector of that role (agreed once per role)
  bool cu_reserve_failed = false;    // the ranks tried to mask their streams and one of them could not: all are back on plain streams, nobody asks again
  bool mgs_leave_req = false;        // a flag wait of this rank's grid timed out on its own: the next sweep asks all ranks to leave the persistent path
  int mgs_local_timeouts = 0;
  int mgs_last_fused = 0, mgs_fused_launches = 0;  // the last persistent sweep had the velocity triangular solves inside (k_ilu_mgs)
  int ilu_mgs_cap[3] = {-1, -1, -1}, ilu_mgs_cap_rows = -1;  // resident-grid limits of the fused kernel's two instantiations (for the LDS request of the current schedule)
  int mgs_last_e = 0, mgs_last_nwg = 0, mgs_last_dist = 0, mgs_max_e_seen = 0;  // the last persistent sweep: entries per thread, grid, collective inside (nsx_path_info)
  int cgd_agreed = -1;               // two-launch Schur CG: -1 not decided for the current schedules, 0 / 1 the ranks' common answer
  nsx::DevBuf<double> ext_self;           // development (NSX_EXT_SELF_P2P): operands of the self-addressed send / receive in front of the sweep's collective
  hipStream_t stream_plain = nullptr;  // the compute stream of the handle's creation once `stream` has been replaced by one with a CU mask (comm_reserve_cus)
  std::vector<hipStream_t> retired_streams;  // streams RCCL has launched on and the handle no longer uses: destroyed behind ncclCommDestroy (comm_destroy)
  int cu_reserved = 0;               // CUs (one per XCD) the compute stream leaves to the communication stream's kernels
  int mgs_dist_cap_reserved[3] = {0, 0, 0};  // grid limits of the distributed sweep's instantiations on the masked compute stream
  int comm_probe_local = -1;         // -1 not probed, 0 / 1: kernels of the communication stream run beside a waiting kernel of the compute stream (comm_prepare_streams)
  int mgs_dist_state = -1;           // -1 not decided yet, 0 two-pass sweep (mgs_lowsync), 1 the collective inside the persistent grid
  int mgs_max_wg_dist[3] = {0, 0, 0};  // resident-grid limits of the distributed instantiations (8 / 10 / 12 entries per thread), room left for the collective
  long long n_allreduce = 0, n_halo = 0;  // collectives issued (nsx_comm_counters)
  int self_p2p = 0;                       // development (NSX_EXT_SELF_P2P, read by nsx_comm_init): self-addressed send / receive pairs in front of collectives
  int n_ext_collectives = 0;              // collectives inside a persistent grid so far (fault injection: NSX_EXT_LATE_RELEASE)
  bool mgs_redo_ahead = false;         // a sweep fell back to the chain after work depending on its w had been enqueued
  // persistent Schur-complement CG (nsx_cg.hip: k_cg_schur): mailbox regions, work vectors (d double-buffered, h)
  nsx::DevBuf<unsigned long long> cg_box;
  nsx::CgPlan cgplan;
  nsx::DevBuf<double> cg_vec;
  int cg_parity = 0, cg_max_wg = 0, cg_max_wg_res[2] = {0, 0};  // _res: resident-grid limits of the variants that keep the block inverses in registers (96 / 128 rows)
  int cg_max_wg_lres[2] = {0, 0};                      // ... and of the variants that keep the operator's rows in LDS as well
  bool cg_resident = false;                            // the last Schur CG ran with the block inverses in registers
  bool cg_lds_resident = false;                        // ... and with the operator in LDS
  bool cg_variant_said = false;
  bool cg_disabled = false;
  nsx::DevBuf<double> cgd_raw;              // ... per-block partial sums of a rank with more Schur blocks than entries of a partial-sum array (k_cgd_fold)
  int cg_last_path = 0;                     // the last Schur CG: 0 none yet, 1 one launch per operation, 2 one persistent launch, 3 two launches per iteration
  nsx::DevBuf<double> cgd_vec, cgd_parts;   // distributed Schur CG in two launches per iteration (cg_schur_fused): g, h, S d, d (double-buffered, with ghosts); partial sums
  // ---- force evaluation (compute_forces): obstacle faces + face-quadrature tables
  int ff_n = 0, ff_nq = 0;
  nsx::DevBuf<int32_t> ff_cells, ff_lf;
  nsx::DevBuf<double> ff_N2, ff_dN2, ff_N1, ff_w, ff_out;
  // ---- profiling
  bool prof_on = false;
  std::map<std::string, nsx::ProfEntry> prof;
  std::vector<std::string> prof_names;
  // ---- comm
  nsx::Comm *comm = nullptr;
  bool defer_red = false;             // distributed runs: all-reduces of finished reductions are held back ...
  std::vector<int> pending_red;       // ... for these slots, and merged when released (defer_reductions)
};

namespace nsx {

// RAII launch timer used by every kernel launch site.
struct LaunchScope {
  nsx_handle *h;
  ProfEntry *e = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  LaunchScope(nsx_handle *h_, const char *name, double bytes) : h(h_) {
    if (!h->prof_on) return;
    e = &h->prof[name];
    e->bytes += bytes;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, h->stream);
  }
  ~LaunchScope() {
    if (!e) return;
    (void)hipEventRecord(b, h->stream);
    e->launches++;
    e->pending.emplace_back(a, b);
  }
};

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- kernels / steps implemented across the .hip files
void setup_ilu_schedule(nsx_handle *h, const Csr &g, const std::vector<int32_t> &block_ptr, IluSchedule &s, int blocks_per_wave, bool allow_dense = false,
                        int ncomp = 1);
void build_schur_graph(nsx_handle *h);
// host vectors of the C-ABI are in the CALLER's numbering (global [n_u_glob | n_p_glob]); device block vectors in the internal one
void vec_from_caller(nsx_handle *h, double *dev, const double *host, bool with_ghosts);  // owned (+ ghost) entries of a block vector
void vec_to_caller(nsx_handle *h, const double *dev, double *host);                      // owned entries only
void part_from_caller(nsx_handle *h, int which, double *dev, const double *host);        // one space, single-process handles: 0 velocity [n_u], 1 pressure [n_p]
void part_to_caller(nsx_handle *h, int which, const double *dev, double *host);
inline int32_t node_to_internal(const nsx_handle *h, int32_t local_node) { return h->layout_on ? h->perm2_h[local_node] : local_node; }
inline int32_t pnode_to_internal(const nsx_handle *h, int32_t local_node) { return h->layout_on ? h->perm1_h[local_node] : local_node; }
void ensure_schedules(nsx_handle *h);  // (re)build the ILU schedules if the rank / Schur block tables changed

// assembly (nsx_assemble.hip)
void run_assemble(nsx_handle *h, bool first, int flags);
void run_dirichlet(nsx_handle *h, int n, const int32_t *dofs, const double *vals);

// sparse (nsx_sparse.hip)
bool blocked_usable(const nsx_handle *h);                                                   // F->vmult goes through the LDS-staged SpMV
void spmv_F(nsx_handle *h, const double *vals, const double *x, double *y);                 // y_u = A x_u   (dim comps)
void spmv_saddle(nsx_handle *h, const double *x, double *y);                                // full block vmult
void spmv_G(nsx_handle *h, const double *xp, double *yu, bool accumulate);                  // y_u (+)= block(0,1) x_p
void spmv_B(nsx_handle *h, const double *xu, double *yp);                                   // y_p = block(1,0) x_u
void spmv_S(nsx_handle *h, const double *x, double *y);                                     // y = negative_S x
void schur_numeric(nsx_handle *h, const double *w);                                         // S = B diag(w) G
void ilu_factor(nsx_handle *h, const DevCsr &g, IluSchedule &s, const double *vals, double *lu, const char *name);
void ilu_check(nsx_handle *h);  // after a synchronisation: throws if a factorisation kernel reported a failure
// dot_slot >= 0: also leave b.x in that scalar slot when the packed kernel can do it; returns whether it did
bool ilu_solve(nsx_handle *h, const DevCsr &g, const IluSchedule &s, const double *lu, const double *b, double *x, int ncomp,
               const char *name, int dot_slot = -1);
void extract_diag(nsx_handle *h, const DevCsr &g, const double *vals, double *d);           // scalar diag
void abs_rowsum(nsx_handle *h, const DevCsr &g, const double *vals, double *d);

// Index span of a BLAS-1 operation: n owned entries; entries at i >= split sit `gap` further (the ghost velocity block
// between the owned velocity and the owned pressure part of a distributed block vector).  Plain vectors: Span(n).
struct Span {
  int n, split, gap;
  Span(int n_) : n(n_), split(n_), gap(0) {}
  Span(int n_, int split_, int gap_) : n(n_), split(split_), gap(gap_) {}
};
inline Span blk_span(const nsx_handle *h) { return Span(h->n_u + h->n_p, h->n_u, h->g_u); }

// DPP lane permutation of a double (two 32-bit moves, pure VALU); CTRL as in the ISA: 0xB1/0x4E quad_perm,
// 0x141 row_half_mirror, 0x140 row_mirror, 0x120+n row_ror:n
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// same, but only the rows of 16 lanes selected by ROW_MASK receive a value; the other rows get 0 (used with the row
// broadcasts 0x142 row_bcast:15 and 0x143 row_bcast:31, which hand the last lane of a row / of the lower half to the rows above)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64_rows(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}

// BLAS-1 (nsx_blas.hip): scalars live in h->scal[slot]
void v_copy(nsx_handle *h, int n, double *d, const double *s);  // raw copy of n contiguous entries
void v_zero(nsx_handle *h, int n, double *d);
void v_add(nsx_handle *h, Span n, double *d, double a, const double *v);                     // d += a v
void v_add_dev(nsx_handle *h, Span n, double *d, double a, int slot, const double *v);       // d += a*scal[slot]*v
void v_sadd(nsx_handle *h, Span n, double *d, double s, double a, const double *v);          // d = s d + a v
void v_scale(nsx_handle *h, Span n, double *d, double a);
void v_scale_dev_inv(nsx_handle *h, Span n, double *d, int slot);                            // d *= 1/scal[slot]
void v_scale_vec(nsx_handle *h, int n, double *d, const double *f);
void v_dot(nsx_handle *h, Span n, const double *a, const double *b, int slot);               // scal[slot] = a.b
void v_add_and_dot(nsx_handle *h, Span n, double *d, double a, int aslot, const double *v, const double *w, int slot);
                                                                                            // d += a*scal[aslot]*v ; scal[slot] = d.w
// modified Gram-Schmidt sweep of SolverGMRES (w against v_0..v_{dim-1}): out[i] = h(i), out[dim] = |w|^2 afterwards (host
// values; scal[slot0+i] holds them too).  normalize: also w *= 1/|w| if the sweep runs as one launch; returns whether it did.
// after_launch (optional) runs between the launch and the wait for the coefficients, only when the sweep is one launch
// AND normalises w: the caller may enqueue work that depends on the finished w alone.
bool v_mgs(nsx_handle *h, Span n, double *w, int dim, double *const *vs, int slot0, bool normalize, double *out,
           const std::function<void()> *after_launch = nullptr, bool consider = false, double *gram = nullptr, const double *ilu_rhs = nullptr);
// ilu_rhs: the sweep's input is w = (LU_F)^-1 ilu_rhs, computed first (inside the sweep's own launch where possible: k_ilu_mgs)
// consider: also out[dim+1] = |w|^2 BEFORE the sweep (SolverGMRES' re-orthogonalisation test); a single-launch sweep then
// normalises w only if the test does not ask for a second sweep.
void v_axpy_multi(nsx_handle *h, Span n, double *x, int k, double *const *vs, const double *coef_host);
void finalize_slots(nsx_handle *h, int slot0, int count);
// for kernels that leave nb <= 512 per-workgroup partial sums of a scalar themselves: where to put them, and the
// bookkeeping (and the all-reduce of a distributed run) once the kernel is launched
double *red_out(nsx_handle *h, int slot, int nb);
void after_reduction(nsx_handle *h, int slot, int nb);
void defer_reductions(nsx_handle *h, bool on);  // hold back / release (merged) the all-reduces of a distributed run
double read_scalar(nsx_handle *h, int slot);
void read_scalars(nsx_handle *h, int slot0, int count, double *out);
unsigned long long publish_scalars(nsx_handle *h, int slot0, int count);  // asynchronous half of read_scalars
void collect_published(nsx_handle *h, unsigned long long seq, int slot0, int count, double *out);
void wait_published(nsx_handle *h, unsigned long long seq);  // host waits for the sequence number of a publication
// persistent CG on the Schur complement (nsx_cg.hip); false: not applicable here, use the launch-per-operation solver
void build_cg_plan(nsx_handle *h);   // with the ILU schedules
void cg_pack_values(nsx_handle *h);  // after every schur_numeric
bool cg_schur_persistent(nsx_handle *h, double *x, const double *b, double rtol, int maxiter, int *steps, double *last, int *status);
bool cg_schur_fused(nsx_handle *h, double *x, const double *b, double rtol, int maxiter, int *steps, double *last, int *status);  // distributed runs: two launches per iteration
void write_scalar(nsx_handle *h, int slot, double v);

// solver (nsx_solve.hip)
void prec_initialize(nsx_handle *h, int type);
void prec_confirm(nsx_handle *h);  // after the synchronisation behind prec_initialize: ilu_check + the Schur values become reusable
void prec_vmult(nsx_handle *h, int type, double inner_rtol, int inner_maxiter, double *dst, const double *src,
                nsx_solve_stats *st);
void solve_time_step(nsx_handle *h, int type, double tol, double inner_rtol, int maxiter, int inner_maxiter,
                     nsx_solve_stats *st);

// comm (nsx_comm.hip)
void comm_allreduce_scalars(nsx_handle *h, int slot0, int count);
void comm_allreduce_partials(nsx_handle *h, double *partials, int count);  // in place, same count on every rank
bool comm_agree_all(nsx_handle *h, bool mine);  // true iff `mine` is true on every rank (one collective): path choices that change the collective sequence
void comm_release_cus(nsx_handle *h);  // back to the plain compute stream and an unmasked communication stream
bool comm_reserve_cus(nsx_handle *h);  // replace the compute stream by one whose CU mask leaves one CU per XCD free (RCCL's kernel beside a persistent grid)
bool comm_streams_concurrent(nsx_handle *h);  // probe + agreement of all ranks (one collective): may a compute kernel wait for the communication stream?
bool comm_on_stream(const nsx_handle *h);  // RCCL backend: collectives are stream operations (the callback backend runs them on the host)
// the collective inside a persistent grid's exchange: on the communication stream wait for `arrive` to reach `expected`, all-reduce
// vals[0..count), store `seq` in `flag` (nsx_comm.hip)
void comm_ext_allreduce(nsx_handle *h, double *vals, int count, int fail_word, unsigned int *arrive, unsigned int expected, unsigned long long *flag,
                        unsigned long long seq);
void comm_halo(nsx_handle *h, HaloPlan &plan, double *x, int ncomp);
// the same exchange in two halves: begin enqueues pack + send/receive on the communication stream (after everything the
// compute stream holds so far), finish makes the compute stream wait for the ghosts; kernels launched in between overlap it
// packer (optional): fills the send buffer itself on the given stream (values that exist nowhere as a vector yet: the CG direction)
void comm_halo_begin(nsx_handle *h, HaloPlan &plan, double *x, int ncomp,
                     const std::function<void(hipStream_t, double *sendbuf, const int32_t *send_idx, int n_send)> *packer = nullptr);
void comm_halo_finish(nsx_handle *h, HaloPlan &plan, double *x, int ncomp);
void build_row_splits(nsx_handle *h);
inline void comm_halo_u(nsx_handle *h, const double *x) { if (h->dist) comm_halo(h, h->haloU, const_cast<double *>(x), h->dim); }
inline void comm_halo_p(nsx_handle *h, const double *x) { if (h->dist) comm_halo(h, h->haloP, const_cast<double *>(x), 1); }
void comm_destroy(nsx_handle *h);

}  // namespace nsx
