// nsx_setup.hip — handle life cycle and the host-side setup products of libnsx:
// sparsity graphs (reference NavierStokes3D.cpp:109-124), cell geometry, deterministic gather maps,
// per-rank ILU(0) level schedules (Ifpack overlap-0 semantics) and the Schur pattern
// (EpetraExt::MatrixMatrix::Multiply structural product, reference Preconditioners.hpp:144,248,358,468).
#include <algorithm>
#include <cmath>
#include <numeric>

#include "nsx_internal.hpp"
#include "../host/ilu_stream.hpp"
#include "../host/layout.hpp"

using namespace nsx;

static thread_local std::string g_create_error;

namespace nsx {

static void upload_csr(nsx_handle *h, DevCsr &d, bool square) {
  d.rowptr.upload(d.host.rowptr, h->stream);
  d.colind.upload(d.host.colind, h->stream);
  if (square) {
    std::vector<int32_t> diag(d.host.n_rows);
    for (int32_t i = 0; i < d.host.n_rows; ++i) {
      diag[i] = find_in_row(d.host, i, i);
      if (diag[i] < 0) NSX_THROW(NSX_ERR_ARG, "square graph has no diagonal entry in row %d", i);
    }
    d.diag.upload(diag, h->stream);
  }
}

// contributions of every cell-local pair (r,c) to CSR entry (rows[r], cols[c]); src = plane*stride*n_cells + cell
static void build_gather(nsx_handle *h, const Csr &g, int n_cells, int per_r, const int32_t *cell_r, int per_c,
                         const int32_t *cell_c, int plane_stride, bool plane_rc_swapped, GatherMap &gm) {
  const int64_t nnz = g.nnz();
  std::vector<int32_t> ptr(nnz + 1, 0);
  std::vector<int32_t> epos((size_t)n_cells * per_r * per_c);
  for (int c = 0; c < n_cells; ++c)
    for (int a = 0; a < per_r; ++a)
      for (int b = 0; b < per_c; ++b) {
        const int32_t row = cell_r[(size_t)c * per_r + a];
        if (row >= g.n_rows) {  // ghost row: assembled by its owner
          epos[((size_t)c * per_r + a) * per_c + b] = -1;
          continue;
        }
        const int32_t e = find_in_row(g, row, cell_c[(size_t)c * per_c + b]);
        if (e < 0) NSX_THROW(NSX_ERR_ARG, "internal: cell pair not in graph");
        epos[((size_t)c * per_r + a) * per_c + b] = e;
        ptr[e + 1]++;
      }
  for (int64_t e = 0; e < nnz; ++e) ptr[e + 1] += ptr[e];
  std::vector<int32_t> src(ptr[nnz]);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int c = 0; c < n_cells; ++c)
    for (int a = 0; a < per_r; ++a)
      for (int b = 0; b < per_c; ++b) {
        const int32_t e = epos[((size_t)c * per_r + a) * per_c + b];
        if (e < 0) continue;
        const int64_t plane = plane_rc_swapped ? ((int64_t)b * per_r + a) : ((int64_t)a * per_c + b);
        const int64_t off = plane * plane_stride * n_cells + c;
        if (off > INT32_MAX) NSX_THROW(NSX_ERR_UNSUPPORTED, "mesh too large for int32 gather offsets");
        src[fill[e]++] = (int32_t)off;
      }
  gm.n_out = nnz;
  gm.n_src = (int64_t)src.size();
  gm.ptr.upload(ptr, h->stream);
  gm.src.upload(src, h->stream);
}

void setup_ilu_schedule(nsx_handle *h, const Csr &g, const std::vector<int32_t> &bptr, IluSchedule &s, int blocks_per_wave, bool allow_dense, int ncomp) {
  const int nb = (int)bptr.size() - 1;
  s.n_blocks = nb;
  s.block_ptr_h = bptr;
  s.max_rows = 0;
  std::vector<int32_t> lev(g.n_rows, 0);
  std::vector<int32_t> f_ptr{0}, f_rows, b_ptr{0}, b_rows, off_f(nb + 1, 0), off_b(nb + 1, 0);
  f_rows.reserve(g.n_rows);
  b_rows.reserve(g.n_rows);
  s.max_levels = 0;
  std::vector<std::vector<int32_t>> buckets;
  for (int b = 0; b < nb; ++b) {
    const int r0 = bptr[b], r1 = bptr[b + 1];
    s.max_rows = std::max(s.max_rows, r1 - r0);
    for (int pass = 0; pass < 2; ++pass) {
      int nl = 0;
      if (pass == 0) {
        for (int i = r0; i < r1; ++i) {
          int l = 0;
          for (int k = g.rowptr[i]; k < g.rowptr[i + 1]; ++k) {
            const int j = g.colind[k];
            if (j >= i) break;
            if (j >= r0) l = std::max(l, lev[j] + 1);
          }
          lev[i] = l;
          nl = std::max(nl, l + 1);
        }
      } else {
        for (int i = r1 - 1; i >= r0; --i) {
          int l = 0;
          for (int k = g.rowptr[i + 1] - 1; k >= g.rowptr[i]; --k) {
            const int j = g.colind[k];
            if (j <= i) break;
            if (j < r1) l = std::max(l, lev[j] + 1);
          }
          lev[i] = l;
          nl = std::max(nl, l + 1);
        }
      }
      if (r1 == r0) nl = 0;
      buckets.assign(nl, {});
      for (int i = r0; i < r1; ++i) buckets[lev[i]].push_back(i);
      auto &ptr = pass == 0 ? f_ptr : b_ptr;
      auto &rows = pass == 0 ? f_rows : b_rows;
      for (int l = 0; l < nl; ++l) {
        rows.insert(rows.end(), buckets[l].begin(), buckets[l].end());
        ptr.push_back((int32_t)rows.size());
      }
      (pass == 0 ? off_f : off_b)[b + 1] = (int32_t)ptr.size() - 1;
      s.max_levels = std::max(s.max_levels, nl);
    }
  }
  s.in_block_nnz = 0;
  for (int b = 0; b < nb; ++b)
    for (int i = bptr[b]; i < bptr[b + 1]; ++i)
      for (int q = g.rowptr[i]; q < g.rowptr[i + 1]; ++q) s.in_block_nnz += g.colind[q] >= bptr[b] && g.colind[q] < bptr[b + 1];
  // ---- the in-block part of every row is one contiguous range of its (sorted) entries: [in_lo, in_hi); in_cptr: running count of
  //      in-block entries (the compact numbering k_ilu_factor_lds stages a block with)
  std::vector<int32_t> lo(g.n_rows), hi(g.n_rows);
  {
    std::vector<int32_t> cptr(g.n_rows + 1, 0);
    s.max_block_nnz = 0;
    for (int b = 0; b < nb; ++b) {
      for (int i = bptr[b]; i < bptr[b + 1]; ++i) {
        const int32_t *cb = g.colind.data() + g.rowptr[i], *ce = g.colind.data() + g.rowptr[i + 1];
        lo[i] = (int32_t)(std::lower_bound(cb, ce, bptr[b]) - g.colind.data());
        hi[i] = (int32_t)(std::lower_bound(cb, ce, bptr[b + 1]) - g.colind.data());
        cptr[i + 1] = cptr[i] + (hi[i] - lo[i]);
      }
      s.max_block_nnz = std::max(s.max_block_nnz, cptr[bptr[b + 1]] - cptr[bptr[b]]);
    }
    std::vector<int32_t> cpos((size_t)cptr[g.n_rows]);
    for (int i = 0; i < g.n_rows; ++i)
      for (int p = lo[i]; p < hi[i]; ++p) cpos[cptr[i] + (p - lo[i])] = p;
    s.in_lo.upload(lo, h->stream);
    s.in_hi.upload(hi, h->stream);
    s.in_cptr.upload(cptr, h->stream);
    s.in_cpos.upload(cpos, h->stream);
    // blocks in order of descending work: the factorisation's workgroups are dispatched in this order, the big blocks first
    std::vector<int32_t> order(nb);
    for (int b = 0; b < nb; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cptr[bptr[x + 1]] - cptr[bptr[x]] > cptr[bptr[y + 1]] - cptr[bptr[y]]; });
    s.fac_order.upload(order, h->stream);
  }
  // ---- blocks beyond what one wave (packed stream) or one workgroup can hold: merged level lists, one launch per level
  const int levelled_min = getenv("NSX_LEVELLED_MIN") ? atoi(getenv("NSX_LEVELLED_MIN")) : 4096;
  s.levelled = s.max_rows > levelled_min;
  s.packed_ok = false;
  s.dense = false;
  if (s.levelled) {
    for (int pass = 0; pass < 2; ++pass) {
      const auto &ptr = pass == 0 ? f_ptr : b_ptr;
      const auto &rows = pass == 0 ? f_rows : b_rows;
      const auto &off = pass == 0 ? off_f : off_b;
      std::vector<int32_t> gptr(s.max_levels + 1, 0), grows(rows.size());
      for (int b = 0; b < nb; ++b)
        for (int l = off[b]; l < off[b + 1]; ++l) gptr[l - off[b] + 1] += ptr[l + 1] - ptr[l];
      for (int l = 0; l < s.max_levels; ++l) gptr[l + 1] += gptr[l];
      std::vector<int32_t> fill(gptr.begin(), gptr.end() - 1);
      for (int b = 0; b < nb; ++b)
        for (int l = off[b]; l < off[b + 1]; ++l)
          for (int q = ptr[l]; q < ptr[l + 1]; ++q) grows[fill[l - off[b]]++] = rows[q];
      (pass == 0 ? s.gl_f_ptr_h : s.gl_b_ptr_h) = gptr;
      (pass == 0 ? s.gl_f_rows : s.gl_b_rows).upload(grows, h->stream);
      // one record per row in level order -- {row, first entry, end of entries, diagonal} of the sweep's triangle -- so that a level
      // kernel learns everything about its row in ONE coalesced load (it used to chase rows -> diag, in_lo/in_hi -> entries)
      std::vector<int32_t> rec(4 * grows.size());
      for (size_t r = 0; r < grows.size(); ++r) {
        const int i = grows[r], d = find_in_row(g, i, i);
        rec[4 * r + 0] = i;
        rec[4 * r + 1] = pass == 0 ? lo[i] : d + 1;
        rec[4 * r + 2] = pass == 0 ? d : hi[i];
        rec[4 * r + 3] = d;
      }
      (pass == 0 ? s.gl_f_rec : s.gl_b_rec).upload(rec, h->stream);
    }
    s.block_ptr.upload(bptr, h->stream);
    if (getenv("NSX_DEBUG"))
      fprintf(stderr, "[nsx] ilu schedule: rows %d blocks %d max_rows %d -> levelled, %d levels\n", g.n_rows, nb, s.max_rows, s.max_levels);
    return;
  }
  s.block_ptr.upload(bptr, h->stream);
  s.fwd_lvl_ptr.upload(f_ptr, h->stream);
  s.fwd_rows.upload(f_rows, h->stream);
  s.bwd_lvl_ptr.upload(b_ptr, h->stream);
  s.bwd_rows.upload(b_rows, h->stream);
  s.blk_lvl_off.upload(off_f, h->stream);
  s.blk_lvl_off_b.upload(off_b, h->stream);

  // ---- explicit block inverses.  Worth it when the blocks are few, small and deep: the dense matrices of all blocks
  // together (sum n_b^2) must not cost more traffic than the chain of the sparse sweep costs time.  Used for the Schur
  // matrix (one component, ~100-row blocks with ~95 dependency levels); the velocity blocks stay sparse.
  s.dense = false;
  if (allow_dense && ncomp == 1) {
    std::vector<int64_t> off(nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
      const int64_t n = bptr[b + 1] - bptr[b];
      off[b + 1] = off[b] + n * n;
    }
    // (blocks of at most 256 rows -- what the Schur CG kernels take -- cost at most 256 entries per row whatever the mesh size: no limit on their total;
    // the 10.6 M-DoF mesh on one GPU has 4 833 blocks of <= 96 rows, 42 M entries = 335 MB)
    const int64_t limit = getenv("NSX_DENSE_MAX") ? atoll(getenv("NSX_DENSE_MAX")) : ((int64_t)32 << 20);  // entries (256 MB)
    if (off[nb] > 0 && (off[nb] <= limit || (s.max_rows <= 256 && !getenv("NSX_DENSE_MAX"))) && s.max_rows <= 4096) {
      s.dense = true;
      s.dn_entries = off[nb];
      s.dn_off.upload(off, h->stream);
      s.dn_P.alloc((size_t)off[nb]);
      if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] ilu schedule: explicit block inverses, %lld entries (%.1f MB)\n", (long long)off[nb], 8e-6 * off[nb]);
    }
  }
  // ---- the packed stream of the solve kernel (host/ilu_stream.hpp, k_ilu_solve_lanes) -- unless it would never be used: the
  // schedule is served by the explicit block inverses above (the Schur matrix), or its largest block alone does not fit the 16-bit
  // LDS addresses of the stream (few large ranks: up to 4096-row blocks run through the workgroup-per-block kernel).  Building it
  // is list scheduling per wave plus an upload the size of the factor; slot_of stays null, so the factorisation kernels skip the
  // stream writes.
  const bool stream_fits = ((size_t)s.max_rows + 64) * sizeof(double) * ncomp <= 65536;
  s.n_waves = 0;
  s.n_slabs = 0;
  s.stream_ncomp = 0;
  if (!s.dense && stream_fits) {
    IluStream st;
    const int ept = getenv("NSX_ILU_EPT") ? std::max(1, std::min(4, atoi(getenv("NSX_ILU_EPT")))) : 2;
    build_ilu_stream(g, bptr, std::max(1, blocks_per_wave), ncomp, 2, st, ept);
    s.stream_ncomp = ncomp;
    s.stream_epl = st.epl;
    s.blocks_per_wave = blocks_per_wave;
    s.n_waves = st.n_waves;
    s.max_wave_rows = st.max_wave_rows;
    s.n_slabs = st.n_slabs;
    s.packed_ok = st.ok;
    if (getenv("NSX_DEBUG"))
      fprintf(stderr, "[nsx] ilu schedule: rows %d blocks %d max_rows %d levels(max) %d lane-owner stream: %d waves (%d blocks each), %d entries per tick, slabs %lld (max/wave %lld) fill %.2f, "
                      "%.1f MB, in-block entries %lld, LDS rows/wave <= %d%s\n",
              g.n_rows, nb, s.max_rows, s.max_levels, st.n_waves, blocks_per_wave, st.epl, (long long)st.n_slabs, (long long)st.max_wave_slabs,
              (double)st.used_slots / (double)std::max<int64_t>(1, st.n_slabs * 64 * st.epl), 1e-6 * (double)st.n_slabs * 64 * (8 * st.epl + 4 * ilu_meta_words(st.epl)),
              (long long)st.in_block_nnz, st.max_wave_rows, st.ok ? "" : " (too many: not used)");
    s.pk_row_ptr.upload(st.row_ptr, h->stream);
    s.pk_rows.upload(st.rows, h->stream);
    {
      std::vector<int32_t> slot(g.n_rows);
      for (size_t k = 0; k < st.rows.size(); ++k) slot[st.rows[k]] = (int32_t)k;
      s.pk_dinv_slot.upload(slot, h->stream);
    }
    s.pk_slab_ptr.upload(st.slab_ptr, h->stream);
    s.pk_meta.upload(reinterpret_cast<const int32_t *>(st.meta.data()), st.meta.size(), h->stream);
    s.pk_slot_of.upload(st.slot_of, h->stream);
    s.pk_val.alloc((size_t)(st.n_slabs + ILU_STREAM_PAD) * 64 * st.epl);
    s.pk_val.zero(h->stream);
    s.pk_dinv.alloc(g.n_rows);
  }

}

// chunks = [bounds[c], bounds[c+1])
// n_own: columns below it are owned by this handle (one GPU: all of them)
void build_blocked(nsx_handle *h, const Csr &g, const std::vector<int32_t> &bounds, SpmvBlocked &b, int n_own) {
  b.n_chunks = (int)bounds.size() - 1;
  std::vector<int32_t> cptr((size_t)b.n_chunks + 1, 0), ucols, tmp;
  std::vector<uint16_t> lidx(g.nnz());
  b.max_ucols = b.max_rows = 0;
  for (int c = 0; c < b.n_chunks; ++c) {
    const int r0 = bounds[c], r1 = bounds[c + 1];
    b.max_rows = std::max(b.max_rows, r1 - r0);
    tmp.assign(g.colind.begin() + g.rowptr[r0], g.colind.begin() + g.rowptr[r1]);
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    if (tmp.size() > 65535) NSX_THROW(NSX_ERR_UNSUPPORTED, "blocked SpMV: chunk touches more than 65535 columns");
    for (int p = g.rowptr[r0]; p < g.rowptr[r1]; ++p)
      lidx[p] = (uint16_t)(std::lower_bound(tmp.begin(), tmp.end(), g.colind[p]) - tmp.begin());
    ucols.insert(ucols.end(), tmp.begin(), tmp.end());
    cptr[c + 1] = (int32_t)ucols.size();
    b.max_ucols = std::max<int>(b.max_ucols, (int)tmp.size());
  }
  b.ucols_total = (double)ucols.size();
  // launch tables.  One-GPU handles: one table over all chunks.  Distributed handles (n_own < number of columns): the chunks whose
  // staged columns are all owned go into the first table -- launched while the ghost exchange is in flight -- and the chunks that stage
  // a ghost column into the second one, launched behind it (the Epetra_Import + local multiply of every vmult, Preconditioners.hpp:382,405).
  // Same kernel, same lanes, same sums per row as on one GPU.
  auto make_desc = [&](const std::vector<int32_t> &chunks, DevBuf<int32_t> &out, int &grid, std::vector<int32_t> *order) {
    // XCD k takes the chunks [cut[k], cut[k+1]) of the list: boundaries where the running non-zero count passes k / 8 of the total
    const int nc = (int)chunks.size();
    auto cnnz = [&](int c) { return (int64_t)(g.rowptr[bounds[c + 1]] - g.rowptr[bounds[c]]); };
    int64_t total = 0;
    for (int c : chunks) total += cnnz(c);
    int cut[9];
    cut[0] = 0;
    int64_t run = 0;
    for (int k = 1, j = 0; k <= 8; ++k) {
      while (j < nc && (run + cnnz(chunks[j])) * 8 <= total * k) run += cnnz(chunks[j++]);
      cut[k] = k == 8 ? nc : j;
    }
    int per_xcd = 0;
    for (int k = 0; k < 8; ++k) per_xcd = std::max(per_xcd, cut[k + 1] - cut[k]);
    grid = 8 * per_xcd;
    if (order) order->assign((size_t)grid, -1);
    std::vector<int32_t> desc((size_t)grid * 4, 0), list;
    static const bool largest_first = !(getenv("NSX_SPMV_ORDER") && atoi(getenv("NSX_SPMV_ORDER")) == 0);
    for (int k = 0; k < 8; ++k) {
      list.assign(chunks.begin() + cut[k], chunks.begin() + cut[k + 1]);
      if (largest_first) std::stable_sort(list.begin(), list.end(), [&](int32_t a, int32_t c) { return cnnz(a) > cnnz(c); });
      for (size_t j = 0; j < list.size(); ++j) {
        const int blk = (int)j * 8 + k, c = list[j];
        if (order) (*order)[blk] = c;
        desc[4 * (size_t)blk + 0] = cptr[c];
        desc[4 * (size_t)blk + 1] = cptr[c + 1] - cptr[c];
        desc[4 * (size_t)blk + 2] = bounds[c];
        desc[4 * (size_t)blk + 3] = bounds[c + 1];
      }
    }
    out.upload(desc, h->stream);
  };
  {
    std::vector<int32_t> first, second;
    int64_t nnz_second = 0;
    for (int c = 0; c < b.n_chunks; ++c) {
      const bool ghost = cptr[c + 1] > cptr[c] && ucols[(size_t)cptr[c + 1] - 1] >= n_own;  // the list is sorted: its last entry is the largest column
      (ghost ? second : first).push_back(c);
      if (ghost) nnz_second += g.rowptr[bounds[c + 1]] - g.rowptr[bounds[c]];
    }
    make_desc(first, b.desc, b.grid, &b.order);
    b.n_chunks_if = (int)second.size();
    b.grid_if = 0;
    b.frac_if = g.nnz() ? (double)nnz_second / (double)g.nnz() : 0.0;
    if (!second.empty()) make_desc(second, b.desc_if, b.grid_if, nullptr);
  }
  b.crow.upload(bounds, h->stream);
  b.cptr.upload(cptr, h->stream);
  b.ucols.upload(ucols, h->stream);
  b.lidx.upload(lidx, h->stream);
  if (getenv("NSX_DEBUG"))
    fprintf(stderr, "[nsx] blocked spmv: rows %d chunks %d (max %d rows; %d of them stage a ghost column: %.1f %% of the non-zeros) unique cols/chunk avg %.0f max %d (nnz/unique %.1f)\n",
            g.n_rows, b.n_chunks, b.max_rows, b.n_chunks_if, 100.0 * b.frac_if, b.ucols_total / std::max(1, b.n_chunks), b.max_ucols, (double)g.nnz() / b.ucols_total);
}

// Chunk boundaries of the LDS-staged SpMV.  With a rank table the chunks are unions of consecutive rank blocks: a rank's rows
// are the nodes of one subdomain, so a chunk that ends where a subdomain ends stages fewer columns than one that straddles
// three subdomains (6.3 instead of 5.5 non-zeros per staged entry at one ~85-row rank per chunk).  Small chunks win: two or
// three ranks per chunk stage even less (7.9) but measured slower (38 and 43 against 34 us: fewer, longer workgroups).
// Without a rank table, or with ranks too large for a chunk: a fixed row count.
static std::vector<int32_t> spmv_chunks(nsx_handle *h) {
  const int n = h->gA.host.n_rows;
  const int target = getenv("NSX_SPMV_R") ? std::max(16, std::min(448, atoi(getenv("NSX_SPMV_R")))) : 130;
  std::vector<int32_t> bounds{0};
  const std::vector<int32_t> &rk = h->rank_u_h;
  const bool by_rank = !(getenv("NSX_SPMV_BY_RANK") && atoi(getenv("NSX_SPMV_BY_RANK")) == 0) && rk.size() > 2 &&
                       (double)n / (double)(rk.size() - 1) <= target;
  // ranks larger than a chunk (fewer, larger virtual ranks: --ranks 2048 has ~170 rows per block) are cut into equal parts of at most
  // `target` rows: a chunk still ends where a subdomain ends
  const bool split_ranks = !by_rank && !(getenv("NSX_SPMV_BY_RANK") && atoi(getenv("NSX_SPMV_BY_RANK")) == 0) && rk.size() > 2 &&
                           (double)n / (double)(rk.size() - 1) <= 4.0 * target;
  if (by_rank) {
    int start = 0;
    for (size_t r = 1; r < rk.size(); ++r) {
      // close the chunk in front of a rank that would take it past the target (a single oversized rank is cut below)
      if (rk[r] - start > target && rk[r - 1] > start) {
        bounds.push_back(rk[r - 1]);
        start = rk[r - 1];
      }
    }
    bounds.push_back(n);
  } else if (split_ranks) {
    for (size_t r = 0; r + 1 < rk.size(); ++r) {
      const int rows = rk[r + 1] - rk[r];
      if (rows <= 0) continue;
      const int parts = (rows + target - 1) / target;
      for (int q = 1; q <= parts; ++q) bounds.push_back(rk[r] + (int)((int64_t)rows * q / parts));
    }
    if (bounds.back() != n) bounds.push_back(n);
  } else {
    for (int r = 128; r < n; r += 128) bounds.push_back(r);
    bounds.push_back(n);
  }
  // no chunk above 448 rows (the kernel's row-pointer buffer)
  std::vector<int32_t> out{0};
  for (size_t c = 1; c < bounds.size(); ++c) {
    while (bounds[c] - out.back() > 448) out.push_back(out.back() + 224);
    if (bounds[c] > out.back()) out.push_back(bounds[c]);
  }
  return out;
}

void build_schur_graph(nsx_handle *h) {
  // structural product block(1,0) * block(0,1); block(0,1) has the transposed pattern of block(1,0), and block(1,0)
  // is stored for every local (owned + ghost) pressure row, so the product is formed from B and B^T alone
  const Csr &B = h->gB.host;
  std::vector<int32_t> tp((size_t)h->N2_loc + 1, 0), tr(B.nnz());
  for (int64_t k = 0; k < B.nnz(); ++k) tp[B.colind[k] + 1]++;
  for (int i = 0; i < h->N2_loc; ++i) tp[i + 1] += tp[i];
  {
    std::vector<int32_t> fill(tp.begin(), tp.end() - 1);
    for (int j = 0; j < B.n_rows; ++j)
      for (int k = B.rowptr[j]; k < B.rowptr[j + 1]; ++k) tr[fill[B.colind[k]]++] = j;
  }
  Csr S;
  S.n_rows = h->NP;
  S.n_cols = h->NP_loc;
  S.rowptr.assign((size_t)h->NP + 1, 0);
  std::vector<int32_t> mark(h->NP_loc, -1), cols;
  for (int i = 0; i < h->NP; ++i) {
    cols.clear();
    for (int kb = B.rowptr[i]; kb < B.rowptr[i + 1]; ++kb) {
      const int k = B.colind[kb];
      for (int q = tp[k]; q < tp[k + 1]; ++q) {
        const int j = tr[q];
        if (mark[j] != i) {
          mark[j] = i;
          cols.push_back(j);
        }
      }
    }
    std::sort(cols.begin(), cols.end());
    S.colind.insert(S.colind.end(), cols.begin(), cols.end());
    S.rowptr[i + 1] = (int32_t)S.colind.size();
  }
  h->gS.host = std::move(S);
  upload_csr(h, h->gS, true);
  h->vSchur.alloc(h->gS.nnz());
  h->luS.alloc(h->gS.nnz());
}

static void default_ranks(nsx_handle *h) {
  h->in_rank_u_h = {0, h->N2};
  h->in_rank_p_h = {0, h->NP};
  h->in_sblk_h.clear();
}

// The rank tables changed: the Dirichlet scan needs them at once, the ILU schedules (seconds of host work at 1 M DoF) are
// rebuilt when a preconditioner is next initialised, so nsx_set_mesh + nsx_set_ranks + nsx_set_schur_blocks build them once.
static void refresh_rank_products(nsx_handle *h) {
  h->rank_u.upload(h->rank_u_h, h->stream);
  h->dbar.alloc(h->rank_u_h.size() - 1);
  h->sched_dirty = true;
  h->prec_ready = false;
  h->schur_valid = false;  // the Schur ILU blocks follow the tables
  h->mgs_dist_fit.clear();  // sizes may have changed: what the ranks agreed on for the old ones is asked again (every rank gets here alike)
}

void ensure_schedules(nsx_handle *h) {
  if (!h->sched_dirty) return;
  build_blocked(h, h->gA.host, spmv_chunks(h), h->blkA, h->N2);
  // blocks per wave: enough rows to keep the 64 lanes of the lane-owner stream busy (~680 rows: 8 blocks of ~85 rows measured
  // best at 1 M DoF / 4096 ranks: fewer blocks per wave leave idle slots in the stream, more leave too few waves)
  auto blocks_per_wave = [](int n_rows, size_t n_blocks) { return std::max(1, std::min(64, (int)(680.0 * (double)n_blocks / std::max(1, n_rows) + 0.5))); };
  const std::vector<int32_t> &sb = h->sblk_h.empty() ? h->rank_p_h : h->sblk_h;
  const int bpwF = getenv("NSX_BPW_F") ? atoi(getenv("NSX_BPW_F")) : blocks_per_wave(h->N2, h->rank_u_h.size() - 1);
  const int bpwS = getenv("NSX_BPW_S") ? atoi(getenv("NSX_BPW_S")) : blocks_per_wave(h->NP, sb.size() - 1);
  setup_ilu_schedule(h, h->gA.host, h->rank_u_h, h->schedF, bpwF, false, h->dim);
  const bool denseS = !(getenv("NSX_DENSE_S") && atoi(getenv("NSX_DENSE_S")) == 0);
  setup_ilu_schedule(h, h->gS.host, sb, h->schedS, bpwS, denseS, 1);
  build_cg_plan(h);
  h->cgd_agreed = -1;  // path choices that the ranks make together are made again for the new schedules
  h->sched_dirty = false;
}


// caller order <-> internal order of one space: entry (node, c) of the caller sits at (perm[node], c) inside libnsx
__global__ void k_perm_vec(int n, int ncomp, const int32_t *__restrict__ perm, double *caller, double *internal, int to_internal) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = i / ncomp, c = i - node * ncomp;
  const int64_t j = (int64_t)perm[node] * ncomp + c;
  if (to_internal) internal[j] = caller[i];
  else caller[i] = internal[j];
}

static void perm_launch(nsx_handle *h, int which, double *caller_dev, double *internal_dev, bool to_internal) {
  const int ncomp = which == 0 ? h->dim : 1, n = which == 0 ? h->n_u : h->n_p;
  if (n) hipLaunchKernelGGL(k_perm_vec, dim3(cdiv(n, 256)), dim3(256), 0, h->stream, n, ncomp, (which == 0 ? h->perm2_d : h->perm1_d).p, caller_dev, internal_dev, to_internal ? 1 : 0);
}

void part_from_caller(nsx_handle *h, int which, double *dev, const double *host) {
  const int n = which == 0 ? h->n_u : h->n_p;
  if (!h->layout_on) {
    HIP_CHECK(hipMemcpyAsync(dev, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  } else {
    h->io_stage.alloc((size_t)h->n_u + h->n_p);
    HIP_CHECK(hipMemcpyAsync(h->io_stage.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    perm_launch(h, which, h->io_stage.p, dev, true);
  }
  HIP_CHECK(hipStreamSynchronize(h->stream));  // host arrays are borrowed for the call only
}

void part_to_caller(nsx_handle *h, int which, const double *dev, double *host) {
  const int n = which == 0 ? h->n_u : h->n_p;
  if (!h->layout_on) {
    HIP_CHECK(hipMemcpyAsync(host, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    h->io_stage.alloc((size_t)h->n_u + h->n_p);
    perm_launch(h, which, h->io_stage.p, const_cast<double *>(dev), false);
    HIP_CHECK(hipMemcpyAsync(host, h->io_stage.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

void vec_from_caller(nsx_handle *h, double *dev, const double *host, bool with_ghosts) {
  const int dim = h->dim;
  if (!h->dist) {
    if (!h->layout_on) {
      HIP_CHECK(hipMemcpyAsync(dev, host, ((size_t)h->n_u + h->n_p) * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
      h->io_stage.alloc((size_t)h->n_u + h->n_p);
      HIP_CHECK(hipMemcpyAsync(h->io_stage.p, host, ((size_t)h->n_u + h->n_p) * sizeof(double), hipMemcpyHostToDevice, h->stream));
      perm_launch(h, 0, h->io_stage.p, dev, true);
      perm_launch(h, 1, h->io_stage.p + h->n_u, dev + h->off_p, true);
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    return;
  }
  std::vector<double> loc(h->len_blk, 0.0);
  for (int i = 0; i < h->N2; ++i)
    for (int c = 0; c < dim; ++c) loc[(size_t)dim * node_to_internal(h, i) + c] = host[(size_t)dim * (h->goff_u + i) + c];
  for (int i = 0; i < h->NP; ++i) loc[h->off_p + pnode_to_internal(h, i)] = host[(size_t)h->n_u_glob + h->goff_p + i];
  if (with_ghosts) {
    for (size_t g = 0; g < h->ghost_u.size(); ++g)
      for (int c = 0; c < dim; ++c) loc[h->n_u + g * dim + c] = host[(size_t)dim * h->ghost_u[g] + c];
    for (size_t g = 0; g < h->ghost_p.size(); ++g) loc[h->off_p + h->n_p + g] = host[(size_t)h->n_u_glob + h->ghost_p[g]];
  }
  HIP_CHECK(hipMemcpyAsync(dev, loc.data(), loc.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

void vec_to_caller(nsx_handle *h, const double *dev, double *host) {
  const int dim = h->dim;
  if (!h->dist) {
    if (!h->layout_on) {
      HIP_CHECK(hipMemcpyAsync(host, dev, ((size_t)h->n_u + h->n_p) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
      h->io_stage.alloc((size_t)h->n_u + h->n_p);
      perm_launch(h, 0, h->io_stage.p, const_cast<double *>(dev), false);
      perm_launch(h, 1, h->io_stage.p + h->n_u, const_cast<double *>(dev) + h->off_p, false);
      HIP_CHECK(hipMemcpyAsync(host, h->io_stage.p, ((size_t)h->n_u + h->n_p) * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_CHECK(hipStreamSynchronize(h->stream));
    return;
  }
  std::vector<double> loc(h->len_blk);
  HIP_CHECK(hipMemcpyAsync(loc.data(), dev, loc.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  for (int i = 0; i < h->N2; ++i)
    for (int c = 0; c < dim; ++c) host[(size_t)dim * (h->goff_u + i) + c] = loc[(size_t)dim * node_to_internal(h, i) + c];
  for (int i = 0; i < h->NP; ++i) host[(size_t)h->n_u_glob + h->goff_p + i] = loc[h->off_p + pnode_to_internal(h, i)];
}

}  // namespace nsx

#define NSX_TRY(h_)                 \
  if (!(h_)) return NSX_ERR_ARG;    \
  try {
#define NSX_CATCH(h_)                                      \
  }                                                        \
  catch (const nsx::Error &e) {                            \
    (h_)->err = e.msg;                                     \
    return e.code;                                         \
  }                                                        \
  catch (const std::exception &e) {                        \
    (h_)->err = e.what();                                  \
    return NSX_ERR_ARG;                                    \
  }                                                        \
  return NSX_OK;

extern "C" {

const char *nsx_version(void) { return "nsx 0.1 (gfx950)"; }

const char *nsx_last_error(const nsx_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int nsx_create(const nsx_params *p, nsx_handle **out) {
  if (!p || !out) return NSX_ERR_ARG;
  *out = nullptr;
  if (p->dim != 2 && p->dim != 3) {
    g_create_error = "dim must be 2 or 3";
    return NSX_ERR_ARG;
  }
  if (!(p->nu > 0) || !(p->deltat > 0)) {
    g_create_error = "nu and deltat must be positive";
    return NSX_ERR_ARG;
  }
  auto *h = new nsx_handle;
  h->prm = *p;
  h->dim = p->dim;
  if (getenv("NSX_GX_DROP_WG")) h->gx_drop_wg = atoi(getenv("NSX_GX_DROP_WG"));  // fault injection for the tests of the time-out fallbacks
  try {
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev < 1) NSX_THROW(NSX_ERR_HIP, "no HIP device visible: libnsx has no CPU fallback");
    if (p->device < 0 || p->device >= ndev) NSX_THROW(NSX_ERR_ARG, "device %d out of range (%d visible)", p->device, ndev);
    HIP_CHECK(hipSetDevice(p->device));
    HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->scal.alloc(N_SLOTS);
    h->scal.zero(h->stream);
    h->red_partial.alloc((size_t)N_SLOTS * 1024);
    HIP_CHECK(hipHostMalloc((void **)&h->scal_host, N_SLOTS * sizeof(double), hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **)&h->pub_host, (N_SLOTS + 8) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->pub_host, 0, (N_SLOTS + 8) * sizeof(double));
    HIP_CHECK(hipHostGetDevicePointer((void **)&h->pub_dev, h->pub_host, 0));
    h->pub_counter.alloc(1);
    h->pub_counter.zero(h->stream);
  } catch (const nsx::Error &e) {
    g_create_error = e.msg;
    delete h;
    return e.code;
  }
  *out = h;
  return NSX_OK;
}

int nsx_destroy(nsx_handle *h) {
  if (!h) return NSX_OK;
  (void)hipSetDevice(h->prm.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  nsx::comm_destroy(h);
  for (auto &kv : h->prof)
    for (auto &ev : kv.second.pending) {
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
  for (auto *b : h->pool) delete b;
  if (h->scal_host) (void)hipHostFree(h->scal_host);
  if (h->pub_host) (void)hipHostFree(h->pub_host);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->stream_plain) (void)hipStreamDestroy(h->stream_plain);
  delete h;
  return NSX_OK;
}

int nsx_set_tables(nsx_handle *h, int n_q, int n_p2, int n_p1, const double *N2, const double *dN2, const double *N1,
                   const double *w) {
  NSX_TRY(h)
  if (!N2 || !dN2 || !N1 || !w || n_q < 1) NSX_THROW(NSX_ERR_ARG, "null table / n_q < 1");
  const int dim = h->dim;
  if (n_p1 != dim + 1 || n_p2 != (dim == 2 ? 6 : 10))
    NSX_THROW(NSX_ERR_UNSUPPORTED, "only Taylor-Hood P2/P1 on simplices (n_p2=%d n_p1=%d for dim=%d)", dim == 2 ? 6 : 10, dim + 1, dim);
  HIP_CHECK(hipSetDevice(h->prm.device));
  h->n_q = n_q;
  h->np2 = n_p2;
  h->np1 = n_p1;
  h->N2_h.assign(N2, N2 + (size_t)n_q * n_p2);
  h->dN2_h.assign(dN2, dN2 + (size_t)n_q * n_p2 * dim);
  h->N1_h.assign(N1, N1 + (size_t)n_q * n_p1);
  h->w_h.assign(w, w + n_q);
  h->tab_N2.upload(h->N2_h, h->stream);
  h->tab_dN2.upload(h->dN2_h, h->stream);
  h->tab_N1.upload(h->N1_h, h->stream);
  h->tab_w.upload(h->w_h, h->stream);
  {
    std::vector<double> NT((size_t)n_q * n_p2), dNT((size_t)n_q * n_p2 * dim);
    for (int q = 0; q < n_q; ++q)
      for (int a = 0; a < n_p2; ++a) {
        NT[(size_t)a * n_q + q] = N2[(size_t)q * n_p2 + a];
        for (int k = 0; k < dim; ++k) dNT[((size_t)a * n_q + q) * dim + k] = dN2[((size_t)q * n_p2 + a) * dim + k];
      }
    h->tab_N2T.upload(NT, h->stream);
    h->tab_dN2T.upload(dNT, h->stream);
  }
  h->have_tables = true;
  NSX_CATCH(h)
}

}  // extern "C"

namespace {
using namespace nsx;

void truncate_rows(nsx::Csr &g, int n_rows) {
  if (g.n_rows <= n_rows) return;
  g.colind.resize(g.rowptr[n_rows]);
  g.rowptr.resize((size_t)n_rows + 1);
  g.n_rows = n_rows;
}

// scalar connectivity (local node ids) from the FESystem dof table; `loc2`/`loc1` map a global node to its local id
template <class M2, class M1>
void connectivity(nsx_handle *h, int n_cells, int dpc, const int32_t *cell_dofs, int n_u_glob, int n_p_glob, M2 loc2, M1 loc1) {
  const int dim = h->dim, nv = dim + 1, np2 = h->np2, np1 = h->np1;
  h->cell_n2_h.resize((size_t)n_cells * np2);
  h->cell_n1_h.resize((size_t)n_cells * np1);
  for (int c = 0; c < n_cells; ++c) {
    const int32_t *cd = cell_dofs + (size_t)c * dpc;
    for (int a = 0; a < np2; ++a) {
      const int base = a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv);
      const int32_t d0 = cd[base];
      if (d0 < 0 || d0 >= n_u_glob || d0 % dim) NSX_THROW(NSX_ERR_ARG, "cell %d: velocity dof %d breaks the dim*node+c numbering contract", c, d0);
      for (int k = 1; k < dim; ++k)
        if (cd[base + k] != d0 + k) NSX_THROW(NSX_ERR_ARG, "cell %d: velocity components of one node are not consecutive", c);
      const int32_t l = loc2(d0 / dim);
      if (l < 0) NSX_THROW(NSX_ERR_ARG, "cell %d: P2 node %d is neither owned nor a known ghost", c, d0 / dim);
      h->cell_n2_h[(size_t)c * np2 + a] = l;
    }
    for (int v = 0; v < nv; ++v) {
      const int32_t d = cd[(dim + 1) * v + dim] - n_u_glob;
      if (d < 0 || d >= n_p_glob) NSX_THROW(NSX_ERR_ARG, "cell %d: pressure dof out of range", c);
      const int32_t l = loc1(d);
      if (l < 0) NSX_THROW(NSX_ERR_ARG, "cell %d: P1 node %d is neither owned nor a known ghost", c, d);
      h->cell_n1_h[(size_t)c * np1 + v] = l;
    }
  }
}

// everything that follows the connectivity: graphs, geometry, gather maps, buffers, Schur pattern, ILU schedules
void setup_mesh(nsx_handle *h, int n_cells, int n_cells1, const double *cell_coords) {
  using namespace nsx;
  const int dim = h->dim, nv = dim + 1, np2 = h->np2, np1 = h->np1;
  h->n_cells = n_cells;
  h->n_cells1 = n_cells1;
  h->n_u = dim * h->N2;
  h->n_p = h->NP;
  h->g_u = dim * (h->N2_loc - h->N2);
  h->g_p = h->NP_loc - h->NP;
  h->off_p = h->n_u + h->g_u;
  h->len_u = h->n_u + h->g_u;
  h->len_p = h->n_p + h->g_p;
  h->len_blk = h->len_u + h->len_p;
  // graphs (reference NavierStokes3D.cpp:109-124, pressure mass :127-142): owned rows, local (owned + ghost) columns
  const int32_t *c2 = h->cell_n2_h.data(), *c1 = h->cell_n1_h.data();
  h->gA.host = build_graph(n_cells, np2, c2, h->N2_loc, np2, c2, h->N2_loc);
  truncate_rows(h->gA.host, h->N2);
  h->gG.host = build_graph(n_cells, np2, c2, h->N2_loc, np1, c1, h->NP_loc);
  truncate_rows(h->gG.host, h->N2);
  h->gB.host = build_graph(n_cells, np1, c1, h->NP_loc, np2, c2, h->N2_loc);  // all local rows: the Schur product needs ghost rows
  h->gPM.host = build_graph(n_cells, np1, c1, h->NP_loc, np1, c1, h->NP_loc);
  truncate_rows(h->gPM.host, h->NP);
  upload_csr(h, h->gA, true);
  upload_csr(h, h->gG, false);
  upload_csr(h, h->gB, false);
  upload_csr(h, h->gPM, true);
  {  // SoA cell tables + geometry
    std::vector<int32_t> t2((size_t)n_cells * np2), t1((size_t)n_cells * np1);
    for (int c = 0; c < n_cells; ++c) {
      for (int a = 0; a < np2; ++a) t2[(size_t)a * n_cells + c] = c2[(size_t)c * np2 + a];
      for (int v = 0; v < np1; ++v) t1[(size_t)v * n_cells + c] = c1[(size_t)c * np1 + v];
    }
    h->cell_n2.upload(t2, h->stream);
    h->cell_n1.upload(t1, h->stream);
    const int ng = dim * dim + 1;
    std::vector<double> geo((size_t)ng * n_cells);
    for (int c = 0; c < n_cells; ++c) {
      const double *X = cell_coords + (size_t)c * nv * dim;
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
      if (!(std::fabs(det) > 0)) NSX_THROW(NSX_ERR_ARG, "cell %d is degenerate", c);
      for (int k = 0; k < dim; ++k)
        for (int d = 0; d < dim; ++d) geo[(size_t)(k * dim + d) * n_cells + c] = Ji[k][d];
      geo[(size_t)(dim * dim) * n_cells + c] = std::fabs(det);
    }
    h->geo.upload(geo, h->stream);
  }
  // gather maps (cell order ascending inside every list -> same summation order as the reference's cell loop)
  build_gather(h, h->gA.host, n_cells, np2, c2, np2, c2, 1, false, h->gmA);
  build_gather(h, h->gG.host, n_cells, np2, c2, np1, c1, dim, false, h->gmG);   // planes (a*np1+v)*dim
  build_gather(h, h->gB.host, n_cells, np1, c1, np2, c2, dim, true, h->gmB);    // same planes, indexed (v,a)
  build_gather(h, h->gPM.host, n_cells, np1, c1, np1, c1, 1, false, h->gmPM);
  const size_t planes = std::max((size_t)np2 * np2, (size_t)np2 * np1 * dim);
  h->cellbuf.alloc(planes * n_cells);
  const int64_t nA = h->gA.nnz();
  h->vS0.alloc(nA);
  h->vMass.alloc(nA);
  h->vStiff.alloc(nA);
  h->vConv.alloc(nA);
  h->vF.alloc(nA);
  h->luF.alloc(nA);
  h->vG.alloc(h->gG.nnz() * dim);
  h->vB.alloc(h->gB.nnz() * dim);
  h->vPM.alloc(h->gPM.nnz());
  for (auto *v : {&h->sol, &h->sol_owned, &h->prev_sol, &h->rhs}) {
    v->alloc(h->len_blk);
    v->zero(h->stream);
  }
  for (auto *v : {&h->diag_D, &h->diag_D_inv, &h->neg_diag_D_inv, &h->lump_M, &h->schur_w, &h->dirmask}) {
    v->alloc(h->len_u);
    v->zero(h->stream);
  }
  build_schur_graph(h);
  build_row_splits(h);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  h->have_mesh = true;
  h->assembled = false;
  h->prec_ready = false;
  h->schur_valid = false;
  h->bc_cache.clear();
  for (int w = 0; w < 2; ++w) {
    h->caller_graph[w] = nsx::Csr();
    h->caller_pos[w].clear();
  }
}

// The node numbering libnsx works in, the set-up products in that numbering, and the rank tables that go with it.
// Without a layout request: the caller's numbering and the caller's tables.  With one (nsx_set_internal_layout): inside the
// range of every rank of the caller the owned nodes are dealt to virtual ranks (bisection of the cells' centroids, lowest-id
// ownership, first-touch order, colour order: host/layout.hpp) and renumbered rank by rank; ghosts keep their places.
void install_mesh(nsx_handle *h) {
  using namespace nsx;
  const int dim = h->dim, nv = dim + 1, np2 = h->np2, np1 = h->np1;
  const int n_cells = (int)(h->cell_n2_in.size() / np2);
  h->cell_n2_h = h->cell_n2_in;
  h->cell_n1_h = h->cell_n1_in;
  h->layout_on = false;
  h->layout_colours = h->layout_colours_p = 0;
  LayoutOut lay;
  if (h->layout_req_ranks > 0) {
    std::vector<double> cen((size_t)n_cells * dim, 0.0);
    for (int c = 0; c < n_cells; ++c)
      for (int k = 0; k < nv; ++k)
        for (int d = 0; d < dim; ++d) cen[(size_t)c * dim + d] += h->cell_coords_in[((size_t)c * nv + k) * dim + d] / nv;
    LayoutIn in;
    in.dim = dim;
    in.n_cells = n_cells;
    in.np2 = np2;
    in.np1 = np1;
    in.c2 = h->cell_n2_in.data();
    in.c1 = h->cell_n1_in.data();
    in.cen = cen.data();
    in.N2 = h->N2;
    in.NP = h->NP;
    in.N2_all = h->N2_loc;
    in.in_u_ptr = h->in_rank_u_h;
    in.in_p_ptr = h->in_rank_p_h;
    build_layout(in, h->layout_req_ranks, h->layout_req_order, h->layout_req_schur, lay);
    h->perm2_h = lay.perm2;
    h->perm1_h = lay.perm1;
    h->iperm2_h.resize(h->N2);
    h->iperm1_h.resize(h->NP);
    for (int i = 0; i < h->N2; ++i) h->iperm2_h[h->perm2_h[i]] = i;
    for (int i = 0; i < h->NP; ++i) h->iperm1_h[h->perm1_h[i]] = i;
    for (auto &v : h->cell_n2_h)
      if (v < h->N2) v = h->perm2_h[v];
    for (auto &v : h->cell_n1_h)
      if (v < h->NP) v = h->perm1_h[v];
    h->perm2_d.upload(h->perm2_h, h->stream);
    h->perm1_d.upload(h->perm1_h, h->stream);
    h->layout_on = true;
    h->layout_colours = lay.n_colours;
    h->layout_colours_p = lay.n_colours_p;
  }
  setup_mesh(h, n_cells, h->n_cells1, h->cell_coords_in.data());
  if (h->layout_on) {
    h->rank_u_h = lay.u_ptr;
    h->rank_p_h = lay.p_ptr;
    h->sblk_h = lay.schur_ptr;
    if (h->sblk_h.size() == h->rank_p_h.size()) h->sblk_h.clear();  // nothing was merged
  } else {
    h->rank_u_h = h->in_rank_u_h;
    h->rank_p_h = h->in_rank_p_h;
    h->sblk_h = h->in_sblk_h;
  }
  refresh_rank_products(h);
  for (HaloPlan *p : {&h->haloU, &h->haloP}) {
    if (p->send_idx_in.empty()) continue;
    std::vector<int32_t> idx(p->send_idx_in);
    if (h->layout_on)
      for (auto &v : idx) v = (p == &h->haloU ? h->perm2_h : h->perm1_h)[v];
    p->send_idx.upload(idx, h->stream);
  }
  if (getenv("NSX_DEBUG") && h->layout_on)
    fprintf(stderr, "[nsx] internal layout: %d virtual ranks inside %d rank(s) of the caller, %d Schur blocks, %d / %d colours\n", (int)h->rank_u_h.size() - 1,
            (int)h->in_rank_u_h.size() - 1, (int)(h->sblk_h.empty() ? h->rank_p_h.size() : h->sblk_h.size()) - 1, h->layout_colours, h->layout_colours_p);
}

void halo_plan(nsx_handle *h, nsx::HaloPlan &p, int n_own, int goff, const std::vector<int32_t> &ghosts, const int32_t *gpu_ptr,
               int n_nbr, const int32_t *nbr, const int32_t *send_ptr, const int32_t *send_nodes, int ncomp) {
  p.n_own = n_own;
  p.nbr.assign(nbr, nbr + n_nbr);
  p.send_ptr.assign(send_ptr, send_ptr + n_nbr + 1);
  std::vector<int32_t> idx(send_ptr[n_nbr]);
  for (int k = 0; k < send_ptr[n_nbr]; ++k) {
    const int32_t l = send_nodes[k] - goff;
    if (l < 0 || l >= n_own) NSX_THROW(NSX_ERR_ARG, "halo plan: node %d to send is not owned by this rank", send_nodes[k]);
    idx[k] = l;
  }
  p.send_idx_in = idx;  // uploaded by install_mesh, through the internal layout if there is one
  p.sendbuf.alloc((size_t)std::max<int32_t>(1, send_ptr[n_nbr]) * ncomp);
  // ghosts are sorted by global id, i.e. grouped by owner in rank order
  p.recv_ptr.assign((size_t)n_nbr + 1, 0);
  size_t g = 0;
  for (int k = 0; k < n_nbr; ++k) {
    const int r = nbr[k];
    if (k > 0 && nbr[k] <= nbr[k - 1]) NSX_THROW(NSX_ERR_ARG, "halo plan: neighbours must be ascending");
    size_t g1 = g;
    while (g1 < ghosts.size() && ghosts[g1] < gpu_ptr[r + 1]) {
      if (ghosts[g1] < gpu_ptr[r]) NSX_THROW(NSX_ERR_ARG, "halo plan: ghost node %d belongs to a rank that is not a neighbour", ghosts[g1]);
      ++g1;
    }
    p.recv_ptr[k + 1] = (int32_t)g1;
    g = g1;
  }
  if (g != ghosts.size()) NSX_THROW(NSX_ERR_ARG, "halo plan: %zu ghost nodes have no neighbour to come from", ghosts.size() - g);
}

}  // namespace

extern "C" {

int nsx_set_mesh(nsx_handle *h, int n_cells, int dpc, const int32_t *cell_dofs, const double *cell_coords, int n_u, int n_p) {
  NSX_TRY(h)
  if (!h->have_tables) NSX_THROW(NSX_ERR_ARG, "nsx_set_tables must be called before nsx_set_mesh");
  if (!cell_dofs || !cell_coords || n_cells < 1) NSX_THROW(NSX_ERR_ARG, "empty mesh");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int dim = h->dim, nv = dim + 1, nl = dim == 2 ? 3 : 6;
  if (dpc != nv * (dim + 1) + nl * dim) NSX_THROW(NSX_ERR_ARG, "dofs_per_cell %d does not match FESystem(P2^%d,P1)", dpc, dim);
  if (n_u % dim) NSX_THROW(NSX_ERR_ARG, "n_u not a multiple of dim");
  h->dpc = dpc;
  h->dist = false;
  h->rank = 0;
  h->world = 1;
  h->goff_u = h->goff_p = 0;
  h->n_u_glob = n_u;
  h->n_p_glob = n_p;
  h->N2 = h->N2_loc = n_u / dim;
  h->NP = h->NP_loc = n_p;
  h->ghost_u.clear();
  h->ghost_p.clear();
  connectivity(h, n_cells, dpc, cell_dofs, n_u, n_p, [](int32_t g) { return g; }, [](int32_t g) { return g; });
  h->cell_n2_in = h->cell_n2_h;
  h->cell_n1_in = h->cell_n1_h;
  h->cell_coords_in.assign(cell_coords, cell_coords + (size_t)n_cells * nv * dim);
  h->n_cells1 = n_cells;
  h->haloU.send_idx_in.clear();
  h->haloP.send_idx_in.clear();
  default_ranks(h);
  install_mesh(h);
  NSX_CATCH(h)
}

int nsx_set_mesh_distributed(nsx_handle *h, int n_cells, int n_cells1, int dpc, const int32_t *cell_dofs, const double *cell_coords,
                             int n_u_glob, int n_p_glob, int world, int rank, const int32_t *gpu_u_ptr, const int32_t *gpu_p_ptr,
                             int n_nbr, const int32_t *nbr, const int32_t *send_u_ptr, const int32_t *send_u_nodes,
                             const int32_t *send_p_ptr, const int32_t *send_p_nodes) {
  NSX_TRY(h)
  if (!h->have_tables) NSX_THROW(NSX_ERR_ARG, "nsx_set_tables must be called before nsx_set_mesh_distributed");
  if (!cell_dofs || !cell_coords || n_cells < 1 || n_cells1 < 1 || n_cells1 > n_cells) NSX_THROW(NSX_ERR_ARG, "empty mesh");
  if (world < 1 || rank < 0 || rank >= world || !gpu_u_ptr || !gpu_p_ptr) NSX_THROW(NSX_ERR_ARG, "bad rank / ownership table");
  if (n_nbr < 0 || (n_nbr > 0 && (!nbr || !send_u_ptr || !send_p_ptr))) NSX_THROW(NSX_ERR_ARG, "bad halo plan");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int dim = h->dim, nv = dim + 1, nl = dim == 2 ? 3 : 6, np2 = h->np2;
  if (dpc != nv * (dim + 1) + nl * dim) NSX_THROW(NSX_ERR_ARG, "dofs_per_cell %d does not match FESystem(P2^%d,P1)", dpc, dim);
  if (gpu_u_ptr[0] != 0 || gpu_p_ptr[0] != 0 || gpu_u_ptr[world] * dim != n_u_glob || gpu_p_ptr[world] != n_p_glob)
    NSX_THROW(NSX_ERR_ARG, "ownership ranges must cover all global nodes");
  h->dpc = dpc;
  h->dist = world > 1;
  h->rank = rank;
  h->world = world;
  h->goff_u = gpu_u_ptr[rank];
  h->goff_p = gpu_p_ptr[rank];
  h->n_u_glob = n_u_glob;
  h->n_p_glob = n_p_glob;
  h->N2 = gpu_u_ptr[rank + 1] - gpu_u_ptr[rank];
  h->NP = gpu_p_ptr[rank + 1] - gpu_p_ptr[rank];
  // ghost sets = nodes of the local cells that this rank does not own, sorted by global id
  {
    std::vector<int32_t> g2, g1;
    for (int c = 0; c < n_cells; ++c) {
      const int32_t *cd = cell_dofs + (size_t)c * dpc;
      for (int a = 0; a < np2; ++a) {
        const int32_t n = cd[a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv)] / dim;
        if (n < gpu_u_ptr[rank] || n >= gpu_u_ptr[rank + 1]) g2.push_back(n);
      }
      for (int v = 0; v < nv; ++v) {
        const int32_t n = cd[(dim + 1) * v + dim] - n_u_glob;
        if (n < gpu_p_ptr[rank] || n >= gpu_p_ptr[rank + 1]) g1.push_back(n);
      }
    }
    std::sort(g2.begin(), g2.end());
    g2.erase(std::unique(g2.begin(), g2.end()), g2.end());
    std::sort(g1.begin(), g1.end());
    g1.erase(std::unique(g1.begin(), g1.end()), g1.end());
    h->ghost_u = std::move(g2);
    h->ghost_p = std::move(g1);
  }
  h->N2_loc = h->N2 + (int)h->ghost_u.size();
  h->NP_loc = h->NP + (int)h->ghost_p.size();
  auto loc = [](int32_t g, int32_t off, int32_t n_own, const std::vector<int32_t> &ghosts) -> int32_t {
    if (g >= off && g < off + n_own) return g - off;
    auto it = std::lower_bound(ghosts.begin(), ghosts.end(), g);
    return (it != ghosts.end() && *it == g) ? n_own + (int32_t)(it - ghosts.begin()) : -1;
  };
  connectivity(h, n_cells, dpc, cell_dofs, n_u_glob, n_p_glob,
               [&](int32_t g) { return loc(g, h->goff_u, h->N2, h->ghost_u); }, [&](int32_t g) { return loc(g, h->goff_p, h->NP, h->ghost_p); });
  // layer-1 cells must be exactly those touching an owned P2 node (all of them first)
  for (int c = 0; c < n_cells; ++c) {
    bool own = false;
    for (int a = 0; a < np2 && !own; ++a) own = h->cell_n2_h[(size_t)c * np2 + a] < h->N2;
    if (own != (c < n_cells1)) NSX_THROW(NSX_ERR_ARG, "cell %d: layer-1 cells (touching an owned node) must come first", c);
  }
  halo_plan(h, h->haloU, h->N2, h->goff_u, h->ghost_u, gpu_u_ptr, n_nbr, nbr, send_u_ptr, send_u_nodes, dim);
  halo_plan(h, h->haloP, h->NP, h->goff_p, h->ghost_p, gpu_p_ptr, n_nbr, nbr, send_p_ptr, send_p_nodes, 1);
  h->cell_n2_in = h->cell_n2_h;
  h->cell_n1_in = h->cell_n1_h;
  h->cell_coords_in.assign(cell_coords, cell_coords + (size_t)n_cells * nv * dim);
  h->n_cells1 = n_cells1;
  default_ranks(h);
  install_mesh(h);
  NSX_CATCH(h)
}

int nsx_set_ranks(nsx_handle *h, int n_ranks, const int32_t *u_ptr, const int32_t *p_ptr) {
  NSX_TRY(h)
  if (!h->have_mesh) NSX_THROW(NSX_ERR_ARG, "nsx_set_mesh first");
  if (n_ranks < 1 || !u_ptr || !p_ptr) NSX_THROW(NSX_ERR_ARG, "bad rank table");
  if (u_ptr[0] != h->goff_u || p_ptr[0] != h->goff_p || u_ptr[n_ranks] != h->goff_u + h->N2 || p_ptr[n_ranks] != h->goff_p + h->NP)
    NSX_THROW(NSX_ERR_ARG, "rank ranges must cover exactly the nodes owned by this handle");
  for (int r = 0; r < n_ranks; ++r)
    if (u_ptr[r + 1] < u_ptr[r] || p_ptr[r + 1] < p_ptr[r]) NSX_THROW(NSX_ERR_ARG, "rank ranges must be ascending");
  HIP_CHECK(hipSetDevice(h->prm.device));
  h->in_rank_u_h.resize((size_t)n_ranks + 1);
  h->in_rank_p_h.resize((size_t)n_ranks + 1);
  for (int r = 0; r <= n_ranks; ++r) {
    h->in_rank_u_h[r] = u_ptr[r] - h->goff_u;
    h->in_rank_p_h[r] = p_ptr[r] - h->goff_p;
  }
  if (h->layout_req_ranks > 0) {
    install_mesh(h);  // the virtual ranks are a refinement of the caller's: lay the nodes out again inside the new ranges
  } else {
    h->rank_u_h = h->in_rank_u_h;
    h->rank_p_h = h->in_rank_p_h;
    refresh_rank_products(h);
  }
  NSX_CATCH(h)
}

int nsx_set_schur_blocks(nsx_handle *h, int n_blocks, const int32_t *p_ptr) {
  NSX_TRY(h)
  if (!h->have_mesh) NSX_THROW(NSX_ERR_ARG, "nsx_set_mesh first");
  if (n_blocks < 1 || !p_ptr || p_ptr[0] != h->goff_p || p_ptr[n_blocks] != h->goff_p + h->NP) NSX_THROW(NSX_ERR_ARG, "bad Schur block table");
  for (int r = 0; r < n_blocks; ++r)
    if (p_ptr[r + 1] < p_ptr[r]) NSX_THROW(NSX_ERR_ARG, "Schur block ranges must be ascending");
  if (h->layout_req_ranks > 0)
    NSX_THROW(NSX_ERR_ARG, "with an internal layout the Schur ILU blocks are unions of virtual ranks: pass schur_max_rows to nsx_set_internal_layout");
  HIP_CHECK(hipSetDevice(h->prm.device));
  h->in_sblk_h.resize((size_t)n_blocks + 1);
  for (int r = 0; r <= n_blocks; ++r) h->in_sblk_h[r] = p_ptr[r] - h->goff_p;
  h->sblk_h = h->in_sblk_h;
  refresh_rank_products(h);
  NSX_CATCH(h)
}

// ---- state: host vectors use the caller's GLOBAL numbering [n_u_glob | n_p_glob]; a rank reads owned + ghost entries and writes owned ones
static int vec_io(nsx_handle *h, nsx::DevBuf<double> &v, double *out, const double *in) {
  NSX_TRY(h)
  if (!h->have_mesh) NSX_THROW(NSX_ERR_ARG, "nsx_set_mesh first");
  HIP_CHECK(hipSetDevice(h->prm.device));
  if (in) {
    nsx::vec_from_caller(h, v.p, in, true);
  } else {
    if (!out) NSX_THROW(NSX_ERR_ARG, "null output");
    nsx::vec_to_caller(h, v.p, out);
  }
  NSX_CATCH(h)
}

int nsx_set_solution(nsx_handle *h, const double *s) {
  if (!h || !s) return NSX_ERR_ARG;
  int rc = vec_io(h, h->sol_owned, nullptr, s);
  if (rc) return rc;
  return vec_io(h, h->sol, nullptr, s);
}
int nsx_get_solution(nsx_handle *h, double *s) { return h ? vec_io(h, h->sol_owned, s, nullptr) : NSX_ERR_ARG; }
int nsx_get_solution_ghosted(nsx_handle *h, double *s) { return h ? vec_io(h, h->sol, s, nullptr) : NSX_ERR_ARG; }
int nsx_get_rhs(nsx_handle *h, double *s) { return h ? vec_io(h, h->rhs, s, nullptr) : NSX_ERR_ARG; }
int nsx_set_rhs(nsx_handle *h, const double *s) { return (h && s) ? vec_io(h, h->rhs, nullptr, s) : NSX_ERR_ARG; }

// ---- exports: graphs and values in the CALLER's numbering
// scalar velocity graph (0) / Schur graph (1) as the caller numbers their rows and columns, columns sorted, and for every entry its
// position in the internal CSR (identity without an internal layout)
static const nsx::Csr &caller_graph(nsx_handle *h, int which) {
  const nsx::Csr &g = which == 0 ? h->gA.host : h->gS.host;
  if (!h->layout_on) return g;
  if (h->caller_graph[which].n_rows == g.n_rows && (int64_t)h->caller_pos[which].size() == g.nnz()) return h->caller_graph[which];
  const std::vector<int32_t> &perm = which == 0 ? h->perm2_h : h->perm1_h, &iperm = which == 0 ? h->iperm2_h : h->iperm1_h;
  const int n = g.n_rows;
  nsx::Csr c;
  c.n_rows = n;
  c.n_cols = g.n_cols;
  c.rowptr.assign((size_t)n + 1, 0);
  c.colind.resize(g.nnz());
  std::vector<int32_t> &pos = h->caller_pos[which];
  pos.resize(g.nnz());
  std::vector<std::pair<int32_t, int32_t>> row;
  for (int i = 0; i < n; ++i) {
    const int ii = perm[i];
    row.clear();
    for (int k = g.rowptr[ii]; k < g.rowptr[ii + 1]; ++k) row.emplace_back(g.colind[k] < n ? iperm[g.colind[k]] : g.colind[k], k);
    std::sort(row.begin(), row.end());
    c.rowptr[i + 1] = c.rowptr[i] + (int32_t)row.size();
    for (size_t q = 0; q < row.size(); ++q) {
      c.colind[c.rowptr[i] + q] = row[q].first;
      pos[c.rowptr[i] + q] = row[q].second;
    }
  }
  h->caller_graph[which] = std::move(c);
  return h->caller_graph[which];
}
static void values_to_caller(nsx_handle *h, int which, const nsx::DevBuf<double> &src, double *values) {
  const int64_t nnz = which == 0 ? h->gA.nnz() : h->gS.nnz();
  if (!h->layout_on) {
    src.download(values, nnz, h->stream);
    return;
  }
  caller_graph(h, which);
  std::vector<double> v(nnz);
  src.download(v.data(), nnz, h->stream);
  const std::vector<int32_t> &pos = h->caller_pos[which];
  for (int64_t k = 0; k < nnz; ++k) values[k] = v[pos[k]];
}

int nsx_scalar_graph_nnz(nsx_handle *h, int which, int64_t *nnz) {
  if (!h || !nnz || !h->have_mesh || which < 0 || which > 1) return NSX_ERR_ARG;
  *nnz = which == 0 ? h->gA.nnz() : h->gS.nnz();
  return NSX_OK;
}
int nsx_scalar_graph(nsx_handle *h, int which, int32_t *rowptr, int32_t *colind) {
  if (!h || !rowptr || !colind || !h->have_mesh || which < 0 || which > 1) return NSX_ERR_ARG;
  const nsx::Csr &g = caller_graph(h, which);
  std::copy(g.rowptr.begin(), g.rowptr.end(), rowptr);
  std::copy(g.colind.begin(), g.colind.end(), colind);
  return NSX_OK;
}
int nsx_ilu_get(nsx_handle *h, int which, double *values) {
  NSX_TRY(h)
  if (!values || which < 0 || which > 1 || !h->prec_ready) NSX_THROW(NSX_ERR_ARG, "no factors: call nsx_prec_initialize first");
  HIP_CHECK(hipSetDevice(h->prm.device));
  values_to_caller(h, which, which == 0 ? h->luF : h->luS, values);
  NSX_CATCH(h)
}
int nsx_schur_nnz(nsx_handle *h, int64_t *nnz) { return nsx_scalar_graph_nnz(h, 1, nnz); }
int nsx_schur_get(nsx_handle *h, int32_t *rowptr, int32_t *colind, double *values) {
  NSX_TRY(h)
  if (!h->prec_ready) NSX_THROW(NSX_ERR_ARG, "no Schur matrix: call nsx_prec_initialize first");
  HIP_CHECK(hipSetDevice(h->prm.device));
  int rc = nsx_scalar_graph(h, 1, rowptr, colind);
  if (rc) NSX_THROW(rc, "bad arguments");
  values_to_caller(h, 1, h->vSchur, values);
  NSX_CATCH(h)
}

int nsx_export_block(nsx_handle *h, int which, int block, int n_rows, const int32_t *rowptr, const int32_t *colind, double *values) {
  NSX_TRY(h)
  if (!h->assembled) NSX_THROW(NSX_ERR_ARG, "nothing assembled yet");
  if (h->dist) NSX_THROW(NSX_ERR_UNSUPPORTED, "nsx_export_block works on a single-process handle only");
  if (!rowptr || !colind || !values) NSX_THROW(NSX_ERR_ARG, "null graph");
  HIP_CHECK(hipSetDevice(h->prm.device));
  const int dim = h->dim;
  const int64_t nnz = rowptr[n_rows];
  std::fill(values, values + nnz, 0.0);
  // the caller's graph is in the caller's numbering: node ids go through the internal layout, components stay
  auto n2 = [&](int32_t node) { return nsx::node_to_internal(h, node); };
  auto n1 = [&](int32_t node) { return nsx::pnode_to_internal(h, node); };
  if (which == 4) {
    if (block != 3 || n_rows != h->n_p) NSX_THROW(NSX_ERR_ARG, "pressure_mass lives in block (1,1)");
    std::vector<double> v(h->gPM.nnz());
    h->vPM.download(v.data(), v.size(), h->stream);
    for (int i = 0; i < n_rows; ++i)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        if (colind[k] < 0 || colind[k] >= h->n_p) NSX_THROW(NSX_ERR_ARG, "column %d out of range in row %d", colind[k], i);
        const int e = nsx::find_in_row(h->gPM.host, n1(i), n1(colind[k]));
        if (e >= 0) values[k] = v[e];
      }
  } else if (block == 0) {
    if (n_rows != h->n_u) NSX_THROW(NSX_ERR_ARG, "block (0,0) has n_u rows");
    nsx::DevBuf<double> *src = which == 0 ? &h->vF : which == 1 ? &h->vMass : which == 2 ? &h->vConv : which == 3 ? &h->vStiff : nullptr;
    if (!src) NSX_THROW(NSX_ERR_ARG, "bad matrix id");
    std::vector<double> v(h->gA.nnz());
    src->download(v.data(), v.size(), h->stream);
    for (int i = 0; i < n_rows; ++i) {
      const int node = n2(i / dim), c = i % dim;
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        if (colind[k] < 0 || colind[k] >= h->n_u) NSX_THROW(NSX_ERR_ARG, "column %d out of range in row %d", colind[k], i);
        if (colind[k] % dim != c) continue;  // cross-component slots are structural zeros in the reference
        const int e = nsx::find_in_row(h->gA.host, node, n2(colind[k] / dim));
        if (e >= 0) values[k] = v[e];
      }
    }
  } else if (block == 1 || block == 2) {
    if (which != 0) return NSX_OK;  // only system_matrix carries the B blocks (mass/convection/stiffness store zeros there)
    const bool isG = block == 1;
    if (n_rows != (isG ? h->n_u : h->n_p)) NSX_THROW(NSX_ERR_ARG, "row count does not match the block");
    const nsx::DevCsr &g = isG ? h->gG : h->gB;
    std::vector<double> v(g.nnz() * dim);
    (isG ? h->vG : h->vB).download(v.data(), v.size(), h->stream);
    for (int i = 0; i < n_rows; ++i)
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        if (colind[k] < 0 || colind[k] >= (isG ? h->n_p : h->n_u)) NSX_THROW(NSX_ERR_ARG, "column %d out of range in row %d", colind[k], i);
        if (isG) {
          const int e = nsx::find_in_row(g.host, n2(i / dim), n1(colind[k]));
          if (e >= 0) values[k] = v[(size_t)e * dim + i % dim];
        } else {
          const int e = nsx::find_in_row(g.host, n1(i), n2(colind[k] / dim));
          if (e >= 0) values[k] = v[(size_t)e * dim + colind[k] % dim];
        }
      }
  } else {
    NSX_THROW(NSX_ERR_ARG, "bad block id");
  }
  NSX_CATCH(h)
}

// ---- internal layout
int nsx_set_internal_layout(nsx_handle *h, int n_virtual_ranks, int order, int schur_max_rows) {
  NSX_TRY(h)
  if (n_virtual_ranks < 0 || order < NSX_ORDER_FIRST_TOUCH || order > NSX_ORDER_COLOUR_ALL || schur_max_rows < 0)
    NSX_THROW(NSX_ERR_ARG, "bad layout request (ranks %d, order %d, Schur rows %d)", n_virtual_ranks, order, schur_max_rows);
  if (h->layout_req_ranks == n_virtual_ranks && h->layout_req_order == order && h->layout_req_schur == schur_max_rows) return NSX_OK;
  h->layout_req_ranks = n_virtual_ranks;
  h->layout_req_order = order;
  h->layout_req_schur = schur_max_rows;
  if (h->have_mesh) {  // otherwise nsx_set_mesh(_distributed) applies the request
    HIP_CHECK(hipSetDevice(h->prm.device));
    if (n_virtual_ranks > 0) h->in_sblk_h.clear();
    install_mesh(h);
  }
  NSX_CATCH(h)
}
int nsx_layout_info(nsx_handle *h, int info[5]) {
  if (!h || !info || !h->have_mesh) return NSX_ERR_ARG;
  info[0] = h->layout_on ? 1 : 0;
  info[1] = (int)h->rank_u_h.size() - 1;
  info[2] = (int)(h->sblk_h.empty() ? h->rank_p_h.size() : h->sblk_h.size()) - 1;
  info[3] = h->layout_colours;
  info[4] = h->layout_colours_p;
  return NSX_OK;
}
int nsx_layout_get(nsx_handle *h, int32_t *node_perm, int32_t *pnode_perm, int32_t *u_ptr, int32_t *p_ptr, int32_t *schur_ptr) {
  if (!h || !h->have_mesh) return NSX_ERR_ARG;
  if (node_perm)
    for (int i = 0; i < h->N2; ++i) node_perm[i] = h->goff_u + nsx::node_to_internal(h, i);
  if (pnode_perm)
    for (int i = 0; i < h->NP; ++i) pnode_perm[i] = h->goff_p + nsx::pnode_to_internal(h, i);
  if (u_ptr)
    for (size_t r = 0; r < h->rank_u_h.size(); ++r) u_ptr[r] = h->goff_u + h->rank_u_h[r];
  if (p_ptr)
    for (size_t r = 0; r < h->rank_p_h.size(); ++r) p_ptr[r] = h->goff_p + h->rank_p_h[r];
  if (schur_ptr) {
    const std::vector<int32_t> &sb = h->sblk_h.empty() ? h->rank_p_h : h->sblk_h;
    for (size_t r = 0; r < sb.size(); ++r) schur_ptr[r] = h->goff_p + sb[r];
  }
  return NSX_OK;
}

// ---- profiling
int nsx_profile_enable(nsx_handle *h, int on) {
  if (!h) return NSX_ERR_ARG;
  h->prof_on = on != 0;
  return NSX_OK;
}
static void prof_collect(nsx_handle *h) {
  (void)hipSetDevice(h->prm.device);
  (void)hipStreamSynchronize(h->stream);
  for (auto &kv : h->prof) {
    for (auto &ev : kv.second.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) kv.second.ms += ms;
      (void)hipEventDestroy(ev.first);
      (void)hipEventDestroy(ev.second);
    }
    kv.second.pending.clear();
  }
}
int nsx_profile_reset(nsx_handle *h) {
  if (!h) return NSX_ERR_ARG;
  prof_collect(h);
  h->prof.clear();
  return NSX_OK;
}
int nsx_profile_count(nsx_handle *h) {
  if (!h) return NSX_ERR_ARG;
  prof_collect(h);
  h->prof_names.clear();
  for (auto &kv : h->prof) h->prof_names.push_back(kv.first);
  return (int)h->prof_names.size();
}
int nsx_profile_get(nsx_handle *h, int i, const char **name, int64_t *launches, double *total_ms, double *bytes) {
  if (!h || i < 0 || i >= (int)h->prof_names.size()) return NSX_ERR_ARG;
  const auto &e = h->prof[h->prof_names[i]];
  if (name) *name = h->prof_names[i].c_str();
  if (launches) *launches = e.launches;
  if (total_ms) *total_ms = e.ms;
  if (bytes) *bytes = e.launches ? e.bytes / (double)e.launches : 0.0;  // average algorithmic bytes per launch
  return NSX_OK;
}

}  // extern "C"
