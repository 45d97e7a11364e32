// layout.hpp — the sub-partition ("virtual MPI ranks") and the node order inside a rank, on plain arrays.
//
// Shared by the host front-end (nsxh_mesh_partition, nsxh_distribute_dofs_ordered: the numbering a caller may choose to
// hand over) and by the device library (nsx_set_internal_layout: the numbering libnsx gives itself behind the C-ABI when
// the caller keeps deal.II's own, reference Navier-Stokes/src/NavierStokes3D.cpp:16-19,58-69).  Both run the SAME code, so a
// serial first-touch numbering + nsx_set_internal_layout(n, NSX_ORDER_COLOUR) is, node for node, the numbering the front-end
// produces for partition(1, n) + NSXH_ORDER_COLOUR (tests/test_layout.py).
//
// The pieces:
//   rcb              recursive coordinate bisection of cell centroids (stands in for METIS, NavierStokes3D.cpp:16)
//   colour_perm      greedy colouring of the P2 nodes of every block, nodes of a block stably sorted by colour: the ILU(0)
//                    dependency graph of a block is then as deep as the number of colours
//   colour_perm_schur the same for the P1 nodes on the graph of the Schur complement B D^-1 B^T
//   merge_blocks     consecutive ranks merged up to a row limit (the Schur ILU blocks)
//   build_layout     cells -> sub-partition inside every real rank's range -> lowest-sub-id ownership (deal.II's rule) ->
//                    first-touch order (cell by cell, vertices then lines) -> colour order
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace nsx {

enum { LAYOUT_FIRST_TOUCH = 0, LAYOUT_COLOUR = 1, LAYOUT_COLOUR_ALL = 2 };

inline void rcb(const std::vector<double> &cen, int dim, std::vector<int32_t> &idx, size_t lo, size_t hi, int nparts, int first,
                std::vector<int32_t> &out) {
  if (nparts <= 1 || hi - lo <= 1) {
    for (size_t k = lo; k < hi; ++k) out[idx[k]] = first;
    return;
  }
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (size_t k = lo; k < hi; ++k)
    for (int d = 0; d < dim; ++d) {
      mn[d] = std::min(mn[d], cen[(size_t)idx[k] * dim + d]);
      mx[d] = std::max(mx[d], cen[(size_t)idx[k] * dim + d]);
    }
  int ax = 0;
  for (int d = 1; d < dim; ++d)
    if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
  const int nl = nparts / 2;
  const size_t mid = lo + (size_t)((double)(hi - lo) * nl / nparts + 0.5);
  std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int32_t a, int32_t b) {
    const double xa = cen[(size_t)a * dim + ax], xb = cen[(size_t)b * dim + ax];
    return xa != xb ? xa < xb : a < b;
  });
  rcb(cen, dim, idx, lo, mid, nl, first, out);
  rcb(cen, dim, idx, mid, hi, nparts - nl, first + nl, out);
}

// entity -> cells incidence restricted to the entities [0, n_ent): ids >= n_ent (ghosts) are skipped
inline void incidence_owned(int32_t n_cells, int32_t per_cell, const int32_t *conn, int32_t n_ent, std::vector<int32_t> &ptr,
                            std::vector<int32_t> &cells) {
  ptr.assign((size_t)n_ent + 1, 0);
  for (int64_t k = 0; k < (int64_t)n_cells * per_cell; ++k)
    if (conn[k] < n_ent) ptr[conn[k] + 1]++;
  for (int32_t i = 0; i < n_ent; ++i) ptr[i + 1] += ptr[i];
  cells.resize(ptr[n_ent]);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int32_t c = 0; c < n_cells; ++c)
    for (int32_t a = 0; a < per_cell; ++a) {
      const int32_t e = conn[(int64_t)c * per_cell + a];
      if (e < n_ent) cells[fill[e]++] = c;
    }
}

// nodes of every block [bptr[b], bptr[b+1]) stably sorted by colour: perm[old] = new
inline void sort_blocks_by_colour(const std::vector<int32_t> &colour, const std::vector<int32_t> &bptr, std::vector<int32_t> &perm) {
  perm.resize(colour.size());
  std::vector<int32_t> idx;
  for (size_t s = 0; s + 1 < bptr.size(); ++s) {
    const int r0 = bptr[s], r1 = bptr[s + 1];
    idx.resize(r1 - r0);
    std::iota(idx.begin(), idx.end(), r0);
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return colour[a] < colour[b]; });
    for (int k = 0; k < r1 - r0; ++k) perm[idx[k]] = r0 + k;
  }
}

// Greedy colouring of the nodes [0, n_own) in index order; two nodes are adjacent when they share a cell AND a block
// (block_of[i], blocks = contiguous index ranges bptr).  Returns the number of colours; perm[old] = new sorts every block by colour.
inline int colour_perm(int32_t n_own, int32_t n_cells, int32_t per_cell, const int32_t *conn, const int32_t *block_of,
                       const std::vector<int32_t> &bptr, std::vector<int32_t> &perm) {
  std::vector<int32_t> nptr, ncell;
  incidence_owned(n_cells, per_cell, conn, n_own, nptr, ncell);
  std::vector<int32_t> colour(n_own, -1);
  std::vector<uint8_t> used;
  int max_col = 0;
  for (int32_t i = 0; i < n_own; ++i) {
    used.assign((size_t)max_col + 2, 0);
    for (int32_t k = nptr[i]; k < nptr[i + 1]; ++k)
      for (int32_t a = 0; a < per_cell; ++a) {
        const int32_t j = conn[(int64_t)ncell[k] * per_cell + a];
        if (j != i && j < n_own && block_of[j] == block_of[i] && colour[j] >= 0) used[colour[j]] = 1;
      }
    int c = 0;
    while (used[c]) ++c;
    colour[i] = c;
    max_col = std::max(max_col, c + 1);
  }
  sort_blocks_by_colour(colour, bptr, perm);
  return max_col;
}

// The same for the P1 nodes [0, n1_own) on the graph of the Schur complement: two pressure nodes are adjacent when some
// P2 node shares a cell with each of them (and they sit in the same block).  conn2 may use any numbering of the P2 nodes
// (n2_all of them, ghosts included): it only serves to find the cells around a P2 node.
inline int colour_perm_schur(int32_t n1_own, int32_t n2_all, int32_t n_cells, int32_t np2, int32_t nv, const int32_t *conn2,
                             const int32_t *conn1, const int32_t *block_of, const std::vector<int32_t> &bptr, std::vector<int32_t> &perm) {
  std::vector<int32_t> pptr, pcell, nptr, ncell;
  incidence_owned(n_cells, nv, conn1, n1_own, pptr, pcell);
  incidence_owned(n_cells, np2, conn2, n2_all, nptr, ncell);
  std::vector<int32_t> colour(n1_own, -1), seen2(n2_all, -1), seenc(n_cells, -1);
  std::vector<uint8_t> used;
  int max_col = 0;
  for (int32_t i = 0; i < n1_own; ++i) {
    used.assign((size_t)max_col + 2, 0);
    for (int32_t k = pptr[i]; k < pptr[i + 1]; ++k)
      for (int32_t a = 0; a < np2; ++a) {
        const int32_t m2 = conn2[(int64_t)pcell[k] * np2 + a];
        if (seen2[m2] == i) continue;
        seen2[m2] = i;
        for (int32_t q = nptr[m2]; q < nptr[m2 + 1]; ++q) {
          const int32_t c2 = ncell[q];
          if (seenc[c2] == i) continue;
          seenc[c2] = i;
          for (int32_t b = 0; b < nv; ++b) {
            const int32_t j = conn1[(int64_t)c2 * nv + b];
            if (j != i && j < n1_own && block_of[j] == block_of[i] && colour[j] >= 0) used[colour[j]] = 1;
          }
        }
      }
    int c = 0;
    while (used[c]) ++c;
    colour[i] = c;
    if (c + 1 > max_col) max_col = c + 1;
  }
  sort_blocks_by_colour(colour, bptr, perm);
  return max_col;
}

// Coarser blocks as unions of CONSECUTIVE ranks: ranks are added to a block while it stays within max_rows rows (a rank
// larger than that stays a block of its own); a block never crosses one of the `fences` (ascending row indices, e.g. the
// boundaries of the real MPI ranks).  ptr: [n+1] row ranges.
inline std::vector<int32_t> merge_blocks(const std::vector<int32_t> &ptr, int max_rows, const std::vector<int32_t> &fences) {
  std::vector<int32_t> out{ptr.front()};
  size_t f = 0;
  for (size_t k = 1; k < ptr.size(); ++k) {
    while (f < fences.size() && fences[f] <= out.back()) ++f;
    const bool fence = f < fences.size() && fences[f] < ptr[k];  // a fence inside (out.back(), ptr[k])
    if ((ptr[k] - out.back() > max_rows || fence) && ptr[k - 1] > out.back()) out.push_back(ptr[k - 1]);
  }
  if (out.back() != ptr.back()) out.push_back(ptr.back());
  return out;
}

struct LayoutIn {
  int dim = 0, n_cells = 0, np2 = 0, np1 = 0;
  const int32_t *c2 = nullptr;  // [n_cells][np2] P2 nodes of every cell, local ids: owned < N2 <= ghosts; vertices first, then lines
  const int32_t *c1 = nullptr;  // [n_cells][np1] P1 nodes, owned < NP <= ghosts
  const double *cen = nullptr;  // [n_cells][dim] centroids
  int N2 = 0, NP = 0, N2_all = 0;               // owned counts; N2_all = owned + ghost P2 nodes
  std::vector<int32_t> in_u_ptr, in_p_ptr;      // the caller's (real) ranks: owned node ranges, local ids
};

struct LayoutOut {
  std::vector<int32_t> perm2, perm1;   // caller-local owned node -> internal node
  std::vector<int32_t> u_ptr, p_ptr;   // [n_sub+1] internal node ranges of the virtual ranks (a refinement of the caller's ranks)
  std::vector<int32_t> schur_ptr;      // Schur ILU blocks (unions of consecutive virtual ranks); empty: the ranks themselves
  std::vector<int32_t> sub_of_cell;    // [n_cells] virtual rank of every cell (-1: touches no owned node)
  int n_colours = 0, n_colours_p = 0;
};

// n_virtual: virtual ranks over the whole handle, dealt to the caller's ranks in proportion to their owned P2 nodes.
inline void build_layout(const LayoutIn &in, int n_virtual, int order, int schur_max_rows, LayoutOut &out) {
  const int dim = in.dim, nc = in.n_cells, np2 = in.np2, nv = in.np1, N2 = in.N2, NP = in.NP;
  const int R = (int)in.in_u_ptr.size() - 1;
  // ---- real rank of every owned node and of every cell (= the highest rank among its owned nodes: deal.II hands a node to
  //      the LOWEST subdomain touching it, so the cells of subdomain r are exactly those whose highest node owner is r)
  std::vector<int32_t> rk2(N2), rk1(NP), rr(nc, -1);
  for (int r = 0; r < R; ++r) {
    for (int i = in.in_u_ptr[r]; i < in.in_u_ptr[r + 1]; ++i) rk2[i] = r;
    for (int i = in.in_p_ptr[r]; i < in.in_p_ptr[r + 1]; ++i) rk1[i] = r;
  }
  for (int c = 0; c < nc; ++c)
    for (int a = 0; a < np2; ++a) {
      const int32_t i = in.c2[(size_t)c * np2 + a];
      if (i < N2) rr[c] = std::max(rr[c], rk2[i]);
    }
  // ---- sub-partition of every real rank's cells
  std::vector<std::vector<int32_t>> members(R);
  for (int c = 0; c < nc; ++c)
    if (rr[c] >= 0) members[rr[c]].push_back(c);
  std::vector<double> cen(in.cen, in.cen + (size_t)nc * dim);
  std::vector<int32_t> first(R + 1, 0);
  out.sub_of_cell.assign(nc, -1);
  for (int r = 0; r < R; ++r) {
    const int64_t n_r = in.in_u_ptr[r + 1] - in.in_u_ptr[r];
    int k = (int)((double)n_virtual * (double)n_r / (double)std::max(1, N2) + 0.5);
    k = std::max(1, std::min<int>(k, std::max<size_t>(1, members[r].size())));
    first[r + 1] = first[r] + k;
    rcb(cen, dim, members[r], 0, members[r].size(), k, first[r], out.sub_of_cell);
  }
  const int n_sub = first[R];
  // ---- ownership: the lowest virtual rank among the cells of the node's own real rank that touch it
  auto owners = [&](int n_own, int per, const int32_t *conn, const std::vector<int32_t> &rk, std::vector<int32_t> &own) {
    own.assign(n_own, INT32_MAX);
    for (int c = 0; c < nc; ++c) {
      if (rr[c] < 0) continue;
      for (int a = 0; a < per; ++a) {
        const int32_t i = conn[(size_t)c * per + a];
        if (i < n_own && rk[i] == rr[c]) own[i] = std::min(own[i], out.sub_of_cell[c]);
      }
    }
    for (int i = 0; i < n_own; ++i)
      if (own[i] == INT32_MAX) own[i] = first[rk[i]];  // no cell of its own rank touches it (not a deal.II numbering): first virtual rank of its range
  };
  std::vector<int32_t> own2, own1;
  owners(N2, np2, in.c2, rk2, own2);
  owners(NP, nv, in.c1, rk1, own1);
  // ---- first-touch order inside a virtual rank: cells by (virtual rank, cell index), per cell vertices then lines
  std::vector<int32_t> order_c;
  order_c.reserve(nc);
  for (int c = 0; c < nc; ++c)
    if (rr[c] >= 0) order_c.push_back(c);
  std::stable_sort(order_c.begin(), order_c.end(), [&](int32_t a, int32_t b) { return out.sub_of_cell[a] < out.sub_of_cell[b]; });
  std::vector<int64_t> key2(N2, -1), key1(NP, -1);
  int64_t t2 = 0, t1 = 0;
  for (int32_t c : order_c) {
    const int s = out.sub_of_cell[c];
    for (int a = 0; a < np2; ++a) {
      const int32_t i = in.c2[(size_t)c * np2 + a];
      if (i < N2 && own2[i] == s && key2[i] < 0) key2[i] = t2++;
    }
    for (int v = 0; v < nv; ++v) {
      const int32_t j = in.c1[(size_t)c * nv + v];
      if (j < NP && own1[j] == s && key1[j] < 0) key1[j] = t1++;
    }
  }
  auto number = [&](int n_own, const std::vector<int32_t> &own, std::vector<int64_t> &key, int64_t t, std::vector<int32_t> &ft, std::vector<int32_t> &ptr) {
    for (int i = 0; i < n_own; ++i)
      if (key[i] < 0) key[i] = t++;
    std::vector<int32_t> idx(n_own);
    std::iota(idx.begin(), idx.end(), 0);
    std::sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return own[a] != own[b] ? own[a] < own[b] : key[a] < key[b]; });
    ft.resize(n_own);
    ptr.assign((size_t)n_sub + 1, 0);
    for (int k = 0; k < n_own; ++k) {
      ft[idx[k]] = k;
      ptr[own[idx[k]] + 1]++;
    }
    for (int s = 0; s < n_sub; ++s) ptr[s + 1] += ptr[s];
  };
  std::vector<int32_t> ft2, ft1;
  number(N2, own2, key2, t2, ft2, out.u_ptr);
  number(NP, own1, key1, t1, ft1, out.p_ptr);
  out.perm2 = ft2;
  out.perm1 = ft1;
  out.n_colours = out.n_colours_p = 0;
  // ---- colour order inside a virtual rank
  if (order == LAYOUT_COLOUR || order == LAYOUT_COLOUR_ALL) {
    std::vector<int32_t> conn((size_t)nc * np2), blk(N2), perm;
    for (size_t k = 0; k < conn.size(); ++k) conn[k] = in.c2[k] < N2 ? ft2[in.c2[k]] : in.c2[k];
    for (int i = 0; i < N2; ++i) blk[ft2[i]] = own2[i];
    out.n_colours = colour_perm(N2, nc, np2, conn.data(), blk.data(), out.u_ptr, perm);
    for (int i = 0; i < N2; ++i) out.perm2[i] = perm[ft2[i]];
  }
  if (order == LAYOUT_COLOUR_ALL) {
    std::vector<int32_t> conn((size_t)nc * nv), blk(NP), perm;
    for (size_t k = 0; k < conn.size(); ++k) conn[k] = in.c1[k] < NP ? ft1[in.c1[k]] : in.c1[k];
    for (int i = 0; i < NP; ++i) blk[ft1[i]] = own1[i];
    out.n_colours_p = colour_perm_schur(NP, in.N2_all, nc, np2, nv, in.c2, conn.data(), blk.data(), out.p_ptr, perm);
    for (int i = 0; i < NP; ++i) out.perm1[i] = perm[ft1[i]];
  }
  // ---- Schur ILU blocks: consecutive virtual ranks merged up to schur_max_rows pressure rows.  A block may span the boundary
  // between two ranks of the caller that live on ONE handle (a handle is one GPU: nothing crosses GPUs): forcing a cut at every
  // such boundary costs a partial block each, and 513 instead of 511 blocks at 1.09 M DoF / 8 ranks no longer fit the 512
  // resident workgroups of the persistent Schur CG (one launch per solve -> five per iteration, +0.5 ms per outer iteration)
  out.schur_ptr.clear();
  if (schur_max_rows > 0) out.schur_ptr = merge_blocks(out.p_ptr, schur_max_rows, std::vector<int32_t>());
}

}  // namespace nsx
