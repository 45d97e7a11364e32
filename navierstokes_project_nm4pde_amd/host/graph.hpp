// graph.hpp — small CSR graph helpers shared by the host front-end and the device library's
// host-side setup.  Builds "row entity -> column entity" couplings induced by a cell loop, i.e.
// what DoFTools::make_sparsity_pattern does for one pair of scalar spaces
// (reference Navier-Stokes/src/NavierStokes3D.cpp:109-124).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace nsx {

struct Csr {
  int32_t n_rows = 0, n_cols = 0;
  std::vector<int32_t> rowptr;  // n_rows + 1
  std::vector<int32_t> colind;  // sorted ascending inside a row
  int64_t nnz() const { return rowptr.empty() ? 0 : rowptr.back(); }
};

// entity -> cells incidence (CSR), cells listed in ascending order.
inline void build_incidence(int32_t n_cells, int32_t per_cell, const int32_t *cell_ent, int32_t n_ent,
                            std::vector<int32_t> &ptr, std::vector<int32_t> &cells) {
  ptr.assign(n_ent + 1, 0);
  for (int64_t k = 0; k < (int64_t)n_cells * per_cell; ++k) ptr[cell_ent[k] + 1]++;
  for (int32_t i = 0; i < n_ent; ++i) ptr[i + 1] += ptr[i];
  cells.resize(ptr[n_ent]);
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int32_t c = 0; c < n_cells; ++c)
    for (int32_t a = 0; a < per_cell; ++a) cells[fill[cell_ent[(int64_t)c * per_cell + a]]++] = c;
}

// rows = entities of kind R (per_r per cell), cols = entities of kind C (per_c per cell).
inline Csr build_graph(int32_t n_cells, int32_t per_r, const int32_t *cell_r, int32_t n_r, int32_t per_c,
                       const int32_t *cell_c, int32_t n_c) {
  std::vector<int32_t> ptr, cells;
  build_incidence(n_cells, per_r, cell_r, n_r, ptr, cells);
  Csr g;
  g.n_rows = n_r;
  g.n_cols = n_c;
  g.rowptr.assign(n_r + 1, 0);
  std::vector<int32_t> tmp;
  // pass 1: count
  for (int32_t i = 0; i < n_r; ++i) {
    tmp.clear();
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) {
      const int32_t *cc = cell_c + (int64_t)cells[k] * per_c;
      tmp.insert(tmp.end(), cc, cc + per_c);
    }
    std::sort(tmp.begin(), tmp.end());
    g.rowptr[i + 1] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
  }
  for (int32_t i = 0; i < n_r; ++i) g.rowptr[i + 1] += g.rowptr[i];
  g.colind.resize(g.rowptr[n_r]);
  for (int32_t i = 0; i < n_r; ++i) {
    tmp.clear();
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) {
      const int32_t *cc = cell_c + (int64_t)cells[k] * per_c;
      tmp.insert(tmp.end(), cc, cc + per_c);
    }
    std::sort(tmp.begin(), tmp.end());
    auto e = std::unique(tmp.begin(), tmp.end());
    std::copy(tmp.begin(), e, g.colind.begin() + g.rowptr[i]);
  }
  return g;
}

// position of column j in row i (binary search); -1 if absent.
inline int32_t find_in_row(const Csr &g, int32_t i, int32_t j) {
  const int32_t *b = g.colind.data() + g.rowptr[i], *e = g.colind.data() + g.rowptr[i + 1];
  const int32_t *p = std::lower_bound(b, e, j);
  return (p != e && *p == j) ? (int32_t)(p - g.colind.data()) : -1;
}

}  // namespace nsx
