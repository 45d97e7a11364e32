// frontend.cpp — host-side mesh / DoF / reference-element front-end (C API in include/nsx_host.h).
//
// Re-creates, on plain arrays, what the reference obtains from deal.II in NavierStokes::setup()
// (reference Navier-Stokes/src/NavierStokes3D.cpp:2-157; 2D: NavierStokes2D.cpp:2-157;
// convergence: Convergence3D.cpp:28-181).  gmsh/METIS/deal.II are not available, so:
//   * meshes come from a block-structured generator using the constants of mesh/*.geo,
//   * partitioning is recursive coordinate bisection,
//   * DoF numbering follows the deal.II rules restated in DESIGN.md (first-touch per cell:
//     vertices then lines; subdomain-major; component_wise by block).
#include "../../include/nsx_host.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include <sys/stat.h>

#include "graph.hpp"
#include "ilu_stream.hpp"
#include "layout.hpp"

namespace {

// deal.II reference-cell line numbering (ReferenceCells::Triangle / Tetrahedron).
const int TRI_LINES[3][2] = {{0, 1}, {1, 2}, {2, 0}};
const int TET_LINES[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}};
// deal.II reference-cell face numbering.
const int TRI_FACES[3][2] = {{0, 1}, {1, 2}, {2, 0}};
const int TET_FACES[4][3] = {{0, 1, 2}, {1, 0, 3}, {0, 2, 3}, {2, 1, 3}};

inline uint64_t edge_key(int a, int b) {
  if (a > b) std::swap(a, b);
  return ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
}

}  // namespace

struct nsxh_mesh {
  int dim = 0;
  std::vector<double> vertices;
  std::vector<int32_t> cells;
  std::vector<int32_t> bfaces, bface_ids, bface_cells;
  std::vector<int32_t> subdomain;
  int n_subdomains = 1;
  int n_vertices() const { return (int)(vertices.size() / dim); }
  int n_cells() const { return (int)(cells.size() / (dim + 1)); }
  int n_bfaces() const { return (int)bface_ids.size(); }
};

namespace {

double simplex_measure_signed(const nsxh_mesh &m, const int32_t *c) {
  const double *v = m.vertices.data();
  if (m.dim == 2) {
    const double *p0 = v + 2 * c[0], *p1 = v + 2 * c[1], *p2 = v + 2 * c[2];
    return 0.5 * ((p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1]));
  }
  const double *p0 = v + 3 * c[0], *p1 = v + 3 * c[1], *p2 = v + 3 * c[2], *p3 = v + 3 * c[3];
  double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  double b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  double d[3] = {p3[0] - p0[0], p3[1] - p0[1], p3[2] - p0[2]};
  return (a[0] * (b[1] * d[2] - b[2] * d[1]) - a[1] * (b[0] * d[2] - b[2] * d[0]) + a[2] * (b[0] * d[1] - b[1] * d[0])) / 6.0;
}

void orient_cells(nsxh_mesh &m) {
  const int nv = m.dim + 1;
  for (int c = 0; c < m.n_cells(); ++c) {
    int32_t *cc = m.cells.data() + (size_t)c * nv;
    if (simplex_measure_signed(m, cc) < 0) std::swap(cc[nv - 2], cc[nv - 1]);
  }
}

// classify boundary faces with a user functor on the face's vertex coordinates.
template <class F>
void extract_boundary(nsxh_mesh &m, F classify) {
  const int dim = m.dim, nv = dim + 1, nf = dim + 1;
  struct FaceRec {
    std::array<int32_t, 3> key;
    int32_t cell, lf;
  };
  std::vector<FaceRec> recs;
  recs.reserve((size_t)m.n_cells() * nf);
  for (int c = 0; c < m.n_cells(); ++c) {
    const int32_t *cc = m.cells.data() + (size_t)c * nv;
    for (int f = 0; f < nf; ++f) {
      FaceRec r;
      r.key = {-1, -1, -1};
      for (int k = 0; k < dim; ++k) r.key[k] = cc[dim == 2 ? TRI_FACES[f][k] : TET_FACES[f][k]];
      std::sort(r.key.begin(), r.key.begin() + dim);
      r.cell = c;
      r.lf = f;
      recs.push_back(r);
    }
  }
  std::sort(recs.begin(), recs.end(), [](const FaceRec &a, const FaceRec &b) {
    if (a.key != b.key) return a.key < b.key;
    return a.cell < b.cell;
  });
  struct BF {
    int32_t cell, lf;
  };
  std::vector<BF> bf;
  for (size_t i = 0; i < recs.size();) {
    size_t j = i + 1;
    while (j < recs.size() && recs[j].key == recs[i].key) ++j;
    if (j - i == 1) bf.push_back({recs[i].cell, recs[i].lf});
    i = j;
  }
  // keep mesh (cell, face) order, as a deal.II cell/face loop would visit them
  std::sort(bf.begin(), bf.end(), [](const BF &a, const BF &b) { return a.cell != b.cell ? a.cell < b.cell : a.lf < b.lf; });
  m.bfaces.clear();
  m.bface_ids.clear();
  m.bface_cells.clear();
  for (auto &b : bf) {
    const int32_t *cc = m.cells.data() + (size_t)b.cell * nv;
    int32_t fv[3];
    double xs[9];
    for (int k = 0; k < dim; ++k) {
      fv[k] = cc[dim == 2 ? TRI_FACES[b.lf][k] : TET_FACES[b.lf][k]];
      for (int d = 0; d < dim; ++d) xs[k * dim + d] = m.vertices[(size_t)fv[k] * dim + d];
    }
    for (int k = 0; k < dim; ++k) m.bfaces.push_back(fv[k]);
    m.bface_ids.push_back(classify(xs));
    m.bface_cells.push_back(b.cell);
  }
}

// ---- 2D building blocks: vertices + triangles ----
struct Tri2D {
  std::vector<double> xy;
  std::vector<int32_t> tri;
  int nv() const { return (int)xy.size() / 2; }
};

void add_quad(Tri2D &t, int v0, int v1, int v2, int v3) {
  auto d2 = [&](int a, int b) {
    double dx = t.xy[2 * a] - t.xy[2 * b], dy = t.xy[2 * a + 1] - t.xy[2 * b + 1];
    return dx * dx + dy * dy;
  };
  if (d2(v0, v2) <= d2(v1, v3) * (1 + 1e-12)) {
    t.tri.insert(t.tri.end(), {v0, v1, v2, v0, v2, v3});
  } else {
    t.tri.insert(t.tri.end(), {v0, v1, v3, v1, v2, v3});
  }
}

std::vector<double> graded(double a, double b, int n, double ratio) {
  // n cells on [a,b]; last/first cell size = ratio (geometric).
  std::vector<double> x(n + 1);
  if (n == 1 || std::fabs(ratio - 1.0) < 1e-12) {
    for (int i = 0; i <= n; ++i) x[i] = a + (b - a) * i / n;
    return x;
  }
  const double q = std::pow(ratio, 1.0 / (n - 1));
  double h0 = (b - a) * (q - 1) / (std::pow(q, n) - 1), h = h0;
  x[0] = a;
  for (int i = 1; i <= n; ++i) {
    x[i] = x[i - 1] + h;
    h *= q;
  }
  x[n] = b;
  return x;
}

Tri2D channel_2d(double L, double H, double xc, double yc, double R, double a, int m, int nr, int nxu, int nxd, int nyb,
                 int nyt, double grade_x, double grade_r) {
  Tri2D t;
  const int nx = nxu + m + nxd, ny = nyb + m + nyt;
  const int i0 = nxu, i1 = nxu + m, j0 = nyb, j1 = nyb + m;
  std::vector<double> xs, ys;
  {
    auto u = graded(0, xc - a, nxu, 1.0), s = graded(xc - a, xc + a, m, 1.0), d = graded(xc + a, L, nxd, grade_x);
    xs.insert(xs.end(), u.begin(), u.end());
    xs.insert(xs.end(), s.begin() + 1, s.end());
    xs.insert(xs.end(), d.begin() + 1, d.end());
    auto b = graded(0, yc - a, nyb, 1.0), sy = graded(yc - a, yc + a, m, 1.0), tp = graded(yc + a, H, nyt, 1.0);
    ys.insert(ys.end(), b.begin(), b.end());
    ys.insert(ys.end(), sy.begin() + 1, sy.end());
    ys.insert(ys.end(), tp.begin() + 1, tp.end());
  }
  std::vector<int32_t> cart((size_t)(nx + 1) * (ny + 1), -1);
  auto cid = [&](int i, int j) -> int32_t & { return cart[(size_t)j * (nx + 1) + i]; };
  for (int j = 0; j <= ny; ++j)
    for (int i = 0; i <= nx; ++i) {
      if (i > i0 && i < i1 && j > j0 && j < j1) continue;
      cid(i, j) = t.nv();
      t.xy.push_back(xs[i]);
      t.xy.push_back(ys[j]);
    }
  // perimeter walk, counter-clockwise from the bottom-left corner of the square
  const int nt = 4 * m;
  std::vector<int32_t> per(nt);
  for (int k = 0; k < nt; ++k) {
    int i, j;
    if (k < m) { i = i0 + k; j = j0; }
    else if (k < 2 * m) { i = i1; j = j0 + (k - m); }
    else if (k < 3 * m) { i = i1 - (k - 2 * m); j = j1; }
    else { i = i0; j = j1 - (k - 3 * m); }
    per[k] = cid(i, j);
  }
  // radial parameter: s=0 on the circle, s=1 on the square
  std::vector<double> s = graded(0, 1, nr, grade_r);
  std::vector<int32_t> ring((size_t)(nr + 1) * nt);
  for (int k = 0; k < nt; ++k) ring[(size_t)nr * nt + k] = per[k];
  for (int r = 0; r < nr; ++r)
    for (int k = 0; k < nt; ++k) {
      const double px = t.xy[2 * per[k]], py = t.xy[2 * per[k] + 1];
      double th_act = std::atan2(py - yc, px - xc);
      double th_uni = -0.75 * M_PI + 0.5 * M_PI * k / m;
      // unwrap actual angle next to the uniform one
      while (th_act - th_uni > M_PI) th_act -= 2 * M_PI;
      while (th_act - th_uni < -M_PI) th_act += 2 * M_PI;
      const double th = 0.5 * (th_act + th_uni);
      const double cx = xc + R * std::cos(th), cy = yc + R * std::sin(th);
      ring[(size_t)r * nt + k] = t.nv();
      t.xy.push_back(cx + s[r] * (px - cx));
      t.xy.push_back(cy + s[r] * (py - cy));
    }
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      if (i >= i0 && i < i1 && j >= j0 && j < j1) continue;
      add_quad(t, cid(i, j), cid(i + 1, j), cid(i + 1, j + 1), cid(i, j + 1));
    }
  for (int r = 0; r < nr; ++r)
    for (int k = 0; k < nt; ++k) {
      const int k1 = (k + 1) % nt;
      add_quad(t, ring[(size_t)r * nt + k], ring[(size_t)(r + 1) * nt + k], ring[(size_t)(r + 1) * nt + k1],
               ring[(size_t)r * nt + k1]);
    }
  return t;
}

Tri2D box_2d(const std::vector<double> &xs, const std::vector<double> &ys) {
  Tri2D t;
  const int nx = (int)xs.size() - 1, ny = (int)ys.size() - 1;
  for (int j = 0; j <= ny; ++j)
    for (int i = 0; i <= nx; ++i) {
      t.xy.push_back(xs[i]);
      t.xy.push_back(ys[j]);
    }
  auto id = [&](int i, int j) { return j * (nx + 1) + i; };
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i) {
      // alternate diagonals (union-jack) so the mesh has no preferred direction
      if ((i + j) % 2 == 0)
        t.tri.insert(t.tri.end(), {id(i, j), id(i + 1, j), id(i + 1, j + 1), id(i, j), id(i + 1, j + 1), id(i, j + 1)});
      else
        t.tri.insert(t.tri.end(), {id(i, j), id(i + 1, j), id(i, j + 1), id(i + 1, j), id(i + 1, j + 1), id(i, j + 1)});
    }
  return t;
}

// extrude triangles into prisms, prisms into 3 tets with the smallest-index rule
// (Dompierre, Labbe, Vallet, Camarero 1999) so that quadrilateral faces are split consistently.
void extrude(const Tri2D &t, const std::vector<double> &zs, nsxh_mesh &m) {
  m.dim = 3;
  const int nv2 = t.nv(), nz = (int)zs.size() - 1;
  m.vertices.resize((size_t)nv2 * (nz + 1) * 3);
  for (int k = 0; k <= nz; ++k)
    for (int v = 0; v < nv2; ++v) {
      double *p = m.vertices.data() + ((size_t)k * nv2 + v) * 3;
      p[0] = t.xy[2 * v];
      p[1] = t.xy[2 * v + 1];
      p[2] = zs[k];
    }
  const int ntri = (int)t.tri.size() / 3;
  m.cells.reserve((size_t)ntri * nz * 12);
  for (int k = 0; k < nz; ++k)
    for (int e = 0; e < ntri; ++e) {
      int32_t V[6];
      for (int a = 0; a < 3; ++a) {
        V[a] = k * nv2 + t.tri[3 * e + a];
        V[3 + a] = (k + 1) * nv2 + t.tri[3 * e + a];
      }
      int p = (int)(std::min_element(V, V + 6) - V);
      int32_t W[6];
      if (p < 3) {
        for (int a = 0; a < 3; ++a) {
          W[a] = V[(p + a) % 3];
          W[3 + a] = V[3 + (p + a) % 3];
        }
      } else {
        p -= 3;
        for (int a = 0; a < 3; ++a) {
          W[a] = V[3 + (p + a) % 3];
          W[3 + a] = V[(p + a) % 3];
        }
      }
      const int32_t V1 = W[0], V2 = W[1], V3 = W[2], V4 = W[3], V5 = W[4], V6 = W[5];
      if (std::min(V2, V6) < std::min(V3, V5))
        m.cells.insert(m.cells.end(), {V1, V2, V3, V6, V1, V2, V6, V5, V1, V5, V6, V4});
      else
        m.cells.insert(m.cells.end(), {V1, V2, V3, V5, V1, V5, V3, V6, V1, V5, V6, V4});
    }
}

void from_2d(const Tri2D &t, nsxh_mesh &m) {
  m.dim = 2;
  m.vertices = t.xy;
  m.cells = t.tri;
}

void finish_mesh(nsxh_mesh &m) {
  orient_cells(m);
  m.subdomain.assign(m.n_cells(), 0);
  m.n_subdomains = 1;
}

using nsx::rcb;  // host/layout.hpp: the bisection libnsx runs behind nsx_set_internal_layout as well

// Recursive coordinate bisection that balances what a subdomain will OWN, not how many cells it has.  deal.II gives an
// interface node to the lowest subdomain id touching it, so with equal cell counts the low ids own far more nodes than the
// high ones (175 against a mean of 85 rows per ILU block at 4096 subdomains: the largest block sets the time of the
// wave-per-block triangular solve).  The recursion visits subdomains in id order; `claimed` marks the vertices owned by the
// subdomains finished so far.  At every cut the cells are sorted along the longest axis and the cut is placed where the
// lower half has claimed its share of the P2 nodes (vertices and edges: the "entities" of a cell) that are still free.
void rcb_owned(const std::vector<double> &cen, int dim, int nv /* entities per cell */, const std::vector<int32_t> &cells /* cell -> entities */,
               std::vector<int32_t> &idx, size_t lo, size_t hi,
               int nparts, int first, std::vector<int32_t> &out, std::vector<char> &claimed, std::vector<int32_t> &stamp, int32_t &stamp_id) {
  if (nparts <= 1 || hi - lo <= 1) {
    for (size_t k = lo; k < hi; ++k) {
      out[idx[k]] = first;
      for (int v = 0; v < nv; ++v) claimed[cells[(size_t)idx[k] * nv + v]] = 1;
    }
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
  std::sort(idx.begin() + lo, idx.begin() + hi, [&](int32_t a, int32_t b) {
    const double xa = cen[(size_t)a * dim + ax], xb = cen[(size_t)b * dim + ax];
    return xa != xb ? xa < xb : a < b;
  });
  // free vertices first touched by the k-th cell of the sweep
  ++stamp_id;
  std::vector<int32_t> fresh(hi - lo, 0);
  int64_t total = 0;
  for (size_t k = lo; k < hi; ++k)
    for (int v = 0; v < nv; ++v) {
      const int32_t x = cells[(size_t)idx[k] * nv + v];
      if (!claimed[x] && stamp[x] != stamp_id) {
        stamp[x] = stamp_id;
        ++fresh[k - lo];
        ++total;
      }
    }
  const int nl = nparts / 2;
  const double want = (double)total * nl / nparts;
  size_t mid = lo;
  int64_t got = 0;
  while (mid < hi && (double)got < want) got += fresh[mid++ - lo];
  // every subdomain keeps at least one cell
  mid = std::max(mid, lo + (size_t)nl);
  mid = std::min(mid, hi - (size_t)(nparts - nl));
  rcb_owned(cen, dim, nv, cells, idx, lo, mid, nl, first, out, claimed, stamp, stamp_id);
  rcb_owned(cen, dim, nv, cells, idx, mid, hi, nparts - nl, first + nl, out, claimed, stamp, stamp_id);
}

}  // namespace

// ---------------------------------------------------------------- mesh C API
extern "C" {

nsxh_mesh *nsxh_mesh_cylinder(int dim, int m, int nr, int nxu, int nxd, int nyb, int nyt, int nz, double grade_x,
                              double grade_r) {
  if ((dim != 2 && dim != 3) || m < 2 || (m % 2) || nr < 1 || nxu < 1 || nxd < 1 || nyb < 1 || nyt < 1) return nullptr;
  if (dim == 3 && nz < 1) return nullptr;
  const double H = 0.41, R = 0.05, a = 0.1;
  const double L = dim == 2 ? 2.2 : 2.5, xc = dim == 2 ? 0.2 : 0.5, yc = 0.2;
  Tri2D t = channel_2d(L, H, xc, yc, R, a, m, nr, nxu, nxd, nyb, nyt, grade_x, grade_r);
  auto *mesh = new nsxh_mesh;
  if (dim == 2)
    from_2d(t, *mesh);
  else
    extrude(t, graded(0, H, nz, 1.0), *mesh);
  finish_mesh(*mesh);
  const double tol = 1e-9;
  extract_boundary(*mesh, [&](const double *xs) {
    bool x0 = true, x1 = true, cyl = true;
    for (int k = 0; k < dim; ++k) {
      const double x = xs[k * dim], y = xs[k * dim + 1];
      x0 = x0 && std::fabs(x) < tol;
      x1 = x1 && std::fabs(x - L) < tol;
      cyl = cyl && std::fabs(std::hypot(x - xc, y - yc) - R) < 1e-7;
    }
    return x0 ? 0 : x1 ? 1 : cyl ? 3 : 2;
  });
  return mesh;
}

nsxh_mesh *nsxh_mesh_cylinder_level(int dim, int level) {
  if (level < 1) level = 1;
  const int m = 2 * (1 + level);              // cells per square side
  const int nr = 1 + level;                   // radial layers
  const int nyb = std::max(1, level), nyt = std::max(1, level);
  if (dim == 2) {
    const int nxu = std::max(1, level), nxd = 6 + 7 * level;
    return nsxh_mesh_cylinder(2, m, nr, nxu, nxd, nyb, nyt, 0, 2.0, 2.0);
  }
  const int nxu = 1 + 2 * level, nxd = 4 + 6 * level, nz = 2 + 2 * level;
  return nsxh_mesh_cylinder(3, m, nr, nxu, nxd, nyb, nyt, nz, 2.5, 2.0);
}

nsxh_mesh *nsxh_mesh_box(int dim, int nx, int ny, int nz, const double *lo, const double *hi) {
  if ((dim != 2 && dim != 3) || nx < 1 || ny < 1 || (dim == 3 && nz < 1)) return nullptr;
  Tri2D t = box_2d(graded(lo[0], hi[0], nx, 1.0), graded(lo[1], hi[1], ny, 1.0));
  auto *mesh = new nsxh_mesh;
  if (dim == 2)
    from_2d(t, *mesh);
  else
    extrude(t, graded(lo[2], hi[2], nz, 1.0), *mesh);
  finish_mesh(*mesh);
  const double tol = 1e-9 * (1 + std::fabs(hi[0] - lo[0]));
  const double x0v = lo[0], x1v = hi[0];
  extract_boundary(*mesh, [&](const double *xs) {
    bool x0 = true, x1 = true;
    for (int k = 0; k < dim; ++k) {
      x0 = x0 && std::fabs(xs[k * dim] - x0v) < tol;
      x1 = x1 && std::fabs(xs[k * dim] - x1v) < tol;
    }
    return x0 ? 0 : x1 ? 1 : 2;
  });
  return mesh;
}

nsxh_mesh *nsxh_mesh_cube(int n) {
  if (n < 1) return nullptr;
  const double lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1};
  nsxh_mesh *mesh = nsxh_mesh_box(3, n, n, n, lo, hi);
  const double tol = 1e-9;
  extract_boundary(*mesh, [&](const double *xs) {
    auto all = [&](int d, double v) {
      for (int k = 0; k < 3; ++k)
        if (std::fabs(xs[k * 3 + d] - v) > tol) return false;
      return true;
    };
    if (all(0, -1)) return 0;
    if (all(0, 1)) return 1;
    if (all(1, 1)) return 2;
    if (all(1, -1)) return 3;
    if (all(2, -1)) return 4;
    return 5;
  });
  return mesh;
}

void nsxh_mesh_free(nsxh_mesh *m) { delete m; }
int nsxh_mesh_dim(const nsxh_mesh *m) { return m->dim; }
int nsxh_mesh_n_vertices(const nsxh_mesh *m) { return m->n_vertices(); }
int nsxh_mesh_n_cells(const nsxh_mesh *m) { return m->n_cells(); }
int nsxh_mesh_n_bfaces(const nsxh_mesh *m) { return m->n_bfaces(); }
const double *nsxh_mesh_vertices(const nsxh_mesh *m) { return m->vertices.data(); }
const int32_t *nsxh_mesh_cells(const nsxh_mesh *m) { return m->cells.data(); }
const int32_t *nsxh_mesh_bfaces(const nsxh_mesh *m) { return m->bfaces.data(); }
const int32_t *nsxh_mesh_bface_ids(const nsxh_mesh *m) { return m->bface_ids.data(); }
const int32_t *nsxh_mesh_bface_cells(const nsxh_mesh *m) { return m->bface_cells.data(); }
const int32_t *nsxh_mesh_subdomain(const nsxh_mesh *m) { return m->subdomain.data(); }

int nsxh_mesh_partition_owned(nsxh_mesh *m, int n_parts, int n_sub) {
  if (n_parts < 1 || n_sub < 1) return -1;
  const int dim = m->dim, nv = dim + 1, nc = m->n_cells();
  if ((int64_t)n_parts * n_sub > nc) return -2;
  std::vector<double> cen((size_t)nc * dim, 0.0);
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < nv; ++k)
      for (int d = 0; d < dim; ++d) cen[(size_t)c * dim + d] += m->vertices[(size_t)m->cells[(size_t)c * nv + k] * dim + d] / nv;
  // P2 nodes of every cell: its vertices and its edges (numbered here through a sorted table of vertex pairs)
  const int ne_cell = dim == 2 ? 3 : 6, nent = nv + ne_cell;
  static const int E3[6][2] = {{0, 1}, {1, 2}, {2, 0}, {0, 3}, {1, 3}, {2, 3}}, E2[3][2] = {{0, 1}, {1, 2}, {2, 0}};
  std::vector<std::pair<int64_t, int32_t>> keys((size_t)nc * ne_cell);
  for (int c = 0; c < nc; ++c)
    for (int e = 0; e < ne_cell; ++e) {
      const int a = m->cells[(size_t)c * nv + (dim == 2 ? E2[e][0] : E3[e][0])], b = m->cells[(size_t)c * nv + (dim == 2 ? E2[e][1] : E3[e][1])];
      keys[(size_t)c * ne_cell + e] = {(int64_t)std::min(a, b) * m->n_vertices() + std::max(a, b), (int32_t)((size_t)c * ne_cell + e)};
    }
  std::sort(keys.begin(), keys.end());
  std::vector<int32_t> ents((size_t)nc * nent);
  int32_t n_edges = 0;
  for (size_t k = 0; k < keys.size(); ++k) {
    if (k > 0 && keys[k].first != keys[k - 1].first) ++n_edges;
    const int32_t slot = keys[k].second;
    ents[(size_t)(slot / ne_cell) * nent + nv + slot % ne_cell] = m->n_vertices() + n_edges;
  }
  n_edges += keys.empty() ? 0 : 1;
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nv; ++v) ents[(size_t)c * nent + v] = m->cells[(size_t)c * nv + v];
  const size_t n_ent_total = (size_t)m->n_vertices() + n_edges;
  std::vector<int32_t> idx(nc), sub(nc, 0), stamp(n_ent_total, 0);
  std::iota(idx.begin(), idx.end(), 0);
  std::vector<char> claimed(n_ent_total, 0);
  int32_t stamp_id = 0;
  // one recursion over all n_parts * n_sub subdomains: the first n_sub ids form part 0, and so on (n_sub a power of two keeps
  // the parts on the cuts of the upper levels; any n_sub keeps the ids of a part consecutive)
  rcb_owned(cen, dim, nent, ents, idx, 0, nc, n_parts * n_sub, 0, sub, claimed, stamp, stamp_id);
  m->subdomain = sub;
  m->n_subdomains = n_parts * n_sub;
  return 0;
}

int nsxh_mesh_partition(nsxh_mesh *m, int n_parts, int n_sub) {
  if (n_parts < 1 || n_sub < 1) return -1;
  const int dim = m->dim, nv = dim + 1, nc = m->n_cells();
  if ((int64_t)n_parts * n_sub > nc) return -2;
  std::vector<double> cen((size_t)nc * dim, 0.0);
  for (int c = 0; c < nc; ++c)
    for (int k = 0; k < nv; ++k)
      for (int d = 0; d < dim; ++d) cen[(size_t)c * dim + d] += m->vertices[(size_t)m->cells[(size_t)c * nv + k] * dim + d] / nv;
  std::vector<int32_t> idx(nc), part(nc, 0);
  std::iota(idx.begin(), idx.end(), 0);
  rcb(cen, dim, idx, 0, nc, n_parts, 0, part);
  m->subdomain.assign(nc, 0);
  // second level inside each part
  std::vector<std::vector<int32_t>> members(n_parts);
  for (int c = 0; c < nc; ++c) members[part[c]].push_back(c);
  std::vector<int32_t> sub(nc, 0);
  for (int p = 0; p < n_parts; ++p) {
    auto &mm = members[p];
    if ((int)mm.size() < n_sub) return -2;
    rcb(cen, dim, mm, 0, mm.size(), n_sub, 0, sub);
  }
  for (int c = 0; c < nc; ++c) m->subdomain[c] = part[c] * n_sub + sub[c];
  m->n_subdomains = n_parts * n_sub;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- MSH reader
namespace {
nsxh_mesh *read_msh_impl(const char *path) {
  std::ifstream in(path);
  if (!in) return nullptr;
  std::string line;
  double version = 2.2;
  std::vector<double> xyz;             // by file order
  std::unordered_map<long, int32_t> node_index;
  struct Elem { int type; int phys; std::vector<long> nodes; };
  std::vector<Elem> elems;
  std::map<std::pair<int, int>, int> entity_phys;  // (dim, tag) -> physical (v4)
  while (std::getline(in, line)) {
    if (line.rfind("$MeshFormat", 0) == 0) {
      std::getline(in, line);
      std::istringstream ss(line);
      ss >> version;
    } else if (line.rfind("$Entities", 0) == 0 && version >= 4) {
      size_t np, nc, ns, nvv;
      in >> np >> nc >> ns >> nvv;
      for (size_t i = 0; i < np; ++i) {
        int tag; double x, y, z; size_t nphys;
        in >> tag >> x >> y >> z >> nphys;
        for (size_t k = 0; k < nphys; ++k) { int p; in >> p; }
      }
      auto read_ents = [&](size_t n, int d) {
        for (size_t i = 0; i < n; ++i) {
          int tag; double b[6]; size_t nphys, nb;
          in >> tag;
          for (double &v : b) in >> v;
          in >> nphys;
          for (size_t k = 0; k < nphys; ++k) { int p; in >> p; if (k == 0) entity_phys[{d, tag}] = p; }
          in >> nb;
          for (size_t k = 0; k < nb; ++k) { int t; in >> t; }
        }
      };
      read_ents(nc, 1);
      read_ents(ns, 2);
      read_ents(nvv, 3);
    } else if (line.rfind("$Nodes", 0) == 0) {
      if (version < 4) {
        size_t n; in >> n;
        for (size_t i = 0; i < n; ++i) {
          long id; double x, y, z;
          in >> id >> x >> y >> z;
          node_index[id] = (int32_t)(xyz.size() / 3);
          xyz.insert(xyz.end(), {x, y, z});
        }
      } else {
        size_t nblocks, nn, mn, mx;
        in >> nblocks >> nn >> mn >> mx;
        for (size_t b = 0; b < nblocks; ++b) {
          int ed, et, par; size_t cnt;
          in >> ed >> et >> par >> cnt;
          std::vector<long> ids(cnt);
          for (auto &id : ids) in >> id;
          for (size_t i = 0; i < cnt; ++i) {
            double x, y, z; in >> x >> y >> z;
            node_index[ids[i]] = (int32_t)(xyz.size() / 3);
            xyz.insert(xyz.end(), {x, y, z});
          }
        }
      }
    } else if (line.rfind("$Elements", 0) == 0) {
      auto nn_of = [](int type) { return type == 1 ? 2 : type == 2 ? 3 : type == 4 ? 4 : type == 15 ? 1 : -1; };
      if (version < 4) {
        size_t n; in >> n;
        std::getline(in, line);
        for (size_t i = 0; i < n; ++i) {
          std::getline(in, line);
          std::istringstream ss(line);
          long id; int type, ntags;
          ss >> id >> type >> ntags;
          Elem e; e.type = type; e.phys = 0;
          for (int k = 0; k < ntags; ++k) { int tg; ss >> tg; if (k == 0) e.phys = tg; }
          int nn = nn_of(type);
          if (nn < 0) continue;
          e.nodes.resize(nn);
          for (auto &v : e.nodes) ss >> v;
          elems.push_back(e);
        }
      } else {
        size_t nblocks, ne, mn, mx;
        in >> nblocks >> ne >> mn >> mx;
        for (size_t b = 0; b < nblocks; ++b) {
          int ed, et, type; size_t cnt;
          in >> ed >> et >> type >> cnt;
          int nn = nn_of(type);
          auto it = entity_phys.find({ed, et});
          for (size_t i = 0; i < cnt; ++i) {
            long id; in >> id;
            Elem e; e.type = type; e.phys = it == entity_phys.end() ? 0 : it->second;
            if (nn < 0) { std::getline(in, line); continue; }
            e.nodes.resize(nn);
            for (auto &v : e.nodes) in >> v;
            elems.push_back(e);
          }
        }
      }
    }
  }
  bool has_tet = false;
  for (auto &e : elems) has_tet = has_tet || e.type == 4;
  auto *m = new nsxh_mesh;
  m->dim = has_tet ? 3 : 2;
  const int dim = m->dim, ctype = has_tet ? 4 : 2, ftype = has_tet ? 2 : 1;
  // keep only the vertices used by cells, renumbered in file order
  std::vector<int32_t> used(xyz.size() / 3, -1);
  for (auto &e : elems)
    if (e.type == ctype)
      for (long v : e.nodes) used[node_index[v]] = 0;
  int32_t nvk = 0;
  for (auto &u : used)
    if (u == 0) u = nvk++;
  m->vertices.resize((size_t)nvk * dim);
  for (size_t i = 0; i < used.size(); ++i)
    if (used[i] >= 0)
      for (int d = 0; d < dim; ++d) m->vertices[(size_t)used[i] * dim + d] = xyz[3 * i + d];
  std::map<std::array<int32_t, 3>, int> face_phys;
  for (auto &e : elems) {
    if (e.type == ctype) {
      for (long v : e.nodes) m->cells.push_back(used[node_index[v]]);
    } else if (e.type == ftype) {
      std::array<int32_t, 3> key = {-1, -1, -1};
      for (int k = 0; k < dim; ++k) key[k] = used[node_index[e.nodes[k]]];
      std::sort(key.begin(), key.begin() + dim);
      face_phys[key] = e.phys;
    }
  }
  if (m->cells.empty()) { delete m; return nullptr; }
  finish_mesh(*m);
  extract_boundary(*m, [&](const double *) { return 0; });
  for (int f = 0; f < m->n_bfaces(); ++f) {
    std::array<int32_t, 3> key = {-1, -1, -1};
    for (int k = 0; k < dim; ++k) key[k] = m->bfaces[(size_t)f * dim + k];
    std::sort(key.begin(), key.begin() + dim);
    auto it = face_phys.find(key);
    m->bface_ids[f] = it == face_phys.end() ? 0 : it->second;
  }
  return m;
}
}  // namespace

extern "C" nsxh_mesh *nsxh_mesh_read_msh(const char *path) { return read_msh_impl(path); }

// ---------------------------------------------------------------- DoF handler
struct nsxh_dofs {
  int dim = 0, n_cells = 0, dpc = 0, n2 = 0, n1 = 0, n_sub = 1, n_colours = 0, n_colours_p = 0;
  const nsxh_mesh *mesh = nullptr;
  std::vector<int32_t> cell_dofs;
  std::vector<int32_t> cell_nodes2, cell_nodes1;  // scalar connectivity
  std::vector<double> cell_coords, support;
  std::vector<int32_t> node_owner, pnode_owner, owned_u_ptr, owned_p_ptr;
  std::vector<int32_t> vertex_node, vertex_pnode;
  std::unordered_map<uint64_t, int32_t> edge_node;
  std::map<int, std::vector<int32_t>> bdofs;
  nsx::Csr ref[4];
  bool have_ref[4] = {false, false, false, false};
};

// NSXH_ORDER_COLOUR: renumber the P2 nodes inside every subdomain by greedy colour (nodes of one colour share no cell
// inside the subdomain).  The ILU(0) dependency graph of a rank block is then only as deep as the number of colours
// (13 on the 3D channel meshes against up to 93 levels with first-touch numbering), which is what bounds the
// triangular solves on the GPU.  Ownership, the rank ranges and the pressure numbering are untouched.
static void renumber_by_colour(nsxh_dofs *d, int np2) {
  const int n2 = (int)d->node_owner.size();
  std::vector<int32_t> perm;
  d->n_colours = nsx::colour_perm(n2, d->n_cells, np2, d->cell_nodes2.data(), d->node_owner.data(), d->owned_u_ptr, perm);
  for (auto &v : d->vertex_node)
    if (v >= 0) v = perm[v];
  for (auto &kv : d->edge_node) kv.second = perm[kv.second];
  for (auto &v : d->cell_nodes2) v = perm[v];
}

// NSXH_ORDER_COLOUR_ALL: the P1 (pressure) nodes of every subdomain as well, coloured on the graph of the Schur complement
// B D^-1 B^T (two pressure nodes are adjacent when some P2 node shares a cell with each of them), because that is the matrix
// whose per-rank ILU(0) is applied in every CG iteration.  With few large ranks (mpirun -n 1, one rank per GPU) the first-touch
// pressure numbering gives that factorisation thousands of dependency levels; by colour it has as many as there are colours.
static int renumber_pressure_by_colour(nsxh_dofs *d, int np2, int nv) {
  const int n1 = (int)d->pnode_owner.size(), n2 = (int)d->node_owner.size();
  std::vector<int32_t> perm;
  const int max_col = nsx::colour_perm_schur(n1, n2, d->n_cells, np2, nv, d->cell_nodes2.data(), d->cell_nodes1.data(), d->pnode_owner.data(),
                                             d->owned_p_ptr, perm);
  for (auto &v : d->vertex_pnode)
    if (v >= 0) v = perm[v];
  for (auto &v : d->cell_nodes1) v = perm[v];
  return max_col;
}

extern "C" {

nsxh_dofs *nsxh_distribute_dofs(const nsxh_mesh *m) { return nsxh_distribute_dofs_ordered(m, NSXH_ORDER_FIRST_TOUCH); }

nsxh_dofs *nsxh_distribute_dofs_ordered(const nsxh_mesh *m, int ordering) {
  if (!m || (ordering != NSXH_ORDER_FIRST_TOUCH && ordering != NSXH_ORDER_COLOUR && ordering != NSXH_ORDER_COLOUR_ALL)) return nullptr;
  auto *d = new nsxh_dofs;
  const int dim = m->dim, nv = dim + 1, nl = dim == 2 ? 3 : 6, nc = m->n_cells();
  d->dim = dim;
  d->mesh = m;
  d->n_cells = nc;
  d->n_sub = m->n_subdomains;
  const int np2 = nv + nl;
  d->dpc = nv * (dim + 1) + nl * dim;
  d->vertex_node.assign(m->n_vertices(), -1);
  d->vertex_pnode.assign(m->n_vertices(), -1);
  std::vector<int32_t> order(nc);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return m->subdomain[a] < m->subdomain[b]; });
  d->owned_u_ptr.assign(d->n_sub + 1, 0);
  d->owned_p_ptr.assign(d->n_sub + 1, 0);
  d->cell_nodes2.resize((size_t)nc * np2);
  d->cell_nodes1.resize((size_t)nc * nv);
  int32_t next2 = 0, next1 = 0;
  d->edge_node.reserve((size_t)nc * 2);
  for (int32_t oc = 0; oc < nc; ++oc) {
    const int32_t c = order[oc];
    const int s = m->subdomain[c];
    const int32_t *cc = m->cells.data() + (size_t)c * nv;
    // vertices first, then lines (deal.II distribute_dofs order on a cell)
    for (int v = 0; v < nv; ++v) {
      if (d->vertex_node[cc[v]] < 0) {
        d->vertex_node[cc[v]] = next2++;
        d->vertex_pnode[cc[v]] = next1++;
        d->node_owner.push_back(s);
        d->pnode_owner.push_back(s);
      }
      d->cell_nodes2[(size_t)c * np2 + v] = d->vertex_node[cc[v]];
      d->cell_nodes1[(size_t)c * nv + v] = d->vertex_pnode[cc[v]];
    }
    for (int l = 0; l < nl; ++l) {
      const int a = cc[dim == 2 ? TRI_LINES[l][0] : TET_LINES[l][0]], b = cc[dim == 2 ? TRI_LINES[l][1] : TET_LINES[l][1]];
      auto ins = d->edge_node.emplace(edge_key(a, b), next2);
      if (ins.second) {
        ++next2;
        d->node_owner.push_back(s);
      }
      d->cell_nodes2[(size_t)c * np2 + nv + l] = ins.first->second;
    }
    d->owned_u_ptr[s + 1] = next2;
    d->owned_p_ptr[s + 1] = next1;
  }
  for (int s = 0; s < d->n_sub; ++s) {  // empty subdomains inherit
    d->owned_u_ptr[s + 1] = std::max(d->owned_u_ptr[s + 1], d->owned_u_ptr[s]);
    d->owned_p_ptr[s + 1] = std::max(d->owned_p_ptr[s + 1], d->owned_p_ptr[s]);
  }
  d->n2 = next2;
  d->n1 = next1;
  if (ordering == NSXH_ORDER_COLOUR || ordering == NSXH_ORDER_COLOUR_ALL) renumber_by_colour(d, np2);
  if (ordering == NSXH_ORDER_COLOUR_ALL) d->n_colours_p = renumber_pressure_by_colour(d, np2, nv);
  const int32_t n_u = dim * next2;
  d->cell_dofs.resize((size_t)nc * d->dpc);
  d->cell_coords.resize((size_t)nc * nv * dim);
  d->support.assign((size_t)(n_u + next1) * dim, 0.0);
  for (int32_t c = 0; c < nc; ++c) {
    const int32_t *cc = m->cells.data() + (size_t)c * nv;
    int32_t *cd = d->cell_dofs.data() + (size_t)c * d->dpc;
    for (int v = 0; v < nv; ++v) {
      const int32_t node = d->cell_nodes2[(size_t)c * np2 + v];
      for (int k = 0; k < dim; ++k) {
        cd[(dim + 1) * v + k] = dim * node + k;
        d->cell_coords[((size_t)c * nv + v) * dim + k] = m->vertices[(size_t)cc[v] * dim + k];
        for (int q = 0; q < dim; ++q) d->support[(size_t)(dim * node + k) * dim + q] = m->vertices[(size_t)cc[v] * dim + q];
      }
      const int32_t pn = d->cell_nodes1[(size_t)c * nv + v];
      cd[(dim + 1) * v + dim] = n_u + pn;
      for (int q = 0; q < dim; ++q) d->support[(size_t)(n_u + pn) * dim + q] = m->vertices[(size_t)cc[v] * dim + q];
    }
    for (int l = 0; l < nl; ++l) {
      const int32_t node = d->cell_nodes2[(size_t)c * np2 + nv + l];
      const int a = cc[dim == 2 ? TRI_LINES[l][0] : TET_LINES[l][0]], b = cc[dim == 2 ? TRI_LINES[l][1] : TET_LINES[l][1]];
      for (int k = 0; k < dim; ++k) {
        cd[(dim + 1) * nv + dim * l + k] = dim * node + k;
        for (int q = 0; q < dim; ++q)
          d->support[(size_t)(dim * node + k) * dim + q] = 0.5 * (m->vertices[(size_t)a * dim + q] + m->vertices[(size_t)b * dim + q]);
      }
    }
  }
  return d;
}

void nsxh_dofs_free(nsxh_dofs *d) { delete d; }
int nsxh_dofs_per_cell(const nsxh_dofs *d) { return d->dpc; }
int nsxh_n_nodes_p2(const nsxh_dofs *d) { return d->n2; }
int nsxh_n_nodes_p1(const nsxh_dofs *d) { return d->n1; }
int nsxh_n_u(const nsxh_dofs *d) { return d->dim * d->n2; }
int nsxh_n_p(const nsxh_dofs *d) { return d->n1; }
const int32_t *nsxh_cell_dofs(const nsxh_dofs *d) { return d->cell_dofs.data(); }
const double *nsxh_cell_coords(const nsxh_dofs *d) { return d->cell_coords.data(); }
const double *nsxh_support_points(const nsxh_dofs *d) { return d->support.data(); }
const int32_t *nsxh_node_owner(const nsxh_dofs *d) { return d->node_owner.data(); }
const int32_t *nsxh_pnode_owner(const nsxh_dofs *d) { return d->pnode_owner.data(); }
const int32_t *nsxh_owned_u_ptr(const nsxh_dofs *d) { return d->owned_u_ptr.data(); }
const int32_t *nsxh_owned_p_ptr(const nsxh_dofs *d) { return d->owned_p_ptr.data(); }
int nsxh_n_subdomains(const nsxh_dofs *d) { return d->n_sub; }
int nsxh_n_colours(const nsxh_dofs *d) { return d->n_colours; }
int nsxh_n_colours_p(const nsxh_dofs *d) { return d->n_colours_p; }

int nsxh_boundary_dofs(nsxh_dofs *d, int boundary_id, const int32_t **dofs) {
  auto it = d->bdofs.find(boundary_id);
  if (it == d->bdofs.end()) {
    const nsxh_mesh *m = d->mesh;
    const int dim = d->dim;
    std::vector<int32_t> nodes;
    for (int f = 0; f < m->n_bfaces(); ++f) {
      if (m->bface_ids[f] != boundary_id) continue;
      const int32_t *fv = m->bfaces.data() + (size_t)f * dim;
      for (int k = 0; k < dim; ++k) nodes.push_back(d->vertex_node[fv[k]]);
      if (dim == 2) {
        nodes.push_back(d->edge_node.at(edge_key(fv[0], fv[1])));
      } else {
        nodes.push_back(d->edge_node.at(edge_key(fv[0], fv[1])));
        nodes.push_back(d->edge_node.at(edge_key(fv[1], fv[2])));
        nodes.push_back(d->edge_node.at(edge_key(fv[2], fv[0])));
      }
    }
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    std::vector<int32_t> out;
    out.reserve(nodes.size() * dim);
    for (int32_t n : nodes)
      for (int k = 0; k < dim; ++k) out.push_back(dim * n + k);
    it = d->bdofs.emplace(boundary_id, std::move(out)).first;
  }
  *dofs = it->second.data();
  return (int)it->second.size();
}

int nsxh_reference_sparsity(nsxh_dofs *d, int block, const int32_t **rowptr, const int32_t **colind) {
  if (block < 0 || block > 3) return -1;
  if (!d->have_ref[block]) {
    const int dim = d->dim, nv = dim + 1, np2 = nv + (dim == 2 ? 3 : 6);
    const int32_t *c2 = d->cell_nodes2.data(), *c1 = d->cell_nodes1.data();
    nsx::Csr s, out;
    int rmul = 1, cmul = 1;
    if (block == 0) { s = nsx::build_graph(d->n_cells, np2, c2, d->n2, np2, c2, d->n2); rmul = cmul = dim; }
    if (block == 1) { s = nsx::build_graph(d->n_cells, np2, c2, d->n2, nv, c1, d->n1); rmul = dim; }
    if (block == 2) { s = nsx::build_graph(d->n_cells, nv, c1, d->n1, np2, c2, d->n2); cmul = dim; }
    if (block == 3) { s = nsx::build_graph(d->n_cells, nv, c1, d->n1, nv, c1, d->n1); }
    out.n_rows = s.n_rows * rmul;
    out.n_cols = s.n_cols * cmul;
    out.rowptr.assign((size_t)out.n_rows + 1, 0);
    out.colind.resize((size_t)s.nnz() * rmul * cmul);
    int64_t pos = 0;
    for (int32_t i = 0; i < s.n_rows; ++i)
      for (int r = 0; r < rmul; ++r) {
        for (int32_t k = s.rowptr[i]; k < s.rowptr[i + 1]; ++k)
          for (int c = 0; c < cmul; ++c) out.colind[pos++] = s.colind[k] * cmul + c;
        out.rowptr[(size_t)i * rmul + r + 1] = (int32_t)pos;
      }
    d->ref[block] = std::move(out);
    d->have_ref[block] = true;
  }
  *rowptr = d->ref[block].rowptr.data();
  *colind = d->ref[block].colind.data();
  return d->ref[block].n_rows;
}

}  // extern "C"

// ---------------------------------------------------------------- FE tables
struct nsxh_tables {
  int dim = 0, n_q = 0, n_qf = 0, np2 = 0, np1 = 0;
  std::vector<double> pts, w, N2, dN2, N1, dN1;
};

namespace {

void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w) {
  // nodes/weights on [0,1]
  x.resize(n);
  w.resize(n);
  for (int i = 0; i < n; ++i) {
    double z = std::cos(M_PI * (i + 0.75) / (n + 0.5)), pp = 0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1, p2 = 0;
      for (int j = 0; j < n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j + 1) * z * p2 - j * p3) / (j + 1);
      }
      pp = n * (z * p1 - p2) / (z * z - 1);
      const double dz = p1 / pp;
      z -= dz;
      if (std::fabs(dz) < 1e-16) break;
    }
    x[n - 1 - i] = 0.5 * (z + 1);
    w[n - 1 - i] = 1.0 / ((1 - z * z) * pp * pp);
  }
}

void cell_rule(int dim, std::vector<double> &p, std::vector<double> &w) {
  p.clear();
  w.clear();
  if (dim == 1) {  // 3-point Gauss on [0,1]
    std::vector<double> x, ww;
    gauss_legendre(3, x, ww);
    p = x;
    w = ww;
  } else if (dim == 2) {  // Radon 7-point, degree 5 (tools/gen_quadrature.py)
    const double a1 = 0.1012865073234563388, w1 = 0.062969590272413576298, a2 = 0.47014206410511508977,
                 w2 = 0.066197076394253090369, wc = 0.1125;
    p = {1.0 / 3, 1.0 / 3};
    w = {wc};
    for (int o = 0; o < 2; ++o) {
      const double a = o ? a2 : a1, ww = o ? w2 : w1, r = 1 - 2 * a;
      const double q[3][2] = {{a, a}, {r, a}, {a, r}};
      for (auto &s : q) {
        p.push_back(s[0]);
        p.push_back(s[1]);
        w.push_back(ww);
      }
    }
  } else {  // 14-point degree 5 with positive weights (tools/gen_quadrature.py)
    const double a1 = 0.3108859192633006098, w1 = 0.0187813209530026418, a2 = 0.092735250310891226402,
                 w2 = 0.012248840519393658257, b = 0.045503704125649649492, w3 = 0.007091003462846911073;
    for (int o = 0; o < 2; ++o) {
      const double a = o ? a2 : a1, ww = o ? w2 : w1, r = 1 - 3 * a;
      const double q[4][3] = {{a, a, a}, {r, a, a}, {a, r, a}, {a, a, r}};
      for (auto &s : q) {
        p.insert(p.end(), {s[0], s[1], s[2]});
        w.push_back(ww);
      }
    }
    const double c = 0.5 - b;
    const double q[6][3] = {{b, b, c}, {b, c, b}, {c, b, b}, {b, c, c}, {c, b, c}, {c, c, b}};
    for (auto &s : q) {
      p.insert(p.end(), {s[0], s[1], s[2]});
      w.push_back(w3);
    }
  }
}

void conical_rule(int dim, int n, std::vector<double> &p, std::vector<double> &w) {
  std::vector<double> x, ww;
  gauss_legendre(n, x, ww);
  p.clear();
  w.clear();
  if (dim == 2) {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        p.push_back(x[i]);
        p.push_back(x[j] * (1 - x[i]));
        w.push_back(ww[i] * ww[j] * (1 - x[i]));
      }
  } else {
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
          const double u = x[i], v = x[j] * (1 - u), t = x[k] * (1 - u - v);
          p.insert(p.end(), {u, v, t});
          w.push_back(ww[i] * ww[j] * ww[k] * (1 - u) * (1 - u - v));
        }
  }
}

void eval_shapes(nsxh_tables &t) {
  const int dim = t.dim, nv = dim + 1, nl = dim == 2 ? 3 : 6;
  t.np2 = nv + nl;
  t.np1 = nv;
  t.N2.assign((size_t)t.n_q * t.np2, 0);
  t.dN2.assign((size_t)t.n_q * t.np2 * dim, 0);
  t.N1.assign((size_t)t.n_q * t.np1, 0);
  t.dN1.assign((size_t)t.n_q * t.np1 * dim, 0);
  for (int q = 0; q < t.n_q; ++q) {
    double lam[4], dl[4][3];
    lam[0] = 1;
    for (int k = 0; k < dim; ++k) {
      lam[k + 1] = t.pts[(size_t)q * dim + k];
      lam[0] -= lam[k + 1];
    }
    for (int v = 0; v < nv; ++v)
      for (int k = 0; k < dim; ++k) dl[v][k] = v == 0 ? -1.0 : (v - 1 == k ? 1.0 : 0.0);
    for (int v = 0; v < nv; ++v) {
      t.N1[(size_t)q * t.np1 + v] = lam[v];
      t.N2[(size_t)q * t.np2 + v] = lam[v] * (2 * lam[v] - 1);
      for (int k = 0; k < dim; ++k) {
        t.dN1[((size_t)q * t.np1 + v) * dim + k] = dl[v][k];
        t.dN2[((size_t)q * t.np2 + v) * dim + k] = (4 * lam[v] - 1) * dl[v][k];
      }
    }
    for (int l = 0; l < nl; ++l) {
      const int a = dim == 2 ? TRI_LINES[l][0] : TET_LINES[l][0], b = dim == 2 ? TRI_LINES[l][1] : TET_LINES[l][1];
      t.N2[(size_t)q * t.np2 + nv + l] = 4 * lam[a] * lam[b];
      for (int k = 0; k < dim; ++k) t.dN2[((size_t)q * t.np2 + nv + l) * dim + k] = 4 * (lam[a] * dl[b][k] + lam[b] * dl[a][k]);
    }
  }
}

}  // namespace

extern "C" {

nsxh_tables *nsxh_tables_create(int dim, int rule, int order) {
  if (dim != 2 && dim != 3) return nullptr;
  auto *t = new nsxh_tables;
  t->dim = dim;
  if (rule == 0) {
    cell_rule(dim, t->pts, t->w);
    t->n_q = (int)t->w.size();
  } else if (rule == 2) {
    conical_rule(dim, order < 1 ? 5 : order, t->pts, t->w);
    t->n_q = (int)t->w.size();
  } else if (rule == 1) {
    std::vector<double> fp, fw;
    cell_rule(dim - 1, fp, fw);
    double sum = 0;
    for (double v : fw) sum += v;
    const int nqf = (int)fw.size(), nf = dim + 1;
    t->n_qf = nqf;
    t->n_q = nqf * nf;
    const double RV[4][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int f = 0; f < nf; ++f)
      for (int q = 0; q < nqf; ++q) {
        double bary[3];  // barycentric on the face
        if (dim == 2) {
          bary[1] = fp[q];
          bary[0] = 1 - bary[1];
        } else {
          bary[1] = fp[2 * q];
          bary[2] = fp[2 * q + 1];
          bary[0] = 1 - bary[1] - bary[2];
        }
        for (int k = 0; k < dim; ++k) {
          double x = 0;
          for (int v = 0; v < dim; ++v) x += bary[v] * RV[dim == 2 ? TRI_FACES[f][v] : TET_FACES[f][v]][k];
          t->pts.push_back(x);
        }
        t->w.push_back(fw[q] / sum);  // normalised: JxW = w * |physical face|
      }
  } else {
    delete t;
    return nullptr;
  }
  eval_shapes(*t);
  return t;
}

void nsxh_tables_free(nsxh_tables *t) { delete t; }
int nsxh_tables_n_q(const nsxh_tables *t) { return t->n_q; }
int nsxh_tables_n_qf(const nsxh_tables *t) { return t->n_qf; }
int nsxh_tables_n_p2(const nsxh_tables *t) { return t->np2; }
int nsxh_tables_n_p1(const nsxh_tables *t) { return t->np1; }
const double *nsxh_tables_points(const nsxh_tables *t) { return t->pts.data(); }
const double *nsxh_tables_weights(const nsxh_tables *t) { return t->w.data(); }
const double *nsxh_tables_N2(const nsxh_tables *t) { return t->N2.data(); }
const double *nsxh_tables_dN2(const nsxh_tables *t) { return t->dN2.data(); }
const double *nsxh_tables_N1(const nsxh_tables *t) { return t->N1.data(); }
const double *nsxh_tables_dN1(const nsxh_tables *t) { return t->dN1.data(); }

}  // extern "C"

// ---------------------------------------------------------------- per-rank view for multi-GPU runs
// One GPU = n_sub consecutive subdomains ("virtual ranks").  A rank keeps every cell that touches one of its owned
// P2 nodes (layer 1: enough to assemble all owned rows without any exchange, replacing compress(VectorOperation::add),
// reference NavierStokes3D.cpp:314-319,506-511) plus the cells touching a node of those (layer 2: makes the rows of
// block(1,0) complete for every ghost pressure node the Schur product B D^-1 B^T of an owned row can reach).
// The halo plan (what Epetra_Import does inside every vmult) is computed from the replicated serial mesh, exactly as
// the reference replicates it on every rank before partitioning (NavierStokes3D.cpp:8-19): no setup communication.
struct nsxh_rank_view {
  int rank = 0, world = 1, dim = 0, dpc = 0, n_cells = 0, n_cells_layer1 = 0;
  std::vector<int32_t> cell_ids, cell_dofs, gpu_u_ptr, gpu_p_ptr, rank_u_ptr, rank_p_ptr;
  std::vector<double> cell_coords;
  std::vector<int32_t> nbr, send_u_ptr, send_u_nodes, send_p_ptr, send_p_nodes;
};

namespace {

struct LocalSets {
  std::vector<int32_t> cells1, cells2;      // layer-1 / extra layer-2 cells (ascending)
  std::vector<int32_t> nodes2, nodes1;      // all local P2 / P1 nodes (sorted global ids)
};

LocalSets local_sets(const nsxh_dofs *d, int32_t u0, int32_t u1) {
  const int dim = d->dim, nv = dim + 1, np2 = nv + (dim == 2 ? 3 : 6);
  const int nc = d->n_cells;
  LocalSets s;
  std::vector<char> node_in(d->n2, 0), cell_in(nc, 0);
  for (int c = 0; c < nc; ++c) {
    const int32_t *cn = d->cell_nodes2.data() + (size_t)c * np2;
    bool own = false;
    for (int a = 0; a < np2 && !own; ++a) own = cn[a] >= u0 && cn[a] < u1;
    if (own) {
      cell_in[c] = 1;
      s.cells1.push_back(c);
    }
  }
  for (int32_t c : s.cells1)
    for (int a = 0; a < np2; ++a) node_in[d->cell_nodes2[(size_t)c * np2 + a]] = 1;
  for (int c = 0; c < nc; ++c) {
    if (cell_in[c]) continue;
    const int32_t *cn = d->cell_nodes2.data() + (size_t)c * np2;
    bool touch = false;
    for (int a = 0; a < np2 && !touch; ++a) touch = node_in[cn[a]];
    if (touch) {
      cell_in[c] = 2;
      s.cells2.push_back(c);
    }
  }
  std::vector<char> n2(d->n2, 0), n1(d->n1, 0);
  for (int c = 0; c < nc; ++c) {
    if (!cell_in[c]) continue;
    for (int a = 0; a < np2; ++a) n2[d->cell_nodes2[(size_t)c * np2 + a]] = 1;
    for (int v = 0; v < nv; ++v) n1[d->cell_nodes1[(size_t)c * nv + v]] = 1;
  }
  for (int32_t i = 0; i < d->n2; ++i)
    if (n2[i]) s.nodes2.push_back(i);
  for (int32_t i = 0; i < d->n1; ++i)
    if (n1[i]) s.nodes1.push_back(i);
  return s;
}

}  // namespace

extern "C" {

nsxh_rank_view *nsxh_rank_view_create(const nsxh_dofs *d, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world || d->n_sub % world) return nullptr;
  const int n_sub = d->n_sub / world, dim = d->dim, nv = dim + 1;
  auto *v = new nsxh_rank_view;
  v->rank = rank;
  v->world = world;
  v->dim = dim;
  v->dpc = d->dpc;
  for (int r = 0; r <= world; ++r) {
    v->gpu_u_ptr.push_back(d->owned_u_ptr[(size_t)r * n_sub]);
    v->gpu_p_ptr.push_back(d->owned_p_ptr[(size_t)r * n_sub]);
  }
  for (int s = 0; s <= n_sub; ++s) {
    v->rank_u_ptr.push_back(d->owned_u_ptr[(size_t)rank * n_sub + s]);
    v->rank_p_ptr.push_back(d->owned_p_ptr[(size_t)rank * n_sub + s]);
  }
  LocalSets mine = local_sets(d, v->gpu_u_ptr[rank], v->gpu_u_ptr[rank + 1]);
  v->n_cells_layer1 = (int)mine.cells1.size();
  v->cell_ids = mine.cells1;
  v->cell_ids.insert(v->cell_ids.end(), mine.cells2.begin(), mine.cells2.end());
  v->n_cells = (int)v->cell_ids.size();
  v->cell_dofs.reserve((size_t)v->n_cells * d->dpc);
  v->cell_coords.reserve((size_t)v->n_cells * nv * dim);
  for (int32_t c : v->cell_ids) {
    v->cell_dofs.insert(v->cell_dofs.end(), d->cell_dofs.begin() + (size_t)c * d->dpc, d->cell_dofs.begin() + (size_t)(c + 1) * d->dpc);
    v->cell_coords.insert(v->cell_coords.end(), d->cell_coords.begin() + (size_t)c * nv * dim,
                          d->cell_coords.begin() + (size_t)(c + 1) * nv * dim);
  }
  // send lists: what every other rank's local set contains of MY nodes (sorted by global id = the order in which the
  // receiver stores its ghosts)
  v->send_u_ptr.push_back(0);
  v->send_p_ptr.push_back(0);
  for (int s = 0; s < world; ++s) {
    if (s == rank) continue;
    LocalSets other = local_sets(d, v->gpu_u_ptr[s], v->gpu_u_ptr[s + 1]);
    std::vector<int32_t> su, sp;
    for (int32_t n : other.nodes2)
      if (n >= v->gpu_u_ptr[rank] && n < v->gpu_u_ptr[rank + 1]) su.push_back(n);
    for (int32_t n : other.nodes1)
      if (n >= v->gpu_p_ptr[rank] && n < v->gpu_p_ptr[rank + 1]) sp.push_back(n);
    // a neighbour is also a rank I receive from, even if I send it nothing
    bool recv_from = false;
    for (int32_t n : mine.nodes2) recv_from = recv_from || (n >= v->gpu_u_ptr[s] && n < v->gpu_u_ptr[s + 1]);
    for (int32_t n : mine.nodes1) recv_from = recv_from || (n >= v->gpu_p_ptr[s] && n < v->gpu_p_ptr[s + 1]);
    if (su.empty() && sp.empty() && !recv_from) continue;
    v->nbr.push_back(s);
    v->send_u_nodes.insert(v->send_u_nodes.end(), su.begin(), su.end());
    v->send_p_nodes.insert(v->send_p_nodes.end(), sp.begin(), sp.end());
    v->send_u_ptr.push_back((int32_t)v->send_u_nodes.size());
    v->send_p_ptr.push_back((int32_t)v->send_p_nodes.size());
  }
  return v;
}
void nsxh_rank_view_free(nsxh_rank_view *v) { delete v; }
int nsxh_rank_view_n_cells(const nsxh_rank_view *v) { return v->n_cells; }
int nsxh_rank_view_n_cells_layer1(const nsxh_rank_view *v) { return v->n_cells_layer1; }
const int32_t *nsxh_rank_view_cell_ids(const nsxh_rank_view *v) { return v->cell_ids.data(); }
const int32_t *nsxh_rank_view_cell_dofs(const nsxh_rank_view *v) { return v->cell_dofs.data(); }
const double *nsxh_rank_view_cell_coords(const nsxh_rank_view *v) { return v->cell_coords.data(); }
const int32_t *nsxh_rank_view_gpu_u_ptr(const nsxh_rank_view *v) { return v->gpu_u_ptr.data(); }
const int32_t *nsxh_rank_view_gpu_p_ptr(const nsxh_rank_view *v) { return v->gpu_p_ptr.data(); }
int nsxh_rank_view_n_virtual_ranks(const nsxh_rank_view *v) { return (int)v->rank_u_ptr.size() - 1; }
const int32_t *nsxh_rank_view_rank_u_ptr(const nsxh_rank_view *v) { return v->rank_u_ptr.data(); }
const int32_t *nsxh_rank_view_rank_p_ptr(const nsxh_rank_view *v) { return v->rank_p_ptr.data(); }
int nsxh_rank_view_n_neighbors(const nsxh_rank_view *v) { return (int)v->nbr.size(); }
const int32_t *nsxh_rank_view_neighbors(const nsxh_rank_view *v) { return v->nbr.data(); }
const int32_t *nsxh_rank_view_send_u_ptr(const nsxh_rank_view *v) { return v->send_u_ptr.data(); }
const int32_t *nsxh_rank_view_send_u_nodes(const nsxh_rank_view *v) { return v->send_u_nodes.data(); }
const int32_t *nsxh_rank_view_send_p_ptr(const nsxh_rank_view *v) { return v->send_p_ptr.data(); }
const int32_t *nsxh_rank_view_send_p_nodes(const nsxh_rank_view *v) { return v->send_p_nodes.data(); }

}  // extern "C"

// ------------------------------------------------------------------ post-processing (N4: output side of the reference)
extern "C" {

int nsxh_write_vtu(const nsxh_dofs *d, const double *solution, const char *directory, const char *basename, unsigned counter) {
  if (!d || !solution || !directory || !basename) return -1;
  const int dim = d->dim, nv = dim + 1, nc = d->n_cells;
  std::string dir(directory);
  if (!dir.empty() && dir.back() != '/') dir += '/';
  for (size_t pos = 1; pos < dir.size(); ++pos)  // mkdir -p
    if (dir[pos] == '/') ::mkdir(dir.substr(0, pos).c_str(), 0777);
  const std::string stem = std::string(basename) + "_" + std::to_string(counter);
  const std::string piece = stem + ".0.vtu";
  std::ofstream f(dir + piece);
  if (!f) return -1;
  f << std::setprecision(17);
  const int32_t *sub = d->mesh->subdomain.empty() ? nullptr : d->mesh->subdomain.data();
  f << "<?xml version=\"1.0\"?>\n<VTKFile type=\"UnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n<UnstructuredGrid>\n"
    << "<Piece NumberOfPoints=\"" << (int64_t)nc * nv << "\" NumberOfCells=\"" << nc << "\">\n";
  f << "<Points>\n<DataArray type=\"Float64\" NumberOfComponents=\"3\" format=\"ascii\">\n";
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nv; ++v) {
      const double *x = d->cell_coords.data() + ((size_t)c * nv + v) * dim;
      f << x[0] << ' ' << x[1] << ' ' << (dim == 3 ? x[2] : 0.0) << '\n';
    }
  f << "</DataArray>\n</Points>\n<Cells>\n<DataArray type=\"Int32\" Name=\"connectivity\" format=\"ascii\">\n";
  for (int64_t k = 0; k < (int64_t)nc * nv; ++k) f << k << ((k + 1) % nv ? ' ' : '\n');
  f << "</DataArray>\n<DataArray type=\"Int32\" Name=\"offsets\" format=\"ascii\">\n";
  for (int c = 1; c <= nc; ++c) f << (int64_t)c * nv << (c % 16 ? ' ' : '\n');
  f << "\n</DataArray>\n<DataArray type=\"UInt8\" Name=\"types\" format=\"ascii\">\n";
  for (int c = 1; c <= nc; ++c) f << (dim == 3 ? 10 : 5) << (c % 32 ? ' ' : '\n');  // VTK_TETRA / VTK_TRIANGLE
  f << "\n</DataArray>\n</Cells>\n<PointData Scalars=\"scalars\">\n";
  f << "<DataArray type=\"Float64\" Name=\"velocity\" NumberOfComponents=\"3\" format=\"ascii\">\n";
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nv; ++v) {
      const int32_t *cd = d->cell_dofs.data() + (size_t)c * d->dpc + (size_t)(dim + 1) * v;
      f << solution[cd[0]] << ' ' << solution[cd[1]] << ' ' << (dim == 3 ? solution[cd[2]] : 0.0) << '\n';
    }
  f << "</DataArray>\n<DataArray type=\"Float64\" Name=\"pressure\" format=\"ascii\">\n";
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nv; ++v) f << solution[d->cell_dofs[(size_t)c * d->dpc + (size_t)(dim + 1) * v + dim]] << (v + 1 == nv ? '\n' : ' ');
  f << "</DataArray>\n<DataArray type=\"Float64\" Name=\"partitioning\" format=\"ascii\">\n";
  for (int c = 0; c < nc; ++c)
    for (int v = 0; v < nv; ++v) f << (sub ? sub[c] : 0) << (v + 1 == nv ? '\n' : ' ');
  f << "</DataArray>\n</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n";
  f.close();
  std::ofstream pv(dir + stem + ".pvtu");
  if (!pv) return -1;
  pv << "<?xml version=\"1.0\"?>\n<VTKFile type=\"PUnstructuredGrid\" version=\"0.1\" byte_order=\"LittleEndian\">\n"
     << "<PUnstructuredGrid GhostLevel=\"0\">\n<PPointData Scalars=\"scalars\">\n"
     << "<PDataArray type=\"Float64\" Name=\"velocity\" NumberOfComponents=\"3\" format=\"ascii\"/>\n"
     << "<PDataArray type=\"Float64\" Name=\"pressure\" format=\"ascii\"/>\n"
     << "<PDataArray type=\"Float64\" Name=\"partitioning\" format=\"ascii\"/>\n</PPointData>\n"
     << "<PPoints>\n<PDataArray type=\"Float64\" NumberOfComponents=\"3\"/>\n</PPoints>\n"
     << "<Piece Source=\"" << piece << "\"/>\n</PUnstructuredGrid>\n</VTKFile>\n";
  return pv ? 0 : -1;
}

int nsxh_pressure_difference(const nsxh_dofs *d, const double *solution, const double *pa, const double *pb, double *diff) {
  if (!d || !solution || !pa || !pb || !diff) return -1;
  const int dim = d->dim, nv = dim + 1;
  double val[2] = {0.0, 0.0};
  bool found[2] = {false, false};
  const double *pts[2] = {pa, pb};
  for (int c = 0; c < d->n_cells && !(found[0] && found[1]); ++c) {
    const double *X = d->cell_coords.data() + (size_t)c * nv * dim;
    // barycentric coordinates: x = X0 + J lambda
    double J[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < dim; ++i)
      for (int j = 0; j < dim; ++j) J[i][j] = X[(size_t)(j + 1) * dim + i] - X[i];
    const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                       J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    if (det == 0.0) continue;
    for (int k = 0; k < 2; ++k) {
      if (found[k]) continue;
      double r[3] = {0, 0, 0}, lam[3] = {0, 0, 0};
      for (int i = 0; i < dim; ++i) r[i] = pts[k][i] - X[i];
      // Cramer
      for (int j = 0; j < 3; ++j) {
        double M[3][3];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) M[a][b] = b == j ? r[a] : J[a][b];
        lam[j] = (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                  M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
      }
      double l0 = 1.0;
      bool inside = true;
      for (int j = 0; j < dim; ++j) {
        l0 -= lam[j];
        inside = inside && lam[j] >= -1e-12;
      }
      if (!inside || l0 < -1e-12) continue;
      const int32_t *cd = d->cell_dofs.data() + (size_t)c * d->dpc;
      double p = l0 * solution[cd[dim]];
      for (int j = 0; j < dim; ++j) p += lam[j] * solution[cd[(size_t)(dim + 1) * (j + 1) + dim]];
      val[k] = p;
      found[k] = true;
    }
  }
  *diff = val[0] - val[1];
  return (int)found[0] + (int)found[1];
}

}  // extern "C"

// ---------------------------------------------------------------- test hooks: schedule of the packed ILU(0) solve
// (host/ilu_stream.hpp is what the device library's set-up runs; these entry points let the CPU tests build a stream for any
// graph / block table and replay it tick by tick against plain sequential sweeps)
namespace {
nsx::Csr csr_of(int n_rows, const int32_t *rowptr, const int32_t *colind) {
  nsx::Csr g;
  g.n_rows = g.n_cols = n_rows;
  g.rowptr.assign(rowptr, rowptr + n_rows + 1);
  g.colind.assign(colind, colind + rowptr[n_rows]);
  return g;
}
}  // namespace

extern "C" int nsxh_ilu_stream_stats(int n_rows, const int32_t *rowptr, const int32_t *colind, int n_blocks, const int32_t *block_ptr,
                                     int blocks_per_wave, int ncomp, int gap, int entries_per_tick, int64_t out[6]) {
  if (!rowptr || !colind || !block_ptr || !out || n_rows < 0 || n_blocks < 1 || ncomp < 1 || gap < 1 || entries_per_tick < 1 || entries_per_tick > 4) return -1;
  try {
    nsx::IluStream s;
    nsx::build_ilu_stream(csr_of(n_rows, rowptr, colind), std::vector<int32_t>(block_ptr, block_ptr + n_blocks + 1), blocks_per_wave, ncomp, gap, s, entries_per_tick);
    out[0] = s.n_slabs, out[1] = s.max_wave_slabs, out[2] = s.in_block_nnz, out[3] = s.max_wave_rows, out[4] = s.n_waves, out[5] = s.used_slots;
    return s.ok ? 0 : -3;
  } catch (const std::exception &) {
    return -1;
  }
}

extern "C" int nsxh_ilu_stream_apply(int n_rows, const int32_t *rowptr, const int32_t *colind, int n_blocks, const int32_t *block_ptr,
                                     int blocks_per_wave, int ncomp, int gap, int entries_per_tick, const double *lu, const double *b, double *x) {
  if (!rowptr || !colind || !block_ptr || !lu || !b || !x || ncomp < 1 || gap < 1 || entries_per_tick < 1 || entries_per_tick > 4) return -1;
  try {
    const nsx::Csr g = csr_of(n_rows, rowptr, colind);
    const std::vector<int32_t> bptr(block_ptr, block_ptr + n_blocks + 1);
    nsx::IluStream s;
    nsx::build_ilu_stream(g, bptr, blocks_per_wave, ncomp, gap, s, entries_per_tick);
    if (!s.ok) return -3;
    nsx::replay_ilu_stream(g, bptr, s, lu, b, x);
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

// ---------------------------------------------------------------- test hook: the internal layout of the device library
// (host/layout.hpp is what nsx_set_internal_layout runs; this entry point lets the CPU tests build a layout for any serial DoF table)
extern "C" int nsxh_internal_layout(int dim, int n_cells, int dofs_per_cell, const int32_t *cell_dofs, const double *cell_coords, int n_u, int n_p,
                                    int n_in_ranks, const int32_t *in_u_ptr, const int32_t *in_p_ptr, int n_virtual, int order, int schur_max_rows,
                                    int32_t *node_perm, int32_t *pnode_perm, int32_t *n_ranks, int32_t *u_ptr, int32_t *p_ptr, int32_t *n_schur,
                                    int32_t *schur_ptr, int32_t *colours) {
  if ((dim != 2 && dim != 3) || n_cells < 1 || !cell_dofs || !cell_coords || n_in_ranks < 1 || !in_u_ptr || !in_p_ptr || n_virtual < 1 || n_u % dim) return -1;
  const int nv = dim + 1, nl = dim == 2 ? 3 : 6, np2 = nv + nl;
  if (dofs_per_cell != nv * (dim + 1) + nl * dim) return -1;
  std::vector<int32_t> c2((size_t)n_cells * np2), c1((size_t)n_cells * nv);
  std::vector<double> cen((size_t)n_cells * dim, 0.0);
  for (int c = 0; c < n_cells; ++c) {
    const int32_t *cd = cell_dofs + (size_t)c * dofs_per_cell;
    for (int a = 0; a < np2; ++a) c2[(size_t)c * np2 + a] = cd[a < nv ? (dim + 1) * a : nv * (dim + 1) + dim * (a - nv)] / dim;
    for (int v = 0; v < nv; ++v) c1[(size_t)c * nv + v] = cd[(dim + 1) * v + dim] - n_u;
    for (int k = 0; k < nv; ++k)
      for (int d = 0; d < dim; ++d) cen[(size_t)c * dim + d] += cell_coords[((size_t)c * nv + k) * dim + d] / nv;
  }
  nsx::LayoutIn in;
  in.dim = dim;
  in.n_cells = n_cells;
  in.np2 = np2;
  in.np1 = nv;
  in.c2 = c2.data();
  in.c1 = c1.data();
  in.cen = cen.data();
  in.N2 = in.N2_all = n_u / dim;
  in.NP = n_p;
  in.in_u_ptr.assign(in_u_ptr, in_u_ptr + n_in_ranks + 1);
  in.in_p_ptr.assign(in_p_ptr, in_p_ptr + n_in_ranks + 1);
  nsx::LayoutOut out;
  nsx::build_layout(in, n_virtual, order, schur_max_rows, out);
  std::copy(out.perm2.begin(), out.perm2.end(), node_perm);
  std::copy(out.perm1.begin(), out.perm1.end(), pnode_perm);
  *n_ranks = (int32_t)out.u_ptr.size() - 1;
  std::copy(out.u_ptr.begin(), out.u_ptr.end(), u_ptr);
  std::copy(out.p_ptr.begin(), out.p_ptr.end(), p_ptr);
  const std::vector<int32_t> &sb = out.schur_ptr.empty() ? out.p_ptr : out.schur_ptr;
  *n_schur = (int32_t)sb.size() - 1;
  std::copy(sb.begin(), sb.end(), schur_ptr);
  if (colours) {
    colours[0] = out.n_colours;
    colours[1] = out.n_colours_p;
  }
  return 0;
}
