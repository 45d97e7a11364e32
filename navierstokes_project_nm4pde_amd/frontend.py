"""numpy view of the host front-end (include/nsx_host.h): meshes, DoF tables, FE tables.

Mirrors the outputs of the reference's ``NavierStokes::setup()``
(reference Navier-Stokes/src/NavierStokes3D.cpp:2-157).
"""
import ctypes as C

import numpy as np

from ._lib import HOST_SO, load

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def _lib():
    lib = load(HOST_SO)
    if getattr(lib, "_nsxh_ready", False):
        return lib
    vp = C.c_void_p
    sig = {
        "nsxh_mesh_cylinder": (vp, [C.c_int] * 8 + [C.c_double] * 2),
        "nsxh_mesh_cylinder_level": (vp, [C.c_int, C.c_int]),
        "nsxh_mesh_cube": (vp, [C.c_int]),
        "nsxh_mesh_box": (vp, [C.c_int] * 4 + [_f64p, _f64p]),
        "nsxh_mesh_read_msh": (vp, [C.c_char_p]),
        "nsxh_mesh_free": (None, [vp]),
        "nsxh_mesh_dim": (C.c_int, [vp]),
        "nsxh_mesh_n_vertices": (C.c_int, [vp]),
        "nsxh_mesh_n_cells": (C.c_int, [vp]),
        "nsxh_mesh_n_bfaces": (C.c_int, [vp]),
        "nsxh_mesh_vertices": (_f64p, [vp]),
        "nsxh_mesh_cells": (_i32p, [vp]),
        "nsxh_mesh_bfaces": (_i32p, [vp]),
        "nsxh_mesh_bface_ids": (_i32p, [vp]),
        "nsxh_mesh_bface_cells": (_i32p, [vp]),
        "nsxh_mesh_subdomain": (_i32p, [vp]),
        "nsxh_mesh_partition": (C.c_int, [vp, C.c_int, C.c_int]),
        "nsxh_mesh_partition_owned": (C.c_int, [vp, C.c_int, C.c_int]),
        "nsxh_distribute_dofs": (vp, [vp]),
        "nsxh_distribute_dofs_ordered": (vp, [vp, C.c_int]),
        "nsxh_n_colours": (C.c_int, [vp]),
        "nsxh_n_colours_p": (C.c_int, [vp]),
        "nsxh_write_vtu": (C.c_int, [vp, C.POINTER(C.c_double), C.c_char_p, C.c_char_p, C.c_uint]),
        "nsxh_pressure_difference": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "nsxh_dofs_free": (None, [vp]),
        "nsxh_dofs_per_cell": (C.c_int, [vp]),
        "nsxh_n_nodes_p2": (C.c_int, [vp]),
        "nsxh_n_nodes_p1": (C.c_int, [vp]),
        "nsxh_n_u": (C.c_int, [vp]),
        "nsxh_n_p": (C.c_int, [vp]),
        "nsxh_cell_dofs": (_i32p, [vp]),
        "nsxh_cell_coords": (_f64p, [vp]),
        "nsxh_support_points": (_f64p, [vp]),
        "nsxh_node_owner": (_i32p, [vp]),
        "nsxh_pnode_owner": (_i32p, [vp]),
        "nsxh_owned_u_ptr": (_i32p, [vp]),
        "nsxh_owned_p_ptr": (_i32p, [vp]),
        "nsxh_n_subdomains": (C.c_int, [vp]),
        "nsxh_boundary_dofs": (C.c_int, [vp, C.c_int, C.POINTER(_i32p)]),
        "nsxh_reference_sparsity": (C.c_int, [vp, C.c_int, C.POINTER(_i32p), C.POINTER(_i32p)]),
        "nsxh_tables_create": (vp, [C.c_int, C.c_int, C.c_int]),
        "nsxh_tables_free": (None, [vp]),
        "nsxh_tables_n_q": (C.c_int, [vp]),
        "nsxh_tables_n_qf": (C.c_int, [vp]),
        "nsxh_tables_n_p2": (C.c_int, [vp]),
        "nsxh_tables_n_p1": (C.c_int, [vp]),
        "nsxh_tables_points": (_f64p, [vp]),
        "nsxh_tables_weights": (_f64p, [vp]),
        "nsxh_tables_N2": (_f64p, [vp]),
        "nsxh_tables_dN2": (_f64p, [vp]),
        "nsxh_tables_N1": (_f64p, [vp]),
        "nsxh_tables_dN1": (_f64p, [vp]),
        "nsxh_ilu_stream_stats": (C.c_int, [C.c_int, _i32p, _i32p, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
        "nsxh_ilu_stream_apply": (C.c_int, [C.c_int, _i32p, _i32p, C.c_int, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p]),
        "nsxh_internal_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, _i32p, _f64p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, C.c_int, C.c_int,
                                           _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p, _i32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._nsxh_ready = True
    return lib


def _arr(ptr, shape, dtype, copy=True):
    """`shape` items of `dtype` behind a C pointer.  copy=False returns a VIEW of the C++ object's storage (valid while
    that object lives): first-touch page faults make a 100-MB copy cost seconds in sandboxed containers."""
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    buf = (C.c_char * nbytes).from_address(C.cast(ptr, C.c_void_p).value)
    a = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    return a.copy() if copy else a


class Mesh:
    """Simplicial mesh with boundary ids (0 inlet, 1 outlet, 2 walls, 3 obstacle)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("mesh construction failed (bad parameters or unreadable file)")
        self._h = handle
        self._lib = _lib()
        self.refresh()

    def refresh(self):
        L, h = self._lib, self._h
        self.dim = L.nsxh_mesh_dim(h)
        nv, nc, nb = L.nsxh_mesh_n_vertices(h), L.nsxh_mesh_n_cells(h), L.nsxh_mesh_n_bfaces(h)
        self.vertices = _arr(L.nsxh_mesh_vertices(h), (nv, self.dim), np.float64, copy=False)   # views: owned by self._h
        self.cells = _arr(L.nsxh_mesh_cells(h), (nc, self.dim + 1), np.int32, copy=False)
        self.bfaces = _arr(L.nsxh_mesh_bfaces(h), (nb, self.dim), np.int32)
        self.bface_ids = _arr(L.nsxh_mesh_bface_ids(h), (nb,), np.int32)
        self.bface_cells = _arr(L.nsxh_mesh_bface_cells(h), (nb,), np.int32)
        self.subdomain = _arr(L.nsxh_mesh_subdomain(h), (nc,), np.int32)

    @classmethod
    def cylinder(cls, dim, level=1, **kw):
        L = _lib()
        if kw:
            a = dict(m=4, nr=2, nxu=2, nxd=10, nyb=1, nyt=1, nz=4, grade_x=2.0, grade_r=2.0)
            a.update(kw)
            return cls(L.nsxh_mesh_cylinder(dim, a["m"], a["nr"], a["nxu"], a["nxd"], a["nyb"], a["nyt"], a["nz"],
                                            a["grade_x"], a["grade_r"]))
        return cls(L.nsxh_mesh_cylinder_level(dim, level))

    @classmethod
    def cube(cls, n):
        return cls(_lib().nsxh_mesh_cube(n))

    @classmethod
    def box(cls, dim, n, lo=None, hi=None):
        n = list(n) + [1] * (3 - len(n))
        lo = np.asarray(lo if lo is not None else [0.0] * dim, dtype=np.float64)
        hi = np.asarray(hi if hi is not None else [1.0] * dim, dtype=np.float64)
        return cls(_lib().nsxh_mesh_box(dim, n[0], n[1], n[2], lo.ctypes.data_as(_f64p), hi.ctypes.data_as(_f64p)))

    @classmethod
    def read_msh(cls, path):
        return cls(_lib().nsxh_mesh_read_msh(str(path).encode()))

    def partition(self, n_parts=1, n_sub=1, balance="cells"):
        """balance = "cells": equal cell counts (METIS-like); "owned": equal numbers of owned nodes per subdomain."""
        fn = self._lib.nsxh_mesh_partition_owned if balance == "owned" else self._lib.nsxh_mesh_partition
        rc = fn(self._h, n_parts, n_sub)
        if rc != 0:
            raise ValueError("partition failed (rc=%d)" % rc)
        self.refresh()
        return self

    def __del__(self):
        try:
            self._lib.nsxh_mesh_free(self._h)
        except Exception:
            pass


class DoFs:
    """Taylor-Hood P2/P1 DoF tables (FESystem local order; velocity block then pressure block)."""

    ORDERINGS = {"first_touch": 0, "colour": 1, "colour_all": 2}

    def __init__(self, mesh, ordering="first_touch"):
        L = _lib()
        self._lib, self.mesh = L, mesh
        self.ordering = ordering
        self._h = h = L.nsxh_distribute_dofs_ordered(mesh._h, self.ORDERINGS[ordering])
        if not h:
            raise ValueError("nsxh_distribute_dofs_ordered failed")
        self.n_colours = L.nsxh_n_colours(h)
        self.n_colours_p = L.nsxh_n_colours_p(h)
        self.dim = mesh.dim
        nc = mesh.cells.shape[0]
        self.n_cells = nc
        self.dofs_per_cell = L.nsxh_dofs_per_cell(h)
        self.n_nodes_p2, self.n_nodes_p1 = L.nsxh_n_nodes_p2(h), L.nsxh_n_nodes_p1(h)
        self.n_u, self.n_p = L.nsxh_n_u(h), L.nsxh_n_p(h)
        self.n_dofs = self.n_u + self.n_p
        self.cell_dofs = _arr(L.nsxh_cell_dofs(h), (nc, self.dofs_per_cell), np.int32, copy=False)      # views: owned by self._h
        self.cell_coords = _arr(L.nsxh_cell_coords(h), (nc, self.dim + 1, self.dim), np.float64, copy=False)
        self.support_points = _arr(L.nsxh_support_points(h), (self.n_dofs, self.dim), np.float64, copy=False)
        self.n_subdomains = L.nsxh_n_subdomains(h)
        self.node_owner = _arr(L.nsxh_node_owner(h), (self.n_nodes_p2,), np.int32)
        self.pnode_owner = _arr(L.nsxh_pnode_owner(h), (self.n_nodes_p1,), np.int32)
        self.owned_u_ptr = _arr(L.nsxh_owned_u_ptr(h), (self.n_subdomains + 1,), np.int32)
        self.owned_p_ptr = _arr(L.nsxh_owned_p_ptr(h), (self.n_subdomains + 1,), np.int32)

    def write_vtu(self, solution, directory, basename, counter):
        """NavierStokes::output: <directory>/<basename>_<counter>.0.vtu + .pvtu (reference NavierStokes3D.cpp:643-683)."""
        x = np.ascontiguousarray(solution, dtype=np.float64)
        assert x.shape == (self.n_dofs,)
        if self._lib.nsxh_write_vtu(self._h, x.ctypes.data_as(C.POINTER(C.c_double)), str(directory).encode(), basename.encode(), int(counter)):
            raise OSError("nsxh_write_vtu failed for %s" % directory)

    def pressure_difference(self, solution, point_a, point_b):
        """NavierStokes::compute_pressure_difference (reference NavierStokes3D.cpp:849-923): (p(a) - p(b), points found)."""
        x = np.ascontiguousarray(solution, dtype=np.float64)
        a, b = (np.ascontiguousarray(p, dtype=np.float64) for p in (point_a, point_b))
        out = C.c_double(0.0)
        dp = C.POINTER(C.c_double)
        n = self._lib.nsxh_pressure_difference(self._h, x.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp), C.byref(out))
        return out.value, n

    def boundary_dofs(self, boundary_id):
        p = _i32p()
        n = self._lib.nsxh_boundary_dofs(self._h, int(boundary_id), C.byref(p))
        return _arr(p, (n,), np.int32)

    def reference_sparsity(self, block):
        """CSR graph of block 0=(0,0), 1=(0,1), 2=(1,0), 3=pressure mass, in the reference's padded layout."""
        rp, ci = _i32p(), _i32p()
        n = self._lib.nsxh_reference_sparsity(self._h, int(block), C.byref(rp), C.byref(ci))
        rowptr = _arr(rp, (n + 1,), np.int32)
        return rowptr, _arr(ci, (int(rowptr[-1]),), np.int32)

    def rank_view(self, rank, world):
        """Per-GPU view (owned + two ghost layers of cells, halo plan) for a run with `world` processes."""
        L = self._lib
        vp = C.c_void_p
        if not getattr(L, "_rv_ready", False):
            L.nsxh_rank_view_create.restype = vp
            L.nsxh_rank_view_create.argtypes = [vp, C.c_int, C.c_int]
            L.nsxh_rank_view_free.argtypes = [vp]
            for name in ("n_cells", "n_cells_layer1", "n_virtual_ranks", "n_neighbors"):
                f = getattr(L, "nsxh_rank_view_" + name)
                f.restype, f.argtypes = C.c_int, [vp]
            for name in ("cell_ids", "cell_dofs", "gpu_u_ptr", "gpu_p_ptr", "rank_u_ptr", "rank_p_ptr", "neighbors",
                         "send_u_ptr", "send_u_nodes", "send_p_ptr", "send_p_nodes"):
                f = getattr(L, "nsxh_rank_view_" + name)
                f.restype, f.argtypes = _i32p, [vp]
            L.nsxh_rank_view_cell_coords.restype = _f64p
            L.nsxh_rank_view_cell_coords.argtypes = [vp]
            L._rv_ready = True
        v = L.nsxh_rank_view_create(self._h, rank, world)
        if not v:
            raise ValueError("rank view needs n_subdomains to be a multiple of world")
        nc, nn, ns = L.nsxh_rank_view_n_cells(v), L.nsxh_rank_view_n_neighbors(v), L.nsxh_rank_view_n_virtual_ranks(v)
        out = {
            "rank": rank, "world": world, "n_cells": nc, "n_cells_layer1": L.nsxh_rank_view_n_cells_layer1(v),
            "cell_ids": _arr(L.nsxh_rank_view_cell_ids(v), (nc,), np.int32),
            "cell_dofs": _arr(L.nsxh_rank_view_cell_dofs(v), (nc, self.dofs_per_cell), np.int32),
            "cell_coords": _arr(L.nsxh_rank_view_cell_coords(v), (nc, self.dim + 1, self.dim), np.float64),
            "gpu_u_ptr": _arr(L.nsxh_rank_view_gpu_u_ptr(v), (world + 1,), np.int32),
            "gpu_p_ptr": _arr(L.nsxh_rank_view_gpu_p_ptr(v), (world + 1,), np.int32),
            "rank_u_ptr": _arr(L.nsxh_rank_view_rank_u_ptr(v), (ns + 1,), np.int32),
            "rank_p_ptr": _arr(L.nsxh_rank_view_rank_p_ptr(v), (ns + 1,), np.int32),
            "neighbors": _arr(L.nsxh_rank_view_neighbors(v), (nn,), np.int32),
            "send_u_ptr": _arr(L.nsxh_rank_view_send_u_ptr(v), (nn + 1,), np.int32),
            "send_p_ptr": _arr(L.nsxh_rank_view_send_p_ptr(v), (nn + 1,), np.int32),
        }
        out["send_u_nodes"] = _arr(L.nsxh_rank_view_send_u_nodes(v), (int(out["send_u_ptr"][-1]),), np.int32)
        out["send_p_nodes"] = _arr(L.nsxh_rank_view_send_p_nodes(v), (int(out["send_p_ptr"][-1]),), np.int32)
        L.nsxh_rank_view_free(v)
        return out

    def __del__(self):
        try:
            self._lib.nsxh_dofs_free(self._h)
        except Exception:
            pass


class Tables:
    """Reference-element shape tables at quadrature points (data, not code, for the device kernels)."""

    CELL, FACE, HIGH = 0, 1, 2

    def __init__(self, dim, rule=0, order=0):
        L = _lib()
        h = L.nsxh_tables_create(dim, rule, order)
        if not h:
            raise ValueError("bad table request")
        self.dim = dim
        nq, n2, n1 = L.nsxh_tables_n_q(h), L.nsxh_tables_n_p2(h), L.nsxh_tables_n_p1(h)
        self.n_q, self.n_p2, self.n_p1, self.n_qf = nq, n2, n1, L.nsxh_tables_n_qf(h)
        self.points = _arr(L.nsxh_tables_points(h), (nq, dim), np.float64)
        self.weights = _arr(L.nsxh_tables_weights(h), (nq,), np.float64)
        self.N2 = _arr(L.nsxh_tables_N2(h), (nq, n2), np.float64)
        self.dN2 = _arr(L.nsxh_tables_dN2(h), (nq, n2, dim), np.float64)
        self.N1 = _arr(L.nsxh_tables_N1(h), (nq, n1), np.float64)
        self.dN1 = _arr(L.nsxh_tables_dN1(h), (nq, n1, dim), np.float64)
        L.nsxh_tables_free(h)


def ilu_stream_stats(rowptr, colind, block_ptr, blocks_per_wave=8, ncomp=3, gap=2, entries_per_tick=1):
    """Schedule statistics of the packed ILU(0) solve for a graph / block table (include/nsx_host.h: nsxh_ilu_stream_stats)."""
    rp, ci, bp = (np.ascontiguousarray(a, dtype=np.int32) for a in (rowptr, colind, block_ptr))
    out = (C.c_int64 * 6)()
    rc = _lib().nsxh_ilu_stream_stats(len(rp) - 1, rp.ctypes.data_as(_i32p), ci.ctypes.data_as(_i32p), len(bp) - 1, bp.ctypes.data_as(_i32p),
                                      int(blocks_per_wave), int(ncomp), int(gap), int(entries_per_tick), out)
    if rc:
        raise ValueError("nsxh_ilu_stream_stats failed (%d)" % rc)
    keys = ("slabs", "max_wave_slabs", "in_block_nnz", "max_wave_rows", "waves", "used_slots")
    d = dict(zip(keys, (int(v) for v in out)))
    d["fill"] = d["used_slots"] / max(1, 64 * d["slabs"] * int(entries_per_tick))
    d["stream_bytes"] = d["slabs"] * 64 * (8 * int(entries_per_tick) + 4 * ((int(entries_per_tick) + 2) // 2))
    return d


def ilu_stream_apply(rowptr, colind, block_ptr, lu, b, ncomp=1, blocks_per_wave=8, gap=2, entries_per_tick=1):
    """Host replay of the packed ILU(0) solve stream (nsxh_ilu_stream_apply): x = U^-1 D^-1 L^-1 b per block."""
    rp, ci, bp = (np.ascontiguousarray(a, dtype=np.int32) for a in (rowptr, colind, block_ptr))
    lu, b = np.ascontiguousarray(lu, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty_like(b)
    rc = _lib().nsxh_ilu_stream_apply(len(rp) - 1, rp.ctypes.data_as(_i32p), ci.ctypes.data_as(_i32p), len(bp) - 1, bp.ctypes.data_as(_i32p),
                                      int(blocks_per_wave), int(ncomp), int(gap), int(entries_per_tick), lu.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p),
                                      x.ctypes.data_as(_f64p))
    if rc:
        raise ValueError("nsxh_ilu_stream_apply failed (%d)" % rc)
    return x


def merge_ranks(ptr, max_rows):
    """Coarser ILU blocks as unions of CONSECUTIVE ranks: ranks are added to a block while it stays within `max_rows` rows
    (a rank larger than that stays a block of its own).  ptr: [n_ranks + 1] node ranges; returns the block table."""
    ptr = np.asarray(ptr)
    out = [int(ptr[0])]
    for k in range(1, len(ptr)):
        if ptr[k] - out[-1] > max_rows and ptr[k - 1] > out[-1]:
            out.append(int(ptr[k - 1]))
    if out[-1] != ptr[-1]:
        out.append(int(ptr[-1]))
    return np.array(out, dtype=np.int32)


def internal_layout(dofs, n_virtual, order="colour", schur_max_rows=0):
    """The layout libnsx gives itself behind nsx_set_internal_layout (host/layout.hpp) for the serial DoF table `dofs` whose
    subdomains are the caller's ranks: dict with node_perm / pnode_perm (caller node -> internal node), u_ptr / p_ptr (internal
    node ranges of the virtual ranks), schur_ptr (Schur ILU blocks), colours."""
    L = _lib()
    cd = np.ascontiguousarray(dofs.cell_dofs, dtype=np.int32)
    cc = np.ascontiguousarray(dofs.cell_coords, dtype=np.float64)
    iu, ip = (np.ascontiguousarray(a, dtype=np.int32) for a in (dofs.owned_u_ptr, dofs.owned_p_ptr))
    cap = int(n_virtual) + len(iu) + 1
    perm2, perm1 = np.empty(dofs.n_nodes_p2, np.int32), np.empty(dofs.n_nodes_p1, np.int32)
    u_ptr, p_ptr, s_ptr = (np.empty(cap, np.int32) for _ in range(3))
    nr, ns, col = C.c_int32(), C.c_int32(), (C.c_int32 * 2)()
    rc = L.nsxh_internal_layout(dofs.dim, dofs.n_cells, dofs.dofs_per_cell, cd.ctypes.data_as(_i32p), cc.ctypes.data_as(_f64p), dofs.n_u, dofs.n_p,
                                len(iu) - 1, iu.ctypes.data_as(_i32p), ip.ctypes.data_as(_i32p), int(n_virtual), DoFs.ORDERINGS[order], int(schur_max_rows),
                                perm2.ctypes.data_as(_i32p), perm1.ctypes.data_as(_i32p), C.byref(nr), u_ptr.ctypes.data_as(_i32p),
                                p_ptr.ctypes.data_as(_i32p), C.byref(ns), s_ptr.ctypes.data_as(_i32p), col)
    if rc:
        raise ValueError("nsxh_internal_layout failed (%d)" % rc)
    return {"node_perm": perm2, "pnode_perm": perm1, "u_ptr": u_ptr[:nr.value + 1].copy(), "p_ptr": p_ptr[:nr.value + 1].copy(),
            "schur_ptr": s_ptr[:ns.value + 1].copy(), "colours": (int(col[0]), int(col[1]))}


class PermutedDoFs:
    """The DoF table `dofs` with its P2 / P1 nodes renumbered (node_perm / pnode_perm: old node -> new node) and a new rank table:
    the same mesh, cells in the same order, every index array mapped.  What a caller would hand over had it numbered its DoFs the way
    libnsx's internal layout does; the oracle runs on it when the device was given `dofs` + nsx_set_internal_layout."""

    def __init__(self, dofs, node_perm, pnode_perm, u_ptr, p_ptr):
        self.base, self.mesh = dofs, dofs.mesh
        self.dim, self.n_cells, self.dofs_per_cell = dofs.dim, dofs.n_cells, dofs.dofs_per_cell
        self.n_nodes_p2, self.n_nodes_p1, self.n_u, self.n_p, self.n_dofs = dofs.n_nodes_p2, dofs.n_nodes_p1, dofs.n_u, dofs.n_p, dofs.n_dofs
        self.node_perm, self.pnode_perm = np.asarray(node_perm, dtype=np.int64), np.asarray(pnode_perm, dtype=np.int64)
        dim = self.dim
        dmap = np.empty(self.n_dofs, dtype=np.int64)      # old dof -> new dof
        for c in range(dim):
            dmap[c:self.n_u:dim] = dim * self.node_perm + c
        dmap[self.n_u:] = self.n_u + self.pnode_perm
        self.dof_map = dmap
        self.cell_dofs = dmap[np.asarray(dofs.cell_dofs)].astype(np.int32)
        self.cell_coords = dofs.cell_coords
        sp = np.empty_like(np.asarray(dofs.support_points))
        sp[dmap] = dofs.support_points
        self.support_points = sp
        self.owned_u_ptr, self.owned_p_ptr = np.asarray(u_ptr, dtype=np.int32), np.asarray(p_ptr, dtype=np.int32)
        self.n_subdomains = len(self.owned_u_ptr) - 1
        self._ref = {}

    def to_new(self, x):
        """vector in the old numbering -> the same finite-element function in the new one"""
        out = np.empty_like(np.asarray(x, dtype=np.float64))
        out[self.dof_map] = x
        return out

    def to_old(self, y):
        return np.asarray(y)[self.dof_map]

    def boundary_dofs(self, boundary_id):
        return np.sort(self.dof_map[self.base.boundary_dofs(boundary_id)]).astype(np.int32)

    def reference_sparsity(self, block):
        if block not in self._ref:
            rp, ci = self.base.reference_sparsity(block)
            nu = self.n_u
            rmap = self.dof_map[:nu] if block in (0, 1) else self.dof_map[nu:] - nu
            cmap = self.dof_map[:nu] if block in (0, 2) else self.dof_map[nu:] - nu
            n = len(rp) - 1
            rows = np.repeat(np.arange(n), np.diff(rp))
            nr, ncol = rmap[rows], cmap[ci]
            order = np.lexsort((ncol, nr))
            cnt = np.bincount(nr, minlength=n)
            self._ref[block] = (np.concatenate(([0], np.cumsum(cnt))).astype(np.int32), ncol[order].astype(np.int32))
        return self._ref[block]
