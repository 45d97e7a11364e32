"""ctypes binding of the device library (include/nsx.h -> csrc/libnsx.so).

Plumbing for tests and bench.py only.  There is no CPU fallback: without a HIP device `Nsx(...)` raises,
and without the built library the import of the symbols raises.
"""
import ctypes as C

import numpy as np

from ._lib import DEV_SO, load

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)

TEMAM, DOUBLE_CONVECTION = 1, 2
YOSIDA, SIMPLE, AYOSIDA, ASIMPLE = 0, 1, 2, 3

# every symbol include/nsx.h declares (tests check that the library exports all of them)
API = [
    "nsx_create", "nsx_destroy", "nsx_last_error", "nsx_version", "nsx_set_tables", "nsx_set_mesh", "nsx_set_ranks",
    "nsx_set_schur_blocks", "nsx_set_solution", "nsx_get_solution", "nsx_get_solution_ghosted", "nsx_get_rhs",
    "nsx_set_rhs", "nsx_assemble", "nsx_assemble_time_step", "nsx_add_rhs", "nsx_apply_boundary_values",
    "nsx_solve_time_step", "nsx_prec_initialize", "nsx_prec_vmult", "nsx_system_vmult", "nsx_ilu_apply",
    "nsx_export_block", "nsx_schur_nnz", "nsx_schur_get", "nsx_scalar_graph_nnz", "nsx_scalar_graph", "nsx_ilu_get",
    "nsx_profile_enable", "nsx_profile_reset", "nsx_profile_count", "nsx_profile_get", "nsx_persistent_state", "nsx_path_info", "nsx_comm_self_halo_test", "nsx_comm_unique_id",
    "nsx_comm_init", "nsx_comm_init_callbacks", "nsx_comm_counters", "nsx_set_mesh_distributed", "nsx_set_force_faces", "nsx_compute_forces",
    "nsx_set_internal_layout", "nsx_layout_info", "nsx_layout_get", "nsx_gram_schmidt_cycle",
]
FIRST_TOUCH, COLOUR, COLOUR_ALL = 0, 1, 2  # node order of nsx_set_internal_layout

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _f64p, C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_f64p), C.POINTER(C.c_int),
                          C.POINTER(_f64p), C.POINTER(C.c_int))


class Params(C.Structure):
    _fields_ = [("dim", C.c_int), ("device", C.c_int), ("nu", C.c_double), ("deltat", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int), ("inner_F_iterations", C.c_int), ("inner_S_iterations", C.c_int),
                ("n_F_solves", C.c_int), ("n_S_solves", C.c_int), ("final_residual", C.c_double),
                ("t_prec", C.c_double), ("t_solve", C.c_double), ("status", C.c_int), ("persistent_fallbacks", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class NsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("nsx error %d: %s" % (code, msg))
        self.code = code


def lib():
    L = load(DEV_SO)
    if getattr(L, "_nsx_ready", False):
        return L
    vp = C.c_void_p
    L.nsx_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.nsx_destroy.argtypes = [vp]
    L.nsx_last_error.restype = C.c_char_p
    L.nsx_last_error.argtypes = [vp]
    L.nsx_version.restype = C.c_char_p
    L.nsx_set_tables.argtypes = [vp, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p]
    L.nsx_set_mesh.argtypes = [vp, C.c_int, C.c_int, _i32p, _f64p, C.c_int, C.c_int]
    L.nsx_set_ranks.argtypes = [vp, C.c_int, _i32p, _i32p]
    L.nsx_set_schur_blocks.argtypes = [vp, C.c_int, _i32p]
    L.nsx_set_internal_layout.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.nsx_layout_info.argtypes = [vp, C.POINTER(C.c_int)]
    L.nsx_layout_get.argtypes = [vp, _i32p, _i32p, _i32p, _i32p, _i32p]
    L.nsx_gram_schmidt_cycle.argtypes = [vp, C.c_int, C.c_int, _f64p, C.c_double, _f64p, _f64p]
    for f in ("nsx_set_solution", "nsx_get_solution", "nsx_get_solution_ghosted", "nsx_get_rhs", "nsx_set_rhs"):
        getattr(L, f).argtypes = [vp, _f64p]
    L.nsx_assemble.argtypes = [vp, C.c_int]
    L.nsx_assemble_time_step.argtypes = [vp, C.c_int]
    L.nsx_add_rhs.argtypes = [vp, C.c_int, _i32p, _f64p]
    L.nsx_apply_boundary_values.argtypes = [vp, C.c_int, _i32p, _f64p]
    L.nsx_solve_time_step.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(Stats)]
    L.nsx_prec_initialize.argtypes = [vp, C.c_int]
    L.nsx_prec_vmult.argtypes = [vp, C.c_int, C.c_double, C.c_int, _f64p, _f64p, C.POINTER(Stats)]
    L.nsx_system_vmult.argtypes = [vp, _f64p, _f64p]
    L.nsx_ilu_apply.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.nsx_export_block.argtypes = [vp, C.c_int, C.c_int, C.c_int, _i32p, _i32p, _f64p]
    L.nsx_schur_nnz.argtypes = [vp, C.POINTER(C.c_int64)]
    L.nsx_schur_get.argtypes = [vp, _i32p, _i32p, _f64p]
    L.nsx_scalar_graph_nnz.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.nsx_scalar_graph.argtypes = [vp, C.c_int, _i32p, _i32p]
    L.nsx_ilu_get.argtypes = [vp, C.c_int, _f64p]
    L.nsx_profile_enable.argtypes = [vp, C.c_int]
    L.nsx_profile_reset.argtypes = [vp]
    L.nsx_profile_count.argtypes = [vp]
    L.nsx_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), _f64p, _f64p]
    L.nsx_persistent_state.argtypes = [vp, C.POINTER(C.c_int)]
    L.nsx_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.nsx_comm_init.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    L.nsx_comm_init_callbacks.argtypes = [vp, C.c_int, C.c_int, ALLREDUCE_FN, EXCHANGE_FN, C.c_void_p]
    L.nsx_comm_counters.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.nsx_set_mesh_distributed.argtypes = [vp, C.c_int, C.c_int, C.c_int, _i32p, _f64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           _i32p, _i32p, C.c_int, _i32p, _i32p, _i32p, _i32p, _i32p]
    L.nsx_set_force_faces.argtypes = [vp, C.c_int, _i32p, _i32p, C.c_int, _f64p, _f64p, _f64p, _f64p]
    L.nsx_compute_forces.argtypes = [vp, _f64p, _f64p]
    L._nsx_ready = True
    return L


def _i(a):
    return a.ctypes.data_as(_i32p)


def _d(a):
    return a.ctypes.data_as(_f64p)


def _ci(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _cd(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def gloo_callbacks():
    """Host-buffer communicator callbacks on torch.distributed (any backend with CPU tensors, e.g. gloo)."""
    import torch
    import torch.distributed as dist

    def allreduce(ctx, buf, count):
        t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(count,)))
        dist.all_reduce(t)
        return 0

    def exchange(ctx, n, ranks, send, scount, recv, rcount):
        reqs, keep = [], []
        for k in range(n):
            if rcount[k]:
                t = torch.from_numpy(np.ctypeslib.as_array(recv[k], shape=(rcount[k],)))
                keep.append(t)
                reqs.append(dist.irecv(t, src=int(ranks[k])))
            if scount[k]:
                t = torch.from_numpy(np.ctypeslib.as_array(send[k], shape=(scount[k],)).copy())
                keep.append(t)
                reqs.append(dist.isend(t, dst=int(ranks[k])))
        for q in reqs:
            q.wait()
        return 0

    return ALLREDUCE_FN(allreduce), EXCHANGE_FN(exchange)


class Nsx:
    """One device-side `NavierStokes` problem (a handle of libnsx)."""

    def __init__(self, dofs, tables, nu, deltat, device=0, rank=0, world=1, comm="rccl", layout=None):
        """world == 1: the whole problem on one GPU.  world > 1: this process holds rank `rank` of a run with one
        process per GPU (mesh partitioned with Mesh.partition(world, n_sub)); `comm` = "rccl" (needs an initialised
        torch.distributed group to broadcast the unique id) or "callbacks" (host buffers over torch.distributed).
        layout = (n_virtual_ranks, order, schur_max_rows): nsx_set_internal_layout, requested before the mesh is handed over."""
        L = lib()
        self.L = L
        self._h = C.c_void_p()
        prm = Params(dofs.dim, device, float(nu), float(deltat))
        rc = L.nsx_create(C.byref(prm), C.byref(self._h))
        if rc:
            raise NsxError(rc, (L.nsx_last_error(None) or b"").decode())
        self.dim, self.n_u, self.n_p = dofs.dim, dofs.n_u, dofs.n_p
        self.n = self.n_u + self.n_p
        self.dofs = dofs
        self.rank, self.world = rank, world
        N2, dN2, N1, w = _cd(tables.N2), _cd(tables.dN2), _cd(tables.N1), _cd(tables.weights)
        self._ck(L.nsx_set_tables(self._h, tables.n_q, tables.n_p2, tables.n_p1, _d(N2), _d(dN2), _d(N1), _d(w)))
        if layout:
            self.set_internal_layout(*layout)
        if world == 1:
            cd, cc = _ci(dofs.cell_dofs), _cd(dofs.cell_coords)
            self._ck(L.nsx_set_mesh(self._h, dofs.n_cells, dofs.dofs_per_cell, _i(cd), _d(cc), dofs.n_u, dofs.n_p))
            if dofs.n_subdomains > 1:
                self.set_ranks(dofs.owned_u_ptr, dofs.owned_p_ptr)
            return
        v = dofs.rank_view(rank, world)
        self.view = v
        cd, cc = _ci(v["cell_dofs"]), _cd(v["cell_coords"])
        arrs = [_ci(v[k]) for k in ("gpu_u_ptr", "gpu_p_ptr", "neighbors", "send_u_ptr", "send_u_nodes", "send_p_ptr", "send_p_nodes")]
        self._ck(L.nsx_set_mesh_distributed(self._h, v["n_cells"], v["n_cells_layer1"], dofs.dofs_per_cell, _i(cd), _d(cc),
                                            dofs.n_u, dofs.n_p, world, rank, _i(arrs[0]), _i(arrs[1]), len(arrs[2]), _i(arrs[2]),
                                            _i(arrs[3]), _i(arrs[4]), _i(arrs[5]), _i(arrs[6])))
        if len(v["rank_u_ptr"]) > 2:  # one rank of the caller per GPU is the handle's default table
            self.set_ranks(v["rank_u_ptr"], v["rank_p_ptr"])
        if comm == "callbacks":
            self._cb = gloo_callbacks()           # keep the CFUNCTYPE objects alive
            self._ck(L.nsx_comm_init_callbacks(self._h, rank, world, self._cb[0], self._cb[1], None))
        else:
            import torch.distributed as dist
            ident = (C.c_uint8 * 128)()
            if rank == 0:
                self._ck(L.nsx_comm_unique_id(ident))
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            self._ck(L.nsx_comm_init(self._h, rank, world, ident))

    def persistent_state(self):
        """dict: sweep / cg on their single-launch path, time-outs so far, non-empty mailbox words (nsx_persistent_state)"""
        st = (C.c_int * 4)()
        self._ck(self.L.nsx_persistent_state(self._h, st))
        return {"sweep_persistent": bool(st[0]), "cg_persistent": bool(st[1]), "fallbacks": int(st[2]), "dirty_mailbox_words": int(st[3])}

    PATH_KEYS = ("spmv_lds_staged", "spmv_chunks", "spmv_chunks_behind_halo", "sweep_entries_per_thread", "sweep_grid", "sweep_collective_inside",
                 "sweep_entries_per_thread_max", "cus_reserved", "schur_cg_path", "schur_blocks", "neighbours", "nodes_sent_per_exchange", "ghost_nodes",
                 "schur_dense_inverses", "sweep_off", "fallbacks", "rccl_sweep_velocity_plain", "rccl_sweep_velocity_masked", "rccl_sweep_block_plain",
                 "rccl_sweep_block_masked", "schur_blocks_per_partial", "sweep_velocity_one_gpu", "owned_p2_nodes", "owned_p1_nodes", "sweep_with_ilu_inside", "fused_launches")

    def path_info(self):
        """dict: which code paths the handle's products and solves take (nsx_path_info; schur_cg_path: 1 launch per operation,
        2 persistent, 3 two launches per iteration)"""
        v = (C.c_int * 32)()
        self._ck(self.L.nsx_path_info(self._h, v))
        return {k: int(v[i]) for i, k in enumerate(self.PATH_KEYS)}

    def comm_counters(self):
        """(all-reduces, ghost exchanges) issued since the communicator was set"""
        c = (C.c_longlong * 2)()
        self._ck(self.L.nsx_comm_counters(self._h, c))
        return int(c[0]), int(c[1])

    def comm_self_halo_test(self, n_own, n_ghost, ncomp):
        """largest error of a self-addressed RCCL ghost exchange (nsx_comm_self_halo_test; needs comm_init_single)"""
        e = C.c_double(0.0)
        self._ck(self.L.nsx_comm_self_halo_test(self._h, int(n_own), int(n_ghost), int(ncomp), C.byref(e)))
        return float(e.value)

    def comm_init_single(self):
        """1-rank RCCL communicator on this handle: every dot product then goes through ncclAllReduce (API self-test)."""
        ident = (C.c_uint8 * 128)()
        self._ck(self.L.nsx_comm_unique_id(ident))
        self._ck(self.L.nsx_comm_init(self._h, 0, 1, ident))

    def gather_solution(self):
        """Global solution vector on every rank (owned parts summed over the process group)."""
        x = np.zeros(self.n)
        self._ck(self.L.nsx_get_solution(self._h, _d(x)))
        if self.world > 1:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(x)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t)
            x = t.cpu().numpy()
        return x

    def _ck(self, rc):
        if rc:
            raise NsxError(rc, (self.L.nsx_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self.L.nsx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ---------------------------------------------------------------------------------
    def set_ranks(self, u_ptr, p_ptr):
        u_ptr, p_ptr = _ci(u_ptr), _ci(p_ptr)
        self._ck(self.L.nsx_set_ranks(self._h, len(u_ptr) - 1, _i(u_ptr), _i(p_ptr)))

    def set_internal_layout(self, n_virtual_ranks, order=COLOUR, schur_max_rows=0):
        self._ck(self.L.nsx_set_internal_layout(self._h, int(n_virtual_ranks), int(order), int(schur_max_rows)))

    def layout_info(self):
        info = (C.c_int * 5)()
        self._ck(self.L.nsx_layout_info(self._h, info))
        return {"on": bool(info[0]), "ranks": int(info[1]), "schur_blocks": int(info[2]), "colours": int(info[3]), "colours_p": int(info[4])}

    def layout(self):
        """node_perm / pnode_perm (owned caller node -> internal node, global ids), u_ptr / p_ptr / schur_ptr (nsx_layout_get)"""
        info = self.layout_info()
        if self.world == 1:
            n2, n1 = self.n_u // self.dim, self.n_p
        else:
            n2 = int(self.view["gpu_u_ptr"][self.rank + 1] - self.view["gpu_u_ptr"][self.rank])
            n1 = int(self.view["gpu_p_ptr"][self.rank + 1] - self.view["gpu_p_ptr"][self.rank])
        out = {"node_perm": np.empty(n2, np.int32), "pnode_perm": np.empty(n1, np.int32), "u_ptr": np.empty(info["ranks"] + 1, np.int32),
               "p_ptr": np.empty(info["ranks"] + 1, np.int32), "schur_ptr": np.empty(info["schur_blocks"] + 1, np.int32)}
        self._ck(self.L.nsx_layout_get(self._h, *[_i(out[k]) for k in ("node_perm", "pnode_perm", "u_ptr", "p_ptr", "schur_ptr")]))
        out.update(info)
        return out

    def set_schur_blocks(self, p_ptr):
        p_ptr = _ci(p_ptr)
        self._ck(self.L.nsx_set_schur_blocks(self._h, len(p_ptr) - 1, _i(p_ptr)))

    # -- state ---------------------------------------------------------------------------------
    def set_solution(self, v):
        v = _cd(v)
        assert v.shape == (self.n,)
        self._ck(self.L.nsx_set_solution(self._h, _d(v)))

    def _get(self, fn):
        v = np.zeros(self.n)
        self._ck(fn(self._h, _d(v)))
        return v

    @property
    def solution_owned(self):
        return self._get(self.L.nsx_get_solution)

    @property
    def solution(self):
        return self._get(self.L.nsx_get_solution_ghosted)

    @property
    def rhs(self):
        return self._get(self.L.nsx_get_rhs)

    def set_rhs(self, v):
        v = _cd(v)
        self._ck(self.L.nsx_set_rhs(self._h, _d(v)))

    # -- hot path ------------------------------------------------------------------------------
    def assemble(self, flags=0):
        self._ck(self.L.nsx_assemble(self._h, flags))

    def assemble_time_step(self, flags=0):
        self._ck(self.L.nsx_assemble_time_step(self._h, flags))

    def add_rhs(self, dofs, vals):
        dofs, vals = _ci(dofs), _cd(vals)
        self._ck(self.L.nsx_add_rhs(self._h, len(dofs), _i(dofs), _d(vals)))

    def apply_boundary_values(self, dofs, vals):
        dofs, vals = _ci(dofs), _cd(vals)
        self._ck(self.L.nsx_apply_boundary_values(self._h, len(dofs), _i(dofs), _d(vals)))

    def solve_time_step(self, prec=YOSIDA, tol_abs=1e-4, inner_rtol=1e-2, maxiter=100000, inner_maxiter=100000,
                        check=True):
        st = Stats()
        rc = self.L.nsx_solve_time_step(self._h, prec, tol_abs, inner_rtol, maxiter, inner_maxiter, C.byref(st))
        if rc and (check or rc != -4):
            self._ck(rc)
        return st.as_dict()

    def prec_initialize(self, prec):
        self._ck(self.L.nsx_prec_initialize(self._h, prec))

    def prec_vmult(self, prec, src, inner_rtol=1e-2, inner_maxiter=100000, dst0=None):
        src = _cd(src)
        dst = np.zeros_like(src) if dst0 is None else _cd(dst0).copy()
        st = Stats()
        self._ck(self.L.nsx_prec_vmult(self._h, prec, inner_rtol, inner_maxiter, _d(dst), _d(src), C.byref(st)))
        return dst, st.as_dict()

    def system_vmult(self, src):
        src = _cd(src)
        dst = np.zeros_like(src)
        self._ck(self.L.nsx_system_vmult(self._h, _d(dst), _d(src)))
        return dst

    def ilu_apply(self, which, src):
        src = _cd(src)
        dst = np.empty_like(src)
        self._ck(self.L.nsx_ilu_apply(self._h, which, _d(dst), _d(src)))
        return dst

    def gram_schmidt_cycle(self, vectors, norm_guard=-1.0):
        """nsx_gram_schmidt_cycle: (orthonormalised vectors, coefficients [m][m], |w|^2 after each sweep)"""
        v = np.ascontiguousarray(vectors, dtype=np.float64).copy()
        m, n = v.shape
        coeffs, norms2 = np.zeros((m, m)), np.zeros(m)
        self._ck(self.L.nsx_gram_schmidt_cycle(self._h, n, m, _d(v), float(norm_guard), _d(coeffs), _d(norms2)))
        return v, coeffs, norms2

    # -- forces --------------------------------------------------------------------------------
    def set_force_faces(self, cells, lfaces, ftab):
        """obstacle faces as (global cell id, local face); in a multi-process run only the faces of owned cells are kept."""
        cells, lfaces = np.asarray(cells), np.asarray(lfaces)
        if self.world > 1:
            n_sub = self.dofs.n_subdomains // self.world
            mine = (self.dofs.mesh.subdomain[cells] // n_sub) == self.rank
            pos = {int(c): i for i, c in enumerate(self.view["cell_ids"])}
            cells = np.array([pos[int(c)] for c in cells[mine]], dtype=np.int32)
            lfaces = lfaces[mine]
        cells, lfaces = _ci(cells), _ci(lfaces)
        N2, dN2, N1, w = _cd(ftab.N2), _cd(ftab.dN2), _cd(ftab.N1), _cd(ftab.weights[:ftab.n_qf])
        self._ck(self.L.nsx_set_force_faces(self._h, len(cells), _i(cells), _i(lfaces), ftab.n_qf, _d(N2), _d(dN2), _d(N1), _d(w)))

    def compute_forces(self):
        d, l = C.c_double(), C.c_double()
        self._ck(self.L.nsx_compute_forces(self._h, C.byref(d), C.byref(l)))
        return d.value, l.value

    # -- export --------------------------------------------------------------------------------
    def export_block(self, which, block, graph=None):
        """values of matrix `which` in the reference's padded block-CSR graph (default: the front-end's)."""
        rowptr, colind = graph if graph is not None else self.dofs.reference_sparsity(3 if which == 4 else block)
        rowptr, colind = _ci(rowptr), _ci(colind)
        vals = np.empty(len(colind))
        self._ck(self.L.nsx_export_block(self._h, which, block, len(rowptr) - 1, _i(rowptr), _i(colind), _d(vals)))
        return vals

    def scalar_graph(self, which):
        nnz = C.c_int64()
        self._ck(self.L.nsx_scalar_graph_nnz(self._h, which, C.byref(nnz)))
        n = (self.n_u // self.dim) if which == 0 else self.n_p
        rp, ci = np.empty(n + 1, np.int32), np.empty(nnz.value, np.int32)
        self._ck(self.L.nsx_scalar_graph(self._h, which, _i(rp), _i(ci)))
        return rp, ci

    def schur(self):
        import scipy.sparse as sp
        rp, ci = self.scalar_graph(1)
        v = np.empty(len(ci))
        self._ck(self.L.nsx_schur_get(self._h, _i(rp), _i(ci), _d(v)))
        return sp.csr_matrix((v, ci, rp), shape=(self.n_p, self.n_p))

    def ilu(self, which):
        rp, ci = self.scalar_graph(which)
        v = np.empty(len(ci))
        self._ck(self.L.nsx_ilu_get(self._h, which, _d(v)))
        return rp, ci, v

    # -- measurement ---------------------------------------------------------------------------
    def profile(self, on=True):
        self._ck(self.L.nsx_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._ck(self.L.nsx_profile_reset(self._h))

    def profile_table(self):
        out = {}
        for i in range(self.L.nsx_profile_count(self._h)):
            name, n, ms, b = C.c_char_p(), C.c_int64(), C.c_double(), C.c_double()
            self.L.nsx_profile_get(self._h, i, C.byref(name), C.byref(n), C.byref(ms), C.byref(b))
            out[name.value.decode()] = {"launches": n.value, "total_ms": ms.value, "bytes_per_launch": b.value}
        return out
