"""CPU tests of the schedule of the packed block-ILU(0) triangular solve (host/ilu_stream.hpp: the code libnsx runs at set-up).
The stream is replayed on the host exactly as the device kernel consumes it -- tick by tick, the LDS reads of a tick taken before
the writes of the tick in front of it land -- and compared with plain sequential sweeps per rank block
(Ifpack_ILU::ApplyInverse, overlap 0: what TrilinosWrappers::PreconditionILU::vmult computes on every MPI rank, reference
Preconditioners.hpp:215-216,382,405).  A tick that read a row before that row's last tick had been written would give a
different vector."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import Problem


def scalar_graph(dofs, which):
    """which = 0: scalar P2 x P2 pattern of block (0,0); 1: pattern of B B^T on the P1 nodes (the Schur complement's)"""
    d = dofs
    first = d.cell_dofs[0]
    sel_u = [k for k in range(len(first)) if first[k] < d.n_u and first[k] % d.dim == 0]
    sel_p = [k for k in range(len(first)) if first[k] >= d.n_u]
    n2 = d.cell_dofs[:, sel_u] // d.dim
    n1 = d.cell_dofs[:, sel_p] - d.n_u

    def coupling(r, c, nr, nc):
        rr, cc = np.repeat(r, c.shape[1], axis=1).ravel(), np.tile(c, (1, r.shape[1])).ravel()
        m = sp.coo_matrix((np.ones(len(rr)), (rr, cc)), shape=(nr, nc)).tocsr()
        m.data[:] = 1.0
        return m

    if which == 0:
        a = coupling(n2, n2, d.n_nodes_p2, d.n_nodes_p2)
    else:
        b = coupling(n1, n2, d.n_nodes_p1, d.n_nodes_p2)
        a = (b @ b.T).tocsr()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32)


def reference_apply(rp, ci, bptr, lu, b, ncomp):
    """x = U^-1 D^-1 L^-1 b block by block with SciPy's triangular solves; lu in Ifpack's storage (L, 1/d, U/d)."""
    n = len(rp) - 1
    m = sp.csr_matrix((lu, ci, rp), shape=(n, n))
    x = np.empty((n, ncomp))
    bb = b.reshape(n, ncomp)
    for k in range(len(bptr) - 1):
        r0, r1 = bptr[k], bptr[k + 1]
        if r1 == r0:
            continue
        blk = m[r0:r1, r0:r1].tocsr()
        low = (sp.tril(blk, -1) + sp.identity(r1 - r0)).tocsr()
        upp = (sp.triu(blk, 1) + sp.identity(r1 - r0)).tocsr()
        y = spla.spsolve_triangular(low, bb[r0:r1], lower=True)
        y = y * blk.diagonal()[:, None]
        x[r0:r1] = spla.spsolve_triangular(upp, y, lower=False)
    return x.ravel()


CASES = [("cylinder", 3, 1, 4, "colour"), ("cylinder", 3, 1, 1, "first_touch"), ("cylinder", 3, 2, 24, "colour"),
         ("cylinder", 2, 2, 6, "first_touch"), ("cube", 3, 3, 5, "colour")]


@pytest.mark.parametrize("kind,dim,level,n_sub,ordering", CASES)
@pytest.mark.parametrize("which", [0, 1])
def test_stream_replay_equals_sequential_block_sweeps(kind, dim, level, n_sub, ordering, which):
    from navierstokes_project_nm4pde_amd.frontend import ilu_stream_apply, ilu_stream_stats
    p = Problem(kind, dim, level, n_sub=n_sub, ordering=ordering)
    rp, ci = scalar_graph(p.dofs, which)
    bptr = np.asarray(p.dofs.owned_u_ptr if which == 0 else p.dofs.owned_p_ptr, dtype=np.int32)
    n = len(rp) - 1
    rng = np.random.default_rng(11 + which)
    lu = 0.3 * rng.standard_normal(len(ci)) / np.sqrt(np.diff(rp).mean())
    rows = np.repeat(np.arange(n), np.diff(rp))
    lu[rows == ci] = 1.0 / (1.5 + rng.random(n))         # diagonal slot holds 1/d
    blk = np.searchsorted(bptr, np.arange(n), side="right") - 1
    in_block = blk[rows] == blk[ci]
    ran = 0
    for ncomp in ((1, dim) if which == 0 else (1,)):
        b = rng.standard_normal(n * ncomp)
        ref = reference_apply(rp, ci, bptr, lu, b, ncomp)
        for bpw, gap, ept in ((1, 2, 1), (2, 2, 2), (5, 2, 3), (3, 3, 4), (2, 2, 4)):
            try:
                st = ilu_stream_stats(rp, ci, bptr, bpw, ncomp, gap, ept)
            except ValueError as e:             # the rows of a wave do not fit 16-bit LDS addresses: refused, the device library
                assert "-3" in str(e)           # then uses its workgroup-per-block kernel (test_a_wave_that_cannot_...)
                assert (np.diff(bptr).max() * bpw + 64) * 8 * ncomp > 65536
                continue
            ran += 1
            assert st["used_slots"] == int((in_block & (rows != ci)).sum())       # every in-block off-diagonal entry exactly once
            assert st["in_block_nnz"] == int(in_block.sum())
            assert st["waves"] == -(-(len(bptr) - 1) // bpw) and st["slabs"] * 64 * ept >= st["used_slots"]
            x = ilu_stream_apply(rp, ci, bptr, lu, b, ncomp, bpw, gap, ept)
            assert np.abs(x - ref).max() < 1e-12 * max(1.0, np.abs(ref).max()), (ncomp, bpw, gap, ept)
    assert ran >= 2


def test_stream_of_a_diagonal_block_table_and_of_empty_blocks():
    """edge cases of the block table: one row per block (no in-block off-diagonal entry at all: zero slabs) and empty blocks"""
    from navierstokes_project_nm4pde_amd.frontend import ilu_stream_apply, ilu_stream_stats
    p = Problem("cylinder", 2, 1)
    rp, ci = scalar_graph(p.dofs, 0)
    n = len(rp) - 1
    rng = np.random.default_rng(5)
    lu = rng.standard_normal(len(ci))
    rows = np.repeat(np.arange(n), np.diff(rp))
    lu[rows == ci] = 2.0
    b = rng.standard_normal(n)
    one = np.arange(n + 1, dtype=np.int32)
    st = ilu_stream_stats(rp, ci, one, 8, 1)
    assert st["slabs"] == 0 and st["used_slots"] == 0 and st["in_block_nnz"] == n
    assert np.array_equal(ilu_stream_apply(rp, ci, one, lu, b, 1, 8), 2.0 * b)
    ragged = np.array([0, 0, n // 3, n // 3, n // 3, n, n], dtype=np.int32)   # empty blocks at both ends and in the middle
    x = ilu_stream_apply(rp, ci, ragged, lu * 0.05 + (rows == ci) * 0.9, b, 1, 2)
    ref = reference_apply(rp, ci, ragged, lu * 0.05 + (rows == ci) * 0.9, b, 1)
    assert np.abs(x - ref).max() < 1e-12 * np.abs(ref).max()


def test_a_wave_that_cannot_address_its_rows_is_refused():
    """16-bit LDS byte addresses: (rows of the wave + 64 scratch rows) * 8 * ncomp must stay below 64 KiB"""
    from navierstokes_project_nm4pde_amd.frontend import ilu_stream_stats
    p = Problem("cylinder", 3, 2)                        # one block of ~5 500 P2 nodes
    rp, ci = scalar_graph(p.dofs, 0)
    n = len(rp) - 1
    assert n * 24 > 65536
    with pytest.raises(ValueError, match="-3"):
        ilu_stream_stats(rp, ci, np.array([0, n], dtype=np.int32), 1, 3)
    assert ilu_stream_stats(rp, ci, np.unique(np.append(np.arange(0, n, 100), n)).astype(np.int32), 4, 3)["slabs"] > 0


def test_stream_keeps_the_lanes_busy_and_the_waves_balanced():
    """the point of the lane-owner stream: on a colour-ordered layout of ~85-row blocks with eight blocks per wave more than 70 %
    of the slots carry an entry (the round-2 lane-group stream: 49 %), and no wave is much longer than the mean although the
    first subdomains own twice the mean number of rows (blocks are dealt longest first)"""
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, ilu_stream_stats
    mesh = Mesh.cylinder(3, 3).partition(1, 256)
    d = DoFs(mesh, "colour")
    rp, ci = scalar_graph(d, 0)
    s1, s8 = ilu_stream_stats(rp, ci, d.owned_u_ptr, 1, 3), ilu_stream_stats(rp, ci, d.owned_u_ptr, 8, 3)
    assert s8["fill"] > 0.70 and s8["fill"] > 2 * s1["fill"] and s8["slabs"] < s1["slabs"]
    assert s8["max_wave_slabs"] < 1.35 * s8["slabs"] / s8["waves"]
