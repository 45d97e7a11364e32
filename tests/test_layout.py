"""host/layout.hpp on the CPU (through the front-end's test hook nsxh_internal_layout): the sub-partition + node order that
libnsx builds behind nsx_set_internal_layout when the caller keeps deal.II's own numbering (reference NavierStokes3D.cpp:16-19,58-69).
The device-side use of the same code is covered by tests/test_gpu_layout.py."""
import numpy as np
import pytest

from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, PermutedDoFs, internal_layout, merge_ranks


@pytest.mark.parametrize("dim,level,n_virtual", [(3, 1, 4), (3, 2, 16), (2, 2, 8), (3, 3, 64), (2, 4, 40)])
@pytest.mark.parametrize("order", ["first_touch", "colour", "colour_all"])
def test_serial_first_touch_plus_layout_is_the_front_ends_numbering(dim, level, n_virtual, order):
    """one rank of the caller: the layout reproduces, node for node, partition(1, n) + the ordered DoF table of the front-end"""
    d0 = DoFs(Mesh.cylinder(dim, level), "first_touch")
    lay = internal_layout(d0, n_virtual, order, 40)
    d1 = DoFs(Mesh.cylinder(dim, level).partition(1, n_virtual), order)
    pd = PermutedDoFs(d0, lay["node_perm"], lay["pnode_perm"], lay["u_ptr"], lay["p_ptr"])
    assert (pd.cell_dofs == d1.cell_dofs).all()
    assert (lay["u_ptr"] == d1.owned_u_ptr).all() and (lay["p_ptr"] == d1.owned_p_ptr).all()
    assert (lay["schur_ptr"] == merge_ranks(d1.owned_p_ptr, 40)).all()
    assert lay["colours"] == (d1.n_colours, d1.n_colours_p)
    assert (pd.support_points == d1.support_points).all()


@pytest.mark.parametrize("dim,level,r_in,n_virtual", [(3, 2, 4, 24), (2, 3, 3, 10), (3, 2, 8, 8), (3, 1, 2, 64)])
def test_layout_refines_the_ranks_of_the_caller(dim, level, r_in, n_virtual):
    import scipy.sparse as sp
    d0 = DoFs(Mesh.cylinder(dim, level).partition(1, r_in), "first_touch")
    lay = internal_layout(d0, n_virtual, "colour", 30)
    n2, n1 = d0.n_nodes_p2, d0.n_nodes_p1
    assert sorted(lay["node_perm"]) == list(range(n2)) and sorted(lay["pnode_perm"]) == list(range(n1))
    for ptr, perm, mine in ((d0.owned_u_ptr, lay["node_perm"], lay["u_ptr"]), (d0.owned_p_ptr, lay["pnode_perm"], lay["p_ptr"])):
        assert mine[0] == 0 and mine[-1] == len(perm) and (np.diff(mine) >= 0).all()
        for r in range(r_in):
            img = perm[ptr[r]:ptr[r + 1]]
            assert len(img) == 0 or (img.min() == ptr[r] and img.max() == ptr[r + 1] - 1)   # a rank's nodes stay in that rank's range
            assert ptr[r] in mine                                                          # ... which is a union of virtual ranks
    assert all(b in lay["p_ptr"] for b in lay["schur_ptr"])                                 # Schur blocks: unions of virtual ranks
    # colour order inside a virtual rank: what the ILU(0) sees -- inside a rank the dependency depth of the block (longest chain of
    # lower neighbours) is at most the number of colours
    pd = PermutedDoFs(d0, lay["node_perm"], lay["pnode_perm"], lay["u_ptr"], lay["p_ptr"])
    nv = dim + 1
    np2 = nv + (3 if dim == 2 else 6)
    base = [(dim + 1) * a if a < nv else nv * (dim + 1) + dim * (a - nv) for a in range(np2)]
    c2 = np.asarray(pd.cell_dofs)[:, base] // dim
    rank_of = np.searchsorted(lay["u_ptr"], np.arange(n2), side="right") - 1
    depth = np.zeros(n2, dtype=np.int64)
    rows = np.repeat(c2, np2, axis=1).ravel()
    cols = np.tile(c2, (1, np2)).ravel()
    A = sp.csr_matrix((np.ones(len(rows), dtype=np.int8), (rows, cols)), shape=(n2, n2))
    A.sum_duplicates()
    A.sort_indices()
    for i in range(n2):
        js = A.indices[A.indptr[i]:A.indptr[i + 1]]
        js = js[(js < i) & (rank_of[js] == rank_of[i])]
        depth[i] = depth[js].max() + 1 if len(js) else 0
    assert depth.max() + 1 <= lay["colours"][0]


def test_layout_rejects_bad_requests():
    d0 = DoFs(Mesh.cylinder(2, 1), "first_touch")
    with pytest.raises(ValueError):
        internal_layout(d0, 0)
