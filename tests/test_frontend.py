"""CPU suite: the host front-end (mesh generator, DoF numbering contract, sparsity, partitioner, .msh reader)."""
import numpy as np
import pytest

from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, Tables


def _volumes(m):
    X = m.vertices[m.cells]
    return np.linalg.det(X[:, 1:] - X[:, :1]) / (2 if m.dim == 2 else 6)


@pytest.mark.parametrize("dim,level", [(2, 1), (2, 3), (3, 1), (3, 2)])
def test_cylinder_mesh_is_valid(dim, level):
    m = Mesh.cylinder(dim, level)
    vol = _volumes(m)
    assert vol.min() > 0                                           # positively oriented, no degenerate cells
    exact = (2.2 * 0.41 - np.pi * 0.05 ** 2) if dim == 2 else (2.5 * 0.41 - np.pi * 0.05 ** 2) * 0.41
    assert abs(vol.sum() - exact) < 2e-3 * exact                   # polygonal approximation of the circle only
    assert set(np.unique(m.bface_ids)) == {0, 1, 2, 3}             # inlet, outlet, walls, obstacle (mesh/*.geo)
    # conforming: every interior face is shared by exactly two cells
    from collections import Counter
    fid = [(0, 1), (1, 2), (2, 0)] if dim == 2 else [(0, 1, 2), (1, 0, 3), (0, 2, 3), (2, 1, 3)]
    cnt = Counter(tuple(sorted(c[list(f)])) for c in m.cells.tolist() for f in [None] if False)
    faces = np.sort(np.concatenate([m.cells[:, list(f)] for f in fid]), axis=1)
    _, counts = np.unique(faces, axis=0, return_counts=True)
    assert counts.max() == 2 and (counts == 1).sum() == len(m.bface_ids)
    # obstacle faces lie on the circle, inlet faces on x = 0
    xc = 0.2 if dim == 2 else 0.5
    ob = m.vertices[m.bfaces[m.bface_ids == 3].ravel()]
    assert np.allclose(np.hypot(ob[:, 0] - xc, ob[:, 1] - 0.2), 0.05, atol=1e-12)
    assert np.allclose(m.vertices[m.bfaces[m.bface_ids == 0].ravel()][:, 0], 0.0)


@pytest.mark.parametrize("dim", [2, 3])
def test_dof_numbering_contract(dim):
    m = Mesh.cylinder(dim, 1).partition(1, 3)
    d = DoFs(m)
    nv, nl = dim + 1, 3 if dim == 2 else 6
    assert d.dofs_per_cell == nv * (dim + 1) + nl * dim == (15 if dim == 2 else 34)
    cd = d.cell_dofs
    # FESystem local order: vertex v -> (dim+1)v + {u_c, p}; line l -> nv(dim+1) + dim l + c; components consecutive
    for v in range(nv):
        base = cd[:, (dim + 1) * v]
        assert (base % dim == 0).all() and (base < d.n_u).all()
        for c in range(1, dim):
            assert (cd[:, (dim + 1) * v + c] == base + c).all()
        assert (cd[:, (dim + 1) * v + dim] >= d.n_u).all()
    for l in range(nl):
        base = cd[:, nv * (dim + 1) + dim * l]
        assert (base % dim == 0).all()
    assert len(np.unique(cd)) == d.n_dofs == d.n_u + d.n_p          # every dof is used, numbering is gap-free
    # Euler: P2 nodes = vertices + edges
    edges = set()
    pairs = [(0, 1), (1, 2), (2, 0)] if dim == 2 else [(0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3)]
    for a, b in pairs:
        e = np.sort(m.cells[:, [a, b]], axis=1)
        edges.update(map(tuple, e.tolist()))
    assert d.n_nodes_p2 == len(m.vertices) + len(edges) and d.n_nodes_p1 == len(m.vertices)
    # support points: vertex dofs sit on vertices, line dofs on edge midpoints
    X = m.vertices[m.cells]
    for l, (a, b) in enumerate(pairs):
        mid = 0.5 * (X[:, a] + X[:, b])
        assert np.allclose(d.support_points[cd[:, nv * (dim + 1) + dim * l]], mid)
    assert np.allclose(d.support_points[cd[:, dim]], X[:, 0])
    # subdomain-major numbering (what an MPI run with one rank per subdomain produces): owner ranges are contiguous
    assert (np.diff(d.node_owner) >= 0).all() and (np.diff(d.pnode_owner) >= 0).all()
    assert d.owned_u_ptr[-1] == d.n_nodes_p2 and d.owned_p_ptr[-1] == d.n_nodes_p1
    for s in range(d.n_subdomains):
        assert (d.node_owner[d.owned_u_ptr[s]:d.owned_u_ptr[s + 1]] == s).all()
    # interface entities belong to the lowest subdomain touching them
    nodes = cd[:, [(dim + 1) * v for v in range(nv)] + [nv * (dim + 1) + dim * l for l in range(nl)]] // dim
    low = np.full(d.n_nodes_p2, 10 ** 9)
    np.minimum.at(low, nodes.ravel(), np.repeat(m.subdomain, nodes.shape[1]))
    assert (low == d.node_owner).all()


def test_single_rank_numbering_is_first_touch():
    m = Mesh.box(2, [2, 1])
    d = DoFs(m)
    seen, nxt = {}, 0
    for cell in d.cell_dofs:
        order = [cell[0], cell[3], cell[6], cell[9], cell[11], cell[13]]     # vertices then lines (x-components)
        for dof in order:
            node = dof // 2
            if node not in seen:
                assert node == nxt
                seen[node] = True
                nxt += 1


@pytest.mark.parametrize("dim", [2, 3])
def test_colour_ordering_is_a_rank_local_renumbering(dim):
    m = Mesh.cylinder(dim, 1).partition(1, 4)
    d0, d1 = DoFs(m), DoFs(m, "colour")
    assert d0.n_colours == 0 and d1.n_colours >= (6 if dim == 2 else 10)   # a cell's P2 nodes are mutually adjacent
    nv, nl = dim + 1, 3 if dim == 2 else 6
    cols = [(dim + 1) * v for v in range(nv)] + [nv * (dim + 1) + dim * l for l in range(nl)]
    n0, n1 = d0.cell_dofs[:, cols] // dim, d1.cell_dofs[:, cols] // dim
    # a permutation of the P2 nodes that keeps every node inside its owner's range; pressure untouched
    perm = np.full(d0.n_nodes_p2, -1)
    perm[n0.ravel()] = n1.ravel()
    assert sorted(perm.tolist()) == list(range(d0.n_nodes_p2))
    assert (d1.node_owner == d0.node_owner).all() and (d1.node_owner[perm] == d0.node_owner).all()
    assert (d1.owned_u_ptr == d0.owned_u_ptr).all() and (d1.owned_p_ptr == d0.owned_p_ptr).all()
    pc = [(dim + 1) * v + dim for v in range(nv)]
    assert (d1.cell_dofs[:, pc] == d0.cell_dofs[:, pc]).all()
    assert np.allclose(d1.support_points[d1.cell_dofs], d0.support_points[d0.cell_dofs])
    # inside a rank the nodes come colour by colour: a run of consecutive nodes without mutual adjacency.  Hence the
    # ILU(0) dependency depth of a rank block (longest chain i1 < i2 < ... of adjacent nodes) is at most n_colours.
    import scipy.sparse as sp
    rows = np.repeat(n1, n1.shape[1], axis=1).ravel()
    colsn = np.tile(n1, (1, n1.shape[1])).ravel()
    A = sp.csr_matrix((np.ones(len(rows)), (rows, colsn)), shape=(d1.n_nodes_p2,) * 2)
    A.sum_duplicates()
    depth = np.zeros(d1.n_nodes_p2, dtype=int)
    for i in range(d1.n_nodes_p2):
        nb = A.indices[A.indptr[i]:A.indptr[i + 1]]
        nb = nb[(nb < i) & (d1.node_owner[nb] == d1.node_owner[i])]
        depth[i] = depth[nb].max() + 1 if len(nb) else 0
    assert depth.max() + 1 <= d1.n_colours
    assert d1.boundary_dofs(3).size == d0.boundary_dofs(3).size
    assert np.allclose(np.sort(d1.support_points[d1.boundary_dofs(3)], axis=0), np.sort(d0.support_points[d0.boundary_dofs(3)], axis=0))


@pytest.mark.parametrize("dim", [2, 3])
def test_reference_sparsity_matches_cell_couplings(dim):
    import scipy.sparse as sp
    m = Mesh.cylinder(dim, 1)
    d = DoFs(m)
    n = d.n_dofs
    rows = np.repeat(d.cell_dofs, d.dofs_per_cell, axis=1).ravel()
    cols = np.tile(d.cell_dofs, (1, d.dofs_per_cell)).ravel()
    full = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    full.data[:] = 1
    nu = d.n_u
    for blk, (r, c) in enumerate([((0, nu), (0, nu)), ((0, nu), (nu, n)), ((nu, n), (0, nu)), ((nu, n), (nu, n))]):
        rp, ci = d.reference_sparsity(blk)
        ref = full[r[0]:r[1], c[0]:c[1]].tocsr()
        ref.sort_indices()
        assert (ref.indptr == rp).all() and (ref.indices == ci).all()


def test_partition_is_balanced_and_complete():
    m = Mesh.cylinder(3, 2).partition(2, 8)
    sizes = np.bincount(m.subdomain, minlength=16)
    assert sizes.min() > 0 and sizes.max() - sizes.min() <= 2
    # two-level: GPU part = subdomain // 8; parts are balanced too
    parts = np.bincount(m.subdomain // 8)
    assert abs(int(parts[0]) - int(parts[1])) <= 1


def test_boundary_dofs_are_whole_nodes():
    m = Mesh.cylinder(3, 1)
    d = DoFs(m)
    for bid in (0, 2, 3):
        bd = d.boundary_dofs(bid)
        assert len(bd) % 3 == 0 and (bd[0::3] % 3 == 0).all() and (bd[1::3] == bd[0::3] + 1).all()
    x_in = d.support_points[d.boundary_dofs(0)]
    assert np.allclose(x_in[:, 0], 0.0)


def test_msh_reader_round_trip(tmp_path):
    m = Mesh.cylinder(3, 1)
    p = tmp_path / "cyl.msh"
    with open(p, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(m.vertices))
        for i, v in enumerate(m.vertices):
            f.write("%d %.17g %.17g %.17g\n" % (i + 1, *v))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(m.bfaces) + len(m.cells)))
        k = 1
        for fv, fid in zip(m.bfaces, m.bface_ids):
            f.write("%d 2 2 %d %d %s\n" % (k, fid, fid, " ".join(str(v + 1) for v in fv)))
            k += 1
        for c in m.cells:
            f.write("%d 4 2 10 1 %s\n" % (k, " ".join(str(v + 1) for v in c)))
            k += 1
        f.write("$EndElements\n")
    r = Mesh.read_msh(p)
    assert (r.cells == m.cells).all() and np.allclose(r.vertices, m.vertices)
    assert sorted(map(tuple, np.c_[np.sort(r.bfaces, 1), r.bface_ids].tolist())) == \
        sorted(map(tuple, np.c_[np.sort(m.bfaces, 1), m.bface_ids].tolist()))


def test_face_tables():
    t = Tables(3, Tables.FACE)
    assert t.n_q == 4 * t.n_qf and np.allclose(t.weights.reshape(4, -1).sum(1), 1.0)
    # points of face f lie on face f of the reference tetrahedron
    lam = np.c_[1 - t.points.sum(1), t.points]
    opp = [3, 2, 1, 0]   # vertex opposite to deal.II face f = {012, 103, 023, 213}
    for f in range(4):
        assert np.allclose(lam[f * t.n_qf:(f + 1) * t.n_qf, opp[f]], 0.0, atol=1e-15)


@pytest.mark.parametrize("dim", [2, 3])
def test_vtu_output_matches_the_field(dim, tmp_path):
    """NavierStokes::output (NavierStokes3D.cpp:643-683): one linear patch per cell, velocity / pressure / partitioning."""
    import xml.etree.ElementTree as ET
    m = Mesh.cylinder(dim, 1).partition(1, 3)
    d = DoFs(m)
    X = d.support_points
    sol = np.zeros(d.n_dofs)
    for c in range(dim):
        sol[c:d.n_u:dim] = (c + 1) * X[c:d.n_u:dim, 0] - 0.5 * X[c:d.n_u:dim, 1]
    sol[d.n_u:] = 3.0 + X[d.n_u:, 0] * X[d.n_u:, 1]
    d.write_vtu(sol, tmp_path / "out" / "nested", "output-navier-stokes-%dD" % dim, 20)
    piece = tmp_path / "out" / "nested" / ("output-navier-stokes-%dD_20.0.vtu" % dim)
    record = tmp_path / "out" / "nested" / ("output-navier-stokes-%dD_20.pvtu" % dim)
    root = ET.parse(piece).getroot()
    pc = root.find("UnstructuredGrid/Piece")
    nv = dim + 1
    assert int(pc.get("NumberOfCells")) == d.n_cells and int(pc.get("NumberOfPoints")) == d.n_cells * nv
    arr = {a.get("Name"): np.array(a.text.split(), dtype=float) for a in pc.iter("DataArray") if a.get("Name")}
    pts = np.array(pc.find("Points/DataArray").text.split(), dtype=float).reshape(-1, 3)
    assert np.allclose(pts[:, :dim].reshape(d.n_cells, nv, dim), d.cell_coords) and (dim == 3 or (pts[:, 2] == 0).all())
    assert (arr["types"] == (10 if dim == 3 else 5)).all() and (arr["offsets"] == nv * np.arange(1, d.n_cells + 1)).all()
    assert (arr["connectivity"] == np.arange(d.n_cells * nv)).all()
    vel = arr["velocity"].reshape(-1, 3)
    for c in range(dim):
        assert np.allclose(vel[:, c], (c + 1) * pts[:, 0] - 0.5 * pts[:, 1], atol=1e-14)
    assert np.allclose(arr["pressure"], 3.0 + pts[:, 0] * pts[:, 1], atol=1e-14)
    assert (arr["partitioning"].reshape(d.n_cells, nv) == m.subdomain[:, None]).all()
    rec = ET.parse(record).getroot()
    assert rec.find("PUnstructuredGrid/Piece").get("Source") == piece.name
    assert {a.get("Name") for a in rec.iter("PDataArray") if a.get("Name")} == {"velocity", "pressure", "partitioning"}


@pytest.mark.parametrize("dim", [2, 3])
def test_pressure_difference_is_the_p1_interpolant(dim):
    """compute_pressure_difference (NavierStokes3D.cpp:849-923): exact for a linear pressure, 0 for points outside."""
    from navierstokes_project_nm4pde_amd.problem import pressure_difference
    m = Mesh.cylinder(dim, 1)
    d = DoFs(m)
    X = d.support_points[d.n_u:]
    sol = np.zeros(d.n_dofs)
    g = np.array([2.0, -1.0, 0.5][:dim])
    sol[d.n_u:] = 7.0 + X @ g
    a, b = np.array([0.45, 0.2, 0.205][:dim]), np.array([0.55, 0.2, 0.205][:dim])
    diff, found = d.pressure_difference(sol, a, b)
    assert found == 2 and abs(diff - (a - b) @ g) < 1e-12
    assert abs(diff - pressure_difference(m, d, sol, a, b)) < 1e-12      # the numpy version used by the convergence driver
    diff, found = d.pressure_difference(sol, a, np.array([9.0, 9.0, 9.0][:dim]))
    assert found == 1 and abs(diff - (7.0 + a @ g)) < 1e-12              # a point nobody holds contributes 0 (MPI_MAX of zeros)


def test_partition_owned_balances_the_ilu_blocks():
    """nsxh_mesh_partition_owned: equal numbers of OWNED P2 nodes per subdomain under the lowest-id ownership rule (the ILU(0)
    block sizes), where the cell-balanced bisection gives the low ids up to twice the mean."""
    sizes = {}
    for balance in ("cells", "owned"):
        m = Mesh.cylinder(3, 2).partition(2, 32, balance=balance)
        assert set(np.unique(m.subdomain)) == set(range(64))
        d = DoFs(m)
        sizes[balance] = np.diff(d.owned_u_ptr)
        assert sizes[balance].sum() == d.n_u // 3 and sizes[balance].min() > 0
    assert sizes["owned"].max() < 1.2 * sizes["owned"].mean()
    assert sizes["cells"].max() > 1.4 * sizes["cells"].mean()
    assert sizes["owned"].max() < sizes["cells"].max()


@pytest.mark.parametrize("dim", [2, 3])
def test_colour_all_also_orders_the_pressure_nodes_by_colour_of_the_schur_graph(dim):
    """NSXH_ORDER_COLOUR_ALL: velocity nodes as NSXH_ORDER_COLOUR, pressure nodes of every rank sorted by a colouring of
    the graph of B D^-1 B^T (two P1 nodes adjacent when some P2 node shares a cell with each).  Rank-local permutation;
    the ILU(0) of a rank's Schur block is then at most n_colours_p levels deep."""
    import scipy.sparse as sp
    m = Mesh.cylinder(dim, 1).partition(1, 3)
    d1, d2 = DoFs(m, "colour"), DoFs(m, "colour_all")
    assert d1.n_colours_p == 0 and d2.n_colours_p >= dim + 1 and d2.n_colours == d1.n_colours
    nv, nl = dim + 1, 3 if dim == 2 else 6
    ucols = [(dim + 1) * v + k for v in range(nv) for k in range(dim)] + [nv * (dim + 1) + dim * l + k for l in range(nl) for k in range(dim)]
    pc = [(dim + 1) * v + dim for v in range(nv)]
    assert (d2.cell_dofs[:, ucols] == d1.cell_dofs[:, ucols]).all()               # velocity numbering as "colour"
    p1, p2 = d1.cell_dofs[:, pc] - d1.n_u, d2.cell_dofs[:, pc] - d2.n_u
    perm = np.full(d1.n_p, -1)
    perm[p1.ravel()] = p2.ravel()
    assert sorted(perm.tolist()) == list(range(d1.n_p))
    assert (d2.pnode_owner == d1.pnode_owner).all() and (d2.pnode_owner[perm] == d1.pnode_owner).all()
    assert (d2.owned_p_ptr == d1.owned_p_ptr).all() and (d2.owned_u_ptr == d1.owned_u_ptr).all()
    assert np.allclose(d2.support_points[d2.cell_dofs], d1.support_points[d1.cell_dofs])
    # pattern of B (P1 x P2 nodes) from the cells, S = B B^T, depth of the in-rank lower triangle
    cols2 = [(dim + 1) * v for v in range(nv)] + [nv * (dim + 1) + dim * l for l in range(nl)]
    n2 = d2.cell_dofs[:, cols2] // dim
    r = np.repeat(p2, n2.shape[1], axis=1).ravel()
    c = np.tile(n2, (1, p2.shape[1])).ravel()
    B = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(d2.n_p, d2.n_nodes_p2))
    S = (B @ B.T).tocsr()
    depth = np.zeros(d2.n_p, dtype=int)
    for i in range(d2.n_p):
        nb = S.indices[S.indptr[i]:S.indptr[i + 1]]
        nb = nb[(nb < i) & (d2.pnode_owner[nb] == d2.pnode_owner[i])]
        depth[i] = depth[nb].max() + 1 if len(nb) else 0
    assert depth.max() + 1 <= d2.n_colours_p


def test_merge_ranks_builds_blocks_of_consecutive_ranks_within_a_row_limit():
    """Schur ILU blocks of bench.py: consecutive ranks merged up to a row limit (nsx_set_schur_blocks takes unions of ranks)."""
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh, merge_ranks
    d = DoFs(Mesh.cylinder(3, 2).partition(1, 64), "colour")
    ptr = np.asarray(d.owned_p_ptr)
    for limit in (1, 20, 96, 10 ** 9):
        t = merge_ranks(ptr, limit)
        assert t[0] == ptr[0] and t[-1] == ptr[-1] and (np.diff(t) > 0).all()
        assert np.isin(t, ptr).all()                                    # block boundaries are rank boundaries
        sizes, ranks = np.diff(t), np.diff(ptr)
        assert (sizes <= max(limit, ranks.max())).all()                 # a rank larger than the limit stays a block of its own
        if limit >= ptr[-1] - ptr[0]:
            assert len(t) == 2
    t = merge_ranks(ptr, 96)
    # greedy: a block could not have taken the next rank as well
    nxt = np.searchsorted(ptr, t[1:-1])
    assert ((ptr[nxt + 1] - t[:-2]) > 96).all()
    empty = np.array([0, 0, 5, 5, 9], dtype=np.int32)                   # empty ranks do not produce empty blocks
    assert merge_ranks(empty, 4).tolist() == [0, 5, 9]
