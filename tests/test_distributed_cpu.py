"""CPU suite, world_size 2 over gloo: the host logic of the N > 1 path (rank views, halo plans, owned-range reductions).
The device kernels cannot run here; what is checked is exactly what the GPU ranks will be told to exchange."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, dim, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from navierstokes_project_nm4pde_amd.frontend import DoFs, Mesh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mesh = Mesh.cylinder(dim, 1).partition(world, 3)
    d = DoFs(mesh)
    v = d.rank_view(rank, world)
    nv, nl = dim + 1, (3 if dim == 2 else 6)
    vel = [(dim + 1) * a for a in range(nv)] + [nv * (dim + 1) + dim * l for l in range(nl)]
    u0, u1 = v["gpu_u_ptr"][rank], v["gpu_u_ptr"][rank + 1]
    nodes = np.unique(v["cell_dofs"][:, vel] // dim)
    ghosts = nodes[(nodes < u0) | (nodes >= u1)]
    # a global field only the owner knows: f(node) = sin(node); ghosts must arrive through the plan
    f = lambda n: np.sin(0.37 * n.astype(np.float64))
    recv = {}
    reqs, keep = [], []
    for k, nb in enumerate(v["neighbors"]):
        gk = ghosts[(ghosts >= v["gpu_u_ptr"][nb]) & (ghosts < v["gpu_u_ptr"][nb + 1])]
        if len(gk):
            t = torch.zeros(len(gk), dtype=torch.float64)
            recv[int(nb)] = (gk, t)
            reqs.append(dist.irecv(t, src=int(nb)))
        send_nodes = v["send_u_nodes"][v["send_u_ptr"][k]:v["send_u_ptr"][k + 1]]
        assert ((send_nodes >= u0) & (send_nodes < u1)).all() and (np.diff(send_nodes) > 0).all()
        if len(send_nodes):
            t = torch.from_numpy(f(send_nodes))
            keep.append(t)
            reqs.append(dist.isend(t, dst=int(nb)))
    for q in reqs:
        q.wait()
    got = sum(len(g) for g, _ in recv.values())
    assert got == len(ghosts)                      # every ghost has exactly one neighbour to come from
    for nb, (gk, t) in recv.items():
        assert np.array_equal(t.numpy(), f(gk))    # in the order the receiver stores them (ascending global id)
    # layer structure: layer-1 cells are exactly those touching an owned node, and they come first
    own = ((v["cell_dofs"][:, vel] // dim >= u0) & (v["cell_dofs"][:, vel] // dim < u1)).any(1)
    assert own[:v["n_cells_layer1"]].all() and not own[v["n_cells_layer1"]:].any()
    # owned-range reduction == global reduction (what the distributed dot product relies on)
    x = np.cos(0.11 * np.arange(d.n_nodes_p2))
    part = torch.tensor([float(x[u0:u1] @ x[u0:u1])], dtype=torch.float64)
    dist.all_reduce(part)
    assert abs(part.item() - float(x @ x)) < 1e-12 * float(x @ x)
    # every cell is layer-1 for at least one rank, and the virtual-rank ranges tile this rank's range
    flag = torch.zeros(d.n_cells, dtype=torch.int32)
    flag[torch.from_numpy(v["cell_ids"][:v["n_cells_layer1"]].astype(np.int64))] = 1
    dist.all_reduce(flag)
    assert int(flag.min()) >= 1
    assert v["rank_u_ptr"][0] == u0 and v["rank_u_ptr"][-1] == u1 and (np.diff(v["rank_u_ptr"]) >= 0).all()
    open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    dist.destroy_process_group()


@pytest.mark.parametrize("dim", [2, 3])
def test_rank_views_and_halo_plans_over_gloo(tmp_path, dim):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, 29611 + dim, dim, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / ("ok%d" % r)) for r in range(world))
