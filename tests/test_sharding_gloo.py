"""
world_size-2 gloo test of the N>1 path (solve_sharded): the batch scatter, the per-rank shard and the
eigenvalue gather, on CPU.  The device solver is replaced by the oracle through ``solver_factory``
(this is a test: it checks the sharding / collectives, not the arithmetic).
"""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, queue):
    import torch.distributed as dist

    from oracle import enm_oracle as orc
    from springcraft_amd.batch import solve_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def factory(n_atoms, batch):
        def run(coords):
            ws = []
            for c in coords:
                h, _ = orc.compute_hessian(c, orc.invariant_ff(8.0))
                ws.append(orc.eigen(h)[0])
            return np.array(ws), None
        return run

    coords = None
    if rank == 0:
        coords = np.stack([orc.synthetic_coord(30, s, 12.0) for s in range(n_items)])
    w, _ = solve_sharded(coords, None, dim=3, solver_factory=factory)
    if rank == 0:
        queue.put(w)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 2, 1])
def test_solve_sharded_two_ranks(n_items):
    import torch.multiprocessing as mp

    from oracle import enm_oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    w = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert w.shape == (n_items, 90)
    for s in range(n_items):
        h, _ = orc.compute_hessian(orc.synthetic_coord(30, s, 12.0), orc.invariant_ff(8.0))
        assert np.allclose(w[s], orc.eigen(h)[0], atol=1e-10)
